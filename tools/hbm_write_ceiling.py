import torch, time
dev = torch.device("cuda", 0)
x = torch.empty(1 << 32, dtype=torch.uint8, device=dev)
y = torch.empty(1 << 32, dtype=torch.uint8, device=dev)
for name, fn, bytes_ in (("fill (write only)", lambda: x.fill_(7), x.numel()), ("zero_ (memset)", lambda: x.zero_(), x.numel()), ("copy (read+write)", lambda: y.copy_(x), 2 * x.numel())):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:20s} {bytes_ / ms / 1e9:8.2f} TB/s")
