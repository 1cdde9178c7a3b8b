"""Does RCCL take two ranks on ONE GPU?  (It would let the RCCL transport run with a real peer on the one-GPU box.)
torch.distributed 'nccl' backend, two processes, both on cuda:0, one all_reduce.  Prints the outcome; never hangs longer
than 60 s."""
import datetime
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=40), device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok -> {t.tolist()}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:400]}", flush=True)
    os._exit(0)


if __name__ == "__main__":
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ps = [mp.get_context("spawn").Process(target=worker, args=(r, port)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            print("a rank had to be killed after 60 s")
