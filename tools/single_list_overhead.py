"""Cost of the stepping API (imt_itree_batch_*) itself: one rank, no collectives, 2^16 insertions per
batch, against the fused path.  The difference is what the single-list multi-GPU mode pays per rank
before any communication: per-level launches from Python, no batch pipeline, full value arrays kept."""
import os, sys, time, importlib.util
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imt_amd
spec = importlib.util.spec_from_file_location("imt_sharded", os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "sharded.py"))
sharded = importlib.util.module_from_spec(spec); spec.loader.exec_module(sharded)
ctx = imt_amd.Context(0)
dev = torch.device("cuda", 0)
bs, nb = 1 << 16, 8
rng = np.random.default_rng(9)
raw = rng.integers(0, 256, size=(nb * bs, 32), dtype=np.uint8); raw[:, 31] &= 0x0f; raw[:, 0] |= 1
vals = torch.from_numpy(raw).to(dev)
t = imt_amd.IndexedTree(ctx, 32, 1 << 20)
rep = sharded.ReplicatedIndexedTree(imt_amd, ctx, t)
rep.insert_batch(vals[:bs])
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in range(1, nb):
    rep.insert_batch(vals[b * bs:(b + 1) * bs])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"stepping API, 1 rank: {dt / (nb - 1) * 1e3:.2f} ms per 2^16 batch = {(nb - 1) * bs / dt / 1e6:.2f} M insertions/s")
