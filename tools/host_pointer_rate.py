import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, imt_amd, oracle_lib
ctx = imt_amd.Context(0)
n = 1 << 16
vals = oracle_lib.synth_values(6 * n, 0x494D5402)
t = imt_amd.IndexedTree(ctx, 32, 1 << 20)
t.insert_batch(vals[:n])
ts = []
for i in range(1, 6):
    t0 = time.perf_counter(); t.insert_batch(vals[i * n:(i + 1) * n]); ts.append(time.perf_counter() - t0)
print("host-pointer insert_batch 2^16, depth 32 (incl. H2D values, D2H roots+proofs, numpy alloc): ms", [round(x * 1e3, 1) for x in ts],
      "-> %.2f M insertions/s" % (n / min(ts) / 1e6))
