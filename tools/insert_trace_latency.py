"""Wall time of imt_insert_trace_batch (host pointers) for a few insertions at depth 32."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, imt_amd, oracle_lib
c = imt_amd.Context(0)
t = imt_amd.IndexedTree(c, 32, 1 << 12)
t.insert_batch(oracle_lib.synth_values(100, 5))
for n in (1, 8, 64):
    r = t.insert_batch(oracle_lib.synth_values(n, 100 + n))
    f = lambda: c.insert_trace(r["low_leaf"], r["low_index"], r["low_sib"], r["new_leaf"], r["new_index"], r["new_sib"], 32)
    f()
    t0 = time.perf_counter(); tr = f(); dt = time.perf_counter() - t0
    print(f"insert_trace n={n:3d} depth 32: {dt * 1e3:7.1f} ms  ({tr.nbytes / 1e6:.0f} MB of rows)")
