import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import imt_amd, oracle_lib
ctx = imt_amd.Context(0)
rng = random.Random(123)
vals = oracle_lib.synth_values(900, 0x494D5407)
a = imt_amd.IndexedTree(ctx, 32, 1024)
b = imt_amd.IndexedTree(ctx, 32, 1024)
pos = 0
while pos < len(vals):
    n = min(rng.choice([1, 2, 5, 33, 64, 200]), len(vals) - pos)
    chunk = vals[pos:pos + n]
    print("batch", pos, n, flush=True)
    ra = a.insert_batch(chunk)
    rb = b.insert_batch(chunk, gpu_prep=True)
    ok = all((ra[k] == rb[k]).all() for k in ra)
    print("   equal", ok, flush=True)
    pos += n
print("done", a.root() == b.root())
