#!/usr/bin/env python3
"""Runs tools/microbench/lane_mont (one Montgomery product spread over the 16 lanes of a DPP row) and checks its chain of
products against Python integers; prints the measured time per product for the row form and for the library's
single-lane product.  Usage: python tools/lane_mont_check.py [iters]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
R = 1 << 261
A = [0x117fd374, 0x1e0f51b7, 0x8cc954f, 0xc82714c, 0x16a3b0d4, 0x1446f350, 0x3d8a09d, 0xbe39f62, 0x17cb76]
B = [0x1814e8a2, 0x12938803, 0x7d96a37, 0x12b39c7e, 0x1e968617, 0x1f43c599, 0x1d14686b, 0x1ad25db9, 0xff508]
val = lambda limbs: sum(v << (29 * i) for i, v in enumerate(limbs))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out = subprocess.run([os.path.join(ROOT, "tools", "microbench", "lane_mont"), str(iters)], capture_output=True, text=True, timeout=120)
print(out.stdout, out.stderr[-500:])
assert out.returncode == 0
x = val(A)
rinv = pow(R, -1, P)
for _ in range(iters):
    x = x * val(B) * rinv % P
ok = True
for line in out.stdout.splitlines():
    if " limbs " in line:
        kind = line.split()[0]
        limbs = [int(t, 16) for t in line.split(" limbs ")[1].split()]
        got = val(limbs)
        good = got % P == x and got < 4 * P and all(v < (1 << 29) + 64 for v in limbs)
        print(f"{kind} form after {iters} dependent products: {'equal to the integers (mod p), limbs in range' if good else 'WRONG'}")
        ok = ok and good
sys.exit(0 if ok else 1)
