#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of the gfx950 code objects, from hipcc's own
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed).  Usage: python tools/kernel_resources.py [file.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc")
files = [os.path.abspath(f) for f in sys.argv[1:]] or [os.path.join(CSRC, "imt_kernels.hip"), os.path.join(CSRC, "imt_prep.hip")]
print(f"{'kernel':44s} {'VGPR':>5s} {'spill':>6s} {'scratch B':>9s} {'LDS B':>6s} {'waves/SIMD':>10s}")
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", f, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=CSRC)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = cur.replace("(anonymous namespace)::", "")
            cur = re.sub(r"^void ", "", cur)
            cur = re.sub(r"\((?!.*\().*$", "", cur).replace("imt::", "").replace("prep::", "prep/")
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1)] = int(m.group(2))
    for k, v in rows.items():
        if "rocprim" in k or not v:
            continue
        print(f"{k[:44]:44s} {v.get('VGPRs', 0):5d} {v.get('VGPRs Spill', 0):6d} {v.get('ScratchSize [bytes/lane]', 0):9d} "
              f"{v.get('LDS Size [bytes/block]', 0):6d} {v.get('Occupancy [waves/SIMD]', 0):10d}")
