#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> a compact table the schedule model (tests/hwq_model.py, tools/hwq_calibrate.py) can be
calibrated against and a reader can diff: one line per dispatch, `queue stream name start_us end_us`, times relative to
the first dispatch, kernel names cut down to the bare function name.  The raw traces are tens of megabytes; these are
what is kept under profiles/.

Usage: python tools/trace_compact.py <dir with *kernel_trace.csv> <out.csv.gz> [--from-marker NAME]"""
import csv
import glob
import gzip
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:<>~ ,]+?)\(", name)
    base = m.group(1) if m else name
    base = base.split("<")[0] if base.startswith("rocprim") or "trampoline_kernel" in base else base
    if "trampoline_kernel" in name:
        m = re.search(r"wrapped_([a-z_]+)_config", name)
        base = "rocprim::" + (m.group(1) if m else "kernel")
    return base.replace("imt::", "").strip()


def main():
    src, out = sys.argv[1], sys.argv[2]
    files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit(f"no *kernel_trace.csv under {src}")
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            for r in rd:
                rows.append((int(r.get("Queue_Id", 0) or 0), int(r.get("Stream_Id", 0) or 0), short(r["Kernel_Name"]),
                             int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Thread_Id", 0) or 0)))
    rows.sort(key=lambda x: x[3])
    t0 = rows[0][3]
    queues = {q: i for i, q in enumerate(sorted({r[0] for r in rows}))}
    streams = {s: i for i, s in enumerate(sorted({r[1] for r in rows}))}
    with gzip.open(out, "wt") as fh:
        fh.write("queue,stream,kernel,start_us,end_us\n")
        for q, s, n, a, b, _ in rows:
            fh.write(f"{queues[q]},{streams[s]},{n},{(a - t0) / 1e3:.2f},{(b - t0) / 1e3:.2f}\n")
    print(f"{len(rows)} dispatches on {len(queues)} queues / {len(streams)} streams over {(rows[-1][4] - t0) / 1e6:.1f} ms -> {out}")


if __name__ == "__main__":
    main()
