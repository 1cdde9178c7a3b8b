#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --hip-runtime-trace CSVs -> for every kernel, when the HOST issued it (the hipLaunchKernel /
hipModuleLaunchKernel / hipMemcpyAsync ... call with the same correlation id) and when it RAN: where a kernel that starts
late was held up -- issued late (the host was elsewhere) or issued early and stuck in its hardware queue.
Prints, per kernel name pattern given, the distribution of (start - issue) and lists the first kernels of every step's
preparation.  Usage: python tools/trace_issue_lag.py <dir> [pattern ...]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))
    m = re.match(r"([A-Za-z_0-9:<>~ ,]+?)\(", name)
    base = m.group(1) if m else name
    if "trampoline_kernel" in name:
        m = re.search(r"wrapped_([a-z_]+)_config", name)
        base = "rocprim::" + (m.group(1) if m else "kernel")
    return base.replace("imt::", "").strip()


def main():
    d = sys.argv[1]
    pats = sys.argv[2:] or ["prep::k_scatter"]
    api = {}
    calls = []
    for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            api[r["Correlation_Id"]] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
            calls.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"], r["Correlation_Id"]))
    calls.sort()
    kname = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            kname[r["Correlation_Id"]] = (short(r["Kernel_Name"]), int(r.get("Queue_Id", 0) or 0), int(r.get("Stream_Id", 0) or 0),
                                          int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    if calls:
        import collections
        t00 = calls[0][0]
        tot = collections.Counter()
        for a, b, fn, th, cid in calls:
            tot[fn] += b - a
        print("host time per HIP call (ms, whole run):", {k: round(v / 1e6, 1) for k, v in tot.most_common(8)})
        print("kernel launches / copies / event calls that held the host for more than 0.5 ms (what was launched, when it then ran):")
        for a, b, fn, th, cid in calls:
            if b - a > 500_000 and fn in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipEventRecord",
                                          "hipStreamWaitEvent", "hipExtLaunchKernel", "hipEventSynchronize", "hipStreamSynchronize", "hipStreamQuery"):
                k = kname.get(cid)
                what = f"{k[0]} on queue {k[1]} stream {k[2]}, ran {(k[3] - b) / 1e3:.1f} us after the call returned" if k else ""
                print(f"   {fn:22s} at {(a - t00) / 1e3:10.1f} us for {(b - a) / 1e3:8.1f} us  {what}")
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            a = api.get(r["Correlation_Id"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Queue_Id", 0) or 0),
                         int(r.get("Stream_Id", 0) or 0), a[1] if a else None, a[0] if a else "?"))
    rows.sort()
    t0 = rows[0][0]
    print(f"{len(rows)} dispatches, {sum(1 for r in rows if r[5] is not None)} matched to an API call")
    sweeps = [(r[0], r[1]) for r in rows if r[2] == "k_sweep"]
    # ---- per step: when the host issued the preparation, when the device finished it, when the host issued the step's
    # hash launches -- the host's loop against the device's
    scat = [r for r in rows if r[2] == "prep::k_scatter" and r[5] is not None]
    steps = [scat[i] for i in range(len(scat)) if i == 0 or scat[i][5] - scat[i - 1][5] > 5_000_000]
    print("per step (ms from the step's first preparation launch): preparation done on the device | host issues the period's hash launches from .. to | "
          "k_sweep launches of the period | device idle (no k_sweep running) inside the step")
    for i, st in enumerate(steps[:-1]):
        a, b = st[5], steps[i + 1][5]
        prep_end = max((r[1] for r in rows if r[2] == "k_merge_level" and a <= r[5] < b), default=None)
        sw = [r for r in rows if r[2] == "k_sweep" and r[5] is not None and a <= r[5] < b]
        ev = sorted([(r[0], 1) for r in rows if r[2] == "k_sweep" and r[1] > a and r[0] < b] + [(r[1], -1) for r in rows if r[2] == "k_sweep" and r[1] > a and r[0] < b])
        h, last, idle = 0, a, 0
        for t, dd in ev:
            if h == 0:
                idle += max(0, min(t, b) - max(last, a))
            h += dd
            last = t
        if sw and prep_end:
            print(f"   step {i}: period {(b - a) / 1e6:6.2f} | {(prep_end - a) / 1e6:6.2f} | {(sw[0][5] - a) / 1e6:6.2f} .. {(sw[-1][5] - a) / 1e6:6.2f} | {len(sw):3d} | {idle / 1e6:5.2f}")
    if len(steps) >= 4 and calls:
        import collections
        a, b = steps[-4][5], steps[-1][5]
        agg = collections.defaultdict(lambda: [0, 0, 0])
        for c0, c1, fn, th, cid in calls:
            if a <= c0 < b:
                g = agg[fn]
                g[0] += 1
                g[1] += c1 - c0
                g[2] = max(g[2], c1 - c0)
        print(f"the host's HIP calls over the last three steps ({(b - a) / 1e6:.1f} ms): function: calls, total ms, longest us")
        for fn, g in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"   {fn:28s} {g[0]:6d} {g[1] / 1e6:8.2f} {g[2] / 1e3:9.1f}")
        inside = sum(g[1] for g in agg.values())
        print(f"   inside HIP calls {inside / 1e6:.1f} ms of {(b - a) / 1e6:.1f} ms")
    for pat in pats:
        sel = [r for r in rows if pat in r[2] and r[5] is not None]
        print(f"== {pat}: {len(sel)} dispatches; issued -> started (us), and how many k_sweep launches ran in between")
        for r in sel[:40]:
            between = sum(1 for a, b in sweeps if a < r[0] and b > r[5])
            print(f"   issued at {(r[5] - t0) / 1e3:10.1f} us   started {(r[0] - r[5]) / 1e3:9.1f} us later on queue {r[3]} stream {r[4]} ({between} sweeps overlapped the wait)")


if __name__ == "__main__":
    main()
