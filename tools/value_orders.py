#!/usr/bin/env python3
"""Throughput on ORDERED inputs at config-2 size (VERDICT r5 "missing" 5 / "next" 3).

Every published rate of this repo is on random values.  A sequencer in front of an indexed tree typically submits
sorted batches, and order changes the shape of everything that replaces the reference's scan
(/root/reference/src/indexed_merkle_tree.rs:639-658): ascending values make every insertion's low leaf the insertion
before it (one chain through the batch), descending values make leaf 0 the low leaf of ALL of them (one node with 2^16
versions per level: one (node, time) run for k_merge_level), clustered values share their high limbs (the 256-bit
compare of the sort and of the lower bound goes to the last limb every time).

This runs bench.py's own one-GPU leg (bench.bench_subtrees: same tree, same calls, same pipelining, same verification by
the independent witness kernels) with bench.synth_values replaced by an ordered variant, and prints insertions/s, the
step's wall time, the GPU time per kernel class (imt_profile: k_sweep by kind, k_merge_level, k_writeback) and the host's
time inside the call.     python tools/value_orders.py [--steps 12 --warmup 3] [--only descending]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ORDERS = ("random", "ascending", "descending", "sorted_batches", "reverse_sorted_batches", "clustered", "clustered_ascending",
          "interleaved")
_random = bench.synth_values


def by_value(v):
    limbs = v.view("<u8").reshape(-1, 4)
    return np.lexsort((limbs[:, 0], limbs[:, 1], limbs[:, 2], limbs[:, 3]))


def ordered_values(kind, total, residue=0, modulus=1, seed=0x494D5402):
    """uint8 [total, 32]: the bench's random draw, re-ordered / re-shaped per `kind` (all values distinct, 0 < v < p)"""
    v = _random(total, residue, modulus, seed)
    B = bench.BATCH
    if kind == "random":
        return v
    if kind == "ascending":                      # the whole run ascending: each value larger than everything before it
        return np.ascontiguousarray(v[by_value(v)])
    if kind == "descending":                     # each value smaller than everything before it: the low leaf is always leaf 0
        return np.ascontiguousarray(v[by_value(v)[::-1]])
    if kind in ("sorted_batches", "reverse_sorted_batches"):     # sorted within a batch, random across batches
        out = v.copy()
        for b in range(0, total, B):
            o = by_value(v[b:b + B])
            out[b:b + B] = v[b:b + B][o if kind == "sorted_batches" else o[::-1]]
        return out
    if kind in ("clustered", "clustered_ascending"):             # the three high limbs shared: values differ in the low 64 bits only
        out = v.copy()
        limbs = out.view("<u8").reshape(-1, 4)
        limbs[:, 1:] = limbs[0, 1:]
        low = np.unique(limbs[:, 0])
        while low.size < total:                  # (collisions of 64-bit draws: practically never)
            low = np.unique(np.concatenate([low, np.random.default_rng(seed + low.size).integers(1, 1 << 63, total, dtype=np.uint64)]))
        low = low[:total]
        if kind == "clustered":
            low = np.random.default_rng(seed).permutation(low)
        limbs[:, 0] = low
        return out
    if kind == "interleaved":                    # two ascending runs merged alternately: low leaves alternate between two chains
        s = v[by_value(v)]
        out = np.empty_like(s)
        out[0::2], out[1::2] = s[:(total + 1) // 2], s[(total + 1) // 2:]
        return out
    raise ValueError(kind)


def run(kind, steps, warmup):
    import torch
    bench.synth_values = lambda total, residue, modulus, seed: ordered_values(kind, total, residue, modulus, 0x494D5402)
    try:
        args = argparse.Namespace(gpus=1, steps=steps, warmup=warmup, no_cpu_baseline=True)
        env = bench.Env(args)
        r = bench.bench_subtrees(env)
    finally:
        bench.synth_values = _random
    k = r["kernels"]
    line = dict(order=kind, value=r["value"], ms_per_step=r["ms_per_step"], verified=bool(r["verified"]),
                gpu_kernel_ms_per_step=r["gpu_kernel_ms_per_step"], host_call_ms_per_step=r["host_call_ms_per_step"],
                sweep_leaves_ms=k["k_sweep[leaves]"]["ms_total"] / steps,
                sweep_low_ms=k["k_sweep[level<l0]"]["ms_total"] / steps, sweep_high_ms=k["k_sweep[level>=l0]"]["ms_total"] / steps,
                merge_level_ms=k["index(k_merge_level)"]["ms_total"] / steps, writeback_ms=k["k_writeback"]["ms_total"] / steps,
                alone_ms=r["alone_ms"], pipe_ms=r["pipe_ms"])
    r["be"].tree.close()
    r["be"].sets = r["be"].structs = None
    r["ctx"].close()
    del r
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.environ.setdefault("IMT_BENCH_NO_TRACE", "1")
    kinds = [a.only] if a.only else ORDERS
    print(f"# bench.py's one-GPU leg (depth 32, 2^16 insertions per step, {a.steps} timed steps after {a.warmup}; every witness written, "
          f"last + middle step verified by the witness kernels) on ordered values; ms are per step")
    print(f"{'order':24s} {'M ins/s':>8s} {'vs random':>9s} {'ms/step':>8s} {'verified':>8s} {'leaves':>7s} {'lv<l0':>7s} {'lv>=l0':>7s} "
          f"{'merge':>7s} {'wrback':>7s} {'host call':>9s} {'sweep alone / piped ms':>22s}")
    base = None
    rows = []
    for kind in kinds:
        r = run(kind, a.steps, a.warmup)
        rows.append(r)
        if kind == "random":
            base = r["value"]
        rel = f"{r['value'] / base:9.3f}" if base else "        -"
        print(f"{kind:24s} {r['value'] / 1e6:8.3f} {rel} {r['ms_per_step']:8.2f} {str(r['verified']):>8s} {r['sweep_leaves_ms']:7.2f} "
              f"{r['sweep_low_ms']:7.2f} {r['sweep_high_ms']:7.2f} {r['merge_level_ms']:7.2f} {r['writeback_ms']:7.2f} "
              f"{r['host_call_ms_per_step']:9.2f} {(r['alone_ms'] or 0):10.3f} / {(r['pipe_ms'] or 0):.3f}", flush=True)
    if not all(r["verified"] for r in rows):
        sys.exit(1)


if __name__ == "__main__":
    main()
