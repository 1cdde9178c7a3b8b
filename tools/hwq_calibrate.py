#!/usr/bin/env python3
"""The hardware-queue model of the sliced schedule (tests/hwq_model.py) against what one MI355X measured, and then asked
about eight.  CPU only.

  python tools/hwq_calibrate.py            the table: model vs measurement for every calibration point, then N = 2 / 4 / 8
                                           distributed with the link model and per-rank skew
  python tools/hwq_calibrate.py --fit      re-fit the free parameters (coordinate descent on the calibration points)

Calibration points (profiles/r05_sliced_costs.txt, profiles/r05_rank_emulation.txt; same box, same build):
  in-process replicas on one GPU, N = 1 / 2 / 4     tools/sliced_costs.py           (all ranks share the process's four queues)
  one rank of N = 4 / 8 alone on the GPU             tools/rank_emulation.py         (free collectives / 40 us + bytes / 48 GB/s)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hwq_model as M  # noqa: E402
import sliced_sim  # noqa: E402

B, DEPTH = 1 << 16, 32
MEASURED = {            # M insertions/s (per rank for "emu"): ONE box, one session, the round's final build (profiles/r05_calibration_session.txt)
    ("inproc", 1): 2.950, ("inproc", 2): 2.943, ("inproc", 4): 2.911,
    # one rank of N alone on the GPU, everything in the normal pool's four queues (IMT_SLICED_POOLS=0): means of two runs
    ("emu", 8, 0, "free"): 2.875, ("emu", 8, 0, "links"): 2.843,
    ("emu", 4, 0, "free"): 2.937, ("emu", 4, 0, "links"): 2.929,
    # the three priority pools (what one process per GPU gets): round streams HIGH, collectives LOW, preparation on the round stream
    ("emu-own", 8, 0, "free"): 2.799, ("emu-own", 8, 0, "links"): 2.810,
    ("emu-own", 4, 0, "free"): 2.890, ("emu-own", 4, 0, "links"): 2.870,
    ("emu-own", 2, 0, "free"): 2.925, ("emu-own", 2, 0, "links"): 2.925,
}
# Points the model is NOT fitted to, and why (none at present).  Shown in the table all the same.
EXCLUDED = {}
# ... and points it is fitted to but misses by more than the others: the bound the test holds them to
KNOWN_MISS = {("emu", 8, 0, "links"): 0.08}     # one pool at N = 8: the model prices the gather's time in the round's queue too high (-6 %)
lib = sliced_sim.load()


def timed_rate(progs, sh, qmap, costs, world, per_rank_insertions, rounds, **kw):
    T = M.Timed(progs, sh, qmap, costs, world, **kw)
    T.run()
    t = [st[-1] - st[-2] for st in T.sync_times if len(st) >= 2]
    return rounds * per_rank_insertions / (max(t) * 1e-6) / 1e6, T


def script(rounds, warm):
    return [("step",)] * warm + [("flush",)] + [("step",)] * rounds + [("flush",)]


def inproc(world, costs, rounds=12, warm=2):
    progs, sh, _ = M.record(lib, world, DEPTH, B, script(rounds, warm), hosts="one", comm_streams=4, costs=costs)
    r, T = timed_rate(progs, sh, M.QueueMap(one_device=True), costs, world, world * B, rounds)
    return r


def emu(world, rank, links, costs, rounds=12, warm=4, prep_on_round=False, **qm):
    c = costs
    if not links:
        c = type("C", (type(costs),), {})()
        c.__dict__.update(costs.__dict__)
        c.link_gbps = 0.0
    progs, sh, _ = M.record(lib, world, DEPTH, B, script(rounds, warm), transport="emu", comm_streams=4, costs=c, only_ranks=[rank],
                            prep_on_round=prep_on_round)
    r, T = timed_rate(progs, sh, M.QueueMap(**qm), c, world, B, rounds)
    return r


def emu_variant(world, rank, costs, lag=None, comm_streams=4, apply_streams=False, rounds=12, warm=4, **qm):
    progs, sh, _ = M.record(lib, world, DEPTH, B, script(rounds, warm), lag=lag, transport="emu", comm_streams=comm_streams,
                            apply_streams=apply_streams, costs=costs, only_ranks=[rank])
    r, T = timed_rate(progs, sh, M.QueueMap(**qm), costs, world, B, rounds)
    return r


# Held out: variants of the emulated rank 0 of 8 (modelled links) that round 4 measured (profiles/r04_rank_emulation.txt,
# "AFTER (precise waits)" block) and that no parameter was set from.  As ratios to the default of the same session.
HELD_OUT = [("lag 3 instead of 2", dict(lag=3), 2.376 / 2.817),
            ("collectives on the round streams (comm_streams = 0)", dict(comm_streams=0), 2.839 / 2.817),
            ("applies on streams of their own", dict(apply_streams=True), 2.708 / 2.817),
            ("the LAST rank instead of the first (r05: 2.817 / 2.731)", dict(rank=7), 2.817 / 2.731)]


# Held out as well (round 5, profiles/r05_prep_repeated_experiment.txt): the bulk of a step's preparation enqueued TWICE per step by a
# throw-away build (the sort, the gap search, the sparse-table levels and the index merge: 1.8 x its device time, 1.85 x
# its launches).  (queues of their own?, modelled links?) -> measured ratio, means of two runs each.
PREP_TWICE = {(True, True): (2.669 + 2.629) / (2.734 + 2.702), (True, False): (2.613 + 2.687) / (2.735 + 2.702),
              (False, True): (2.709 + 2.726) / (2.761 + 2.765), (False, False): (2.795 + 2.801) / (2.895 + 2.890)}
PREP_TWICE_BATCH_PIPELINE = (3.057 + 3.056) / (3.079 + 3.082)        # python bench.py: the one-GPU batch pipeline (no model of it here)


def prep_twice(costs):
    c = type("C", (type(costs),), {})()
    c.__dict__.update(costs.__dict__)
    c.prep_us = (costs.prep_us[0] * 1.8, costs.prep_us[1] * 2)
    c.prep_kernels = int(costs.prep_kernels * 1.85)
    return c


def show_held_out(costs):
    print("\nheld out: a step's preparation enqueued twice, one emulated rank of 8 (ratio to once)")
    print("  collectives' streams on       links      model   measured")
    c2 = prep_twice(costs)
    for (own, links), meas in PREP_TWICE.items():
        v = emu(8, 0, links, c2, comm_own_queues=own) / emu(8, 0, links, costs, comm_own_queues=own)
        print(f"  {'queues of their own' if own else 'their rounds queues':28s}  {'48 GB/s' if links else 'free   '}   {v:6.3f}   {meas:6.3f}")
    print(f"  (the one-GPU batch pipeline of bench.py, which this model does not describe: measured {PREP_TWICE_BATCH_PIPELINE:.3f})")
    base = emu_variant(8, 0, costs)
    print("\nheld-out variants of one emulated rank of 8 (ratio to the default; no parameter was fitted to these)")
    print("  variant                                                    model   measured")
    for name, kw, meas in HELD_OUT:
        kw = dict(kw)
        rank = kw.pop("rank", 0)
        v = emu_variant(8, rank, costs, **kw)
        print(f"  {name:58s} {v / base:6.3f}   {meas:6.3f}")


def distributed(world, costs, rounds=12, warm=4, speed=None, host_speed=None, transport="rccl", comm_streams=4, prep_on_round=False, **qm):
    progs, sh, _ = M.record(lib, world, DEPTH, B, script(rounds, warm), transport=transport, comm_streams=comm_streams, costs=costs,
                            rccl_internal=qm.get("rccl_dev") is not None, prep_on_round=prep_on_round)
    r, T = timed_rate(progs, sh, M.QueueMap(**qm), costs, world, world * B, rounds, speed=speed, host_speed=host_speed)
    return r, T


def points(costs, box=True):
    """the model's value for every calibration point; box=True: scaled to the speed of the box the session ran on (boxes
    differ by +-1.5 %, this one by -3 %: kernel durations in Costs are a typical box's) -- the scale is the one-GPU
    in-process point, which is therefore matched by construction"""
    out = _points(costs)
    if box:
        k = MEASURED[("inproc", 1)] / out[("inproc", 1)]
        out = {key: v * k for key, v in out.items()}
    return out


def _points(costs):
    out = {}
    for w in (1, 2, 4):
        out[("inproc", w)] = inproc(w, costs)
    for w in (8, 4):
        out[("emu", w, 0, "free")] = emu(w, 0, False, costs)
        out[("emu", w, 0, "links")] = emu(w, 0, True, costs)
    for w in (8, 4, 2):
        out[("emu-own", w, 0, "free")] = emu(w, 0, False, costs, comm_own_queues=True, prep_on_round=True)
        out[("emu-own", w, 0, "links")] = emu(w, 0, True, costs, comm_own_queues=True, prep_on_round=True)
    return out


def show(costs):
    t0 = time.time()
    got = points(costs)
    worst = 0.0
    print("calibration point                         model   measured   model / measured")
    for k, m in MEASURED.items():
        g = got[k]
        note = "   (not fitted: " + EXCLUDED[k] + ")" if k in EXCLUDED else "   (known miss)" if k in KNOWN_MISS else ""
        if k not in EXCLUDED and k not in KNOWN_MISS:
            worst = max(worst, abs(g / m - 1))
        print(f"  {str(k):38s} {g:6.3f}   {m:6.3f}     {g / m:6.3f}{note}")
    print(f"worst deviation of the fitted points {worst * 100:.1f} %   ({time.time() - t0:.0f} s)")
    return got, worst


def main():
    costs = M.Costs()
    show(costs)
    show_held_out(costs)
    if "--fit" in sys.argv:
        return
    print("\none process per GPU, RCCL semantics (a collective holds its queue until every rank's has reached the head of its own; every")
    print("communicator brackets its collectives with a stream of its own, tools/microbench/rccl_streams_probe.hip),")
    print(f"links {costs.link_latency_us:.0f} us + bytes / {costs.link_gbps:.0f} GB/s per peer")
    POOLS = dict(comm_own_queues=True, prep_on_round=True)

    one_gpu = inproc(1, costs)                     # the model's own one-GPU figure (a typical box: Costs' kernel durations)

    def line(w, r):
        return f"N = {w}: {r:6.2f} ({r / w / one_gpu:.3f})"

    def row(label, worlds=(2, 4, 8), **qm):
        print(f"  {label:92s} " + "   ".join(line(w, distributed(w, costs, **qm)[0]) for w in worlds), flush=True)

    print("M insertions/s (and the fraction of N x the one-GPU figure)")
    print(" three priority pools -- the library's placement for one process per GPU (IMT_SLICED_OPT_POOLS): rounds HIGH, collectives LOW")
    row("RCCL's streams in the normal pool, one queue each", rccl_dev=[8, 9, 10, 11], **POOLS)
    row("RCCL's four streams on ONE queue of the normal pool", rccl_dev=[8, 8, 8, 8], **POOLS)
    row("the IPC transport instead (counters polled by the GPUs, the peers' payloads read in one launch)", transport="ipc", **POOLS)
    print(" round AND collectives' streams in the HIGH pool, the collectives' on their rounds' queues (IMT_SLICED_OPT_POOLS 2)")
    row("RCCL's streams in the normal pool (the model misses this layout's emulated rank of 8 by -6 %)", rccl_dev=[8, 9, 10, 11])
    print(" one pool of four queues, collectives' streams on their rounds' queues (IMT_SLICED_OPT_POOLS 0, the runtime's defaults)")
    row("no stream of RCCL's own (the model before this was known)")
    row("the IPC transport", transport="ipc")
    row("RCCL's stream of channel c on round c's queue", rccl_dev=[0, 1, 2, 3])
    row("RCCL's stream of channel c on ANOTHER round's queue", rccl_dev=[1, 2, 3, 0])
    print(" one pool of eight queues, collectives' streams on queues of their own (GPU_MAX_HW_QUEUES=8: this round's earlier default)")
    row("no stream of RCCL's own", comm_own_queues=True)
    row("RCCL's stream of channel c on its collective stream's queue", comm_own_queues=True, rccl_dev=[4, 5, 6, 7])
    row("RCCL's stream of channel c on round c's queue", comm_own_queues=True, rccl_dev=[0, 1, 2, 3])
    row("RCCL's stream of channel c on another round's queue", comm_own_queues=True, rccl_dev=[1, 2, 3, 0])
    row("RCCL's streams where a round-robin over eight queues puts them (6, 3, 0, 5)", comm_own_queues=True, rccl_dev=[6, 3, 0, 5])
    import random
    rng = random.Random(5)
    print(" three priority pools, N = 8, GPUs and hosts of unequal speed")
    for skew in (0.015, 0.03):
        speed = {g: 1.0 + rng.uniform(-skew, skew) for g in range(8)}
        r, T = distributed(8, costs, speed=speed, host_speed={g: 1.0 + rng.uniform(-0.2, 0.2) for g in range(8)}, rccl_dev=[8, 9, 10, 11], **POOLS)
        print(f"  per-GPU speeds within +-{skew * 100:.1f} % and host speeds within +-20 %: {r:6.2f} M insertions/s "
              f"(slowest GPU {min(speed.values()):.3f})")
    print(" IMT_SLICED_OPT_POOLS 2, N = 8, GPUs and hosts of unequal speed (every tick a barrier across ranks: host jitter gets through)")
    rng = random.Random(5)
    for skew in (0.015, 0.03):
        speed = {g: 1.0 + rng.uniform(-skew, skew) for g in range(8)}
        r, T = distributed(8, costs, speed=speed, host_speed={g: 1.0 + rng.uniform(-0.2, 0.2) for g in range(8)}, rccl_dev=[8, 9, 10, 11])
        print(f"  per-GPU speeds within +-{skew * 100:.1f} % and host speeds within +-20 %: {r:6.2f} M insertions/s "
              f"(slowest GPU {min(speed.values()):.3f})")
    print(" one pool of four queues, N = 8, no stream of RCCL's own")
    for name, qm in (("collectives' streams on the NEXT slot's queue (comm_shift = 1)", dict(comm_shift=1)),
                     ("ranks with different rotations of the queue map", dict(rot={g: g % 4 for g in range(8)}))):
        r, T = distributed(8, costs, **qm)
        print(f"  {name}: {r:6.2f} M insertions/s")


if __name__ == "__main__":
    main()
