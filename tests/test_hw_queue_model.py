"""CPU: the multi-GPU single-list schedule of libimt_hip.so (csrc/imt_sliced_sched.hpp, the code imt_sliced_step runs) on a
model of the HIP runtime's HARDWARE QUEUES (tests/hwq_model.py): K in-order queues per device shared by all streams of a
process, event waits that block their whole queue, collectives that hold their queue until every peer's matching
collective has reached the head of ITS queue (RCCL; the GPU-polled IPC transport's flag waits likewise; the host-polled
form with a worker thread per rank), one host per rank, each advancing on its own.

 * progress: world 2 / 4 / 8, depth 32 at world 8, the default lag and one more / less, every rotation of the queue map
   (and a different one per rank), the helper streams on their own or on the round streams, collectives' streams moved
   onto ANOTHER round's queue, fewer communicators than round slots, hosts that wait for old rounds and flush in the
   middle -- under an adversarial scheduler every run drains and every unit still sees exactly the earlier slices
   (one list, slices in insertion order: /root/reference/src/indexed_merkle_tree.rs:632-660, :715);
 * what the model catches: ranks whose call sequences differ (the old rank-dependent tick of imt_sliced_wait, rebuilt as
   mutation 7; a rank that calls imt_sliced_wait when its peer does not) end in a reported cycle or in collectives that
   do not match;
 * timing: the discrete-event form with measured kernel durations reproduces what one MI355X delivered (in-process
   replicas N = 1 / 2 / 4, one emulated rank of 2, 4 and 8) within 3 %, and is then asked what 2 / 4 / 8 GPUs deliver.
"""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import hwq_model as M  # noqa: E402
import sliced_sim  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return sliced_sim.load()


def script(rounds=6, waits=False, flush_at=(), back=2):
    s = []
    for r in range(rounds):
        s.append(("step",))
        if waits and r >= back:
            s.append(("wait", r - back))
        if r in flush_at:
            s.append(("flush",))
    s.append(("flush",))
    return s


def check_replicas(sh, world, rounds, depth):
    for rank, rp in sh.ranks.items():
        assert sorted(rp.computed) == [(r * world + rank, q) for r in range(rounds) for q in range(depth + 1)], rank
        for lvl in rp.levels:
            assert lvl == list(range(rounds * world)), (rank, lvl[:8])


def lags_around_default(lib, world, depth):
    d = sliced_sim.Schedule(lib, world, depth + 1).lag
    out = [None]
    for lag in (d - 1, d + 1):
        try:
            sliced_sim.Schedule(lib, world, depth + 1, lag)
            out.append(lag)
        except ValueError:
            pass
    return out


@pytest.mark.parametrize("world,depth", [(2, 8), (4, 8), (8, 32), (3, 5)])
@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_hw_queue_model_progress(lib, world, depth, transport):
    """every run drains, whatever the queue map and however the hosts and queues are interleaved"""
    rounds = 6 if world < 8 else 5
    full = os.environ.get("IMT_SIM_FULL") == "1"            # the whole grid (minutes); the default is a covering sample
    seeds = range(int(os.environ.get("IMT_SIM_SEEDS", "2" if full else "1")))
    runs = 0
    all_maps = [M.QueueMap(K=4, rot={g: off for g in range(world)}) for off in range(4)]
    all_maps += [M.QueueMap(K=4, rot={g: g % 4 for g in range(world)}),              # another rotation on every rank
                 M.QueueMap(K=4, comm_shift=1), M.QueueMap(K=4, comm_shift=2, apply_shift=1),
                 M.QueueMap(K=2), M.QueueMap(K=1), M.QueueMap(K=4, comm_own_queues=True)]
    combo = 0
    for lag in lags_around_default(lib, world, depth):
        for apply_streams in (False, True):
            for comm_streams in ((4, 0) if transport == "rccl" else (4,)):
                sc = script(rounds, waits=(lag is None), flush_at=(2,) if apply_streams else ())
                if full:
                    maps = all_maps
                elif lag is None and not apply_streams and comm_streams == 4:
                    maps = all_maps if world < 8 else [all_maps[0], all_maps[4], all_maps[5], all_maps[9]]       # the product's layout: every map (a sample at world 8)
                else:
                    maps = [all_maps[(combo + k) % len(all_maps)] for k in (0, 5)]       # the others: two maps each, rotating
                combo += 1
                for qm in maps:
                    for seed in seeds:
                        progs, sh, _ = M.record(lib, world, depth, 4, sc, lag=lag, transport=transport, comm_streams=comm_streams,
                                                apply_streams=apply_streams, real_sizes=False)
                        n = M.replay_adversarial(progs, sh, qm, seed=seed * 7919 + runs, world=world)
                        assert n > 0
                        check_replicas(sh, world, rounds, depth)
                        runs += 1
    assert runs >= 8


@pytest.mark.parametrize("world,depth", [(2, 8), (4, 8), (8, 32), (3, 5)])
def test_hw_queue_model_progress_with_rccls_own_streams(lib, world, depth):
    """RCCL brackets every collective with a stream of the communicator's own (the user's stream waits for an event of
    it, it waits for an event behind the kernel: tools/microbench/rccl_streams_probe.hip finds three streams per
    communicator in the normal-priority pool): wherever those streams sit -- a queue nothing else is on, their own
    slot's queue, ANOTHER round's queue, all four on one round's queue, a different place on every rank -- every run
    drains and every unit sees exactly the earlier slices.  (Both waits point at something issued earlier on the same
    rank, so the progress argument of DESIGN 8a covers them; what they cost is the next test.)"""
    rounds = 6 if world < 8 else 5
    maps = [M.QueueMap(K=4, rccl_dev=[0, 1, 2, 3]), M.QueueMap(K=4, rccl_dev=[1, 2, 3, 0]), M.QueueMap(K=4, rccl_dev=[2, 2, 2, 2]),
            M.QueueMap(K=4, comm_own_queues=True, rccl_dev=[8, 9, 10, 11]), M.QueueMap(K=4, comm_own_queues=True, rccl_dev=[6, 3, 0, 5]),
            M.QueueMap(K=4, comm_own_queues=True, rccl_dev={g: [(g + c) % 8 for c in range(4)] for g in range(world)}),
            M.QueueMap(K=1, rccl_dev=[0, 0, 0, 0]), M.QueueMap(K=2, comm_shift=1, rccl_dev=[1, 0, 1, 0])]
    runs = 0
    for lag in lags_around_default(lib, world, depth):
        for comm_streams, channels in ((4, 0), (0, 0), (4, 2)):
            sc = script(rounds, waits=(lag is None), flush_at=(2,) if comm_streams == 0 else ())
            for qm in (maps if (lag is None and world < 8) else maps[runs % 3::3]):
                progs, sh, _ = M.record(lib, world, depth, 4, sc, lag=lag, transport="rccl", comm_streams=comm_streams, channels=channels,
                                        real_sizes=False, rccl_internal=True)
                assert M.replay_adversarial(progs, sh, qm, seed=4001 + runs, world=world) > 0
                check_replicas(sh, world, rounds, depth)
                runs += 1
    assert runs >= 12


def test_hw_queue_model_rccls_own_streams_must_stay_off_the_rounds_queues(lib):
    """in time: one of RCCL's streams on a round's hardware queue puts that round's hash kernels behind every collective
    of its communicator and the collective behind the round's backlog -- the reason for the three priority pools
    (IMT_SLICED_OPT_POOLS: rounds HIGH, collectives LOW, everybody else's streams normal)"""
    import hwq_calibrate as C
    costs = M.Costs()
    pools, _ = C.distributed(4, costs, rounds=6, warm=3, comm_own_queues=True, prep_on_round=True, rccl_dev=[8, 9, 10, 11])
    crowded, _ = C.distributed(4, costs, rounds=6, warm=3, comm_own_queues=True, prep_on_round=True, rccl_dev=[8, 8, 8, 8])
    one_pool, _ = C.distributed(4, costs, rounds=6, warm=3, rccl_dev=[1, 2, 3, 0])
    eight, _ = C.distributed(4, costs, rounds=6, warm=3, comm_own_queues=True, rccl_dev=[1, 2, 3, 0])
    assert crowded > 0.97 * pools                 # RCCL's streams sharing ONE queue among themselves: harmless
    assert one_pool < 0.85 * pools and eight < 0.85 * pools, (pools, one_pool, eight)


@pytest.mark.parametrize("world", [2, 4])
def test_hw_queue_model_progress_host_polled_and_depth_32(lib, world):
    """the HOST-polled form of the IPC transport (ranks that share a GPU: a worker thread per rank watches the peers'
    counters and enqueues each payload's copies; the issuing host waits in the fence for the worker and for the peers'
    acknowledgements -- hosts wait for hosts, nothing on the device waits for a peer), two hosts per rank; and the
    other transports at depth 32 for the small worlds"""
    runs = 0
    for depth, transports in ((8, ("ipc-host",)), (32, ("ipc-host", "rccl", "ipc"))):
        for transport in transports:
            for k, qm in enumerate((M.QueueMap(K=4), M.QueueMap(K=4, rot={g: g % 4 for g in range(world)}), M.QueueMap(K=1),
                                    M.QueueMap(K=4, comm_own_queues=True), M.QueueMap(K=4, comm_shift=1))):
                if depth == 32 and k % 2:
                    continue
                progs, sh, _ = M.record(lib, world, depth, 4, script(6, waits=True, flush_at=(2,)), transport=transport, real_sizes=False)
                assert len(progs) == (2 * world if transport == "ipc-host" else world)
                M.replay_adversarial(progs, sh, qm, seed=17 * runs + world, world=world)
                check_replicas(sh, world, 6, depth)
                runs += 1
    assert runs >= 12


@pytest.mark.parametrize("channels", [1, 2, 3])
def test_fewer_communicators_than_round_slots(lib, channels):
    """round slots that share a channel of the transport (an RCCL communicator) enqueue on ONE stream, so every rank's
    communicator sees the same sequence of calls (sizes included), at world 4 with waits in the middle"""
    world, depth, rounds = 4, 8, 6
    for seed in range(3):
        for qm in (M.QueueMap(K=4), M.QueueMap(K=4, rot={g: g % 4 for g in range(world)}), M.QueueMap(K=2)):
            progs, sh, _ = M.record(lib, world, depth, 4, script(rounds, waits=True), transport="rccl", channels=channels, real_sizes=False)
            M.replay_adversarial(progs, sh, qm, seed=seed, world=world)
            check_replicas(sh, world, rounds, depth)


def one_host_in_process(lib):
    """all replicas in one process (the rehearsal form: one host, the replicas' streams share the process's four queues)"""
    for world in (2, 4):
        for seed in range(2):
            progs, sh, _ = M.record(lib, world, 8, 4, script(6, waits=True), hosts="one", real_sizes=False)
            M.replay_adversarial(progs, sh, M.QueueMap(K=4, one_device=True), seed=seed, world=world)
            check_replicas(sh, world, 6, 8)


def test_in_process_world_on_four_shared_queues(lib):
    one_host_in_process(lib)


def mutated_lib(tmp_path, mutation):
    src = os.path.join(ROOT, "tests", "native", "sliced_sym.cpp")
    so = str(tmp_path / f"libslicedsym_m{mutation}.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DIMT_TEST_BUILD", f"-DIMT_SCHED_MUTATION={mutation}", "-o", so, src], check=True)
    mlib = ctypes.CDLL(so)
    ref = sliced_sim.load()
    for name in ("sym_schedule", "sym_unit_of", "sym_payload_units", "sym_world_create", "sym_world_create_channels", "sym_world_step",
                 "sym_world_flush", "sym_world_run_all", "sym_world_wait", "sym_world_collectives", "sym_world_destroy", "sym_set_layout",
                 "sym_world_tick"):
        fn, r = getattr(mlib, name), getattr(ref, name)
        fn.argtypes, fn.restype = r.argtypes, r.restype
    return mlib


def outcome(lib, world, depth, sc, qm, seeds=4, **kw):
    for seed in range(seeds):
        try:
            progs, sh, _ = M.record(lib, world, depth, 4, sc, real_sizes=False, **kw)
            M.replay_adversarial(progs, sh, qm, seed=seed, world=world)
            check_replicas(sh, world, sum(1 for c in sc if c[0] == "step"), depth)
        except M.Deadlock as e:
            return "deadlock", str(e)
        except AssertionError as e:
            return "mismatch", str(e)
    return "passed", ""


def test_the_model_catches_rank_dependent_ticks(lib, tmp_path):
    """Mutation 7 = imt_sliced_wait as it was until round 5: it advanced the schedule to THIS rank's last compute tick, so
    the global tick -- and with it the next round's start and the order in which collectives of different rounds are
    issued -- came to depend on the rank.  With round slots sharing a communicator the ranks' collectives no
    longer match; with every slot on a communicator of its own the hosts' different orders meet on shared hardware
    queues and the ranks wait for each other: the model reports the cycle.  The repaired code passes the same runs."""
    m7 = mutated_lib(tmp_path, 7)
    sc = script(7, waits=True, back=1)          # a wait for the round before the last: its last units are not issued yet
    cases = [dict(world=2, depth=8, qm=M.QueueMap(K=1), kw=dict(transport="rccl")),
             dict(world=4, depth=32, qm=M.QueueMap(K=1), kw=dict(transport="rccl")),
             dict(world=4, depth=32, qm=M.QueueMap(K=4), kw=dict(transport="rccl", channels=2)),
             dict(world=4, depth=32, qm=M.QueueMap(K=4), kw=dict(transport="rccl", channels=1))]
    seen = set()
    for c in cases:
        assert outcome(lib, c["world"], c["depth"], sc, c["qm"], **c["kw"])[0] == "passed"
        kind, text = outcome(m7, c["world"], c["depth"], sc, c["qm"], **c["kw"])
        assert kind in ("deadlock", "mismatch"), f"mutation 7 went unnoticed in {c}"
        seen.add(kind)
        if kind == "deadlock":
            assert "holds the queue until every rank runs collective" in text
    assert "deadlock" in seen or "mismatch" in seen


def test_the_model_catches_unequal_call_sequences(lib):
    """the contract of include/imt.h: the same sequence of imt_sliced_step / _wait / _flush on every rank.  A rank that waits
    for an old round when its peer does not issues more ticks before its next step than the peer: their collectives part
    ways -- reported, not hung"""
    world, depth = 2, 8
    same = script(7, waits=True, back=1)
    other = script(7, waits=False)
    assert outcome(lib, world, depth, same, M.QueueMap(K=2), transport="rccl", channels=2)[0] == "passed"
    kind, text = outcome(lib, world, depth, same, M.QueueMap(K=2), transport="rccl", channels=2, rank_scripts={0: same, 1: other})
    assert kind in ("deadlock", "mismatch"), "ranks with different call sequences went unnoticed"


def test_misplaced_helper_streams_cannot_hang(lib):
    """the collectives' streams on ANOTHER round's hardware queue, or ranks whose queue maps differ: every dependency of the
    schedule points to something issued earlier (or to the same-numbered call on a peer, issued in the same order
    everywhere), so no placement can deadlock it -- the adversarial runs above drain for every map -- and in time the
    world-8 model finishes for each of them; which physical queue is "queue 0" on a rank does not matter at all"""
    import hwq_calibrate as C
    costs = M.Costs()
    base, _ = C.distributed(8, costs, rounds=6, warm=3)
    rotated, _ = C.distributed(8, costs, rounds=6, warm=3, rot={g: g % 4 for g in range(8)})
    assert abs(rotated / base - 1) < 0.01, (base, rotated)
    for shift in (1, 2, 3):
        shifted, _ = C.distributed(8, costs, rounds=6, warm=3, comm_shift=shift)
        assert 0.7 * base < shifted < 1.15 * base, (shift, base, shifted)
    own, _ = C.distributed(8, costs, rounds=6, warm=3, comm_own_queues=True)
    assert own > base                       # a gather on a queue of its own overlaps its round's next units


def test_hw_queue_model_timing_is_calibrated():
    """the discrete-event form against what one MI355X measured this round (profiles/r05_sliced_costs.txt,
    profiles/r05_emu_*.txt; tools/hwq_calibrate.py prints the table): every calibration point within 3.5 % (worst as fitted: 2.8 %)"""
    import hwq_calibrate as C
    got = C.points(M.Costs())
    for k, m in C.MEASURED.items():
        if k in C.EXCLUDED:            # (listed with its reason in tools/hwq_calibrate.py)
            continue
        # the fit stands at 2 %; the model is a discrete schedule, a parameter's last digit moves a point by 0.5 %
        assert abs(got[k] / m - 1) < C.KNOWN_MISS.get(k, 0.035), (k, got[k], m)
    # and what it says about one process per GPU with RCCL semantics and the link model: past north_star's 10^7 at N = 8
    r8, _ = C.distributed(8, M.Costs(), rounds=8, warm=4)
    assert r8 > 10.0, r8
