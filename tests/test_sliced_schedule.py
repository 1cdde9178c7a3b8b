"""CPU: the multi-GPU single-list schedule of libimt_hip.so (csrc/imt_sliced_sched.hpp: Schedule, Rank, World,
LocalTransport -- what imt_sliced_step / imt_sliced_flush run) compiled over a symbolic backend
(tests/native/sliced_sym.cpp, tests/sliced_sim.py).  What is under test is everything that is not hashing: the systolic
schedule, which payload is applied where and when, the events between rounds, buffer reuse, the line-up of the
collectives across ranks.

 * all ranks in one process, deferred execution in random interleavings (only stream order and event waits are
   respected): every slice's every level must see exactly the slices before it -- the rule that makes the replicas equal
   the reference's sequential list (src/indexed_merkle_tree.rs:632-660), at world 1..16, several depths, lags and seeds;
 * three planted mutations of the schedule code (a dropped ordering rule each) must be caught;
 * one rank per process over gloo with 2, 4 and 8 processes: the same assertions, real collectives.
The GPU tests (tests/test_gpu_sliced.py) run the same classes over HIP against the one-GPU tree and the oracle."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sliced_sim  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return sliced_sim.load()


def test_schedule_arithmetic(lib):
    for world in (1, 2, 3, 4, 8, 16):
        for units in (4, 9, 33):
            sc = sliced_sim.Schedule(lib, world, units)
            assert -(-sc.round_ticks // sc.period) <= sliced_sim.ROUNDS
            seen = set()
            for rt in range(sc.round_ticks):
                for g in range(world):
                    q = sc.unit_of(g, rt)
                    if q is not None:
                        assert (g, q) not in seen
                        seen.add((g, q))
                        assert rt < sc.gathers
            assert len(seen) == world * units                         # every unit of every rank exactly once
            # every payload is carried by exactly one collective and consumed `lag` ticks later, inside the round
            carried = [(g, q) for rt in range(sc.gathers) for g, q in enumerate(sc.payload_units(rt)) if q >= 0]
            assert sorted(carried) == sorted((g, q) for g in range(world) for q in range(1, units))
            assert all(sc.has_gather(rt) == any(q >= 0 for q in sc.payload_units(rt)) for rt in range(sc.round_ticks))
    with pytest.raises(ValueError):
        sliced_sim.Schedule(lib, 1, 33, lag=2)       # 17 rounds in flight
    sc = sliced_sim.Schedule(lib, 8, 33)
    assert (sc.lag, sc.period, sc.gathers) == (2, 16, 47)
    assert [sliced_sim.Schedule(lib, w, 33).lag for w in (1, 2, 4)] == [11, 6, 3]


def run_world(lib, world, depth, lag, seed, rounds=7, batch=4):
    sim = sliced_sim.Sim(immediate=False, seed=seed)
    w = sliced_sim.SymWorld(lib, sim, world, depth, batch, lag)
    for r in range(rounds):
        assert w.step() == r
        if seed % 2 and r == 3:
            sim.run()                    # a caller that synchronises in the middle
        if seed % 4 == 2 and r in (1, 4):
            w.flush()                    # ... or runs the schedule dry and goes on (bench.py: warm-up, then the timed region)
        if seed % 4 == 3 and r >= 2:
            w.wait(r % world, r - 2)     # ... or waits for one rank's witnesses of an older round
    w.flush()
    for rank, rp in w.reps.items():
        assert sorted(rp.computed) == [(r * world + rank, q) for r in range(rounds) for q in range(depth + 1)]
        for lvl in rp.levels:
            assert lvl == list(range(rounds * world))      # every replica holds every slice's write-back, in order
    assert w.collectives > 0 or world == 1
    w.close()


@pytest.mark.parametrize("world,depth,lag", [(1, 8, None), (2, 8, None), (2, 8, 5), (4, 8, None), (4, 32, None), (8, 32, None),
                                             (8, 32, 4), (3, 8, None), (16, 32, None), (8, 8, 1), (2, 3, 2)])
def test_every_level_sees_exactly_the_earlier_slices(lib, world, depth, lag):
    for seed in range(int(os.environ.get("IMT_SIM_SEEDS", "4" if world < 16 else "2"))):
        run_world(lib, world, depth, lag, seed)


def test_the_simulator_catches_planted_mutations(tmp_path):
    """the schedule code rebuilt with one ordering rule dropped each time -- round R's units behind round R - 1's applies
    and own units (1), round R's applies behind them (2), the fence that keeps a send buffer until every peer has copied
    it (3), a unit behind its own tick's apply now that applies have a stream of their own (4), a collective behind the
    apply that last read its receive buffer (5), round R writing a level as soon as round R - 1 has written it, without
    waiting for that round to have read it (6) -- must fail under the adversarial scheduler for some seed; the unmodified
    code passes the same runs"""
    src = os.path.join(ROOT, "tests", "native", "sliced_sym.cpp")

    def outcome(mutation, extra=(), apply_on_round=False):
        so = str(tmp_path / f"libslicedsym_m{mutation}{len(extra)}{int(apply_on_round)}.so")
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DIMT_TEST_BUILD", f"-DIMT_SCHED_MUTATION={mutation}", *extra, "-o", so, src],
                       check=True)
        mlib = ctypes.CDLL(so)
        for name in ("sym_schedule", "sym_unit_of", "sym_payload_units", "sym_world_create", "sym_world_step", "sym_world_flush",
                     "sym_world_run_all", "sym_world_wait", "sym_world_collectives", "sym_world_destroy"):
            ref = getattr(sliced_sim.load(), name)
            fn = getattr(mlib, name)
            fn.argtypes, fn.restype = ref.argtypes, ref.restype
        mlib.sym_set_layout.argtypes, mlib.sym_set_layout.restype = [ctypes.c_int, ctypes.c_int], None
        if apply_on_round:              # the product's default layout: applies on the round's own stream
            mlib.sym_set_layout(sliced_sim.ROUNDS, 0)
        for world, depth, lag in ((2, 8, None), (4, 8, None), (8, 8, 1), (3, 5, 2)):
            for seed in range(6):
                try:
                    run_world(mlib, world, depth, lag, seed)
                except AssertionError:
                    return "caught"
        return "passed"

    assert outcome(0) == "passed"
    # the A/B build that issues every wait, also those another wait on the same stream implies (docs/LAB_NOTES.md)
    assert outcome(0, ("-DIMT_SCHED_ALL_WAITS",)) == "passed"
    for m in (1, 2, 3, 4, 5, 6):
        assert outcome(m) == "caught", f"mutation {m} went unnoticed"
    # the waits that are NOT issued because another one implies them (round 5): the unmodified code passes in the layout
    # where they are dropped (applies on the round's stream); a note that claims more than was waited for (8) is caught
    # there, a tick event taken for implied although the previous round applies on another stream (9) in the other
    assert outcome(0, apply_on_round=True) == "passed"
    assert outcome(8, apply_on_round=True) == "caught", "mutation 8 went unnoticed"
    assert outcome(9) == "caught", "mutation 9 went unnoticed"


def _worker(rank, world, port, depth, lag, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = sliced_sim.load()
    sim = sliced_sim.Sim(immediate=True)
    w = sliced_sim.SymWorld(lib, sim, world, depth, 4, lag, first_rank=rank, n_local=1, dist=dist)
    rounds = 6
    for r in range(rounds):
        assert w.step() == r
        if r == 2:
            w.flush()
    w.flush()
    rp = w.reps[rank]
    ok = (sorted(rp.computed) == [(r * world + rank, u) for r in range(rounds) for u in range(depth + 1)]
          and all(lvl == list(range(rounds * world)) for lvl in rp.levels))
    q.put((rank, ok, w.collectives, w.sched.gathers))
    dist.barrier()
    dist.destroy_process_group()


# (a gloo all-gather between 8 processes on 8 cores takes ~0.3 s: the 8-rank case keeps the depth small; depth 32 at
# world 8 and 16 runs in-process above, under the adversarial scheduler)
@pytest.mark.parametrize("world,depth,lag", [(2, 8, None), (4, 6, 2), (8, 10, None)])
def test_gloo_ranks_line_up(world, depth, lag):
    import socket
    sliced_sim.build_lib()               # once, before the ranks race to build it
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, depth, lag, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]      # 8 spawned interpreters import torch on as many cores
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert len({c for _, _, c, _ in res}) == 1          # the same number of collectives on every rank
    assert res[0][2] > 0


def test_small_schedules_exhaustively(lib):
    """every admissible (world <= 8, depth in 1..8, lag in 1..7) schedule, with flushes at changing places, under the
    adversarial stream scheduler: the corner cases of the tick arithmetic (a world larger than the tree is deep, a lag
    longer than a slice, rounds that overlap four deep or not at all)"""
    done = 0
    for world in range(1, 9):
        for depth in (1, 2, 3, 5, 8):
            for lag in (1, 2, 3, 4, 7):
                try:
                    sliced_sim.Schedule(lib, world, depth + 1, lag)
                except ValueError:
                    continue
                sim = sliced_sim.Sim(immediate=False, seed=world * 100 + depth * 10 + lag)
                w = sliced_sim.SymWorld(lib, sim, world, depth, 2, lag)
                rounds = 6
                for r in range(rounds):
                    w.step()
                    if (r + lag) % 3 == 0:
                        w.flush()
                w.flush()
                for rank, rp in w.reps.items():
                    assert sorted(rp.computed) == [(r * world + rank, q) for r in range(rounds) for q in range(depth + 1)]
                    assert all(lvl == list(range(rounds * world)) for lvl in rp.levels), (world, depth, lag)
                w.close()
                done += 1
    assert done > 150


def test_a_test_switch_does_not_compile_into_a_product_build(tmp_path):
    """IMT_SCHED_MUTATION / IMT_SCHED_ALL_WAITS / IMT_SLICED_ROUNDS_BUILD change the shipped schedule.  A
    build that defines one of them without IMT_TEST_BUILD (a stray CXXFLAGS) must fail at the header, not ship."""
    hdr = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc", "imt_sliced_sched.hpp")
    src = tmp_path / "x.cpp"
    src.write_text(f'#include "{hdr}"\nint main() {{ return imt::sliced::ROUNDS; }}\n')

    def compiles(*defs):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *defs, str(src)], capture_output=True, text=True)
        return r.returncode == 0, r.stderr
    assert compiles()[0]
    assert compiles("-DIMT_SCHED_MUTATION=0", "-DIMT_SLICED_ROUNDS_BUILD=4")[0]        # the product's own values are not a switch
    for d in ("-DIMT_SCHED_MUTATION=3", "-DIMT_SCHED_ALL_WAITS", "-DIMT_SLICED_ROUNDS_BUILD=6"):
        ok, err = compiles(d)
        assert not ok and "test switches" in err, (d, err)
        assert compiles(d, "-DIMT_TEST_BUILD")[0], d
