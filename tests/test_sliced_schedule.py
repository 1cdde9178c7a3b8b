"""CPU: the multi-GPU single-list driver (indexed-merkle-tree-halo2_amd/sliced.py) with a symbolic backend
(tests/sliced_sim.py).  What is under test is everything that is not hashing: the systolic schedule, which payload is
applied where and when, the events between rounds, buffer reuse, the line-up of the collectives across ranks.

 * LocalWorld, deferred execution in random interleavings (only stream order and event waits are respected): every
   slice's every level must see exactly the slices before it -- the rule that makes the replicas equal the reference's
   sequential list (src/indexed_merkle_tree.rs:632-660), at world 1..16, several depths, lags and seeds.
 * DistTransport over gloo with 2 and 4 processes: the same assertions, real collectives.
The GPU tests (tests/test_gpu_sliced.py) run the same driver with the HIP backend against the one-GPU tree."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_sliced():
    import importlib.util
    path = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "sliced.py")
    spec = importlib.util.spec_from_file_location("imt_sliced", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)         # sliced.py itself does not need the HIP library
    return mod


def test_schedule_arithmetic():
    sl = load_sliced()
    for world in (1, 2, 3, 4, 8, 16):
        for units in (4, 9, 33):
            sc = sl.SliceSchedule(world, units)
            assert -(-sc.round_ticks // sc.period) <= sc.STREAMS
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
        sl.SliceSchedule(1, 33, lag=2)       # 17 rounds in flight
    sc = sl.SliceSchedule(8, 33)
    assert (sc.lag, sc.period, sc.gathers) == (2, 16, 47)


@pytest.mark.parametrize("world,depth,lag", [(1, 8, None), (2, 8, None), (2, 8, 5), (4, 8, None), (4, 32, None), (8, 32, None),
                                             (8, 32, 4), (3, 8, None), (16, 32, None), (8, 8, 1), (2, 3, 2)])
def test_every_level_sees_exactly_the_earlier_slices(world, depth, lag):
    import sliced_sim
    sl = load_sliced()
    for seed in range(int(os.environ.get("IMT_SIM_SEEDS", "4"))):
        sim = sliced_sim.Sim(immediate=False, seed=seed)
        bes = [sliced_sim.SymbolicBackend(sim, depth, 4, world, g) for g in range(world)]
        w = sl.LocalWorld(bes, lag)
        rounds = 7
        for r in range(rounds):
            assert w.step([FakeVals(4 * world)] * world) == r
            if seed % 2 and r == 3:
                sim.run()                    # a caller that synchronises in the middle
            if seed % 4 == 2 and r in (1, 4):
                w.flush()                    # ... or runs the schedule dry and goes on (bench.py: warm-up, then the timed region)
        w.flush()
        for be in bes:
            assert sorted(be.computed) == [(r * world + be.rank, q) for r in range(rounds) for q in range(depth + 1)]
            for lvl in be.levels:
                assert lvl == list(range(rounds * world))      # every replica holds every slice's write-back, in order
        assert w.tp.collectives > 0 or world == 1


class FakeVals:
    def __init__(self, n):
        self.shape = (n, 32)


def _worker(rank, world, port, depth, lag, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sliced_sim
    sl = load_sliced()
    sim = sliced_sim.Sim(immediate=True)
    be = sliced_sim.SymbolicBackend(sim, depth, 4, world, rank)
    tree = sl.SlicedIndexedTree(be, world, rank, sl.DistTransport(dist, via_host=True), lag)
    rounds = 6
    for r in range(rounds):
        assert tree.step(FakeVals(4 * world)) == r
        if r == 2:
            tree.flush()
    tree.flush()
    ok = (sorted(be.computed) == [(r * world + rank, u) for r in range(rounds) for u in range(depth + 1)]
          and all(lvl == list(range(rounds * world)) for lvl in be.levels))
    q.put((rank, ok, tree.tp.collectives, tree.sched.gathers))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,depth,lag", [(2, 8, None), (4, 6, 2), (8, 32, None)])
def test_gloo_ranks_line_up(world, depth, lag):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, depth, lag, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert len({c for _, _, c, _ in res}) == 1          # the same number of collectives on every rank


def test_small_schedules_exhaustively():
    """every admissible (world <= 8, depth in 1..8, lag in 1..7) schedule, with flushes at changing places, under the
    adversarial stream scheduler: the corner cases of the tick arithmetic (a world larger than the tree is deep, a lag
    longer than a slice, rounds that overlap four deep or not at all)"""
    import sliced_sim
    sl = load_sliced()
    done = 0
    for world in range(1, 9):
        for depth in (1, 2, 3, 5, 8):
            for lag in (1, 2, 3, 4, 7):
                try:
                    sl.SliceSchedule(world, depth + 1, lag)
                except ValueError:
                    continue
                sim = sliced_sim.Sim(immediate=False, seed=world * 100 + depth * 10 + lag)
                bes = [sliced_sim.SymbolicBackend(sim, depth, 2, world, g) for g in range(world)]
                w = sl.LocalWorld(bes, lag)
                rounds = 6
                for r in range(rounds):
                    w.step([FakeVals(2 * world)] * world)
                    if (r + lag) % 3 == 0:
                        w.flush()
                w.flush()
                for be in bes:
                    assert sorted(be.computed) == [(r * world + be.rank, q) for r in range(rounds) for q in range(depth + 1)]
                    assert all(lvl == list(range(rounds * world)) for lvl in be.levels), (world, depth, lag)
                done += 1
    assert done > 150
