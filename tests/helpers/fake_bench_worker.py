"""Stand-in for bench.py's worker process in tests/test_bench_supervisor.py: no GPU, no torch.  Behaves as the scenario in
FAKE_SCENARIO says for (IMT_BENCH_ATTEMPT, RANK) and records the environment it was started with."""
import json
import os
import sys
import time

rank, attempt = int(os.environ["RANK"]), int(os.environ["IMT_BENCH_ATTEMPT"])
kind, layout = os.environ.get("IMT_BENCH_SLICED_TRANSPORT", "-"), os.environ.get("IMT_BENCH_LAYOUT", "-")
scenario = os.environ["FAKE_SCENARIO"]
with open(os.path.join(os.environ["FAKE_LOG_DIR"], f"worker_a{attempt}_r{rank}.json"), "w") as f:
    json.dump({k: os.environ.get(k) for k in ("IMT_BENCH_WORKER", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE", "IMT_BENCH_MODE",
                                              "WORLD_SIZE", "LOCAL_RANK", "IMT_BENCH_SLICED_TRANSPORT", "IMT_BENCH_LAYOUT")}, f)

SUB = {"value": 5.0e6, "verified": True}


def line(value, why=None, subtrees=True):
    d = {"metric": "m", "value": value, "verified": value is not None, "ms_per_step": 1.0 if value else None, "n_gpus": 2,
         "modes": {"single_list": {"schedule": {"transport": kind, "pools": 1 if layout == "pools" else 0, "comm_streams": 4}}}}
    if why:
        d["value_failed"] = why
    # (a real worker runs the single list FIRST: its line carries the subtree leg only when the headline leg did not fail)
    if subtrees and os.environ.get("IMT_BENCH_MODE") != "single-list":
        d["modes"]["subtrees"] = SUB
    print(json.dumps(d), flush=True)


if scenario == "second_attempt":
    if attempt == 0:
        if rank == 1:
            print("rank 1: collective NOT COMPLETE on channel 2 (fake dump)", file=sys.stderr, flush=True)
            sys.exit(3)
        time.sleep(120)              # rank 0 hangs in a collective its peer never joins: the supervisor must kill it
    if rank == 0:
        line(123.0)
    sys.exit(0)
if scenario == "all_fail":
    if os.environ.get("IMT_BENCH_MODE") == "subtrees":          # the supervisors' last run: the other leg alone, in workers of its own
        if rank == 0:
            line(5.0e6)
        sys.exit(0)
    if rank == 0:
        line(None, why=f"ImtError: -13 in attempt {attempt}", subtrees=False)
    print(f"rank {rank}: world state (fake)", file=sys.stderr, flush=True)
    sys.exit(1)
if scenario == "unverified_then_ok":     # every worker exits 0 but rank 0's line carries no value: still a failed attempt
    if rank == 0:
        line(None if attempt == 0 else 77.0, why="did not verify" if attempt == 0 else None)
    sys.exit(0)
if scenario.startswith("explore_"):      # attempt 0 verifies at 100; the exploratory attempt (other layout) is better / worse / fails
    if layout == "pools":
        if rank == 0:
            line(100.0)
        sys.exit(0)
    if scenario == "explore_fails":
        print(f"rank {rank}: preflight failed (fake)", file=sys.stderr, flush=True)
        sys.exit(1)
    if rank == 0:
        line(120.0 if scenario == "explore_better" else 90.0)
    sys.exit(0)
if scenario == "orphan":                 # hangs for good and says who it is: the test ends its supervisor and looks for it
    with open(os.path.join(os.environ["FAKE_LOG_DIR"], f"pid_r{rank}"), "w") as f:
        f.write(str(os.getpid()))
    time.sleep(300)
if scenario == "silent_hang":            # nobody fails, nobody finishes: the per-attempt limit ends it
    if attempt == 0:
        time.sleep(120)
    if rank == 0:
        line(9.0)
    sys.exit(0)
sys.exit(99)
