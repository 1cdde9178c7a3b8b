"""bench.py's supervisor (the GPU-free process every launched rank is): the retry logic over attempts
in FRESH worker processes, exercised without a GPU -- two supervisors over a rendezvous store, the worker replaced by
tests/helpers/fake_bench_worker.py.  What must hold: `value` comes from the first attempt in which EVERY rank's worker
succeeded and rank 0's line verified; a worker that hangs because its peer died is killed by its own supervisor (exact
process group); the line carries `attempts`; the exit status is non-zero on every rank unless an attempt verified."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "helpers", "fake_bench_worker.py")
DRIVER = ("import sys, argparse; sys.path.insert(0, %r); import bench; "
          "a = argparse.Namespace(gpus=2, steps=2, warmup=1, no_cpu_baseline=True); "
          "sys.exit(bench.supervise(a, worker_cmd=[sys.executable, %r]))" % (ROOT, FAKE))


def run_supervisors(tmp_path, scenario, extra=None, world=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FAKE_SCENARIO=scenario, FAKE_LOG_DIR=str(tmp_path), IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_PEER_GRACE="1",
                   TORCHELASTIC_USE_AGENT_STORE="False", **(extra or {}))
        for k in ("IMT_BENCH_WORKER", "IMT_BENCH_ATTEMPTS", "IMT_BENCH_SLICED_TRANSPORT", "IMT_BENCH_MODE"):
            if k not in (extra or {}):
                env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, "-c", DRIVER.replace("gpus=2", f"gpus={world}")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    return [p.returncode for p in procs], outs, port


def the_line(outs):
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0]
    assert not any(l.startswith("{") for l in outs[1][0].splitlines())
    return json.loads(lines[0])


def test_second_attempt_in_fresh_workers(tmp_path):
    rcs, outs, port = run_supervisors(tmp_path, "second_attempt", {"IMT_BENCH_ATTEMPTS": "rccl:pools,ipc:pools,rccl:one-pool"})
    assert rcs == [0, 0], outs
    res = the_line(outs)
    assert res["value"] == 123.0 and res["verified"] is True
    at = res["attempts"]
    assert [a["outcome"] for a in at] == ["failed", "verified"]             # the third of the plan is never started
    assert at[0]["asked_for"] == "rccl" and at[1]["transport"] == "ipc" and at[1]["layout"] == "pools"
    assert at[0]["exit_status"][1] == 3 and at[0]["exit_status"][0] != 0       # rank 0's hung worker was killed, not waited for
    assert at[0]["seconds"] < 60 and "fake dump" in at[0]["dump_tail"]
    assert "rank 1: exit status 3" in at[0]["why"] and "a peer's worker had failed" in at[0]["why"]
    assert "collective NOT COMPLETE" in outs[1][1]                           # the worker's stderr is relayed as it comes
    w0, w1 = (json.load(open(tmp_path / f"worker_a{k}_r0.json")) for k in (0, 1))
    for w in (w0, w1):
        assert w["IMT_BENCH_WORKER"] == "1" and w["TORCHELASTIC_USE_AGENT_STORE"] is None and w["WORLD_SIZE"] == "2"
        assert int(w["MASTER_PORT"]) != port                                # the workers rendezvous on a port of their own ...
    assert w0["MASTER_PORT"] != w1["MASTER_PORT"]                            # ... a fresh one per attempt,
    assert json.load(open(tmp_path / "worker_a1_r1.json"))["MASTER_PORT"] == w1["MASTER_PORT"]      # the same on every rank
    assert not (tmp_path / "worker_a2_r0.json").exists()


def test_every_attempt_fails(tmp_path):
    rcs, outs, _ = run_supervisors(tmp_path, "all_fail")
    assert rcs == [1, 1]
    res = the_line(outs)
    assert res["value"] is None and res["verified"] is False and res["ms_per_step"] is None
    assert [a["asked_for"] + ":" + a["layout"] for a in res["attempts"]] == ["ipc:pools", "ipc:one-pool"]      # the gloo rehearsal's plan
    assert all(a["outcome"] == "failed" and "-13" in a["why"] and "world state" in a["dump_tail"] for a in res["attempts"])
    assert "-13 in attempt 1" in res["value_failed"]
    # the other leg was never reached (a worker runs the headline leg first and leaves when it fails): the supervisors
    # measure it at the end in workers of its own, and its figure goes under `modes` only
    assert json.load(open(tmp_path / "worker_a0_r1.json"))["IMT_BENCH_MODE"] is None
    assert json.load(open(tmp_path / "worker_a1_r1.json"))["IMT_BENCH_MODE"] is None
    assert json.load(open(tmp_path / "worker_a3_r1.json"))["IMT_BENCH_MODE"] == "subtrees"      # (index 2 = the exploratory attempt's)
    assert res["modes"]["subtrees"]["value"] == 5.0e6 and "its own workers" in res["modes"]["subtrees"]["measured_in_attempt"]
    assert len(res["attempts"]) == 2


def test_an_unverified_line_is_a_failed_attempt_and_an_explicit_transport_is_one_attempt(tmp_path):
    rcs, outs, _ = run_supervisors(tmp_path, "unverified_then_ok")
    assert rcs == [0, 0]
    res = the_line(outs)
    assert res["value"] == 77.0 and [a["outcome"] for a in res["attempts"]] == ["failed", "verified"]
    assert res["attempts"][0]["why"] == "did not verify" and res["attempts"][0]["exit_status"] == [0, 0]
    sub = tmp_path / "one"
    sub.mkdir()
    rcs, outs, _ = run_supervisors(sub, "unverified_then_ok", {"IMT_BENCH_SLICED_TRANSPORT": "stall"})
    assert rcs == [1, 1]
    res = the_line(outs)
    assert res["value"] is None and len(res["attempts"]) == 1 and res["attempts"][0]["asked_for"] == "stall"


def test_the_attempt_limit_ends_a_silent_hang(tmp_path):
    rcs, outs, _ = run_supervisors(tmp_path, "silent_hang", {"IMT_BENCH_ATTEMPT_TIMEOUT": "3"})
    assert rcs == [0, 0]
    res = the_line(outs)
    assert res["value"] == 9.0 and "killed by its supervisor after 3 s" in res["attempts"][0]["why"]


def test_plan():
    sys.path.insert(0, ROOT)
    import bench
    for k in ("IMT_BENCH_ATTEMPTS", "IMT_BENCH_SLICED_TRANSPORT"):
        os.environ.pop(k, None)
    assert bench.attempts_plan("nccl") == [("rccl", "pools"), ("ipc", "pools"), ("rccl", "one-pool"), ("ipc", "one-pool"), ("rccl", "one-comm")]
    assert bench.attempts_plan("gloo") == [("ipc", "pools"), ("ipc", "one-pool")]
    assert bench.attempts_plan("nccl", 1) == [("rccl", "pools"), ("local", "pools")] and bench.attempts_plan("gloo", 1) == [("local", "pools")]
    os.environ["IMT_BENCH_ATTEMPTS"] = "stall,local:one-pool"
    try:
        assert bench.attempts_plan("nccl") == [("stall", "pools"), ("local", "one-pool")]
        os.environ["IMT_BENCH_ATTEMPTS"] = "rccl:fast"
        with pytest.raises(SystemExit):
            bench.attempts_plan("nccl")
    finally:
        del os.environ["IMT_BENCH_ATTEMPTS"]


@pytest.mark.parametrize("sig", ["TERM", "KILL"])
def test_a_worker_does_not_outlive_its_supervisor(tmp_path, sig):
    """the launcher ends the job (its time limit, a failed peer): SIGTERM to the supervisor takes the worker's process group
    with it; a supervisor killed outright (SIGKILL) takes its worker along through PR_SET_PDEATHSIG -- no orphan keeps a GPU"""
    import signal
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAKE_SCENARIO="orphan",
               FAKE_LOG_DIR=str(tmp_path), IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_FORCE_DIST="1")
    for k in ("IMT_BENCH_WORKER", "IMT_BENCH_ATTEMPTS", "IMT_BENCH_SLICED_TRANSPORT", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    drv = DRIVER.replace("gpus=2", "gpus=1")
    sup = subprocess.Popen([sys.executable, "-c", drv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pidfile = tmp_path / "pid_r0"
    for _ in range(600):
        if pidfile.exists() and pidfile.read_text():
            break
        time.sleep(0.1)
    wpid = int(pidfile.read_text())
    os.kill(wpid, 0)                                   # the worker lives
    sup.send_signal(getattr(signal, "SIG" + sig))
    sup.wait(timeout=30)
    assert sup.returncode != 0
    for _ in range(100):
        try:
            os.kill(wpid, 0)
        except ProcessLookupError:
            break
        # (a zombie re-parented to init still answers kill 0 until it is reaped: look at its state)
        try:
            if open(f"/proc/{wpid}/stat").read().split()[2] == "Z":
                break
        except FileNotFoundError:
            break
        time.sleep(0.1)
    else:
        os.kill(wpid, signal.SIGKILL)
        raise AssertionError("the worker outlived its supervisor")


def test_eight_supervisors(tmp_path):
    """the driver's largest form: eight ranks.  Attempt 0: rank 1's worker dies, the seven others hang and are killed by
    their supervisors; attempt 1 verifies; every supervisor exits 0 and only rank 0 prints."""
    rcs, outs, _ = run_supervisors(tmp_path, "second_attempt", {"IMT_BENCH_ATTEMPTS": "rccl:pools,ipc:pools"}, world=8)
    assert rcs == [0] * 8, [o[1][-300:] for o in outs]
    res = the_line(outs)
    assert res["value"] == 123.0 and [a["outcome"] for a in res["attempts"]] == ["failed", "verified"]
    st = res["attempts"][0]["exit_status"]
    assert len(st) == 8 and st[1] == 3 and all(x != 0 for x in st)
    ports = {json.load(open(tmp_path / f"worker_a1_r{r}.json"))["MASTER_PORT"] for r in range(8)}
    assert len(ports) == 1


@pytest.mark.parametrize("scenario,want_value,want_from", [("explore_better", 120.0, 2), ("explore_worse", 100.0, 0), ("explore_fails", 100.0, 0)])
def test_one_look_at_the_other_stream_layout(tmp_path, scenario, want_value, want_from):
    """After a quick verified attempt the same transport is measured once in the OTHER stream layout (fresh workers, the
    single list alone); `value` is the better of the two verified figures, both are on the line, and a failing exploratory
    attempt costs nothing.  (On by default with RCCL; asked for here.)"""
    rcs, outs, _ = run_supervisors(tmp_path, scenario, {"IMT_BENCH_EXPLORE": "1"})
    assert rcs == [0, 0], outs
    res = the_line(outs)
    at = res["attempts"]
    assert len(at) == 2 and at[0]["outcome"] == "verified" and at[0]["value"] == 100.0 and at[1].get("exploratory") is True
    assert (at[0]["layout"], at[1]["layout"]) == ("pools", "one-pool") and at[1]["attempt"] == 2
    assert res["value"] == want_value and res["value_from_attempt"] == want_from and res["verified"] is True
    assert at[1]["outcome"] == ("failed" if scenario == "explore_fails" else "verified")
    assert res["modes"]["subtrees"]["value"] == 5.0e6                       # measured once, in the first verified attempt
    w = json.load(open(tmp_path / "worker_a2_r1.json"))
    assert w["IMT_BENCH_MODE"] == "single-list" and w["IMT_BENCH_LAYOUT"] == "one-pool" and w["IMT_BENCH_SLICED_TRANSPORT"] == "ipc"


def test_no_exploration_when_the_plan_was_given_or_time_is_short(tmp_path):
    rcs, outs, _ = run_supervisors(tmp_path, "explore_better", {"IMT_BENCH_EXPLORE": "1", "IMT_BENCH_EXPLORE_WITHIN": "0"})
    assert rcs == [0, 0] and len(the_line(outs)["attempts"]) == 1
    sub = tmp_path / "given"
    sub.mkdir()
    rcs, outs, _ = run_supervisors(sub, "explore_better", {"IMT_BENCH_EXPLORE": "1", "IMT_BENCH_ATTEMPTS": "ipc:pools"})
    assert rcs == [0, 0] and len(the_line(outs)["attempts"]) == 1 and the_line(outs)["value"] == 100.0
