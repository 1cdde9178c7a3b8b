#!/bin/bash
# GPU sessions of round 6 (one 1-GPU box per call): pick the part with $1.  Steps are chained with && and carry their own
# timeouts; everything lands under gpurun_out/r06/ (summaries are copied into profiles/ by tools/collect_profiles.sh or by hand).
#   newtests  this round's new / changed GPU tests (bench supervisor + preflight, destroy after a timeout, sticky gather error,
#             pool presets) and the files they live in
#   orders    tools/value_orders.py: throughput on ordered inputs at config-2 size
#   orderprof rocprofv3 kernel stats of one order ($2)
#   tests     the whole -m gpu suite
#   bench     bench.py N = 1 + rocprofv3 kernel stats + PMC passes for k_sweep
#   multi     2- and 4-process rehearsals of bench.py --gpus N on the one GPU (supervisors + workers, IPC transport)
#   emu       one rank of N = 2 / 4 / 8 alone on the GPU: memcpy fill vs the RCCL-shaped gather kernel (+ CU-masked streams)
#   flag      A/B: device-side flag between a gather and its apply instead of the cross-queue event wait
set -o pipefail
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
part=${1:-tests}
case $part in
newtests)
  timeout -k 10 1100 python -m pytest tests/test_gpu_sharded_procs.py tests/test_gpu_sliced.py -m gpu -x -q --durations=15 > $O/newtests.log 2>&1
  rc=$?; tail -40 $O/newtests.log; exit $rc ;;
orders)
  timeout -k 10 900 python tools/value_orders.py --steps 12 --warmup 3 > $O/value_orders.txt 2> $O/value_orders.err
  rc=$?; cat $O/value_orders.txt; tail -5 $O/value_orders.err; exit $rc ;;
orderprof)
  cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$2 -o $2 -- python3 $GRAFT_REPO_ROOT/tools/value_orders.py --only $2 --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/$O/orderprof_$2.txt 2>&1
  rc=$?; cd $GRAFT_REPO_ROOT; find $O/prof_$2 -name "*kernel_stats.csv" -exec cp {} $O/order_$2_kernel_stats.csv \; ; head -25 $O/order_$2_kernel_stats.csv | cut -c1-200; exit $rc ;;
tests)
  timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=25 > $O/gpu_tests.log 2>&1
  rc=$?; tail -45 $O/gpu_tests.log; exit $rc ;;
emu)   # one rank of N alone on the GPU under the production layout (three priority pools): today's fill (a one-wave sleep + N blits)
       # against ONE RCCL-shaped kernel per collective (7 / 28 / 56 workgroups x 512 lanes polling a flag, then copying)
  ( export EMU_RANKS=first EMU_ROUNDS=16
    for fill in "memcpy 0" "rccl 7" "rccl 28" "rccl 56" "memcpy 0"; do set -- $fill
      echo "== fill $1, $2 workgroups"
      EMU_FILL=$1 EMU_GATHER_WGS=$2 timeout -k 10 300 python tools/rank_emulation.py 8 4 2 2>&1 | grep "^N =" | cut -c1-150 || exit 1
    done ) > $O/emu_rccl_shape.txt 2>&1
  rc=$?; cat $O/emu_rccl_shape.txt; exit $rc ;;
flag)  # "the gather is complete" as a device-side counter polled by a one-wave kernel on the round's
       # stream (IMT_SLICED_GATHER_FLAGS=1) instead of an event wait across hardware queues.  Correctness first, then one rank
       # of 8 / 4 (production layout, both fills), the two forms alternating on ONE box.
  ( IMT_SLICED_GATHER_FLAGS=1 timeout -k 10 600 python -m pytest tests/test_gpu_sliced.py -m gpu -x -q -k "sequential_oracle or equals_one_gpu_tree or bench_size or over_ipc or pool_presets" > $O/flag_tests.log 2>&1 ) &&
  tail -3 $O/flag_tests.log &&
  ( export EMU_RANKS=first EMU_ROUNDS=16
    for rep in 1 2 3; do for fl in 0 1; do for fill in memcpy rccl; do
      echo "== rep $rep: gather flags $fl, fill $fill"
      IMT_SLICED_GATHER_FLAGS=$fl EMU_FILL=$fill EMU_GATHER_WGS=28 timeout -k 10 300 python tools/rank_emulation.py 8 4 2>&1 | grep "^N =" | grep links | cut -c1-110 || exit 1
    done; done; done ) > $O/ab_gather_flags.txt 2>&1
  rc=$?; cat $O/ab_gather_flags.txt; exit $rc ;;
emuslack)   # how much later than the link model may a collective complete before one rank of 8 / 4 slows down?  (the slowest peer's
            # kernel waiting for wave slots: up to ~450 us in profiles/r05_barrier_wakeup.txt; the schedule consumes a gather lag ticks later)
  ( export EMU_RANKS=first EMU_ROUNDS=16 EMU_FILL=rccl EMU_GATHER_WGS=28
    for extra in 0 100 200 450 900 1500 2500 0; do
      echo "== every collective completes $extra us later than 40 us + bytes / 48 GB/s"
      EMU_EXTRA_WAIT_US=$extra timeout -k 10 300 python tools/rank_emulation.py 8 4 2>&1 | grep "^N =" | grep links | cut -c1-110 || exit 1
    done ) > $O/emu_slack.txt 2>&1
  rc=$?; cat $O/emu_slack.txt; exit $rc ;;
emulag)   # the same sensitivity with a longer lag (more ticks between a gather and its use, fewer rounds in flight): what to try on
          # hardware if the collectives turn out to be late (IMT_BENCH_LAG)
  ( export EMU_RANKS=first EMU_ROUNDS=16 EMU_FILL=rccl EMU_GATHER_WGS=28
    for lag in 2 3 4; do for extra in 0 900 1500 2500; do
      echo "== lag $lag, every collective $extra us late"
      EMU_LAG=$lag EMU_EXTRA_WAIT_US=$extra timeout -k 10 300 python tools/rank_emulation.py 8 2>&1 | grep "^N =" | grep links | cut -c1-110 || exit 1
    done; done ) > $O/emu_lag_slack.txt 2>&1
  rc=$?; cat $O/emu_lag_slack.txt; exit $rc ;;
arena)   # the IPC transport's send arena in uncached device memory (the default now) against ordinary hipMalloc memory: the IPC GPU
         # tests, then the 2- and 4-process rehearsals on the one GPU with both, alternating
  timeout -k 10 600 python -m pytest tests/test_gpu_sliced.py tests/test_gpu_sharded_procs.py -m gpu -x -q -k "ipc or vanished or gather_reports or two_ranks or c_example or gpus2 or gpus4" > $O/arena_tests.log 2>&1 &&
  tail -3 $O/arena_tests.log &&
  ( for rep in 1 2; do for kind in uncached default; do for n in 2 4; do
      echo "== rep $rep: arena $kind, N = $n"
      ( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo IMT_BENCH_MODE=single-list IMT_IPC_ARENA=$kind IMT_SLICED_TIMING=1
        timeout -k 10 400 python3 bench.py --gpus $n --steps 12 --warmup 3 --no-cpu-baseline 2> $O/arena.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(d['value']/1e6,3), 'verified', d['verified'], 'attempts', len(d['attempts']))" ) || exit 1
      grep -o "arena in [a-z-]* memory" $O/arena.err | sort | uniq -c
    done; done; done ) > $O/ab_ipc_arena.txt 2>&1
  rc=$?; cat $O/ab_ipc_arena.txt; exit $rc ;;
bench|multi|scale|soak|aux)   # the standing parts: tools/gpu_round4.sh writing into this round's directory
  ROUND_DIR=$O bash tools/gpu_round4.sh $part ;;
*) echo "unknown part $part"; exit 2 ;;
esac
