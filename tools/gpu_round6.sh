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
*) echo "unknown part $part"; exit 2 ;;
esac
