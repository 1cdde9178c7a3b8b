#!/bin/bash
# Repeats the typed multi-process rehearsal (launcher -> torch.distributed.run -> GPU-free supervisors -> workers on ONE GPU) to
# look for flakiness in the process orchestration itself (ports, the rendezvous store, worker start / teardown).  Stops at the
# first run that fails.   bash tools/supervisor_soak.sh [runs-of-2] [runs-of-4]
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo IMT_BENCH_NO_TRACE=1 IMT_BENCH_EXPLORE=1
n2=${1:-8}; n4=${2:-4}
: > $O/supervisor_soak.txt
for i in $(seq 1 $n2); do
  timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2> $O/supervisor_soak.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['attempts']; assert d['verified'] and d['value'] and all(x['outcome']=='verified' for x in a), d; print('N=2 run', $i, 'value', round(d['value']/1e6,3), 'attempts', [(x['layout'], round(x['value']/1e6,3)) for x in a], 'from', d['value_from_attempt'])" >> $O/supervisor_soak.txt || { echo "N=2 run $i FAILED" >> $O/supervisor_soak.txt; tail -30 $O/supervisor_soak.err; exit 1; }
done
for i in $(seq 1 $n4); do
  timeout -k 10 300 python3 bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/supervisor_soak.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['attempts']; assert d['verified'] and d['value'] and all(x['outcome']=='verified' for x in a), d; print('N=4 run', $i, 'value', round(d['value']/1e6,3), 'attempts', [(x['layout'], round(x['value']/1e6,3)) for x in a], 'from', d['value_from_attempt'])" >> $O/supervisor_soak.txt || { echo "N=4 run $i FAILED" >> $O/supervisor_soak.txt; tail -30 $O/supervisor_soak.err; exit 1; }
done
cat $O/supervisor_soak.txt
