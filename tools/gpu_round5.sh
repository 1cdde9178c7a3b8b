#!/bin/bash
# GPU sessions of round 5 (one 1-GPU box per call): pick the part with $1.  Steps are chained with && and carry their own
# timeouts; everything lands under gpurun_out/r05/ (summaries are copied into profiles/ by hand or tools/collect_profiles.sh).
#   qmap      tools/microbench/queue_map_probe: which streams share a hardware queue (the runtime's placement rule)
#   traces    rocprofv3 kernel traces of the sliced mode (1 / 2 / 4 in-process replicas, one emulated rank of 8), compacted
#   poolstrace  the same for one emulated rank of 8 with the library's default for one process per GPU (three priority pools)
#   prio / couple / own8 / rehearse / polled4 / longsoak / issue   the round's experiments (see the case labels)
#   tests     the whole -m gpu suite
#   bench     bench.py N = 1 + rocprofv3 kernel stats + PMC passes for k_sweep
#   multi     2- and 4-process rehearsals of bench.py --gpus N on the one GPU (IPC transport) + in-process replicas
#   emu       one rank of an N = 2 / 4 / 8 single-list run alone on the GPU over a modelled transport
#   aux       latency tables, secondary rates, kernel resources
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
part=${1:-tests}
trace() {   # trace NAME -- cmd...: kernel trace of cmd, compacted to $O/NAME_trace.csv.gz
  local name=$1; shift; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/raw_$name -o $name -- "$@" > $O/${name}_under_rocprof.txt 2> $O/${name}_rocprof.err &&
  python tools/trace_compact.py $O/raw_$name $O/${name}_trace.csv.gz && rm -rf $O/raw_$name
}
case $part in
qmap)
  timeout -k 10 120 tools/microbench/queue_map_probe > $O/queue_map_probe.txt 2>&1 && echo "probe ok" &&
  ( export GPU_MAX_HW_QUEUES=8; timeout -k 10 120 tools/microbench/queue_map_probe > $O/queue_map_probe_8q.txt 2>&1 ) && echo "probe (8 queues) ok"
  cat $O/queue_map_probe.txt ;;
traces)
  timeout -k 10 400 python tools/sliced_costs.py 1 2 4 > $O/sliced_costs.txt 2>&1 && echo "costs ok" &&
  trace sliced1 -- python3 tools/sliced_costs.py 1 && echo "trace 1 ok" &&
  trace sliced2 -- python3 tools/sliced_costs.py 2 && echo "trace 2 ok" &&
  trace sliced4 -- python3 tools/sliced_costs.py 4 && echo "trace 4 ok" &&
  ( export EMU_RANKS=first EMU_LINK_GBPS=0 EMU_ROUNDS=12; trace emu8 -- python3 tools/rank_emulation.py 8 ) && echo "trace emu8 ok" &&
  ( export EMU_RANKS=first EMU_ROUNDS=12; timeout -k 10 300 python tools/rank_emulation.py 8 > $O/rank_emulation_first8.txt 2>&1 ) && echo "emu ok"
  echo "traces part exit $?"; cat $O/sliced_costs.txt ;;
poolstrace)  # one rank of 8 with the three priority pools (the library's default for one process per GPU), modelled links: the kernel trace
  ( export EMU_RANKS=first EMU_ROUNDS=12; trace emu8_pools -- python3 tools/rank_emulation.py 8 ) && echo "trace emu8 pools ok"
  tail -3 $O/emu8_pools_under_rocprof.txt | cut -c1-160 ;;
prio)   # where the collectives' streams live: the normal pool (sharing the rounds' four queues) or the low-priority pool (four queues of their own)
  for cfg in "0 comm" "1 comm" "1 round" "0 round" "1 side"; do set -- $cfg
    ( export EMU_RANKS=first EMU_ROUNDS=12 IMT_SLICED_COMM_PRIO=$1 IMT_SLICED_PREP_STREAM=$2; echo "== comm prio $1, prep on $2"; timeout -k 10 200 python tools/rank_emulation.py 8 4 2>&1 | grep "^N =" | cut -c1-120 ) >> $O/emu_comm_prio.txt || break
  done
  cat $O/emu_comm_prio.txt ;;
couple)  # the only experiment with REAL peer coupling a one-GPU box offers: 2 (and 4) processes over IPC, the collectives'
         # streams on their rounds' queues (default) or in the low-priority pool (queues of their own), host- and GPU-polled
  for cfg in "0 1" "1 1" "0 0" "1 0"; do set -- $cfg
    ( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo IMT_SLICED_COMM_PRIO=$1 IMT_IPC_HOST_POLL=$2 IMT_BENCH_MODE=single-list
      echo "== comm prio $1 (1 = low-priority pool: own queues), IPC host poll $2"
      timeout -k 10 300 python3 bench.py --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline 2> $O/couple_$1_$2.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2 value', d['value'], 'verified', d['verified'], d['modes']['single_list'].get('schedule',{}).get('queue_map'))" ) >> $O/couple.txt 2>&1 || break
  done
  cat $O/couple.txt ;;
own8)   # the collectives' streams on queues of their own in the NORMAL pool (GPU_MAX_HW_QUEUES=8): one emulated rank, and the
        # placement with 0 .. 3 streams created before the world
  ( export GPU_MAX_HW_QUEUES=8 EMU_RANKS="first last" EMU_ROUNDS=12; echo "== GPU_MAX_HW_QUEUES=8, collectives on queues of their own"; timeout -k 10 400 python tools/rank_emulation.py 8 4 2 2>&1 | grep "^N =" | cut -c1-130 ) > $O/emu_own_queues.txt &&
  ( export GPU_MAX_HW_QUEUES=8 IMT_SLICED_COMM_PLACEMENT=1 EMU_RANKS="first" EMU_ROUNDS=12; echo "== GPU_MAX_HW_QUEUES=8, collectives on their rounds' queues"; timeout -k 10 400 python tools/rank_emulation.py 8 4 2>&1 | grep "^N =" | cut -c1-130 ) >> $O/emu_own_queues.txt &&
  ( export EMU_RANKS="first last" EMU_ROUNDS=12; echo "== four queues (the runtime's default): collectives on their rounds' queues"; timeout -k 10 400 python tools/rank_emulation.py 8 4 2 2>&1 | grep "^N =" | cut -c1-130 ) >> $O/emu_own_queues.txt &&
  for d in 0 1 2 3; do timeout -k 10 120 python tools/placement_check.py $d 65536 12 >> $O/placement_rates.txt 2>/dev/null || break; done &&
  for d in 0 1 2 3; do ( export GPU_MAX_HW_QUEUES=8; timeout -k 10 120 python tools/placement_check.py $d 65536 12 >> $O/placement_rates.txt 2>/dev/null ) || break; done
  cat $O/emu_own_queues.txt $O/placement_rates.txt ;;
rehearse)  # what changed the multi-process rehearsals on the one GPU since round 4: the watchdog's polling waits, eight queues
  for cfg in "8 120000" "4 120000" "8 0" "4 0"; do set -- $cfg
    for n in 2 4; do
      ( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo GPU_MAX_HW_QUEUES=$1 IMT_BENCH_LIBRARY_WATCHDOG_S=$(( $2 / 1000 )) IMT_BENCH_MODE=single-list
        echo "== N=$n GPU_MAX_HW_QUEUES=$1 watchdog $2 ms"
        timeout -k 10 400 python3 bench.py --gpus $n --steps 12 --warmup 3 --no-cpu-baseline 2> $O/rehearse.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['modes']['single_list']; print('value', round(d['value']/1e6,3), 'verified', d['verified'], s['schedule']['queue_map'][1], 'host wait ms', round(s.get('host_wait_ms_per_step',0),1), 'issue', round(s['host_call_ms_per_step'],1))" ) >> $O/rehearse.txt 2>&1 || break 2
    done
  done
  cat $O/rehearse.txt ;;
polled4)  # the 4-process GPU-polled rehearsal collapses because a waiting kernel holds a round's queue -- then
          # it should recover when the collectives' streams have queues of their own (low-priority pool; four queues per process)
  for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
    ( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo IMT_SLICED_COMM_PRIO=$1 IMT_IPC_HOST_POLL=$2 IMT_BENCH_MODE=single-list
      echo "== N=4, collectives' streams priority $1 (1 = low-priority pool: queues of their own), IPC host poll $2"
      timeout -k 10 400 python3 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline 2> $O/polled4.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['modes']['single_list']; print('value', round(d['value']/1e6,3), 'verified', d['verified'], s['schedule']['queue_map'][1])" ) >> $O/polled4.txt 2>&1 || break
  done
  cat $O/polled4.txt ;;
longsoak)
  SOAK_SECONDS=420 timeout -k 10 600 python tools/sliced_soak.py > $O/sliced_soak_420s.txt 2>&1 && echo "sliced soak ok" &&
  SOAK_SECONDS=240 timeout -k 10 400 python tools/differential_soak.py > $O/differential_soak_240s.txt 2>&1 && echo "soak ok"
  tail -2 $O/sliced_soak_420s.txt $O/differential_soak_240s.txt ;;
issue)   # when does the host ISSUE a step's preparation, and when does it run? (rocprofv3 kernel + HIP runtime trace of one emulated rank of 8)
  ( export EMU_RANKS=first EMU_LINK_GBPS=0 EMU_ROUNDS=6; timeout -k 10 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/raw_issue -o issue -- python3 tools/rank_emulation.py 8 > $O/issue_under_rocprof.txt 2> $O/issue_rocprof.err ) &&
  python tools/trace_issue_lag.py $O/raw_issue "prep::k_scatter" "k_convert" > $O/issue_lag.txt 2>&1; rm -rf $O/raw_issue
  cat $O/issue_lag.txt | head -60 ;;
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
  tail -4 $O/tests.log ;;
bench|multi|scale|soak|emu|aux)   # the standing parts: tools/gpu_round4.sh writing into this round's directory
  ROUND_DIR=$O bash tools/gpu_round4.sh $part ;;
esac
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
