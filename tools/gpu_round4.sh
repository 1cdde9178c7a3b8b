#!/bin/bash
# GPU sessions of round 4 (one 1-GPU box per call): pick the parts with $1.  Steps are chained with && and carry their own
# timeouts; everything lands under gpurun_out/r04/ (tools/collect_profiles.sh copies the summaries into profiles/).
#   tests     the whole -m gpu suite
#   bench     bench.py N = 1 + rocprofv3 kernel stats (alone, pipelined) + PMC passes for k_sweep
#   multi     2- and 4-process rehearsals of bench.py --gpus N on the one GPU (IPC transport) + in-process replicas
#   scale     the timed path on trees of 2^20 .. 2^27 leaves (+ rocprofv3 at 2^26)
#   soak      differential soaks against the sequential oracle
#   aux       secondary rates and latency tables, kernel resources
#   emu       one rank of an N = 2 / 4 / 8 single-list run alone on the GPU over a modelled transport (timing emulation)
set -o pipefail
O=${ROUND_DIR:-gpurun_out/r04}
mkdir -p $O
export TMPDIR=/tmp
part=${1:-tests}
case $part in
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
  tail -4 $O/tests.log ;;
bench)
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err && echo "bench ok" &&
  ( export IMT_NO_PIPELINE=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_alone -o alone -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_alone_under_rocprof.json 2> $O/prof_alone.err ) && echo "prof alone ok" &&
  ( export IMT_BENCH_NO_ATTRIBUTION=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipe -o pipe -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_pipe_under_rocprof.json 2> $O/prof_pipe.err ) && echo "prof pipe ok" &&
  ( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err ) && echo "pmc fetch ok" &&
  ( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err ) && echo "pmc write ok" &&
  ( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_valu_alone -o v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_valu_alone.err ) && echo "pmc valu (alone) ok" &&
  ( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU --output-format csv -d $O/pmc_classes -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_classes.err ) && echo "pmc classes ok" &&
  ( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $O/pmc_other -o o -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_other.err ) && echo "pmc other ok" &&
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_bench_summary.txt && python tools/pmc_summary.py $O/pmc_valu_alone > $O/pmc_valu_alone_summary.txt && python tools/pmc_summary.py $O/pmc_classes $O/pmc_other > $O/pmc_classes_summary.txt
  echo "bench part exit $?" ;;
multi)
  ( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo; timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_rehearsal_ipc.json 2> $O/bench_2rank.err ) && echo "2-rank ok" &&
  ( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo; timeout -k 10 600 python3 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_4rank_rehearsal_ipc.json 2> $O/bench_4rank.err ) && echo "4-rank ok" &&
  timeout -k 10 400 python tools/sliced_costs.py 1 2 4 > $O/sliced_costs.txt 2>&1 && echo "sliced costs ok"
  echo "multi part exit $?" ;;
scale)
  timeout -k 10 1000 python tools/scale_check.py > $O/scale_check.txt 2>&1 && echo "scale ok" &&
  ( export SCALE_SIZES=26 SCALE_STEPS=20; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scale26 -o s26 -- python3 tools/scale_check.py > $O/scale_check_2pow26_under_rocprof.txt 2> $O/prof_scale26.err ) && echo "prof scale ok"
  echo "scale part exit $?"; tail -12 $O/scale_check.txt ;;
soak)
  SOAK_SECONDS=60 timeout -k 10 400 python tools/differential_soak.py > $O/differential_soak.txt 2>&1 && echo "soak ok" &&
  SOAK_SECONDS=120 timeout -k 10 400 python tools/sliced_soak.py > $O/sliced_soak.txt 2>&1 && echo "sliced soak ok"
  echo "soak part exit $?" ;;
emu)
  timeout -k 10 900 python tools/rank_emulation.py > $O/rank_emulation.txt 2>&1 && echo "emulation ok"
  echo "emu part exit $?"; grep "^N =" $O/rank_emulation.txt | cut -c1-110 ;;
aux)
  timeout -k 10 300 python tools/latency_vs_cpu.py > $O/latency_vs_cpu.txt 2>&1 && echo "latency vs cpu ok" &&
  timeout -k 10 300 python tools/bench_aux.py > $O/bench_aux.txt 2>&1 && echo "aux ok" &&
  timeout -k 10 300 python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1 && echo "resources ok"
  echo "aux part exit $?" ;;
esac
# keep the merged output small: the raw traces are large
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
