#!/bin/bash
# One GPU session of round 3: tests, bench, rocprofv3 kernel stats (alone + pipelined), PMC passes (HBM traffic and VALU
# issue of k_sweep), rehearsals of the multi-GPU modes on the one GPU, latency tables.  Steps are chained with && and
# carry their own timeouts; everything lands under gpurun_out/r03/ (tools/collect_profiles.sh copies the summaries).
set -o pipefail
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err && echo "bench ok" &&
( export IMT_NO_PIPELINE=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_alone -o alone -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_alone_under_rocprof.json 2> $O/prof_alone.err ) && echo "prof alone ok" &&
( export IMT_BENCH_NO_ATTRIBUTION=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipe -o pipe -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_pipe_under_rocprof.json 2> $O/prof_pipe.err ) && echo "prof pipe ok" &&
( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err ) && echo "pmc fetch ok" &&
( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err ) && echo "pmc write ok" &&
( export IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_valu -o v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_pmc_valu.json 2> $O/pmc_valu.err ) && echo "pmc valu (pipelined) ok" &&
( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1 IMT_BENCH_NO_TRACE=1; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_valu_alone -o v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_valu_alone.err ) && echo "pmc valu (alone) ok" &&
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_bench_summary.txt && python tools/pmc_summary.py $O/pmc_valu > $O/pmc_valu_summary.txt && python tools/pmc_summary.py $O/pmc_valu_alone > $O/pmc_valu_alone_summary.txt &&
( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo; timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank.err ) && echo "2-rank ok" &&
( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo; timeout -k 10 600 python3 bench.py --gpus 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_4rank_rehearsal.json 2> $O/bench_4rank.err ) && echo "4-rank ok" &&
timeout -k 10 400 python tools/sliced_costs.py 1 2 4 8 > $O/sliced_costs.txt 2>&1 && echo "sliced costs ok" &&
timeout -k 10 300 python tools/latency_vs_cpu.py > $O/latency_vs_cpu.txt 2>&1 && echo "latency vs cpu ok" &&
timeout -k 10 300 python tools/verify_latency.py > $O/verify_latency.txt 2>&1 && echo "verify latency ok" &&
timeout -k 10 300 python tools/witness_check_latency.py > $O/witness_check_latency.txt 2>&1 && timeout -k 10 300 python tools/insert_trace_latency.py > $O/insert_trace_latency.txt 2>&1 && echo "check / trace latency ok" &&
timeout -k 10 300 python tools/bench_aux.py > $O/bench_aux.txt 2>&1 && echo "aux ok" &&
SOAK_SECONDS=60 timeout -k 10 400 python tools/differential_soak.py > $O/differential_soak.txt 2>&1 && echo "soak ok" &&
SOAK_SECONDS=120 timeout -k 10 400 python tools/sliced_soak.py > $O/sliced_soak.txt 2>&1 && echo "sliced soak ok"
echo "exit $?"
# keep the merged output small: the raw traces are large
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
