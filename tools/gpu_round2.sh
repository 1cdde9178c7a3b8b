#!/bin/bash
# One GPU session of round 2: tests, bench, rocprofv3 kernel stats (alone + pipelined), PMC traffic passes for the
# insertion kernel and for the trace kernel, 2-rank rehearsal.  Steps are chained with && and carry their own
# timeouts; everything lands under gpurun_out/r02/.
set -o pipefail
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err && echo "bench ok" &&
( export IMT_NO_PIPELINE=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_alone -o alone -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_alone_under_rocprof.json 2> $O/prof_alone.err ) && echo "prof alone ok" &&
( export IMT_BENCH_NO_ATTRIBUTION=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipe -o pipe -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_pipe_under_rocprof.json 2> $O/prof_pipe.err ) && echo "prof pipe ok" &&
( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err ) && echo "pmc fetch ok" &&
( export IMT_NO_PIPELINE=1 IMT_BENCH_NO_ATTRIBUTION=1; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err ) && echo "pmc write ok" &&
timeout -k 10 300 python tools/trace_rate.py 16 17 18 19 > $O/trace_rate.txt 2>&1 && echo "trace rate ok" &&
( timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -o t -- python3 tools/trace_rate.py 18 > /dev/null 2> $O/prof_trace.err ) && echo "prof trace ok" &&
( timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_trace_fetch -o f -- python3 tools/trace_rate.py 18 > /dev/null 2> $O/pmc_trace_fetch.err ) &&
( timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_trace_write -o w -- python3 tools/trace_rate.py 18 > /dev/null 2> $O/pmc_trace_write.err ) && echo "pmc trace ok" &&
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_bench_summary.txt && python tools/pmc_summary.py $O/pmc_trace_fetch $O/pmc_trace_write > $O/pmc_trace_summary.txt &&
( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo; timeout -k 10 400 python3 bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank.err ) && echo "2-rank ok" &&
( export IMT_BENCH_DEVICE=0 IMT_BENCH_COLLECTIVE=gloo; timeout -k 10 400 python3 bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_4rank_rehearsal.json 2> $O/bench_4rank.err ) && echo "4-rank ok" &&
timeout -k 10 300 python tools/verify_latency.py > $O/verify_latency.txt 2>&1 && echo "verify latency ok" &&
timeout -k 10 300 python tools/witness_check_latency.py > $O/witness_check_latency.txt 2>&1 && timeout -k 10 300 python tools/insert_trace_latency.py > $O/insert_trace_latency.txt 2>&1 && echo "check / trace latency ok" &&
timeout -k 10 300 python tools/small_batch_rate.py 8 16 > $O/small_batch.txt 2>&1 && echo "small batch ok" &&
timeout -k 10 300 python tools/bench_aux.py > $O/bench_aux.txt 2>&1 && echo "aux ok" &&
SOAK_SECONDS=60 timeout -k 10 400 python tools/differential_soak.py > $O/differential_soak.txt 2>&1 && echo "soak ok" &&
timeout -k 10 300 python tools/scale_check.py > $O/scale_check.txt 2>&1 && echo "scale check ok"
echo "exit $?"
# keep the merged output small: the raw traces are large
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
