#!/bin/bash
# Copies what tools/gpu_round2.sh left under gpurun_out/r02/ (scratch) into profiles/ (tracked) under the round's names.
# The hand-annotated summaries (r02_pmc_*_traffic.txt, r02_small_batch_rates.txt) quote these raw files.
set -e
O=gpurun_out/r02
P=profiles
cp $O/bench_n1.json $P/r02_bench_n1.json
cp $O/bench_alone_under_rocprof.json $P/r02_alone_under_rocprof.json
cp $O/prof_alone/alone_kernel_stats.csv $P/r02_alone_kernel_stats.csv
cp $O/bench_pipe_under_rocprof.json $P/r02_pipelined_under_rocprof.json
cp $O/prof_pipe/pipe_kernel_stats.csv $P/r02_pipelined_kernel_stats.csv
cp $O/prof_trace/t_kernel_stats.csv $P/r02_trace_kernel_stats.csv
cp $O/trace_rate.txt $P/r02_trace_rate.txt
cp $O/bench_2rank_rehearsal.json $P/r02_bench_2rank_rehearsal_one_gpu.json
cp $O/bench_4rank_rehearsal.json $P/r02_bench_4rank_rehearsal_one_gpu.json
cp $O/verify_latency.txt $P/r02_verify_latency.txt
grep check $O/witness_check_latency.txt > $P/r02_witness_check_latency.txt
grep insert_trace $O/insert_trace_latency.txt > $P/r02_insert_trace_latency.txt
cp $O/bench_aux.txt $P/r02_bench_aux.txt
cp $O/differential_soak.txt $P/r02_differential_soak.txt
cp $O/scale_check.txt $P/r02_scale_check.txt
cp $O/pmc_bench_summary.txt $P/r02_pmc_bench_summary_raw.txt
cp $O/pmc_trace_summary.txt $P/r02_pmc_trace_summary_raw.txt
ls $P | grep r02_
