#!/bin/bash
# Copies what tools/gpu_round3.sh left under gpurun_out/r03/ (scratch) into profiles/ (tracked) under the round's names.
set -e
O=gpurun_out/r03
P=profiles
cp $O/bench_n1.json $P/r03_bench_n1.json
cp $O/bench_alone_under_rocprof.json $P/r03_alone_under_rocprof.json
cp $O/prof_alone/alone_kernel_stats.csv $P/r03_alone_kernel_stats.csv
cp $O/bench_pipe_under_rocprof.json $P/r03_pipelined_under_rocprof.json
cp $O/prof_pipe/pipe_kernel_stats.csv $P/r03_pipelined_kernel_stats.csv
cp $O/bench_2rank_rehearsal.json $P/r03_bench_2rank_rehearsal_one_gpu.json
cp $O/bench_4rank_rehearsal.json $P/r03_bench_4rank_rehearsal_one_gpu.json
grep -v amdgpu.ids $O/sliced_costs.txt > $P/r03_sliced_costs.txt
grep -v amdgpu.ids $O/latency_vs_cpu.txt > $P/r03_latency_vs_cpu.txt
cp $O/verify_latency.txt $P/r03_verify_latency.txt
grep check $O/witness_check_latency.txt > $P/r03_witness_check_latency.txt
grep insert_trace $O/insert_trace_latency.txt > $P/r03_insert_trace_latency.txt
cp $O/bench_aux.txt $P/r03_bench_aux.txt
grep "differential soak" $O/differential_soak.txt > $P/r03_differential_soak.txt
[ -f $O/sliced_soak.txt ] && grep "sliced soak" $O/sliced_soak.txt > $P/r03_sliced_soak.txt
grep -E "k_sweep|k_merge_level|k_writeback|k_events|k_insert" $O/pmc_bench_summary.txt > $P/r03_pmc_hbm_traffic_raw.txt
{ echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  echo "# (tools/pmc_summary.py: mean per dispatch, summed over the 8 XCDs / 32 SEs of the device; kernels are serialised under --pmc)"
  echo "## default run (IMT_PIPELINE requested; counters serialise the launches)"; grep -E "k_sweep|k_mad_peak|k_insert_chains" $O/pmc_valu_summary.txt
  echo "## IMT_NO_PIPELINE=1"; grep -E "k_sweep|k_mad_peak|k_insert_chains" $O/pmc_valu_alone_summary.txt; } > $P/r03_pmc_valu_sweep_raw.txt
python tools/kernel_resources.py > $P/r03_kernel_resources.txt
ls $P | grep r03_
