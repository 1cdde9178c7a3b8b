#!/bin/bash
# Copies what tools/gpu_round4.sh left under gpurun_out/r04/ (scratch) into profiles/ (tracked) under the round's names.
set -e
R=${ROUND:-04}; O=gpurun_out/r$R
P=profiles
cp $O/bench_n1.json $P/r${R}_bench_n1.json
cp $O/bench_alone_under_rocprof.json $P/r${R}_alone_under_rocprof.json
cp $O/prof_alone/alone_kernel_stats.csv $P/r${R}_alone_kernel_stats.csv
cp $O/bench_pipe_under_rocprof.json $P/r${R}_pipelined_under_rocprof.json
cp $O/prof_pipe/pipe_kernel_stats.csv $P/r${R}_pipelined_kernel_stats.csv
grep "^{" $O/bench_2rank_rehearsal_ipc.json > $P/r${R}_bench_2rank_rehearsal_ipc.json
grep "^{" $O/bench_4rank_rehearsal_ipc.json > $P/r${R}_bench_4rank_rehearsal_ipc.json
grep -v amdgpu.ids $O/sliced_costs.txt > $P/r${R}_sliced_costs.txt
grep -v amdgpu.ids $O/latency_vs_cpu.txt > $P/r${R}_latency_vs_cpu.txt
grep -v amdgpu.ids $O/bench_aux.txt > $P/r${R}_bench_aux.txt
grep "differential soak" $O/differential_soak.txt > $P/r${R}_differential_soak.txt
grep "sliced soak" $O/sliced_soak.txt > $P/r${R}_sliced_soak.txt
grep -E "k_sweep|k_merge_level|k_writeback|k_events|k_insert" $O/pmc_bench_summary.txt > $P/r${R}_pmc_hbm_traffic_raw.txt
{ echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline   (IMT_NO_PIPELINE=1)"
  echo "# (tools/pmc_summary.py: mean per dispatch, summed over the 8 XCDs / 32 SEs of the device; kernels are serialised under --pmc)"
  grep -E "k_sweep|k_mad_peak|k_insert_chains" $O/pmc_valu_alone_summary.txt
  echo "# instruction classes of the same launches: --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU  and  --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
  echo "# (a k_sweep launch = 2048 waves of 64 hashes: divide by 2048 for wave-instructions per hash)"
  grep -E "k_sweep" $O/pmc_classes_summary.txt; } > $P/r${R}_pmc_valu_sweep_raw.txt
python tools/kernel_resources.py > $P/r${R}_kernel_resources.txt
ls $P | grep r${R}_
