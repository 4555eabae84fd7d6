# Copy the figures of gpurun_out/r03 (tools/round3_final.sh) that are to be judged into profiles/ (tracked).
set -e
O=gpurun_out/r03
P=profiles
cp $O/bench_default.json $P/r03_bench_default.json
cp $O/kernel_stats.csv $P/r03_bench_default_kernel_stats.csv
cp $O/kernel_summary.txt $P/r03_bench_default_summary.txt
cp $O/prof.json $P/r03_bench_default_profiled.json
cp $O/hbm_traffic.json $P/r03_hbm_traffic.json
{ echo "== SQ waits (rocprofv3 --pmc, bench.py default, wino kernels)"; cat $O/pmc_sq_waits.txt; echo "== instruction mix"; cat $O/pmc_inst_mix.txt; echo "== GRBM_GUI_ACTIVE (clock = value / 8 XCDs / duration)"; cat $O/pmc_clock.txt; } > $P/r03_pmc_wino4.txt
cp $O/parity_report.txt $P/r03_parity_report.txt
for M in fp16 fp16s bf16 bf16x3; do cp $O/tagged_$M.json $P/r03_bench_tagged_$M.json; done
cp $O/x_fp16/kernel_summary.txt $P/r03_fp16_kernel_summary.txt
cp $(ls -t $O/x_fp16/prof/*/*kernel_stats.csv | head -1) $P/r03_fp16_kernel_stats.csv
cp $O/pmc_fp16.log $P/r03_pmc_conv16_fp16.txt
cp $O/y_fp16s/kernel_summary.txt $P/r03_fp16s_kernel_summary.txt
cp $O/pmc_g1_fp16s.log $P/r03_pmc_gemm1x1_fp16s.txt
cp $O/b1_prof.log $P/r03_batch1_kernel_summary.txt
cp $O/cfg_nuscene.json $P/r03_bench_config_nuscene.json
cp $O/cfg_ntusl_10cm.json $P/r03_bench_config_ntusl_10cm.json
cp $O/cfg_nuscene_10class.json $P/r03_bench_config_nuscene_10class.json
ls -la $P | grep r03
