# Copy the figures of gpurun_out/r04 (tools/round4_final.sh) that are to be judged into profiles/ (tracked).
set -e
O=gpurun_out/r04
P=profiles
cp $O/bench_default.json $P/r04_bench_default.json
cp $O/kernel_stats.csv $P/r04_bench_default_kernel_stats.csv
cp $O/kernel_summary.txt $P/r04_bench_default_summary.txt
cp $O/prof.json $P/r04_bench_default_profiled.json
cp $O/hbm_traffic.json $P/r04_hbm_traffic.json
{ echo "== SQ waits (rocprofv3 --pmc, bench.py default, wino kernels)"; cat $O/pmc_sq_waits.txt; echo "== instruction mix"; cat $O/pmc_inst_mix.txt; echo "== GRBM_GUI_ACTIVE (clock = value / 8 XCDs / duration)"; cat $O/pmc_clock.txt; } > $P/r04_pmc_wino.txt
cp $O/parity_report.txt $P/r04_parity_report.txt
for M in fp16 fp16s bf16 bf16x3; do cp $O/tagged_$M.json $P/r04_bench_tagged_$M.json; done
cp $O/b1_prof.log $P/r04_batch1_kernel_summary.txt
cp $O/cfg_nuscene.json $P/r04_bench_config_nuscene.json
cp $O/cfg_ntusl_10cm.json $P/r04_bench_config_ntusl_10cm.json
cp $O/cfg_nuscene_10class.json $P/r04_bench_config_nuscene_10class.json

cp $O/batch64.json $P/r04_bench_batch64.json
ls -la $P | grep r04
