set -e
R=$PWD
export PP_TUNE_CACHE=$R/gpurun_out/tune.cache
rm -f $PP_TUNE_CACHE
python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/warm.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcg -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmcg.json 2> $R/gpurun_out/pmcg.err
cd $R
python tools/pmc_summary.py gpurun_out/pmcg mfma
python tools/pmc_summary.py gpurun_out/pmcg gemm1x1
