set -e
R=$PWD
export PP_TUNE_CACHE=$R/gpurun_out/tune.cache
python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/warm.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcf3 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmcf3.json 2> $R/gpurun_out/pmcf3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcw3 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmcw3.json 2> $R/gpurun_out/pmcw3.err
cd $R
python tools/pmc_summary.py gpurun_out/pmcf3 mfma
python tools/pmc_summary.py gpurun_out/pmcf3 gemm1x1
python tools/pmc_summary.py gpurun_out/pmcw3 mfma
python tools/pmc_summary.py gpurun_out/pmcw3 gemm1x1
