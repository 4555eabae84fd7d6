set -e
R=$PWD
export PP_TUNE_CACHE=$R/gpurun_out/tune.cache
python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/warm.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof2.json 2> $R/gpurun_out/prof2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcf2 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmcf2.json 2> $R/gpurun_out/pmcf2.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcw2 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmcw2.json 2> $R/gpurun_out/pmcw2.err
cd $R
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/prof2.json gpurun_out/bench_default.json
