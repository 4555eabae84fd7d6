# One profiling round on the GPU box (run through gpurun): kernel-trace stats of the default bench, HBM traffic
# PMC passes of the roofline layer for the tilings the tuner chooses between, and the default bench line.
set -e
R=$PWD
export PP_TUNE_CACHE=$R/gpurun_out/tune.cache
rm -f $PP_TUNE_CACHE
python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/warm.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof.json 2> $R/gpurun_out/prof.err
unset PP_TUNE_CACHE
i=0
for T in "wino tw8 w1x4 bx1 kc4" "wino tw8 w1x4 bx1 kc8" "wino tw8 w1x4 bx2 kc8" "wino tw4 w1x4 bx2 kc8"; do
  export PP_FORCE_VARIANT="$T"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$i -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $R/gpurun_out/pmcf_$i.json 2> $R/gpurun_out/pmcf_$i.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$i -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $R/gpurun_out/pmcw_$i.json 2> $R/gpurun_out/pmcw_$i.err
  i=$((i+1))
done
unset PP_FORCE_VARIANT
cd $R
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/prof.json gpurun_out/bench_default.json
