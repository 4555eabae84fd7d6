# PP_CONV_DBG ablations of the direct conv kernels under rocprofv3 --kernel-trace (16: no step loops, 4: no epilogue, 1: no staging);
# (the first run tunes and fills PP_TUNE_CACHE; the later ones reuse its picks)
set -e
R=$PWD
mkdir -p $R/gpurun_out; export PP_TUNE_CACHE=$R/gpurun_out/tune.cache
cd /tmp && export TMPDIR=/tmp
for D in 16 4 1; do
PP_CONV_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dbg_$D -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $R/gpurun_out/dbg_$D.json 2> $R/gpurun_out/dbg_$D.err
done
