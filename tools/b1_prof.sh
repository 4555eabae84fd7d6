# kernel-trace --stats of the batch-1 (BASELINE config 2) loop: bash tools/b1_prof.sh <tag>
set -e
R=$PWD
O=$R/gpurun_out/${1:-b1prof}
mkdir -p $O
export PP_TUNE_CACHE=$O/tune.cache
python tools/batch1_probe.py > $O/warm.txt 2> $O/warm.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/batch1_probe.py > $O/prof.txt 2> $O/prof.err
cd $R
python tools/rocprof_stats.py $O/prof 121 40 > $O/kernel_summary.txt
cat $O/warm.txt
cat $O/kernel_summary.txt
