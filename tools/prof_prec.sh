# kernel-trace --stats of a tagged reduced-precision bench run: bash tools/prof_prec.sh <mode> <tag>
set -e
R=$PWD
M=${1:-fp16}
O=$R/gpurun_out/${2:-r03}_$M
mkdir -p $O
export PP_TUNE_CACHE=$O/tune.cache
python bench.py --precision $M --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $O/warm.json 2> $O/warm.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --precision $M --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $O/prof.json 2> $O/prof.err
cd $R
python tools/rocprof_stats.py $O/prof $((24 * 32)) 30 > $O/kernel_summary.txt
cat $O/kernel_summary.txt
python -c "import json;d=json.load(open('$O/warm.json'));print(d['value'],d['value_host_start'],d['roofline']['kernel'],d['roofline']['avg_launch_ms'])"
