export PP_TUNE_CACHE=$PWD/gpurun_out/tune.cache
rm -f $PP_TUNE_CACHE
for cfg in "3 1 16" "0 2 8" "0 3 8" "1 2 8" "0 2 16"; do
  set -- $cfg
  echo "aux=$1 streams=$2 batch=$3: $(PP_AUX_STREAMS=$1 python bench.py --no-cpu-baseline --streams $2 --batch $3 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["achieved"])')"
done
