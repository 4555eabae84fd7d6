# frames-per-pass sweep of the default bench (fixed tiling cache so only the batch changes)
set -e
export PP_TUNE_CACHE=/tmp/tc_sweep.txt
python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > /dev/null
for B in 8 16 24 32 48; do
  S=$((640 / B))
  python bench.py --no-extras --no-cpu-baseline --batch $B --steps $S --warmup 3 > gpurun_out/sweep_b$B.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_b$B.json").read().strip().splitlines()[-1]); print("batch", $B, d["value"], "frames/s", d["ms_per_step"], "ms/step")
PY
done
