# Fixed-tiling A/B of two builds of libpp_hip.so on the GPU box (run through gpurun): the working tree's library against a
# baseline built by hand from another revision's conv.hip into 3d_object_detection_amd/csrc/_ab/libpp_base.so (git-ignored,
# travels with the snapshot; PP_HIP_LIB selects it).  The first run tunes and fills PP_TUNE_CACHE, the others reuse its
# picks, so only the code differs; alternating runs give +-0.1 % repeatability.
set -e
B=3d_object_detection_amd/csrc/_ab/libpp_base.so
export PP_TUNE_CACHE=/tmp/tc.txt
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/ab_new0.json
for i in 1 2; do
PP_HIP_LIB=$B python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 3 > gpurun_out/ab_base$i.json
python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 3 > gpurun_out/ab_new$i.json
done
python - <<'PY'
import json
for n in ("new0","base1","new1","base2","new2"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"])
PY
