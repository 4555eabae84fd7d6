# Final measurement round of round 2 (through gpurun from the repository root): the whole GPU suite with its parity
# report, the profiling round, the other configurations and the tagged reduced-precision lines.  Everything lands in
# gpurun_out/r02/; copy what is to be judged to profiles/.
set -e
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1
grep -E "^\.*\[(frame parity|golden samples|rotated iou|precision)" $O/gpu_tests.log | sed 's/^\.*//' > $O/parity_report.txt || true
tail -1 $O/gpu_tests.log
timeout -k 10 700 bash tools/profile_round2.sh > $O/profile.log 2>&1
tail -2 $O/profile.log | cut -c1-300
timeout -k 10 200 python bench.py --config nuscene --no-cpu-baseline --no-extras > $O/cfg_nuscene.json 2> $O/cfg_nuscene.err
timeout -k 10 300 python bench.py --config ntusl_10cm --batch 16 --no-cpu-baseline --no-extras > $O/cfg_ntusl_10cm.json 2> $O/cfg_ntusl_10cm.err
timeout -k 10 200 python bench.py --precision bf16x3 --no-cpu-baseline --no-extras > $O/tagged_bf16x3.json 2> $O/tagged_bf16x3.err
timeout -k 10 200 python bench.py --precision bf16 --no-cpu-baseline --no-extras > $O/tagged_bf16.json 2> $O/tagged_bf16.err
for f in cfg_nuscene cfg_ntusl_10cm tagged_bf16x3 tagged_bf16; do echo "$f: $(python tools/print_bench.py $O/$f.json)"; done
