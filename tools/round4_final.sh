# Final measurement round of round 4 (through gpurun from the repository root): the whole GPU suite with its parity report,
# the profiling round of the fp32 headline (kernel trace, SQ / clock / FETCH / WRITE PMC passes, default bench line), the tagged
# reduced-precision lines with the kernel trace and SQ passes of the fp16 mode, and the other configurations.  Everything
# lands in gpurun_out/r04/; copy what is to be judged to profiles/.
set -e
O=gpurun_out/r04
mkdir -p $O
export PP_PARITY_REPORT=$PWD/$O/parity_report.txt
rm -f $PP_PARITY_REPORT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1
unset PP_PARITY_REPORT
grep -E "^\.*\[(rotated iou)" $O/gpu_tests.log | sed 's/^\.*//' >> $O/parity_report.txt || true
tail -1 $O/gpu_tests.log
timeout -k 10 900 bash tools/profile_round4.sh > $O/profile.log 2>&1
tail -2 $O/profile.log | cut -c1-300
for M in bf16x3 bf16 fp16 fp16s; do
  timeout -k 10 300 python bench.py --precision $M --no-cpu-baseline > $O/tagged_$M.json 2> $O/tagged_$M.err
  echo "tagged $M: $(python tools/print_bench.py $O/tagged_$M.json | head -1)"
done
timeout -k 10 200 bash tools/b1_prof.sh r04/b1 > $O/b1_prof.log 2>&1 || true
timeout -k 10 200 python bench.py --config nuscene --no-cpu-baseline --no-extras > $O/cfg_nuscene.json 2> $O/cfg_nuscene.err
timeout -k 10 300 python bench.py --config ntusl_10cm --batch 16 --no-cpu-baseline --no-extras > $O/cfg_ntusl_10cm.json 2> $O/cfg_ntusl_10cm.err
timeout -k 10 200 python bench.py --config nuscene_10class --no-cpu-baseline --no-extras > $O/cfg_nuscene_10class.json 2> $O/cfg_nuscene_10class.err
timeout -k 10 300 python bench.py --batch 64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/batch64.json 2> $O/batch64.err
echo "64 frames per pass: $(python tools/print_bench.py $O/batch64.json | head -1)"
for f in cfg_nuscene cfg_nuscene_10class cfg_ntusl_10cm; do echo "$f: $(python tools/print_bench.py $O/$f.json | head -1)"; done
