# Timing-only ablations of wino4_mfma (PP_W4_DIAG builds: make -C 3d_object_detection_amd/csrc variant NAME=w4d<N> DEFS=-DPP_W4_DIAG=<N>).
# Usage (through gpurun, repository root): bash tools/w4_diag_run.sh 4 2 16 64 80 86 128 256 384
for d in "$@"; do
  PP_HIP_LIB=$PWD/3d_object_detection_amd/csrc/_build/libpp_w4d$d.so timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/diag_$d.json 2> gpurun_out/diag_$d.err || exit 1
  echo "diag $d: $(python tools/print_bench.py gpurun_out/diag_$d.json)"
done
