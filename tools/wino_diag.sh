# timing-only ablations of the Winograd loop (developer tool; results of these builds are wrong by design)
# build the ablation libraries first, in 3d_object_detection_amd/csrc (k = bit set of PP_WINO_DIAG, see conv.hip):
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -DPP_WINO_DIAG=$k -c conv.hip -o _build/conv_diag$k.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o _build/libpp_diag$k.so _build/{pp_api,voxelize,anchor_mask,pfn_scatter,postprocess,frame,eval_host}.o _build/conv_diag$k.o
# tools/wino_stamp.py needs the same with -DPP_WINO_STAMP=1 into _build/libpp_stamp.so
R=$PWD
for k in ${DIAGS:-0 15}; do
  for dbg in ${DBGS:-0 1 4 5}; do
    if [ $k = 0 ]; then L=$R/3d_object_detection_amd/csrc/libpp_hip.so; else L=$R/3d_object_detection_amd/csrc/_build/libpp_diag$k.so; fi
    echo "== diag $k dbg $dbg"
    PP_HIP_LIB=$L PP_CONV_DBG=$dbg PP_VERBOSE=1 timeout -k 10 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "${PAT:-wino tw8 w1x4 bx1 kc8 }" | grep -v -e retime -e "->"
  done
done
