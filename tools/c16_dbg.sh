# timing-only ablations of the conv16 kernels through the tuner's verbose table (16 frames per launch): bash tools/c16_dbg.sh
for D in 0; do
  echo "== PP_CONV_DBG=$D"
  PP_CONV_DBG=$D PP_VERBOSE=1 timeout -k 10 200 python tools/prec_probe.py eight_20cm 16 fp16 2>&1 | grep "autotune" | grep -v retime | grep "c16" | awk '{print $5,$6,$7,$8,$9, $(NF-6), $(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
done
