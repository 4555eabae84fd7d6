set -e
R=$PWD
O=$R/gpurun_out/wdb/pmcb1
mkdir -p $O
export PP_TUNE_CACHE=$R/gpurun_out/wdb/b1d/tune.cache
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/ps -- python3 $R/tools/batch1_probe.py > /dev/null 2> $O/ps.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-trace --output-format csv -d $O/pi -- python3 $R/tools/batch1_probe.py > /dev/null 2> $O/pi.err
cd $R
for k in nms_reduce_b post_topk_b scan_cols_b vox_insert_b; do for d in ps pi; do python tools/pmc_summary.py $O/$d $k; done; done
