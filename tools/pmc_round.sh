set -e
R=$PWD
export PP_TUNE_CACHE=$R/gpurun_out/tune.cache
python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcs -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmcs.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmci -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmci.err
cd $R
python tools/pmc_summary.py gpurun_out/pmcs wino_mfma
python tools/pmc_summary.py gpurun_out/pmci wino_mfma
python tools/pmc_summary.py gpurun_out/pmcs gemm1x1
python tools/pmc_summary.py gpurun_out/pmci gemm1x1
