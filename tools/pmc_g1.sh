# L2 / fabric traffic and wait counters of the 1x1 GEMMs (upsamplers, head) in a reduced-precision bench run: bash tools/pmc_g1.sh <mode> <tag>
set -e
R=$PWD
M=${1:-fp16s}
O=$R/gpurun_out/${2:-r03}_pmcg1_$M
mkdir -p $O
export PP_TUNE_CACHE=$O/tune.cache
python bench.py --precision $M --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/warm.json 2> $O/warm.err
cd /tmp && export TMPDIR=/tmp
A="--precision $M --no-cpu-baseline --no-extras --steps 3 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/bench.py $A > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pw -- python3 $R/bench.py $A > /dev/null 2> $O/pw.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/ps -- python3 $R/bench.py $A > /dev/null 2> $O/ps.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pi -- python3 $R/bench.py $A > /dev/null 2> $O/pi.err
cd $R
for d in pf pw ps pi; do echo "== $d"; python tools/pmc_summary.py $O/$d ${3:-gemm1x1}; done
