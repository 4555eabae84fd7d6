# SQ wait / instruction-mix / clock PMC passes of a reduced-precision bench run (separate passes, kernel-trace only): bash tools/pmc_prec.sh <mode> <tag>
set -e
R=$PWD
M=${1:-fp16}
O=$R/gpurun_out/${2:-r03}_pmc_$M
mkdir -p $O
export PP_TUNE_CACHE=$O/tune.cache
python bench.py --precision $M --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/warm.json 2> $O/warm.err
cd /tmp && export TMPDIR=/tmp
A="--precision $M --no-cpu-baseline --no-extras --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmcs -- python3 $R/bench.py $A > /dev/null 2> $O/pmcs.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmci -- python3 $R/bench.py $A > /dev/null 2> $O/pmci.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcg -- python3 $R/bench.py $A > /dev/null 2> $O/pmcg.err
cd $R
for d in pmcs pmci pmcg; do echo "== $d"; python tools/pmc_summary.py $O/$d ${3:-conv16}; done
