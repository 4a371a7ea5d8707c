#!/bin/bash
# Instruction-mix passes only (2 rocprofv3 runs per variant) over several ab_bench variants: where do the instructions of a block go?
# usage: tools/pmc_insts2.sh <tag> <variant> [<variant> ...]   -> gpurun_out/insts_<tag>/<n>/summary.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM"
n=0
for VAR in "$@"; do
    n=$((n + 1))
    OUT=$R/gpurun_out/insts_$TAG/$n
    mkdir -p $OUT
    echo "$VAR" > $OUT/variant.txt
    i=0
    for P in "$P1" "$P2" "$P3"; do
        i=$((i + 1))
        mkdir -p $OUT/pass$i
        timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $R/tools/ab_bench.py "$VAR" --rounds 2 > $OUT/pass$i/out.log 2> $OUT/pass$i/err.log || echo "pass $i failed"
    done
    python3 $R/tools/pmc_sq_summary.py $OUT | grep -A 30 "emit_kernel" > $OUT/summary.txt
    echo "== $VAR"; cat $OUT/summary.txt
    rm -rf $OUT/pass*/*/  # raw csv not needed back home
done
