#!/bin/bash
# Runs on the GPU box (through gpurun): SQ / TCP counter passes over the A/B harness (one variant,
# few rounds), one rocprofv3 run per pass -- counters only with --kernel-trace, as the pool requires.
# usage: tools/pmc_sq.sh <tag> [ab_bench variant | script.py [args...]]   -> gpurun_out/sq_<tag>/pass*/
TAG=${1:-sq}
VAR=${2:-base}
shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [[ "$VAR" == *.py ]]; then PROG="$R/$VAR $*"; else PROG="$R/tools/ab_bench.py $VAR --rounds 2"; fi
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
P4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
P5="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
P6="SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
    i=$((i + 1))
    mkdir -p $OUT/pass$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $PROG > $OUT/pass$i/out.log 2> $OUT/pass$i/err.log || echo "pass $i failed"
done
python3 $R/tools/pmc_sq_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
