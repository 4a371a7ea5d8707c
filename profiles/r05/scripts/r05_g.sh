#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_g
mkdir -p $OUT
cd $R
for opt in "--pipeline" "--pipeline --two-streams" "--pipeline --comm" "--pipeline --comm --two-streams" "--pipeline --two-streams --no-stage-events" "--pipeline --no-stage-events"; do
  echo "=== rank_step.py $opt 8" >> $OUT/rank_step_two_streams.txt
  timeout -k 10 200 python tools/rank_step.py $opt 8 2>&1 | grep -v amdgpu.ids >> $OUT/rank_step_two_streams.txt
done
cat $OUT/rank_step_two_streams.txt
# item 4: sustained-load A/B of the vertex-once default: three alternating pairs of 300-step bench runs
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-indexed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('once=1 ms_per_step %.4f emit %.4f' % (d['ms_per_step'], d['kernels']['emit']['avg_ms']))" >> $OUT/ab_sustained_once.txt
  VTMC_BENCH_TUNING="emit_once=0" timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-indexed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('once=0 ms_per_step %.4f emit %.4f' % (d['ms_per_step'], d['kernels']['emit']['avg_ms']))" >> $OUT/ab_sustained_once.txt
done
cat $OUT/ab_sustained_once.txt
