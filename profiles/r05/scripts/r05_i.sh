#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_i
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_bench_modes.py tests/test_gpu_parity.py tests/test_lifecycle.py -m gpu -x -q -p no:cacheprovider -k "bench or rccl or lifecycle or stream" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-indexed > $OUT/bench_streams2_$i.json 2> $OUT/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-indexed --streams 1 > $OUT/bench_streams1_$i.json 2>> $OUT/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_i/bench_streams*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], d['ms_per_step'], d['value'], 'roofline', d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline']['frac'], 'iso', d['roofline']['isolated']['avg_ms'], d['roofline']['isolated']['frac'], {k:(v['avg_ms'],v['isolated_ms']) for k,v in d['kernels'].items()}, 'path', d['path_roofline']['frac_of_peak'])
PY
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/rocprof_streams2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-indexed > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
ls $OUT/rocprof_streams2/*/ | head
find $OUT/rocprof_streams2 -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {}
