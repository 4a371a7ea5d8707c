#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_o
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_bench_modes.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_bench_modes.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_bench_modes.log
tail -5 $OUT/pytest_bench_modes.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-indexed > $OUT/bench_default.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_o/bench_default.json'))
print(d['ms_per_step'], d['value'], d['one_stream_ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels'], d['path_roofline'])
PY
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/rocprof_default --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-indexed > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
tail -3 $OUT/rocprof.err
