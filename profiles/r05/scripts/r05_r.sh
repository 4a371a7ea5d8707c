#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_r
mkdir -p $OUT
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=5 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
timeout -k 10 300 python bench.py --config stream2048 > $OUT/bench_stream2048.json 2> $OUT/stream.err
python -c "
import json; d=json.load(open('$OUT/bench_stream2048.json')); print(d['ms_per_step'], d['value'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
timeout -k 10 300 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['indexed_output']['ms_per_step'], d['indexed_output']['kernels_ms'])"
