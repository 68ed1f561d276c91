#!/bin/bash
# GPU box: analysis parity tests, then the analysis rate for several lane counts and concurrent segments.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 560 python -u -m pytest tests/test_analysis_gpu.py -x -q --timeout 240 --timeout-method thread > gpurun_out/an_tests.log 2>&1
rc=$?
tail -5 gpurun_out/an_tests.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/an_bench.log
for l in 1 2 3 4 6; do
  timeout -k 10 120 python -u tools/bench_analysis.py --qp 32 --pictures 5 --lanes $l >> gpurun_out/an_bench.log 2>&1 || exit 1
done
for s in 2 4 8; do
  timeout -k 10 180 python -u tools/bench_analysis.py --qp 32 --pictures 4 --segments $s --lanes 2 >> gpurun_out/an_bench.log 2>&1 || exit 1
done
timeout -k 10 120 python -u tools/bench_analysis.py --qp 22 --pictures 4 --lanes 3 >> gpurun_out/an_bench.log 2>&1 || exit 1
cat gpurun_out/an_bench.log
