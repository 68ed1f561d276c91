#!/bin/bash
# GPU box: analysis parity tests, then the analysis rate for several polling times, lane counts and concurrent segments.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 560 python -u -m pytest tests/test_analysis_gpu.py -x -q --timeout 240 --timeout-method thread > gpurun_out/an_tests.log 2>&1
rc=$?
tail -5 gpurun_out/an_tests.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/an_bench.log
for sp in 0 50 200; do
  timeout -k 10 120 python -u tools/bench_analysis.py --qp 32 --pictures 5 --spin-us $sp >> gpurun_out/an_bench.log 2>&1 || exit 1
done
timeout -k 10 120 python -u tools/bench_analysis.py --qp 32 --pictures 5 --spin-us 200 --lanes 2 >> gpurun_out/an_bench.log 2>&1 || exit 1
for s in 2 8; do
  for sp in 0 100; do
    timeout -k 10 180 python -u tools/bench_analysis.py --qp 32 --pictures 4 --segments $s --spin-us $sp >> gpurun_out/an_bench.log 2>&1 || exit 1
  done
done
timeout -k 10 120 python -u tools/bench_analysis.py --qp 22 --pictures 4 --spin-us 200 >> gpurun_out/an_bench.log 2>&1 || exit 1
cut -c1-420 gpurun_out/an_bench.log
