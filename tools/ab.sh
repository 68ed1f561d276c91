#!/bin/bash
# usage: tools/ab.sh <variant> [<variant> ...]  -- per-kernel times (rocprofv3 kernel trace) of bench.py per library variant
export TMPDIR=/tmp
repo=$PWD
for v in "$@"; do
  out=$repo/gpurun_out/ab_$v; mkdir -p $out
  if [ "$v" = "base" ]; then unset X265_AMD_LIB; else export X265_AMD_LIB=$repo/build/variants/libx265_amd_$v.so; fi
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $v -- python3 $repo/bench.py --no-cpu-baseline --headline-only --steps ${STEPS:-10} --warmup 2 $BENCH_ARGS > $out/bench.log 2>&1) || { echo "variant $v failed"; tail -5 $out/bench.log; exit 1; }
  echo "== $v"; python3 $repo/tools/kstats.py $out/${v}_kernel_stats.csv; python3 $repo/tools/series.py $out/${v}_kernel_trace.csv
done
