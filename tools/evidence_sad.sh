#!/bin/bash
# usage: tools/evidence_sad.sh <tag>   (GPU box, through gpurun) -> gpurun_out/evidence_<tag>/
# Steady-state evidence for the SAD surface kernel: per-launch durations of a long burst (product library and the
# store-free / compute-free variants), per-workgroup clocks after bursts of different length, and the microbenchmarks
# that give the store and issue bounds quoted for the kernel. Variant libraries are built beforehand with tools/build_variant.sh
# (timeline: -DXMI_EXP_TIMELINE, nostore: -DXMI_EXP_NOSTORE, nocompute: -DXMI_EXP_NOCOMPUTE).
tag=$1
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/evidence_$tag
mkdir -p $out
for v in base nostore nocompute; do
  if [ "$v" = "base" ]; then unset X265_AMD_LIB; else export X265_AMD_LIB=$repo/build/variants/libx265_amd_$v.so; fi
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$v -o $v -- python3 $repo/bench.py --no-cpu-baseline --headline-only --settle-ms 0 --steps 1500 --warmup 2 > $out/trace_$v.log 2>&1) || exit 1
  { echo "== $v: bench.py --headline-only --settle-ms 0 --steps 1500 --warmup 2 under rocprofv3 --kernel-trace"; python3 tools/kstats.py $out/trace_$v/${v}_kernel_stats.csv | grep sad_surface; python3 tools/series.py $out/trace_$v/${v}_kernel_trace.csv; } >> $out/sad_launch_series.txt
  rm -rf $out/trace_$v
done
export X265_AMD_LIB=$repo/build/variants/libx265_amd_timeline.so
for n in 8 60 600 2000; do
  echo "== after $n back-to-back launches" >> $out/sad_timeline.txt
  LAUNCHES=$n timeout -k 10 100 python3 tools/timeline.py >> $out/sad_timeline.txt 2>&1 || exit 1
done
unset X265_AMD_LIB
{ echo "== store_pitch"; timeout -k 10 60 tools/microbench/store_pitch; echo "== store_pitch4"; timeout -k 10 60 tools/microbench/store_pitch4; echo "== store_shapes"; timeout -k 10 60 tools/microbench/store_shapes; echo "== qsad_cycles"; timeout -k 10 60 tools/microbench/qsad_cycles; } > $out/microbench.txt 2>&1
cat $out/sad_launch_series.txt $out/sad_timeline.txt $out/microbench.txt
