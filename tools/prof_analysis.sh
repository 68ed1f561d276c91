#!/bin/bash
# usage: tools/prof_analysis.sh <tag> [bench_analysis args...]   (run on the GPU box through gpurun)
# rocprofv3 kernel-trace stats of one tools/bench_analysis.py run -> gpurun_out/prof_<tag>/kernel_stats.csv
set -e
tag=$1; shift
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $repo/tools/bench_analysis.py "$@" > $out/bench.log 2>&1
cd $repo
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -24 $out/kernel_stats.csv | cut -c1-200
tail -2 $out/bench.log
