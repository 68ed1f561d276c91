#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# Writes the rocprofv3 kernel-trace stats of one bench.py run under gpurun_out/prof_<tag>/.
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $OLDPWD/bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
cd $OLDPWD
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -12 $out/kernel_stats.csv
tail -2 $out/bench.log
