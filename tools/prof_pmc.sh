#!/bin/bash
# usage: tools/prof_pmc.sh <tag> "<counter list>" [bench args...]   (run on the GPU box through gpurun)
# One rocprofv3 --pmc pass (counters only, no tracing) over a short bench.py run; prints per-kernel means.
set -e
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
repo=$PWD
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -o $tag -- python3 $repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/bench.log 2>&1
cd $repo
python3 tools/pmc_summary.py $out
