#!/bin/bash
# kernel-trace statistics of tools/bench_pipeline.py (GPU box): which kernels make up a picture's 0.2 ms
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/pipeline_stats
rm -rf $out; mkdir -p $out
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s -o p -- python3 $repo/tools/bench_pipeline.py > $out/run.log 2>&1 || exit 1
cd $repo
python3 tools/kstats.py $out/s/p_kernel_stats.csv
