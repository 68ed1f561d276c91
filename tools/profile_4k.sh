#!/bin/bash
# usage: tools/profile_4k.sh <tag>     (GPU box) - the 3840x2160, search range 128 workload: kernel-trace stats of a short
# bench.py run and the two HBM-byte PMC passes. Output in gpurun_out/profile4k_<tag>/.
tag=$1
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/profile4k_$tag
mkdir -p $out
args="--width 3840 --height 2160 --range 128 --frames 3 --no-cpu-baseline"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $repo/bench.py $args --steps 10 --warmup 2 > $out/bench_stats_run.log 2>&1 || exit 1
cp $out/stats/bench_kernel_stats.csv $out/kernel_stats.csv
for n in fetch write; do
  c=FETCH_SIZE; [ $n = write ] && c=WRITE_SIZE
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$n -o $n -- python3 $repo/bench.py $args --steps 3 --warmup 1 > $out/pmc_$n.log 2>&1 || exit 1
  python3 $repo/tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt
done
cd $repo
python3 tools/kstats.py $out/kernel_stats.csv
cat $out/pmc_*.txt
tail -1 $out/bench_stats_run.log
