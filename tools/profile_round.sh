#!/bin/bash
# usage: tools/profile_round.sh <tag>     (GPU box, through gpurun)
# rocprofv3 kernel-trace stats of the default bench.py run, then separate PMC passes (HBM bytes, SQ/LDS
# counters) over a short run. Everything lands in gpurun_out/profile_<tag>/ for copying into profiles/.
tag=$1
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $repo/bench.py --no-cpu-baseline > $out/bench_stats_run.log 2>&1 || exit 1
cp $out/stats/bench_kernel_stats.csv $out/kernel_stats.csv
pass() { # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$n -o $n -- python3 $repo/bench.py --no-cpu-baseline --headline-only --settle-ms 0 --steps 4 --warmup 1 > $out/pmc_$n.log 2>&1 || return 1
  python3 $repo/tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && \
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT && \
pass sq2 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES && \
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
# the SATD batch kernels (bench.py's satd block): LDS activity and bank conflicts
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_satd -o satd -- python3 $repo/bench.py --no-cpu-baseline --no-analysis --settle-ms 0 --steps 2 --warmup 1 > $out/pmc_satd.log 2>&1 && python3 $repo/tools/pmc_summary.py $out/pmc_satd > $out/pmc_satd.txt
cd $repo
python3 tools/kstats.py $out/kernel_stats.csv
cat $out/pmc_*.txt
tail -1 $out/bench_stats_run.log
