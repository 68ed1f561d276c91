#!/bin/bash
# usage: tools/profile_secondary.sh <tag>   (GPU box) - kernel-trace stats + MFMA/VALU counters of tools/bench_kernels.py
tag=$1
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/secondary_$tag
mkdir -p $out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 $repo/tools/bench_kernels.py > $out/bench_kernels.jsonl 2>&1 || exit 1
cp $out/stats/k_kernel_stats.csv $out/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $out/pmc -o k -- python3 $repo/tools/bench_kernels.py > $out/pmc.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -o k -- python3 $repo/tools/bench_kernels.py > $out/pmc2.log 2>&1 || exit 1
export XMI_BENCH_ONLY=tu
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -o k -- python3 $repo/tools/bench_kernels.py > $out/pmc3.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc4 -o k -- python3 $repo/tools/bench_kernels.py > $out/pmc4.log 2>&1 || exit 1
unset XMI_BENCH_ONLY
cd $repo
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt
python3 tools/pmc_summary.py $out/pmc3 >> $out/pmc_summary.txt
python3 tools/pmc_summary.py $out/pmc4 >> $out/pmc_summary.txt
python3 tools/pmc_summary.py $out/pmc2 >> $out/pmc_summary.txt
python3 tools/kstats.py $out/kernel_stats.csv
grep -A9 "tu_mfma" $out/pmc_summary.txt | head -60
