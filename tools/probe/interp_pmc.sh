#!/bin/bash
# usage: tools/probe/interp_pmc.sh   (GPU box) - counters of the interpolation and motion-search part of tools/bench_kernels.py
export TMPDIR=/tmp XMI_BENCH_ONLY=me
repo=$PWD
out=$repo/gpurun_out/interp_pmc
mkdir -p $out
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq -o k -- python3 $repo/tools/bench_kernels.py > $out/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -o k -- python3 $repo/tools/bench_kernels.py > $out/grbm.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o k -- python3 $repo/tools/bench_kernels.py > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o k -- python3 $repo/tools/bench_kernels.py > $out/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/sq2 -o k -- python3 $repo/tools/bench_kernels.py > $out/sq2.log 2>&1 || exit 1
cd $repo
for d in sq sq2 grbm fetch write; do python3 tools/pmc_summary.py $out/$d; done > $out/summary.txt
grep -A9 "^interp_kernel" $out/summary.txt
