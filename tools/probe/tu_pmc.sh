#!/bin/bash
# usage: tools/probe/tu_pmc.sh [tag]   (GPU box) - counters of the transform-unit part of tools/bench_kernels.py
export TMPDIR=/tmp XMI_BENCH_ONLY=tu
tag=${1:-base}
repo=$PWD
out=$repo/gpurun_out/tu_pmc_$tag
mkdir -p $out
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq -o k -- python3 $repo/tools/bench_kernels.py > $out/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA --output-format csv -d $out/sq2 -o k -- python3 $repo/tools/bench_kernels.py > $out/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES --output-format csv -d $out/sq3 -o k -- python3 $repo/tools/bench_kernels.py > $out/sq3.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -o k -- python3 $repo/tools/bench_kernels.py > $out/grbm.log 2>&1 || exit 1
cd $repo
for d in sq sq2 sq3 grbm; do python3 tools/pmc_summary.py $out/$d; done > $out/summary.txt
grep -A9 "^tu_mfma4_kernel<0>\|^tu_mfma_kernel<8, 0>" $out/summary.txt
