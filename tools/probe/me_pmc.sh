#!/bin/bash
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/me_pmc
rm -rf $out; mkdir -p $out
cd /tmp
export ME_PROBE_ONLY=$1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/a -o a -- python3 $repo/tools/probe/me_probe.py > $out/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $out/b -o b -- python3 $repo/tools/probe/me_probe.py > $out/b.log 2>&1 || exit 1
cd $repo
python3 tools/pmc_summary.py $out/a | grep -A9 "^me_search"
python3 tools/pmc_summary.py $out/b | grep -A9 "^me_search"
