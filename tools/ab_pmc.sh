#!/bin/bash
# usage: tools/ab_pmc.sh "<counters>" <variant>...   -- one PMC pass per library variant (short bench run)
export TMPDIR=/tmp
repo=$PWD
ctrs=$1; shift
for v in "$@"; do
  out=$repo/gpurun_out/abpmc_$v; rm -rf $out; mkdir -p $out
  if [ "$v" = "base" ]; then unset X265_AMD_LIB; else export X265_AMD_LIB=$repo/build/variants/libx265_amd_$v.so; fi
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o $v -- python3 $repo/bench.py --no-cpu-baseline --steps 30 --warmup 5 $BENCH_ARGS > $out/bench.log 2>&1) || { echo "variant $v failed"; tail -5 $out/bench.log; exit 1; }
  echo "== $v"; python3 $repo/tools/pmc_summary.py $out; python3 - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_trace.csv",recursive=True)
if f:
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f[0])) if 'sad_surface' in r['Kernel_Name']]
    print('   duration us: avg %.1f  last10 avg %.1f'%(sum(d)/len(d), sum(d[-10:])/10))
PY
done
