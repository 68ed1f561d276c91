#!/bin/bash
# usage: tools/ab_bench.sh <out-name> <variant> [<variant> ...]   (GPU box) - bench.py's headline leg per library variant
# ("base" = the product library, others = build/variants/libx265_amd_<variant>.so), steady state, plain HIP-event timing.
# BENCH_ARGS adds bench.py options (e.g. "--width 3840 --height 2160 --range 128 --frames 3 --steps 60").
out=gpurun_out/$1; shift
mkdir -p $(dirname $out)
: > $out
for v in "$@"; do
  if [ "$v" = "base" ]; then unset X265_AMD_LIB; else export X265_AMD_LIB=$PWD/build/variants/libx265_amd_$v.so; fi
  echo "== $v $BENCH_ARGS" >> $out
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --headline-only $BENCH_ARGS >> $out 2>&1 || { echo "variant $v failed" >> $out; exit 1; }
done
python3 - $out <<'P'
import json, sys
name = None
for line in open(sys.argv[1]):
    if line.startswith("== "):
        name = line[3:].strip()
    elif line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(f"{name:40s} kernel_ms avg {r['kernel_ms_avg']:.4f} min {r['kernel_ms_min']:.4f}  frac {r['frac']:.4f}  value {d['value']:.1f} Gcand/s")
P
