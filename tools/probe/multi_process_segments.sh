#!/bin/bash
# usage: multi.sh NPROC PICTURES
n=$1; pics=$2
for i in $(seq 1 $n); do
  ( timeout -k 10 200 python3 tools/bench_encoder.py segments 1 $pics 2 0 2>/dev/null | tail -1 > gpurun_out/r03/mp_$i.txt ) &
done
wait
python3 - $n <<'P'
import json,sys
n=int(sys.argv[1]); tot=0; ptot=0
for i in range(1,n+1):
    d=json.loads(open(f"gpurun_out/r03/mp_{i}.txt").read())
    tot+=d["encode_fps"]; ptot+=d["p_pictures_per_s"]
print(f"{n} processes: sum encode_fps {tot:.1f}, sum p_pictures_per_s {ptot:.1f}")
P
