#!/bin/bash
# usage: tools/probe/warm_round.sh   (GPU box) - every P round launched twice (build with -DXMI_EXP_WARM_ROUND): kernel trace of both
export TMPDIR=/tmp
repo=$PWD
out=$repo/gpurun_out/warm_round
mkdir -p $out
cat > /tmp/warm_run.py <<PY
import sys
sys.path.insert(0, "$repo/tools"); sys.path.insert(0, "$repo")
import bench_encoder
d = bench_encoder.run(1920, 1080, 10, 32, lanes=1)
print(d["p_picture_ms"], d["round_trips_per_p_picture"], d["stream_matches_fixture"])
PY
cd /tmp
X265_AMD_LIB=$repo/build/variants/libx265_amd_warm.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o warm -- python3 /tmp/warm_run.py > $out/run.log 2>&1 || exit 1
cd $repo
tail -1 $out/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "analysis_round_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
cold, warm = d[0::2], d[1::2]
n = min(len(cold), len(warm))
print("launch pairs", n, "cold mean us %.2f" % (sum(cold[:n]) / n), "warm mean us %.2f" % (sum(warm[:n]) / n))
big = [(c, w) for c, w in zip(cold, warm) if c > 30]
if big: print("pairs with cold > 30 us:", len(big), "cold %.2f warm %.2f" % (sum(c for c, _ in big) / len(big), sum(w for _, w in big) / len(big)))
PY
