#!/bin/bash
# usage: tools/copy_profiles.sh <tag> <round dir>   e.g. tools/copy_profiles.sh r01h profiles/r01
# Copies the summaries tools/profile_round.sh and tools/profile_secondary.sh left under gpurun_out/ into the tracked
# profiles directory and refreshes the PMC traffic figure bench.py reports.
set -e
tag=$1; dst=$2
P=gpurun_out/profile_$tag; S=gpurun_out/secondary_$tag
cp $P/kernel_stats.csv $dst/bench_kernel_stats.csv
grep -o '^{.*}' $P/bench_stats_run.log | tail -1 > $dst/bench_under_rocprof.json
cat $P/pmc_fetch.txt $P/pmc_write.txt $P/pmc_sq1.txt $P/pmc_sq2.txt $P/pmc_grbm.txt > $dst/pmc_summary.txt
[ -f $P/pmc_satd.txt ] && cat $P/pmc_satd.txt >> $dst/pmc_summary.txt
E=gpurun_out/evidence_$(basename $dst)
if [ -d $E ]; then cp $E/sad_launch_series.txt $E/sad_timeline.txt $E/microbench.txt $dst/; fi
cp $P/pmc_fetch/fetch_counter_collection.csv $dst/pmc_fetch_counter_collection.csv
cp $P/pmc_write/write_counter_collection.csv $dst/pmc_write_counter_collection.csv
if [ -d $S ]; then
  cp $S/kernel_stats.csv $dst/secondary_kernel_stats.csv
  grep '^{' $S/bench_kernels.jsonl > $dst/secondary_kernels.jsonl
  cp $S/pmc_summary.txt $dst/secondary_pmc_summary.txt
fi
python3 - "$P" "$dst" <<'PY'
import json, re, sys
p, dst = sys.argv[1], sys.argv[2]
def mean(path, name):
    txt = open(path).read()
    m = re.search(r"sad_surface_kernel<64[^>]*>\s*\n\s*" + name + r"\s+mean\s+([0-9.]+)", txt)
    return float(m.group(1))
w, f = mean(p + "/pmc_write.txt", "WRITE_SIZE"), mean(p + "/pmc_fetch.txt", "FETCH_SIZE")
path = "x265_amd/measured_traffic.json"
d = json.load(open(path))
src = d["1920x1088_R64"].get("source", "")
src = re.sub(r"^profiles/\w+/", dst.rstrip("/") + "/", src)
d["1920x1088_R64"].update(write_size_kb=w, fetch_size_kb=f, bytes_per_launch=int(round((w + 2 * f) * 1024)), source=src)
json.dump(d, open(path, "w"), indent=1)
print("traffic per launch:", d["1920x1088_R64"]["bytes_per_launch"])
PY
