import sys, json
sys.path.insert(0, "tools")
import bench_analysis
for sp in (1, 2):
    r = bench_analysis.run_segments(1920, 1080, 32, 4, 8, speculate=sp, workers=2)
    print("speculate", sp, r.get("analysis_fps"))
