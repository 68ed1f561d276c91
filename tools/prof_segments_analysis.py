#!/usr/bin/env python3
"""The 8-segment leg of bench.py's analysis block on its own (for runs under rocprofv3): usage prof_segments_analysis.py [LEVEL]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402,F401  (first HIP runtime of the process, as in bench.py)
import bench_analysis  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print(json.dumps(bench_analysis.run_segments(1920, 1080, 32, 3, 8, speculate=level, workers=2)), flush=True)
