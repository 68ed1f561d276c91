#!/usr/bin/env python3
"""Board power and clocks while the SAD surface kernel runs back to back (GPU box): starts a launch loop in a child process
(tools/timeline.py's setup, LAUNCHES launches) and samples `rocm-smi` beside it.
usage: power_probe.py [variant ...]   ("base" = the product library, else build/variants/libx265_amd_<variant>.so)"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOP = r"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, %r)
import bench as B
from x265_amd import capi, layout
L = capi.load(); capi.check(L.xmi_init(0))
W, H, R, PAD = 1920, 1088, 64, 80
ref = torch.from_numpy(B.pad_frame(B.synth_frame(W, H, 0, 7), PAD)).cuda(); cur = torch.from_numpy(B.synth_frame(W, H, 1, 7)).cuda()
nctu = (W // 64) * (H // 64)
planes = [torch.empty(layout.plane_elems(nctu, R, lv), dtype=torch.int16 if lv < 2 else torch.int32, device="cuda") for lv in range(4)]
geom = capi.SadGeom(W, H, W, W + 2 * PAD, R); s = torch.cuda.current_stream().cuda_stream
t_end = time.time() + float(sys.argv[1]); n = 0; t0 = time.time()
print("started", flush=True)
while time.time() < t_end:
    for _ in range(200):
        capi.check(L.xmi_sad_surface_dev(C.byref(geom), cur.data_ptr(), ref.data_ptr() + PAD * (W + 2 * PAD) + PAD, planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), planes[3].data_ptr(), 0, nctu, s))
    torch.cuda.synchronize(); n += 200
print("us_per_launch %%.1f" %% ((time.time() - t0) / n * 1e6), flush=True)
""" % ROOT


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    vals = {}
    for line in out.splitlines():
        m = re.search(r"GPU\[0\]\s*:\s*(.*?):\s*(.*)", line)
        if m:
            vals[m.group(1).strip()] = m.group(2).strip()
    return vals


def main():
    for v in sys.argv[1:] or ["base"]:
        env = dict(os.environ)
        if v != "base":
            env["X265_AMD_LIB"] = os.path.join(ROOT, "build", "variants", f"libx265_amd_{v}.so")
        else:
            env.pop("X265_AMD_LIB", None)
        p = subprocess.Popen([sys.executable, "-c", LOOP, "9"], env=env, stdout=subprocess.PIPE, text=True)
        assert p.stdout.readline().strip() == "started"
        time.sleep(2.0)
        samples = [smi() for _ in range(4)]
        rate = p.stdout.readline().strip()
        p.wait()
        print(f"== {v}: {rate}")
        keys = [k for k in samples[0] if any(w in k.lower() for w in ("power", "sclk", "mclk", "fclk", "socclk", "temperature (sensor junction)", "temperature (sensor memory)"))]
        for k in keys:
            print(f"   {k}: " + " | ".join(s.get(k, "?") for s in samples))
    idle = smi()
    print("== idle: " + "; ".join(f"{k}: {v}" for k, v in idle.items() if "power" in k.lower() or "sclk" in k.lower()))


if __name__ == "__main__":
    main()
