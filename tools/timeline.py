#!/usr/bin/env python3
"""Per-workgroup timeline of sad_surface_kernel<64> on the bench workload. Needs a library built with
-DXMI_EXP_TIMELINE (tools/build_variant.sh timeline -DXMI_EXP_TIMELINE; X265_AMD_LIB=build/variants/...).
s_memtime ticks are core-clock cycles (calibrated with tools/microbench/memtime_rate.hip), s_memrealtime is 100 MHz."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from x265_amd import capi, layout  # noqa: E402

L = capi.load()
capi.check(L.xmi_init(0))
W, H, R, PAD = 1920, 1088, 64, 80
f0, f1 = B.synth_frame(W, H, 0, 7), B.synth_frame(W, H, 1, 7)
ref = torch.from_numpy(B.pad_frame(f0, PAD)).cuda()
cur = torch.from_numpy(f1).cuda()
nctu = (W // 64) * (H // 64)
planes = [torch.empty(layout.plane_elems(nctu, R, lv), dtype=torch.int16 if lv < 2 else torch.int32, device="cuda") for lv in range(4)]
geom = capi.SadGeom(W, H, W, W + 2 * PAD, R)
s = torch.cuda.current_stream().cuda_stream
import time
torch.cuda.synchronize()
n_launch = int(os.environ.get("LAUNCHES", "30"))
wall0 = time.perf_counter()
for it in range(n_launch):     # sustained load, the last launch is analysed
    capi.check(L.xmi_sad_surface_dev(C.byref(geom), cur.data_ptr(), ref.data_ptr() + PAD * (W + 2 * PAD) + PAD, planes[0].data_ptr(),
                                     planes[1].data_ptr(), planes[2].data_ptr(), planes[3].data_ptr(), 0, nctu, s))
torch.cuda.synchronize()
print(f"{n_launch} launches back to back: {(time.perf_counter() - wall0) / n_launch * 1e6:.1f} us per launch (host clock)")
nwg = nctu * (R // 32 if R >= 64 else 1)      # workgroups of a launch: one per CTU and slice of the dx range
buf = (C.c_uint64 * (6 * nwg))()
L.xmi_debug_timeline.argtypes = [C.c_void_p, C.c_int]
assert L.xmi_debug_timeline(buf, nwg) == 0
t = np.frombuffer(buf, np.uint64).reshape(nwg, 6).astype(np.int64)
dur, stg, loop, tail = t[:, 3] - t[:, 0], t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
r0 = t[:, 4].min()
rs, re_ = t[:, 4] - r0, t[:, 5] - r0
first = rs < 0.25 * re_.max()                 # workgroups that started with the launch; the others took a slot that became free
steps = 136
qsad_cycles = steps * 64 * 16                 # issue cycles of one wavefront's v_qsad_pk_u16_u8 over the main loop (one wavefront per SIMD and workgroup)
print(f"kernel span {re_.max() / 100:.1f} us; {int(first.sum())} of {nwg} workgroups start within {rs[first].max() / 100:.2f} us")
q = [0, 10, 25, 50, 75, 90, 100]
print("start times, us, percentiles", q, ":", " ".join(f"{x / 100:.1f}" for x in np.percentile(rs, q)))
print("end times,   us, percentiles", q, ":", " ".join(f"{x / 100:.1f}" for x in np.percentile(re_, q)))
print("durations,   us, percentiles", q, ":", " ".join(f"{x / 100:.1f}" for x in np.percentile(re_ - rs, q)))
clk = dur / ((t[:, 5] - t[:, 4]) * 10.0)
print(f"core clock inside the kernel (s_memtime / s_memrealtime over each workgroup): median {float(np.median(clk)):.3f} GHz, "
      f"10th..90th percentile {float(np.percentile(clk, 10)):.3f}..{float(np.percentile(clk, 90)):.3f}")
for name, m in (("workgroups of the first round", first), ("workgroups of the second round", ~first)):
    if m.sum() == 0:
        continue
    print(f"{name}: n {m.sum()}  end at {re_[m].mean() / 100:.1f} us  core cycles: total {int(dur[m].mean())}  staging {int(stg[m].mean())}  "
          f"main loop {int(loop[m].mean())}  dx=+R pass {int(tail[m].mean())}")
print(f"two wavefronts share a SIMD: their v_qsad_pk_u16_u8 alone need 2 x {qsad_cycles} issue cycles per main loop; "
      f"main loop of the first round {int(loop[first].mean())} cycles ({2 * qsad_cycles / loop[first].mean():.1%} of them SAD issue)")
