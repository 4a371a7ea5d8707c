#!/usr/bin/env python3
"""Where the streaming classify kernel's time goes INSIDE a launch: every wave of a diagnostic build (-DVTMC_TIMELINE) leaves its start
and end time (s_memrealtime, 100 MHz) and its XCD; this prints the launch's span, the number of waves alive over time (ramp, plateau,
tail), brick durations by start time, and when each XCD ran dry.  Question (round 4): the kernel takes 780 us for 512 chunks and 129 us
for 64 (97.5 if it scaled) -- where are the 31 us?
    python -c "from volumetricterrain_amd import build; build.build_variant('tools/_ab/libvtmc_timeline.so', ['-DVTMC_TIMELINE'])"
    VTMC_LIB=tools/_ab/libvtmc_timeline.so python tools/classify_timeline.py [n_chunks ...]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import _lib, sharding  # noqa: E402

n, c, dim = 1024, 128, 130
L = _lib.load()
L.vtmc_debug_timeline.argtypes = [ctypes.c_void_p]
L.vtmc_debug_timeline.restype = ctypes.c_int32
ex = vt.Extractor(0)
org_all = sharding.chunk_origins(n, c)
d = torch.empty(len(org_all) * dim ** 3, dtype=torch.float32, device="cuda")
ex.density_fill_device(vt.density_params("perlin3d", n), org_all, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
tuning = [a for a in sys.argv[1:] if "=" in a]   # e.g. emit_ablate=1: the emit kernel of the step before leaves no dirty lines behind
for item in tuning:
    k, v = item.split("=")
    ex.set_tuning(**{k: int(v)})
    print("tuning: %s" % item)
for n_chunks in [int(a) for a in sys.argv[1:] if "=" not in a] or [512, 64]:
    n_bricks = n_chunks * 16 * 16 * 2
    buf = torch.zeros(n_bricks * 4, dtype=torch.int64, device="cuda")
    assert L.vtmc_debug_timeline(buf.data_ptr()) == 0
    for _ in range(4):
        ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), n_chunks, dim ** 3)
    ms = ex.last_stage_ms()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n_bricks, 4)
    assert L.vtmc_debug_timeline(None) == 0
    t0, t1, xcc = t[:, 0], t[:, 1], t[:, 3]
    origin = t0.min()
    s, e = (t0 - origin) / 100.0, (t1 - origin) / 100.0     # microseconds
    span = e.max()
    print("== %d chunks, %d bricks: classify stage %.1f us by HIP events; first wave start -> last wave end %.1f us" % (n_chunks, n_bricks, ms["classify"] * 1e3, span))
    dur = e - s
    print("   brick duration (wave start -> end): mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us" % (
        dur.mean(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
    # waves alive over time, 1 us bins
    nb = int(np.ceil(span)) + 1
    alive = np.zeros(nb + 1)
    np.add.at(alive, np.floor(s).astype(int), 1)
    np.add.at(alive, np.minimum(np.floor(e).astype(int) + 1, nb), -1)
    alive = np.cumsum(alive)[:nb]
    peak = alive.max()
    up = int(np.argmax(alive >= 0.9 * peak))
    down = nb - 1 - int(np.argmax(alive[::-1] >= 0.9 * peak))
    print("   waves alive: peak %d; reaches 90 %% of it at %d us, falls below 90 %% for good at %d us (tail %.0f us)" % (peak, up, down, span - down))
    # bricks finished per 10 % of the span, and their mean duration by start decile
    q = np.linspace(0, span, 11)
    fin = np.histogram(e, q)[0]
    print("   bricks finished per tenth of the span : " + " ".join("%6d" % x for x in fin))
    md = [dur[(s >= q[i]) & (s < q[i + 1])].mean() if ((s >= q[i]) & (s < q[i + 1])).any() else 0 for i in range(10)]
    print("   mean duration by start tenth (us)      : " + " ".join("%6.2f" % x for x in md))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("   XCC %d: %6d bricks, first start %6.1f, last end %6.1f us, mean duration %.2f" % (x, m.sum(), s[m].min(), e[m].max(), dur[m].mean()))
    # time between a workgroup slot's consecutive bricks is not observable here; the gap between the first start and the first end is
    print("   first end at %.1f us; last start at %.1f us" % (e.min(), s.max()))
