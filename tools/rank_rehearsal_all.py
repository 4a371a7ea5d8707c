#!/usr/bin/env python3
"""EVERY rank of an N = 2 / 4 / 8 run of BASELINE configs[3] rehearsed on one GPU, in ONE process, with the shipped N > 1
default of bench.py (four steps in flight on own-queue streams, the world-of-one RCCL all-gather on the rank's one collective stream, the
pinned read-back, the host's offsets) -- for the default chunk rule c % N and for the balanced cut (sharding.balanced_assignment), plus the
one-stream / two-deep configuration the fallback would run.  Predicted strong scaling = the whole world's step in the same process and
configuration / the SLOWEST rank's step.  A rehearsal, not a scaling measurement: no second GPU is involved.

    python tools/rank_rehearsal_all.py [--steps 96] [--gather-stream side|main] > profiles/r06/rank_rehearsal_all.txt
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--gather-stream", default="side", choices=["side", "main"])
ap.add_argument("--json", default=None, help="also write the raw numbers here")
a = ap.parse_args()

n, c, dim = 1024, 128, 130
n_chunks = (n // c) ** 3
bpv = (c // 8) ** 3
d_field = torch.empty(n_chunks * dim ** 3, dtype=torch.float32, device="cuda")
raw = {}


def rehearse(depth, own_queue, gather_stream, assign):
    pipe = bench.GridPipeline(torch, vt, 0, depth, own_queue, gather_stream, tuning={"place_outputs": 8} if own_queue else None)   # as bench.py: placement trials in the default, none in the fallback
    try:
        pipe.exs[0].density_fill_device(vt.density_params("perlin3d", n), sharding.origins_of(n, c, range(n_chunks)), (dim, dim, dim), (1, dim, dim * dim),
                                        dim ** 3, d_field.data_ptr(), pipe.streams[0].cuda_stream)
        torch.cuda.synchronize()
        pipe.world_of_one_comm()
        pipe.bind(d_field.data_ptr(), n_chunks, c, 1, n_chunks, np.arange(n_chunks, dtype=np.intp), True)
        pipe.run_steps(2 * depth, False)
        tris, active = bench.per_chunk_counts(pipe.slots[0].ex, n_chunks, bpv)
        return bench.rehearse_ranks(torch, pipe, d_field, n_chunks, c, tris, active, steps=a.steps, assign=assign), tris, active
    finally:
        pipe.close()


print("RANK REHEARSAL, EVERY RANK (round 6; tools/rank_rehearsal_all.py, one process, one GPU: %s)" % torch.cuda.get_device_name(0))
print("perlin3d 1024^3 as 512 chunks of 128^3; ms per step, best of three regions of %d steps; T and the gathered counts checked for every rank" % a.steps)
for label, depth, own, gs in ((("SHIPPED N > 1 DEFAULT: 4 steps in flight, a hardware queue per context, every collective on ONE ordinary stream (--gather-stream side)" if a.gather_stream == "side" else
                               "4 steps in flight, a hardware queue per context, every collective behind its emit kernel on the step's own stream (--gather-stream main)"), 4, True, a.gather_stream),
                              ("FALLBACK: 2 steps in flight on one ordinary stream, every collective behind its emit kernel", 2, False, "main")):
    for assign in ("modulo", "balanced"):
        r, tris, active = rehearse(depth, own, gs, assign)
        raw["%s/%s" % ("default" if own else "fallback", assign)] = r
        print()
        print("%s -- chunks cut by %s" % (label, "c %% N" if assign == "modulo" else "the first step's triangle counts (balanced)"))
        print("  whole world (512 chunks) in the same process and configuration: %.4f ms per step" % r["world_step_ms"])
        for w in ("2", "4", "8"):
            q = r["ranks"][w]
            print("  N = %s  rank steps: %s" % (w, " ".join("%.4f" % x for x in q["step_ms"])))
            print("         slowest %.4f  mean %.4f  slowest / mean %.4f   triangles max / mean %.4f   active blocks max / mean %.4f   => predicted %.2fx"
                  % (q["slowest_ms"], q["mean_ms"], q["slowest_over_mean"], q["triangles_max_over_mean"][assign], q["active_blocks_max_over_mean"][assign],
                     q["predicted_scaling"]))
print()
print("triangles per chunk: min %d  median %d  max %d; empty chunks %d of %d" % (tris.min(), int(np.median(tris)), tris.max(), int((tris == 0).sum()), n_chunks))
if a.json:
    json.dump(raw, open(a.json, "w"), indent=1)
