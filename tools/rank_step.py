#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: times the step of rank r of a W-rank run of BASELINE configs[3]
(its 512 / W chunks of the 1024^3 world: queued extract + the one host wait, no collective) next to the
whole 512-chunk step, i.e. the speed-up the sharding leaves before the all-gather's ~tens of microseconds.
    python tools/rank_step.py [--comm | --comm-beside] [W ...]
--comm: every step also queues the C ABI's all-gather (a world-of-one RCCL communicator: its stream ordering, events and
the copy of the gathered array are real, the wire is not) -- the fixed cost of the exchange, behind the emit kernel on the
extract's stream (the library's default); --comm-beside: on the context's second stream beside the emit kernel (opt-in)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 1024, 128, 130
ex = vt.Extractor(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
prm = vt.density_params("perlin3d", n)
out = {}
args = [a for a in sys.argv[1:] if not a.startswith("--comm")]
with_comm = any(a.startswith("--comm") for a in sys.argv[1:])
if with_comm:
    ex.comm_init_rank(ex.comm_unique_id(), 0, 1)
    if "--comm-beside" in sys.argv[1:]:
        ex.set_tuning(gather_beside=1)   # opt-in: the collective on the second stream, beside the emit kernel
for W in [1] + [int(a) for a in args or ["2", "4", "8"]]:
    worst = 0.0
    for r in range(W):
        org = sharding.chunk_origins(n, c, r, W)
        d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
        ex.density_fill_device(prm, org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr(), stream.cuda_stream)

        gathered = torch.zeros(2 * len(org), dtype=torch.int32, device="cuda")
        gathered_host = torch.zeros(2 * len(org), dtype=torch.int32).pin_memory()

        def step():
            ex.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, stream.cuda_stream, 0)
            if with_comm:
                ex.allgather_volume_counts(gathered.data_ptr(), len(org), stream.cuda_stream)
                gathered_host.copy_(gathered, non_blocking=True)   # as bench.py: behind the collective, before the one wait
                stream.synchronize()
            return ex.extract_finish()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 30
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        st = ex.last_stage_ms()
        kern = st["total"]
        worst = max(worst, ms)
        if r in (0, W - 1):
            print("W=%d rank %d: %.4f ms/step (kernels %.4f = classify %.4f + scan %.4f + emit %.4f, host + gaps %.4f)"
                  % (W, r, ms, kern, st["classify"], st["scan"], st["emit"], ms - kern))
        del d
        if W >= 4 and r >= 1 and r < W - 1:
            continue
    out[W] = worst
for W, ms in out.items():
    print("W=%d: slowest rank %.4f ms -> speed-up over W=1 %.2fx (before the all-gather)" % (W, ms, out[1] / ms))
