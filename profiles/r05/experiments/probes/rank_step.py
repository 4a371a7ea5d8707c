#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: times the step of rank r of a W-rank run of BASELINE configs[3]
(its 512 / W chunks of the 1024^3 world: queued extract + the one host wait, no collective) next to the
whole 512-chunk step, i.e. the speed-up the sharding leaves before the all-gather's ~tens of microseconds.
    python tools/rank_step.py [--comm | --comm-beside | --comm-side] [--pipeline] [W ...]
--pipeline: two contexts take turns, step k + 1 is queued before the host takes step k (what bench.py does by default; with --comm the
second context borrows the first one's communicator, vtmc_comm_share) -- the throughput per step instead of an isolated step's latency.
--comm: every step also queues the C ABI's all-gather (a world-of-one RCCL communicator: its stream ordering, events and
the copy of the gathered array are real, the wire is not) -- the fixed cost of the exchange, behind the emit kernel on the
extract's stream (the library's default); --comm-beside: on the context's second stream beside the emit kernel (opt-in);
--comm-side: on a second stream of the CALLER's, which the library orders behind the extract's emit launch;
--two-streams (with --pipeline): the two contexts queue their steps on a stream EACH, so step k + 1's classify kernel may start while step
k's emit kernel drains (at full size both stages want the whole chip and nothing overlaps; at rank size a quarter of a 0.14 ms kernel is
ramp-up and drain)."""
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
args = [a for a in sys.argv[1:] if not a.startswith("--")]
with_comm = any(a.startswith("--comm") for a in sys.argv[1:])
pipelined = "--pipeline" in sys.argv[1:]
side = torch.cuda.Stream() if "--comm-side" in sys.argv[1:] else None
two_streams = "--two-streams" in sys.argv[1:]
stream2 = torch.cuda.Stream() if two_streams else stream
no_stage_events = "--no-stage-events" in sys.argv[1:]   # only the step's total is timed on the device: no events between the three kernels
ex2 = vt.Extractor(0) if pipelined else None
if no_stage_events:
    for e in (ex, ex2):
        if e is not None:
            e.set_tuning(stage_events=0)
if with_comm:
    ex.comm_init_rank(ex.comm_unique_id(), 0, 1)
    if "--comm-beside" in sys.argv[1:]:
        ex.set_tuning(gather_beside=1)   # opt-in: the collective on the second stream, beside the emit kernel
    if ex2 is not None:
        ex2.comm_share(ex)
        if "--comm-beside" in sys.argv[1:]:
            ex2.set_tuning(gather_beside=1)
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

        K = 30
        if pipelined:   # two contexts, two result sets: queue step k + 1, then take step k
            slots = [(ex, gathered, gathered_host, torch.cuda.Event()), (ex2, torch.zeros_like(gathered), torch.zeros_like(gathered_host).pin_memory(), torch.cuda.Event())]

            def queue(i):
                e, g, gh, ev = slots[i % 2]
                st_i = stream if i % 2 == 0 else stream2
                e.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, st_i.cuda_stream, 0)
                if with_comm and side is not None:
                    e.allgather_volume_counts(g.data_ptr(), len(org), side.cuda_stream)
                    with torch.cuda.stream(side):
                        gh.copy_(g, non_blocking=True)
                    ev.record(side)
                elif with_comm:
                    e.allgather_volume_counts(g.data_ptr(), len(org), st_i.cuda_stream)
                    with torch.cuda.stream(st_i):
                        gh.copy_(g, non_blocking=True)
                    ev.record(st_i)

            def take(i):
                e, g, gh, ev = slots[i % 2]
                if with_comm:
                    ev.synchronize()
                return e.extract_finish()

            def run(n_steps):
                for i in range(n_steps):
                    queue(i)
                    if i >= 1:
                        take(i - 1)
                take(n_steps - 1)

            run(4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 60
            run(K)
        else:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
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
    print("W=%d: slowest rank %.4f ms -> speed-up over W=1 %.2fx (%s, %s)" % (W, ms, out[1] / ms, "with the world-of-one all-gather" if with_comm else "before the all-gather",
                                                                           "two steps in flight" if pipelined else "isolated steps"))
