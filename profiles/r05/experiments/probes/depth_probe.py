#!/usr/bin/env python3
"""How many steps in flight?  One process, contexts on own-queue streams, d = 1 .. 4 of them taking turns over the same resident field (the
whole 1024^3 world, or one rank's share of W ranks), alternating over several rounds; every step's T checked.
    python tools/depth_probe.py [W=1] [--rounds 5] [--steps 40] [--comm]
--comm: every step also issues the C ABI's all-gather of its per-chunk counts (a world-of-one RCCL communicator the contexts share) on the step's
own stream and the host reads the gathered counts back, as bench.py does at N > 1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 1024, 128, 130
W = int(next((a for a in sys.argv[1:] if a.isdigit()), "1"))
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
K = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40
D = int(sys.argv[sys.argv.index("--max") + 1]) if "--max" in sys.argv else 4
exs = [vt.Extractor(0) for _ in range(D)]
sp = [e.stream_handle(own_queue=True) for e in exs]
org = sharding.chunk_origins(n, c, 0, W)
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
exs[0].density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
COMM = "--comm" in sys.argv
if COMM:
    exs[0].comm_init_rank(exs[0].comm_unique_id(), 0, 1)
    for e in exs[1:]:
        e.comm_share(exs[0])
g = [torch.zeros(2 * len(org), dtype=torch.int32, device="cuda") for _ in range(D)]
gh = [torch.zeros(2 * len(org), dtype=torch.int32).pin_memory() for _ in range(D)]
ss = [torch.cuda.ExternalStream(p) for p in sp]
evs = [torch.cuda.Event() for _ in range(D)]
T0 = None
for e in exs:   # buffers grown, kernels loaded
    T = e.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3)
    T0 = T if T0 is None else T0
    assert T == T0


def take(k):
    assert exs[k].extract_finish() == T0
    if COMM:   # the gathered counts of this step (the collective sits behind the emit kernel on the step's stream, the copy behind it)
        evs[k].synchronize()
        assert int(gh[k][1::2].sum()) == T0


def run(depth, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps + depth - 1):
        if i < steps:
            if i >= depth:
                take(i % depth)
            exs[i % depth].extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, sp[i % depth])
            if COMM:
                exs[i % depth].allgather_volume_counts(g[i % depth].data_ptr(), len(org), sp[i % depth])
                with torch.cuda.stream(ss[i % depth]):
                    gh[i % depth].copy_(g[i % depth], non_blocking=True)
                evs[i % depth].record(ss[i % depth])
    for j in range(max(0, steps - depth), steps):
        take(j % depth)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {k: [] for k in range(1, D + 1)}
for r in range(rounds):
    for depth in range(1, D + 1):
        run(depth, 6)
        res[depth].append(run(depth, K))
print("W = %d (%d chunks), T = %d, %d steps per measurement, %d rounds, ms per step (median | all):" % (W, len(org), T0, K, rounds))
for depth in range(1, D + 1):
    v = sorted(res[depth])
    print("  %d in flight: %.4f | %s" % (depth, v[len(v) // 2], " ".join("%.4f" % x for x in res[depth])))
torch.cuda.synchronize()
del evs, gh, g, ss   # torch objects that touched the contexts' streams go before the contexts
import gc  # noqa: E402
gc.collect()
torch.cuda.empty_cache()
for e in exs:
    e.close()
