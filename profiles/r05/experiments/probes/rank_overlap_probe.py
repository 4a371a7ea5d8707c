#!/usr/bin/env python3
"""One rank of eight (64 chunks of the 1024^3 world) in ONE process, every way of driving its steps in turn: one stream or a stream per
context, with and without the C ABI's all-gather (world-of-one communicator), the collective on the step's own stream or on a third one.
Every step's T is checked.  Rehearsal on one GPU -- not a scaling measurement.
    python tools/rank_overlap_probe.py [W=8] [--rounds 3] [--tune key=value,...] [--only two]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 1024, 128, 130
W = int(next((a for a in sys.argv[1:] if a.isdigit()), "8"))
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
exs = [vt.Extractor(0), vt.Extractor(0)]
if "--tune" in sys.argv:
    kv = {k: int(v) for k, v in (it.split("=") for it in sys.argv[sys.argv.index("--tune") + 1].split(","))}
    for e in exs:
        e.set_tuning(**kv)
    print("tuning:", kv)
exs[0].comm_init_rank(exs[0].comm_unique_id(), 0, 1)
exs[1].comm_share(exs[0])
s1, s2, s3 = torch.cuda.ExternalStream(exs[0].stream_handle()), torch.cuda.ExternalStream(exs[1].stream_handle()), torch.cuda.Stream()   # the contexts' own streams (a hardware queue each) + one for the collectives
org = sharding.chunk_origins(n, c, 0, W)
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
with torch.cuda.stream(s1):
    exs[0].density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr(), s1.cuda_stream)
s1.synchronize()
T0 = exs[0].extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, s1.cuda_stream)
exs[1].extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, s1.cuda_stream)
g = [torch.zeros(2 * len(org), dtype=torch.int32, device="cuda") for _ in range(2)]
gh = [torch.zeros(2 * len(org), dtype=torch.int32).pin_memory() for _ in range(2)]
evs = [torch.cuda.Event(), torch.cuda.Event()]


def run(mode, K):
    two = mode["streams"] == 2
    comm = mode["comm"]   # None | "own" (the step's stream) | "third" (one stream for every collective)

    def queue(i):
        e = exs[i % 2]
        st = s2 if (two and i % 2) else s1
        e.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, st.cuda_stream, 0)
        if comm:
            cs = s3 if comm == "third" else st
            e.allgather_volume_counts(g[i % 2].data_ptr(), len(org), cs.cuda_stream)   # another stream than the extract's: ordered behind its emit launch
            with torch.cuda.stream(cs):
                gh[i % 2].copy_(g[i % 2], non_blocking=True)
            evs[i % 2].record(cs)

    def take(i):
        if comm:
            evs[i % 2].synchronize()
        T = exs[i % 2].extract_finish()
        assert T == T0, (T, T0)
        if comm:
            assert int(gh[i % 2][1::2].sum()) == T0

    for i in range(K):
        queue(i)
        if i >= 1:
            take(i - 1)
    take(K - 1)


MODES = [("one stream, no collective", dict(streams=1, comm=None)),
         ("one stream, collective behind the emit kernel (bench.py's N > 1 default)", dict(streams=1, comm="own")),
         ("one stream, collective on a third stream", dict(streams=1, comm="third")),
         ("a stream per context, no collective", dict(streams=2, comm=None)),
         ("a stream per context, collective on the step's stream", dict(streams=2, comm="own")),
         ("a stream per context, every collective on a third stream", dict(streams=2, comm="third"))]
if "--only" in sys.argv:
    key = sys.argv[sys.argv.index("--only") + 1]
    MODES = [m for m in MODES if (key == "two") == (m[1]["streams"] == 2) and m[1]["comm"] is None] + [m for m in MODES if m[1]["streams"] == 1 and m[1]["comm"] is None and key == "two"]
res = {name: [] for name, _ in MODES}
for name, m in MODES:
    run(m, 6)
torch.cuda.synchronize()
for _ in range(rounds):
    for name, m in MODES:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        run(m, K)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / K * 1e3)
print("rank 0 of %d: %d chunks, T = %d per step; ms per step, two steps in flight, %d rounds of 200 steps per mode (alternating)" % (W, len(org), T0, rounds))
for name, _ in MODES:
    print("  %-75s %s   min %.4f" % (name, " ".join("%.4f" % x for x in res[name]), min(res[name])))
# torch objects that touched the contexts' streams (events, pinned tensors: PyTorch records an event on the stream when it frees one) go before
# the contexts -- left to the interpreter's tear-down they outlive the streams and the process dies in the HIP runtime
torch.cuda.synchronize()
del evs, gh, g, d, s1, s2, s3
import gc  # noqa: E402
gc.collect()
torch.cuda.empty_cache()
for e in reversed(exs):   # the borrower of the communicator first
    e.close()
