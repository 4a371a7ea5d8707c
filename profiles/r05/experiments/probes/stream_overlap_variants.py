#!/usr/bin/env python3
"""Why do the steps of two contexts on two streams overlap in tools/rank_overlap_probe.py and not in bench.py?  The same 512-chunk step
with the streams made in different ways / orders; ms per step, two steps in flight.
    python tools/stream_overlap_variants.py <variant>     (one variant per process: the mapping of streams to hardware queues is a process-wide state)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

variant = sys.argv[1]
n, c, dim = 1024, 128, 130
org = sharding.chunk_origins(n, c, 0, 1)
if variant == "streams_first":
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
exs = [vt.Extractor(0), vt.Extractor(0)]
if variant == "rccl_first":
    exs[0].comm_init_rank(exs[0].comm_unique_id(), 0, 1)
if variant == "priority":
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
elif variant == "many":
    pool = [torch.cuda.Stream() for _ in range(8)]
    s1, s2 = pool[0], pool[5]
elif variant in ("cumask", "cumask_one"):
    import ctypes
    import glob
    hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))[0])
    mask = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))

    class Raw:
        def __init__(self):
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, mask)
            assert rc == 0, rc
            self.cuda_stream = h.value

        def synchronize(self):
            assert hip.hipStreamSynchronize(ctypes.c_void_p(self.cuda_stream)) == 0

    s1 = Raw()
    s2 = Raw() if variant == "cumask" else torch.cuda.Stream()
elif variant == "ctx_own":
    s1, s2 = torch.cuda.ExternalStream(exs[0].stream_handle()), torch.cuda.ExternalStream(exs[1].stream_handle())   # round 5: each on a queue of its own
elif variant != "streams_first":
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
if variant == "set_stream":
    torch.cuda.set_stream(s1)
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
exs[0].density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr(), s1.cuda_stream)
s1.synchronize()


def run(two, K):
    for i in range(K):
        st = s2 if (two and i % 2) else s1
        exs[i % 2].extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, st.cuda_stream, 0)
        if i >= 1:
            exs[(i - 1) % 2].extract_finish()
    exs[(K - 1) % 2].extract_finish()


out = []
for two in (False, True, False, True):
    run(two, 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(two, 100)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 100 * 1e3)
print("%-14s GPU_MAX_HW_QUEUES=%s  one stream %.4f %.4f   two streams %.4f %.4f" % (variant, os.environ.get("GPU_MAX_HW_QUEUES", "-"), out[0], out[2], out[1], out[3]))
