#!/usr/bin/env python3
"""Do the steps of two contexts overlap into each other's memory slack?  Two extractors take turns on the bench workload
(512 chunks of 128^3, resident): one stream (round 2's depth-2 pipeline) against one stream per context, for several residency
splits between the emit kernel (persistent, LDS-bound) and the classify kernel (streaming).  Prints ms per step.

    python tools/overlap_probe.py [--steps 20]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--indexed", type=int, default=0)
    ap.add_argument("--limit", type=int, default=0, help="use only the first K chunks (64: the share of one rank of eight)")
    args = ap.parse_args()
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding

    n, c = args.n, 128
    dim = c + 2
    origins = sharding.chunk_origins(n, c)
    if args.limit:
        origins = origins[:args.limit]
    exs = [vt.Extractor(0) for _ in range(2)]
    d = torch.empty(len(origins) * dim ** 3, dtype=torch.float32, device="cuda")
    exs[0].density_fill_device(vt.density_params("perlin3d", n), origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(two_streams, tuning, steps):
        for e in exs:
            e.set_output_mode(bool(args.indexed))
            e.set_tuning(stage_events=0, **tuning)
        def q(i):
            s = streams[i % 2] if two_streams else streams[0]
            exs[i % 2].extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(origins), dim ** 3, s.cuda_stream, 0)
        T = None
        for i in range(steps):
            q(i)
            if i >= 1:
                T = exs[(i - 1) % 2].extract_finish()
        T = exs[(steps - 1) % 2].extract_finish()
        return T

    base = dict(emit_wgs_per_cu=0, classify_wgs_per_cu=3, emit_once=1)
    variants = [
        ("one stream (r02 pipeline)", False, base),
        ("two streams, defaults", True, base),
        ("two streams, once=0", True, dict(base, emit_once=0)),
        ("two streams, classify uncapped", True, dict(base, classify_wgs_per_cu=0)),
        ("two streams, once=0 emit 3 + classify uncapped", True, dict(base, emit_once=0, emit_wgs_per_cu=3, classify_wgs_per_cu=0)),
        ("two streams, once=0 emit 3 + classify 4", True, dict(base, emit_once=0, emit_wgs_per_cu=3, classify_wgs_per_cu=4)),
        ("two streams, once=0 emit 2 + classify uncapped", True, dict(base, emit_once=0, emit_wgs_per_cu=2, classify_wgs_per_cu=0)),
        ("two streams, once=1 emit 2 + classify uncapped", True, dict(base, emit_wgs_per_cu=2, classify_wgs_per_cu=0)),
        ("two streams, once=1 emit 2 + classify 2", True, dict(base, emit_wgs_per_cu=2, classify_wgs_per_cu=2)),
        ("one stream again", False, base),
    ]
    for name, two, tun in variants:
        run(two, tun, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T = run(two, tun, args.steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print("%-56s %.4f ms per step   T = %d" % (name, ms, T), flush=True)


if __name__ == "__main__":
    main()
