#!/usr/bin/env python3
"""bench.py -- headline benchmark of the marching-cubes extraction path on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): a 1024^3-cell perlin3d grid
held as 8^3 chunks of 128^3 cells (130^3 samples each, 4.50 GB), resident in HBM.  One "step" =
one pass of the hot path over that batch: classify+count -> prefix scan / compaction -> fused
normals + triangle emit (--sweep: the experimental single-pass kernel), ending when the host knows T (and, for N > 1, after the RCCL all-gather
of the per-chunk {vertex, triangle} counts).  Weak scaling: every rank owns 512 chunks of a
1024 x 1024 x (1024*N) world, chunk c -> rank c % N.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
algorithmic bytes / HIP-event time on the kernels' own stream) and `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample of the same device-generated field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", "--n", dest="n", type=int, default=1024, help="cells per axis of one rank's grid")
    ap.add_argument("--chunk", type=int, default=128, help="cells per axis of a chunk")
    ap.add_argument("--kind", default="perlin3d", choices=["perlin3d", "fbm8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chunks", type=int, default=8, help="chunks the CPU oracle is timed on")
    ap.add_argument("--no-dense", action="store_true", help="A/B: force the per-block classify kernel")
    ap.add_argument("--sweep", action="store_true", help="A/B: the experimental single-pass kernel instead of classify -> scan -> emit")
    return ap.parse_args()


def cpu_baseline(d_field, dim, chunk, n_sample, kind_label):
    """Time the CPU oracle (oracle/mc_oracle.c, the restatement of the reference kernels: 'port')
    on the first n_sample chunks of the SAME device-generated field, all host cores + 1 core."""
    import ctypes
    import oracle
    L = oracle.lib()
    threads = oracle.max_threads()
    blocks = oracle.all_blocks(chunk, chunk, chunk)
    vols = [d_field[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy() for v in range(n_sample)]
    sx, sy, sz = 1, dim, dim * dim
    offs = np.empty(len(blocks) + 1, np.int32)
    # size the output once (count pass), reuse the buffer so page faults are not timed
    totals = [L.vto_extract_grid(oracle._p(v), sx, sy, sz, oracle._p(blocks), len(blocks), None, 0,
                                 oracle._p(offs), None, threads) for v in vols]
    buf = np.zeros(max(max(totals), 1), oracle.TRI_DTYPE)

    def run(nthreads, vs):
        t0 = time.perf_counter()
        tris = 0
        for v in vs:
            tris += L.vto_extract_grid(oracle._p(v), sx, sy, sz, oracle._p(blocks), len(blocks), oracle._p(buf),
                                       len(buf), oracle._p(offs), None, nthreads)
        return time.perf_counter() - t0, tris

    best_all = min(run(threads, vols)[0] for _ in range(3))
    t_one, _ = run(1, vols[:1])
    cells = chunk ** 3
    return {
        "value": round(n_sample * cells / best_all / 1e6, 2),
        "unit": "Mvoxels/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d chunks of %d^3 cells (%s, same device-generated field), OpenMP over blocks, best of 3"
                  % (n_sample, chunk, kind_label),
        "mtris_per_s": round(sum(totals) / best_all / 1e6, 2),
        "single_core_mvoxels_per_s": round(cells / t_one / 1e6, 2),
        "cpu_model": _cpu_model(),
        "host_cores": os.cpu_count(),
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the extraction path has no CPU fallback")
    # rehearsal hook for a one-GPU box: every rank on device 0, gloo instead of RCCL (which refuses
    # two ranks on one device); the driver's multi-GPU runs use neither variable
    if os.environ.get("VTMC_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("VTMC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    n, c = args.n, args.chunk
    dim = c + 2
    world_dims = (n, n, n * world)
    origins = sharding.chunk_origins(world_dims, c, rank, world)
    n_chunks = len(origins)
    bpv = (c // 8) ** 3
    ex = vt.Extractor(local)
    # one explicit (non-default) HIP stream for everything: the library's kernels, the counts copy and
    # the RCCL all-gather are ordered by it (a NULL handle would mean "the context's own stream" to the
    # library, which torch's collectives know nothing about)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    prm = vt.density_params(args.kind, n)

    # -- setup (untimed): density field generated on the device, chunk by chunk with halos -------
    d_field = torch.empty(n_chunks * dim ** 3, dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    ex.density_fill_device(prm, origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d_field.data_ptr(),
                           stream.cuda_stream)
    torch.cuda.synchronize()
    sampler_s = time.perf_counter() - t0

    flags = 2 if args.no_dense else 0
    if args.sweep:
        ex.set_tuning(sweep=1)
    fused = args.sweep and not args.no_dense
    counts_dev = torch.zeros((n_chunks, 2), dtype=torch.int32, device="cuda")
    stage_acc = {"classify": 0.0, "scan": 0.0, "emit": 0.0, "total": 0.0}

    def step(accumulate=False):
        T = ex.extract_volumes_device(d_field.data_ptr(), (c, c, c), (1, dim, dim * dim), n_chunks, dim ** 3,
                                      stream.cuda_stream, flags)
        if accumulate:
            for k, v in ex.last_stage_ms().items():
                stage_acc[k] += v
        if world > 1:
            # the only exchange of the path: all-gather of per-chunk {vertices, triangles}
            ex.copy_volume_counts_device(counts_dev.data_ptr(), n_chunks, stream.cuda_stream)
            gathered = sharding.allgather_counts(counts_dev if backend == "nccl" else counts_dev.cpu())
            offs = torch.cumsum(gathered.transpose(0, 1).reshape(-1, 2).to(torch.int64), 0)  # global chunk order
            return T, offs
        return T, None

    for _ in range(args.warmup):
        T, _ = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T, _ = step(accumulate=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el[0])
        tsum = torch.tensor([float(T)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        total_tris = float(tsum[0])
    else:
        total_tris = float(T)

    ms_per_step = elapsed / args.steps * 1e3
    cells_total = float(n) ** 3 * world
    value = cells_total / (elapsed / args.steps) / 1e6

    if rank == 0:
        # -- roofline of the dominant kernel (rank 0's launches, HIP events inside libvtmc) --------
        avg = {k: v / args.steps for k, v in stage_acc.items()}
        samples = n_chunks * dim ** 3
        _, off_ptr, _ = ex.device_results()
        offs = sharding.copy_device_u32(off_ptr, n_chunks * bpv + 1).astype(np.int64)
        n_active = int((np.diff(offs) > 0).sum())
        alg = {
            # DESIGN.md "algorithmic bytes": classify reads every sample once and writes one count per block
            "classify": 4.0 * samples + 4.0 * n_chunks * bpv,
            # emit reads the 10^3 tile of every non-empty block and writes 76 B per triangle
            "emit": 76.0 * T + 4000.0 * n_active,
            "scan": 4.0 * n_chunks * bpv * 3,
        }
        if fused:
            # single-pass kernel: every sample read once, one offset per block and every triangle written once
            alg["sweep"] = 4.0 * samples + 4.0 * n_chunks * bpv + 76.0 * T
            avg["sweep"] = avg["classify"]
        dom = "sweep" if fused else max(("classify", "emit"), key=lambda k: avg[k])
        ach = alg[dom] / (avg[dom] * 1e-3) / 1e9
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file) and n == 1024 and c == 128 and args.kind == "perlin3d":  # the PMC passes were taken on this workload
            try:
                traffic = json.load(open(pmc_file)).get(dom + "_kernel_hbm_bytes")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes": alg[dom], "avg_ms": round(avg[dom], 4)}
        per_kernel = {k: {"avg_ms": round(avg[k], 4), "alg_GBps": round(alg[k] / (avg[k] * 1e-3) / 1e9, 1) if avg[k] > 0 else None}
                      for k in (("sweep",) if fused else ("classify", "scan", "emit"))}
        # SURVEY.md 8d whole-path figure on one rank: 4*S + 76*T + 8*C over the device time of the three stages
        path_bytes = 4.0 * samples + 76.0 * T + 8.0 * n_chunks
        path = {"bytes": path_bytes, "device_ms": round(avg["total"], 4),
                "achieved_GBps": round(path_bytes / (avg["total"] * 1e-3) / 1e9, 1),
                "frac_of_peak": round(path_bytes / (avg["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "read_only_frac_of_peak": round(4.0 * samples / (avg["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            cpu = cpu_baseline(d_field, dim, c, min(args.cpu_sample_chunks, n_chunks), args.kind)
        out = {
            "metric": "marching-cubes extraction throughput on a %d^3 %s grid per GPU (Mvoxels/s)" % (n, args.kind),
            "value": round(value, 1),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s %d^3 cells per GPU as %d chunks of %d^3 (130^3 samples incl. halo), chunk c -> rank c %% N"
                                   % (args.kind, n, n_chunks, c),
                       "grid": n, "chunk": c, "chunks_per_gpu": n_chunks, "kind": args.kind, "seed": 1337,
                       "pipeline": "single-pass sweep kernel" if fused else ("classify(per-block) -> scan -> emit" if args.no_dense else "classify(dense) -> scan -> emit")},
            "mtris_per_s": round(total_tris / (elapsed / args.steps) / 1e6, 1),
            "triangles_per_gpu": int(T),
            "active_blocks_per_gpu": n_active,
            "roofline": roofline,
            "kernels": per_kernel,
            "path_roofline": path,
            "cpu_baseline": cpu,
            "sampler_s": round(sampler_s, 4),
        }
        print(json.dumps(out))
    ex.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
