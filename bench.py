#!/usr/bin/env python3
"""bench.py -- headline benchmark of the marching-cubes extraction path on MI355X.

--config grid1024 (default; BASELINE.json configs[2] at N = 1, configs[3] at N > 1 -- the one `metric`
is quoted on): a 1024^3-cell perlin3d grid held as 8^3 chunks of 128^3 cells (130^3 samples each,
4.50 GB), resident in HBM.  One "step" = one pass of the hot path over the rank's chunks: classify +
count -> prefix scan / compaction -> fused normals + triangle emit, ending when the host knows T and
(N > 1) every chunk's global offsets.  N > 1 is STRONG scaling by default, as configs[3] states it:
the world stays 1024^3, chunk c belongs to rank c % N (64 chunks each at N = 8), and the step ends
with the path's one collective, the RCCL all-gather of per-chunk {vertices, triangles} over xGMI
(vtmc_allgather_volume_counts, queued on the extract's stream; --scaling weak keeps 512 chunks per
rank on a 1024 x 1024 x 1024*N world instead).  Steps are independent passes over the same resident input and
run four deep by default (--pipeline 4): four contexts take turns, each on its own-queue stream, and step k + 3 is queued before the host
takes step k's T and offsets, so the device goes from step to step without waiting for the host -- the way a host that
extracts frame after frame would drive the library; every step still delivers its T, gather and offsets.  `value`
is that throughput; the latency of an isolated step is reported next to it (`step_latency_ms`, = --pipeline 1).

--config stream2048 (BASELINE.json configs[4]): a 2048^3-cell fbm8 world (36 GB of samples) streamed
as double-buffered batches of 128^3 chunks (two density buffers / contexts, each on its own-queue stream, the host a batch
ahead: sample(k + 1) and extract(k) are queued before batch k - 1's result is taken; the sampler leaves the samples'
sign bits and the classify stage reads those); one step = one pass over the rank's 4096 / N chunks, sampling included.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
algorithmic bytes / HIP-event time on the kernels' own stream) and `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample of the same device-generated field).
"""
import os

# the CPU leg pins its OpenMP threads; libgomp reads these when it is first mapped (import torch) and
# binds THIS thread to the first place from then on -- so the process's CPU share is read before that
_AFFINITY_AT_START = sorted(os.sched_getaffinity(0))
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import argparse  # noqa: E402
import json  # noqa: E402
import statistics  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
# vector-ALU ISSUE peak: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = one vector instruction per SIMD every four cycles, packed or not (the guide's
# 157.3 TFLOP/s FP32 vector figure = this x 2 halves of a v_pk_fma_f32 x 2 flops).  The sampler's model below counts issue slots: round 5 packed
# its fmas, so an octave pair's three fmas are three slots, not six.
VALU_PEAK_LANE_OPS = 39.3e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="grid1024", choices=["grid1024", "stream2048"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: the same 1024^3 world sharded (BASELINE configs[3]) or 512 chunks per rank")
    ap.add_argument("--grid", "--n", dest="n", type=int, default=None, help="cells per axis (reduced sizes for tests)")
    ap.add_argument("--chunk", type=int, default=128, help="cells per axis of a chunk")
    ap.add_argument("--batch", type=int, default=256, help="stream2048: chunks per double-buffered batch")
    ap.add_argument("--stream-one-queue", action="store_true", help="stream2048: one stream for both contexts (rounds 2-4) instead of each context's sampler + extract on its own-queue stream")
    ap.add_argument("--sampler-wgs", type=int, default=None, help="stream2048: residency cap of the sampler kernel (workgroups per CU; default: the stream's own)")
    ap.add_argument("--kind", default=None, choices=["perlin3d", "fbm8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chunks", type=int, default=32, help="chunks the CPU oracle is timed on")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0: physical cores in this process's CPU share, at most 16 per GPU)")
    ap.add_argument("--pipeline", type=int, default=4, choices=[1, 2, 3, 4],
                    help="grid1024: steps in flight = contexts taking turns (each with its own result buffers); step k + d - 1 is queued before the "
                         "host takes step k's result -- the device never waits for the host.  Default 4 (rounds 2-5a: 2): in ONE process, "
                         "alternating, 4 in flight run a rank's step of an 8-rank run 8 %% faster than 2 (0.2154 against 0.2341 ms with the collective "
                         "and the host's read-back in the loop) and the whole 1024^3 step 1-2 %% (tools/depth_probe.py, profiles/r05/depth_probe.txt).  "
                         "At N > 1 all contexts issue their all-gather through ONE communicator (vtmc_comm_share).  1: every step ends with its host "
                         "wait (the latency of an isolated step, also reported as step_latency_ms)")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="grid1024, --pipeline > 1: HIP streams the contexts queue their steps on.  2 (default): a stream EACH (its own-queue stream) -- step k + 1's "
                         "classify kernel starts on the CUs step k's emit kernel leaves as it drains (the emit kernel is bound by issued "
                         "instructions, the classify kernel by memory: profiles/r05/rank_overlap_probe.txt: -4 %% of a step at N = 1, -20 %% of a "
                         "rank's step of an 8-rank run); 1: one stream for both (rounds 2-4)")
    ap.add_argument("--gather-stream", default="main", choices=["side", "main"],
                    help="N > 1: the stream the all-gather and the copy of its result are queued on.  main (default): behind the emit kernel on the "
                         "extract's own stream; with --streams 2 the collectives of the rank's ONE communicator then alternate between two streams, "
                         "and the library chains them by events (a collective waits on the device for the one before it: csrc/comm.hip) -- RCCL sees "
                         "them one after the other, in the same order on every rank, exactly as on one stream.  side: ONE third stream for every "
                         "collective, each ordered behind its extract's emit launch by an event")
    ap.add_argument("--gather-beside", action="store_true",
                    help="N > 1, opt-in: the all-gather on the context's second stream beside the emit kernel (tuning key gather_beside)")
    ap.add_argument("--no-dense", action="store_true", help="A/B: force the per-block classify kernel")
    ap.add_argument("--no-indexed", action="store_true", help="skip the extra indexed-output steps at N = 1")
    args = ap.parse_args()
    if args.config == "stream2048":
        args.n = args.n or 2048
        args.kind = args.kind or "fbm8"
        args.steps = 3 if args.steps is None else args.steps
        args.warmup = 1 if args.warmup is None else args.warmup
    else:
        args.n = args.n or 1024
        args.kind = args.kind or "perlin3d"
        args.steps = 100 if args.steps is None else args.steps   # ~0.2 s of timed region: long enough for a power / busy sample to see it
        args.warmup = 5 if args.warmup is None else args.warmup
    if args.pipeline == 1:
        args.streams = 1
    return args


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (oracle/mc_oracle.c, "port") on this box's host cores
# ------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """Physical cores this process may use: /proc/cpuinfo (physical id, core id) pairs of the CPUs in
    sched_getaffinity, capped by the cgroup's cpu.max quota when one is set."""
    aff = _AFFINITY_AT_START
    cores, cur = {}, {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur:
                    cores[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
                cur = {}
                continue
            k, v = line.split(":", 1)
            cur[k.strip()] = v.strip()
        if "processor" in cur:
            cores[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
    except OSError:
        pass
    physical = len({cores.get(c, ("0", str(c))) for c in aff})
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period)))
    except (OSError, ValueError):
        pass
    return {"affinity_cpus": len(aff), "physical_cores": physical, "cgroup_quota_cores": quota}


def cpu_baseline(volumes, dim, chunk, kind_label, want_threads, n_gpus_on_box):
    """volumes: list of host float32 arrays (one 130^3 chunk each) of the SAME device-generated field.
    All-cores leg: >= 5 repetitions of >= 0.5 s each (the sample is looped inside a repetition until
    0.5 s have passed), every repetition's rate reported with min / median; then a one-thread leg."""
    import oracle
    L = oracle.lib()
    share = cpu_share()
    cap = 16 * max(1, n_gpus_on_box)   # the pool's CPU share per GPU; more threads than that measure the neighbours
    threads = want_threads or min(x for x in (share["physical_cores"], share["cgroup_quota_cores"] or 1 << 30, cap, oracle.max_threads()))
    blocks = oracle.all_blocks(chunk, chunk, chunk)
    sx, sy, sz = 1, dim, dim * dim
    offs = np.empty(len(blocks) + 1, np.int32)
    # count pass first: sizes the output once and touches every input page before anything is timed
    totals = [L.vto_extract_grid(oracle._p(v), sx, sy, sz, oracle._p(blocks), len(blocks), None, 0,
                                 oracle._p(offs), None, threads) for v in volumes]
    buf = np.zeros(max(max(totals), 1), oracle.TRI_DTYPE)
    cells = chunk ** 3

    def repetition(nthreads, vs, min_s=0.5):
        t0 = time.perf_counter()
        done = 0
        while True:
            for v in vs:
                L.vto_extract_grid(oracle._p(v), sx, sy, sz, oracle._p(blocks), len(blocks), oracle._p(buf),
                                   len(buf), oracle._p(offs), None, nthreads)
                done += 1
            dt = time.perf_counter() - t0
            if dt >= min_s:
                return done * cells / dt / 1e6, dt

    repetition(threads, volumes, 0.0)   # warm-up: thread team, output pages
    all_runs = [repetition(threads, volumes) for _ in range(5)]
    one_runs = [repetition(1, volumes[:1]) for _ in range(3)]
    rates = [r for r, _ in all_runs]
    return {
        "value": round(statistics.median(rates), 2),
        "unit": "Mvoxels/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d chunks of %d^3 cells (%s, same device-generated field), OpenMP over blocks pinned close/cores, "
                  "5 repetitions of >= 0.5 s, median" % (len(volumes), chunk, kind_label),
        "repetitions_mvoxels_per_s": [round(r, 2) for r in rates],
        "repetition_seconds": [round(s, 3) for _, s in all_runs],
        "min_mvoxels_per_s": round(min(rates), 2),
        "max_mvoxels_per_s": round(max(rates), 2),
        "mtris_per_s": round(statistics.median(rates) * sum(totals) / (len(volumes) * cells), 2),
        "single_core_mvoxels_per_s": round(statistics.median(r for r, _ in one_runs), 2),
        "single_core_repetitions": [round(r, 2) for r, _ in one_runs],
        "cpu_model": _cpu_model(),
        "host_cpus": os.cpu_count(),
        "cpu_share": share,
        "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
    }


def cpu_baseline_stream(n, chunk, kind, origins, want_threads, n_gpus_on_box, n_sample=8):
    """The CPU leg of the streaming config (BASELINE configs[4]): the oracle's per-sample sampler + its extractor, chunk by chunk, on a
    bounded sample of the same world -- `n_sample` chunks spread evenly over this rank's list -- both stages on the stated thread count.
    Repetitions of >= 0.5 s as cpu_baseline(); a one-thread leg on one chunk of the sample."""
    import ctypes
    import oracle
    L = oracle.lib()
    share = cpu_share()
    cap = 16 * max(1, n_gpus_on_box)
    threads = want_threads or min(x for x in (share["physical_cores"], share["cgroup_quota_cores"] or 1 << 30, cap, oracle.max_threads()))
    dim = chunk + 2
    prm = oracle.density_params(kind, n)
    pick = [origins[i] for i in sorted({int(round(k * (len(origins) - 1) / max(n_sample - 1, 1))) for k in range(n_sample)})]
    blocks = oracle.all_blocks(chunk, chunk, chunk)
    offs = np.empty(len(blocks) + 1, np.int32)
    vol = np.empty(dim ** 3, np.float32)
    sx, sy, sz = 1, dim, dim * dim

    def one_chunk(org, nthreads, buf):
        L.vto_density_fill_threads(ctypes.byref(prm), int(org[0]), int(org[1]), int(org[2]), dim, dim, dim, sx, sy, sz, oracle._p(vol), nthreads)
        return L.vto_extract_grid(oracle._p(vol), sx, sy, sz, oracle._p(blocks), len(blocks), oracle._p(buf) if buf is not None else None,
                                  len(buf) if buf is not None else 0, oracle._p(offs), None, nthreads)

    totals = [one_chunk(o, threads, None) for o in pick]      # count pass: sizes the output, touches every page
    buf = np.zeros(max(max(totals), 1), oracle.TRI_DTYPE)
    cells = chunk ** 3

    def repetition(nthreads, orgs, min_s=0.5):
        t0 = time.perf_counter()
        done = 0
        while True:
            for o in orgs:
                one_chunk(o, nthreads, buf)
                done += 1
            dt = time.perf_counter() - t0
            if dt >= min_s:
                return done * cells / dt / 1e6, dt

    repetition(threads, pick, 0.0)
    all_runs = [repetition(threads, pick) for _ in range(5)]
    one_runs = [repetition(1, pick[len(pick) // 2:len(pick) // 2 + 1], 0.0)]   # one chunk on one thread: seconds
    rates = [r for r, _ in all_runs]
    return {
        "value": round(statistics.median(rates), 2),
        "unit": "Mvoxels/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d chunks of %d^3 cells spread over the rank's %d (%s, sampled AND extracted by the oracle: oracle/density_ref.c + "
                  "oracle/mc_oracle.c), OpenMP pinned close/cores, 5 repetitions of >= 0.5 s, median" % (len(pick), chunk, len(origins), kind),
        "repetitions_mvoxels_per_s": [round(r, 2) for r in rates],
        "repetition_seconds": [round(s, 3) for _, s in all_runs],
        "min_mvoxels_per_s": round(min(rates), 2),
        "max_mvoxels_per_s": round(max(rates), 2),
        "triangles_in_sample": int(sum(totals)),
        "single_core_mvoxels_per_s": round(statistics.median(r for r, _ in one_runs), 2),
        "cpu_model": _cpu_model(),
        "host_cpus": os.cpu_count(),
        "cpu_share": share,
        "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
    }


# ------------------------------------------------------------------------------------------------
def pmc_traffic(dom, matches):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes -- a constant taken
    on the builder's lease of the same workload, NOT measured in this run (traffic_source says so)."""
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not matches or not os.path.exists(f):
        return None, None
    try:
        j = json.load(open(f))
        from volumetricterrain_amd import build as vt_build
        now = vt_build.kernel_source_hash()
        if j.get("kernel_source_sha256") != now:   # the kernels changed since the PMC passes were taken: no stale constant
            return None, "none: profiles/pmc_traffic.json was measured on other kernel sources (sha256 %s..., now %s...)" % (
                str(j.get("kernel_source_sha256"))[:12], now[:12])
        return j.get(dom + "_hbm_bytes"), "committed constant: " + j.get("source", "profiles/pmc_traffic.json")
    except Exception:
        return None, None


def init_distributed(args, torch, dist):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # main() starts the ranks itself when there is no launcher; a launcher with another world size is a mistake
        raise SystemExit("bench.py: --gpus %d inside a launcher's world of %d ranks (WORLD_SIZE): refusing to print a line for another N" % (args.gpus, world))
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
    return rank, world, local, backend


def native_comm(ex, rank, world, backend, dist, torch):
    """The library's own RCCL communicator (vtmc_comm_init_rank): rank 0 draws the id, torch.distributed
    only carries its 128 bytes to the other ranks.  Returns False when the native path is unavailable
    (VTMC_BENCH_NATIVE_RCCL=0, two ranks rehearsing on one device, or any rank failing to join -- the ranks
    agree on that with one all-reduce), the torch collective is used then."""
    if world == 1 or os.environ.get("VTMC_BENCH_NATIVE_RCCL", "1") != "1" or backend != "nccl":
        return False
    box = [None]
    if rank == 0:
        try:
            box[0] = ex.comm_unique_id()
        except Exception as e:   # noqa: BLE001 -- e.g. librccl missing: every rank takes the fallback
            print("native RCCL unavailable (%s): falling back to torch.distributed" % e, file=sys.stderr)
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return False
    ok = 1
    try:
        ex.comm_init_rank(box[0], rank, world)
    except Exception as e:   # noqa: BLE001
        print("rank %d: vtmc_comm_init_rank failed (%s)" % (rank, e), file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag[0]) == 0:
        if ok:
            ex.comm_destroy()
        return False
    return True


def reduce_max_sum(elapsed, tris, world, backend, torch, dist):
    if world == 1:
        return elapsed, float(tris)
    dev = "cuda" if backend == "nccl" else "cpu"
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ts = torch.tensor([float(tris)], dtype=torch.float64, device=dev)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    return float(el[0]), float(ts[0])


# ------------------------------------------------------------------------------------------------
# grid1024: resident grid, extraction only (configs[2] / configs[3])
# ------------------------------------------------------------------------------------------------
def run_grid(args, torch, dist):
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding
    rank, world, local, backend = init_distributed(args, torch, dist)
    n, c = args.n, args.chunk
    dim = c + 2
    strong = world > 1 and args.scaling == "strong"
    world_dims = (n, n, n) if (strong or world == 1) else (n, n, n * world)
    origins = sharding.chunk_origins(world_dims, c, rank, world)
    n_chunks = len(origins)
    n_chunks_total = (world_dims[0] // c) * (world_dims[1] // c) * (world_dims[2] // c)
    per_rank = (n_chunks_total + world - 1) // world   # slots per rank in the gathered array (zero-padded)
    bpv = (c // 8) ** 3
    depth = args.pipeline
    exs = [vt.Extractor(local) for _ in range(depth)]   # depth 2: the contexts take turns, each with its own result buffers; they share ONE communicator
    ex = exs[0]
    if args.gather_beside:
        for e in exs:
            e.set_tuning(gather_beside=1)
    if os.environ.get("VTMC_BENCH_TUNING"):   # A/B of kernel variants under the bench's sustained load, e.g. VTMC_BENCH_TUNING="emit_once=0"
        kv = {k: int(v) for k, v in (item.split("=") for item in os.environ["VTMC_BENCH_TUNING"].split(","))}
        for e in exs:
            e.set_tuning(**kv)
    # the contexts' streams (vtmc_context_stream; --streams 2: the ones on a hardware queue of their own -- ordinary HIP streams may share a
    # queue and then run strictly in turn), wrapped for torch's copies and events: one per context in flight (--streams 2: the steps of the contexts
    # overlap where one kernel drains and the next ramps up), or the first context's for everything (--streams 1)
    # (never torch's CURRENT stream: what torch allocates while a stream is current belongs to that stream in its caching allocator, and
    # these streams die with their contexts -- torch work is put on them with `with torch.cuda.stream(...)` only where it must be)
    streams = [torch.cuda.ExternalStream(exs[i].stream_handle(own_queue=args.streams == 2)) for i in range(depth if args.streams == 2 else 1)]
    stream = streams[0]
    prm = vt.density_params(args.kind, n)

    # -- setup (untimed): density field generated on the device, chunk by chunk with halos -------
    d_field = torch.empty(max(n_chunks, 1) * dim ** 3, dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    ex.density_fill_device(prm, origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d_field.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    sampler_s = time.perf_counter() - t0
    sampler_kernel_ms = ex.last_fill_ms()

    # ONE communicator per rank, whatever the depth: the first context owns it, the others issue their all-gathers through it
    # (vtmc_comm_share) -- on the one stream everything runs on, the collectives of consecutive steps are in program order on every rank
    native = native_comm(exs[0], rank, world, backend, dist, torch)
    # rehearsal hook for a one-GPU box: the N > 1 host path (collective, pinned copy, offsets) through a world-of-one communicator
    force_comm = world == 1 and os.environ.get("VTMC_BENCH_FORCE_COMM") == "1"
    if force_comm:
        exs[0].comm_init_rank(exs[0].comm_unique_id(), 0, 1)
        native = True
    if native:
        for e in exs[1:]:
            e.comm_share(exs[0])
    exchange = world > 1 or force_comm
    flags = 2 if args.no_dense else 0
    # rank r holds chunks r, r + N, ...: chunk c sits in slot c // N of rank c % N
    perm = np.array([(ch % world) * per_rank + ch // world for ch in range(n_chunks_total)], np.intp)
    d_ptr = d_field.data_ptr()

    class Slot:   # what one step in flight owns besides its context
        def __init__(self, e, st):
            self.ex = e
            self.stream = st
            self.s_ptr = st.cuda_stream
            self.gathered = torch.zeros((world, per_rank, 2), dtype=torch.int32, device="cuda")
            self.counts_dev = torch.zeros((per_rank, 2), dtype=torch.int32, device="cuda")   # fallback collective's send buffer
            self.gathered_host = torch.zeros((world, per_rank, 2), dtype=torch.int32).pin_memory()
            self.host_rows = self.gathered_host.numpy().reshape(-1, 2)   # a view of the pinned words
            self.offs = np.zeros((n_chunks_total + 1, 2), np.int64)
            self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.copied = torch.cuda.Event()
            self.timed_gather = False
            self.timed_stages = True

    slots = [Slot(e, streams[i % len(streams)]) for i, e in enumerate(exs)]
    side = torch.cuda.Stream() if (exchange and native and args.gather_stream == "side" and not args.gather_beside) else None
    stage_acc = {"classify": 0.0, "scan": 0.0, "emit": 0.0, "total": 0.0}
    stage_steps = [0]   # steps whose three kernels were timed one by one
    # N > 1: a rank's kernels take 0.14 ms each, and the HIP events between them cost 2 % of its step (tools/rank_step.py): only every eighth
    # step carries them (the same steps whose collective is timed); at N = 1 every step does (0.3 % of a step)
    sample_stages = world > 1
    gather_ms = []

    def queue(sl, timed_gather=False, timed_stages=True):
        """classify -> scan -> emit [-> all-gather -> copy of the gathered pairs into pinned words]; nothing waits."""
        if sample_stages:
            sl.ex.set_tuning(stage_events=1 if timed_stages else 0)
        sl.timed_stages = timed_stages
        sl.ex.extract_volumes_device_async(d_ptr, (c, c, c), (1, dim, dim * dim), n_chunks, dim ** 3, sl.s_ptr, flags)
        if exchange:
            sl.timed_gather = timed_gather
            if timed_gather:
                sl.ev0.record(sl.stream)
            if native and side is not None:
                # the path's one collective, behind the C ABI, on a stream of its own: the library orders it behind this extract's emit
                # launch (an event), the copy of the gathered pairs follows it there, and the main stream goes straight on to the next
                # step's classify kernel -- nothing on it waits for the collective, only the host does (`copied`)
                sl.ex.allgather_volume_counts(sl.gathered.data_ptr(), per_rank, side.cuda_stream)
                if timed_gather:
                    sl.ev1.record(side)
                with torch.cuda.stream(side):
                    sl.gathered_host.copy_(sl.gathered, non_blocking=True)
                sl.copied.record(side)
                return
            with torch.cuda.stream(sl.stream):
                if native:     # --gather-stream main: behind the emit kernel on the extract's own stream (or beside it with --gather-beside)
                    sl.ex.allgather_volume_counts(sl.gathered.data_ptr(), per_rank, sl.s_ptr)
                else:
                    sl.ex.copy_volume_counts_device(sl.counts_dev.data_ptr(), per_rank, sl.s_ptr)
                    if backend == "nccl":
                        dist.all_gather_into_tensor(sl.gathered.view(-1), sl.counts_dev.view(-1))
                    else:   # gloo rehearsal: through the host (torch allocates nothing while a context's stream is current: see above)
                        sl.stream.synchronize()
                        with torch.cuda.stream(torch.cuda.default_stream()):
                            g_dev = sharding.allgather_counts(sl.counts_dev.cpu()).to("cuda")
                            torch.cuda.default_stream().synchronize()
                        sl.gathered.copy_(g_dev)
                        sl.stream.synchronize()
                        del g_dev
                if timed_gather:
                    sl.ev1.record(sl.stream)
                sl.gathered_host.copy_(sl.gathered, non_blocking=True)
                sl.copied.record(sl.stream)

    def complete(sl, accumulate=False):
        """The one host wait of a step: its gathered pairs are in pinned memory (or, without an exchange, its extract is done)."""
        if exchange:
            sl.copied.synchronize()
        T = sl.ex.extract_finish()
        offs = None
        if exchange:   # every rank's local exclusive scan over the chunks in global order
            np.cumsum(sl.host_rows[perm], axis=0, dtype=np.int64, out=sl.offs[1:])
            offs = sl.offs
        if accumulate and sl.timed_stages:
            ms = sl.ex.last_stage_ms()
            for k in stage_acc:
                stage_acc[k] += ms[k]
            stage_steps[0] += 1
            if exchange and sl.timed_gather:
                gather_ms.append(sl.ev0.elapsed_time(sl.ev1))
        return T, offs

    def run_steps(k_steps, accumulate):
        """k_steps steps, `depth` in flight: step i + 1 is queued before the host takes step i."""
        T = offs = None
        for i in range(k_steps):
            queue(slots[i % depth], timed_gather=accumulate and exchange and i % 8 == 0,   # the collective's own events on every eighth step
                  timed_stages=(not sample_stages) or i % 8 == 0)
            if i >= depth - 1:
                T, offs = complete(slots[(i - depth + 1) % depth], accumulate)
        for i in range(max(k_steps - depth + 1, 0), k_steps):
            T, offs = complete(slots[i % depth], accumulate)
        return T, offs

    n_warm = max(args.warmup, depth)   # every context once at least: output buffers grow to their size, RCCL builds its channels
    T, offs = run_steps(n_warm, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T, offs = run_steps(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_tris = reduce_max_sum(elapsed, T, world, backend, torch, dist)
    if exchange:
        assert int(offs[-1, 1]) == int(total_tris), "gathered chunk counts do not add up to the ranks' triangle totals"

    ms_per_step = elapsed / args.steps * 1e3
    # The kernels' own durations.  With a stream per context the HIP events around a kernel also see the time it waits for CUs beside the
    # other context's kernels (a classify kernel "takes" 1.9 ms there): the roofline of a KERNEL needs it alone on the chip.  A second
    # region of the same K steps, the same contexts taking turns, all on the first context's stream (rounds 1-4's timed region, two contexts then).
    live = {k: v / max(stage_steps[0], 1) for k, v in stage_acc.items()}
    serial_ms_per_step = ms_per_step

    def set_streams(one):
        for i, sl in enumerate(slots):
            st = streams[0] if one else streams[i % len(streams)]
            sl.stream, sl.s_ptr = st, st.cuda_stream

    if len(streams) > 1:
        set_streams(True)
        for k in stage_acc:
            stage_acc[k] = 0.0
        stage_steps[0] = 0
        run_steps(depth, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(args.steps, True)
        torch.cuda.synchronize()
        serial_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
        set_streams(False)
    # the latency of an isolated step (queue, one host wait), outside the timed region: what --pipeline 1 measures
    lat = []
    iso = {"classify": [], "scan": [], "emit": [], "total": []}   # the three kernels with nothing beside them (no second step in flight)
    for _ in range(10):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        queue(slots[0])
        complete(slots[0])
        lat.append((time.perf_counter() - t1) * 1e3)
        for k, v in slots[0].ex.last_stage_ms().items():
            iso[k].append(v)
    step_latency_ms = statistics.median(lat)
    iso = {k: statistics.median(v) for k, v in iso.items()}
    cells_total = float(world_dims[0]) * world_dims[1] * world_dims[2]
    value = cells_total / (elapsed / args.steps) / 1e6

    if rank == 0:
        # -- roofline of the dominant kernel (rank 0's launches, HIP events inside libvtmc) --------
        avg = {k: v / max(stage_steps[0], 1) for k, v in stage_acc.items()}
        samples = n_chunks * dim ** 3
        _, off_ptr, _ = ex.device_results()
        boffs = ex.copy_u32(off_ptr, n_chunks * bpv + 1).astype(np.int64)
        n_active = int((np.diff(boffs) > 0).sum())
        alg = {
            # DESIGN.md "algorithmic bytes": classify reads every sample once and writes one count per block
            "classify": 4.0 * samples + 4.0 * n_chunks * bpv,
            # emit reads the 10^3 tile of every non-empty block and writes 76 B per triangle
            "emit": 76.0 * T + 4000.0 * n_active,
            "scan": 4.0 * n_chunks * bpv * 3,
        }
        dom = max(("classify", "emit"), key=lambda k: avg[k])
        ach = alg[dom] / (avg[dom] * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(dom + "_kernel", world == 1 and n == 1024 and c == 128 and args.kind == "perlin3d" and not args.no_dense)
        roofline = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes": alg[dom], "avg_ms": round(avg[dom], 4)}
        roofline["measured"] = ("HIP events on the kernels' stream over %d steps, the contexts taking turns on ONE stream" % args.steps) + (
            "" if len(streams) == 1 else " -- a second region behind the timed one (%.4f ms per step there): in the timed region the contexts have a "
            "stream each and a kernel's events also see the time it shares the chip with the other context's kernels, see kernels.*.two_queue_span_ms" % serial_ms_per_step)
        per_kernel = {k: {"avg_ms": round(avg[k], 4), "alg_GBps": round(alg[k] / (avg[k] * 1e-3) / 1e9, 1) if avg[k] > 0 else None,
                          "two_queue_span_ms": round(live[k], 4) if len(streams) > 1 else None, "isolated_step_ms": round(iso[k], 4)}
                      for k in ("classify", "scan", "emit")}
        # SURVEY.md 8d whole-path figure on one rank: 4*S + 76*T + 8*C over the time a step takes (wall clock of the timed region: with
        # two streams the kernels of neighbouring steps overlap, the sum of their durations is more than a step)
        path_bytes = 4.0 * samples + 76.0 * T + 8.0 * n_chunks
        path = {"bytes": path_bytes, "step_ms": round(ms_per_step, 4), "kernel_ms_sum": round(avg["total"], 4),
                "achieved_GBps": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                "frac_of_peak": round(path_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "read_only_frac_of_peak": round(4.0 * samples / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # the same workload in the welded (indexed) output format -- 24 B per vertex + 12 B per triangle instead of 76 B
        # per triangle: a few steps after the timed region, N = 1 only (not part of `value`)
        indexed = None
        if world == 1 and not args.no_indexed:
            for e in exs:
                e.set_output_mode(True)
            try:
                # driven exactly as the timed soup steps: `depth` steps in flight, the contexts taking turns
                def run_indexed(k_steps, acc):
                    Ti = None
                    for i in range(k_steps + depth - 1):
                        if i < k_steps:
                            exs[i % depth].extract_volumes_device_async(d_ptr, (c, c, c), (1, dim, dim * dim), n_chunks, dim ** 3, slots[i % depth].s_ptr, flags)
                        if i >= depth - 1:
                            e = exs[(i - depth + 1) % depth]
                            Ti = e.extract_finish()
                            if acc is not None:
                                for k, v in e.last_stage_ms().items():
                                    acc[k] += v / k_steps
                    return Ti

                run_indexed(2 * depth, None)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                K = max(4, args.steps // 2)
                Ti = run_indexed(K, None)
                torch.cuda.synchronize()
                ms_i = (time.perf_counter() - t0) / K * 1e3
                acc = {"classify": 0.0, "scan": 0.0, "emit": 0.0, "total": 0.0}   # the kernels' own durations: the same steps on one stream (see above)
                set_streams(True)
                run_indexed(K, acc)
                torch.cuda.synchronize()
                set_streams(False)
                V = ex.last_vertex_count()
                ibytes = 4.0 * samples + 24.0 * V + 12.0 * Ti + 8.0 * n_chunks
                indexed = {"ms_per_step": round(ms_i, 4), "mvoxels_per_s": round(cells_total / (ms_i * 1e-3) / 1e6, 1),
                           "vertices": int(V), "triangles": int(Ti), "output_bytes": 24.0 * V + 12.0 * Ti,
                           "output_bytes_vs_soup": round((24.0 * V + 12.0 * Ti) / (76.0 * Ti), 4),
                           "kernels_ms": {k: round(v, 4) for k, v in acc.items()},
                           "path_roofline": {"bytes": ibytes, "achieved_GBps": round(ibytes / (ms_i * 1e-3) / 1e9, 1),
                                             "frac_of_peak": round(ibytes / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                             "read_only_frac_of_peak": round(4.0 * samples / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                           "speedup_over_soup_step": round(ms_per_step / ms_i, 3)}
            finally:
                for e in exs:
                    e.set_output_mode(False)
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            k = min(args.cpu_sample_chunks, n_chunks)
            vols = [d_field[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy() for v in range(k)]
            cpu = cpu_baseline(vols, dim, c, args.kind, args.cpu_threads, torch.cuda.device_count())
        if world == 1:
            wl = "%s %d^3 cells as %d chunks of %d^3 (%d^3 samples incl. halo), resident in HBM" % (args.kind, n, n_chunks, c, dim)
        elif strong:
            wl = ("%s %d^3 cells as %d chunks of %d^3, chunk c -> rank c %% %d (%d per rank), all-gather of per-chunk counts"
                  % (args.kind, n, n_chunks_total, c, world, n_chunks))
        else:
            wl = "%s %d^3 cells per GPU as %d chunks of %d^3, chunk c -> rank c %% N (weak scaling world 1024 x 1024 x 1024N)" % (args.kind, n, n_chunks, c)
        out = {
            "metric": "marching-cubes extraction throughput on a %d^3 %s grid (Mvoxels/s)" % (n, args.kind),
            "value": round(value, 1),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": n_warm,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl, "grid": n, "chunk": c, "chunks_per_gpu": n_chunks, "kind": args.kind, "seed": 1337,
                       "pipeline": "classify(per-block) -> scan -> emit" if args.no_dense else "classify(dense) -> scan -> emit",
                       "collective": None if not exchange else (("rccl all-gather via libvtmc (vtmc_allgather_volume_counts), %s" % ("on a second stream behind the emit launch's event" if side is not None else "on the extract's stream")) if native
                                                              else "torch.distributed all_gather (%s)" % backend)},
            "mtris_per_s": round(total_tris / (elapsed / args.steps) / 1e6, 1),
            "triangles_rank0": int(T),
            "triangles_total": int(total_tris),
            "active_blocks_rank0": n_active,
            "roofline": roofline,
            "kernels": per_kernel,
            "path_roofline": path,
            "allgather_ms": None if not gather_ms else {"avg": round(statistics.mean(gather_ms), 4), "max": round(max(gather_ms), 4),
                                                     "note": "rank 0, HIP events from the end of the emit kernel to the end of the collective (sampled on every eighth step).  With --gather-stream main (default) the collective sits on the extract's stream, between this step's emit kernel and the next step's classify kernel; with --gather-stream side it runs on a second stream beside the latter"},
            "pipeline_depth": depth,
            "streams": len(streams),
            "one_stream_ms_per_step": round(serial_ms_per_step, 4),
            "step_latency_ms": round(step_latency_ms, 4),
            "indexed_output": indexed,
            "cpu_baseline": cpu,
            "sampler_s": round(sampler_s, 4),
            "sampler_kernel_ms": round(sampler_kernel_ms, 3),
        }
        emit_line(json.dumps(out))
    # nothing of torch's may outlive the contexts' streams: drain, drop every tensor / event that was used on them, hand cached blocks back
    dbg = (lambda m: print("bench.py[%d]: %s" % (rank, m), file=sys.stderr, flush=True)) if os.environ.get("VTMC_BENCH_DEBUG") else (lambda m: None)
    torch.cuda.synchronize()
    dbg("synchronised")
    sl = None   # a loop variable above may still hold the last slot (its events were recorded on a context's stream)
    del slots, d_field
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    dbg("torch side released")
    for e in exs:
        e.close()
    dbg("contexts closed")
    if world > 1:
        dist.destroy_process_group()
    dbg("process group destroyed")
    del streams, stream
    dbg("streams dropped")


# ------------------------------------------------------------------------------------------------
# stream2048: sampler + extractor, double-buffered batches (configs[4])
# ------------------------------------------------------------------------------------------------
def run_stream(args, torch, dist):
    from volumetricterrain_amd.streaming import ChunkStream
    rank, world, local, backend = init_distributed(args, torch, dist)
    n, c = args.n, args.chunk
    dim = c + 2
    with ChunkStream(n, c, args.batch, args.kind, n, rank=rank, world_size=world, device=local, sampler_wgs_per_cu=args.sampler_wgs, two_queues=not args.stream_one_queue) as st:
        n_chunks = len(st.origins)
        origins_rank0 = [tuple(int(v) for v in o) for o in st.origins]
        cells_total = float(n) ** 3
        for _ in range(max(args.warmup, 1)):   # buffers grow to their steady size
            st.run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            total, counts = st.run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        elapsed, total_tris = reduce_max_sum(elapsed, total, world, backend, torch, dist)
        if world > 1:   # the same single exchange as config 4: per-chunk counts of every rank
            from volumetricterrain_amd import sharding
            per_rank = ((n // c) ** 3 + world - 1) // world
            loc = torch.zeros((per_rank, 2), dtype=torch.int32)
            loc[:n_chunks] = torch.from_numpy(counts.astype(np.int32))
            g = sharding.allgather_counts(loc.cuda() if backend == "nccl" else loc)
            assert int(g[..., 1].sum()) == int(total_tris)
        # diagnostic pass, serialised (fill waits, then extract): per-kernel device times by HIP events
        fill_ms, stage = [], {"classify": 0.0, "scan": 0.0, "emit": 0.0, "total": 0.0}
        t0 = time.perf_counter()
        for k in range(st.n_batches()):
            org = st._origins_of(k)
            st._ex[0].density_fill_device(st.params, org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, st._buf[0].data_ptr())
            fill_ms.append(st._ex[0].last_fill_ms())
            st._ex[0].extract_volumes_device(st._buf[0].data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3)
            for kk, v in st._ex[0].last_stage_ms().items():
                stage[kk] += v
        serial_s = time.perf_counter() - t0
    if rank == 0:
        step_s = elapsed / args.steps
        samples = n_chunks * dim ** 3
        nb = len(fill_ms)
        kern = {"density_column_kernel": sum(fill_ms), "classify_dense_kernel": stage["classify"], "scan": stage["scan"], "emit_kernel": stage["emit"]}
        dom = max(kern, key=kern.get)
        octaves = st.params.octaves
        if dom == "density_column_kernel":
            alg_bytes = 4.0 * samples / nb          # per launch: every sample written once
        elif dom == "classify_dense_kernel":
            alg_bytes = (4.0 * samples + 4.0 * n_chunks * st.bpv) / nb
        else:
            alg_bytes = 76.0 * total / nb           # + 4000 B per non-empty block, not counted here
        avg_ms = kern[dom] / nb
        ach = alg_bytes / (avg_ms * 1e-3) / 1e9
        # the sampler's own bound is the vector ALU's issue slots.  Vector instructions per sample, counted in the kernel's ISA (DESIGN.md 4,
        # profiles/r05/sampler_valu_bound.txt): 3 packed fmas per octave PAIR + 7 for the step (sum, sign compare, two v_writelane for the sign
        # word, store address, LDS address, one spare); a rebuilt octave is 26 (a face 21: 4 addresses, 2 + 8 for the corner dot products, 6 packed
        # lerps, 1 packed amplitude; 5 for the derived constants), at f * lacunarity^o rebuilds per sample and octave; a step with any rebuild
        # costs 7 (mask words to scalars, the constants' sum); a walk starts with two faces per octave and ~25 per octave of column set-up.
        # The counter (SQ_INSTS_VALU) read 43 per sample for config 5 before the last two trims (-3), this model says 38.
        rates = [min(1.0, st.params.frequency * (st.params.lacunarity ** o)) for o in range(octaves)]
        lane_ops = samples * (1.5 * octaves + 7.0 + 26.0 * sum(rates) + 7.0 * max(rates) + (2 * 26.0 + 25.0) * octaves / st.dim)
        sampler_flops = samples * (2.0 * 3.0 * octaves + 2.0 * 34.0 * sum(rates))   # the arithmetic itself: 3 fmas per octave and sample, ~34 flop-pairs a face
        out = {
            "metric": "streamed sampler + marching-cubes extraction throughput on a %d^3 %s world (Mvoxels/s)" % (n, args.kind),
            "value": round(cells_total / step_s / 1e6, 1),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": max(args.warmup, 1),
            "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s %d^3 cells streamed as %d chunks of %d^3 per rank (chunk c -> rank c %% %d), double-buffered batches of %d chunks, "
                                   "sampling + per-vertex normals + extraction" % (args.kind, n, n_chunks, c, world, st.batch),
                       "grid": n, "chunk": c, "chunks_per_gpu": n_chunks, "kind": args.kind, "seed": 1337, "batch_chunks": st.batch},
            "mtris_per_s": round(total_tris / step_s / 1e6, 1),
            "triangles_total": int(total_tris),
            "samples_GB_rank0": round(samples * 4 / 1e9, 2),
            # the dominant kernel's own bound: the sampler is bound by vector-ALU issue (its figure is the instruction model of
            # `sampler_valu`, in lane-instructions per second against the plain-FP32 issue peak), the extract stages by HBM
            "roofline": ({"bound": "valu", "kernel": dom, "achieved": round(lane_ops / (sum(fill_ms) * 1e-3) / 1e12, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 3),
                          "unit": "T lane-instructions/s", "frac": round(lane_ops / (sum(fill_ms) * 1e-3) / VALU_PEAK_LANE_OPS, 4), "traffic": None,
                          "traffic_source": None, "hbm_GBps_of_its_stores": round(ach, 1), "avg_ms": round(avg_ms, 4), "launches_per_step": nb}
                         if dom == "density_column_kernel" else
                         {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                          "algorithmic_bytes": alg_bytes, "avg_ms": round(avg_ms, 4), "launches_per_step": nb}),
            "sampler_valu": {"lane_ops_per_step": lane_ops, "achieved_lane_ops_per_s": round(lane_ops / (sum(fill_ms) * 1e-3), 1),
                             "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS, "frac": round(lane_ops / (sum(fill_ms) * 1e-3) / VALU_PEAK_LANE_OPS, 4),
                             "fp32_tflops": round(sampler_flops / (sum(fill_ms) * 1e-3) / 1e12, 2), "fp32_vector_peak_tflops": 157.3},
            "kernels_ms_per_step_serialised": {k: round(v, 3) for k, v in kern.items()},
            "serialised_step_ms": round(serial_s * 1e3, 3),
            "overlap_gain": round(serial_s / step_s, 3),
            "cpu_baseline": (cpu_baseline_stream(n, c, args.kind, origins_rank0, args.cpu_threads, torch.cuda.device_count())
                             if (world == 1 and not args.no_cpu_baseline) else None),
        }
        emit_line(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


_REAL_STDOUT = None


def emit_line(line):
    """The ONE line of the contract goes to the process's real stdout; see main()."""
    data = (line + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD `python -m torch.distributed.run`
    (exactly the command the task statement gives) before this process has touched the GPU or imported torch, hand its stdout
    through (rank 0's JSON line) and exit with its code.  A process that initialised the GPU is never replaced (no exec), and a
    line with n_gpus = 1 is never printed for a run that asked for N."""
    import signal
    import subprocess
    env = dict(os.environ)
    env["VTMC_BENCH_SELF_LAUNCHED"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    # --standalone: torchrun picks a free rendezvous port itself (no bind-then-close window another process could take)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    sys.stderr.flush()
    # the launcher and its ranks in a process group of their own; a SIGTERM / SIGINT to this process is forwarded to the whole group, so a
    # killed bench never leaves ranks behind holding the GPUs
    child = subprocess.Popen(cmd, env=env, start_new_session=True)

    def forward(signum, _frame):
        try:
            os.killpg(child.pid, signum)
        except ProcessLookupError:
            pass

    old = {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT)}
    try:
        while True:
            try:
                return child.wait()
            except KeyboardInterrupt:   # the handler above has already forwarded it
                continue
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass


def main():
    global _REAL_STDOUT
    args = parse()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("VTMC_BENCH_SELF_LAUNCHED") != "1":
        sys.exit(self_launch(args))
    # Libraries write to stdout on their own (RCCL prints a five-line version banner whenever a communicator is created): everything
    # but the JSON line is sent to stderr by pointing fd 1 there; the line itself goes to a duplicate of the original fd 1.
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if args.config == "stream2048":
        run_stream(args, torch, dist)
    else:
        run_grid(args, torch, dist)


if __name__ == "__main__":
    main()
