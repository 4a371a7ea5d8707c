#!/usr/bin/env python3
"""bench.py -- headline benchmark of the marching-cubes extraction path on MI355X.

--config grid1024 (default; BASELINE.json configs[2] at N = 1, configs[3] at N > 1 -- the one `metric`
is quoted on): a 1024^3-cell perlin3d grid held as 8^3 chunks of 128^3 cells (130^3 samples each,
4.50 GB), resident in HBM.  One "step" = one pass of the hot path over the rank's chunks: classify +
count -> prefix scan / compaction -> fused normals + triangle emit, ending when the host knows T and
(N > 1) every chunk's global offsets.  N > 1 is STRONG scaling by default, as configs[3] states it:
the world stays 1024^3 and the step ends with the path's one collective, the RCCL all-gather of
per-chunk {vertices, triangles} over xGMI (vtmc_allgather_volume_counts; --scaling weak keeps 512
chunks per rank on a 1024 x 1024 x 1024*N world instead).  The chunks start as c -> rank c % N; after
the first warm-up step every rank holds every chunk's counts (that all-gather) and the chunks are cut
again so that the ranks' triangle totals are even (--assign balanced, the default: c % 8 leaves the
heaviest rank 3 % above the mean, and the step ends when the heaviest rank does).  Steps are
independent passes over the same resident input and run four deep by default (--pipeline 4): four
contexts take turns, each on its own-queue stream, and step k + 3 is queued before the host takes step
k's T and offsets, so the device goes from step to step without waiting for the host -- the way a host
that extracts frame after frame would drive the library; every step still delivers its T, gather and
offsets.  `value` is that throughput; the latency of an isolated step is reported next to it
(`step_latency_ms`, = --pipeline 1).  Each context's output buffer is chosen among eight allocations in its first
warm-up step (--place-outputs 8, at most 16: the library's placement trials; the identical emit kernel runs 0.86 ... 1.00 ms
by which allocation it writes, profiles/r06/placement_probe.txt; every candidate's time is in the line).

--config stream2048 (BASELINE.json configs[4]): a 2048^3-cell fbm8 world (36 GB of samples) streamed
as double-buffered batches of 128^3 chunks (two density buffers / contexts, each on its own-queue
stream, the host a batch ahead; the sampler leaves the samples' sign bits and the classify stage reads
those); one step = one pass over the rank's 4096 / N chunks, sampling included.  The grid1024 line at
N = 1 carries a short form of it (`stream2048`: one warm and two timed passes, no CPU leg).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
algorithmic bytes / HIP-event time on the kernels' own stream), `cpu_baseline` (the CPU oracle timed
on this box's host cores on a bounded sample of the same device-generated field), `box` (what this
box's memory delivers for plain streams: tools/calib/mix2, a fresh process after the timed regions)
and, at N = 1, `predicted_scaling` (every rank of an N = 2 / 4 / 8 run rehearsed on this one GPU).

PROCESSES (round 6).  The process the driver (or torchrun) starts is a SUPERVISOR that never imports
torch and never touches a GPU; it starts the measuring WORKER as a child and reads its stage reports
and its JSON line from a pipe.  Every worker carries a watchdog (a thread: when a stage exceeds its
bound it prints the stage and every thread's stack and leaves with os._exit(3)).  If the worker fails
before its timed region is complete, the supervisor starts ONE fresh worker with the most conservative
configuration (--pipeline 2 --streams 1, the collective behind the emit kernel on the step's stream)
and labels the line "fallback": true; at N > 1 the ranks of the second attempt meet through a
prefix of the launcher's store.  A process that has initialised a GPU is never re-executed.
--direct runs the worker in this process (profilers; no fallback).
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
import threading  # noqa: E402
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

# the most conservative way to drive the same steps: two contexts taking turns on ONE ordinary stream, every collective behind its emit kernel
FALLBACK_ARGS = ["--pipeline", "2", "--streams", "1", "--gather-stream", "main", "--assign", "modulo", "--place-outputs", "0"]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="grid1024", choices=["grid1024", "stream2048"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: the same 1024^3 world sharded (BASELINE configs[3]) or 512 chunks per rank")
    ap.add_argument("--assign", default="balanced", choices=["balanced", "modulo"],
                    help="N > 1, strong scaling: modulo = chunk c -> rank c %% N throughout; balanced (default) = that for the first warm-up step, then the chunks are cut "
                         "again by the triangle counts every rank holds after that step's all-gather, so that the ranks' totals are even "
                         "(volumetricterrain_amd.sharding.balanced_assignment; every rank derives the same partition, nothing else is exchanged)")
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
                         "and the host's read-back in the loop) and the whole 1024^3 step 1-2 %% (profiles/r05/depth_probe.txt).  "
                         "At N > 1 all contexts issue their all-gather through ONE communicator (vtmc_comm_share).  1: every step ends with its host "
                         "wait (the latency of an isolated step, also reported as step_latency_ms)")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="grid1024, --pipeline > 1: the MODE of the contexts' streams.  2 (default): a stream EACH (the context's own-queue stream, so --pipeline 4 "
                         "means four streams) -- step k + 1's classify kernel starts on the CUs step k's emit kernel leaves as it drains (the emit kernel is bound by issued "
                         "instructions, the classify kernel by memory: profiles/r05/rank_overlap_probe.txt: -4 %% of a step at N = 1, -20 %% of a "
                         "rank's step of an 8-rank run); 1: one ordinary stream for all of them (rounds 2-4)")
    ap.add_argument("--gather-stream", default="side", choices=["side", "main"],
                    help="N > 1: the stream the all-gather and the copy of its result are queued on.  side (default, round 6): ONE ordinary stream for every "
                         "collective of the rank, each ordered behind its extract's emit launch by an event (the library's), the pinned read-back behind it on the "
                         "same stream -- RCCL sees one stream, nothing pinned rides an own-queue stream, and the steps' own streams never wait for a collective.  "
                         "main: behind the emit kernel on the step's own stream; with a stream per context the collectives of the rank's ONE communicator then "
                         "alternate between streams and the library chains them by events (csrc/comm.hip); the read-back still travels on an ordinary stream")
    ap.add_argument("--gather-beside", action="store_true",
                    help="N > 1, opt-in: the all-gather on the context's second stream beside the emit kernel (tuning key gather_beside)")
    ap.add_argument("--place-outputs", type=int, default=8,
                    help="grid1024: the library's output placement trials (tuning key place_outputs): when a context has just allocated its output buffers -- its first warm-up "
                         "step -- the emit stage is run into this many candidate allocations and the fastest is kept.  The emit kernel's time is a property of the pair "
                         "(input field's allocation, output allocation): 0.86 ... 1.00 ms for the identical kernel (profiles/r06/placement_probe.txt).  0 / 1: take what "
                         "hipMalloc gives (rounds 1-5).  The line reports every candidate's time (output_placement)")
    ap.add_argument("--no-dense", action="store_true", help="A/B: force the per-block classify kernel")
    ap.add_argument("--no-indexed", action="store_true", help="skip the extra indexed-output steps at N = 1")
    ap.add_argument("--no-rehearsal", action="store_true", help="N = 1: skip the rehearsal of every rank of an N = 2 / 4 / 8 run (predicted_scaling)")
    ap.add_argument("--no-stream-record", action="store_true", help="N = 1: skip the short stream2048 run behind the grid (the `stream2048` sub-record)")
    ap.add_argument("--stream-record-cells", type=int, default=2048, help="cells per axis of the `stream2048` sub-record (reduced sizes for tests)")
    ap.add_argument("--no-terrain-record", action="store_true", help="N = 1: skip the `terrain` sub-record (world build + interactive edits on a grid resident in HBM: SURVEY 8f rank 1)")
    ap.add_argument("--no-box", action="store_true", help="skip the memory calibration of this box (tools/calib/mix2 box, a fresh process after the timed regions)")
    ap.add_argument("--no-fallback", action="store_true", help="a failed worker is not followed by a second, conservative attempt")
    ap.add_argument("--direct", action="store_true", help="no supervisor: measure in this process (for profilers; no fallback, the watchdog still applies)")
    args = ap.parse_args(argv)
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
# worker side of the supervisor's pipe, and the watchdog
# ------------------------------------------------------------------------------------------------
_REPORT_FD = None       # the supervisor's pipe (VTMC_BENCH_REPORT_FD); None: --direct, the line goes to stdout
_REAL_STDOUT = None     # --direct: a duplicate of the original fd 1 (fd 1 itself is pointed at stderr, see main())
_REPORT_LOCK = threading.Lock()


def report(kind, **kw):
    """One JSON message to the supervisor: {"k": "stage" | "line" | "done" | "watchdog", ...}.  Without a supervisor only the line matters."""
    if _REPORT_FD is None:
        return
    msg = dict(kw)
    msg["k"] = kind
    data = (json.dumps(msg) + "\n").encode()
    with _REPORT_LOCK:
        try:
            os.write(_REPORT_FD, data)
        except OSError:
            pass


def emit_partial(obj, what):
    """A first / intermediate form of the line, for the supervisor only (it prints the LAST line it received): if anything behind this point
    fails, what was measured so far is not lost.  --direct prints nothing until the full line."""
    if _REPORT_FD is not None:
        emit_line(dict(obj, partial=what))


def emit_line(obj):
    """The ONE line of the contract.  Under a supervisor it travels through the pipe (the supervisor prints the LAST one it received, once, when the
    worker has ended -- so the worker may send a first, short form right after the timed region and the full one later); --direct: real stdout."""
    line = json.dumps(obj)
    if _REPORT_FD is not None:
        report("line", line=line)
    elif _REAL_STDOUT is not None:
        os.write(_REAL_STDOUT, (line + "\n").encode())
    else:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()


def test_hang(wd, stage):
    """Test hook (tests/test_bench_modes.py): VTMC_BENCH_TEST_HANG="<stage>:<seconds>" makes the FIRST attempt's worker stop answering in that stage
    with the stage's bound cut to <seconds> -- a real worker, on the GPU, whose watchdog fires and whose supervisor starts the real fallback."""
    spec = os.environ.get("VTMC_BENCH_TEST_HANG")
    if not spec or os.environ.get("VTMC_BENCH_FALLBACK") == "1":
        return
    name, _, secs = spec.partition(":")
    if name == stage:
        wd.stage(stage, float(secs or 3) / wd.scale)
        time.sleep(1e6)


class Refusal(Exception):
    """The run cannot happen at all here (no GPU, a launcher's world of another size): exit code 4, and no second attempt."""


class Watchdog:
    """A thread that ends the process when a stage of the run takes longer than its bound: prints the stage reached and every thread's stack
    (faulthandler) to stderr, tells the supervisor, os._exit(3).  A collective that never completes, a kernel whose waves never finish or a
    rendezvous nobody joins then costs its bound, not the driver's whole time limit -- and the supervisor, which never touched a GPU, can still
    start the conservative second attempt.  Bounds are generous multiples of what the stages take (a first `import torch` on a fresh box: two
    minutes; everything after it: seconds) and scale with VTMC_BENCH_WATCHDOG_SCALE."""

    def __init__(self, tag):
        self.tag = tag
        self.scale = float(os.environ.get("VTMC_BENCH_WATCHDOG_SCALE", "1"))
        self.name = "start"
        self.deadline = None
        self.t0 = time.monotonic()
        self.lock = threading.Lock()
        self.thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        self.thread.start()

    def stage(self, name, budget_s):
        with self.lock:
            self.name = name
            self.budget = budget_s * self.scale
            self.deadline = time.monotonic() + self.budget
        report("stage", name=name, budget_s=round(self.budget, 2), t=round(time.monotonic() - self.t0, 2))

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        import faulthandler
        while True:
            time.sleep(0.2)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                name, budget = self.name, getattr(self, "budget", 0.0)
            if late:
                try:
                    sys.stderr.write("bench.py[%s]: WATCHDOG: stage '%s' exceeded its bound of %.0f s -- leaving with exit code 3; stacks of all threads follow\n" % (self.tag, name, budget))
                    sys.stderr.flush()
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    sys.stderr.flush()
                except Exception:   # noqa: BLE001
                    pass
                report("watchdog", name=name, budget_s=budget)
                os._exit(3)


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


def pick_representative_chunks(counts, n_sample):
    """Indices of `n_sample` chunks whose triangles per cell are as close to the whole list's as a bounded sample gets: the world of
    config 5 is a terrain -- its surface sits in a fifth of the chunks (cy = 7, 8 of 16) and the rest is empty --, so an evenly spread
    sample can miss the surface altogether (round 5's did: triangles_in_sample = 0, the "CPU leg" never emitted a triangle).
    Stratified by the chunks' own triangle counts (the GPU pass that was just timed delivered them): the sample takes
    round(n_sample x share of non-empty chunks) non-empty chunks (at least one when there is a surface at all), spread over the
    quantiles of the non-empty chunks' counts, then the LAST of them is replaced by the non-empty chunk that brings the sample's mean
    closest to the list's mean; the rest of the sample are empty chunks spread evenly over the list."""
    counts = np.asarray(counts, np.int64)
    n = len(counts)
    n_sample = max(1, min(n_sample, n))
    full = np.flatnonzero(counts > 0)
    empty = np.flatnonzero(counts == 0)
    k = 0
    if len(full):
        k = int(round(n_sample * len(full) / n))
        k = max(1, min(k, len(full), n_sample))
    k = max(k, n_sample - len(empty))
    pick = []
    if k:
        by_count = full[np.argsort(counts[full], kind="stable")]
        pick = [int(by_count[min(len(by_count) - 1, int((q + 0.5) * len(by_count) / k))]) for q in range(k)]
        pick = list(dict.fromkeys(pick))
        target = counts.mean() * n_sample                      # what the sample's triangle sum should be
        rest = sum(int(counts[i]) for i in pick[:-1])
        cand = [int(i) for i in by_count if int(i) not in pick[:-1]]
        pick[-1] = min(cand, key=lambda i: (abs(rest + int(counts[i]) - target), i))
    m = n_sample - len(pick)
    if m > 0 and len(empty):
        pick += [int(empty[min(len(empty) - 1, int((q + 0.5) * len(empty) / m))]) for q in range(m)]
    return sorted(dict.fromkeys(pick))


def cpu_baseline_stream(n, chunk, kind, origins, chunk_tris, want_threads, n_gpus_on_box, n_sample=8):
    """The CPU leg of the streaming config (BASELINE configs[4]): the oracle's per-sample sampler + its extractor, chunk by chunk, on a
    bounded sample of the same world -- `n_sample` chunks of this rank's list chosen by pick_representative_chunks() from the per-chunk
    triangle counts of the pass that was just timed, so that the sample's triangles per cell match the world's (the ratio is printed) --
    both stages on the stated thread count.  Repetitions of >= 0.5 s as cpu_baseline(); a one-thread leg on one chunk of the sample."""
    import ctypes
    import oracle
    L = oracle.lib()
    share = cpu_share()
    cap = 16 * max(1, n_gpus_on_box)
    threads = want_threads or min(x for x in (share["physical_cores"], share["cgroup_quota_cores"] or 1 << 30, cap, oracle.max_threads()))
    dim = chunk + 2
    prm = oracle.density_params(kind, n)
    idx = pick_representative_chunks(chunk_tris, n_sample)
    pick = [origins[i] for i in idx]
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
    heavy = pick[int(np.argmax(totals))]
    one_runs = [repetition(1, [heavy], 0.0)]   # the sample's heaviest chunk on one thread: seconds
    rates = [r for r, _ in all_runs]
    world_tpc = float(np.sum(chunk_tris)) / (len(chunk_tris) * cells)
    sample_tpc = float(sum(totals)) / (len(pick) * cells)
    return {
        "value": round(statistics.median(rates), 2),
        "unit": "Mvoxels/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d chunks of %d^3 cells out of the rank's %d (%s, sampled AND extracted by the oracle: oracle/density_ref.c + "
                  "oracle/mc_oracle.c), stratified by the chunks' triangle counts so that the sample's triangles per cell match the world's, "
                  "OpenMP pinned close/cores, 5 repetitions of >= 0.5 s, median" % (len(pick), chunk, len(origins), kind),
        "sample_chunks": [int(i) for i in idx],
        "repetitions_mvoxels_per_s": [round(r, 2) for r in rates],
        "repetition_seconds": [round(s, 3) for _, s in all_runs],
        "min_mvoxels_per_s": round(min(rates), 2),
        "max_mvoxels_per_s": round(max(rates), 2),
        "triangles_in_sample": int(sum(totals)),
        "triangles_per_cell": {"sample": round(sample_tpc, 6), "world": round(world_tpc, 6),
                               "sample_over_world": round(sample_tpc / world_tpc, 4) if world_tpc > 0 else None},
        "mtris_per_s": round(statistics.median(rates) * sample_tpc, 2),
        "single_core_mvoxels_per_s": round(statistics.median(r for r, _ in one_runs), 2),
        "single_core_sample": "the sample's heaviest chunk (%d triangles)" % max(totals),
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


def init_distributed(args, torch, dist, need_gpu=True):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # main() starts the ranks itself when there is no launcher; a launcher with another world size is a mistake
        raise Refusal("bench.py: --gpus %d inside a launcher's world of %d ranks (WORLD_SIZE): refusing to print a line for another N" % (args.gpus, world))
    if need_gpu and not torch.cuda.is_available():
        raise Refusal("bench.py needs an MI355X: the extraction path has no CPU fallback")
    # rehearsal hook for a one-GPU box: every rank on device 0, gloo instead of RCCL (which refuses
    # two ranks on one device); the driver's multi-GPU runs use neither variable
    if os.environ.get("VTMC_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("VTMC_BENCH_BACKEND", "nccl" if need_gpu else "gloo")
    if need_gpu:
        torch.cuda.set_device(local)
    if world > 1:
        import datetime
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}   # RCCL over xGMI
        if os.environ.get("VTMC_BENCH_FALLBACK") == "1":
            # the second attempt's ranks meet through a PREFIX of the rendezvous store: under torchrun the store lives in the launcher's agent
            # (TORCHELASTIC_USE_AGENT_STORE) and has outlived the first attempt's workers, whose keys must not be seen again; without an
            # agent, rank 0 serves a new store on the same port (the first attempt's server died with its worker)
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
            store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world, is_master=(rank == 0 and not agent),
                                  timeout=datetime.timedelta(seconds=600), wait_for_workers=False)
            dist.init_process_group(backend, store=dist.PrefixStore("vtmc_fallback/", store), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=600), **kw)
        else:
            dist.init_process_group(backend, **kw)
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
class GridPipeline:
    """`depth` contexts of one device taking turns over ONE resident batch of chunks: step i is queued (classify -> scan -> emit
    [-> all-gather -> pinned copy of the gathered pairs]) before the host takes step i - depth + 1.  The timed region, the
    one-stream region behind it and the rank rehearsal are all this object, bound to different chunk sets.

    Streams: own_queue_streams = a stream per context, each on a hardware queue of its own (vtmc_context_stream); otherwise the first
    context's ordinary stream for everything.  Never torch's CURRENT stream: what torch allocates while a stream is current belongs to
    that stream in its caching allocator.  The collective: gather_stream "side" = ONE ordinary torch stream for every collective and
    its pinned read-back (the library orders each behind its extract's emit launch); "main" = behind the emit kernel on the step's own
    stream, the pinned read-back on an ordinary stream behind the collective's event (pinned copies never ride an own-queue stream)."""

    def __init__(self, torch, vt, device, depth, own_queue_streams, gather_stream="side", gather_beside=False, tuning=None, no_dense=False):
        self.torch, self.depth = torch, depth
        self.exs = [vt.Extractor(device) for _ in range(depth)]
        if gather_beside:
            for e in self.exs:
                e.set_tuning(gather_beside=1)
        if tuning:
            for e in self.exs:
                e.set_tuning(**tuning)
        self.own_queue = bool(own_queue_streams)
        self.streams = [torch.cuda.ExternalStream(self.exs[i].stream_handle(own_queue=self.own_queue)) for i in range(depth if self.own_queue else 1)]
        self.gather_stream, self.gather_beside = gather_stream, gather_beside
        self.flags = 2 if no_dense else 0
        self.native = False
        self.side = self.copy_stream = None
        self.slots = []
        self.exchange = False
        self.sample_stages = False
        self.reset_stats()

    # -- the communicator: ONE per rank whatever the depth; the first context owns it, the others borrow it (vtmc_comm_share)
    def set_comm(self, native):
        self.native = native
        if native:
            for e in self.exs[1:]:
                e.comm_share(self.exs[0])

    def world_of_one_comm(self):
        """Rehearsal on one GPU: the N > 1 host path (collective, pinned copy, offsets) through a communicator of one rank."""
        self.exs[0].comm_init_rank(self.exs[0].comm_unique_id(), 0, 1)
        self.set_comm(True)

    def reset_stats(self):
        self.stage_acc = {"classify": 0.0, "scan": 0.0, "emit": 0.0, "total": 0.0}
        self.stage_steps = 0     # steps whose three kernels were timed one by one
        self.gather_ms = []

    def bind(self, d_ptr, n_chunks, chunk, gather_world, per_rank, perm, exchange, sample_stages=False):
        """The chunk set the steps run over: `n_chunks` volumes of (chunk + 2)^3 samples at d_ptr; with `exchange`, every step ends with the
        all-gather into a (gather_world x per_rank x 2) array and perm[c] = chunk c's slot in it (sharding.slot_permutation)."""
        torch = self.torch
        self.d_ptr, self.n_chunks, self.c, self.dim = d_ptr, n_chunks, chunk, chunk + 2
        self.per_rank, self.perm, self.exchange, self.sample_stages = per_rank, perm, exchange, sample_stages
        n_total = len(perm) if perm is not None else 0

        class Slot:   # what one step in flight owns besides its context
            pass

        self.slots = []
        for i, e in enumerate(self.exs):
            sl = Slot()
            sl.ex = e
            sl.stream = self.streams[i % len(self.streams)]
            sl.s_ptr = sl.stream.cuda_stream
            sl.gathered = torch.zeros((gather_world, per_rank, 2), dtype=torch.int32, device="cuda")
            sl.counts_dev = torch.zeros((per_rank, 2), dtype=torch.int32, device="cuda")   # the torch collective's send buffer
            sl.gathered_host = torch.zeros((gather_world, per_rank, 2), dtype=torch.int32).pin_memory()
            sl.host_rows = sl.gathered_host.numpy().reshape(-1, 2)   # a view of the pinned words
            sl.offs = np.zeros((n_total + 1, 2), np.int64)
            sl.ev0, sl.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sl.gathered_ev, sl.copied = torch.cuda.Event(), torch.cuda.Event()
            sl.timed_gather, sl.timed_stages = False, True
            self.slots.append(sl)
        if exchange and self.native and self.gather_stream == "side" and not self.gather_beside:
            self.side = self.side or torch.cuda.Stream()
        else:
            self.side = None
        if exchange and self.side is None:
            self.copy_stream = self.copy_stream or torch.cuda.Stream()
        self.set_one_stream(False)

    def set_one_stream(self, one):
        for i, sl in enumerate(self.slots):
            st = self.streams[0] if one else self.streams[i % len(self.streams)]
            sl.stream, sl.s_ptr = st, st.cuda_stream

    def queue(self, sl, timed_gather=False, timed_stages=True, backend="nccl", dist=None, sharding=None):
        """classify -> scan -> emit [-> all-gather -> copy of the gathered pairs into pinned words]; nothing waits."""
        torch, c, dim = self.torch, self.c, self.dim
        if self.sample_stages:
            sl.ex.set_tuning(stage_events=1 if timed_stages else 0)
        sl.timed_stages = timed_stages
        sl.ex.extract_volumes_device_async(self.d_ptr, (c, c, c), (1, dim, dim * dim), self.n_chunks, dim ** 3, sl.s_ptr, self.flags)
        if not self.exchange:
            return
        sl.timed_gather = timed_gather
        if timed_gather:
            sl.ev0.record(sl.stream)
        if self.native and self.side is not None:
            # the path's one collective, behind the C ABI, on the rank's ONE collective stream: the library orders it behind this extract's
            # emit launch (an event), the copy of the gathered pairs follows it there, and the step's stream goes straight on to its next
            # step's classify kernel -- nothing on it waits for the collective, only the host does (`copied`)
            sl.ex.allgather_volume_counts(sl.gathered.data_ptr(), self.per_rank, self.side.cuda_stream)
            if timed_gather:
                sl.ev1.record(self.side)
            with torch.cuda.stream(self.side):
                sl.gathered_host.copy_(sl.gathered, non_blocking=True)
            sl.copied.record(self.side)
            return
        with torch.cuda.stream(sl.stream):
            if self.native:     # --gather-stream main: behind the emit kernel on the extract's own stream (or beside it with --gather-beside)
                sl.ex.allgather_volume_counts(sl.gathered.data_ptr(), self.per_rank, sl.s_ptr)
            else:
                sl.ex.copy_volume_counts_device(sl.counts_dev.data_ptr(), self.per_rank, sl.s_ptr)
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
        # the read-back of the gathered pairs: an ORDINARY stream carries the pinned copy, behind the collective's event (round 5 queued it on
        # the step's own-queue stream -- the one kind of stream a plain C++ host's pinned copies once hung on at exit)
        sl.gathered_ev.record(sl.stream)
        self.copy_stream.wait_event(sl.gathered_ev)
        with torch.cuda.stream(self.copy_stream):
            sl.gathered_host.copy_(sl.gathered, non_blocking=True)
        sl.copied.record(self.copy_stream)

    def complete(self, sl, accumulate=False):
        """The one host wait of a step: its gathered pairs are in pinned memory (or, without an exchange, its extract is done)."""
        if self.exchange:
            sl.copied.synchronize()
        T = sl.ex.extract_finish()
        offs = None
        if self.exchange:   # every rank's local exclusive scan over the chunks in global order
            np.cumsum(sl.host_rows[self.perm], axis=0, dtype=np.int64, out=sl.offs[1:])
            offs = sl.offs
        if accumulate and sl.timed_stages:
            ms = sl.ex.last_stage_ms()
            for k in self.stage_acc:
                self.stage_acc[k] += ms[k]
            self.stage_steps += 1
            if self.exchange and sl.timed_gather:
                self.gather_ms.append(sl.ev0.elapsed_time(sl.ev1))
        return T, offs

    def run_steps(self, k_steps, accumulate, **kw):
        """k_steps steps, `depth` in flight: step i + depth - 1 is queued before the host takes step i."""
        T = offs = None
        depth, slots = self.depth, self.slots
        for i in range(k_steps):
            self.queue(slots[i % depth], timed_gather=accumulate and self.exchange and i % 8 == 0,   # the collective's own events on every eighth step
                       timed_stages=(not self.sample_stages) or i % 8 == 0, **kw)
            if i >= depth - 1:
                T, offs = self.complete(slots[(i - depth + 1) % depth], accumulate)
        for i in range(max(k_steps - depth + 1, 0), k_steps):
            T, offs = self.complete(slots[i % depth], accumulate)
        return T, offs

    def close(self):
        """Drain, drop every torch object that was used on the contexts' streams, close the contexts (the borrowers of the communicator first).
        Since round 6 the order no longer matters for safety -- the library parks its streams instead of destroying them, so an event or a
        pinned tensor that outlives its context is harmless -- but a bench that is about to start a child process leaves nothing behind."""
        torch = self.torch
        torch.cuda.synchronize()
        self.slots = []
        import gc
        gc.collect()
        for e in reversed(self.exs):
            e.close()
        self.exs = []


def per_chunk_counts(ex, n_chunks, bpv):
    """(triangles per chunk, non-empty blocks per chunk) of the extract `ex` has just finished, from its device-side results."""
    _, off_ptr, vc_ptr = ex.device_results()
    tris = ex.copy_u32(vc_ptr, 2 * n_chunks).reshape(-1, 2)[:, 1].astype(np.int64)
    boffs = ex.copy_u32(off_ptr, n_chunks * bpv + 1).astype(np.int64)
    active = (np.diff(boffs) > 0).reshape(n_chunks, bpv).sum(axis=1).astype(np.int64)
    return tris, active


def rehearse_ranks(torch, pipe, d_field, n_chunks_world, chunk, chunk_tris, chunk_active, worlds=(2, 4, 8), steps=48, assign="balanced", log=None):
    """EVERY rank of an N-rank run, one after the other, on this one GPU with the configuration `pipe` was built with (the shipped N > 1 default:
    four steps in flight on own-queue streams, the world-of-one collective on the rank's collective stream, the pinned read-back, the host's
    offsets): rank r's chunks are gathered into one buffer and stepped `steps` times.  predicted strong scaling = the whole world's step in
    the same process and configuration / the SLOWEST rank's step -- a rehearsal, not a scaling measurement: a real run adds the all-gather's
    cross-GPU latency (one 512-byte message per rank, off the steps' streams) and a real box's neighbours."""
    from volumetricterrain_amd import sharding
    dim = chunk + 2
    field2d = d_field.view(n_chunks_world, dim ** 3)
    ident = np.arange(n_chunks_world, dtype=np.intp)
    total_T = int(np.sum(chunk_tris))

    def step_ms(d_ptr, ids):
        n = len(ids)
        pipe.bind(d_ptr, n, chunk, 1, n, np.arange(n, dtype=np.intp), True, sample_stages=True)   # stage events on every eighth step only, as a real rank
        T, offs = pipe.run_steps(2 * pipe.depth, False)
        want = int(sum(int(chunk_tris[c]) for c in ids))
        assert T == want and int(offs[-1, 1]) == want, "rehearsal: rank's triangle total %d, its chunks' counts say %d" % (T, want)
        best = None
        for _ in range(3):   # the best of three regions: one rank in ten draws a region 5 % slow
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run_steps(steps, False)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            best = ms if best is None else min(best, ms)
        return best

    world_ms = step_ms(d_field.data_ptr(), ident)
    out = {"world_step_ms": round(world_ms, 4), "assignment": assign, "steps_per_rank": steps,
           "configuration": "%d steps in flight, %s, collective %s (world-of-one communicator), pinned read-back, host offsets"
                            % (pipe.depth, "a hardware queue per context" if pipe.own_queue else "one stream",
                               "on the rank's one collective stream" if pipe.side is not None else "behind the emit kernel on the step's stream"),
           "ranks": {}}
    scratch = torch.empty(((n_chunks_world + 1) // 2) * dim ** 3, dtype=torch.float32, device="cuda")
    for w in worlds:
        if n_chunks_world < w:
            continue
        modulo = sharding.modulo_assignment(n_chunks_world, w)
        parts = sharding.balanced_assignment(chunk_tris, w) if assign == "balanced" else modulo
        ms = []
        for r in range(w):
            ids = parts[r]
            idx = torch.tensor(ids, dtype=torch.int64, device="cuda")
            buf = scratch[:len(ids) * dim ** 3].view(len(ids), dim ** 3)
            torch.index_select(field2d, 0, idx, out=buf)
            torch.cuda.synchronize()
            ms.append(step_ms(buf.data_ptr(), ids))
            if log:
                log("rehearsal N = %d rank %d: %d chunks, %d triangles, %.4f ms per step" % (w, r, len(ids), sum(int(chunk_tris[c]) for c in ids), ms[-1]))
        out["ranks"][str(w)] = {
            "step_ms": [round(x, 4) for x in ms],
            "slowest_ms": round(max(ms), 4), "mean_ms": round(sum(ms) / len(ms), 4),
            "slowest_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4),
            "predicted_scaling": round(world_ms / max(ms), 3),
            "triangles_max_over_mean": {"modulo": round(sharding.imbalance(chunk_tris, modulo), 4), "balanced": round(sharding.imbalance(chunk_tris, sharding.balanced_assignment(chunk_tris, w)), 4)},
            "active_blocks_max_over_mean": {"modulo": round(sharding.imbalance(chunk_active, modulo), 4), "balanced": round(sharding.imbalance(chunk_active, sharding.balanced_assignment(chunk_tris, w)), 4)},
        }
    del scratch
    assert total_T == int(np.sum(chunk_tris))
    return out


def measure_stream(torch, dist, args, rank, world, local, backend, wd=None, passes=None, warm=None):
    """One ChunkStream over the rank's share of the world: `warm` untimed passes, then `passes` timed ones between barriers; then a serialised
    diagnostic pass (fill waits, then extract) for per-kernel device times.  Returns a dict of raw numbers."""
    from volumetricterrain_amd.streaming import ChunkStream
    n, c = args.n, args.chunk
    dim = c + 2
    passes = args.steps if passes is None else passes
    warm = max(args.warmup, 1) if warm is None else warm
    with ChunkStream(n, c, args.batch, args.kind, n, rank=rank, world_size=world, device=local, sampler_wgs_per_cu=args.sampler_wgs, two_queues=not args.stream_one_queue) as st:
        n_chunks = len(st.origins)
        origins = [tuple(int(v) for v in o) for o in st.origins]
        for _ in range(warm):   # buffers grow to their steady size
            st.run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pass_ms = []
        t0 = time.perf_counter()
        for _ in range(passes):
            t1 = time.perf_counter()
            total, counts = st.run()
            pass_ms.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
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
        return {"n_chunks": n_chunks, "origins": origins, "elapsed": elapsed, "passes": passes, "pass_ms": pass_ms, "total": total, "counts": counts,
                "fill_ms": fill_ms, "stage": stage, "serial_s": serial_s, "batch": st.batch, "bpv": st.bpv, "dim": st.dim,
                "octaves": st.params.octaves, "frequency": st.params.frequency, "lacunarity": st.params.lacunarity}


def stream_sub_record(torch, args, local):
    """BASELINE configs[4] on this one GPU, short: one warm and two timed passes over the 2048^3 fbm8 world, no CPU leg -- so that the driver's
    own run times config 5's single-GPU number too (the full line: --config stream2048)."""
    cells = args.stream_record_cells
    sa = argparse.Namespace(n=cells, chunk=128, batch=256, kind="fbm8", steps=2, warmup=1, sampler_wgs=None, stream_one_queue=False)
    m = measure_stream(torch, None, sa, 0, 1, local, "nccl", passes=2, warm=1)
    step_s = m["elapsed"] / m["passes"]
    return {"config": "fbm8 %d^3 cells streamed as %d chunks of 128^3 in double-buffered batches of %d, sampling + extraction, 1 GPU" % (cells, m["n_chunks"], m["batch"]),
            "ms_per_pass": round(step_s * 1e3, 3), "passes_ms": [round(x, 3) for x in m["pass_ms"]], "warm_passes": 1,
            "mvoxels_per_s": round(float(cells) ** 3 / step_s / 1e6, 1), "mtris_per_s": round(m["total"] / step_s / 1e6, 1), "triangles_total": int(m["total"]),
            "kernels_ms_per_pass_serialised": {"density_column_kernel": round(sum(m["fill_ms"]), 3), "classify_dense_kernel": round(m["stage"]["classify"], 3),
                                               "scan": round(m["stage"]["scan"], 3), "emit_kernel": round(m["stage"]["emit"], 3)}}


def terrain_sub_record(vt):
    """SURVEY 8f rank 1 in the driver's own run: the density grid resident in HBM, VoxelTerrain.Update on the device.  (a) the world build as
    TerrainEngine.Init does it (TerrainEngine.cs:87-99): 1024 x 256 x 1024 cells, one IslandModifier (512^2 heightmap) + 40 river cylinders, ONE
    vtmc_terrain_update (density write of every modifier + extraction of every block); (b) the interactive edit of SceneManager.cs:121-129 on
    the demo world (256 x 72 x 256): 200 sphere edits r = 10, one vtmc_terrain_update each (density write + dirty set + classify + scan +
    emit + T on the host).  Wall clock around the blocking calls; the CPU side of both is tools/world_build_bench.py / tools/edit_latency.py."""
    rng = np.random.default_rng(3)
    W, E, H = 1024, 256, 1024
    u = np.linspace(-1, 1, 512, dtype=np.float32)[:, None]
    v = np.linspace(-1, 1, 512, dtype=np.float32)[None, :]
    hm = (0.55 * np.exp(-2.5 * (u * u + v * v)) + 0.06 * np.sin(7 * u) * np.cos(5 * v) + 0.12).astype(np.float32)
    owners = [vt.IslandModifier(hm * E, float(W), float(H), float(E), True)]
    for _ in range(40):
        start = (float(rng.uniform(0.2, 0.8) * W), float(rng.uniform(0.25, 0.5) * E), float(rng.uniform(0.2, 0.8) * H))
        d = (float(rng.normal()), float(rng.normal() * 0.1), float(rng.normal()))
        owners.append(vt.CylinderModifier(start, d, float(rng.uniform(0.05, 0.15) * W), float(rng.uniform(1.5, 3.0)), False))
    mods = [m.to_struct() for m in owners]   # the structs borrow the heightmap array: `owners` stays alive
    with vt.Extractor(0) as ex:
        times = []
        for _ in range(4):
            ex.terrain_init(W, E, H, 1.0, (0.0, 0.0, 0.0), 5)
            t0 = time.perf_counter()
            nd, T = ex.terrain_update(mods)
            times.append(time.perf_counter() - t0)
        build = {"world": "%dx%dx%d cells, IslandModifier (512^2 heightmap) + 40 river cylinders, one Update" % (W, E, H),
                 "update_ms": round(min(times[1:]) * 1e3, 3), "dirty_blocks": int(nd), "triangles": int(T),
                 "msamples_per_s": round((W + 2) * (E + 2) * (H + 2) / min(times[1:]) / 1e6, 1)}
    erng = np.random.default_rng(1)
    with vt.Extractor(0) as ex:
        ex.terrain_init(256, 72, 256, 1.0, (0.0, 0.0, 0.0), 1)
        ex.terrain_update([vt.PlaneModifier(30.5, (-1, -1), (300, 300), True)])
        lat, tris = [], []
        for i in range(220):
            cpos = (float(erng.uniform(20, 236)), 30.0 + float(erng.uniform(-4, 4)), float(erng.uniform(20, 236)))
            m = vt.SphereModifier(cpos, 10.0, bool(i & 1))
            t0 = time.perf_counter()
            nd, T = ex.terrain_update([m])
            if i >= 20:
                lat.append((time.perf_counter() - t0) * 1e6)
                tris.append(T)
        edits = {"world": "256x72x256 cells, plane + 200 sphere edits r = 10 (alternating add / erode)", "edit_latency_us_median": round(statistics.median(lat), 1),
                 "edit_latency_us_p90": round(float(np.percentile(lat, 90)), 1), "triangles_median": int(statistics.median(tris))}
    return {"world_build": build, "edits": edits}


def run_grid(args, torch, dist, wd):
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding
    wd.stage("init", 300)
    rank, world, local, backend = init_distributed(args, torch, dist)
    wd.stage("setup", 180)
    n, c = args.n, args.chunk
    dim = c + 2
    strong = world > 1 and args.scaling == "strong"
    world_dims = (n, n, n) if (strong or world == 1) else (n, n, n * world)
    n_chunks_total = (world_dims[0] // c) * (world_dims[1] // c) * (world_dims[2] // c)
    per_rank = (n_chunks_total + world - 1) // world   # slots per rank in the gathered array (zero-padded)
    assignment = sharding.modulo_assignment(n_chunks_total, world)   # rank r holds chunks r, r + N, ...
    my_chunks = assignment[rank]
    n_chunks = len(my_chunks)
    bpv = (c // 8) ** 3
    depth = args.pipeline
    tuning = None
    if os.environ.get("VTMC_BENCH_TUNING"):   # A/B of kernel variants under the bench's sustained load, e.g. VTMC_BENCH_TUNING="emit_once=0"
        tuning = {k: int(v) for k, v in (item.split("=") for item in os.environ["VTMC_BENCH_TUNING"].split(","))}
    if args.place_outputs > 1 and not (tuning and "place_outputs" in tuning):
        tuning = dict(tuning or {}, place_outputs=min(args.place_outputs, 16))
    pipe = GridPipeline(torch, vt, local, depth, args.streams == 2, args.gather_stream, args.gather_beside, tuning, args.no_dense)
    dbg = (lambda m: print("bench.py[%d]: %s" % (rank, m), file=sys.stderr, flush=True)) if os.environ.get("VTMC_BENCH_DEBUG") else (lambda m: None)
    d_field = None
    try:
        ex = pipe.exs[0]
        stream = pipe.streams[0]
        prm = vt.density_params(args.kind, n)

        # -- setup (untimed): density field generated on the device, chunk by chunk with halos -------
        d_field = torch.empty(max(per_rank, 1) * dim ** 3, dtype=torch.float32, device="cuda")

        def generate(chunk_ids):
            t0 = time.perf_counter()
            ex.density_fill_device(prm, sharding.origins_of(world_dims, c, chunk_ids), (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d_field.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        sampler_s = generate(my_chunks)
        sampler_kernel_ms = ex.last_fill_ms()

        # ONE communicator per rank, whatever the depth: the first context owns it, the others issue their all-gathers through it
        pipe.set_comm(native_comm(pipe.exs[0], rank, world, backend, dist, torch))
        # rehearsal hook for a one-GPU box: the N > 1 host path (collective, pinned copy, offsets) through a world-of-one communicator
        force_comm = world == 1 and os.environ.get("VTMC_BENCH_FORCE_COMM") == "1"
        if force_comm:
            pipe.world_of_one_comm()
        exchange = world > 1 or force_comm
        # N > 1: a rank's kernels take 0.14 ms each, and the HIP events between them cost 2 % of its step (DESIGN_HISTORY.md, round 3): only every eighth
        # step carries them (the same steps whose collective is timed); at N = 1 every step does (0.3 % of a step)
        sample_stages = world > 1
        kw = {"backend": backend, "dist": dist, "sharding": sharding}
        pipe.bind(d_field.data_ptr(), n_chunks, c, world, per_rank, sharding.slot_permutation(assignment, per_rank), exchange, sample_stages)

        wd.stage("warmup", 240)
        test_hang(wd, "warmup")
        n_warm = max(args.warmup, depth)   # every context once at least: output buffers grow to their size, RCCL builds its channels
        T, offs = pipe.run_steps(depth, False, **kw)
        balance = None
        if strong and exchange:
            costs = np.diff(offs[:, 1])   # every chunk's triangle count, global chunk order: the same array on every rank
            balance = {"rule": args.assign, "triangles_max_over_mean_modulo": round(sharding.imbalance(costs, assignment), 4)}
            if args.assign == "balanced":
                assignment = sharding.balanced_assignment(costs, world)
                balance["triangles_max_over_mean"] = round(sharding.imbalance(costs, assignment), 4)
                my_chunks = assignment[rank]
                n_chunks = len(my_chunks)
                torch.cuda.synchronize()
                generate(my_chunks)
                pipe.bind(d_field.data_ptr(), n_chunks, c, world, per_rank, sharding.slot_permutation(assignment, per_rank), exchange, sample_stages)
                T, offs = pipe.run_steps(depth, False, **kw)
                assert np.array_equal(np.diff(offs[:, 1]), costs), "the re-cut world's per-chunk counts differ from the first cut's"
        if n_warm > depth:
            T, offs = pipe.run_steps(n_warm - depth, False, **kw)
        placement = None
        if args.place_outputs > 1:
            trials = [e.last_placement() for e in pipe.exs]
            placement = {"candidates": min(args.place_outputs, 16),
                         "emit_ms_by_context": [t[0] for t in trials], "kept": [t[1] for t in trials],
                         "note": "tuning key place_outputs: at each context's first warm-up step the emit stage was run into this many allocations of the output buffer "
                                 "and the fastest kept (the emit kernel's time is a property of the pair input allocation / output allocation; candidate 0 is what hipMalloc gave first)"}
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wd.stage("timed", 120 + 0.5 * args.steps)
        t0 = time.perf_counter()
        T, offs = pipe.run_steps(args.steps, True, **kw)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        elapsed, total_tris = reduce_max_sum(elapsed, T, world, backend, torch, dist)
        if exchange:
            assert int(offs[-1, 1]) == int(total_tris), "gathered chunk counts do not add up to the ranks' triangle totals"

        ms_per_step = elapsed / args.steps * 1e3
        cells_total = float(world_dims[0]) * world_dims[1] * world_dims[2]
        value = cells_total / (elapsed / args.steps) / 1e6
        if world == 1:
            wl = "%s %d^3 cells as %d chunks of %d^3 (%d^3 samples incl. halo), resident in HBM" % (args.kind, n, n_chunks, c, dim)
        elif strong:
            wl = ("%s %d^3 cells as %d chunks of %d^3, %d per rank (%s), all-gather of per-chunk counts"
                  % (args.kind, n, n_chunks_total, c, n_chunks, "cut by the first step's triangle counts" if args.assign == "balanced" else "chunk c -> rank c %% %d" % world))
        else:
            wl = "%s %d^3 cells per GPU as %d chunks of %d^3, chunk c -> rank c %% N (weak scaling world 1024 x 1024 x 1024N)" % (args.kind, n, n_chunks, c)
        fallback = os.environ.get("VTMC_BENCH_FALLBACK") == "1"
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
                       "collective": None if not exchange else (("rccl all-gather via libvtmc (vtmc_allgather_volume_counts), %s" % ("on the rank's one collective stream behind the emit launch's event" if pipe.side is not None else "on the extract's stream")) if pipe.native
                                                              else "torch.distributed all_gather (%s)" % backend),
                       "chunk_assignment": balance},
            "mtris_per_s": round(total_tris / (elapsed / args.steps) / 1e6, 1),
            "triangles_rank0": int(T),
            "triangles_total": int(total_tris),
            "pipeline_depth": depth,
            "streams_mode": "a stream per context, each on a hardware queue of its own (--streams 2)" if pipe.own_queue else "one ordinary stream for every context (--streams 1)",
            "stream_count": len(pipe.streams),
            "output_placement": placement,
            "fallback": fallback,
        }
        if fallback:
            out["fallback_reason"] = os.environ.get("VTMC_BENCH_FALLBACK_REASON", "")
        report("done")
        if rank == 0:   # a first, short form of the line: if anything behind the timed region fails, the measurement itself is not lost
            emit_partial(out, "everything behind the timed region (kernel rooflines, indexed output, CPU leg, rehearsal) is missing: the worker ended before it sent the full line")

        # The kernels' own durations.  With a stream per context the HIP events around a kernel also see the time it waits for CUs beside the
        # other context's kernels (a classify kernel "takes" 1.9 ms there): the roofline of a KERNEL needs it alone on the chip.  A second
        # region of the same K steps, the same contexts taking turns, all on the first context's stream (rounds 1-4's timed region, two contexts then).
        wd.stage("one_stream_region", 120 + 0.5 * args.steps)
        live = {k: v / max(pipe.stage_steps, 1) for k, v in pipe.stage_acc.items()}
        gather_ms = list(pipe.gather_ms)
        serial_ms_per_step = ms_per_step
        if len(pipe.streams) > 1:
            pipe.set_one_stream(True)
            pipe.reset_stats()
            pipe.run_steps(depth, False, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipe.run_steps(args.steps, True, **kw)
            torch.cuda.synchronize()
            serial_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
            pipe.set_one_stream(False)
        avg = {k: v / max(pipe.stage_steps, 1) for k, v in pipe.stage_acc.items()}
        # the latency of an isolated step (queue, one host wait), outside the timed region: what --pipeline 1 measures
        lat = []
        iso = {"classify": [], "scan": [], "emit": [], "total": []}   # the three kernels with nothing beside them (no second step in flight)
        for _ in range(10):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipe.queue(pipe.slots[0], **kw)
            pipe.complete(pipe.slots[0])
            lat.append((time.perf_counter() - t1) * 1e3)
            for k, v in pipe.slots[0].ex.last_stage_ms().items():
                iso[k].append(v)
        step_latency_ms = statistics.median(lat)
        iso = {k: statistics.median(v) for k, v in iso.items()}

        if rank == 0:
            # -- roofline of the dominant kernel (rank 0's launches, HIP events inside libvtmc) --------
            samples = n_chunks * dim ** 3
            chunk_tris, chunk_active = per_chunk_counts(pipe.slots[0].ex, n_chunks, bpv)
            n_active = int(chunk_active.sum())
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
                        "algorithmic_bytes": alg[dom], "avg_ms": round(avg[dom], 4),
                        "region": "timed" if len(pipe.streams) == 1 else "one_stream"}
            roofline["measured"] = ("HIP events on the kernels' stream over %d steps, the contexts taking turns on ONE stream" % args.steps) + (
                "" if len(pipe.streams) == 1 else " -- a second region behind the timed one (%.4f ms per step there; `value` comes from the timed region): in the timed region the contexts have a "
                "stream each and a kernel's events also see the time it shares the chip with the other context's kernels, see kernels.*.two_queue_span_ms" % serial_ms_per_step)
            per_kernel = {k: {"avg_ms": round(avg[k], 4), "alg_GBps": round(alg[k] / (avg[k] * 1e-3) / 1e9, 1) if avg[k] > 0 else None,
                              "two_queue_span_ms": round(live[k], 4) if len(pipe.streams) > 1 else None, "isolated_step_ms": round(iso[k], 4)}
                          for k in ("classify", "scan", "emit")}
            # SURVEY.md 8d whole-path figure on one rank: 4*S + 76*T + 8*C over the time a step takes (wall clock of the timed region: with
            # two streams the kernels of neighbouring steps overlap, the sum of their durations is more than a step)
            path_bytes = 4.0 * samples + 76.0 * T + 8.0 * n_chunks
            path = {"bytes": path_bytes, "step_ms": round(ms_per_step, 4), "kernel_ms_sum": round(avg["total"], 4),
                    "achieved_GBps": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                    "frac_of_peak": round(path_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "read_only_frac_of_peak": round(4.0 * samples / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            out.update({
                "active_blocks_rank0": n_active,
                "roofline": roofline,
                "kernels": per_kernel,
                "path_roofline": path,
                "allgather_ms": None if not gather_ms else {"avg": round(statistics.mean(gather_ms), 4), "max": round(max(gather_ms), 4),
                                                         "note": "rank 0, HIP events from the end of the emit kernel to the end of the collective (sampled on every eighth step)"},
                "one_stream_ms_per_step": round(serial_ms_per_step, 4),
                "step_latency_ms": round(step_latency_ms, 4),
                "sampler_s": round(sampler_s, 4),
                "sampler_kernel_ms": round(sampler_kernel_ms, 3),
            })
            emit_partial(out, "indexed output, CPU leg, rehearsal and stream record are missing: the worker ended before it sent the full line")

            # the same workload in the welded (indexed) output format -- 24 B per vertex + 12 B per triangle instead of 76 B
            # per triangle: a few steps after the timed region, N = 1 only (not part of `value`)
            indexed = None
            if world == 1 and not args.no_indexed:
                wd.stage("indexed", 120)
                exs, slots = pipe.exs, pipe.slots
                d_ptr, flags = pipe.d_ptr, pipe.flags
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
                    pipe.set_one_stream(True)
                    run_indexed(K, acc)
                    torch.cuda.synchronize()
                    pipe.set_one_stream(False)
                    V = ex.last_vertex_count()
                    ibytes = 4.0 * samples + 24.0 * V + 12.0 * Ti + 8.0 * n_chunks
                    ebytes = 24.0 * V + 12.0 * Ti + 4000.0 * n_active
                    indexed = {"ms_per_step": round(ms_i, 4), "mvoxels_per_s": round(cells_total / (ms_i * 1e-3) / 1e6, 1),
                               "vertices": int(V), "triangles": int(Ti), "output_bytes": 24.0 * V + 12.0 * Ti,
                               "output_bytes_vs_soup": round((24.0 * V + 12.0 * Ti) / (76.0 * Ti), 4),
                               "kernels_ms": {k: round(v, 4) for k, v in acc.items()},
                               "emit_alg_GBps": round(ebytes / (acc["emit"] * 1e-3) / 1e9, 1) if acc["emit"] > 0 else None,
                               "path_roofline": {"bytes": ibytes, "achieved_GBps": round(ibytes / (ms_i * 1e-3) / 1e9, 1),
                                                 "frac_of_peak": round(ibytes / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "read_only_frac_of_peak": round(4.0 * samples / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                               "speedup_over_soup_step": round(ms_per_step / ms_i, 3)}
                finally:
                    for e in exs:
                        e.set_output_mode(False)
            out["indexed_output"] = indexed

            # every rank of an N = 2 / 4 / 8 run rehearsed on this GPU, with the N > 1 default configuration (this pipeline's own contexts)
            predicted = None
            if world == 1 and not args.no_rehearsal and n_chunks >= 8:
                wd.stage("rehearsal", 240)
                try:
                    if not pipe.native:
                        pipe.world_of_one_comm()
                    reh = rehearse_ranks(torch, pipe, d_field, n_chunks, c, chunk_tris, chunk_active, assign=args.assign, log=dbg)
                    predicted = {"2": reh["ranks"].get("2", {}).get("predicted_scaling"), "4": reh["ranks"].get("4", {}).get("predicted_scaling"),
                                 "8": reh["ranks"].get("8", {}).get("predicted_scaling"),
                                 "note": "REHEARSAL on one GPU, not a measurement: the whole world's step / the slowest rank's step, same process, same configuration (rank_rehearsal)",
                                 }
                    out["rank_rehearsal"] = reh
                except Exception as e:   # noqa: BLE001 -- the rehearsal is an extra: its failure is reported, the line stays
                    predicted = {"error": "%s: %s" % (type(e).__name__, e)}
            out["predicted_scaling"] = predicted

            cpu = None
            if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
                wd.stage("cpu_baseline", 240)
                k = min(args.cpu_sample_chunks, n_chunks)
                vols = [d_field[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy() for v in range(k)]
                cpu = cpu_baseline(vols, dim, c, args.kind, args.cpu_threads, torch.cuda.device_count())
            out["cpu_baseline"] = cpu
    finally:
        # nothing of this run stays on the device: contexts closed (communicator's borrowers first), the field and torch's cache released --
        # in a `finally`, so a failed assertion above ends as its own traceback and not as an abort in the interpreter's tear-down
        wd.stage("teardown", 120)
        dbg("releasing")
        try:
            pipe.close()
            d_field = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            print("bench.py[%d]: release failed: %s" % (rank, e), file=sys.stderr)
        dbg("contexts closed")
    if rank == 0:
        if world == 1 and not args.no_stream_record and c == 128 and (n == 1024 or args.stream_record_cells != 2048):
            wd.stage("stream_record", 240)
            emit_partial(out, "the stream2048 sub-record is missing: the worker ended before it sent the full line")
            try:
                out["stream2048"] = stream_sub_record(torch, args, local)
            except Exception as e:   # noqa: BLE001
                out["stream2048"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            out["stream2048"] = None
        if world == 1 and not args.no_terrain_record:
            wd.stage("terrain_record", 240)
            emit_partial(out, "the terrain sub-record is missing: the worker ended before it sent the full line")
            try:
                import volumetricterrain_amd as vt_
                out["terrain"] = terrain_sub_record(vt_)
            except Exception as e:   # noqa: BLE001
                out["terrain"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            out["terrain"] = None
        if _REPORT_FD is None and world == 1 and not args.no_box:   # --direct: no supervisor to do it; the contexts are closed by now
            wd.stage("box", 180)
            out = add_box(out)
        emit_line(out)
    wd.stage("teardown", 120)
    if world > 1:
        dist.destroy_process_group()
    dbg("process group destroyed")


# ------------------------------------------------------------------------------------------------
# stream2048: sampler + extractor, double-buffered batches (configs[4])
# ------------------------------------------------------------------------------------------------
def run_stream(args, torch, dist, wd):
    wd.stage("init", 300)
    rank, world, local, backend = init_distributed(args, torch, dist)
    n, c = args.n, args.chunk
    wd.stage("timed", 600)
    m = measure_stream(torch, dist, args, rank, world, local, backend)
    n_chunks, total, counts, fill_ms, stage = m["n_chunks"], m["total"], m["counts"], m["fill_ms"], m["stage"]
    elapsed, total_tris = reduce_max_sum(m["elapsed"], total, world, backend, torch, dist)
    if world > 1:   # the same single exchange as config 4: per-chunk counts of every rank
        from volumetricterrain_amd import sharding
        per_rank = ((n // c) ** 3 + world - 1) // world
        loc = torch.zeros((per_rank, 2), dtype=torch.int32)
        loc[:n_chunks] = torch.from_numpy(counts.astype(np.int32))
        g = sharding.allgather_counts(loc.cuda() if backend == "nccl" else loc)
        assert int(g[..., 1].sum()) == int(total_tris)
    report("done")
    if rank == 0:
        step_s = elapsed / args.steps
        cells_total = float(n) ** 3
        samples = n_chunks * m["dim"] ** 3
        nb = len(fill_ms)
        kern = {"density_column_kernel": sum(fill_ms), "classify_dense_kernel": stage["classify"], "scan": stage["scan"], "emit_kernel": stage["emit"]}
        dom = max(kern, key=kern.get)
        octaves = m["octaves"]
        if dom == "density_column_kernel":
            alg_bytes = 4.0 * samples / nb          # per launch: every sample written once
        elif dom == "classify_dense_kernel":
            alg_bytes = (4.0 * samples + 4.0 * n_chunks * m["bpv"]) / nb
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
        rates = [min(1.0, m["frequency"] * (m["lacunarity"] ** o)) for o in range(octaves)]
        lane_ops = samples * (1.5 * octaves + 7.0 + 26.0 * sum(rates) + 7.0 * max(rates) + (2 * 26.0 + 25.0) * octaves / m["dim"])
        sampler_flops = samples * (2.0 * 3.0 * octaves + 2.0 * 34.0 * sum(rates))   # the arithmetic itself: 3 fmas per octave and sample, ~34 flop-pairs a face
        fallback = os.environ.get("VTMC_BENCH_FALLBACK") == "1"
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
                                   "sampling + per-vertex normals + extraction" % (args.kind, n, n_chunks, c, world, m["batch"]),
                       "grid": n, "chunk": c, "chunks_per_gpu": n_chunks, "kind": args.kind, "seed": 1337, "batch_chunks": m["batch"]},
            "mtris_per_s": round(total_tris / step_s / 1e6, 1),
            "triangles_total": int(total_tris),
            "samples_GB_rank0": round(samples * 4 / 1e9, 2),
            "passes_ms": [round(x, 3) for x in m["pass_ms"]],
            "fallback": fallback,
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
            "serialised_step_ms": round(m["serial_s"] * 1e3, 3),
            "overlap_gain": round(m["serial_s"] / step_s, 3),
        }
        if fallback:
            out["fallback_reason"] = os.environ.get("VTMC_BENCH_FALLBACK_REASON", "")
        emit_partial(out, "the CPU leg is missing: the worker ended before it sent the full line")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            wd.stage("cpu_baseline", 300)
            cpu = cpu_baseline_stream(n, c, args.kind, m["origins"], counts[:, 1], args.cpu_threads, torch.cuda.device_count())
        out["cpu_baseline"] = cpu
        emit_line(out)
    wd.stage("teardown", 120)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# stub worker: the launch / watchdog / fallback logic on a box without a GPU (tests/test_bench_launch.py)
# ------------------------------------------------------------------------------------------------
def run_stub(args, wd, spec):
    """VTMC_BENCH_STUB='{"hang_stage": "warmup", "hang_rank": 1, "attempts": [0], "budget_s": 3}': no GPU, no extraction, no number -- the same
    stages, reports, rendezvous (gloo) and final reduction as a real worker, with a rank that stops answering where the test says.  Its
    line says "stub": true and carries value 0: it cannot be mistaken for a measurement."""
    import torch
    import torch.distributed as dist
    attempt = 1 if os.environ.get("VTMC_BENCH_FALLBACK") == "1" else 0
    budget = float(spec.get("budget_s", 5))

    def maybe_hang(stage, rank):
        if spec.get("hang_stage") == stage and rank == int(spec.get("hang_rank", 0)) and attempt in spec.get("attempts", [0]):
            time.sleep(1e6)
        if spec.get("crash_stage") == stage and rank == int(spec.get("crash_rank", 0)) and attempt in spec.get("attempts", [0]):
            os._exit(int(spec.get("crash_code", 134)))

    wd.stage("init", 60)
    rank, world, _, backend = init_distributed(args, torch, dist, need_gpu=False)
    wd.stage("setup", budget)
    maybe_hang("setup", rank)
    wd.stage("warmup", budget)
    maybe_hang("warmup", rank)
    if world > 1:
        dist.barrier()
    wd.stage("timed", budget)
    maybe_hang("timed", rank)
    t0 = time.perf_counter()
    elapsed, total = reduce_max_sum(time.perf_counter() - t0 + 1e-3, rank + 1, world, "gloo", torch, dist)
    report("done")
    if rank == 0:
        emit_line({"metric": "stub", "stub": True, "value": 0.0, "unit": "none", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "fallback": attempt == 1, "fallback_reason": os.environ.get("VTMC_BENCH_FALLBACK_REASON", ""), "ranks_sum": int(total),
                   "pipeline_depth": args.pipeline, "streams_arg": args.streams, "gather_stream": args.gather_stream, "assign": args.assign, "place_outputs": args.place_outputs})
    wd.stage("teardown", budget)
    maybe_hang("teardown", rank)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# the box's memory, measured by a fresh process after the timed regions (tools/calib/mix2 box)
# ------------------------------------------------------------------------------------------------
def box_calibration(timeout_s=120):
    """{read, write, copy, mix}_TBps of THIS box from tools/calib/mix2 (built by __graft_entry__.build()): plain float4 streams over 4 GiB, the
    best of a few launch shapes each, median of 5 launches -- the numbers `roofline.frac_of_box` divides by.  Run by a process that has not
    touched the GPU (the supervisor, after the worker has ended), or by the --direct worker after it has closed its contexts."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "calib", "mix2")
    if not os.path.exists(exe):
        return {"error": "tools/calib/mix2 is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    try:
        p = subprocess.run([exe, "4", "box"], capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "tools/calib/mix2 box did not finish within %d s" % timeout_s}
    rows = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not rows:
        return {"error": "tools/calib/mix2 box: exit code %d: %s" % (p.returncode, p.stderr.strip()[-300:])}
    r = rows[-1]
    return {"read_TBps": r["read_TBps"], "write_TBps": r["write_TBps"], "copy_TBps": r["copy_TBps"], "mix_TBps": r["mix_4r7w_TBps"],
            "emit_shape_TBps": r["emit_shape_TBps"], "device": r.get("device"),
            "how": "tools/calib/mix2 4 box in a fresh process after the timed regions: plain float4 read / write / copy streams and a 4 : 7 read : write mix "
                   "(the emit kernel moves 36 : 64) over 4 GiB, best launch shape of each, median of 5 launches; emit_shape = 40-byte rows gathered + 76-byte records streamed, no arithmetic"}


def add_box(out):
    """`box` and the two kernels' rates as fractions of what the box delivers: classify against the plain read stream, emit against the mix."""
    box = box_calibration()
    out["box"] = box
    if "error" in box or not out.get("kernels"):
        return out
    k = out["kernels"]
    fob = {}
    if k.get("classify", {}).get("alg_GBps"):
        fob["classify_vs_read"] = round(k["classify"]["alg_GBps"] / (box["read_TBps"] * 1e3), 4)
    if k.get("emit", {}).get("alg_GBps"):
        fob["emit_vs_mix"] = round(k["emit"]["alg_GBps"] / (box["mix_TBps"] * 1e3), 4)
    ix = out.get("indexed_output") or {}
    if ix.get("emit_alg_GBps"):
        fob["indexed_emit_vs_mix"] = round(ix["emit_alg_GBps"] / (box["mix_TBps"] * 1e3), 4)
    if isinstance(out.get("roofline"), dict):
        out["roofline"]["frac_of_box"] = fob.get("classify_vs_read" if out["roofline"].get("kernel") == "classify_kernel" else "emit_vs_mix")
        out["roofline"]["frac_of_box_all"] = fob
    return out


# ------------------------------------------------------------------------------------------------
# processes: supervisor -> worker (-> one conservative second attempt)
# ------------------------------------------------------------------------------------------------
def release_library_streams():
    """Every context of this process is closed and nothing of torch's refers to their streams any more (run_grid / ChunkStream release in a
    `finally`): the library's parked streams are destroyed now.  Without it a worker that is profiled (rocprofv3 -- python3 bench.py --direct)
    ends with own-queue streams alive and crashes in the profiler's finalisation, profile lost."""
    try:
        import gc
        gc.collect()
        vt = sys.modules.get("volumetricterrain_amd")
        if vt is not None:
            vt.release_streams()
    except Exception as e:   # noqa: BLE001
        print("bench.py: vtmc_release_streams failed: %s" % e, file=sys.stderr)


def worker_main(args):
    """The measuring process.  Under a supervisor: VTMC_BENCH_REPORT_FD names the pipe.  --direct: the line goes to the real stdout."""
    global _REPORT_FD, _REAL_STDOUT
    import faulthandler
    faulthandler.enable()
    fd = os.environ.get("VTMC_BENCH_REPORT_FD")
    if fd is not None:
        _REPORT_FD = int(fd)
    else:
        # Libraries write to stdout on their own (RCCL prints a five-line version banner whenever a communicator is created): everything
        # but the JSON line is sent to stderr by pointing fd 1 there; the line itself goes to a duplicate of the original fd 1.
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    wd = Watchdog(os.environ.get("RANK", "0"))
    wd.stage("import", 420)    # the first `import torch` on a fresh box pages the image in: one to two minutes
    try:
        stub = os.environ.get("VTMC_BENCH_STUB")
        if stub:
            run_stub(args, wd, json.loads(stub))
        else:
            import torch
            import torch.distributed as dist
            if args.config == "stream2048":
                run_stream(args, torch, dist, wd)
            else:
                run_grid(args, torch, dist, wd)
    except Refusal as e:
        print(str(e), file=sys.stderr)
        wd.disarm()
        return 4
    finally:
        release_library_streams()
    wd.disarm()
    return 0


def _forwarding(child):
    """SIGTERM / SIGINT to this process go to the child's whole process group: a killed bench never leaves ranks behind holding GPUs."""
    import signal

    def forward(signum, _frame):
        try:
            os.killpg(child.pid, signum)
        except ProcessLookupError:
            pass

    return {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT)}


def run_worker(argv, extra_env, tag):
    """Starts one worker (a fresh `python bench.py ...` with VTMC_BENCH_ROLE=worker, in a process group of its own, its stdout pointed at our
    stderr) and reads its reports until it ends.  Returns {"rc", "line", "done", "stage", "watchdog", "killed"}.  Backstop to the worker's own
    watchdog: a worker that is `grace` seconds past the bound of the stage it reported (its watchdog thread never ran) is killed."""
    import select
    import signal
    import subprocess
    r, w = os.pipe()
    env = dict(os.environ)
    env.update(extra_env)
    env["VTMC_BENCH_ROLE"] = "worker"
    env["VTMC_BENCH_REPORT_FD"] = str(w)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, pass_fds=(w,), stdout=sys.stderr, start_new_session=True)
    os.close(w)
    old = _forwarding(child)
    res = {"rc": None, "line": None, "done": False, "stage": "start", "watchdog": None, "killed": False}
    grace = 30.0 * float(os.environ.get("VTMC_BENCH_WATCHDOG_SCALE", "1"))
    deadline = time.monotonic() + 600
    buf = b""
    def feed(chunk):
        nonlocal buf, deadline
        buf += chunk
        while b"\n" in buf:
            raw, buf = buf.split(b"\n", 1)
            try:
                msg = json.loads(raw)
            except ValueError:
                continue
            k = msg.get("k")
            if k == "stage":
                res["stage"] = msg["name"]
                deadline = time.monotonic() + float(msg["budget_s"]) + grace
            elif k == "line":
                res["line"] = msg["line"]
            elif k == "done":
                res["done"] = True
            elif k == "watchdog":
                res["watchdog"] = msg["name"]

    try:
        eof = False
        while True:
            if not eof:
                ready, _, _ = select.select([r], [], [], 0.25)
                if ready:
                    chunk = os.read(r, 1 << 16)
                    if chunk:
                        feed(chunk)
                    else:
                        eof = True
            else:
                time.sleep(0.1)
            if child.poll() is not None:
                while not eof and select.select([r], [], [], 0)[0]:   # what the worker wrote before it ended
                    chunk = os.read(r, 1 << 16)
                    if not chunk:
                        break
                    feed(chunk)
                break
            if time.monotonic() > deadline:
                print("bench.py[%s]: supervisor: the worker is %.0f s past the bound of stage '%s' and its own watchdog has not ended it: SIGKILL" % (tag, grace, res["stage"]), file=sys.stderr)
                res["killed"] = True
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                deadline = time.monotonic() + 3600
        res["rc"] = child.wait()
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            res["rc"] = child.wait()
        os.close(r)
    return res


def supervise(args, argv):
    """Never imports torch, never touches a GPU.  One worker; if it fails before its timed region is complete (no "done" report), ONE more
    with FALLBACK_ARGS.  Prints the last line the worker(s) sent -- exactly once -- with the box calibration merged in (N = 1)."""
    rank = os.environ.get("RANK", "0")
    tag = "rank " + rank
    first = run_worker(argv, {}, tag)
    res, attempts = first, 1
    if first["rc"] not in (0, 4) and not first["done"] and not args.no_fallback and args.config == "grid1024":   # 4: refused to run at all
        reason = "first attempt: %s in stage '%s'" % (("watchdog" if first["watchdog"] else "killed by the supervisor" if first["killed"] else "exit code %s" % first["rc"]), first["watchdog"] or first["stage"])
        print("bench.py[%s]: supervisor: %s -- starting ONE fresh worker with the conservative configuration (%s)" % (tag, reason, " ".join(FALLBACK_ARGS)), file=sys.stderr)
        res = run_worker(argv + FALLBACK_ARGS, {"VTMC_BENCH_FALLBACK": "1", "VTMC_BENCH_FALLBACK_REASON": reason}, tag)
        attempts = 2
    if res["line"] is not None:
        out = json.loads(res["line"])
        out["worker"] = {"attempts": attempts, "exit_code": res["rc"], "last_stage": res["stage"]}
        if res["rc"] != 0:
            print("bench.py[%s]: supervisor: the worker delivered its line and then ended with exit code %s in stage '%s'" % (tag, res["rc"], res["stage"]), file=sys.stderr)
        if out.get("n_gpus") == 1 and args.config == "grid1024" and not args.no_box and not out.get("stub"):
            out = add_box(out)
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
        return 0 if (res["rc"] == 0 or res["done"]) else (res["rc"] if 0 < res["rc"] < 256 else 1)
    if res["rc"] == 0:
        return 0    # a rank other than 0: it has no line to print
    return res["rc"] if 0 < res["rc"] < 256 else 1


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
    old = _forwarding(child)
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
    args = parse()
    if os.environ.get("VTMC_BENCH_ROLE") == "worker":
        sys.exit(worker_main(args))
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("VTMC_BENCH_SELF_LAUNCHED") != "1":
        sys.exit(self_launch(args))
    if args.direct:
        rc = worker_main(args)
        sys.exit(rc)
    sys.exit(supervise(args, sys.argv[1:]))


if __name__ == "__main__":
    main()
