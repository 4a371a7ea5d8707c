"""bench.py's modes at reduced size on the GPU box: the N = 1 line (roofline + reproducible
cpu_baseline), BASELINE configs[3] as STRONG scaling (two ranks rehearsed on one device over gloo:
RCCL refuses two ranks per device, the driver's runs use one GPU per rank and the library's own RCCL
all-gather) and configs[4] as the streamed sampler + extractor."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, env=None, drop=()):
    e = dict(os.environ)
    for k in drop:
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_single_gpu_line_carries_roofline_and_reproducible_cpu_leg():
    j = run([sys.executable, "bench.py", "--grid", "256", "--steps", "3", "--warmup", "1", "--cpu-sample-chunks", "2", "--stream-record-cells", "256"])
    assert j["n_gpus"] == 1 and j["unit"] == "Mvoxels/s" and j["value"] > 0 and j["scaling"] == "strong"
    assert "partial" not in j and j["fallback"] is False and j["worker"] == {"attempts": 1, "exit_code": 0, "last_stage": "teardown"}    # supervisor -> worker, first attempt, clean exit
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] in ("classify_kernel", "emit_kernel") and 0 < r["frac"] < 1
    assert "traffic_source" in r and r["region"] == "one_stream"
    assert j["pipeline_depth"] == 4 and j["stream_count"] == 4 and "--streams 2" in j["streams_mode"] and j["step_latency_ms"] > 0      # throughput with four steps in flight on a stream each, isolated-step latency beside it
    assert "ONE stream" in r["measured"] and all(k["isolated_step_ms"] > 0 and k["two_queue_span_ms"] >= k["avg_ms"] * 0.8 for k in j["kernels"].values())
    # round 6: output placement trials, disclosed: every context's candidates' emit times, the fastest kept
    op = j["output_placement"]
    assert op["candidates"] == 8 and len(op["emit_ms_by_context"]) == 4 and all(len(c) == 8 and min(c) > 0 for c in op["emit_ms_by_context"])
    assert all(c[k] == min(c) for c, k in zip(op["emit_ms_by_context"], op["kept"]))
    # round 6: the box's own memory rates (a fresh process after the timed regions) and the kernels' rates as fractions of them
    b = j["box"]
    assert "error" not in b, b
    assert 2.0 < b["read_TBps"] < 8.0 and 2.0 < b["write_TBps"] < 8.0 and 2.0 < b["copy_TBps"] < 8.0 and 2.0 < b["mix_TBps"] < 8.0
    assert r["frac_of_box"] is not None and 0 < r["frac_of_box"] < 1.2 and set(r["frac_of_box_all"]) >= {"classify_vs_read", "emit_vs_mix"}
    # every rank of an N = 2 / 4 / 8 run rehearsed on this GPU (8 chunks here: one per rank at N = 8)
    ps, rr = j["predicted_scaling"], j["rank_rehearsal"]
    assert "error" not in ps, ps
    for w in ("2", "4", "8"):
        q = rr["ranks"][w]
        assert len(q["step_ms"]) == int(w) and all(x > 0 for x in q["step_ms"]) and q["slowest_ms"] == max(q["step_ms"])
        assert abs(ps[w] - rr["world_step_ms"] / q["slowest_ms"]) < 0.01 and 0.5 < ps[w] <= int(w) * 1.5
        assert 1.0 <= q["triangles_max_over_mean"]["balanced"] <= q["triangles_max_over_mean"]["modulo"] + 1e-9
    # SURVEY 8f rank 1 in the driver's line: the world build and interactive edits on a grid resident in HBM
    tr = j["terrain"]
    assert "error" not in tr, tr
    assert tr["world_build"]["update_ms"] > 0 and tr["world_build"]["triangles"] > 0 and tr["edits"]["edit_latency_us_median"] > 0 and tr["edits"]["triangles_median"] > 0
    # BASELINE configs[4] in its short form behind the grid (256^3 here, 2048^3 in the default run)
    s2 = j["stream2048"]
    assert "error" not in s2, s2
    assert s2["ms_per_pass"] > 0 and len(s2["passes_ms"]) == 2 and s2["triangles_total"] > 0 and s2["kernels_ms_per_pass_serialised"]["density_column_kernel"] > 0
    assert j["one_stream_ms_per_step"] > 0
    assert j["path_roofline"]["step_ms"] == j["ms_per_step"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and len(c["repetitions_mvoxels_per_s"]) == 5
    assert all(s >= 0.5 for s in c["repetition_seconds"])
    assert c["min_mvoxels_per_s"] <= c["value"] <= c["max_mvoxels_per_s"]
    assert c["cpu_share"]["affinity_cpus"] >= c["cores"]
    ix = j["indexed_output"]
    assert ix["triangles"] == j["triangles_total"] and 0 < ix["output_bytes_vs_soup"] < 0.45 and ix["ms_per_step"] > 0
    # 256^3 perlin3d as 8 chunks of 128^3: the surface of the one-grid config (2 655 156 triangles with the CPU
    # twin's samples; the device sampler differs from the twin by ~1e-7, which moves the samples that are
    # zero in exact arithmetic -- the noise lattice points -- across the threshold: a few hundred triangles)
    assert abs(j["triangles_total"] - 2655156) < 2000


def test_strong_scaling_two_ranks_on_one_device():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", "bench.py", "--gpus", "2", "--grid", "256", "--steps", "3", "--warmup", "1"]
    j = run(cmd, {"VTMC_BENCH_ONE_DEVICE": "1", "VTMC_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["config"]["chunks_per_gpu"] == 4 and "cut by the first step's triangle counts" in j["config"]["workload"]
    assert abs(j["triangles_total"] - 2655156) < 2000 and 0 < j["triangles_rank0"] < j["triangles_total"]
    assert j["allgather_ms"]["avg"] >= 0 and j["cpu_baseline"] is None
    assert j["pipeline_depth"] == 4   # four contexts take turns; at N > 1 they share one communicator (vtmc_comm_share)
    # round 6: the chunks are cut again by the first step's counts (every rank derives the same partition from the all-gathered pairs)
    ca = j["config"]["chunk_assignment"]
    assert ca["rule"] == "balanced" and 1.0 <= ca["triangles_max_over_mean"] <= ca["triangles_max_over_mean_modulo"] + 1e-9
    jm = run(cmd + ["--assign", "modulo"], {"VTMC_BENCH_ONE_DEVICE": "1", "VTMC_BENCH_BACKEND": "gloo"})
    assert jm["config"]["chunk_assignment"]["rule"] == "modulo" and jm["triangles_total"] == j["triangles_total"] and "c -> rank c % 2" in jm["config"]["workload"]


def test_stream_config_line():
    j = run([sys.executable, "bench.py", "--config", "stream2048", "--grid", "256", "--batch", "3", "--steps", "1"])
    assert j["n_gpus"] == 1 and j["config"]["chunks_per_gpu"] == 8 and j["config"]["kind"] == "fbm8"
    assert j["value"] > 0 and j["triangles_total"] > 0
    cb = j["cpu_baseline"]     # the CPU leg of the stream: the oracle's sampler + extractor on a bounded sample of chunks, thread count stated
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["repetitions_mvoxels_per_s"]) == 5 and "sampled AND extracted" in cb["sample"]
    # round 6: the sample is stratified by the chunks' triangle counts -- it EMITS triangles, about as many per cell as the world does
    assert cb["triangles_in_sample"] > 0 and cb["triangles_per_cell"]["world"] > 0 and cb["triangles_per_cell"]["sample_over_world"] > 0
    assert j["roofline"]["kernel"] in ("density_column_kernel", "classify_dense_kernel", "emit_kernel")
    assert j["roofline"]["bound"] == ("valu" if j["roofline"]["kernel"] == "density_column_kernel" else "hbm")   # the sampler is bound by vector issue
    assert j["sampler_valu"]["frac"] > 0 and j["overlap_gain"] > 0


def test_exchange_path_through_a_world_of_one_communicator():
    """The N > 1 host path of bench.py on one GPU: the library's RCCL all-gather (world of one), the pinned copy + the one
    wait through the ABI, offsets from the gathered pairs -- and the triangle total they must add up to (asserted in bench.py)."""
    j = run([sys.executable, "bench.py", "--grid", "256", "--steps", "9", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record", "--no-box"],
            {"VTMC_BENCH_FORCE_COMM": "1"})
    assert j["n_gpus"] == 1 and abs(j["triangles_total"] - 2655156) < 2000
    assert j["allgather_ms"] is not None and j["allgather_ms"]["avg"] >= 0
    assert j["stream_count"] == 4 and "one collective stream" in j["config"]["collective"]     # the default (round 6): a stream per context, every collective of the rank on ONE ordinary stream behind its emit launch's event
    j0 = run([sys.executable, "bench.py", "--grid", "256", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record", "--no-box", "--streams", "1"],
             {"VTMC_BENCH_FORCE_COMM": "1"})
    assert j0["stream_count"] == 1 and "one collective stream" in j0["config"]["collective"] and j0["triangles_total"] == j["triangles_total"]
    # the collective behind the emit kernel on the step's own stream (the library chains the communicator's collectives across the four streams)
    j3 = run([sys.executable, "bench.py", "--grid", "256", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record", "--no-box", "--gather-stream", "main"],
             {"VTMC_BENCH_FORCE_COMM": "1"})
    assert j3["stream_count"] == 4 and "extract's stream" in j3["config"]["collective"] and j3["triangles_total"] == j["triangles_total"]
    # the fallback's configuration, as the supervisor would start it
    jf = run([sys.executable, "bench.py", "--grid", "256", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record", "--no-box", "--pipeline", "2", "--streams", "1",
              "--gather-stream", "main", "--assign", "modulo"], {"VTMC_BENCH_FORCE_COMM": "1"})
    assert jf["pipeline_depth"] == 2 and jf["stream_count"] == 1 and jf["triangles_total"] == j["triangles_total"]
    j1 = run([sys.executable, "bench.py", "--grid", "256", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record", "--no-box", "--pipeline", "1"],
             {"VTMC_BENCH_FORCE_COMM": "1"})
    assert j1["pipeline_depth"] == 1 and j1["stream_count"] == 1 and j1["triangles_total"] == j["triangles_total"]
    # opt-in: the collective beside the emit kernel, two contexts / communicators taking turns
    j2 = run([sys.executable, "bench.py", "--grid", "256", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record", "--no-box", "--pipeline", "2",
              "--gather-beside"], {"VTMC_BENCH_FORCE_COMM": "1"})
    assert j2["pipeline_depth"] == 2 and j2["triangles_total"] == j["triangles_total"]


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: bench.py starts the two ranks itself as a child process (before it
    touches the GPU) and the line says n_gpus = 2 -- never a world of one under an N = 2 request."""
    env = {"VTMC_BENCH_ONE_DEVICE": "1", "VTMC_BENCH_BACKEND": "gloo"}
    j = run([sys.executable, "bench.py", "--gpus", "2", "--grid", "256", "--steps", "3", "--warmup", "1"], env,
            drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VTMC_BENCH_SELF_LAUNCHED"))
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["chunks_per_gpu"] == 4
    assert abs(j["triangles_total"] - 2655156) < 2000


def test_direct_mode_measures_in_one_process():
    """--direct (profilers put the program itself behind `--`): no supervisor, the worker prints the line itself; the watchdog still applies."""
    j = run([sys.executable, "bench.py", "--direct", "--grid", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal", "--no-stream-record", "--no-terrain-record"])
    assert j["n_gpus"] == 1 and j["value"] > 0 and "worker" not in j and "error" not in j["box"]


def test_a_worker_that_hangs_on_the_gpu_is_replaced_by_the_conservative_one():
    """The real thing on the GPU box: the first worker stops answering in its warm-up (contexts created, field generated, kernels queued), its
    watchdog ends it, and the supervisor -- which never touched the GPU -- starts the fallback worker, whose line is a real measurement with
    the conservative configuration, labelled."""
    e = dict(os.environ, VTMC_BENCH_TEST_HANG="warmup:4")
    p = subprocess.run([sys.executable, "bench.py", "--grid", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-indexed", "--no-rehearsal",
                        "--no-stream-record", "--no-terrain-record", "--no-box"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = lines[0]
    assert j["fallback"] is True and "watchdog in stage 'warmup'" in j["fallback_reason"] and j["worker"]["attempts"] == 2
    assert j["pipeline_depth"] == 2 and j["stream_count"] == 1 and j["value"] > 0 and abs(j["triangles_total"] - 2655156) < 2000
    assert j["output_placement"] is None          # the conservative configuration takes what hipMalloc gives
    assert "WATCHDOG: stage 'warmup'" in p.stderr
