import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite.  The driver runs `pytest -m gpu -x`: whatever comes after the first failure is never
# reached, so the rows of SURVEY.md section 8 come first (goldens -> core parity -> the lifecycle sequence -> configs[1] / [2] ->
# indexed output -> terrain -> chunk files -> config 5) and the auxiliary surfaces last (sampler twin, sign-bit classify,
# the tuning matrix, bench modes, the RCCL world-of-one).  First matching pattern wins; unmatched tests go to the end, in file order.
GPU_ORDER = [
    r"test_golden\.py",
    r"test_gpu_parity\.py::test_exact_mode",
    r"test_gpu_parity\.py::test_exact_zero",
    r"test_gpu_parity\.py::test_tile_batch",
    r"test_gpu_parity\.py::test_grid_in_place",
    r"test_gpu_parity\.py::test_non_cubic",
    r"test_gpu_parity\.py::test_dirty_block_list",
    r"test_gpu_parity\.py::test_plane_sphere_empty",
    r"test_gpu_parity\.py::test_special_values",
    r"test_gpu_parity\.py::test_queued_extract",
    r"test_gpu_parity\.py::test_capacity_and_range",
    r"test_gpu_parity\.py::test_error_behaviour",
    r"test_gpu_parity\.py::test_device_volume_batch",
    r"test_lifecycle\.py",
    r"test_gpu_parity\.py::test_config_256",
    r"test_gpu_parity\.py::test_config_1024",
    r"test_gpu_parity\.py::test_max_size",
    r"test_indexed\.py",
    r"test_random_shapes\.py",
    r"test_terrain\.py",
    r"test_host_mirror\.py",
    r"test_gpu_parity\.py::test_chunk_file",
    r"test_gpu_parity\.py::test_sharded_host_entry",
    r"test_gpu_parity\.py::test_streaming_shards",
    r"test_gpu_parity\.py::test_config_streaming",
    r"test_config5_full\.py",
    r"test_gpu_parity\.py::test_density_sampler",
    r"test_gpu_parity\.py::test_classify_from_the_samplers",
    r"test_tuning_matrix\.py",
    r"test_gpu_parity\.py::test_rccl",
    r"test_bench_modes\.py",
    r"test_own_queue_cpp_host\.py",     # last: a plain C++ process on ROCm's own runtime; a hang there costs nothing that ran before it
]
_GPU_ORDER = [re.compile(p) for p in GPU_ORDER]


def gpu_rank(nodeid):
    for i, p in enumerate(_GPU_ORDER):
        if p.search(nodeid):
            return i
    return len(_GPU_ORDER)


def pytest_collection_modifyitems(session, config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu:
        return
    rest = [it for it in items if not it.get_closest_marker("gpu")]
    gpu.sort(key=lambda it: gpu_rank(it.nodeid))   # stable: file order inside a rank
    items[:] = rest + gpu


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
