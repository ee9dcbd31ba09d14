import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's own DataLoader pin_memory thread calls deprecated Tensor.pin_memory(device)/is_pinned(device)
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor:DeprecationWarning")


@pytest.fixture(scope="session")
def built_lib():
    """libnomad_hip.so (and libnomad_diag.so), (re)built in-tree with hipcc if sources changed."""
    from nomad_amd import build
    build.build_all()
    return build.LIB


@pytest.fixture(scope="session")
def sd0():
    from nomad_amd.weights import seeded_state_dict
    return seeded_state_dict(0)


@pytest.fixture(scope="session")
def sd_peaky():
    from nomad_amd.weights import seeded_state_dict
    return seeded_state_dict(1, qk_gain=6.0)


@pytest.fixture(scope="session")
def engine(built_lib, sd0):
    import torch
    from nomad_amd.engine import Engine
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    eng = Engine(sd0, 0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def engine_peaky(built_lib, sd_peaky):
    import torch
    from nomad_amd.engine import Engine
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    eng = Engine(sd_peaky, 0)
    yield eng
    eng.close()


# Kernel instantiations that exist in the PRODUCT library (what the scoring / training paths can select); every other
# nomad_diag_gemm* tile id is an experiment that lives in libnomad_diag.so only (nomad_amd/build.py).
PRODUCT_TILES = {"f32": {20, 31, 33, 34, 37, 48}, "bf16": {1, 2, 3, 4, 16, 55, 57, 58, 60}, "bf16x3": {7, 8}}


@pytest.fixture(scope="session")
def engine_diag(built_lib, sd0):
    """An engine on libnomad_diag.so, for the tests of experimental kernel instantiations."""
    import torch
    from nomad_amd.engine import Engine
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    eng = Engine(sd0, 0, diag=True)
    yield eng
    eng.close()


@pytest.fixture
def engine_for(engine, request):
    """engine_for(kind, tile) -> the product engine when that instantiation ships in libnomad_hip.so, else the diag one."""
    def pick(kind, tile):
        return engine if tile in PRODUCT_TILES[kind] else request.getfixturevalue("engine_diag")
    return pick
