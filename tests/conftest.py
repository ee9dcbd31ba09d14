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
    """libnomad_hip.so, (re)built in-tree with hipcc if sources changed."""
    from nomad_amd import build
    return build.build_library()


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
