import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ASSETS = os.path.join(ROOT, "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    import subprocess
    if not (os.path.exists(os.path.join(ROOT, "cuda-pathtracer_amd", "libptamd.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "libpt_oracle.so"))):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)


@pytest.fixture(scope="session")
def P():
    _ensure_built()
    import cuda_pathtracer_amd as P
    return P


@pytest.fixture(scope="session")
def O():
    _ensure_built()
    import pt_oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def indoor(P):
    return P.HostScene.load(os.path.join(ASSETS, "indoor.scene"))


@pytest.fixture(scope="session")
def gpu_ctx(P):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback for the render path")
    ctx = P.Context(0)
    yield ctx
    ctx.close()
