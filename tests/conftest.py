import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build(ref=os.path.isdir("/root/reference"))
    return oracle


@pytest.fixture(scope="session")
def proto256():
    p = np.load(os.path.join(GOLDEN, "prototype_M256_m4_r1.npz"))
    return p["h"], p["g"]


@pytest.fixture(scope="session")
def kinect_pcm():
    return np.load(os.path.join(GOLDEN, "kinect_4ch_16k.npz"))["pcm"].astype(np.float32)


@pytest.fixture(scope="session")
def pygolden():
    return np.load(os.path.join(GOLDEN, "pybeamformer_golden.npz"))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
