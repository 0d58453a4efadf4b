import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_ckpt(tag):
    ck = torch.load(os.path.join(GOLDEN, f"synthetic_{tag}.ckpt"), map_location="cpu", weights_only=True)
    return ck["state_dict"], ck["hyper_parameters"]["config"]


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session", params=["k9", "k6"])
def model_case(request):
    tag = request.param
    sd, cfg = load_ckpt(tag)
    return tag, sd, cfg, load_npz(f"stages_{tag}.npz")
