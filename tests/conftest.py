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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name, dtype=None):
    """Returns (arrays dict, params dict[str, Tensor]) of one fixture from tests/golden/."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    arrs = {k: z[k] for k in z.files if not k.startswith("p.")}
    params = {}
    for k in z.files:
        if k.startswith("p."):
            t = torch.from_numpy(z[k])
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            params[k[2:]] = t
    return arrs, params


@pytest.fixture(scope="session")
def oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from oracle import focus_oracle
    return focus_oracle
