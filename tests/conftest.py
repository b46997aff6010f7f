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


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, sd, arr = {}, {}, {}
    for k in z.files:
        v = z[k]
        if k.startswith("cfg/"):
            v = v.tolist()
            cfg[k[4:]] = tuple(v) if isinstance(v, list) and k.endswith("patch_size") and len(v) == 3 else v
        elif k.startswith("sd/"):
            sd[k[3:]] = torch.from_numpy(v)
        else:
            arr[k] = torch.from_numpy(v) if v.dtype.kind in "fiub" else v
    return cfg, sd, arr


@pytest.fixture(scope="session")
def golden():
    return load_golden
