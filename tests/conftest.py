import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (os.path.join(ROOT, "path-tracer_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def veach_sd():
    from pt_hip.scene import SceneData
    return SceneData.load(os.path.join(GOLDEN, "veach_mis_scene.npz"))


@pytest.fixture(scope="session")
def veach_oracle(veach_sd):
    import orc
    return orc.Scene(veach_sd)


@pytest.fixture(scope="session")
def veach_gpu(veach_sd):
    from pt_hip import api
    return api.Scene(veach_sd)
