import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle, build
    build(ref=False)
    return Oracle()


@pytest.fixture(scope="session")
def gpu_ctx():
    from fem_amd import capi
    if capi.device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the GPU box")
    ctx = capi.Context(0)
    yield ctx
    ctx.destroy()
