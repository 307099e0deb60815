import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once
    if not os.path.exists(os.path.join(ROOT, "fem_amd", "lib", "libfem_amd.so")):
        import __graft_entry__
        __graft_entry__.build()


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
