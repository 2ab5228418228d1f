import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the host library and the oracle (cheap, incremental). The HIP library is built by
    __graft_entry__.build(); GPU tests fail loudly if it is missing."""
    from edgegraph3d_amd import build as b
    b.build_host()
    b.build_oracle()
    yield
