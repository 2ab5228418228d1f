import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def pytest_generate_tests(metafunc):
    """Every `-m gpu` test runs once per DLT form (tests/forms.py): against libeg3d.so (6x4, OpenCV <= 3.1, the
    default) and against libeg3d_dlt4x4.so, the oracle in the matching mode each time."""
    if metafunc.definition.get_closest_marker("gpu") is not None:
        import forms
        if "eg3d_form" not in metafunc.fixturenames:
            metafunc.fixturenames.append("eg3d_form")
        metafunc.parametrize("eg3d_form", list(forms.FORMS), ids=[forms.IDS[r] for r in forms.FORMS], indirect=True)


@pytest.fixture
def eg3d_form(request):
    import forms
    with forms.product_form(request.param) as rows:
        yield rows


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the host library and the oracle (cheap, incremental). The HIP library is built by
    __graft_entry__.build(); GPU tests fail loudly if it is missing."""
    from edgegraph3d_amd import build as b
    b.build_host()
    b.build_oracle()
    yield
