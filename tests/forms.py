"""The two forms of cv::triangulatePoints' DLT system and what belongs to each in the tests:
rows = 3 -> the 6x4 system of OpenCV <= 3.1 (the release the reference names): the DEFAULT library libeg3d.so;
rows = 2 -> the 4x4 system of later releases: libeg3d_dlt4x4.so. Golden outputs exist per form."""
import contextlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMS = (3, 2)
IDS = {3: "dlt6x4", 2: "dlt4x4"}


def lib_path(rows):
    # EG3D_TEST_LIB_6X4 / EG3D_TEST_LIB_4X4: run the suite against an experimental build of that form (tools/build_variant.sh)
    over = os.environ.get("EG3D_TEST_LIB_6X4" if rows == 3 else "EG3D_TEST_LIB_4X4")
    if over:
        return over
    return os.path.join(ROOT, "edgegraph3d_amd", "libeg3d.so" if rows == 3 else "libeg3d_dlt4x4.so")


def golden_path(base, rows):
    """base = 'synthetic_tiny_v1' / 'synthetic_tiny_sets_v1': the outputs of the form (inputs live in the base file)."""
    return os.path.join(ROOT, "tests", "golden", base + ("_dlt6x4" if rows == 3 else "") + ".npz")


@contextlib.contextmanager
def oracle_rows(rows):
    from oracle import binding as ob
    L = ob.lib()
    before = L.orc_get_dlt_rows()
    assert L.orc_set_dlt_rows(rows) == 0
    try:
        yield L
    finally:
        L.orc_set_dlt_rows(before)


@contextlib.contextmanager
def product_form(rows):
    """Make `rows` the form of everything a GPU test touches: the product library api.lib() returns (and the
    EG3D_LIB its subprocesses inherit) and the oracle's mode."""
    from edgegraph3d_amd import api
    path = lib_path(rows)
    assert os.path.exists(path), "%s is not built (python -m edgegraph3d_amd.build)" % path
    old_env, old_lib = os.environ.get("EG3D_LIB"), api._LIB
    os.environ["EG3D_LIB"] = path
    api._LIB = None
    try:
        assert api.lib().eg3d_dlt_rows() == rows
        with oracle_rows(rows):
            yield rows
    finally:
        api._LIB = old_lib
        if old_env is None:
            os.environ.pop("EG3D_LIB", None)
        else:
            os.environ["EG3D_LIB"] = old_env
