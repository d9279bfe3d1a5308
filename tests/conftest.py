import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun); everything else runs on CPU")


def pytest_report_header(config):
    # an experiment build selected with GFPP_LIB_PATH (tools/build_variant.sh) must never pass for the shipped library unnoticed
    lib = os.environ.get("GFPP_LIB_PATH")
    return f"WARNING: GFPP_LIB_PATH={lib}: these tests run on an experiment build, not on genefaceplusplus_amd/libgfpp_radnerf.so" if lib else None


def pytest_sessionstart(session):
    # (genefaceplusplus_amd/_lib.py itself refuses anything but build/variants/lib_<name>.so of this checkout)
    lib = os.environ.get("GFPP_LIB_PATH")
    if lib and not os.path.basename(lib).startswith("lib_"):
        raise pytest.UsageError(f"GFPP_LIB_PATH={lib} is neither the shipped library nor a tools/build_variant.sh build (lib_<name>.so)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_python_golden.npz"))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.build()
    return orc
