import importlib
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def svo():
    """The product binding.  `make` first (a no-op when libsvo_amd.so is up to date; hipcc cross-compiles without a GPU),
    so that a fresh checkout tests the real HIP library; importing fails loudly when it is still missing - there is no
    CPU fallback to test instead."""
    import subprocess
    if not os.environ.get("SVO_AMD_LIB"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "octree-raymarcher_amd")], check=True, stdout=subprocess.DEVNULL)
    return importlib.import_module("octree-raymarcher_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding
