"""The variant builds of the library (octree-raymarcher_amd/Makefile `variants`, built by __graft_entry__.build()) must not rot
unseen: the C++ march step that kernel_stack.hip.h calls the readable statement of the algorithm, the timing
build whose lane counters DESIGN.md quotes, the large-pool kernel forced onto small worlds, and the test hooks the shipped library
no longer reads.  Each runs in its own process (one library per process), one after the other."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "octree-raymarcher_amd", "build")
CHECK = os.path.join(ROOT, "tests", "variant_check.py")
VARIANTS = [("cxxstep", "march"), ("wide64", "march"), ("timing", "march"), ("timing", "timing"), ("hooks", "hooks"),
            ("suremiss", "march"), ("suremiss64", "march")]      # (the sure-miss brick test in every wave-step, both addressing variants)


def lib_of(name):
    return os.path.join(BUILD, f"libsvo_{name}.so")


def test_variant_libraries_are_built_and_export_the_abi(svo):
    """CPU: every variant exists (build() made it; `make variants` here if a fresh checkout has not), exports every symbol of
    include/svo.h, and the shipped library reads none of the test hooks' environment variables."""
    if not os.environ.get("SVO_AMD_LIB"):       # (a no-op when they are up to date, like the product library's own make in conftest.py)
        subprocess.run(["make", "-j4", "-C", os.path.join(ROOT, "octree-raymarcher_amd"), "variants"], check=True, stdout=subprocess.DEVNULL)
    for name in sorted({n for n, _ in VARIANTS}):
        out = subprocess.run(["nm", "-D", "--defined-only", lib_of(name)], capture_output=True, text=True, check=True).stdout
        have = {line.split()[-1] for line in out.splitlines() if line.strip()}
        missing = [s for s in svo.ABI_SYMBOLS if s not in have]
        assert not missing, (name, missing)
    shipped = open(svo.LIB_PATH, "rb").read() if not os.environ.get("SVO_AMD_LIB") else open(os.path.join(ROOT, "octree-raymarcher_amd", "libsvo_amd.so"), "rb").read()
    assert b"SVO_TEST_FAIL_WIDE" not in shipped and b"SVO_GRID_WAVES_PER_CU" not in shipped
    assert b"SVO_TEST_FAIL_WIDE" in open(lib_of("hooks"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("name,what", VARIANTS, ids=[f"{n}-{w}" for n, w in VARIANTS])
def test_variant_against_the_oracle(name, what):
    assert os.path.exists(lib_of(name)), f"{lib_of(name)} missing: __graft_entry__.build() makes it (make -C octree-raymarcher_amd variants)"
    r = subprocess.run([sys.executable, CHECK, lib_of(name), what], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
