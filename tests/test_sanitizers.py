"""AddressSanitizer + UBSan over the CPU-side code (GPU ASan is not available on this pool): the oracle's C, and the
product's host generator / validator (terrain.cpp, world.cpp's helpers).  Built with gcc/g++ -fsanitize, run as subprocesses."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1", "-ffp-contract=off"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    subprocess.run(["gcc", "-std=gnu11", *SAN, "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "sanitize_main.c"),
                    os.path.join(ROOT, "oracle", "svo_oracle.c"), "-o", exe, "-lm", "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_host_generator_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_san")
    pkg = os.path.join(ROOT, "octree-raymarcher_amd")
    # world.cpp's validator / classifier are exercised through a tiny shim: compile the two functions' TU without HIP
    shim = str(tmp_path / "world_shim.cpp")
    src = open(os.path.join(pkg, "csrc", "world.cpp")).read()
    start = src.index("namespace svo {")
    end = src.index("static int positive_mod")
    open(shim, "w").write('#include "world.h"\n#include <cmath>\n#include <cstring>\n' + src[start:end] + "} // namespace svo\n")
    subprocess.run(["g++", "-std=c++17", *SAN, "-I", os.path.join(pkg, "csrc"), os.path.join(pkg, "host", "sanitize_host.cpp"),
                    os.path.join(pkg, "csrc", "terrain.cpp"), shim, "-o", exe, "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr
