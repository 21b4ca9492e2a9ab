"""The C++ World/Traverse-shaped adaptor (octree-raymarcher_amd/host/svo_world.hpp) over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "octree-raymarcher_amd", "host")
EXE = os.path.join(HOST, "example_world")


def build():
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I.", "example_world.cpp", "-L..", "-lsvo_amd",
                    "-Wl,-rpath,$ORIGIN/..", "-o", "example_world"], cwd=HOST, check=True)


def test_adaptor_compiles_and_fails_loudly_without_a_device(svo):
    build()
    if svo.device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([EXE, "4"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr        # svo::Error thrown by World::load_gpu, no CPU fallback


@pytest.mark.gpu
def test_adaptor_draw_and_chunkmarch_on_gpu():
    build()
    r = subprocess.run([EXE, "6"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "cursor hit" in r.stdout
