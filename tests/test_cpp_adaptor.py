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


MULTI = os.path.join(HOST, "multi_gpu_example")


def build_multi():
    """The C++ multi-GPU host (multi_gpu.hpp: N devices in one process, svo_trace_rows_frames + svo_gbuffer_pack per device, one
    grouped ncclSend / ncclRecv of the bands into their rows of the frame on device 0) links against librccl."""
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I.", "multi_gpu_example.cpp", "-L..", "-lsvo_amd",
                    "-lrccl", "-Wl,-rpath,$ORIGIN/..", "-o", "multi_gpu_example"], cwd=HOST, check=True)


def test_multi_gpu_host_compiles_links_rccl_and_refuses_missing_devices(svo):
    build_multi()
    r = subprocess.run([MULTI, "64"], capture_output=True, text=True)          # no node has 64 devices
    assert r.returncode == 3 and "devices asked for" in r.stderr
    ldd = subprocess.run(["ldd", MULTI], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "libsvo_amd" in ldd


@pytest.mark.gpu
def test_multi_gpu_host_frame_equals_single_device_frame():
    """With every device present on the box (1 here; the driver's 8-GPU node runs the same binary with 8): the gathered,
    de-interleaved packed frame equals the frame device 0 traces alone, record for record."""
    import ctypes
    build_multi()
    hip = ctypes.CDLL("libamdhip64.so.7")
    n = ctypes.c_int()
    assert hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
    for ndev in sorted({1, min(2, n.value), n.value}):
        r = subprocess.run([MULTI, str(ndev), "8"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "0 records differ" in r.stdout
