"""Shading stage (svo_shade): Blinn-Phong x 3 lights over the G-buffer, against the oracle's restatement
of shaders/World.Fragment.glsl:75-138,180-197.  Float tolerance: powf / sqrtf-division chains differ in the last bits
between glibc and the device, so colours are compared to 2e-5 relative (+1e-6 absolute); depth likewise."""
import ctypes as C

import numpy as np
import pytest

RTOL, ATOL = 2e-5, 1e-6


def test_struct_layout(svo, oracle):
    assert C.sizeof(svo.ShadeParams) == 4 * (15 + 15 + 20 + 8 * 10 + 4)


def test_oracle_shading_known_answer(svo, oracle):
    """One pixel worked by hand: only the directional light on, light straight down, surface normal +y, stone (diffuse .8),
    not shadowed: colour = ambient*albedo + diffuse_light*1*albedo with albedo = 0.8^2.2; shadowed: ambient term only."""
    P = svo.shade_defaults()
    for light in (P.point, P.spot):
        for f in ("ambient", "diffuse", "specular"):
            getattr(light, f)[:] = [0.0, 0.0, 0.0]
    P.directional.direction[:] = [0.0, -1.0, 0.0]
    cam = svo.make_camera((0.0, 10.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0), 60.0, 1, 1)
    g = np.zeros(1, oracle.HIT_DTYPE)
    g["t"], g["normal"], g["material"], g["flags"] = 5.0, (0.0, 1.0, 0.0), 1, 1
    albedo = np.float32(0.8) ** np.float32(2.2)
    out = oracle.shade_image(cam, P, (0, 0, 1, 1), g)[0, 0]
    want = (np.array([0.2, 0.3, 0.4], np.float32) + np.array([0.3, 0.3, 0.6], np.float32)) * albedo
    assert np.allclose(out[:3], want, rtol=1e-5)
    dist = 5.0 - 1.0 / 8192.0
    assert abs(out[3] - (1 / dist - 8.0) / (1 / 8192.0 - 8.0)) < 1e-6
    g["flags"] = 1 | 2 | 4
    out = oracle.shade_image(cam, P, (0, 0, 1, 1), g)[0, 0]
    assert np.allclose(out[:3], np.array([0.2, 0.3, 0.4], np.float32) * albedo, rtol=1e-5)
    g["flags"] = 0
    assert tuple(oracle.shade_image(cam, P, (0, 0, 1, 1), g)[0, 0]) == (0.0, 0.0, 0.0, 1.0)


@pytest.mark.gpu
def test_gpu_shading_matches_oracle(svo, oracle):
    W = svo.World.generate(1, 1, 1, 128, 8)
    W.upload(0)
    cam = svo.make_camera((60.0, 60.0, -30.0), (0.0, -0.4, 0.9), (0.0, 1.0, 0.0), 70.0, 320, 200)   # close to the reference's lights
    g = W.draw(cam, shadow=True)
    P = svo.shade_defaults()
    gb = svo.DeviceBuffer.from_numpy(g)
    out = svo.DeviceBuffer(320 * 200 * 16)
    svo.shade(cam, P, (0, 0, 320, 200), gb.ptr, out.ptr)
    svo.lib.svo_stream_synchronize(None)
    got = out.to_numpy(np.float32, 320 * 200 * 4).reshape(200, 320, 4)
    want = oracle.shade_image(cam, P, (0, 0, 320, 200), g)
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.all(both_nan | (np.abs(got - want) <= ATOL + RTOL * np.abs(want)))
    hit = (g["flags"] & 1) != 0
    assert hit.mean() > 0.2 and np.nanmax(want[..., :3]) > 0.05          # something is actually lit
    assert np.all(got[~hit] == np.array([0, 0, 0, 1], np.float32))
    # a sub-rectangle shades identically (pixel coordinates drive the ray)
    sub = np.ascontiguousarray(g[50:150, 100:260])
    gb2 = svo.DeviceBuffer.from_numpy(sub); out2 = svo.DeviceBuffer(100 * 160 * 16)
    svo.shade(cam, P, (100, 50, 160, 100), gb2.ptr, out2.ptr)
    svo.lib.svo_stream_synchronize(None)
    got2 = out2.to_numpy(np.float32, 100 * 160 * 4).reshape(100, 160, 4)
    assert np.array_equal(got2.view(np.uint32), got[50:150, 100:260].view(np.uint32))
    # shading the packed 8-byte records gives the same picture (the record keeps t, normal, material, flags)
    pk = svo.DeviceBuffer(320 * 200 * 8); out3 = svo.DeviceBuffer(320 * 200 * 16)
    svo.gbuffer_pack(gb.ptr, pk.ptr, 320 * 200)
    svo.shade_packed(cam, P, (0, 0, 320, 200), pk.ptr, out3.ptr)
    svo.lib.svo_stream_synchronize(None)
    got3 = out3.to_numpy(np.float32, 320 * 200 * 4).reshape(200, 320, 4)
    same = (got3.view(np.uint32) == got.view(np.uint32)) | (np.isnan(got3) & np.isnan(got))
    assert np.all(same)
    W.destroy()


@pytest.mark.gpu
def test_depth12_frame_has_no_nan_pixels_with_face_normals(svo, oracle):
    """The reference's cubeNormal is NaN for ~14 % of the hits at depth 11-12 (black speckle in the shaded frame);
    with svo_trace_params.normal_mode = SVO_NORMAL_FACE every hit pixel shades, and equals the oracle's shading."""
    W = svo.World.generate(1, 1, 1, 128, 12)
    W.upload(0)
    cam = svo.make_camera((64.0, 100.0, -40.0), (0.0, -0.5, 0.866), (0.0, 1.0, 0.0), 60.0, 640, 360)
    P = svo.shade_defaults()
    pictures = {}
    for mode in (svo.NORMAL_CUBE, svo.NORMAL_FACE):
        g = W.draw(cam, shadow=True, normal_mode=mode)
        gb = svo.DeviceBuffer.from_numpy(g); out = svo.DeviceBuffer(640 * 360 * 16)
        svo.shade(cam, P, (0, 0, 640, 360), gb.ptr, out.ptr)
        svo.lib.svo_stream_synchronize(None)
        pictures[mode] = (g, out.to_numpy(np.float32, 640 * 360 * 4).reshape(360, 640, 4))
    g0, rgb0 = pictures[svo.NORMAL_CUBE]
    g1, rgb1 = pictures[svo.NORMAL_FACE]
    hit = (g1["flags"] & 1) != 0
    assert hit.mean() > 0.2
    assert np.isnan(rgb0[hit]).any(axis=1).mean() > 0.05            # the reference formula: speckle
    assert not np.isnan(rgb1).any()                                 # the face normal: none
    want = oracle.shade_image(cam, P, (0, 0, 640, 360), g1)
    assert np.all(np.abs(rgb1 - want) <= ATOL + RTOL * np.abs(want))
    W.destroy()
