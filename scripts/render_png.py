"""Render one shaded frame (svo_trace + svo_shade) to a PNG — a human-readable sanity check of the whole path.
    python scripts/render_png.py out.png [depth] [width] [height]
"""
import importlib, os, struct, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
svo = importlib.import_module("octree-raymarcher_amd")


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


out = sys.argv[1] if len(sys.argv) > 1 else "frame.png"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 9
w = int(sys.argv[3]) if len(sys.argv) > 3 else 960
h = int(sys.argv[4]) if len(sys.argv) > 4 else 540
W = svo.World.generate(4, 1, 4, 128, depth, build_device=0); W.upload(0)
cam = svo.make_camera((256.3, 150.0, -40.0), (0.0, -0.5, 0.866), (0.0, 1.0, 0.0), 60.0, w, h)
g = W.draw(cam, shadow=True)
P = svo.shade_defaults()
# the reference's lights sit near the origin of a 4x4x4 world; for a picture, put a stronger sun-like term in
P.directional.diffuse[:] = [0.9, 0.85, 0.7]; P.directional.ambient[:] = [0.25, 0.3, 0.4]
gb = svo.DeviceBuffer.from_numpy(g); rgba = svo.DeviceBuffer(w * h * 16)
svo.shade(cam, P, (0, 0, w, h), gb.ptr, rgba.ptr); svo.lib.svo_stream_synchronize(None)
img = rgba.to_numpy(np.float32, w * h * 4).reshape(h, w, 4)
rgb = np.nan_to_num(img[..., :3], nan=0.0)
hit = (g["flags"] & 1) != 0
sky = np.array([0.45, 0.65, 0.9], np.float32)
rgb = np.where(hit[..., None], rgb, sky)
rgb = np.clip(rgb, 0, 1) ** (1 / 2.2)
write_png(out, (rgb * 255 + 0.5).astype(np.uint8))
print(out, "hits %.1f%%" % (100 * hit.mean()), "shadowed %.1f%%" % (100 * ((g["flags"] & 4) != 0).mean()), "materials", np.unique(g["material"][hit]))
