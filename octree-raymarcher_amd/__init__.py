"""octree-raymarcher_amd — thin ctypes harness over libsvo_amd.so (the C ABI in include/svo.h).

The product is the shared library; this module only exists so that tests/ and bench.py can drive
it.  It mirrors the reference's `World` / `Traverse` surface (src/World.h:44-68, src/Traverse.h:27-30):

    World.generate(w, h, d, chunksize, ...)   <- World::init            src/World.cpp:19-43
    World.upload(device)                      <- World::load_gpu        src/World.cpp:57-94
    World.draw(camera, ...)                   <- World::draw            src/World.cpp:205-266
    World.chunkmarch(origins, dirs)           <- chunkmarch             src/Traverse.cpp:127-171
    World.index / index_float                 <- World::index(_float)   src/World.cpp:288-293,323-332

There is NO CPU fallback: if libsvo_amd.so is missing the import raises, and every device call
raises SvoError when HIP reports no device.

Import with importlib.import_module("octree-raymarcher_amd") (the hyphen is the project's name).
If torch is used in the same process, import torch BEFORE this module so that both share one HIP
runtime (both resolve the soname libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import numpy as np

from . import partition  # noqa: F401  (host-side image partition helpers)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVO_AMD_LIB lets kernel A/B experiments point at an alternative build of the same library.
LIB_PATH = os.environ.get("SVO_AMD_LIB") or os.path.join(_HERE, "libsvo_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()). "
        "There is no CPU fallback for the SVO march.")

lib = C.CDLL(LIB_PATH)

# ---- enums / constants (include/svo.h) -----------------------------------------------------
SVO_OK = 0
OK_LITERAL_ONLY = 1     # svo_world_update / edit_box / shift: applied, but the stack kernel's wide trees could not be rebuilt
ERR_NAMES = {0: "SVO_OK", -1: "SVO_ERR_INVALID_ARG", -2: "SVO_ERR_NO_DEVICE", -3: "SVO_ERR_OUT_OF_MEMORY",
             -4: "SVO_ERR_MALFORMED_TREE", -5: "SVO_ERR_NOT_UPLOADED", -6: "SVO_ERR_UNSUPPORTED", -7: "SVO_ERR_HIP"}
EMPTY, LEAF, BRANCH, TWIG = 0, 1, 2, 3
KERNEL_AUTO, KERNEL_LITERAL, KERNEL_STACK = 0, 1, 2
EDIT_BUILD, EDIT_DESTROY, EDIT_REPLACE = 0, 1, 2     # svo_world_edit_box
HIT_FLAG, SHADOW_TRACED, SHADOWED, FACE_NORMAL, ERR_FLAG = 1, 2, 4, 8, 1 << 15
NORMAL_CUBE, NORMAL_FACE = 0, 1
SEMANTICS_CPU, SEMANTICS_GLSL = 0, 1
CELL_NONE = 0xFF

HIT_DTYPE = np.dtype([("t", "<f4"), ("normal", "<f4", (3,)), ("material", "<u2"), ("flags", "<u2"),
                      ("chunk", "<u4"), ("node", "<u4"), ("cell", "<u4")])
assert HIT_DTYPE.itemsize == 32


class SvoError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = lib.svo_last_error().decode(errors="replace")
        super().__init__(f"{where}: {ERR_NAMES.get(code, code)} ({msg})")


class ChunkDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("size", C.c_float), ("depth", C.c_uint32), ("_pad", C.c_uint32),
                ("tree", C.POINTER(C.c_uint32)), ("trees", C.c_uint64),
                ("twig", C.POINTER(C.c_uint16)), ("twigs", C.c_uint64)]


class TerrainParams(C.Structure):
    _fields_ = [("depth", C.c_uint32), ("pyramid_resolution", C.c_uint32), ("amplitude", C.c_float),
                ("yshift", C.c_float), ("seed", C.c_int32), ("water", C.c_int32), ("water_level", C.c_float),
                ("water_material", C.c_uint32), ("threads", C.c_int32), ("coarse_depth", C.c_uint32),
                ("refine_min", C.c_float * 3), ("refine_max", C.c_float * 3), ("build_device_plus1", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("forward", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("tan_half_x", C.c_float), ("tan_half_y", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


class TraceParams(C.Structure):
    _fields_ = [("eps", C.c_float), ("max_chunk_steps", C.c_int32), ("max_tree_steps", C.c_int32),
                ("max_twig_steps", C.c_int32), ("shadow", C.c_int32), ("light_dir", C.c_float * 3),
                ("kernel", C.c_int32), ("tiles_per_wave", C.c_int32), ("counters_dev", C.c_void_p),
                ("normal_mode", C.c_int32), ("launches_in_flight", C.c_int32),
                ("tile_cost_dev", C.c_void_p), ("tile_order_dev", C.c_void_p), ("semantics", C.c_int32), ("_pad_semantics", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("shininess", C.c_float)]


class _PointLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("ambient", C.c_float * 3), ("diffuse", C.c_float * 3), ("specular", C.c_float * 3),
                ("constant", C.c_float), ("linear", C.c_float), ("quadratic", C.c_float)]


class _DirectionalLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("ambient", C.c_float * 3), ("diffuse", C.c_float * 3),
                ("specular", C.c_float * 3)]


class _Spotlight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("ambient", C.c_float * 3), ("diffuse", C.c_float * 3),
                ("specular", C.c_float * 3), ("cos_phi", C.c_float), ("cos_gamma", C.c_float), ("constant", C.c_float),
                ("linear", C.c_float), ("quadratic", C.c_float)]


class ShadeParams(C.Structure):
    _fields_ = [("point", _PointLight), ("directional", _DirectionalLight), ("spot", _Spotlight), ("materials", Material * 8),
                ("eps", C.c_float), ("gamma", C.c_float), ("near_plane", C.c_float), ("far_plane", C.c_float)]


class WorldInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32), ("chunksize", C.c_int32),
                ("chunkcoordmin", C.c_int32 * 3), ("uploaded_device", C.c_int32),
                ("total_trees", C.c_uint64), ("total_twigs", C.c_uint64),
                ("tree_pool_bytes", C.c_uint64), ("twig_pool_bytes", C.c_uint64), ("mask_pool_bytes", C.c_uint64),
                ("max_chunk_depth", C.c_int32), ("exact_geometry", C.c_int32),
                ("wide_pool_bytes", C.c_uint64), ("wide_nodes", C.c_uint64)]


# every symbol include/svo.h declares (tests check that the library exports exactly these)
MAX_FRAMES = 16                     # SVO_MAX_FRAMES

ABI_SYMBOLS = [
    "svo_world_generate", "svo_world_create", "svo_world_info_get", "svo_world_chunk", "svo_world_destroy",
    "svo_world_index_float", "svo_world_index", "svo_world_upload", "svo_world_update",
    "svo_chunk_write", "svo_chunk_read", "svo_chunk_free", "svo_world_shift", "svo_world_edit_box", "svo_shade", "svo_shade_packed", "svo_shade_defaults", "svo_gbuffer_pack", "svo_gbuffer_unpack",
    "svo_tile_order", "svo_trace", "svo_trace_rows", "svo_trace_frames", "svo_trace_rows_frames", "svo_trace_rays", "svo_trace_last_ray_count",
    "svo_device_count", "svo_device_alloc", "svo_device_free", "svo_device_cache_trim", "svo_memcpy_h2d", "svo_memcpy_d2h",
    "svo_stream_synchronize", "svo_last_error", "svo_abi_version",
]

_P = C.c_void_p
lib.svo_last_error.restype = C.c_char_p
lib.svo_abi_version.restype = C.c_int
lib.svo_world_edit_box.argtypes = [_P, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint16]
lib.svo_world_generate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(TerrainParams), C.POINTER(_P)]
lib.svo_world_create.argtypes = [C.POINTER(ChunkDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(_P)]
lib.svo_world_info_get.argtypes = [_P, C.POINTER(WorldInfo)]
lib.svo_world_chunk.argtypes = [_P, C.c_int, C.POINTER(ChunkDesc)]
lib.svo_world_destroy.argtypes = [_P]
lib.svo_world_destroy.restype = None
lib.svo_world_index_float.argtypes = [_P, C.POINTER(C.c_float), C.POINTER(C.c_int)]
lib.svo_world_index.argtypes = [_P, C.c_int, C.c_int, C.c_int]
lib.svo_chunk_write.argtypes = [C.c_char_p, C.POINTER(ChunkDesc), C.c_uint64, C.c_uint64]
lib.svo_chunk_read.argtypes = [C.c_char_p, C.POINTER(ChunkDesc), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
lib.svo_chunk_free.argtypes = [C.POINTER(ChunkDesc)]
lib.svo_chunk_free.restype = None
lib.svo_world_shift.argtypes = [_P, C.POINTER(C.c_int)]
lib.svo_gbuffer_pack.argtypes = [_P, _P, C.c_int64, _P]
lib.svo_gbuffer_unpack.argtypes = [_P, _P, C.c_int64, _P]
lib.svo_shade_defaults.argtypes = [C.POINTER(ShadeParams)]
lib.svo_shade_defaults.restype = None
lib.svo_shade.argtypes = [C.POINTER(Camera), C.POINTER(ShadeParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]
lib.svo_shade_packed.argtypes = [C.POINTER(Camera), C.POINTER(ShadeParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]
lib.svo_world_upload.argtypes = [_P, C.c_int]
lib.svo_world_update.argtypes = [_P, C.c_int, C.POINTER(ChunkDesc), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
lib.svo_trace.argtypes = [_P, C.POINTER(Camera), C.POINTER(TraceParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]
lib.svo_trace_rows.argtypes = [_P, C.POINTER(Camera), C.POINTER(TraceParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]
lib.svo_trace_frames.argtypes = [_P, C.POINTER(Camera), C.c_int, C.POINTER(TraceParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]
lib.svo_trace_rows_frames.argtypes = [_P, C.POINTER(Camera), C.c_int, C.POINTER(TraceParams), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]
lib.svo_tile_order.argtypes = [_P, _P, _P, C.c_int, _P]
lib.svo_trace_rays.argtypes = [_P, _P, _P, C.c_int64, C.POINTER(TraceParams), _P, _P]
lib.svo_trace_last_ray_count.argtypes = [_P, _P, C.POINTER(C.c_uint64)]
lib.svo_device_count.restype = C.c_int
lib.svo_device_alloc.argtypes = [C.c_size_t]
lib.svo_device_alloc.restype = _P
lib.svo_device_free.argtypes = [_P]
lib.svo_device_free.restype = None
lib.svo_device_cache_trim.argtypes = []
lib.svo_device_cache_trim.restype = None
lib.svo_memcpy_h2d.argtypes = [_P, _P, C.c_size_t]
lib.svo_memcpy_d2h.argtypes = [_P, _P, C.c_size_t]
lib.svo_stream_synchronize.argtypes = [_P]


def _check(rc: int, where: str) -> int:
    if rc < 0:
        raise SvoError(rc, where)
    return rc


def device_count() -> int:
    return lib.svo_device_count()


class DeviceBuffer:
    """A caller-owned HBM buffer (svo_device_alloc)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = lib.svo_device_alloc(self.nbytes)
        if not self.ptr:
            raise SvoError(-3, "svo_device_alloc")

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(max(a.nbytes, 1))
        if a.nbytes:
            _check(lib.svo_memcpy_h2d(buf.ptr, a.ctypes.data, a.nbytes), "svo_memcpy_h2d")
        return buf

    def to_numpy(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            _check(lib.svo_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "svo_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib.svo_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def make_camera(eye, forward, up_hint, vfov_deg: float, width: int, height: int) -> Camera:
    """Pinhole camera: orthonormal basis from (forward, up_hint); tangents from the vertical fov."""
    f = _normalize(forward)
    r = _normalize(np.cross(f, np.asarray(up_hint, dtype=np.float64)))
    u = np.cross(r, f)
    cam = Camera()
    cam.eye[:] = [float(np.float32(x)) for x in eye]
    cam.forward[:] = [float(np.float32(x)) for x in f]
    cam.right[:] = [float(np.float32(x)) for x in r]
    cam.up[:] = [float(np.float32(x)) for x in u]
    ty = math.tan(math.radians(vfov_deg) * 0.5)
    cam.tan_half_y = ty
    cam.tan_half_x = ty * width / height
    cam.width, cam.height = int(width), int(height)
    return cam


def default_camera(world_w: int, world_d: int, chunksize: int, width: int, height: int) -> Camera:
    """SURVEY.md §8d bench camera: eye (world_cx, 150, -40), forward normalize(0,-0.5,0.866), vfov 60."""
    cx = world_w * chunksize * 0.5
    return make_camera((cx, 150.0, -40.0), (0.0, -0.5, 0.866), (0.0, 1.0, 0.0), 60.0, width, height)


def c5_scene() -> dict:
    """BASELINE.json configs[4] as tests and bench.py build it: one depth-16 chunk refined to full depth only inside
    the band 62 <= x <= 66 (depth-10 bricks elsewhere; 13.4 M nodes, 2.5 M bricks), seen by a camera hovering 2 units
    over the band's terrain, so that a third of the primary hits are depth-16 voxels (14 branch levels on the path)."""
    return {
        "generate": dict(pyramid_resolution=4096, water=False, coarse_depth=10, refine_box=((62.0, -1e9, -1e9), (66.0, 1e9, 1e9))),
        "camera": lambda w, h: make_camera((64.3, 12.7, 96.5), (0.0, -0.8, 0.6), (0.0, 1.0, 0.0), 60.0, w, h),
    }


def trace_params(shadow: bool = False, kernel: int = KERNEL_AUTO, light_dir=(1.0, -1.0, 0.0), eps: float = 0.0,
                 caps=(0, 0, 0), counters_dev: Optional[int] = None, tiles_per_wave: int = 0, normal_mode: int = 0,
                 tile_cost_dev: Optional[int] = None, tile_order_dev: Optional[int] = None, launches_in_flight: int = 0,
                 semantics: int = 0) -> TraceParams:
    """semantics: SEMANTICS_CPU (src/Traverse.cpp) / SEMANTICS_GLSL (shaders/Chunkmarch.glsl); eps / caps 0 = that twin's own constants."""
    p = TraceParams()
    p.semantics = semantics
    p.normal_mode = normal_mode
    p.eps = eps
    p.max_chunk_steps, p.max_tree_steps, p.max_twig_steps = caps
    p.shadow = 1 if shadow else 0
    p.light_dir[:] = [float(x) for x in light_dir]
    p.kernel = kernel
    p.counters_dev = counters_dev
    p.tiles_per_wave = tiles_per_wave
    p.tile_cost_dev = tile_cost_dev
    p.tile_order_dev = tile_order_dev
    p.launches_in_flight = launches_in_flight
    return p


def chunk_write(path: str, chunk: dict, treestoragesize: int = 0, twigstoragesize: int = 0):
    """Ocroot::write (src/Octree.cpp:180-187): chunk = dict(position, size, depth, tree, twig)."""
    tree = np.ascontiguousarray(chunk["tree"], dtype=np.uint32)
    twig = np.ascontiguousarray(chunk["twig"], dtype=np.uint16)
    d = ChunkDesc()
    d.position[:] = [float(x) for x in chunk["position"]]
    d.size, d.depth = float(chunk["size"]), int(chunk["depth"])
    d.tree, d.trees = tree.ctypes.data_as(C.POINTER(C.c_uint32)), tree.size
    d.twig, d.twigs = twig.ctypes.data_as(C.POINTER(C.c_uint16)), twig.size // 64
    _check(lib.svo_chunk_write(path.encode(), C.byref(d), treestoragesize, twigstoragesize), "svo_chunk_write")


def chunk_read(path: str) -> dict:
    """Ocroot::read (src/Octree.cpp:189-201) -> dict(position, size, depth, tree, twig, treestoragesize, twigstoragesize)."""
    d = ChunkDesc()
    ts, ws = C.c_uint64(), C.c_uint64()
    _check(lib.svo_chunk_read(path.encode(), C.byref(d), C.byref(ts), C.byref(ws)), "svo_chunk_read")
    try:
        tree = np.ctypeslib.as_array(d.tree, shape=(d.trees,)).copy()
        twig = np.ctypeslib.as_array(d.twig, shape=(d.twigs * 64,)).copy() if d.twigs else np.zeros(0, np.uint16)
        return {"position": tuple(d.position), "size": d.size, "depth": d.depth, "tree": tree, "twig": twig,
                "treestoragesize": ts.value, "twigstoragesize": ws.value}
    finally:
        lib.svo_chunk_free(C.byref(d))


def gbuffer_pack(gbuffer_ptr: int, packed_ptr: int, n: int, stream: int = 0):
    _check(lib.svo_gbuffer_pack(gbuffer_ptr, packed_ptr, n, stream), "svo_gbuffer_pack")


def gbuffer_unpack(packed_ptr: int, gbuffer_ptr: int, n: int, stream: int = 0):
    _check(lib.svo_gbuffer_unpack(packed_ptr, gbuffer_ptr, n, stream), "svo_gbuffer_unpack")


def shade_defaults() -> ShadeParams:
    """The reference's lights (src/Main.cpp:101-131) and material table (shaders/World.Fragment.glsl:63-73)."""
    p = ShadeParams()
    lib.svo_shade_defaults(C.byref(p))
    return p


def shade(cam: Camera, params: ShadeParams, rect, gbuffer_ptr: int, rgba_ptr: int, stream: int = 0):
    x0, y0, w, h = rect
    _check(lib.svo_shade(C.byref(cam), C.byref(params), x0, y0, w, h, gbuffer_ptr, rgba_ptr, stream), "svo_shade")


def shade_packed(cam: Camera, params: ShadeParams, rect, packed_ptr: int, rgba_ptr: int, stream: int = 0):
    """svo_shade over the 8-byte records of gbuffer_pack."""
    x0, y0, w, h = rect
    _check(lib.svo_shade_packed(C.byref(cam), C.byref(params), x0, y0, w, h, packed_ptr, rgba_ptr, stream), "svo_shade_packed")


class World:
    """Host handle of a chunk grid; shaped like the reference's `World` (src/World.h:44-68)."""

    def __init__(self, handle):
        self._h = handle

    # -- construction ----------------------------------------------------------------------
    @classmethod
    def generate(cls, w: int, h: int, d: int, chunksize: int = 128, depth: int = 8, chunkcoordmin=(0, 0, 0),
                 pyramid_resolution: int = 0, amplitude: float = 64.0, yshift: float = 16.0, seed: int = 0,
                 water: bool = True, water_level: float = 6.0, water_material: int = 6, threads: int = 0,
                 coarse_depth: int = 0, refine_box=None, build_device: Optional[int] = None) -> "World":
        """coarse_depth / refine_box=((x0,y0,z0),(x1,y1,z1)): sparse refinement (full depth only inside the box)."""
        tp = TerrainParams(depth, pyramid_resolution, amplitude, yshift, seed, 1 if water else 0, water_level,
                           water_material, threads, coarse_depth)
        if refine_box is not None:
            tp.refine_min[:] = [float(v) for v in refine_box[0]]
            tp.refine_max[:] = [float(v) for v in refine_box[1]]
        tp.build_device_plus1 = 0 if build_device is None else int(build_device) + 1     # None: host threads
        ccm = (C.c_int * 3)(*chunkcoordmin)
        out = _P()
        _check(lib.svo_world_generate(w, h, d, chunksize, ccm, C.byref(tp), C.byref(out)), "svo_world_generate")
        return cls(out)

    @classmethod
    def create(cls, chunks: Sequence[dict], w: int, h: int, d: int, chunksize: int, chunkcoordmin=(0, 0, 0)) -> "World":
        """chunks: dicts with position(3), size, depth, tree (uint32 array), twig (uint16 array, 64 per brick)."""
        descs = (ChunkDesc * len(chunks))()
        keep = []
        for i, c in enumerate(chunks):
            tree = np.ascontiguousarray(c["tree"], dtype=np.uint32)
            twig = np.ascontiguousarray(c.get("twig", np.zeros(0, np.uint16)), dtype=np.uint16)
            keep += [tree, twig]
            descs[i].position[:] = [float(x) for x in c["position"]]
            descs[i].size = float(c["size"])
            descs[i].depth = int(c["depth"])
            descs[i].tree = tree.ctypes.data_as(C.POINTER(C.c_uint32))
            descs[i].trees = tree.size
            descs[i].twig = twig.ctypes.data_as(C.POINTER(C.c_uint16))
            descs[i].twigs = twig.size // 64
        ccm = (C.c_int * 3)(*chunkcoordmin)
        out = _P()
        _check(lib.svo_world_create(descs, len(chunks), w, h, d, chunksize, ccm, C.byref(out)), "svo_world_create")
        return cls(out)

    def destroy(self):
        if self._h:
            lib.svo_world_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- inspection ------------------------------------------------------------------------
    @property
    def info(self) -> WorldInfo:
        wi = WorldInfo()
        _check(lib.svo_world_info_get(self._h, C.byref(wi)), "svo_world_info_get")
        return wi

    def chunk(self, i: int, copy: bool = True) -> dict:
        """Host pools of chunk i.  copy=False borrows the library's arrays (valid until destroy/update)."""
        d = ChunkDesc()
        _check(lib.svo_world_chunk(self._h, i, C.byref(d)), "svo_world_chunk")
        tree = np.ctypeslib.as_array(d.tree, shape=(d.trees,))
        twig = np.ctypeslib.as_array(d.twig, shape=(d.twigs * 64,)) if d.twigs else np.zeros(0, np.uint16)
        if copy:
            tree, twig = tree.copy(), twig.copy()
        return {"position": tuple(d.position), "size": d.size, "depth": d.depth, "tree": tree, "twig": twig}

    def index_float(self, p) -> tuple:
        pp = (C.c_float * 3)(*[float(x) for x in p])
        q = (C.c_int * 3)()
        _check(lib.svo_world_index_float(self._h, pp, q), "svo_world_index_float")
        return tuple(q)

    def index(self, x: int, y: int, z: int) -> int:
        return _check(lib.svo_world_index(self._h, x, y, z), "svo_world_index")

    # -- device ----------------------------------------------------------------------------
    def upload(self, device: int = 0) -> "World":
        self.upload_status = _check(lib.svo_world_upload(self._h, device), "svo_world_upload")     # SVO_OK or OK_LITERAL_ONLY
        return self

    def update(self, chunk: int, desc: dict, tree_range=(0, 0), twig_range=(0, 0), realloc: bool = False):
        tree = np.ascontiguousarray(desc["tree"], dtype=np.uint32)
        twig = np.ascontiguousarray(desc["twig"], dtype=np.uint16)
        d = ChunkDesc()
        d.position[:] = [float(x) for x in desc["position"]]
        d.size, d.depth = float(desc["size"]), int(desc["depth"])
        d.tree, d.trees = tree.ctypes.data_as(C.POINTER(C.c_uint32)), tree.size
        d.twig, d.twigs = twig.ctypes.data_as(C.POINTER(C.c_uint16)), twig.size // 64
        return _check(lib.svo_world_update(self._h, chunk, C.byref(d), tree_range[0], tree_range[1], twig_range[0], twig_range[1],
                                           1 if realloc else 0), "svo_world_update")

    def edit_box(self, chunk: int, op: int, lo, hi, material: int = 0):
        """Ocroot::build / destroy / replace + World::modify on the device (svo_world_edit_box); op = EDIT_BUILD / EDIT_DESTROY / EDIT_REPLACE."""
        return _check(lib.svo_world_edit_box(self._h, int(chunk), int(op), (C.c_float * 3)(*[float(v) for v in lo]),
                                             (C.c_float * 3)(*[float(v) for v in hi]), C.c_uint16(int(material))), "svo_world_edit_box")

    def shift(self, offset):
        """World::shift (src/World.cpp:334-378): slide the grid one chunk along one axis."""
        off = (C.c_int * 3)(*[int(v) for v in offset])
        return _check(lib.svo_world_shift(self._h, off), "svo_world_shift")

    # raw launches on caller-owned device memory (bench.py passes torch tensors' data_ptr())
    def trace(self, cam: Camera, params: TraceParams, rect, out_ptr: int, stream: int = 0):
        x0, y0, w, h = rect
        _check(lib.svo_trace(self._h, C.byref(cam), C.byref(params), x0, y0, w, h, out_ptr, stream), "svo_trace")

    def trace_rows(self, cam: Camera, params: TraceParams, band0: int, band_stride: int, nbands: int, band_height: int,
                   out_ptr: int, stream: int = 0):
        _check(lib.svo_trace_rows(self._h, C.byref(cam), C.byref(params), band0, band_stride, nbands, band_height,
                                  out_ptr, stream), "svo_trace_rows")

    def trace_frames(self, cams, params: TraceParams, rect, out_ptr: int, stream: int = 0):
        """svo_trace_frames: len(cams) frames (<= MAX_FRAMES, one image size) in one launch; out holds the rasters back to back."""
        x0, y0, w, h = rect
        arr = (Camera * len(cams))(*cams)
        _check(lib.svo_trace_frames(self._h, arr, len(cams), C.byref(params), x0, y0, w, h, out_ptr, stream), "svo_trace_frames")

    def trace_rows_frames(self, cams, params: TraceParams, band0: int, band_stride: int, nbands: int, band_height: int,
                          out_ptr: int, stream: int = 0):
        arr = (Camera * len(cams))(*cams)
        _check(lib.svo_trace_rows_frames(self._h, arr, len(cams), C.byref(params), band0, band_stride, nbands, band_height,
                                         out_ptr, stream), "svo_trace_rows_frames")

    def trace_rays(self, origins_ptr: int, dirs_ptr: int, n: int, params: TraceParams, out_ptr: int, stream: int = 0):
        _check(lib.svo_trace_rays(self._h, origins_ptr, dirs_ptr, n, C.byref(params), out_ptr, stream), "svo_trace_rays")

    def tile_order(self, cost_ptr: int, order_ptr: int, ntiles: int, stream: int = 0):
        """svo_tile_order: tile indices by descending cost (of one frame) into order_ptr."""
        _check(lib.svo_tile_order(self._h, cost_ptr, order_ptr, ntiles, stream), "svo_tile_order")

    def last_ray_count(self, stream: int = 0) -> int:
        n = C.c_uint64()
        _check(lib.svo_trace_last_ray_count(self._h, stream, C.byref(n)), "svo_trace_last_ray_count")
        return n.value

    # -- convenience: World::draw / chunkmarch returning numpy -------------------------------
    def draw(self, cam: Camera, rect=None, shadow: bool = False, kernel: int = KERNEL_AUTO, counters: bool = False,
             light_dir=(1.0, -1.0, 0.0), normal_mode: int = 0, semantics: int = 0):
        """Trace a rectangle of the camera image; returns the G-buffer (HIT_DTYPE[h, w]) [+ counters]."""
        x0, y0, w, h = rect if rect is not None else (0, 0, cam.width, cam.height)
        out = DeviceBuffer(max(w * h, 1) * 32)
        cnt = DeviceBuffer(max(w * h, 1) * 16) if counters else None
        prm = trace_params(shadow=shadow, kernel=kernel, light_dir=light_dir, counters_dev=cnt.ptr if cnt else None, normal_mode=normal_mode,
                           semantics=semantics)
        self.trace(cam, prm, (x0, y0, w, h), out.ptr)
        _check(lib.svo_stream_synchronize(None), "svo_stream_synchronize")
        g = out.to_numpy(HIT_DTYPE, w * h).reshape(h, w)
        out.free()
        if counters:
            c = cnt.to_numpy(np.uint32, w * h * 4).reshape(h, w, 4)
            cnt.free()
            return g, c
        return g

    def chunkmarch(self, origins, dirs, shadow: bool = False, kernel: int = KERNEL_AUTO, counters: bool = False,
                   light_dir=(1.0, -1.0, 0.0), eps: float = 0.0, caps=(0, 0, 0), normal_mode: int = 0, semantics: int = 0):
        """chunkmarch over a ray list (src/Traverse.cpp:127-171); returns HIT_DTYPE[n] [+ counters]."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        od, dd = DeviceBuffer.from_numpy(o), DeviceBuffer.from_numpy(d)
        out = DeviceBuffer(max(n, 1) * 32)
        cnt = DeviceBuffer(max(n, 1) * 16) if counters else None
        prm = trace_params(shadow=shadow, kernel=kernel, light_dir=light_dir, eps=eps, caps=caps,
                           counters_dev=cnt.ptr if cnt else None, normal_mode=normal_mode, semantics=semantics)
        self.trace_rays(od.ptr, dd.ptr, n, prm, out.ptr)
        _check(lib.svo_stream_synchronize(None), "svo_stream_synchronize")
        g = out.to_numpy(HIT_DTYPE, n)
        for b in (od, dd, out):
            b.free()
        if counters:
            c = cnt.to_numpy(np.uint32, n * 4).reshape(n, 4)
            cnt.free()
            return g, c
        return g
