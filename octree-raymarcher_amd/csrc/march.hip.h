// march.hip.h — device-side building blocks of the SVO march for gfx950 (wave64).
//
// Float semantics follow the reference's CPU path op for op (src/Traverse.cpp, GLM's generic
// min/max/compare).  This translation unit MUST be compiled with -ffp-contract=off and without
// fast-math: the hit voxel is decided by separately rounded IEEE operations.
#pragma once
#include <hip/hip_runtime.h>
#include "svo_format.h"
#include "../../include/svo.h"

namespace svo {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ V3 operator+(V3 a, float s) { return mk(a.x + s, a.y + s, a.z + s); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }

// glm::min / glm::max generic forms: argument order matters for NaN.
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }

// isInsideCube, src/Traverse.cpp:18-23 — closed box, NaN => false.
__device__ __forceinline__ bool inside(V3 p, V3 lo, V3 hi)
{
    return (p.x >= lo.x) & (p.y >= lo.y) & (p.z >= lo.z) & (hi.x >= p.x) & (hi.y >= p.y) & (hi.z >= p.z);
}

// cubeEscapeDistance, src/Traverse.cpp:25-32.  g = 1/b is hoisted out of the loop: the reference
// recomputes the same correctly rounded quotient on every call.
__device__ __forceinline__ float escape(V3 a, V3 g, V3 lo, V3 hi)
{
    const V3 t0 = (lo - a) * g, t1 = (hi - a) * g;
    const float tx = gmax(t0.x, t1.x), ty = gmax(t0.y, t1.y), tz = gmax(t0.z, t1.z);
    return gmin(tx, gmin(ty, tz));
}

// The GLSL twin's guard (shaders/Chunkmarch.glsl:107-114: `return d < EPS ? BIGEPS : d`); thr = -inf switches it off (NaN: kept).
constexpr float GLSL_BIGEPS = 0.0625f;
__device__ __forceinline__ float guarded(float d, float thr) { return d < thr ? GLSL_BIGEPS : d; }

// cubeEnterDistance, shaders/Chunkmarch.glsl:116-126: slabs by multiplication with g = 1/b, and the box must lie ahead.
__device__ __forceinline__ float enter_glsl(V3 a, V3 g, V3 lo, V3 hi, bool &hit)
{
    const V3 t0 = (lo - a) * g, t1 = (hi - a) * g;
    const float n0 = gmin(t0.x, t1.x), n1 = gmin(t0.y, t1.y), n2 = gmin(t0.z, t1.z);
    const float f0 = gmax(t0.x, t1.x), f1 = gmax(t0.y, t1.y), f2 = gmax(t0.z, t1.z);
    const float tnear = gmax(gmax(n0, n1), n2);
    const float tfar = gmin(gmin(f0, f1), f2);
    hit = (tfar > tnear) & (tnear > 0.0f);
    return tnear;
}

// intersectCube, src/Traverse.cpp:115-125 (true divisions, not the reciprocal).
__device__ __forceinline__ float enter(V3 a, V3 b, V3 lo, V3 hi, bool &hit)
{
    const V3 t0 = (lo - a) / b, t1 = (hi - a) / b;
    const float n0 = gmin(t0.x, t1.x), n1 = gmin(t0.y, t1.y), n2 = gmin(t0.z, t1.z);
    const float f0 = gmax(t0.x, t1.x), f1 = gmax(t0.y, t1.y), f2 = gmax(t0.z, t1.z);
    const float tnear = gmax(gmax(n0, n1), n2);
    const float tfar = gmin(gmin(f0, f1), f2);
    hit = tfar > tnear;
    return tnear;
}

__device__ __forceinline__ V3 recip(V3 b) { return mk(1.0f / b.x, 1.0f / b.y, 1.0f / b.z); }
__device__ __forceinline__ V3 normalize3(V3 v)
{   // glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
    const float d = v.x * v.x + v.y * v.y + v.z * v.z;
    return v * (1.0f / sqrtf(d));
}

// cubeNormal, shaders/Chunkmarch.glsl:128-136, with the build's EPS.
__device__ __forceinline__ V3 cube_normal(V3 s, V3 lo, V3 hi, float eps)
{
    const V3 c = (lo + hi) * 0.5f;
    const V3 p = s - c;
    const V3 dd = lo - hi;
    const V3 d = mk(fabsf(dd.x), fabsf(dd.y), fabsf(dd.z)) * 0.5f;
    const V3 n = (p / d) * (1.0f + eps);
    return normalize3(mk((float)(int)n.x, (float)(int)n.y, (float)(int)n.z));
}

// The build's entered-face normal (svo_trace_params.normal_mode = SVO_NORMAL_FACE; not in the reference): unit vector of the
// voxel face the sample point s lies closest to - axis of the largest |s - centre| (first axis on ties), signed like that
// component, against the ray when it is exactly 0.  Never NaN.
__device__ __forceinline__ V3 face_normal(V3 s, V3 lo, V3 hi, V3 beta)
{
    const V3 c = (lo + hi) * 0.5f;
    const V3 p = s - c;
    const float ax = fabsf(p.x), ay = fabsf(p.y), az = fabsf(p.z);
    int k = 0; float pk = p.x, bk = beta.x, ak = ax;
    if (ay > ak) { k = 1; pk = p.y; bk = beta.y; ak = ay; }
    if (az > ak) { k = 2; pk = p.z; bk = beta.z; ak = az; }
    const float sgn = pk > 0.0f ? 1.0f : pk < 0.0f ? -1.0f : (bk > 0.0f ? -1.0f : 1.0f);
    return mk(k == 0 ? sgn : 0.0f, k == 1 ? sgn : 0.0f, k == 2 ? sgn : 0.0f);
}

__device__ __forceinline__ int pmod(int n, int m) { return (m + (n % m)) % m; }   // src/World.cpp:276-279

// World::index(World::index_float(p)), src/World.cpp:288-293,323-332
__device__ __forceinline__ int chunk_index(const TraceArgs &A, V3 p)
{
    float qx = p.x / A.chunksize, qy = p.y / A.chunksize, qz = p.z / A.chunksize;
    if (qx < 0.0f) qx -= 1.0f;
    if (qy < 0.0f) qy -= 1.0f;
    if (qz < 0.0f) qz -= 1.0f;
    const int ix = (int)qx, iy = (int)qy, iz = (int)qz;
    return pmod(iy, A.dimh) * A.dimw * A.dimd + pmod(iz, A.dimd) * A.dimw + pmod(ix, A.dimw);
}

// The build's pinhole camera (include/svo.h svo_camera).
__device__ __forceinline__ void camera_ray(const FrameCam &c, int imgw, int imgh, int px, int py, V3 &o, V3 &d)
{
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float u = ((fx / (float)imgw) * 2.0f - 1.0f) * c.tanx;
    const float v = (1.0f - (fy / (float)imgh) * 2.0f) * c.tany;
    const V3 dir = (ld3(c.fwd) + ld3(c.right) * u) + ld3(c.up) * v;
    d = normalize3(dir);
    o = ld3(c.eye);
}

// local raster position -> image pixel (rectangles and interleaved bands share this)
__device__ __forceinline__ void local_to_pixel(const TraceArgs &A, int lx, int ly, int &px, int &py)
{
    px = A.x0 + lx;
    py = A.y0 + (ly / A.bh) * A.ystep + (ly % A.bh);
}

struct Voxel { V3 lo; float size; uint32_t material, node, cell; };

// 32-byte record as two 16-byte stores.
__device__ __forceinline__ void store_hit(void *out, int64_t k, float t, V3 n, uint32_t material, uint32_t flags,
                                          uint32_t chunk, uint32_t node, uint32_t cell)
{
    uint4 a, b;
    a.x = __float_as_uint(t); a.y = __float_as_uint(n.x); a.z = __float_as_uint(n.y); a.w = __float_as_uint(n.z);
    b.x = (material & 0xFFFFu) | (flags << 16); b.y = chunk; b.z = node; b.w = cell;
    uint4 *rec = reinterpret_cast<uint4 *>(out) + 2 * k;
    rec[0] = a;
    rec[1] = b;
}
__device__ __forceinline__ void store_miss(void *out, int64_t k, uint32_t flags)
{
    uint4 z; z.x = z.y = z.z = z.w = 0u;
    uint4 b = z; b.x = flags << 16;
    uint4 *rec = reinterpret_cast<uint4 *>(out) + 2 * k;
    rec[0] = z;
    rec[1] = b;
}
__device__ __forceinline__ void store_flags(void *out, int64_t k, uint32_t flags)
{
    reinterpret_cast<uint16_t *>(out)[16 * k + 9] = (uint16_t)flags;
}

constexpr uint32_t STEP_GUARD = 1u << 22;   // total march steps after which a ray is abandoned (SVO_ERR_FLAG)

} // namespace svo
