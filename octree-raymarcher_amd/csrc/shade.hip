// shade.hip — the shading stage over the G-buffer: Blinn-Phong x 3 lights,
// shaders/World.Fragment.glsl:63-138,180-197, as one coalesced kernel (32 B read + 16 B written per pixel:
// HBM-bound).  Albedo from the material table instead of the (unavailable) texture atlas — see include/svo.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

#include "march.hip.h"
#include "world.h"

namespace svo {
namespace {

struct ShadeArgs {
    svo_shade_params P;
    float eye[3], fwd[3], right[3], up[3];
    float tanx, tany;
    int32_t imgw, imgh, x0, y0, w, h;
    const uint4 *gbuffer;
    float4 *rgba;
    // per-launch constants worked out once on the host with the same float expressions the per-pixel code used:
    float gdiffuse[8][3], gspecular[8][3];      // pow(material.diffuse / .specular, gamma), :183-184
    float dir_l[3], spot_axis[3];               // normalize(-directional.direction), normalize(-spot.direction)
    float inv_imgw, inv_imgh, inv_spot_delta, inv_near, inv_depth_range;    // 1/width, 1/height, 1/(cos_phi - cos_gamma), 1/near, 1/(1/far - 1/near)
};

__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// The shading stage is compared with the oracle to 2e-5 relative (tests/test_shading.py), not bit for bit (powf already
// differs between glibc and the device), so reciprocals and inverse square roots are the hardware's 1-ulp instructions
// instead of IEEE division sequences: ~20 VALU less per normalisation, seven normalisations per pixel.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ V3 normalize_fast(V3 v) { return v * __builtin_amdgcn_rsqf(dot3(v, v)); }
__device__ __forceinline__ float maxf0(float x) { return (x < 0.0f) ? 0.0f : x; }          // max(x, 0.0)
__device__ __forceinline__ float attenuation(float kc, float kl, float kq, float d) { return rcp_fast(kc + kl * d + kq * d * d); }   // :75-78
// pow(x, y) for x in [0, 1], y >= 0 as exp2(y * log2(x)).  The exponent y * log2(x) must be good to ~1e-5 absolute
// wherever the result is not negligible, i.e. for |y * log2(x)| < 20: near x = 1 (the only place a shininess of 10000
// leaves anything) log2(x) comes from the series of ln(1 - d), d = 1 - x exact, relative error ~1e-7; elsewhere from the
// hardware's 1-ulp log2, whose absolute error (~1e-7) times y stays below 1e-5 for y <= 100 and is irrelevant above
// (the result underflows).  No library powf: three of them were a third of the kernel.
__device__ __forceinline__ float pow_shiny(float x, float y)
{
    if (y == 0.0f) return 1.0f;                                     // pow(x, 0) = 1, also for x = 0
    if (!(x > 0.0f)) return 0.0f;
    const float d = 1.0f - x;
    const float series = -(d + d * d * (0.5f + d * (0.33333334f + d * 0.25f))) * 1.44269504f;       // log2(1 - d), d < 1/64
    const float l2 = d < 0.015625f ? series : __builtin_amdgcn_logf(x);
    return __builtin_amdgcn_exp2f(y * l2);
}
// normalize(ivec3 in {-1,0,1}^3) from the packed record's 2-bit-per-axis code (bit 6: NaN), as k_gbuffer_unpack
__device__ __forceinline__ V3 normal_from_code(uint32_t code)
{
    if (code & (1u << 6)) { const float q = __uint_as_float(0x7FC00000u); return mk(q, q, q); }
    const float ix = (float)((int)(code & 3u) - 1), iy = (float)((int)((code >> 2) & 3u) - 1), iz = (float)((int)((code >> 4) & 3u) - 1);
    const float dot = ix * ix + iy * iy + iz * iz;
    const float inv = dot == 1.0f ? 1.0f : dot == 2.0f ? __uint_as_float(0x3F3504F3u) : dot == 3.0f ? __uint_as_float(0x3F13CD3Au) : __uint_as_float(0x7FC00000u);
    return mk(ix * inv, iy * inv, iz * inv);
}

// PACKED: the G-buffer is the 8-byte form of svo_gbuffer_pack (8 B read + 16 B written per pixel instead of 32 + 16)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_shade(ShadeArgs A)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (int64_t)A.w * A.h) return;
    uint32_t flags, material;
    float t;
    V3 n;
    if (PACKED) {
        const uint2 r = reinterpret_cast<const uint2 *>(A.gbuffer)[k];
        t = __uint_as_float(r.x); material = r.y & 0xFFFFu; flags = (r.y >> 16) & 0xFFu;
        n = normal_from_code((r.y >> 24) & 0x7Fu);
    } else {
        const uint4 r0 = A.gbuffer[2 * k], r1 = A.gbuffer[2 * k + 1];
        flags = r1.x >> 16; material = r1.x & 0xFFFFu;
        t = __uint_as_float(r0.x);
        n = mk(__uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
    }
    if (!(flags & SVO_HIT_FLAG)) { A.rgba[k] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); return; }       // discard
    const svo_shade_params &P = A.P;
    // the ray of this pixel (same generation as the march) and the shaded point alpha + beta * (sigma - EPS), :174
    const int px = A.x0 + (int)(k % A.w), py = A.y0 + (int)(k / A.w);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float u = ((fx * A.inv_imgw) * 2.0f - 1.0f) * A.tanx;
    const float v = (1.0f - (fy * A.inv_imgh) * 2.0f) * A.tany;
    const V3 eye = ld3(A.eye);
    const V3 beta = normalize_fast((ld3(A.fwd) + ld3(A.right) * u) + ld3(A.up) * v);
    const float sdist = t - P.eps;
    const V3 p = eye + beta * sdist;
    const int mi = material < 8 ? (int)material : 0;
    const float shininess = P.materials[mi].shininess;
    const V3 diffuse = ld3(A.gdiffuse[mi]), specular = ld3(A.gspecular[mi]);                          // :183-184
    const float lit = (flags & SVO_SHADOWED) ? 0.0f : 1.0f;                                          // (1.0 - shadow)
    // normalize(eye - p) = -beta and |p - eye| = sigma - EPS (beta is a unit vector) while the hit lies in front of the eye
    const bool front = sdist > 1.0e-3f;
    const V3 vdir = front ? mk(-beta.x, -beta.y, -beta.z) : normalize_fast(eye - p);
    const float zdist = front ? sdist : sqrtf(dot3(p - eye, p - eye));
    V3 color = mk(0.0f, 0.0f, 0.0f);
    {   // computePointLight_BlinnPhong, :80-97
        const V3 lv = ld3(P.point.position) - p;
        const float l2 = dot3(lv, lv), il = __builtin_amdgcn_rsqf(l2);
        const V3 l = lv * il;
        const V3 hv = normalize_fast(l + vdir);
        const float d = maxf0(dot3(n, l));
        const float s = pow_shiny(maxf0(dot3(vdir, hv)), shininess);
        const float att = attenuation(P.point.constant, P.point.linear, P.point.quadratic, l2 * il);     // |p - position| = l2 / sqrt(l2)
        const V3 amb = ld3(P.point.ambient) * diffuse;
        const V3 dif = ((ld3(P.point.diffuse) * d) * diffuse) * lit;
        const V3 spe = ((ld3(P.point.specular) * s) * specular) * lit;
        color = color + ((amb + dif) + spe) * att;
    }
    {   // computeDirectionalLight_BlinnPhong, :99-114
        const V3 l = ld3(A.dir_l);
        const V3 hv = normalize_fast(l + vdir);
        const float d = maxf0(dot3(n, l));
        const float s = pow_shiny(maxf0(dot3(vdir, hv)), shininess);
        const V3 amb = ld3(P.directional.ambient) * diffuse;
        const V3 dif = ((ld3(P.directional.diffuse) * d) * diffuse) * lit;
        const V3 spe = ((ld3(P.directional.specular) * s) * specular) * lit;
        color = color + ((amb + dif) + spe);
    }
    {   // computeSpotlight_BlinnPhong, :116-138
        const V3 lv = ld3(P.spot.position) - p;
        const float l2 = dot3(lv, lv), il = __builtin_amdgcn_rsqf(l2);
        const V3 l = lv * il;
        const V3 hv = normalize_fast(l + vdir);
        const float d = maxf0(dot3(n, l));
        const float s = pow_shiny(maxf0(dot3(vdir, hv)), shininess);
        const float att = attenuation(P.spot.constant, P.spot.linear, P.spot.quadratic, l2 * il);
        const float theta = dot3(l, ld3(A.spot_axis));
        float intensity = (theta - P.spot.cos_gamma) * A.inv_spot_delta;
        intensity = (intensity < 0.0f) ? 0.0f : intensity;                      // clamp = min(max(x, 0), 1)
        intensity = (1.0f < intensity) ? 1.0f : intensity;
        const V3 amb = ld3(P.spot.ambient) * diffuse;
        const V3 dif = ((ld3(P.spot.diffuse) * d) * diffuse) * lit;
        const V3 spe = ((ld3(P.spot.specular) * s) * specular) * lit;
        color = color + (amb + (dif + spe) * intensity) * att;
    }
    A.rgba[k] = make_float4(color.x, color.y, color.z, (rcp_fast(zdist) - A.inv_near) * A.inv_depth_range);     // :193-197
}

// ---- packed G-buffer -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t axis_code(float v) { return v < 0.0f ? 0u : (v > 0.0f ? 2u : 1u); }

__global__ __launch_bounds__(256) void k_gbuffer_pack(const uint4 *in, uint2 *out, int64_t n)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint4 r0 = in[2 * k], r1 = in[2 * k + 1];
    const float nx = __uint_as_float(r0.y), ny = __uint_as_float(r0.z), nz = __uint_as_float(r0.w);
    const bool nan = (nx != nx) | (ny != ny) | (nz != nz);
    const uint32_t code = nan ? (1u << 6) : (axis_code(nx) | (axis_code(ny) << 2) | (axis_code(nz) << 4));
    uint2 o;
    o.x = r0.x;
    const uint32_t flags = r1.x >> 16;
    o.y = (r1.x & 0xFFFFu) | ((flags & 0xFFu) << 16) | (code << 24) | ((flags & SVO_ERR_FLAG) ? 1u << 31 : 0u);
    out[k] = o;
}

__global__ __launch_bounds__(256) void k_gbuffer_unpack(const uint2 *in, uint4 *out, int64_t n)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint2 p = in[k];
    const uint32_t code = (p.y >> 24) & 0x7Fu, flags = ((p.y >> 16) & 0xFFu) | ((p.y >> 31) ? (uint32_t)SVO_ERR_FLAG : 0u);
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    if (flags & SVO_HIT_FLAG) {
        if (code & (1u << 6)) {
            nx = ny = nz = __uint_as_float(0x7FC00000u);
        } else {
            const float ix = (float)((int)(code & 3u) - 1), iy = (float)((int)((code >> 2) & 3u) - 1), iz = (float)((int)((code >> 4) & 3u) - 1);
            const float dot = ix * ix + iy * iy + iz * iz;       // normalize(ivec3): the same constants as the march kernels
            const float inv = dot == 1.0f ? 1.0f : dot == 2.0f ? __uint_as_float(0x3F3504F3u) : dot == 3.0f ? __uint_as_float(0x3F13CD3Au) : __uint_as_float(0x7FC00000u);
            nx = ix * inv; ny = iy * inv; nz = iz * inv;
        }
    }
    uint4 a, b;
    a.x = p.x; a.y = __float_as_uint(nx); a.z = __float_as_uint(ny); a.w = __float_as_uint(nz);
    b.x = (p.y & 0xFFFFu) | (flags << 16); b.y = 0u; b.z = 0u; b.w = 0u;
    out[2 * k] = a;
    out[2 * k + 1] = b;
}

} // namespace
} // namespace svo

using namespace svo;

extern "C" {

void svo_shade_defaults(svo_shade_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    auto set3 = [](float *d, float x, float y, float z) { d[0] = x; d[1] = y; d[2] = z; };
    // src/Main.cpp:101-109
    set3(p->point.position, 50, 8, 65); set3(p->point.ambient, 0.1f, 0.1f, 0.1f); set3(p->point.diffuse, 0.5f, 0.5f, 0.5f); set3(p->point.specular, 1, 1, 1);
    p->point.constant = 1.0f; p->point.linear = 0.14f; p->point.quadratic = 0.09f;
    // :114-119
    const float inv = 1.0f / std::sqrt(2.0f);
    set3(p->directional.position, 250, 125, 250); set3(p->directional.direction, inv, -inv, 0.0f);
    set3(p->directional.ambient, 0.2f, 0.3f, 0.4f); set3(p->directional.diffuse, 0.3f, 0.3f, 0.6f); set3(p->directional.specular, 0, 0, 0);
    // :121-131
    const float sl = 1.0f / std::sqrt(0.01f + 1.0f + 0.01f);
    set3(p->spot.position, 50, 20, 70); set3(p->spot.direction, -0.1f * sl, -1.0f * sl, -0.1f * sl);
    set3(p->spot.ambient, 0.2f, 0.8f, 0.3f); set3(p->spot.diffuse, 0.2f, 0.8f, 0.3f); set3(p->spot.specular, 1, 1, 1);
    p->spot.cos_phi = (float)std::cos(25.0 * 3.14159265358979323846 / 180.0);
    p->spot.cos_gamma = (float)std::cos(35.0 * 3.14159265358979323846 / 180.0);
    p->spot.constant = 1.0f; p->spot.linear = 0.045f; p->spot.quadratic = 0.0075f;
    // ML[8], shaders/World.Fragment.glsl:63-73
    const float ml[8][10] = {
        { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, { .8f, .8f, .8f, .8f, .8f, .8f, .5f, .5f, .5f, 8 }, { .8f, .8f, .8f, .6f, .6f, .6f, .1f, .1f, .1f, 16 },
        { .8f, .8f, .8f, .7f, .7f, .7f, .15f, .15f, .15f, 32 }, { .8f, .8f, .8f, .9f, .9f, .9f, .7f, .7f, .7f, 10000 },
        { .8f, .8f, .8f, .5f, .5f, .5f, 0, 0, 0, 0 }, { .8f, .8f, .8f, .4f, .4f, .4f, 1, 1, 1, 100 }, { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 } };
    for (int i = 0; i < 8; ++i) {
        std::memcpy(p->materials[i].ambient, &ml[i][0], 12); std::memcpy(p->materials[i].diffuse, &ml[i][3], 12);
        std::memcpy(p->materials[i].specular, &ml[i][6], 12); p->materials[i].shininess = ml[i][9];
    }
    p->eps = 1.0f / 8192.0f; p->gamma = 2.2f; p->near_plane = 0.125f; p->far_plane = 8192.0f;
}

static int pack_common(const void *in, void *out, int64_t n, void *stream, bool pack)
{
    if (n < 0 || (n > 0 && (!in || !out))) { set_error("svo_gbuffer_pack/unpack: bad argument"); return SVO_ERR_INVALID_ARG; }
    if (n == 0) return SVO_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (pack) hipLaunchKernelGGL(k_gbuffer_pack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)in, (uint2 *)out, n);
    else hipLaunchKernelGGL(k_gbuffer_unpack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint2 *)in, (uint4 *)out, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("svo_gbuffer_pack/unpack: ") + hipGetErrorString(e)); return e == hipErrorNoDevice ? SVO_ERR_NO_DEVICE : SVO_ERR_HIP; }
    return SVO_OK;
}

int svo_gbuffer_pack(const svo_hit *gbuffer_dev, uint64_t *packed_dev, int64_t n, void *stream) { return pack_common(gbuffer_dev, packed_dev, n, stream, true); }
int svo_gbuffer_unpack(const uint64_t *packed_dev, svo_hit *gbuffer_dev, int64_t n, void *stream) { return pack_common(packed_dev, gbuffer_dev, n, stream, false); }

static int shade_impl(const svo_camera *cam, const svo_shade_params *p, int x0, int y0, int w, int h,
                      const void *gbuffer_dev, float *rgba_dev, void *stream, bool packed)
{
    if (!cam || !p || !gbuffer_dev || !rgba_dev || w < 0 || h < 0 || x0 < 0 || y0 < 0 || cam->width <= 0 || cam->height <= 0) {
        set_error("svo_shade: bad argument"); return SVO_ERR_INVALID_ARG;
    }
    ShadeArgs A;
    A.P = *p;
    if (A.P.eps == 0.0f) A.P.eps = 1.0f / 8192.0f;
    if (A.P.gamma == 0.0f) A.P.gamma = 2.2f;
    if (A.P.near_plane == 0.0f) A.P.near_plane = 0.125f;
    if (A.P.far_plane == 0.0f) A.P.far_plane = 8192.0f;
    std::memcpy(A.eye, cam->eye, 12); std::memcpy(A.fwd, cam->forward, 12); std::memcpy(A.right, cam->right, 12); std::memcpy(A.up, cam->up, 12);
    A.tanx = cam->tan_half_x; A.tany = cam->tan_half_y; A.imgw = cam->width; A.imgh = cam->height;
    A.x0 = x0; A.y0 = y0; A.w = w; A.h = h;
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 3; ++c) {
            A.gdiffuse[m][c] = std::pow(A.P.materials[m].diffuse[c], A.P.gamma);
            A.gspecular[m][c] = std::pow(A.P.materials[m].specular[c], A.P.gamma);
        }
    auto unit_neg = [](const float v[3], float out[3]) {          // glm::normalize(-v) = -v * (1 / sqrt(dot))
        const float x = -v[0], y = -v[1], z = -v[2];
        const float inv = 1.0f / std::sqrt(x * x + y * y + z * z);
        out[0] = x * inv; out[1] = y * inv; out[2] = z * inv;
    };
    unit_neg(A.P.directional.direction, A.dir_l);
    unit_neg(A.P.spot.direction, A.spot_axis);
    A.inv_imgw = 1.0f / (float)cam->width; A.inv_imgh = 1.0f / (float)cam->height;
    A.inv_spot_delta = 1.0f / (A.P.spot.cos_phi - A.P.spot.cos_gamma);
    A.inv_near = 1.0f / A.P.near_plane;
    A.inv_depth_range = 1.0f / (1.0f / A.P.far_plane - 1.0f / A.P.near_plane);
    A.gbuffer = reinterpret_cast<const uint4 *>(gbuffer_dev);
    A.rgba = reinterpret_cast<float4 *>(rgba_dev);
    const int64_t n = (int64_t)w * h;
    if (n == 0) return SVO_OK;
    if (packed) hipLaunchKernelGGL(k_shade<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_shade<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("svo_shade: ") + hipGetErrorString(e)); return e == hipErrorNoDevice ? SVO_ERR_NO_DEVICE : SVO_ERR_HIP; }
    return SVO_OK;
}

int svo_shade(const svo_camera *cam, const svo_shade_params *p, int x0, int y0, int w, int h,
              const svo_hit *gbuffer_dev, float *rgba_dev, void *stream)
{
    return shade_impl(cam, p, x0, y0, w, h, gbuffer_dev, rgba_dev, stream, false);
}

int svo_shade_packed(const svo_camera *cam, const svo_shade_params *p, int x0, int y0, int w, int h,
                     const uint64_t *packed_dev, float *rgba_dev, void *stream)
{
    return shade_impl(cam, p, x0, y0, w, h, packed_dev, rgba_dev, stream, true);
}

} // extern "C"
