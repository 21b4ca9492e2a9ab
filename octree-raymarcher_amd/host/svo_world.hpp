// svo_world.hpp — C++ host-side mirror of the reference's World / Traverse surface, over the C ABI.
//
// The reference is a C++ program whose hot path sits behind two in-process surfaces
// (src/World.h:44-68, src/Traverse.h:27-30).  This header gives a maintainer the same names, argument
// meaning and hit/miss behaviour, implemented by libsvo_amd.so (include/svo.h) — header-only, no HIP
// headers needed, plain float[3] instead of glm::vec3 so it compiles without GLM:
//
//   svo::World::init(w,h,d,s)        <- World::init            src/World.cpp:19-43
//   svo::World::load_gpu()           <- World::load_gpu        src/World.cpp:57-94
//   svo::World::draw(camera, ...)    <- World::draw (+ draw_shadowmap as the fused shadow ray)  src/World.cpp:162-266
//   svo::World::modify(i, ...)       <- World::modify          src/World.cpp:268-274
//   svo::World::index / index_float  <- src/World.cpp:288-293,323-332
//   svo::World::deinit()             <- World::deinit          src/World.cpp:129-151
//   svo::chunkmarch(alpha,beta,world,&sigma) <- chunkmarch     src/Traverse.cpp:127-171
//
// Errors: the reference asserts / die()s (src/Util.cpp:72-78); here every failure throws svo::Error
// carrying the svo_status and svo_last_error().  There is no CPU fallback.
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/svo.h"

namespace svo {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &where) : std::runtime_error(where + ": " + svo_last_error()), code(c) {}
};
inline void check(int rc, const char *where) { if (rc < 0) throw Error(rc, where); }

struct vec3 { float x, y, z; };
struct ivec3 { int x, y, z; };

// RAII HBM buffer of G-buffer records.
class GBuffer {
public:
    GBuffer() = default;
    GBuffer(int w, int h) { resize(w, h); }
    ~GBuffer() { svo_device_free(dev_); }
    GBuffer(const GBuffer &) = delete;
    GBuffer &operator=(const GBuffer &) = delete;
    void resize(int w, int h)
    {
        svo_device_free(dev_);
        width = w; height = h;
        dev_ = static_cast<svo_hit *>(svo_device_alloc(sizeof(svo_hit) * (size_t)w * h));
        if (!dev_) throw Error(SVO_ERR_OUT_OF_MEMORY, "GBuffer::resize");
    }
    svo_hit *device() const { return dev_; }
    std::vector<svo_hit> download() const
    {
        std::vector<svo_hit> host((size_t)width * height);
        check(svo_memcpy_d2h(host.data(), dev_, host.size() * sizeof(svo_hit)), "GBuffer::download");
        return host;
    }
    int width = 0, height = 0;
private:
    svo_hit *dev_ = nullptr;
};

struct Camera : svo_camera {
    // Camera::init-style construction (src/Camera.cpp): position, unit forward, up hint, vertical fov.
    Camera(vec3 eye_, vec3 fwd, vec3 up_hint, float vfov_deg, int w, int h)
    {
        auto norm = [](double v[3]) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
        double f[3] = { fwd.x, fwd.y, fwd.z }, u[3] = { up_hint.x, up_hint.y, up_hint.z };
        norm(f);
        double r[3] = { f[1] * u[2] - f[2] * u[1], f[2] * u[0] - f[0] * u[2], f[0] * u[1] - f[1] * u[0] };
        norm(r);
        double up2[3] = { r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0] };
        eye[0] = eye_.x; eye[1] = eye_.y; eye[2] = eye_.z;
        for (int i = 0; i < 3; ++i) { forward[i] = (float)f[i]; right[i] = (float)r[i]; up[i] = (float)up2[i]; }
        tan_half_y = (float)std::tan(vfov_deg * 3.14159265358979323846 / 360.0);
        tan_half_x = tan_half_y * (float)w / (float)h;
        width = w; height = h;
    }
};

class World {
public:
    World() = default;
    ~World() { deinit(); }
    World(const World &) = delete;
    World &operator=(const World &) = delete;

    // World::init(w, h, d, s) with the reference's terrain constants as defaults (src/World.cpp:296-321).
    void init(int w, int h, int d, int s, uint32_t tree_max_depth = 8, const int chunkcoordmin[3] = nullptr,
              const svo_terrain_params *terrain = nullptr)
    {
        deinit();
        svo_terrain_params tp;
        if (terrain) tp = *terrain;
        else {
            std::memset(&tp, 0, sizeof tp);
            tp.depth = tree_max_depth; tp.pyramid_resolution = 0; tp.amplitude = 64.0f; tp.yshift = 16.0f;
            tp.water = 1; tp.water_level = 6.0f; tp.water_material = 6;
        }
        check(svo_world_generate(w, h, d, s, chunkcoordmin, &tp, &world_), "World::init");
        width = w; height = h; depth = d; chunksize = s;
        plane = w * d; volume = plane * h;
    }
    // Adopt chunks the caller built itself (Ocroot arrays).
    void init(const std::vector<svo_chunk_desc> &chunks, int w, int h, int d, int s, const int chunkcoordmin[3] = nullptr)
    {
        deinit();
        check(svo_world_create(chunks.data(), (int)chunks.size(), w, h, d, s, chunkcoordmin, &world_), "World::init(chunks)");
        width = w; height = h; depth = d; chunksize = s;
        plane = w * d; volume = plane * h;
    }
    void deinit() { svo_world_destroy(world_); world_ = nullptr; }

    void load_gpu(int device = 0) { check(svo_world_upload(world_, device), "World::load_gpu"); }

    // Which of the reference's two marches draw() / draw_frames() follow: the CPU code's (src/Traverse.cpp; what computeTarget and the
    // edits use - the default) or the fragment shader's (shaders/Chunkmarch.glsl; what World::draw renders with).  SURVEY.md App. B.
    int semantics = SVO_SEMANTICS_CPU;

    // World::draw: march every pixel of the camera image into `out` (asynchronous on `stream`).
    // shadow = true also casts the shadow ray of draw_shadowmap's light direction from every hit.
    void draw(const Camera &cam, GBuffer &out, bool shadow = false, const float light_dir[3] = nullptr, void *stream = nullptr)
    {
        if (out.width != cam.width || out.height != cam.height) out.resize(cam.width, cam.height);
        svo_trace_params p;
        std::memset(&p, 0, sizeof p);
        p.semantics = semantics;
        p.shadow = shadow ? 1 : 0;
        if (light_dir) std::memcpy(p.light_dir, light_dir, sizeof p.light_dir);
        check(svo_trace(world_, &cam, &p, 0, 0, cam.width, cam.height, out.device(), stream), "World::draw");
    }

    // Several views in one launch (svo_trace_frames): `out` receives cams.size() rasters of cam size one after the
    // other (a GBuffer of height cams.size() * cam.height holds them).  All cameras share one image size.
    void draw_frames(const std::vector<Camera> &cams, GBuffer &out, bool shadow = false, const float light_dir[3] = nullptr, void *stream = nullptr)
    {
        if (cams.empty()) return;
        const int w = cams[0].width, h = cams[0].height;
        if (out.width != w || out.height != h * (int)cams.size()) out.resize(w, h * (int)cams.size());
        svo_trace_params p;
        std::memset(&p, 0, sizeof p);
        p.semantics = semantics;
        p.shadow = shadow ? 1 : 0;
        if (light_dir) std::memcpy(p.light_dir, light_dir, sizeof p.light_dir);
        std::vector<svo_camera> plain(cams.begin(), cams.end());
        check(svo_trace_frames(world_, plain.data(), (int)plain.size(), &p, 0, 0, w, h, out.device(), stream), "World::draw_frames");
    }

    // World::modify(i, tree delta, twig delta): re-send an edited chunk (Ocdelta ranges, src/Octree.h:47-54).
    void modify(int i, const svo_chunk_desc &edited, uint64_t tree_left, uint64_t tree_right, uint64_t twig_left, uint64_t twig_right, bool realloc_)
    {
        check(svo_world_update(world_, i, &edited, tree_left, tree_right, twig_left, twig_right, realloc_ ? 1 : 0), "World::modify");
    }

    // world.chunk[i].build / destroy / replace(cmin, cmax, ...) followed by world.modify(i, ...) (src/Main.cpp:340-367) in one
    // call, on the device the world is uploaded to (svo_world_edit_box).
    void build(int i, vec3 cmin, vec3 cmax, uint16_t material) { edit(i, SVO_EDIT_BUILD, cmin, cmax, material, "World::build"); }
    void destroy(int i, vec3 cmin, vec3 cmax) { edit(i, SVO_EDIT_DESTROY, cmin, cmax, 0, "World::destroy"); }
    void replace(int i, vec3 cmin, vec3 cmax, uint16_t material) { edit(i, SVO_EDIT_REPLACE, cmin, cmax, material, "World::replace"); }

    // World::shift(offset): slide the grid by one chunk (src/World.cpp:334-378).
    void shift(ivec3 offset)
    {
        const int o[3] = { offset.x, offset.y, offset.z };
        check(svo_world_shift(world_, o), "World::shift");
    }

    ivec3 index_float(vec3 p) const
    {
        const float pp[3] = { p.x, p.y, p.z };
        int q[3];
        check(svo_world_index_float(world_, pp, q), "World::index_float");
        return { q[0], q[1], q[2] };
    }
    int index(int x, int y, int z) const { return svo_world_index(world_, x, y, z); }
    svo_chunk_desc chunk(int i) const { svo_chunk_desc d; check(svo_world_chunk(world_, i, &d), "World::chunk"); return d; }
    svo_world *handle() const { return world_; }
private:
    void edit(int i, int op, vec3 cmin, vec3 cmax, uint16_t material, const char *where)
    {
        const float lo[3] = { cmin.x, cmin.y, cmin.z }, hi[3] = { cmax.x, cmax.y, cmax.z };
        check(svo_world_edit_box(world_, i, op, lo, hi, material), where);
    }
public:

    int width = 0, height = 0, depth = 0, plane = 0, volume = 0, chunksize = 0;
private:
    svo_world *world_ = nullptr;
};

// bool chunkmarch(vec3 alpha, vec3 beta, const World *world, vec3 *sigma) — src/Traverse.cpp:127-171.
// One ray through the device kernel (the reference uses this for the edit cursor, src/Main.cpp:314-319).
// sigma is written only on a hit, exactly like the reference; `hit_out` optionally receives the voxel record.
inline bool chunkmarch(vec3 alpha, vec3 beta, const World *world, vec3 *sigma, svo_hit *hit_out = nullptr)
{
    struct Scratch {
        float *o = nullptr, *d = nullptr; svo_hit *h = nullptr;
        Scratch() { o = (float *)svo_device_alloc(12); d = (float *)svo_device_alloc(12); h = (svo_hit *)svo_device_alloc(sizeof(svo_hit)); }
        ~Scratch() { svo_device_free(o); svo_device_free(d); svo_device_free(h); }
    };
    static thread_local Scratch s;
    const float a[3] = { alpha.x, alpha.y, alpha.z }, b[3] = { beta.x, beta.y, beta.z };
    check(svo_memcpy_h2d(s.o, a, sizeof a), "chunkmarch"); check(svo_memcpy_h2d(s.d, b, sizeof b), "chunkmarch");
    check(svo_trace_rays(world->handle(), s.o, s.d, 1, nullptr, s.h, nullptr), "chunkmarch");
    svo_hit h;
    check(svo_memcpy_d2h(&h, s.h, sizeof h), "chunkmarch");
    if (hit_out) *hit_out = h;
    if (!(h.flags & SVO_HIT_FLAG)) return false;
    if (sigma) *sigma = { alpha.x + beta.x * h.t, alpha.y + beta.y * h.t, alpha.z + beta.z * h.t };   // src/Traverse.cpp:161
    return true;
}

} // namespace svo
