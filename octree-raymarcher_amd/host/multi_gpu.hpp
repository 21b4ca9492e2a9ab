// multi_gpu.hpp — C++ host side of the multi-GPU frame (BASELINE north_star: "host code stays C++ ...
// the image tile-partitions across the 8 GPUs of one node with an RCCL gather of per-tile G-buffers over xGMI").
//
// One process drives N devices.  Every device holds the whole world (rays are independent, the world is read-only): the
// replicas are generated CONCURRENTLY, each on its own device (svo_terrain_params.build_device_plus1: noise, mips and grow()
// as kernels, pools left in HBM; the generator is deterministic, so the replicas are identical).  The image is cut into 8-row
// bands dealt round-robin (device r traces bands r, r+N, ...: svo_trace_rows_frames), each device packs its bands to the
// lossless 8-byte record (svo_gbuffer_pack), and the exchange of one draw_frames() call is
//
//     N-1 point-to-point operations: ONE ncclSend per peer of ALL its packed bands of ALL the call's frames (contiguous in
//     the peer's buffer) and the matching ncclRecv into a staging buffer on device 0, in one group;
//     F*N strided device-to-device copies on device 0 (hipMemcpy2DAsync: a band is one "row" of 8*width records, the
//     destination pitch is N bands) that put every band at its rows of its frame - no kernel, no per-band message.
//
// (Round 2 sent one message per band: 238 point-to-point operations of 245 KB per 2160p frame on 8 devices; now 7 of 8.3 MB,
// and 7 per call whatever the number of frames.)  xGMI is point-to-point: device 0 receives from its N-1 peers over N-1
// different links at once, per-link bound (8.3 MB / 153 GB/s = 0.05 ms per 2160p frame).
// Calls alternate between SLOTS sets of buffers and streams, so the exchange and de-interleave of one call overlap the
// march of the next (frames in flight, as bench.py does it).
//
// The N > 1 path is UNVERIFIED on hardware: the GPU boxes this was developed on have one device; with one device the
// exchange degenerates to the strided copies (tests/test_cpp_adaptor.py runs that).  Its index arithmetic (band_layout.hpp)
// is checked on a CPU for N = 2, 4, 8 with memcpy standing in for the copies (tests/test_host_units.py).  multi_gpu_example checks the
// gathered frames against single-device frames, record for record, on however many devices it is given.
//
// The reference has no counterpart (one GL context, one GPU: src/Main.cpp); this is the build's extension of
// World::draw to a node, in the reference's host language.  Needs <hip/hip_runtime_api.h> and <rccl/rccl.h>; the march
// itself is behind the C ABI (include/svo.h), nothing here touches a kernel.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "band_layout.hpp"
#include "svo_world.hpp"

namespace svo {

inline void hip_check(hipError_t e, const char *where) { if (e != hipSuccess) throw std::runtime_error(std::string(where) + ": " + hipGetErrorString(e)); }
inline void nccl_check(ncclResult_t r, const char *where) { if (r != ncclSuccess) throw std::runtime_error(std::string(where) + ": " + ncclGetErrorString(r)); }

class MultiGpuWorld {
public:
    static constexpr int SLOTS = 2;                      // draw_frames() calls in flight

    // World::init on every device at once + load_gpu (the device-resident generator leaves the world uploaded).
    MultiGpuWorld(int ndev, int w, int h, int d, int chunksize, uint32_t depth) : n_(ndev)
    {
        int have = 0;
        hip_check(hipGetDeviceCount(&have), "hipGetDeviceCount");
        if (ndev < 1 || ndev > have) throw std::runtime_error("MultiGpuWorld: " + std::to_string(ndev) + " devices asked for, " + std::to_string(have) + " present");
        worlds_.resize((size_t)ndev);
        std::vector<std::string> failed((size_t)ndev);
        std::vector<std::thread> builders;
        for (int r = 0; r < ndev; ++r) {
            worlds_[(size_t)r] = std::make_unique<World>();
            builders.emplace_back([this, r, w, h, d, chunksize, depth, &failed]() {
                try {
                    svo_terrain_params tp;
                    std::memset(&tp, 0, sizeof tp);
                    tp.depth = depth; tp.amplitude = 64.0f; tp.yshift = 16.0f;
                    tp.water = 1; tp.water_level = 6.0f; tp.water_material = 6;          // src/World.cpp:296-321
                    tp.build_device_plus1 = r + 1;
                    worlds_[(size_t)r]->init(w, h, d, chunksize, depth, nullptr, &tp);
                    worlds_[(size_t)r]->load_gpu(r);                                       // (already resident: returns at once)
                } catch (const std::exception &e) { failed[(size_t)r] = e.what(); }
            });
        }
        for (std::thread &t : builders) t.join();
        for (int r = 0; r < ndev; ++r) if (!failed[(size_t)r].empty()) throw std::runtime_error("MultiGpuWorld: device " + std::to_string(r) + ": " + failed[(size_t)r]);
        for (int s = 0; s < SLOTS; ++s) {
            slot_[s].streams.assign((size_t)ndev, nullptr);
            slot_[s].bands.assign((size_t)ndev, nullptr); slot_[s].packed.assign((size_t)ndev, nullptr); slot_[s].staging.assign((size_t)ndev, nullptr);
            for (int r = 0; r < ndev; ++r) {
                hip_check(hipSetDevice(r), "hipSetDevice");
                hip_check(hipStreamCreateWithFlags(&slot_[s].streams[(size_t)r], hipStreamNonBlocking), "hipStreamCreate");
            }
        }
        if (ndev > 1) {
            comms_.resize((size_t)ndev);
            std::vector<int> devs((size_t)ndev);
            for (int r = 0; r < ndev; ++r) devs[(size_t)r] = r;
            nccl_check(ncclCommInitAll(comms_.data(), ndev, devs.data()), "ncclCommInitAll");
        }
    }
    ~MultiGpuWorld()
    {
        for (ncclComm_t c : comms_) if (c) (void)ncclCommDestroy(c);
        for (int s = 0; s < SLOTS; ++s) {
            release(slot_[s]);
            for (int r = 0; r < n_ && r < (int)slot_[s].streams.size(); ++r) {
                (void)hipSetDevice(r);
                if (slot_[s].streams[(size_t)r]) (void)hipStreamDestroy(slot_[s].streams[(size_t)r]);
            }
        }
    }
    MultiGpuWorld(const MultiGpuWorld &) = delete;
    MultiGpuWorld &operator=(const MultiGpuWorld &) = delete;

    int devices() const { return n_; }
    World &world(int r) { return *worlds_[(size_t)r]; }
    // point-to-point operations (send/recv pairs) and strided copies one draw_frames() call of F frames issues
    int exchange_ops() const { return n_ - 1; }
    int copy_ops(int frames) const { return n_ * frames; }

    // World::draw of cams.size() (<= SVO_MAX_FRAMES) views over the node: the packed frames (8-byte records, what
    // svo_gbuffer_unpack / svo_shade_packed read), one after the other, on device 0; frame f starts at
    // frames() + f * frame_stride().  Asynchronous; wait(slot) completes it.  Calls alternate between the SLOTS slots by
    // themselves; a slot's previous frames are overwritten by the call that comes round to it.
    const uint64_t *draw_frames(const std::vector<Camera> &cams, bool shadow = false, const float light_dir[3] = nullptr, int *slot_out = nullptr)
    {
        if (cams.empty() || (int)cams.size() > SVO_MAX_FRAMES) throw std::runtime_error("MultiGpuWorld::draw_frames: 1.." + std::to_string(SVO_MAX_FRAMES) + " cameras");
        const int F = (int)cams.size(), s = next_slot_;
        next_slot_ = (next_slot_ + 1) % SLOTS;
        Slot &S = slot_[s];
        resize(S, cams[0].width, cams[0].height, F);
        svo_trace_params p;
        std::memset(&p, 0, sizeof p);
        p.shadow = shadow ? 1 : 0;
        p.tiles_per_wave = 4;                               // a rank's share is a small raster: keep the waves refilling
        p.launches_in_flight = SLOTS;                       // calls alternate between SLOTS slots
        if (light_dir) std::memcpy(p.light_dir, light_dir, sizeof p.light_dir);
        std::vector<svo_camera> plain(cams.begin(), cams.end());
        const BandLayout L{ n_, S.width, S.height, F };       // (band_layout.hpp: the arithmetic, checked on a CPU for N = 2, 4, 8)
        const size_t share = L.share_records();             // records of one rank, all frames
        for (int r = 0; r < n_; ++r) {                      // every device: its bands of every frame, then the 8-byte form
            hip_check(hipSetDevice(r), "hipSetDevice");
            check(svo_trace_rows_frames(worlds_[(size_t)r]->handle(), plain.data(), F, &p, r, n_, S.nb, BAND, S.bands[(size_t)r], S.streams[(size_t)r]), "svo_trace_rows_frames");
            check(svo_gbuffer_pack(S.bands[(size_t)r], S.packed[(size_t)r], (int64_t)share, S.streams[(size_t)r]), "svo_gbuffer_pack");
        }
        if (n_ > 1) {                                       // ONE message per peer
            nccl_check(ncclGroupStart(), "ncclGroupStart");
            for (int r = 1; r < n_; ++r) {
                nccl_check(ncclSend(S.packed[(size_t)r], share, ncclUint64, 0, comms_[(size_t)r], S.streams[(size_t)r]), "ncclSend");
                nccl_check(ncclRecv(S.staging[(size_t)r], share, ncclUint64, r, comms_[0], S.streams[0]), "ncclRecv");
            }
            nccl_check(ncclGroupEnd(), "ncclGroupEnd");
        }
        // band k of rank r of frame f = rows (k*N + r)*8 .. of that frame: one strided copy per (frame, rank) on device 0
        hip_check(hipSetDevice(0), "hipSetDevice");
        for (int f = 0; f < F; ++f)
            for (int r = 0; r < n_; ++r) {
                const BandCopy c = L.copy(f, r);
                const uint64_t *src = (r == 0 ? S.packed[0] : S.staging[(size_t)r]) + c.src;
                hip_check(hipMemcpy2DAsync(S.frames + c.dst, c.dst_pitch * 8, src, c.src_pitch * 8, c.row_records * 8, c.rows, hipMemcpyDeviceToDevice, S.streams[0]), "hipMemcpy2DAsync");
            }
        if (slot_out) *slot_out = s;
        return S.frames;
    }
    // World::draw: one view.
    const uint64_t *draw(const Camera &cam, bool shadow = false, const float light_dir[3] = nullptr, int *slot_out = nullptr)
    {
        return draw_frames(std::vector<Camera>(1, cam), shadow, light_dir, slot_out);
    }
    size_t frame_stride(int slot = -1) const { return slot_[slot < 0 ? (next_slot_ + SLOTS - 1) % SLOTS : slot].frame_stride; }    // records between frames (rows padded to whole rounds of bands)
    void wait(int slot = -1)
    {
        for (int s = 0; s < SLOTS; ++s) {
            if (slot >= 0 && s != slot) continue;
            for (int r = 0; r < n_; ++r) { hip_check(hipSetDevice(r), "hipSetDevice"); hip_check(hipStreamSynchronize(slot_[s].streams[(size_t)r]), "hipStreamSynchronize"); }
        }
        hip_check(hipSetDevice(0), "hipSetDevice");
    }
    void *stream(int r, int slot = 0) const { return slot_[slot].streams[(size_t)r]; }

private:
    struct Slot {
        std::vector<hipStream_t> streams;
        std::vector<svo_hit *> bands;                    // per device: its bands of every frame, 32-byte records
        std::vector<uint64_t *> packed;                  // per device: the same, 8-byte records (what travels)
        std::vector<uint64_t *> staging;                 // on device 0, per peer: what arrives
        uint64_t *frames = nullptr;                      // on device 0: the frames, de-interleaved
        size_t frame_stride = 0;
        int nb = 0, width = 0, height = 0, frames_cap = 0;
    };
    void release(Slot &S)
    {
        for (int r = 0; r < n_ && r < (int)S.bands.size(); ++r) {
            (void)hipSetDevice(r);
            (void)hipFree(S.bands[(size_t)r]); S.bands[(size_t)r] = nullptr;
            (void)hipFree(S.packed[(size_t)r]); S.packed[(size_t)r] = nullptr;
        }
        (void)hipSetDevice(0);
        for (size_t r = 0; r < S.staging.size(); ++r) { (void)hipFree(S.staging[r]); S.staging[r] = nullptr; }
        (void)hipFree(S.frames); S.frames = nullptr;
        S.width = S.height = S.frames_cap = 0;
    }
    void resize(Slot &S, int w, int h, int frames)
    {
        if (w == S.width && h == S.height && frames <= S.frames_cap) return;
        wait();                                             // nothing of this slot (or the other) may still use the buffers
        release(S);                                         // (pointers are nulled: a failed hipMalloc below leaves nothing dangling)
        const BandLayout L{ n_, w, h, frames };
        S.nb = L.bands_per_rank();
        const size_t share = L.share_records();
        for (int r = 0; r < n_; ++r) {
            hip_check(hipSetDevice(r), "hipSetDevice");
            hip_check(hipMalloc((void **)&S.bands[(size_t)r], share * sizeof(svo_hit)), "hipMalloc");
            hip_check(hipMalloc((void **)&S.packed[(size_t)r], share * 8), "hipMalloc");
        }
        hip_check(hipSetDevice(0), "hipSetDevice");
        for (int r = 1; r < n_; ++r) hip_check(hipMalloc((void **)&S.staging[(size_t)r], share * 8), "hipMalloc");
        S.frame_stride = L.frame_stride();                  // whole rounds of bands: the last one may hang over the image
        hip_check(hipMalloc((void **)&S.frames, (size_t)frames * S.frame_stride * 8), "hipMalloc");
        S.width = w; S.height = h; S.frames_cap = frames;
    }

    int n_ = 0, next_slot_ = 0;
    std::vector<std::unique_ptr<World>> worlds_;
    std::vector<ncclComm_t> comms_;
    Slot slot_[SLOTS];
};

} // namespace svo
