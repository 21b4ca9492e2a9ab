// multi_gpu.hpp — C++ host side of the multi-GPU frame (BASELINE north_star: "host code stays C++ ...
// the image tile-partitions across the 8 GPUs of one node with an RCCL gather of per-tile G-buffers over xGMI").
//
// One process drives N devices.  Every device holds the whole world (rays are independent, the world is read-only);
// the image is cut into 8-row bands dealt round-robin (device r traces bands r, r+N, ...: svo_trace_rows_frames), each
// device packs its bands to the lossless 8-byte record (svo_gbuffer_pack) and the bands travel to device 0 with ONE
// grouped RCCL exchange per frame: ncclSend per band on device r, ncclRecv of that band straight into its rows of the
// frame on device 0 - the de-interleave is the receive address, no extra kernel and no staging copy.  xGMI is
// point-to-point: device 0 receives from its N-1 peers over N-1 different links at once.
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
#include <vector>

#include "svo_world.hpp"

namespace svo {

inline void hip_check(hipError_t e, const char *where) { if (e != hipSuccess) throw std::runtime_error(std::string(where) + ": " + hipGetErrorString(e)); }
inline void nccl_check(ncclResult_t r, const char *where) { if (r != ncclSuccess) throw std::runtime_error(std::string(where) + ": " + ncclGetErrorString(r)); }

constexpr int BAND = 8;                                  // rows per band == tile height of the stack kernel

class MultiGpuWorld {
public:
    // World::init on every device (the generator is deterministic: all replicas are identical) + load_gpu.
    MultiGpuWorld(int ndev, int w, int h, int d, int chunksize, uint32_t depth) : n_(ndev)
    {
        int have = 0;
        hip_check(hipGetDeviceCount(&have), "hipGetDeviceCount");
        if (ndev < 1 || ndev > have) throw std::runtime_error("MultiGpuWorld: " + std::to_string(ndev) + " devices asked for, " + std::to_string(have) + " present");
        worlds_.resize((size_t)ndev);
        streams_.assign((size_t)ndev, nullptr);
        for (int r = 0; r < ndev; ++r) {
            worlds_[(size_t)r] = std::make_unique<World>();
            worlds_[(size_t)r]->init(w, h, d, chunksize, depth);
            worlds_[(size_t)r]->load_gpu(r);
            hip_check(hipSetDevice(r), "hipSetDevice");
            hip_check(hipStreamCreateWithFlags(&streams_[(size_t)r], hipStreamNonBlocking), "hipStreamCreate");
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
        for (int r = 0; r < n_; ++r) {
            (void)hipSetDevice(r);
            if (streams_[(size_t)r]) (void)hipStreamDestroy(streams_[(size_t)r]);
            if (r < (int)bands_.size()) { (void)hipFree(bands_[(size_t)r]); (void)hipFree(packed_[(size_t)r]); }
        }
        (void)hipSetDevice(0);
        (void)hipFree(frame_);
    }
    MultiGpuWorld(const MultiGpuWorld &) = delete;
    MultiGpuWorld &operator=(const MultiGpuWorld &) = delete;

    int devices() const { return n_; }
    World &world(int r) { return *worlds_[(size_t)r]; }

    // World::draw over the node: the packed frame (height x width 8-byte records, svo_gbuffer_unpack / svo_shade_packed
    // read it) on device 0.  Asynchronous; wait() completes it.
    const uint64_t *draw(const Camera &cam, bool shadow = false, const float light_dir[3] = nullptr)
    {
        resize(cam.width, cam.height);
        svo_trace_params p;
        std::memset(&p, 0, sizeof p);
        p.shadow = shadow ? 1 : 0;
        if (light_dir) std::memcpy(p.light_dir, light_dir, sizeof p.light_dir);
        const int64_t band_px = (int64_t)BAND * width_;
        for (int r = 0; r < n_; ++r) {                      // every device: its bands, then the 8-byte form
            hip_check(hipSetDevice(r), "hipSetDevice");
            check(svo_trace_rows(worlds_[(size_t)r]->handle(), &cam, &p, r, n_, nb_, BAND, bands_[(size_t)r], streams_[(size_t)r]), "svo_trace_rows");
            check(svo_gbuffer_pack(bands_[(size_t)r], packed_[(size_t)r], nb_ * band_px, streams_[(size_t)r]), "svo_gbuffer_pack");
        }
        // band k of device r = image rows (k*N + r)*8 ..: receive it where it belongs
        if (n_ > 1) nccl_check(ncclGroupStart(), "ncclGroupStart");
        for (int r = 0; r < n_; ++r) {
            for (int k = 0; k < nb_; ++k) {
                const int row0 = (k * n_ + r) * BAND;
                if (row0 >= height_) continue;
                const int rows = std::min(BAND, height_ - row0);
                const size_t count = (size_t)rows * (size_t)width_;                         // uint64 records
                uint64_t *dst = frame_ + (size_t)row0 * (size_t)width_;
                const uint64_t *src = packed_[(size_t)r] + (size_t)k * (size_t)band_px;
                if (r == 0) {
                    hip_check(hipSetDevice(0), "hipSetDevice");
                    hip_check(hipMemcpyAsync(dst, src, count * 8, hipMemcpyDeviceToDevice, streams_[0]), "hipMemcpyAsync");
                } else {
                    nccl_check(ncclSend(src, count, ncclUint64, 0, comms_[(size_t)r], streams_[(size_t)r]), "ncclSend");
                    nccl_check(ncclRecv(dst, count, ncclUint64, r, comms_[0], streams_[0]), "ncclRecv");
                }
            }
        }
        if (n_ > 1) nccl_check(ncclGroupEnd(), "ncclGroupEnd");
        return frame_;
    }
    void wait()
    {
        for (int r = 0; r < n_; ++r) { hip_check(hipSetDevice(r), "hipSetDevice"); hip_check(hipStreamSynchronize(streams_[(size_t)r]), "hipStreamSynchronize"); }
        hip_check(hipSetDevice(0), "hipSetDevice");
    }
    void *stream(int r) const { return streams_[(size_t)r]; }

private:
    void resize(int w, int h)
    {
        if (w == width_ && h == height_) return;
        const int bands_total = (h + BAND - 1) / BAND;
        nb_ = (bands_total + n_ - 1) / n_;
        bands_.resize((size_t)n_, nullptr); packed_.resize((size_t)n_, nullptr);
        for (int r = 0; r < n_; ++r) {
            hip_check(hipSetDevice(r), "hipSetDevice");
            (void)hipFree(bands_[(size_t)r]); (void)hipFree(packed_[(size_t)r]);
            hip_check(hipMalloc((void **)&bands_[(size_t)r], (size_t)nb_ * BAND * w * sizeof(svo_hit)), "hipMalloc");
            hip_check(hipMalloc((void **)&packed_[(size_t)r], (size_t)nb_ * BAND * w * 8), "hipMalloc");
        }
        hip_check(hipSetDevice(0), "hipSetDevice");
        (void)hipFree(frame_);
        hip_check(hipMalloc((void **)&frame_, (size_t)h * w * 8), "hipMalloc");
        width_ = w; height_ = h;
    }

    int n_ = 0, nb_ = 0, width_ = 0, height_ = 0;
    std::vector<std::unique_ptr<World>> worlds_;
    std::vector<hipStream_t> streams_;
    std::vector<ncclComm_t> comms_;
    std::vector<svo_hit *> bands_;
    std::vector<uint64_t *> packed_;
    uint64_t *frame_ = nullptr;
};

} // namespace svo
