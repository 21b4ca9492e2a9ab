// multi_gpu_example.cpp — World::draw over N devices of one node (multi_gpu.hpp), checked against single-device frames.
//
//   hipcc -std=c++17 -I. multi_gpu_example.cpp -L.. -lsvo_amd -lrccl -Wl,-rpath,'$ORIGIN/..' -o multi_gpu_example
//   ./multi_gpu_example <devices> [depth] [width height]
//
// Two draw_frames() calls in flight (3 views, then 2: both slots), every gathered, de-interleaved frame compared with the
// frame device 0 traces alone.
// exit 0: all equal, record for record;  exit 1: some differ;  exit 3: fewer devices than asked for (nothing was run).
#include <cstdio>
#include <cstdlib>

#include "multi_gpu.hpp"

int main(int argc, char **argv)
{
    const int ndev = argc > 1 ? std::atoi(argv[1]) : 1;
    const uint32_t depth = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 8;
    const int w = argc > 4 ? std::atoi(argv[3]) : 640, h = argc > 4 ? std::atoi(argv[4]) : 356;      // 356: not a multiple of 8 bands * devices
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < ndev) {
        std::fprintf(stderr, "multi_gpu_example: %d devices asked for, %d present\n", ndev, have);
        return 3;
    }
    try {
        svo::MultiGpuWorld node(ndev, 2, 1, 2, 128, depth);
        std::vector<svo::Camera> views;
        for (int f = 0; f < 5; ++f)
            views.emplace_back(svo::vec3{ 128.0f + 7.0f * (float)f, 150.0f - 3.0f * (float)f, -40.0f }, svo::vec3{ 0.03f * (float)f, -0.5f, 0.866f }, svo::vec3{ 0.0f, 1.0f, 0.0f }, 60.0f, w, h);
        const std::vector<svo::Camera> first(views.begin(), views.begin() + 3), second(views.begin() + 3, views.end());
        int slot_a = -1, slot_b = -1;
        const uint64_t *fa = node.draw_frames(first, true, nullptr, &slot_a);
        const size_t stride_a = node.frame_stride(slot_a);
        const uint64_t *fb = node.draw_frames(second, true, nullptr, &slot_b);      // in flight behind the first call
        const size_t stride_b = node.frame_stride(slot_b);
        node.wait();

        size_t differ = 0, hits = 0;
        svo::GBuffer whole(w, h);
        uint64_t *packed = static_cast<uint64_t *>(svo_device_alloc((size_t)w * h * 8));
        std::vector<uint64_t> frame((size_t)w * h), ref((size_t)w * h);
        for (int f = 0; f < 5; ++f) {
            const uint64_t *src = f < 3 ? fa + (size_t)f * stride_a : fb + (size_t)(f - 3) * stride_b;
            svo::hip_check(hipMemcpy(frame.data(), src, frame.size() * 8, hipMemcpyDeviceToHost), "hipMemcpy");
            node.world(0).draw(views[(size_t)f], whole, true);                      // the same view traced by device 0 alone
            svo::check(svo_gbuffer_pack(whole.device(), packed, (int64_t)w * h, nullptr), "svo_gbuffer_pack");
            svo::check(svo_memcpy_d2h(ref.data(), packed, ref.size() * 8), "svo_memcpy_d2h");
            for (size_t i = 0; i < ref.size(); ++i) { differ += frame[i] != ref[i]; hits += (ref[i] >> 48) & 1u; }
        }
        svo_device_free(packed);
        std::printf("multi_gpu_example: %d device(s), 5 frames of %dx%d in two calls (slots %d, %d), depth %u: %d point-to-point ops + %d strided copies per 3-frame call; "
                    "%zu hit pixels, %zu records differ from the single-device frames\n",
                    ndev, w, h, slot_a, slot_b, depth, node.exchange_ops(), node.copy_ops(3), hits, differ);
        return differ ? 1 : 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "multi_gpu_example: %s\n", e.what());
        return 2;
    }
}
