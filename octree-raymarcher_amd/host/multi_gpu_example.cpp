// multi_gpu_example.cpp — World::draw over N devices of one node (multi_gpu.hpp), checked against the single-device frame.
//
//   hipcc -std=c++17 -I. multi_gpu_example.cpp -L.. -lsvo_amd -lrccl -Wl,-rpath,'$ORIGIN/..' -o multi_gpu_example
//   ./multi_gpu_example <devices> [depth] [width height]
//
// exit 0: the gathered, de-interleaved frame equals the frame device 0 traces alone, record for record;
// exit 1: it differs;  exit 3: fewer devices than asked for (nothing was run).
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
        svo::Camera cam({ 128.0f, 150.0f, -40.0f }, { 0.0f, -0.5f, 0.866f }, { 0.0f, 1.0f, 0.0f }, 60.0f, w, h);
        const uint64_t *frame_dev = node.draw(cam, true);
        node.wait();
        std::vector<uint64_t> frame((size_t)w * h);
        svo::hip_check(hipMemcpy(frame.data(), frame_dev, frame.size() * 8, hipMemcpyDeviceToHost), "hipMemcpy");

        // the same frame traced by device 0 alone
        svo::GBuffer whole(w, h);
        node.world(0).draw(cam, whole, true);
        uint64_t *packed = static_cast<uint64_t *>(svo_device_alloc((size_t)w * h * 8));
        svo::check(svo_gbuffer_pack(whole.device(), packed, (int64_t)w * h, nullptr), "svo_gbuffer_pack");
        std::vector<uint64_t> ref((size_t)w * h);
        svo::check(svo_memcpy_d2h(ref.data(), packed, ref.size() * 8), "svo_memcpy_d2h");
        svo_device_free(packed);

        size_t differ = 0, hits = 0;
        for (size_t i = 0; i < ref.size(); ++i) { differ += frame[i] != ref[i]; hits += (ref[i] >> 48) & 1u; }
        std::printf("multi_gpu_example: %d device(s), %dx%d, depth %u: %zu hit pixels, %zu records differ from the single-device frame\n",
                    ndev, w, h, depth, hits, differ);
        return differ ? 1 : 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "multi_gpu_example: %s\n", e.what());
        return 2;
    }
}
