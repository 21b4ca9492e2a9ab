// band_layout_check.cpp — the multi-GPU exchange's index arithmetic (band_layout.hpp) applied on a CPU.
//
//   g++ -std=c++17 -I. band_layout_check.cpp -o band_layout_check
//   ./band_layout_check <ranks> <width> <height> <frames>  > frames.u64
//
// Every rank's buffer is filled the way svo_trace_rows_frames + svo_gbuffer_pack leave it (frame after frame, the rank's bands
// stacked in k order) with records that name their origin - tag(rank, frame, band k, row j, x) - and the very copies
// MultiGpuWorld::draw_frames issues (BandLayout::copy, hipMemcpy2DAsync there) are applied with memcpy.  Checked here: every
// record of every frame is written exactly once, and row y of frame f holds the records of rank (y/8) % N, band (y/8) / N, row
// y % 8.  The gathered frames (frame_stride apart, as on device 0) go to stdout for tests/test_host_units.py, which holds
// them against partition.deinterleave.  exit 0: all good;  1: a check failed;  2: usage.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "band_layout.hpp"

static uint64_t tag(int r, int f, int k, int j, int x) { return ((uint64_t)r << 56) | ((uint64_t)f << 48) | ((uint64_t)k << 32) | ((uint64_t)j << 24) | (uint64_t)x; }

int main(int argc, char **argv)
{
    if (argc != 5) { std::fprintf(stderr, "usage: band_layout_check ranks width height frames\n"); return 2; }
    const svo::BandLayout L{ std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]) };
    const int N = L.ranks, W = L.width, H = L.height, F = L.frames, nb = L.bands_per_rank();
    std::vector<std::vector<uint64_t>> rank_buf((size_t)N, std::vector<uint64_t>(L.share_records()));
    for (int r = 0; r < N; ++r)
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < nb; ++k)
                for (int j = 0; j < svo::BAND; ++j)
                    for (int x = 0; x < W; ++x)
                        rank_buf[(size_t)r][(((size_t)f * nb + k) * svo::BAND + j) * W + x] = tag(r, f, k, j, x);
    std::vector<uint64_t> frames((size_t)F * L.frame_stride(), ~0ull);
    std::vector<uint8_t> written(frames.size(), 0);
    for (int f = 0; f < F; ++f)
        for (int r = 0; r < N; ++r) {
            const svo::BandCopy c = L.copy(f, r);
            for (size_t row = 0; row < c.rows; ++row) {
                const size_t s = c.src + row * c.src_pitch, d = c.dst + row * c.dst_pitch;
                if (s + c.row_records > rank_buf[(size_t)r].size() || d + c.row_records > frames.size()) { std::fprintf(stderr, "copy (%d, %d) row %zu out of range\n", f, r, row); return 1; }
                std::memcpy(&frames[d], &rank_buf[(size_t)r][s], c.row_records * 8);
                for (size_t i = 0; i < c.row_records; ++i) ++written[d + i];
            }
        }
    for (size_t i = 0; i < written.size(); ++i)
        if (written[i] != 1) { std::fprintf(stderr, "record %zu written %d times\n", i, (int)written[i]); return 1; }
    for (int f = 0; f < F; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int b = y / svo::BAND;
                const uint64_t want = tag(b % N, f, b / N, y % svo::BAND, x), got = frames[(size_t)f * L.frame_stride() + (size_t)y * W + x];
                if (got != want) { std::fprintf(stderr, "frame %d row %d x %d: %016llx, expected %016llx\n", f, y, x, (unsigned long long)got, (unsigned long long)want); return 1; }
                if (L.band_first_row(b % N, b / N) != b * svo::BAND) { std::fprintf(stderr, "band_first_row(%d, %d)\n", b % N, b / N); return 1; }
            }
    std::fwrite(frames.data(), 8, frames.size(), stdout);
    return 0;
}
