// band_layout.hpp — where the bands of a multi-GPU frame live: pure index arithmetic, no HIP, no RCCL.
//
// The image is cut into bands of BAND rows dealt round-robin: rank r of N traces bands r, r+N, r+2N, ...  (svo_trace_rows_frames
// with band0 = r, band_stride = N writes them stacked in that order, frame after frame).  multi_gpu.hpp moves every rank's
// buffer to device 0 as ONE message and then puts the bands in place with one strided 2-D copy per (frame, rank); the
// numbers of those copies are what this header computes, so that they can be checked on a CPU for N = 2, 4, 8 without a
// second GPU (tests/test_host_units.py applies them with memcpy) - the N > 1 exchange itself has never run on hardware.
// octree-raymarcher_amd/partition.py is the Python twin (bench.py's RCCL path); the test holds the two against each other.
#pragma once
#include <cstddef>

namespace svo {

constexpr int BAND = 8;                                  // rows per band == tile height of the stack kernel

// One strided copy in units of RECORDS (multiply by the record size for bytes): `rows` rows of `row_records` records,
// the source rows `src_pitch` apart starting at `src`, the destination rows `dst_pitch` apart starting at `dst`.
struct BandCopy { size_t src, dst, row_records, src_pitch, dst_pitch, rows; };

struct BandLayout {
    int ranks = 1, width = 0, height = 0, frames = 1;

    constexpr int bands_total() const { return (height + BAND - 1) / BAND; }
    // bands every rank traces (the same count on every rank; trailing ones may lie below the image: written as misses)
    constexpr int bands_per_rank() const { return (bands_total() + ranks - 1) / ranks; }
    constexpr size_t band_records() const { return (size_t)BAND * (size_t)width; }
    // records one rank produces (and sends) for all frames of a call
    constexpr size_t share_records() const { return (size_t)frames * (size_t)bands_per_rank() * band_records(); }
    // records between two frames of the gathered result: whole rounds of bands, so the last round may hang over the image
    constexpr size_t frame_stride() const { return (size_t)bands_per_rank() * (size_t)ranks * band_records(); }
    // first image row of the k-th band of rank r
    constexpr int band_first_row(int r, int k) const { return (k * ranks + r) * BAND; }

    // the bands of rank r of frame f, from that rank's buffer (frame f starts bands_per_rank() bands in) into the frames:
    // band k goes to rows (k*N + r)*BAND.. of frame f, i.e. a destination pitch of N bands.
    constexpr BandCopy copy(int f, int r) const
    {
        return BandCopy{ (size_t)f * (size_t)bands_per_rank() * band_records(),
                         (size_t)f * frame_stride() + (size_t)r * band_records(),
                         band_records(), band_records(), band_records() * (size_t)ranks, (size_t)bands_per_rank() };
    }
};

} // namespace svo
