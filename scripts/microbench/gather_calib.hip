// gather_calib.hip — what does rocprofv3's FETCH_SIZE count for the march's access pattern?
//
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read and is
// uncalibrated for other access widths.  The SVO march reads single dwords (node words) and 8-byte masks from lines
// scattered over pools far larger than the 256 MiB Infinity Cache, so this program issues known numbers of such reads
// over a 4 GiB table and the profile (scripts/prof_round2.sh, --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum ...) tells what
// the counter makes of them:
//   k_gather_dword<1>   one dword from each of N distinct 128-byte lines (every lane its own line)
//   k_gather_dword<2>   two dwords 64 bytes apart from each of N distinct lines (both 64-byte halves)
//   k_gather_qword      one 8-byte word from each of N distinct lines
//   k_stream_x4         N*128 bytes read as coalesced 16-byte-per-lane loads (the guide's half-count case)
// N = 2^24 lines = 2 GiB of distinct lines out of a 4 GiB table.  Prints the byte counts to compare with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned long long LINES = 1ull << 25;            // 128-byte lines in the table (4 GiB)
constexpr unsigned N = 1u << 24;                            // lines touched per kernel

__device__ __forceinline__ unsigned long long line_of(unsigned i) { return ((unsigned long long)i * 2654435761ull) & (LINES - 1); }   // odd multiplier: a bijection on 2^25

template <int HALVES>
__global__ __launch_bounds__(256) void k_gather_dword(const unsigned *table, unsigned *out)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long w = line_of(i) * 32ull + (i & 7u);            // some dword of the first half
    unsigned v = table[w];
    if (HALVES == 2) v += table[w + 16];
    if (v == 0x12345678u) out[0] = v;
}
__global__ __launch_bounds__(256) void k_gather_qword(const unsigned long long *table, unsigned *out)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long v = table[line_of(i) * 16ull + (i & 3u)];
    if (v == 0x12345678ull) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_stream_x4(const uint4 *table, unsigned *out)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;        // 16 B per lane, N*8 lanes = N*128 B
    const uint4 v = table[i];
    if (v.x == 0x12345678u && v.y == 1u) out[0] = v.z;
}

int main()
{
    unsigned *table, *out;
    CHECK(hipMalloc(&table, LINES * 128ull));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(table, 0, LINES * 128ull));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_gather_dword<1>, dim3(N / 256), dim3(256), 0, 0, table, out);
        hipLaunchKernelGGL(k_gather_dword<2>, dim3(N / 256), dim3(256), 0, 0, table, out);
        hipLaunchKernelGGL(k_gather_qword, dim3(N / 256), dim3(256), 0, 0, (const unsigned long long *)table, out);
        hipLaunchKernelGGL(k_stream_x4, dim3(N * 8 / 256), dim3(256), 0, 0, (const uint4 *)table, out);
        CHECK(hipDeviceSynchronize());
    }
    printf("{\"lines_touched\": %u, \"bytes_if_64_per_line\": %llu, \"bytes_if_128_per_line\": %llu, \"stream_bytes\": %llu}\n",
           N, (unsigned long long)N * 64ull, (unsigned long long)N * 128ull, (unsigned long long)N * 128ull);
    return 0;
}
