// valu_issue.hip — how many cycles does one wave64 VALU instruction hold a gfx950 SIMD for?
//
// W waves per SIMD (W = 1..8) each run a stream of independent VALU instructions (8 accumulators, so that the
// dependent-issue latency never gates the stream).  Per wave: s_memtime before and after; cycles per instruction per
// SIMD = elapsed / (instructions per wave * W) because the W waves of a SIMD share its issue port for that interval.
// The same figure follows from the launch's wall time (HIP events) and the shader clock.  Mixes measured: v_fma_f32,
// v_add_u32, v_bfe_u32, v_cmp+v_cndmask (the march step's compare/select pairs), v_mul_f32 + v_cvt, and the packed
// v_pk_mul_f32 / v_pk_add_f32 (two float operations per lane and instruction) the compiler forms from xyz arithmetic.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue > valu_issue.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_FMA = 0, OP_ADDU = 1, OP_BFE = 2, OP_CMPSEL = 3, OP_MULCVT = 4, OP_PKMUL = 5, OP_PKADD = 6, OP_FMA_BR_NT = 7, OP_FMA_BR_T = 8, OP_FMA_SALU = 9, OP_FMA_NOP = 10, N_OPS = 11 };
static const char *op_name[N_OPS] = { "v_fma_f32", "v_add_u32", "v_bfe_u32", "v_cmp_lt_f32+v_cndmask_b32", "v_mul_f32+v_cvt_i32_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                     "v_fma_f32 + s_cbranch_execz (not taken)", "v_fma_f32 + s_cbranch_execnz (taken, to the next instruction)",
                                     "v_fma_f32 + s_and_b64", "v_fma_f32 + s_nop 1" };
static const int op_insts[N_OPS] = { 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1 };     // VALU instructions per "op" below (the companions of the last four: compare with v_fma_f32 alone)

constexpr int UNROLL = 8;       // independent accumulators
constexpr int INNER = 16;       // ops per accumulator per loop trip (loop overhead: 2 SALU per 128+ VALU)

template <int OP>
__global__ __launch_bounds__(64) void k_issue(int trips, float seed, unsigned long long *cycles, float *sink, int pattern, unsigned long long *stamps)
{
    // which lanes execute the stream (EXEC during the timed loop): 0 all 64, 1 the low 32, 2 the even lanes, 3 the low 16,
    // 4 lane 0 alone - does the SIMD skip a half (quarter) of a wave64 instruction whose lanes are all off?
    const bool on = pattern == 0 ? true : pattern == 1 ? threadIdx.x < 32 : pattern == 2 ? (threadIdx.x & 1) == 0 : pattern == 3 ? threadIdx.x < 16 : threadIdx.x == 0;
    float a[UNROLL];
    unsigned u[UNROLL];
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v pk[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + (float)(threadIdx.x + i); u[i] = threadIdx.x * 7u + i; pk[i].x = a[i]; pk[i].y = a[i] * 0.5f; }
    float2v pm; pm.x = 1.0000001f; pm.y = 0.9999999f;
    const float m = 1.0000001f, c = 1e-7f;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = 0, t1 = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz: when, in wall time, this wave ran
    if (on) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < INNER; ++k) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == OP_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
                if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(u[i]));
                if (OP == OP_CMPSEL) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
                if (OP == OP_MULCVT) asm volatile("v_mul_f32 %0, %0, %2\n\tv_cvt_i32_f32 %1, %0" : "+v"(a[i]), "=v"(u[i]) : "v"(m));
                if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[i]) : "v"(pm));
                if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[i]) : "v"(pm));
                if (OP == OP_FMA_BR_NT) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_cbranch_execz 1f\n1:" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == OP_FMA_BR_T) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_cbranch_execnz 1f\n1:" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == OP_FMA_SALU) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_and_b64 s[40:41], s[40:41], exec" : "+v"(a[i]) : "v"(m), "v"(c) : "s40", "s41", "scc");
                if (OP == OP_FMA_NOP) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 1" : "+v"(a[i]) : "v"(m), "v"(c));
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    }
    float s = 0.0f; unsigned q = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { s += a[i] + pk[i].x + pk[i].y; q += u[i]; }
    if (s == 12345.678f && q == 42u) sink[0] = s;             // keep the streams alive
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        cycles[blockIdx.x] = t1 - t0;
        // where it ran: XCC, SE, CU, SIMD (HW_REG_XCC_ID, HW_REG_HW_ID) - the waves that really shared a SIMD are grouped by this
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        stamps[3 * blockIdx.x] = r0; stamps[3 * blockIdx.x + 1] = r1;
        stamps[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | (hw & 0x0000FF30u);     // se/sh/cu bits 15:8, simd bits 5:4
    }
}

template <int OP>
static void run(int waves_per_simd, int ncu, double clock_ghz, bool last, int pattern = 0)
{
    const int trips = 2000;
    const int blocks = ncu * 4 * waves_per_simd;               // 64-thread blocks: the dispatcher spreads them over the SIMDs
    unsigned long long *d_cyc, *d_st; float *d_sink;
    CHECK(hipMalloc(&d_cyc, blocks * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_st, 3 * blocks * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(64), 0, 0, 10, 1.0f, d_cyc, d_sink, pattern, d_st);    // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(64), 0, 0, trips, 1.0f, d_cyc, d_sink, pattern, d_st);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(blocks);
    CHECK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // the reconciliation of the two views: group the waves by the SIMD they really ran on, take each SIMD's
    // busy interval (first start to last end, 100 MHz stamps) and its real wave count, and the shader clock from the ratio of
    // the two counters of one wave
    std::vector<unsigned long long> st(3 * (size_t)blocks);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::map<unsigned long long, std::vector<int>> simd;
    for (int b = 0; b < blocks; ++b) simd[st[3 * b + 2]].push_back(b);
    double clk_sum = 0; int clk_n = 0;
    for (int b = 0; b < blocks; ++b) if (st[3 * b + 1] > st[3 * b]) { clk_sum += (double)cyc[b] / ((double)(st[3 * b + 1] - st[3 * b]) * 10.0); ++clk_n; }   // cycles per ns
    const double ghz_measured = clk_n ? clk_sum / clk_n : 0.0;
    std::vector<double> per_simd; std::vector<int> occ;
    unsigned long long first = ~0ull, last_end = 0;
    for (auto &kv : simd) {
        unsigned long long a = ~0ull, z = 0;
        for (int b : kv.second) { a = std::min(a, st[3 * b]); z = std::max(z, st[3 * b + 1]); }
        first = std::min(first, a); last_end = std::max(last_end, z);
        occ.push_back((int)kv.second.size());
        per_simd.push_back((double)(z - a) * 10.0 * ghz_measured);                 // busy interval in shader cycles
    }
    std::sort(occ.begin(), occ.end());
    const double insts = (double)trips * INNER * UNROLL * op_insts[OP];      // VALU instructions per wave
    std::vector<double> cpi; { size_t i = 0; for (auto &kv : simd) { cpi.push_back(per_simd[i] / (insts * (double)kv.second.size())); ++i; } }
    std::sort(cpi.begin(), cpi.end());
    std::sort(cyc.begin(), cyc.end());
    const double med = (double)cyc[blocks / 2], p10 = (double)cyc[blocks / 10], p90 = (double)cyc[blocks * 9 / 10];
    // wall-clock view: every SIMD issues insts * W instructions in ms
    const double wall_cyc_per_inst = (ms * 1e-3 * clock_ghz * 1e9) / (insts * waves_per_simd);
    static const char *pat[] = { "all 64 lanes", "low 32 lanes", "even lanes", "low 16 lanes", "lane 0" };
    printf("    {\"op\": \"%s\", \"exec\": \"%s\", \"waves_per_simd\": %d, \"valu_insts_per_wave\": %.0f, \"wave_cycles_median\": %.0f, "
           "\"wave_cycles_p10\": %.0f, \"wave_cycles_p90\": %.0f, \"cycles_per_inst_per_simd\": %.3f, "
           "\"launch_ms\": %.4f, \"cycles_per_inst_per_simd_from_wall_at_%.1fGHz\": %.3f, "
           "\"shader_clock_ghz_measured\": %.3f, \"simds_used\": %d, \"waves_on_a_simd_min_median_max\": [%d, %d, %d], "
           "\"cycles_per_inst_per_simd_grouped_p10_median_p90\": [%.3f, %.3f, %.3f], \"kernel_span_us_by_stamps\": %.1f}%s\n",
           op_name[OP], pat[pattern], waves_per_simd, insts, med, p10, p90, med / (insts * waves_per_simd), ms, clock_ghz, wall_cyc_per_inst,
           ghz_measured, (int)simd.size(), occ.front(), occ[occ.size() / 2], occ.back(), cpi[cpi.size() / 10], cpi[cpi.size() / 2], cpi[cpi.size() * 9 / 10],
           (double)(last_end - first) * 0.01, last ? "" : ",");
    CHECK(hipFree(d_cyc)); CHECK(hipFree(d_sink)); CHECK(hipFree(d_st));
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("{\n  \"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f,\n", prop.name, prop.gcnArchName, ncu, ghz);
    printf("  \"note\": \"cycles_per_inst_per_simd = median wave s_memtime cycles / (VALU insts per wave * waves per SIMD); 8 independent accumulators per wave\",\n");
    printf("  \"runs\": [\n");
    const int ws[] = { 1, 2, 4, 5, 6, 8 };
    for (int w : ws) {
        run<OP_FMA>(w, ncu, ghz, false);
        run<OP_ADDU>(w, ncu, ghz, false);
        run<OP_BFE>(w, ncu, ghz, false);
        run<OP_CMPSEL>(w, ncu, ghz, false);
        run<OP_MULCVT>(w, ncu, ghz, false);
        run<OP_PKMUL>(w, ncu, ghz, false);
        run<OP_PKADD>(w, ncu, ghz, false);
    }
    // EXEC patterns: a partly filled wave costs the SIMD what?
    for (int w : { 1, 5 })
        for (int pattern = 0; pattern < 5; ++pattern) {
            run<OP_FMA>(w, ncu, ghz, false, pattern);
            run<OP_PKMUL>(w, ncu, ghz, false, pattern);
            run<OP_CMPSEL>(w, ncu, ghz, false, pattern);
        }
    // what a scalar companion costs next to a vector instruction: branch (not taken / taken), SALU, s_nop
    for (int w : { 1, 2, 5, 6, 8 }) {
        run<OP_FMA>(w, ncu, ghz, false);
        run<OP_FMA_BR_NT>(w, ncu, ghz, false);
        run<OP_FMA_BR_T>(w, ncu, ghz, false);
        run<OP_FMA_SALU>(w, ncu, ghz, false);
        run<OP_FMA_NOP>(w, ncu, ghz, w == 8);
    }
    printf("  ]\n}\n");
    return 0;
}
