// tools/dualpipe.hip -- round 3: do the FMA-class VALU operations (2 cycles per wave on the SIMD-32 of CDNA4) and the
// "other" operations (v_min3 / v_min / v_cmp ...: 4 cycles) overlap, inside one wave and across the waves of a SIMD?
// The pair loop of the tile kernel is 13 packed FMA-class operations + 8 v_min3_f32 per trip; if the two kinds run side
// by side, its floor is max(13 x 4.3, 8 x 4.2) = 56 cycles per trip instead of the sum, 90.
// Build: hipcc --offload-arch=gfx950 -O3 dualpipe.hip -o dualpipe.   Occupancy is set with dynamic LDS.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

#define PK(d, s) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(s), "v"(c2))
#define PKN(d, s) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(c2), "v"(base))      // fresh destination, like g_k
#define MIN3(m, g) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(m) : "v"(g.x), "v"(g.y))
#define FMA(d, s) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(s), "v"(c1))
#define FMAN(d, s) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(c1), "v"(base.x))

template <int PATTERN>
__global__ __launch_bounds__(256) void k_mix(float* out, float seed)
{
    extern __shared__ float pad[];
    v2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    v2 g0, g1, g2, g3, g4, g5, g6, g7, h0, h1, h2, h3, h4, h5, h6, h7;
    v2 c2 = {seed * 0.25f, seed * 0.5f}, base = {seed, seed + 2.f};
    float c1 = seed * 0.125f;
    float m0 = 1e30f, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
    float f0 = seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    g0 = g1 = g2 = g3 = g4 = g5 = g6 = g7 = a0; h0 = h1 = h2 = h3 = h4 = h5 = h6 = h7 = a1;
    for (int i = 0; i < ITERS; ++i) {
        if (PATTERN == 0) {                     // 16 packed fma
            PK(a0, a1); PK(a1, a2); PK(a2, a3); PK(a3, a4); PK(a4, a5); PK(a5, a6); PK(a6, a7); PK(a7, a0);
            PK(a0, a1); PK(a1, a2); PK(a2, a3); PK(a3, a4); PK(a4, a5); PK(a5, a6); PK(a6, a7); PK(a7, a0);
        } else if (PATTERN == 1) {              // 16 min3
            MIN3(m0, a0); MIN3(m1, a1); MIN3(m2, a2); MIN3(m3, a3); MIN3(m4, a4); MIN3(m5, a5); MIN3(m6, a6); MIN3(m7, a7);
            MIN3(m0, a1); MIN3(m1, a2); MIN3(m2, a3); MIN3(m3, a4); MIN3(m4, a5); MIN3(m5, a6); MIN3(m6, a7); MIN3(m7, a0);
        } else if (PATTERN == 2) {              // 8 packed fma THEN 8 min3 of their results (what hipcc emits today)
            PKN(g0, a0); PKN(g1, a1); PKN(g2, a2); PKN(g3, a3); PKN(g4, a4); PKN(g5, a5); PKN(g6, a6); PKN(g7, a7);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 3) {              // the same 16, alternating, each min3 one packed fma behind its operand
            PKN(g0, a0);
            PKN(g1, a1); MIN3(m0, g0); PKN(g2, a2); MIN3(m1, g1); PKN(g3, a3); MIN3(m2, g2); PKN(g4, a4); MIN3(m3, g3);
            PKN(g5, a5); MIN3(m4, g4); PKN(g6, a6); MIN3(m5, g5); PKN(g7, a7); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 4) {              // software-pipelined: this trip's packed fma beside the LAST trip's min3
            PKN(g0, a0); MIN3(m0, h0); PKN(g1, a1); MIN3(m1, h1); PKN(g2, a2); MIN3(m2, h2); PKN(g3, a3); MIN3(m3, h3);
            PKN(g4, a4); MIN3(m4, h4); PKN(g5, a5); MIN3(m5, h5); PKN(g6, a6); MIN3(m6, h6); PKN(g7, a7); MIN3(m7, h7);
            PKN(h0, a1); MIN3(m0, g0); PKN(h1, a2); MIN3(m1, g1); PKN(h2, a3); MIN3(m2, g2); PKN(h3, a4); MIN3(m3, g3);
            PKN(h4, a5); MIN3(m4, g4); PKN(h5, a6); MIN3(m5, g5); PKN(h6, a7); MIN3(m6, g6); PKN(h7, a0); MIN3(m7, g7);
        } else if (PATTERN == 5) {              // the real mix, block order: 13 packed + 8 min3
            PK(a0, a1); PK(a1, a2); PK(a2, a3); PK(a3, a4); PK(a4, a5);
            PKN(g0, a0); PKN(g1, a1); PKN(g2, a2); PKN(g3, a3); PKN(g4, a4); PKN(g5, a5); PKN(g6, a6); PKN(g7, a7);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 6) {              // the real mix, interleaved and pipelined across trips (min3 of the last trip's g)
            PK(a0, a1); MIN3(m0, h0); PK(a1, a2); MIN3(m1, h1); PK(a2, a3); MIN3(m2, h2); PK(a3, a4); MIN3(m3, h3); PK(a4, a5); MIN3(m4, h4);
            PKN(g0, a0); MIN3(m5, h5); PKN(g1, a1); MIN3(m6, h6); PKN(g2, a2); MIN3(m7, h7); PKN(g3, a3); PKN(g4, a4); PKN(g5, a5); PKN(g6, a6); PKN(g7, a7);
            h0 = g0; h1 = g1; h2 = g2; h3 = g3; h4 = g4; h5 = g5; h6 = g6; h7 = g7;      // (register renaming: free in an unrolled-by-two loop)
        } else if (PATTERN == 7) {              // unpacked: 16 v_fma_f32 + 8 min3, alternating 2:1, pipelined across trips
            FMAN(g0.x, f0); FMAN(g0.y, f1); MIN3(m0, h0); FMAN(g1.x, f1); FMAN(g1.y, f2); MIN3(m1, h1); FMAN(g2.x, f2); FMAN(g2.y, f3); MIN3(m2, h2);
            FMAN(g3.x, f3); FMAN(g3.y, f4); MIN3(m3, h3); FMAN(g4.x, f4); FMAN(g4.y, f5); MIN3(m4, h4); FMAN(g5.x, f5); FMAN(g5.y, f6); MIN3(m5, h5);
            FMAN(g6.x, f6); FMAN(g6.y, f7); MIN3(m6, h6); FMAN(g7.x, f7); FMAN(g7.y, f0); MIN3(m7, h7);
            FMAN(h0.x, f0); FMAN(h0.y, f1); MIN3(m0, g0); FMAN(h1.x, f1); FMAN(h1.y, f2); MIN3(m1, g1); FMAN(h2.x, f2); FMAN(h2.y, f3); MIN3(m2, g2);
            FMAN(h3.x, f3); FMAN(h3.y, f4); MIN3(m3, g3); FMAN(h4.x, f4); FMAN(h4.y, f5); MIN3(m4, g4); FMAN(h5.x, f5); FMAN(h5.y, f6); MIN3(m5, g5);
            FMAN(h6.x, f6); FMAN(h6.y, f7); MIN3(m6, g6); FMAN(h7.x, f7); FMAN(h7.y, f0); MIN3(m7, g7);
        } else if (PATTERN == 8) {              // 16 v_fma_f32 then 8 min3 (block order, unpacked)
            FMAN(g0.x, f0); FMAN(g0.y, f1); FMAN(g1.x, f1); FMAN(g1.y, f2); FMAN(g2.x, f2); FMAN(g2.y, f3); FMAN(g3.x, f3); FMAN(g3.y, f4);
            FMAN(g4.x, f4); FMAN(g4.y, f5); FMAN(g5.x, f5); FMAN(g5.y, f6); FMAN(g6.x, f6); FMAN(g6.y, f7); FMAN(g7.x, f7); FMAN(g7.y, f0);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 10) {             // all unpacked: 10 shared + 16 plane fma + 8 min3
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0); FMA(f0, f2); FMA(f1, f3);
            FMAN(g0.x, f0); FMAN(g0.y, f1); FMAN(g1.x, f1); FMAN(g1.y, f2); FMAN(g2.x, f2); FMAN(g2.y, f3); FMAN(g3.x, f3); FMAN(g3.y, f4);
            FMAN(g4.x, f4); FMAN(g4.y, f5); FMAN(g5.x, f5); FMAN(g5.y, f6); FMAN(g6.x, f6); FMAN(g6.y, f7); FMAN(g7.x, f7); FMAN(g7.y, f0);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 11) {             // 5 packed shared + 16 unpacked plane fma + 8 min3
            PK(a0, a1); PK(a1, a2); PK(a2, a3); PK(a3, a4); PK(a4, a5);
            FMAN(g0.x, f0); FMAN(g0.y, f1); FMAN(g1.x, f1); FMAN(g1.y, f2); FMAN(g2.x, f2); FMAN(g2.y, f3); FMAN(g3.x, f3); FMAN(g3.y, f4);
            FMAN(g4.x, f4); FMAN(g4.y, f5); FMAN(g5.x, f5); FMAN(g5.y, f6); FMAN(g6.x, f6); FMAN(g6.y, f7); FMAN(g7.x, f7); FMAN(g7.y, f0);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 12) {             // 5 packed shared + 4 packed planes + 8 unpacked plane fma + 8 min3
            PK(a0, a1); PK(a1, a2); PK(a2, a3); PK(a3, a4); PK(a4, a5);
            PKN(g0, a0); PKN(g1, a1); PKN(g2, a2); PKN(g3, a3);
            FMAN(g4.x, f4); FMAN(g4.y, f5); FMAN(g5.x, f5); FMAN(g5.y, f6); FMAN(g6.x, f6); FMAN(g6.y, f7); FMAN(g7.x, f7); FMAN(g7.y, f0);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 13) {             // 10 unpacked shared + 8 packed planes + 8 min3
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0); FMA(f0, f2); FMA(f1, f3);
            PKN(g0, a0); PKN(g1, a1); PKN(g2, a2); PKN(g3, a3); PKN(g4, a4); PKN(g5, a5); PKN(g6, a6); PKN(g7, a7);
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 14) {             // 8 unpacked fma + 8 v_cmp (does a compare hide an fma?)
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n"
                         "v_cmp_lt_f32 vcc, %0, %2\n v_cmp_lt_f32 vcc, %1, %3\n v_cmp_lt_f32 vcc, %2, %0\n v_cmp_lt_f32 vcc, %3, %1" :: "v"(m0), "v"(m1), "v"(m2), "v"(m3) : "vcc");
        } else if (PATTERN == 15) {             // 8 unpacked fma + 8 shifts
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
            asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n"
                         "v_lshrrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
        } else if (PATTERN == 16) {             // 24 unpacked fma + 8 v_exp_f32 (8-cycle operations: how many fma does one hide?)
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7));
        } else if (PATTERN == 17) {             // 8 v_add_u32 (a 2-cycle integer operation) + 8 min3
            asm volatile("v_add_u32 %0, %1, %0\n v_add_u32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_add_u32 %3, %0, %3\n"
                         "v_add_u32 %0, %2, %0\n v_add_u32 %1, %3, %1\n v_add_u32 %2, %0, %2\n v_add_u32 %3, %1, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 18) {             // 8 min3 alone (reference for the rows above)
            MIN3(m0, g0); MIN3(m1, g1); MIN3(m2, g2); MIN3(m3, g3); MIN3(m4, g4); MIN3(m5, g5); MIN3(m6, g6); MIN3(m7, g7);
        } else if (PATTERN == 19) {             // 8 v_exp alone
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7));
        } else if (PATTERN == 9) {              // 16 v_fma_f32 alone
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
            FMA(f0, f1); FMA(f1, f2); FMA(f2, f3); FMA(f3, f4); FMA(f4, f5); FMA(f5, f6); FMA(f6, f7); FMA(f7, f0);
        }
    }
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7 + h0 + h1 + h2 + h3 + h4 + h5 + h6 + h7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + pad[threadIdx.x & 1];
}

template <class K>
static void run(K kernel, int waves_per_simd, int fast, int slow, const char* name, float* d_out, int cus)
{
    // 256-thread blocks = one wave per SIMD each; dynamic LDS admits exactly `waves_per_simd` blocks per CU
    const size_t lds = waves_per_simd >= 8 ? 1024 : (size_t)(160 * 1024 / waves_per_simd) - 1024;
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = cus * waves_per_simd * 4;           // four rounds of full residency
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<blocks, 256, lds>>>(d_out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kernel<<<blocks, 256, lds>>>(d_out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per loop trip and SIMD at 2.4 GHz: (time x clock) / (trips each SIMD executes = waves/SIMD x rounds x ITERS)
    const double trips = (double)waves_per_simd * 4 * ITERS;
    const double cyc = ms * 1e-3 * 2.4e9 / trips;
    printf("  %-58s %d waves/SIMD  %7.3f ms  %6.1f cycles/trip/SIMD (2.4 GHz)   [%2d fma-class + %d min3: sum %5.1f, max %5.1f]\n", name,
           waves_per_simd, ms, cyc, fast, slow, fast * 4.3 + slow * 4.2, fast * 4.3 > slow * 4.2 ? fast * 4.3 : slow * 4.2);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d_out;
    hipMalloc(&d_out, (size_t)p.multiProcessorCount * 64 * 256 * 4);
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    for (int w : {4, 8}) {
        run(k_mix<0>, w, 16, 0, "16 v_pk_fma_f32", d_out, p.multiProcessorCount);
        run(k_mix<1>, w, 0, 16, "16 v_min3_f32", d_out, p.multiProcessorCount);
        run(k_mix<9>, w, 8, 0, "16 v_fma_f32 (= 8 packed)", d_out, p.multiProcessorCount);
        run(k_mix<2>, w, 8, 8, "8 pk_fma then 8 min3 (block order, today's loop)", d_out, p.multiProcessorCount);
        run(k_mix<3>, w, 8, 8, "8 pk_fma / 8 min3 alternating, same trip", d_out, p.multiProcessorCount);
        run(k_mix<4>, w, 16, 16, "2 trips: pk_fma beside the previous trip's min3", d_out, p.multiProcessorCount);
        run(k_mix<5>, w, 13, 8, "13 pk + 8 min3, block order (the real mix)", d_out, p.multiProcessorCount);
        run(k_mix<6>, w, 13, 8, "13 pk + 8 min3, interleaved + pipelined", d_out, p.multiProcessorCount);
        run(k_mix<8>, w, 8, 8, "16 v_fma_f32 then 8 min3 (block order)", d_out, p.multiProcessorCount);
        run(k_mix<7>, w, 16, 16, "2 trips: 2 v_fma_f32 : 1 min3 alternating, pipelined", d_out, p.multiProcessorCount);
        run(k_mix<10>, w, 13, 8, "ALL unpacked: 26 v_fma_f32 + 8 min3", d_out, p.multiProcessorCount);
        run(k_mix<11>, w, 13, 8, "5 pk shared + 16 v_fma_f32 planes + 8 min3", d_out, p.multiProcessorCount);
        run(k_mix<12>, w, 13, 8, "5 pk shared + 4 pk + 8 v_fma_f32 planes + 8 min3", d_out, p.multiProcessorCount);
        run(k_mix<13>, w, 13, 8, "10 v_fma_f32 shared + 8 pk planes + 8 min3", d_out, p.multiProcessorCount);
        run(k_mix<18>, w, 0, 8, "8 min3 alone", d_out, p.multiProcessorCount);
        run(k_mix<14>, w, 4, 8, "8 v_fma_f32 + 8 v_cmp_lt_f32", d_out, p.multiProcessorCount);
        run(k_mix<15>, w, 4, 8, "8 v_fma_f32 + 8 shifts", d_out, p.multiProcessorCount);
        run(k_mix<17>, w, 4, 8, "8 v_add_u32 + 8 min3", d_out, p.multiProcessorCount);
        run(k_mix<19>, w, 0, 16, "8 v_exp_f32 alone", d_out, p.multiProcessorCount);
        run(k_mix<16>, w, 12, 16, "24 v_fma_f32 + 8 v_exp_f32", d_out, p.multiProcessorCount);
        printf("\n");
    }
    return 0;
}
