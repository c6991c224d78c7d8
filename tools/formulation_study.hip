// tools/formulation_study.hip -- round 3: measurements that settle the formulation of the cfg2 / cfg4 tile kernel
// (VERDICT r2 "next round" item 1).  Stand-alone: hipcc --offload-arch=gfx950 -O3 formulation_study.hip -o formulation_study
//
//   A  issue rates of the instructions a pair test can be built from, with the clock the chip really ran at
//      (shader cycles from s_memtime against the 100 MHz s_memrealtime): which operations are full rate on the
//      SIMD-32 of CDNA4 and which are not
//   B  does a VALU instruction whose EXEC mask has one 32-lane half (or 16-lane quarter) empty cost less?
//      (the "y-reach buckets by half-wave exec skipping" option)
//   C  ds_min_u32 (no return) rates: conflict-free, random over a 16 KB tile, the address pattern of a z-run scatter,
//      one address; ds_write_b32 beside it               (the "LDS-resident scatter" option)
//   D  an UPPER BOUND for the LDS-resident scatter on cfg2's density: a workgroup owns the 8x8x8 x 8-channel keys of a
//      tile in LDS, lane = (entry, x-y column) work item with its exact z run ALREADY enumerated on the host (the
//      enumeration, the cull and the sort are free here), ds_min_u32 of the bit pattern of d^2 * w per in-range pair,
//      the tile kernel's epilogue and stores.  Useful pairs / s / CU against the 6.6 G of the round-2 kernel.
//
// Prints a report; round 3 copied it to profiles/r3_formulation_study.txt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
struct Clocks { unsigned long long cyc, real; };

__device__ inline unsigned long long rd_cycles() { return __builtin_readcyclecounter(); }                 // s_memtime: shader clock
__device__ inline unsigned long long rd_real() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }   // 100 MHz

#define OP8(fmt) \
    asm volatile(fmt : "+v"(a0) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a2) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a4) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a5) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a6) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a7) : "v"(b), "v"(c));

// MASKSEL: 0 all lanes, 1 lanes 0-31, 2 lanes 0-15, 3 even lanes, 4 lanes 32-63, 5 lanes 0-47
template <int MASKSEL> __device__ inline bool lane_on(int lane)
{
    return MASKSEL == 0 ? true : MASKSEL == 1 ? lane < 32 : MASKSEL == 2 ? lane < 16 : MASKSEL == 3 ? (lane & 1) == 0
         : MASKSEL == 4 ? lane >= 32 : lane < 48;
}

#define UB_KERNEL(name, body)                                                                                   \
    template <int MASKSEL> __global__ __launch_bounds__(256) void name(float* out, float seed, Clocks* clk)      \
    {                                                                                                           \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = seed * 0.5f, c = seed * 0.25f;                                                                \
        const unsigned long long c0 = rd_cycles(), r0 = rd_real();                                              \
        if (lane_on<MASKSEL>(threadIdx.x & 63)) {                                                               \
            for (int i = 0; i < ITERS; ++i) { body }                                                            \
        }                                                                                                       \
        const unsigned long long c1 = rd_cycles(), r1 = rd_real();                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                           \
    }

UB_KERNEL(k_fma, OP8("v_fma_f32 %0, %1, %2, %0") OP8("v_fma_f32 %0, %1, %2, %0"))
UB_KERNEL(k_fmac, OP8("v_fmac_f32 %0, %1, %2") OP8("v_fmac_f32 %0, %1, %2"))
UB_KERNEL(k_add, OP8("v_add_f32 %0, %1, %0") OP8("v_add_f32 %0, %2, %0"))
UB_KERNEL(k_min, OP8("v_min_f32 %0, %1, %0") OP8("v_min_f32 %0, %2, %0"))
UB_KERNEL(k_max, OP8("v_max_f32 %0, %1, %0") OP8("v_max_f32 %0, %2, %0"))
UB_KERNEL(k_min3, OP8("v_min3_f32 %0, %1, %2, %0") OP8("v_min3_f32 %0, %2, %1, %0"))
UB_KERNEL(k_med3, OP8("v_med3_f32 %0, %1, %2, %0") OP8("v_med3_f32 %0, %2, %1, %0"))
UB_KERNEL(k_minu, OP8("v_min_u32 %0, %1, %0") OP8("v_min_u32 %0, %2, %0"))
UB_KERNEL(k_addu, OP8("v_add_u32 %0, %1, %0") OP8("v_add_u32 %0, %2, %0"))
UB_KERNEL(k_and, OP8("v_and_b32 %0, %1, %0") OP8("v_xor_b32 %0, %2, %0"))
UB_KERNEL(k_lshl, OP8("v_lshlrev_b32 %0, 1, %0") OP8("v_lshrrev_b32 %0, 1, %0"))
UB_KERNEL(k_mov, OP8("v_mov_b32 %0, %1") OP8("v_mov_b32 %0, %2"))
UB_KERNEL(k_cnd, OP8("v_cndmask_b32 %0, %1, %0, vcc") OP8("v_cndmask_b32 %0, %2, %0, vcc"))
UB_KERNEL(k_cmp, OP8("v_cmp_lt_f32 vcc, %1, %0") OP8("v_cmp_lt_f32 vcc, %2, %0"))
UB_KERNEL(k_fma_min3, OP8("v_fma_f32 %0, %1, %2, %0") OP8("v_min3_f32 %0, %2, %1, %0"))     // the pair test's mix, unpacked
UB_KERNEL(k_exp, OP8("v_exp_f32 %0, %0") OP8("v_exp_f32 %0, %0"))

#define UBP_KERNEL(name, body)                                                                                  \
    template <int MASKSEL> __global__ __launch_bounds__(256) void name(float* out, float seed, Clocks* clk)      \
    {                                                                                                           \
        typedef float v2 __attribute__((ext_vector_type(2)));                                                   \
        v2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        v2 b = {seed * 0.5f, seed}, c = {seed * 0.25f, seed};                                                   \
        const unsigned long long c0 = rd_cycles(), r0 = rd_real();                                              \
        if (lane_on<MASKSEL>(threadIdx.x & 63)) {                                                               \
            for (int i = 0; i < ITERS; ++i) { body }                                                            \
        }                                                                                                       \
        const unsigned long long c1 = rd_cycles(), r1 = rd_real();                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }                   \
        v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                                        \
    }
UBP_KERNEL(k_pk_fma, OP8("v_pk_fma_f32 %0, %1, %2, %0") OP8("v_pk_fma_f32 %0, %1, %2, %0"))
UBP_KERNEL(k_pk_add, OP8("v_pk_add_f32 %0, %1, %0") OP8("v_pk_add_f32 %0, %2, %0"))

static float* d_out;
static Clocks* d_clk;
static int g_blocks;

template <class K>
static void run_valu(K kernel, int instr_per_iter, const char* name, double* ms_out = nullptr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<g_blocks, 256>>>(d_out, 1.0f, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kernel<<<g_blocks, 256>>>(d_out, 1.0f, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    Clocks c;
    hipMemcpy(&c, d_clk, sizeof c, hipMemcpyDeviceToHost);
    const double mhz = c.real ? (double)c.cyc / ((double)c.real / 100.0) : 0.0;           // shader MHz while the kernel ran
    const double winstr = (double)g_blocks * 4 * ITERS * instr_per_iter;
    const double gps = winstr / (ms * 1e-3) / 1e9;
    const double cyc_real = 1024.0 * (mhz * 1e-3) / gps;
    printf("  %-34s %7.3f ms  %8.1f G wave-instr/s  clock %4.0f MHz  %5.2f cycles/wave-instr/SIMD (%.2f at a nominal 2.4 GHz)\n",
           name, ms, gps, mhz, cyc_real, 1024.0 * 2.4 / gps);
    if (ms_out) *ms_out = ms;
}

// ---------------------------------------------------------------------------------------------------- C: LDS atomics
// PATTERN: 0 conflict-free (lane), 1 random in the 16 KB tile, 2 z-run scatter (lane -> (column, channel), z steps),
//          3 one address, 4 stride 8 dwords
template <int PATTERN, int OP>   // OP: 0 ds_min_u32 (no return), 1 ds_write_b32, 2 ds_min_rtn_u32
__global__ __launch_bounds__(256) void k_lds(unsigned* out, unsigned seed, Clocks* clk)
{
    __shared__ unsigned tile[4096];                              // 16 KB: 8 channels x 512 voxels
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0x7f800000u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned rnd = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
    unsigned acc = 0;
    const unsigned long long c0 = rd_cycles(), r0 = rd_real();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned idx;
            rnd = rnd * 1664525u + 1013904223u;
            if (PATTERN == 0) idx = (unsigned)(threadIdx.x + 64 * j + i) & 4095u;
            else if (PATTERN == 1) idx = (rnd >> 12) & 4095u;
            else if (PATTERN == 2) { const unsigned col = (rnd >> 10) & 63u, ch = (rnd >> 20) & 7u; idx = ch * 512u + col * 8u + (unsigned)((i + j) & 7); }
            else if (PATTERN == 3) idx = 17u;
            else idx = (unsigned)(lane * 8 + j) & 4095u;
            const unsigned val = (rnd >> 3) | 0x3f000000u;
            if (OP == 0) (void)__hip_atomic_fetch_min(&tile[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) tile[idx] = val;
            else acc += __hip_atomic_fetch_min(&tile[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const unsigned long long c1 = rd_cycles(), r1 = rd_real();
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = tile[threadIdx.x] + acc;
}

template <class K>
static void run_lds(K kernel, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<g_blocks, 256>>>((unsigned*)d_out, 1u, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kernel<<<g_blocks, 256>>>((unsigned*)d_out, 1u, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    Clocks c;
    hipMemcpy(&c, d_clk, sizeof c, hipMemcpyDeviceToHost);
    const double mhz = c.real ? (double)c.cyc / ((double)c.real / 100.0) : 0.0;
    const double winstr = (double)g_blocks * 4 * ITERS * 8;
    const double gps = winstr / (ms * 1e-3) / 1e9;
    printf("  %-44s %7.3f ms  %7.1f G wave-instr/s  clock %4.0f MHz  %6.2f cycles/wave-instr/CU  %5.1f lanes/clk/CU\n",
           name, ms, gps, mhz, 256.0 * (mhz * 1e-3) / gps, 64.0 * gps / (256.0 * mhz * 1e-3));
}

// ------------------------------------------------------------------------------------------- D: scatter upper bound
struct Item { float dxy2, ez, w; unsigned code; };        // code: column (6 bits) | channel << 6 | zlo << 9 | zhi << 12

__device__ inline float occ_from_q(float q)               // the tile kernel's epilogue arithmetic (same cost)
{
    const float u = __builtin_amdgcn_rcpf(q);
    const float u2 = u * u, x = u2 * u2 * u2;
    const float big = 1.f - __builtin_amdgcn_exp2f(-x * 1.44269504f);
    const float small = x * (1.f - x * (0.5f - x * 0.16666667f));
    return x >= 0.015625f ? big : small;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_scatter_tile(const Item* __restrict__ items, const unsigned* __restrict__ list_start,
                                                              int nlists, float* __restrict__ out, unsigned out_tiles, unsigned long long* useful)
{
    __shared__ unsigned q[8 * 512];                        // [channel][x][y][z] keys: bit patterns of d^2 * w
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += WAVES * 64) q[i] = 0x7f800000u;
    __syncthreads();
    const int l = blockIdx.x % nlists;
    const unsigned s = list_start[l], e = list_start[l + 1];
    unsigned long long mine = 0;
    for (unsigned i = s + tid; i < e; i += WAVES * 64) {
        const Item it = items[i];
        const unsigned col = it.code & 63u, ch = (it.code >> 6) & 7u;
        const int zlo = (it.code >> 9) & 7, zhi = (it.code >> 12) & 7;
        unsigned* qp = q + ch * 512u + col * 8u;
        for (int z = zlo; z <= zhi; ++z) {                  // the exact in-range run of this (entry, column)
            const float dz = (float)z - it.ez;
            const float t = __builtin_fmaf(dz, dz, it.dxy2) * it.w;
            (void)__hip_atomic_fetch_min(qp + z, __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        mine += (unsigned long long)(zhi - zlo + 1);
    }
    __syncthreads();
    float* o = out + (size_t)(blockIdx.x % out_tiles) * 4096;
    for (int v = tid; v < 512; v += WAVES * 64) {
        float f[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = occ_from_q(__uint_as_float(q[c * 512 + v]));
        float4* p = reinterpret_cast<float4*>(o + v * 8);
        p[0] = make_float4(f[0], f[1], f[2], f[3]);
        p[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (useful && blockIdx.x < (unsigned)nlists) atomicAdd(useful, mine);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  nominal clock=%d kHz\n\n", p.name, p.multiProcessorCount, p.clockRate);
    g_blocks = p.multiProcessorCount * 8;                  // 32 waves per CU
    CHECK(hipMalloc(&d_out, (size_t)g_blocks * 256 * 4));
    CHECK(hipMalloc(&d_clk, sizeof(Clocks)));

    printf("A. issue rates (32 waves per CU, 8 independent chains per wave) and the clock they ran at\n");
    run_valu(k_fma<0>, 16, "v_fma_f32");
    run_valu(k_fmac<0>, 16, "v_fmac_f32 (VOP2)");
    run_valu(k_add<0>, 16, "v_add_f32");
    run_valu(k_pk_fma<0>, 16, "v_pk_fma_f32 (2 per lane)");
    run_valu(k_pk_add<0>, 16, "v_pk_add_f32 (2 per lane)");
    run_valu(k_min<0>, 16, "v_min_f32");
    run_valu(k_max<0>, 16, "v_max_f32");
    run_valu(k_min3<0>, 16, "v_min3_f32");
    run_valu(k_med3<0>, 16, "v_med3_f32");
    run_valu(k_minu<0>, 16, "v_min_u32");
    run_valu(k_addu<0>, 16, "v_add_u32");
    run_valu(k_and<0>, 16, "v_and_b32 / v_xor_b32");
    run_valu(k_lshl<0>, 16, "v_lshlrev_b32 / v_lshrrev_b32");
    run_valu(k_mov<0>, 16, "v_mov_b32");
    run_valu(k_cnd<0>, 16, "v_cndmask_b32");
    run_valu(k_cmp<0>, 16, "v_cmp_lt_f32");
    run_valu(k_exp<0>, 16, "v_exp_f32");
    run_valu(k_fma_min3<0>, 16, "v_fma_f32 + v_min3_f32 alternating");

    printf("\nB. the same loops under a partial EXEC mask (time relative to all 64 lanes)\n");
    struct { const char* name; int sel; } masks[] = {{"lanes 0-31", 1}, {"lanes 32-63", 4}, {"lanes 0-15", 2}, {"lanes 0-47", 5}, {"even lanes", 3}};
    double full[3], part;
    run_valu(k_fma<0>, 16, "v_fma_f32      all lanes", &full[0]);
    run_valu(k_fma<1>, 16, "v_fma_f32      lanes 0-31", &part);   printf("      -> %.2f of the full-wave time\n", part / full[0]);
    run_valu(k_fma<4>, 16, "v_fma_f32      lanes 32-63", &part);  printf("      -> %.2f\n", part / full[0]);
    run_valu(k_fma<2>, 16, "v_fma_f32      lanes 0-15", &part);   printf("      -> %.2f\n", part / full[0]);
    run_valu(k_fma<5>, 16, "v_fma_f32      lanes 0-47", &part);   printf("      -> %.2f\n", part / full[0]);
    run_valu(k_fma<3>, 16, "v_fma_f32      even lanes", &part);   printf("      -> %.2f\n", part / full[0]);
    run_valu(k_min3<0>, 16, "v_min3_f32     all lanes", &full[1]);
    run_valu(k_min3<1>, 16, "v_min3_f32     lanes 0-31", &part);  printf("      -> %.2f\n", part / full[1]);
    run_valu(k_min3<2>, 16, "v_min3_f32     lanes 0-15", &part);  printf("      -> %.2f\n", part / full[1]);
    run_valu(k_min3<3>, 16, "v_min3_f32     even lanes", &part);  printf("      -> %.2f\n", part / full[1]);
    run_valu(k_pk_fma<0>, 16, "v_pk_fma_f32   all lanes", &full[2]);
    run_valu(k_pk_fma<1>, 16, "v_pk_fma_f32   lanes 0-31", &part); printf("      -> %.2f\n", part / full[2]);
    run_valu(k_pk_fma<2>, 16, "v_pk_fma_f32   lanes 0-15", &part); printf("      -> %.2f\n", part / full[2]);
    (void)masks;

    printf("\nC. LDS atomics into a 16 KB tile of keys (8 workgroups of 4 waves per CU)\n");
    run_lds(k_lds<0, 0>, "ds_min_u32  conflict-free (address = lane)");
    run_lds(k_lds<1, 0>, "ds_min_u32  random address in the tile");
    run_lds(k_lds<2, 0>, "ds_min_u32  z-run scatter pattern");
    run_lds(k_lds<4, 0>, "ds_min_u32  stride 8 dwords (8-way conflict)");
    run_lds(k_lds<3, 0>, "ds_min_u32  one address");
    run_lds(k_lds<0, 2>, "ds_min_rtn_u32 conflict-free");
    run_lds(k_lds<1, 2>, "ds_min_rtn_u32 random");
    run_lds(k_lds<0, 1>, "ds_write_b32 conflict-free");
    run_lds(k_lds<1, 1>, "ds_write_b32 random");

    // ---- D: work items of NLISTS independent tiles at cfg2's density (0.1 atoms / A^3, 1.041 entries per atom, the
    //         generator's radii and channel probabilities: tests/synth.py / SURVEY.md section 8d)
    printf("\nD. LDS-resident scatter, upper bound (work items enumerated on the host; 131072 tiles = one 256-grid step)\n");
    const int NLISTS = 64;
    std::mt19937 rng(2);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const float radii[5] = {1.1f, 1.7f, 1.55f, 1.52f, 1.8f}, rp[5] = {.5f, .3f, .08f, .11f, .01f};
    const float cp[7] = {.3f, .05f, .1f, .05f, .02f, .02f, .001f};
    std::vector<Item> items;
    std::vector<unsigned> starts(1, 0u);
    double pairs = 0, entries = 0;
    for (int l = 0; l < NLISTS; ++l) {
        const int natoms = (int)std::lround(0.1 * 18 * 18 * 18);
        for (int a = 0; a < natoms; ++a) {
            const float x = U(rng) * 18.f - 5.f, y = U(rng) * 18.f - 5.f, z = U(rng) * 18.f - 5.f;    // tile voxels at 0..7
            float r = U(rng); int cls = 0; while (cls < 4 && r >= rp[cls]) { r -= rp[cls]; ++cls; }
            const float w = 1.f / (radii[cls] * radii[cls]);
            bool chan[8];
            for (int c = 0; c < 7; ++c) chan[c] = U(rng) < cp[c];
            chan[7] = cls != 0;
            for (int c = 0; c < 8; ++c) {
                if (!chan[c]) continue;
                bool any = false;
                for (int ix = 0; ix < 8; ++ix) for (int iy = 0; iy < 8; ++iy) {
                    const float dxy2 = (ix - x) * (ix - x) + (iy - y) * (iy - y);
                    int zlo = 8, zhi = -1;
                    for (int iz = 0; iz < 8; ++iz) if (dxy2 + (iz - z) * (iz - z) < 25.f) { zlo = std::min(zlo, iz); zhi = std::max(zhi, iz); }
                    if (zhi < 0) continue;
                    items.push_back({dxy2, z, w, (unsigned)(ix * 8 + iy) | ((unsigned)c << 6) | ((unsigned)zlo << 9) | ((unsigned)zhi << 12)});
                    pairs += zhi - zlo + 1; any = true;
                }
                entries += any;
            }
        }
        // the order a sort would leave them in is free to choose: shuffle so that a wave's lanes hold unrelated items
        std::shuffle(items.begin() + starts.back(), items.end(), rng);
        starts.push_back((unsigned)items.size());
    }
    printf("  per tile: %.0f entries within reach, %.0f (entry, column) work items, %.0f in-range pairs = %.1f per voxel (z run %.2f per item)\n",
           entries / NLISTS, (double)items.size() / NLISTS, pairs / NLISTS, pairs / NLISTS / 512, pairs / items.size());
    Item* d_items; unsigned* d_starts; float* d_tiles; unsigned long long* d_useful;
    const unsigned OUT_TILES = 32768;                      // 512 MB of output, written four times over: HBM stores as in the real step
    CHECK(hipMalloc(&d_items, items.size() * sizeof(Item)));
    CHECK(hipMalloc(&d_starts, starts.size() * 4));
    CHECK(hipMalloc(&d_tiles, (size_t)OUT_TILES * 4096 * 4));
    CHECK(hipMalloc(&d_useful, 8));
    CHECK(hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice));
    const unsigned TILES = 131072;
    auto run_scatter = [&](auto kernel, int waves, const char* name) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipMemset(d_useful, 0, 8);
        kernel<<<TILES, waves * 64>>>(d_items, d_starts, NLISTS, d_tiles, OUT_TILES, d_useful);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            kernel<<<TILES, waves * 64>>>(d_items, d_starts, NLISTS, d_tiles, OUT_TILES, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        const double useful_pairs = pairs / NLISTS * TILES;
        printf("  %-30s %7.3f ms per 131072 tiles  -> %6.2f G useful pairs/s/CU  (round-2 tile kernel: 2.17 ms, 6.6 G/s/CU);"
               " as the whole tile kernel: %.3f of the HBM roofline\n",
               name, best, useful_pairs / (best * 1e-3) / 256 / 1e9, 2710.7e6 / (best * 1e-3) / 8e12);
    };
    run_scatter(k_scatter_tile<1>, 1, "1 wave per tile");
    run_scatter(k_scatter_tile<2>, 2, "2 waves per tile");
    run_scatter(k_scatter_tile<4>, 4, "4 waves per tile");
    printf("  (not in this number: finding the entries of a tile, the (entry, column) enumeration -- %.0f column tests per tile --, the z runs)\n",
           entries / NLISTS * 64);
    return 0;
}
