// tools/ubench.hip -- VALU / LDS issue-rate micro-benchmarks for gfx950, used to choose the inner
// loop of k_voxelize_tiles.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
// Prints G wave-instructions/s per op and the per-SIMD cycles per wave-instruction (assuming 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int ITERS = 4096;

#define UB_KERNEL(name, body)                                                          \
    __global__ __launch_bounds__(256) void name(float* out, float seed)               \
    {                                                                                  \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = seed * 0.5f, c = seed * 0.25f;                                       \
        for (int i = 0; i < ITERS; ++i) { body }                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;  \
    }

// 8 independent chains per iteration x 2 = 16 instructions / iteration / wave
#define OP8(fmt) \
    asm volatile(fmt : "+v"(a0) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a2) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a4) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a5) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a6) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a7) : "v"(b), "v"(c));

UB_KERNEL(k_fma, OP8("v_fma_f32 %0, %1, %2, %0") OP8("v_fma_f32 %0, %1, %2, %0"))
UB_KERNEL(k_mul, OP8("v_mul_f32 %0, %1, %0") OP8("v_mul_f32 %0, %2, %0"))
UB_KERNEL(k_add, OP8("v_add_f32 %0, %1, %0") OP8("v_add_f32 %0, %2, %0"))
UB_KERNEL(k_min_f32, OP8("v_min_f32 %0, %1, %0") OP8("v_min_f32 %0, %2, %0"))
UB_KERNEL(k_min_u32, OP8("v_min_u32 %0, %1, %0") OP8("v_min_u32 %0, %2, %0"))
UB_KERNEL(k_min3_u32, OP8("v_min3_u32 %0, %1, %2, %0") OP8("v_min3_u32 %0, %2, %1, %0"))
UB_KERNEL(k_min3_f32, OP8("v_min3_f32 %0, %1, %2, %0") OP8("v_min3_f32 %0, %2, %1, %0"))
UB_KERNEL(k_cmp_cnd, OP8("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %2, %0, vcc") )   // 16 instrs too
UB_KERNEL(k_exp, OP8("v_exp_f32 %0, %0") OP8("v_exp_f32 %0, %0"))
UB_KERNEL(k_rcp, OP8("v_rcp_f32 %0, %0") OP8("v_rcp_f32 %0, %0"))

// packed f32: each chain is a register PAIR
#define UBP_KERNEL(name, body)                                                         \
    __global__ __launch_bounds__(256) void name(float* out, float seed)               \
    {                                                                                  \
        typedef float v2 __attribute__((ext_vector_type(2)));                          \
        v2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        v2 b = {seed * 0.5f, seed}, c = {seed * 0.25f, seed};                          \
        for (int i = 0; i < ITERS; ++i) { body }                                       \
        v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                  \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                               \
    }
UBP_KERNEL(k_pk_fma, OP8("v_pk_fma_f32 %0, %1, %2, %0") OP8("v_pk_fma_f32 %0, %1, %2, %0"))
UBP_KERNEL(k_pk_add, OP8("v_pk_add_f32 %0, %1, %0") OP8("v_pk_add_f32 %0, %2, %0"))
UBP_KERNEL(k_pk_mul, OP8("v_pk_mul_f32 %0, %1, %0") OP8("v_pk_mul_f32 %0, %2, %0"))

// LDS broadcast read (uniform address) of 16 bytes + one dependent VALU op each
__global__ __launch_bounds__(256) void k_lds_b128_uniform(float* out, float seed)
{
    __shared__ float4 buf[256];
    buf[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + threadIdx.x);
    __syncthreads();
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 e = buf[(i + j * 7) & 255];
            acc += e.x + e.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class K>
static double run(K kernel, int blocks, int instr_per_iter, const char* name, float* d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<blocks, 256>>>(d_out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kernel<<<blocks, 256>>>(d_out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double winstr = waves * ITERS * instr_per_iter;
    const double gps = winstr / (ms * 1e-3) / 1e9;
    // 256 CUs x 4 SIMDs, assume 2.4 GHz: cycles per wave-instruction per SIMD
    const double cyc = 1024.0 * 2.4 / gps;
    printf("%-22s %8.3f ms  %9.1f G wave-instr/s   %5.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, gps, cyc);
    return gps;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int blocks = p.multiProcessorCount * 8;      // 32 waves per CU
    float* d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run(k_fma, blocks, 16, "v_fma_f32", d_out);
    run(k_mul, blocks, 16, "v_mul_f32", d_out);
    run(k_add, blocks, 16, "v_add_f32", d_out);
    run(k_min_f32, blocks, 16, "v_min_f32", d_out);
    run(k_min_u32, blocks, 16, "v_min_u32", d_out);
    run(k_min3_u32, blocks, 16, "v_min3_u32", d_out);
    run(k_min3_f32, blocks, 16, "v_min3_f32", d_out);
    run(k_cmp_cnd, blocks, 16, "v_cmp+v_cndmask", d_out);
    run(k_exp, blocks, 16, "v_exp_f32", d_out);
    run(k_rcp, blocks, 16, "v_rcp_f32", d_out);
    run(k_pk_fma, blocks, 16, "v_pk_fma_f32", d_out);
    run(k_pk_add, blocks, 16, "v_pk_add_f32", d_out);
    run(k_pk_mul, blocks, 16, "v_pk_mul_f32", d_out);
    run(k_lds_b128_uniform, blocks, 16, "ds_read_b128 uniform", d_out);
    hipFree(d_out);
    return 0;
}
