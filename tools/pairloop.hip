// tools/pairloop.hip -- round 3: the pair loop of voxelize_tile (kernels.h) in isolation: entry pairs broadcast-read from
// LDS, d0 per lane, one fma per plane, min3 into the class minima -- at the tile kernel's occupancy (64-thread workgroups,
// ~10 KB of LDS each: 4 per SIMD).  Which form of the SAME arithmetic runs fastest?
//   0  today's loop: packed (v_pk_*), loads at the top of every trip
//   1  the same, unpacked (v_fma_f32 / v_sub_f32 ...)
//   2  unpacked, the next pair's three LDS reads issued before this pair's arithmetic (two trips unrolled: no copies)
//   3  packed, prefetched the same way
//   4  unpacked, FOUR entries per trip (two pairs' reads up front, min3 twice per plane)
// Build: hipcc --offload-arch=gfx950 -O3 pairloop.hip -o pairloop
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int K = 8, ESTRIDE = 644, NPAIRS = 300, REPS = 64;
__device__ constexpr float slope(int k) { return -2.f * ((float)k - 3.5f); }

template <int FORM>
__global__ __launch_bounds__(64) void k_pairs(float* out, const float* in)
{
    __shared__ __attribute__((aligned(16))) float sxyz[3 * ESTRIDE];
    __shared__ float pad[2560 - 3 * ESTRIDE];                 // 10 KB in all: the tile kernel's tier-0 footprint
    for (int i = threadIdx.x; i < 3 * ESTRIDE; i += 64) sxyz[i] = in[i];
    if (in[0] < -1e30f) pad[threadIdx.x] = 1.f;
    __syncthreads();
    const int lane = threadIdx.x;
    const float Y = (float)(lane >> 3) - 3.5f, Z = (float)(lane & 7) - 3.5f;
    const v2 Y2 = {Y, Y}, Z2 = {Z, Z};
    float m[K];
#pragma unroll
    for (int k = 0; k < K; ++k) m[k] = 3.0e38f;
    for (int r = 0; r < REPS; ++r) {
        const float* e = sxyz + 2 * (r & 3);
        const float* const e_end = e + 2 * NPAIRS;
        if (FORM == 0) {
#pragma clang loop unroll(disable)
            for (; e != e_end; e += 2) {
                const v2 px = *(const v2*)e, py = *(const v2*)(e + ESTRIDE), pz = *(const v2*)(e + 2 * ESTRIDE);
                const v2 dy = Y2 - py, dz = Z2 - pz;
                const v2 d0 = __builtin_elementwise_fma(px, px, __builtin_elementwise_fma(dy, dy, dz * dz));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const v2 sl = {slope(k), slope(k)};
                    const v2 g = __builtin_elementwise_fma(sl, px, d0);
                    m[k] = __builtin_fminf(__builtin_fminf(m[k], g.x), g.y);
                }
            }
        } else if (FORM == 1) {
#pragma clang loop unroll(disable)
            for (; e != e_end; e += 2) {
                const float ax = e[0], bx = e[1], ay = e[ESTRIDE], by = e[ESTRIDE + 1], az = e[2 * ESTRIDE], bz = e[2 * ESTRIDE + 1];
                const float ady = Y - ay, bdy = Y - by, adz = Z - az, bdz = Z - bz;
                const float a0 = __builtin_fmaf(ax, ax, __builtin_fmaf(ady, ady, adz * adz));
                const float b0 = __builtin_fmaf(bx, bx, __builtin_fmaf(bdy, bdy, bdz * bdz));
#pragma unroll
                for (int k = 0; k < K; ++k)
                    m[k] = __builtin_fminf(__builtin_fminf(m[k], __builtin_fmaf(slope(k), ax, a0)), __builtin_fmaf(slope(k), bx, b0));
            }
        } else if (FORM == 2 || FORM == 3) {
            v2 px = *(const v2*)e, py = *(const v2*)(e + ESTRIDE), pz = *(const v2*)(e + 2 * ESTRIDE);
#pragma clang loop unroll(disable)
            for (; e != e_end; e += 4) {                      // NPAIRS is even
                const v2 qx = *(const v2*)(e + 2), qy = *(const v2*)(e + 2 + ESTRIDE), qz = *(const v2*)(e + 2 + 2 * ESTRIDE);
                {
                    const v2 dy = Y2 - py, dz = Z2 - pz;
                    const v2 d0 = __builtin_elementwise_fma(px, px, __builtin_elementwise_fma(dy, dy, dz * dz));
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const v2 sl = {slope(k), slope(k)};
                        const v2 g = __builtin_elementwise_fma(sl, px, d0);
                        m[k] = __builtin_fminf(__builtin_fminf(m[k], g.x), g.y);
                    }
                }
                px = *(const v2*)(e + 4); py = *(const v2*)(e + 4 + ESTRIDE); pz = *(const v2*)(e + 4 + 2 * ESTRIDE);   // (reads past the end stay inside the array)
                {
                    const v2 dy = Y2 - qy, dz = Z2 - qz;
                    const v2 d0 = __builtin_elementwise_fma(qx, qx, __builtin_elementwise_fma(dy, dy, dz * dz));
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const v2 sl = {slope(k), slope(k)};
                        const v2 g = __builtin_elementwise_fma(sl, qx, d0);
                        m[k] = __builtin_fminf(__builtin_fminf(m[k], g.x), g.y);
                    }
                }
            }
        } else if (FORM == 4) {
#pragma clang loop unroll(disable)
            for (; e != e_end; e += 4) {
                const v2 px = *(const v2*)e, py = *(const v2*)(e + ESTRIDE), pz = *(const v2*)(e + 2 * ESTRIDE);
                const v2 qx = *(const v2*)(e + 2), qy = *(const v2*)(e + 2 + ESTRIDE), qz = *(const v2*)(e + 2 + 2 * ESTRIDE);
                const v2 dy = Y2 - py, dz = Z2 - pz, ey = Y2 - qy, ez = Z2 - qz;
                const v2 d0 = __builtin_elementwise_fma(px, px, __builtin_elementwise_fma(dy, dy, dz * dz));
                const v2 e0 = __builtin_elementwise_fma(qx, qx, __builtin_elementwise_fma(ey, ey, ez * ez));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const v2 sl = {slope(k), slope(k)};
                    const v2 g = __builtin_elementwise_fma(sl, px, d0), h = __builtin_elementwise_fma(sl, qx, e0);
                    m[k] = __builtin_fminf(__builtin_fminf(m[k], g.x), g.y);
                    m[k] = __builtin_fminf(__builtin_fminf(m[k], h.x), h.y);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += m[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class Kn>
static void run(Kn kernel, const char* name, float* d_out, const float* d_in, int cus)
{
    const int blocks = cus * 16 * 8;                          // 16 resident per CU (4 per SIMD), eight rounds
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<blocks, 64>>>(d_out, d_in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kernel<<<blocks, 64>>>(d_out, d_in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double trips_per_simd = 4.0 * 8 * REPS * NPAIRS;    // pair trips (two entries x eight planes) per SIMD
    printf("  %-72s %7.3f ms  %6.1f cycles per pair trip and SIMD (2.4 GHz) = %4.2f per (voxel-plane, entry) wave-test\n", name, ms,
           ms * 1e-3 * 2.4e9 / trips_per_simd, ms * 1e-3 * 2.4e9 / trips_per_simd / 16);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float *d_out, *d_in;
    hipMalloc(&d_out, (size_t)p.multiProcessorCount * 16 * 8 * 64 * 4);
    hipMalloc(&d_in, 3 * ESTRIDE * 4);
    float h[3 * ESTRIDE];
    for (int i = 0; i < 3 * ESTRIDE; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 17.f - 8.5f;
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    printf("pair loop in isolation, 4 waves per SIMD (%s)\n", argc > 1 ? argv[1] : "default build");
    run(k_pairs<0>, "0 packed, loads at the top of the trip (the kernel today)", d_out, d_in, p.multiProcessorCount);
    run(k_pairs<1>, "1 unpacked", d_out, d_in, p.multiProcessorCount);
    run(k_pairs<3>, "3 packed, next pair's reads issued before this pair's arithmetic", d_out, d_in, p.multiProcessorCount);
    run(k_pairs<4>, "4 packed, four entries per trip", d_out, d_in, p.multiProcessorCount);
    return 0;
}
