// tools/store_pattern.hip -- round 5: what the STORE PATTERN of the tile kernel's epilogue costs on its own (VERDICT r4, weak #2:
// a store-only build of the tile kernel's launch shape took 0.96-1.04 ms for 2.15 GB, ~2.1 TB/s -- a second ceiling at
// ~0.33 of the HBM roofline under the VALU one?).  Same launch shape as k_voxelize_tiles on cfg2 (256 grids of 64^3 x 8
// channels: 131 072 one-wave workgroups, XCD-contiguous tile order, lane = (y, z) of an 8 x 8 face, 8 x-planes), nothing but
// the stores:
//   A  today's epilogue: per plane two 16-byte pieces per lane at its voxel (32-byte stride between lanes), non-temporal
//   B  the plane turned through 2 KB of LDS (voxelize_item_tile's epilogue): lane L writes piece L of 1 KB, every store
//      instruction covers four whole 256-byte z-rows, non-temporal
//   C / D  = A / B with plain stores
//   E  B with all 8 planes staged first (16 KB of LDS per wave) and the 16 stores issued back to back
//   F  a linear fill of the same bytes by the same number of waves (the ceiling of this launch shape)
//   G  A with `work` dependent FMAs per plane in front of the stores (stores spread over the wave's life, as in the real kernel)
// hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ inline void store16(float4* p, float4 v)
{
    if (NT) __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
    else *p = v;
}

constexpr int NX = 64, NY = 64, NZ = 64, TN = 8;            // grid, tiles per axis

__device__ inline void tile_of(unsigned lt, int& b, int& x0, int& y0, int& z0)
{
    b = (int)(lt / (TN * TN * TN));
    int t = (int)(lt % (TN * TN * TN));
    z0 = (t % TN) * 8; t /= TN;
    y0 = (t % TN) * 8; x0 = (t / TN) * 8;
}

template <bool NT, int WORK>
__global__ __launch_bounds__(64) void k_direct(float* __restrict__ out, unsigned ntiles, float seed)
{
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lt >= ntiles) return;
    int b, x0, y0, z0;
    tile_of(lt, b, x0, y0, z0);
    const int lane = threadIdx.x, y = y0 + (lane >> 3), z = z0 + (lane & 7);
    const size_t plane = (size_t)NY * NZ;
    size_t vox = (size_t)b * NX * plane + (size_t)x0 * plane + (size_t)y * NZ + z;
    float v = seed + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll 8
        for (int i = 0; i < WORK; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-7f);
        float4* o = reinterpret_cast<float4*>(out + vox * 8);
        store16<NT>(o, make_float4(v, v, v, v));
        store16<NT>(o + 1, make_float4(v, v, v, v));
        vox += plane;
    }
}

template <bool NT, bool ALL_PLANES>
__global__ __launch_bounds__(64) void k_turned(float* __restrict__ out, unsigned ntiles, float seed)
{
    __shared__ float4 tb[(ALL_PLANES ? 8 : 1) * 128];
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lt >= ntiles) return;
    int b, x0, y0, z0;
    tile_of(lt, b, x0, y0, z0);
    const int lane = threadIdx.x;
    const size_t plane = (size_t)NY * NZ;
    size_t t_vox[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int vv = s2 * 32 + (lane >> 1);
        t_vox[s2] = (size_t)b * NX * plane + (size_t)x0 * plane + (size_t)(y0 + (vv >> 3)) * NZ + (size_t)(z0 + (vv & 7));
    }
    const float v = seed + lane;
    if (ALL_PLANES) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { tb[k * 128 + 2 * lane] = make_float4(v, v, v, v); tb[k * 128 + 2 * lane + 1] = make_float4(v, v, v, v); }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                store16<NT>(reinterpret_cast<float4*>(out + (t_vox[s2] + (size_t)k * plane) * 8) + (lane & 1), tb[k * 128 + 2 * (s2 * 32 + (lane >> 1)) + (lane & 1)]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_wave_barrier();
            tb[2 * lane] = make_float4(v, v, v, v);
            tb[2 * lane + 1] = make_float4(v, v, v, v);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                store16<NT>(reinterpret_cast<float4*>(out + (t_vox[s2] + (size_t)k * plane) * 8) + (lane & 1), tb[2 * (s2 * 32 + (lane >> 1)) + (lane & 1)]);
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(64) void k_linear(float* __restrict__ out, unsigned ntiles, float seed)
{
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lt >= ntiles) return;
    float4* o = reinterpret_cast<float4*>(out) + (size_t)lt * 1024 + threadIdx.x;      // 16 KB per wave
    const float v = seed + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) store16<NT>(o + 64 * k, make_float4(v, v, v, v));
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, reps = 20;
    const unsigned ntiles = (unsigned)B * TN * TN * TN;
    const size_t bytes = (size_t)B * NX * NY * NZ * 8 * 4;
    float* out;
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(((ntiles + 7) / 8) * 8), block(64);
    auto run = [&](const char* name, auto kern) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, ntiles, 1.0f);
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, grid, block, 0, 0, out, ntiles, (float)r);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-72s avg %.4f ms  best %.4f ms  %.2f TB/s (avg)\n", name, sum / reps, best, bytes / (sum / reps) / 1e9);
    };
    printf("# %d grids of 64^3 x 8 channels, %u one-wave workgroups, %.2f GB per launch\n", B, ntiles, bytes / 1e9);
    run("A direct, 2 x 16 B per lane at a 32-B stride, non-temporal (today)", k_direct<true, 0>);
    run("C direct, plain stores", k_direct<false, 0>);
    run("B turned through LDS per plane (whole 256-B rows per instruction), nt", k_turned<true, false>);
    run("D turned through LDS per plane, plain stores", k_turned<false, false>);
    run("E turned, all 8 planes staged (16 KB LDS), 16 stores back to back, nt", k_turned<true, true>);
    run("F linear fill, 16 KB per wave, nt", k_linear<true>);
    run("F' linear fill, plain", k_linear<false>);
    run("G direct nt + 256 dependent FMAs per plane (~2 k VALU per wave)", k_direct<true, 256>);
    run("G' direct nt + 1024 dependent FMAs per plane (~8 k VALU per wave: the tile kernel's length)", k_direct<true, 1024>);
    CHECK(hipFree(out));
    return 0;
}
