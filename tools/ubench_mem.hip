// tools/ubench_mem.hip -- memory-side micro-benchmarks behind the binning pre-pass design
// (hipcc --offload-arch=gfx950 -O3 tools/ubench_mem.hip -o /tmp/ubench_mem && /tmp/ubench_mem)
//   * returning atomicAdd on scattered counters (k_bin_count's one atomic per atom)
//   * the same with wave-level aggregation of equal addresses
//   * per-atom 32-byte rows read as 8 strided dwords vs 2 dwordx4
//   * scattered 16-byte stores (k_bin_fill's permutation) vs coalesced ones
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_atomic_ret(const unsigned* __restrict__ idx, unsigned* cnt, unsigned* out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = atomicAdd(&cnt[idx[i]], 1u);
}
__global__ __launch_bounds__(256) void k_atomic_noret(const unsigned* __restrict__ idx, unsigned* cnt, unsigned* out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { atomicAdd(&cnt[idx[i]], 1u); out[i] = 1u; }
}
__global__ __launch_bounds__(256) void k_no_atomic(const unsigned* __restrict__ idx, unsigned* cnt, unsigned* out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = cnt[idx[i]];
}
// four atoms per thread: four atomics in flight per lane
__global__ __launch_bounds__(256) void k_atomic_ret4(const unsigned* __restrict__ idx, unsigned* cnt, unsigned* out, int n)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uint4 ix = *reinterpret_cast<const uint4*>(idx + i);
        uint4 r;
        r.x = atomicAdd(&cnt[ix.x], 1u); r.y = atomicAdd(&cnt[ix.y], 1u);
        r.z = atomicAdd(&cnt[ix.z], 1u); r.w = atomicAdd(&cnt[ix.w], 1u);
        *reinterpret_cast<uint4*>(out + i) = r;
    }
}
__global__ __launch_bounds__(256) void k_rows_strided(const float* __restrict__ rows, float* out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += rows[(size_t)i * 8 + j];
        out[i] = s;
    }
}
__global__ __launch_bounds__(256) void k_rows_vec(const float* __restrict__ rows, float* out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float4 a = reinterpret_cast<const float4*>(rows)[(size_t)i * 2], b = reinterpret_cast<const float4*>(rows)[(size_t)i * 2 + 1];
        out[i] = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    }
}
__global__ __launch_bounds__(256) void k_scatter16(const unsigned* __restrict__ perm, const float4* __restrict__ src, float4* dst, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[perm[i]] = src[i];
}
__global__ __launch_bounds__(256) void k_gather16(const unsigned* __restrict__ perm, const float4* __restrict__ src, float4* dst, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

template <class F>
static float time_ms(F&& launch, int reps = 20)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

void write_bench();

int main()
{
    write_bench();
    const int n = 1600000;
    unsigned *idx_r, *idx_s, *cnt, *out, *perm;
    float *rows, *fout;
    float4 *src, *dst;
    CHECK(hipMalloc(&idx_r, n * 4)); CHECK(hipMalloc(&idx_s, n * 4)); CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&perm, n * 4));
    CHECK(hipMalloc(&rows, (size_t)n * 32)); CHECK(hipMalloc(&fout, n * 4));
    CHECK(hipMalloc(&src, (size_t)n * 16)); CHECK(hipMalloc(&dst, (size_t)n * 16));
    CHECK(hipMemset(rows, 0, (size_t)n * 32)); CHECK(hipMemset(src, 0, (size_t)n * 16));
    for (int ncnt : {42592, 221184, 4000000}) {
        CHECK(hipMalloc(&cnt, (size_t)ncnt * 4));
        std::vector<unsigned> h(n), hs(n), hp(n);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        for (int i = 0; i < n; ++i) { h[i] = (unsigned)(rnd() % (unsigned)ncnt); hs[i] = (unsigned)(((unsigned long long)i * ncnt) / n); hp[i] = i; }
        for (int i = n - 1; i > 0; --i) { const int j = (int)(rnd() % (unsigned)(i + 1)); std::swap(hp[i], hp[j]); }
        CHECK(hipMemcpy(idx_r, h.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(idx_s, hs.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(perm, hp.data(), n * 4, hipMemcpyHostToDevice));
        const int nb = (n + 255) / 256;
        printf("--- %d atoms, %d counters\n", n, ncnt);
        printf("returning atomicAdd, random counters    %8.1f us\n", 1e3f * time_ms([&] { k_atomic_ret<<<nb, 256>>>(idx_r, cnt, out, n); }));
        printf("returning atomicAdd x4/thread, random   %8.1f us\n", 1e3f * time_ms([&] { k_atomic_ret4<<<(nb + 3) / 4, 256>>>(idx_r, cnt, out, n); }));
        printf("non-returning atomicAdd, random         %8.1f us\n", 1e3f * time_ms([&] { k_atomic_noret<<<nb, 256>>>(idx_r, cnt, out, n); }));
        printf("plain gather of the counter, random     %8.1f us\n", 1e3f * time_ms([&] { k_no_atomic<<<nb, 256>>>(idx_r, cnt, out, n); }));
        printf("returning atomicAdd, sorted counters    %8.1f us\n", 1e3f * time_ms([&] { k_atomic_ret<<<nb, 256>>>(idx_s, cnt, out, n); }));
        CHECK(hipFree(cnt));
    }
    const int nb = (n + 255) / 256;
    printf("--- rows / permutation, %d atoms\n", n);
    printf("32-B rows as 8 strided dwords           %8.1f us\n", 1e3f * time_ms([&] { k_rows_strided<<<nb, 256>>>(rows, fout, n); }));
    printf("32-B rows as 2 dwordx4                  %8.1f us\n", 1e3f * time_ms([&] { k_rows_vec<<<nb, 256>>>(rows, fout, n); }));
    printf("16-B scatter (random permutation)       %8.1f us\n", 1e3f * time_ms([&] { k_scatter16<<<nb, 256>>>(perm, src, dst, n); }));
    printf("16-B gather  (random permutation)       %8.1f us\n", 1e3f * time_ms([&] { k_gather16<<<nb, 256>>>(perm, src, dst, n); }));
    return 0;
}

// ---- write-bandwidth ceiling of the feature tensor: plain streaming float4 stores vs the tile kernel's
//      pattern (lane = (y,z) of an 8x8 face, two float4 = 8 channels per voxel, K x-planes per wave) ----
__global__ __launch_bounds__(256) void k_write_linear(float4* dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(64) void k_write_tiles(float* out, int B, int nx, int ny, int nz, int tnx, int tny, int tnz)
{
    const int ntiles = tnx * tny * tnz;
    const unsigned lt = blockIdx.x;
    const int b = lt / ntiles;
    int t = lt - b * ntiles;
    const int tz = t % tnz; t /= tnz;
    const int ty = t % tny, tx = t / tny;
    const int ly = threadIdx.x >> 3, lz = threadIdx.x & 7;
    const int y = ty * 8 + ly, z = tz * 8 + lz;
    const size_t V = (size_t)nx * ny * nz;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = tx * 8 + k;
        if (y < ny && z < nz && x < nx) {
            float4* o = reinterpret_cast<float4*>(out + ((size_t)b * V + ((size_t)x * ny + y) * nz + z) * 8);
            o[0] = make_float4(1.f, 2.f, 3.f, (float)k);
            o[1] = make_float4(5.f, 6.f, 7.f, (float)x);
        }
    }
}

void write_bench()
{
    {
        for (int cfg = 0; cfg < 2; ++cfg) {
            const int B = cfg == 0 ? 1024 : 32, n = cfg == 0 ? 24 : 64;
            const size_t bytes = (size_t)B * n * n * n * 32;
            float* buf; CHECK(hipMalloc(&buf, bytes));
            const int tn = (n + 7) / 8;
            const float lin = time_ms([&] { k_write_linear<<<(unsigned)((bytes / 16 + 255) / 256), 256>>>((float4*)buf, bytes / 16); });
            const float til = time_ms([&] { k_write_tiles<<<B * tn * tn * tn, 64>>>(buf, B, n, n, n, tn, tn, tn); });
            printf("--- feature tensor %d x %d^3 x 8 ch (%.0f MB): linear float4 stores %.1f us = %.0f GB/s ; tile pattern %.1f us = %.0f GB/s\n",
                   B, n, bytes / 1e6, lin * 1e3, bytes / lin / 1e6, til * 1e3, bytes / til / 1e6);
            CHECK(hipFree(buf));
        }
    }
}
