// tools/sqrt_exact.hip -- round 5: is a cheaper correctly-rounded float32 square root possible on gfx950?  The distance kernels are VALU-bound and a
// third of their per-pair instructions is mk_fsqrt_rn_ordinary (v_sqrt_f32 + the Tuckerman correction: 12 issue slots).  Candidates are compared
// with it over EVERY float in [2^-96, inf) (1.9e9 values):
//   A  y0 = v_rsq_f32(x); s = x * y0; r = fma(-s, s, x); return fma(r, 0.5 * y0, s)                    (8 slots)
//   B  s = v_sqrt_f32(x); r = fma(-s, s, x); return fma(r, 0.5 * v_rcp_f32(s), s)                      (11 slots)
//   C  A with the half of y0 folded: h = 0.5 * y0 first, s = x * y0 ... (same count; a different rounding of h is impossible: exact)
// hipcc --offload-arch=gfx950 -O3 tools/sqrt_exact.hip -o tools/sqrt_exact && tools/sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline float ref_root(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_down = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_down = __builtin_fmaf(-s_down, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    const float y = (r_down <= 0.0f) ? s_down : s;
    return (r_up > 0.0f) ? s_up : y;
}
__device__ inline float cand_a(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    float s = x * y0;
    asm volatile("" : "+v"(s));                                   // (no contraction of the product into the fma below)
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, 0.5f * y0, s);
}
__device__ inline float cand_b(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, 0.5f * __builtin_amdgcn_rcpf(s), s);
}
// a second correction step on top of A (two more fma): r2 = fma(-y, y, x); y2 = fma(r2, h, y)
__device__ inline float cand_d(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    float s = x * y0;
    asm volatile("" : "+v"(s));
    const float h = 0.5f * y0;
    const float y = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
}

__global__ void k_check(unsigned lo, unsigned long long n, unsigned long long* bad /* [3] */, unsigned* first /* [3] */)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long k = i; k < n; k += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = lo + (unsigned)k;
        const float x = __uint_as_float(bits), want = ref_root(x);
        const float got[3] = {cand_a(x), cand_b(x), cand_d(x)};
        for (int c = 0; c < 3; ++c)
            if (__float_as_uint(got[c]) != __float_as_uint(want)) {
                if (atomicAdd(&bad[c], 1ull) == 0ull) first[c] = bits;
            }
    }
}

int main()
{
    unsigned long long* bad; unsigned* first;
    CHECK(hipMalloc(&bad, 24)); CHECK(hipMalloc(&first, 12));
    CHECK(hipMemset(bad, 0, 24)); CHECK(hipMemset(first, 0, 12));
    const unsigned lo = 0x0F800000u;                               // 2^-96
    const unsigned long long n = 0x7F800000ull - lo;
    hipLaunchKernelGGL(k_check, dim3(65536), dim3(256), 0, 0, lo, n, bad, first);
    CHECK(hipDeviceSynchronize());
    unsigned long long hb[3]; unsigned hf[3];
    CHECK(hipMemcpy(hb, bad, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hf, first, 12, hipMemcpyDeviceToHost));
    const char* names[3] = {"A rsq, one correction (8 slots)", "B sqrt + rcp, one correction (11 slots)", "D rsq, two corrections (10 slots)"};
    for (int c = 0; c < 3; ++c) printf("%-44s mismatches %llu of %llu (first at bits 0x%08x)\n", names[c], hb[c], n, hf[c]);
    return 0;
}
