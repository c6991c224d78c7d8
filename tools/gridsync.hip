// tools/gridsync.hip -- round 3: what a grid-wide barrier costs on gfx950 (for the one-launch single-molecule path):
//   * empty kernel, normal launch vs hipLaunchCooperativeKernel, one synchronous call at a time
//   * kernel with NB grid barriers (counter + generation flag, one atomic per workgroup), 54 / 512 / 1024 workgroups of 256
//   * the same with a two-level counter (16 sub-counters)
//   * two dependent empty kernels back to back (what a launch boundary costs instead)
// Build: hipcc --offload-arch=gfx950 -O3 gridsync.hip -o gridsync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned* gen, unsigned nwg)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned my = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1u) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_barrier2(unsigned* ctr /* 17 words, 64 B apart */, unsigned* gen, unsigned nwg)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned my = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned grp = blockIdx.x & 15u, in_grp = (nwg >> 4) + ((nwg & 15u) > grp ? 1u : 0u);
        bool last = false;
        if (__hip_atomic_fetch_add(ctr + 16 * (1 + grp), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1u) {
            __hip_atomic_store(ctr + 16 * (1 + grp), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned ngrp = nwg < 16u ? nwg : 16u;
            last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1u;
        }
        if (last) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

// variant 2: relaxed atomics and polls, ONE release fence before arriving and one acquire fence after leaving
__device__ __forceinline__ void grid_barrier_relaxed(unsigned* ctr, unsigned* gen, unsigned nwg, int fences, int sleep)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned my = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1u) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my) { if (sleep) __builtin_amdgcn_s_sleep(8); }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
// the "last workgroup" pattern: everybody releases and draws a ticket, nobody waits
__global__ __launch_bounds__(256) void k_last(unsigned* ctr, int fences, unsigned* out)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            out[0] = 1u;
        }
    }
}
__global__ __launch_bounds__(256) void k_barriers_relaxed(unsigned* ctr, unsigned* gen, int nb, int fences, int sleep, unsigned* out)
{
    for (int i = 0; i < nb; ++i) grid_barrier_relaxed(ctr, gen, gridDim.x, fences, sleep);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1u;
}
__global__ __launch_bounds__(256) void k_empty(unsigned* out) { if (out == nullptr) __builtin_trap(); }
__global__ __launch_bounds__(256) void k_barriers(unsigned* ctr, unsigned* gen, int nb, int two_level, unsigned* out)
{
    for (int i = 0; i < nb; ++i) {
        if (two_level) grid_barrier2(ctr, gen, gridDim.x); else grid_barrier(ctr, gen, gridDim.x);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1u;
}

template <class F> static double per_call_us(F&& f, int reps = 200)
{
    for (int i = 0; i < 20; ++i) { f(); hipDeviceSynchronize(); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { f(); hipDeviceSynchronize(); }
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main()
{
    unsigned *ctr, *gen, *out;
    hipMalloc(&ctr, 4096 * 4); hipMalloc(&gen, 256); hipMalloc(&out, 256);
    hipMemset(ctr, 0, 4096 * 4); hipMemset(gen, 0, 256);
    hipStream_t s; hipStreamCreate(&s);
    int coop = 0; hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0);
    printf("cooperative launch supported: %d\n", coop);
    for (unsigned nwg : {54u, 512u, 1024u}) {
        const double e = per_call_us([&] { k_empty<<<nwg, 256, 0, s>>>(out); });
        void* eargs[] = {&out};
        const double ec = per_call_us([&] { hipLaunchCooperativeKernel((const void*)k_empty, dim3(nwg), dim3(256), eargs, 0, s); });
        const double e2 = per_call_us([&] { k_empty<<<nwg, 256, 0, s>>>(out); k_empty<<<nwg, 256, 0, s>>>(out); });
        const double e4 = per_call_us([&] { for (int i = 0; i < 4; ++i) k_empty<<<nwg, 256, 0, s>>>(out); });
        printf("%4u workgroups: empty kernel %.1f us per synchronous call, cooperative launch %.1f, two kernels %.1f, four kernels %.1f\n", nwg, e, ec, e2, e4);
        for (int two : {0, 1}) {
            double t[4];
            int nbs[4] = {0, 1, 2, 8};
            for (int j = 0; j < 4; ++j) {
                int nb = nbs[j];
                t[j] = per_call_us([&] { k_barriers<<<nwg, 256, 0, s>>>(ctr, gen, nb, two, out); });
            }
            printf("      %s barrier: 0 -> %.1f us, 1 -> %.1f, 2 -> %.1f, 8 -> %.1f   (%.2f us per barrier)\n", two ? "two-level" : "one-counter", t[0], t[1], t[2], t[3],
                   (t[3] - t[0]) / 8);
        }
        for (int fences : {0, 1}) for (int sleep : {0, 1}) {
            double t[3];
            int nbs[3] = {0, 1, 8};
            for (int j = 0; j < 3; ++j) {
                int nb = nbs[j];
                t[j] = per_call_us([&] { k_barriers_relaxed<<<nwg, 256, 0, s>>>(ctr, gen, nb, fences, sleep, out); });
            }
            printf("      relaxed barrier, fences %d, sleep %d: 0 -> %.1f us, 1 -> %.1f, 8 -> %.1f   (%.2f us per barrier)\n", fences, sleep, t[0], t[1], t[2], (t[2] - t[0]) / 8);
        }
        for (int fences : {0, 1})
            printf("      last-workgroup ticket, fences %d: %.1f us per synchronous call\n", fences, per_call_us([&] { k_last<<<nwg, 256, 0, s>>>(ctr, fences, out); }));
    }
    return 0;
}
