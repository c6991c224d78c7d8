// tools/census.hip -- how many workgroups of a given size / LDS footprint are resident per CU on gfx950?
// Each block bumps a per-CU counter on entry, records the maximum it ever sees, spins, and leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int LDS_BYTES>
__global__ void k_census(unsigned* resident, unsigned* maxres, long long spin_cycles)
{
    __shared__ unsigned pad[LDS_BYTES / 4 + 1];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    const unsigned idx = ((xcc & 0xf) * 8 + se) * 32 + sh * 16 + cu;     // generous index space (4096)
    pad[threadIdx.x % (LDS_BYTES / 4 + 1)] = idx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned now = atomicAdd(&resident[idx], 1u) + 1u;
        atomicMax(&maxres[idx], now);
    }
    const long long t0 = clock64();
    while (clock64() - t0 < spin_cycles) __builtin_amdgcn_s_sleep(10);
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(&resident[idx], 1u + (pad[0] == 0xffffffffu));
}

template <int LDS>
void run(int threads, const char* name)
{
    unsigned *res, *mx;
    (void)hipMalloc(&res, 4096 * 4); (void)hipMalloc(&mx, 4096 * 4);
    (void)hipMemset(res, 0, 4096 * 4); (void)hipMemset(mx, 0, 4096 * 4);
    k_census<LDS><<<256 * 48, threads>>>(res, mx, 2000000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(4096);
    (void)hipMemcpy(h.data(), mx, 4096 * 4, hipMemcpyDeviceToHost);
    unsigned best = 0, ncu = 0; unsigned long long sum = 0;
    for (unsigned v : h) if (v) { best = std::max(best, v); ++ncu; sum += v; }
    printf("%-28s threads=%3d lds=%6d B : CUs seen=%u  max resident WGs/CU=%u  mean=%.2f  (waves/CU max=%u)\n",
           name, threads, LDS, ncu, best, ncu ? (double)sum / ncu : 0.0, best * (threads / 64));
    (void)hipFree(res); (void)hipFree(mx);
}

int main()
{
    run<16>(64, "64-thread WG, no LDS");
    run<8192>(64, "64-thread WG, 8 KiB LDS");
    run<12800>(64, "64-thread WG, 12.5 KiB LDS");
    run<16384>(64, "64-thread WG, 16 KiB LDS");
    run<16>(128, "128-thread WG, no LDS");
    run<25600>(128, "128-thread WG, 25 KiB LDS");
    run<16>(256, "256-thread WG, no LDS");
    run<51200>(256, "256-thread WG, 50 KiB LDS");
    return 0;
}
