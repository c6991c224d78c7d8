// host_capi.cpp -- libmkamd_host.so: the library's HOST implementation of calculate_occupancy (cpu_occupancy.h)
// as a shared object of its OWN, built with the plain C++ compiler and linked against nothing but the C++ runtime -- a host
// without ROCm (no hipcc to build libmkamd.so, no libamdhip64 to load it) can still run method="CPU" /
// occupancy_utils.calculate_occupancy_cpu (inside libmkamd.so the entry point needed the HIP runtime it exists to
// do without).  Same two entry points, same contract and bits as the copies libmkamd.so exports (include/mkamd_voxel.h); built
// with -ffp-contract=off: the reference's arithmetic, one rounding per operation.
#include "cpu_occupancy.h"

#include <cstdio>
#include <new>

static thread_local char g_host_error[256] = "";

static int host_fail(int code, const char* msg)
{
    snprintf(g_host_error, sizeof g_host_error, "%s", msg);
    return code;
}

extern "C" {

const char* mkamd_host_last_error(void) { return g_host_error; }

// (status codes of include/mkamd_voxel.h: 0 ok, 1 bad argument, 6 out of host memory, 2 anything else)
int mkamd_calculate_occupancy_cpu_threads(const double* centers, int64_t V, const float* coords, int64_t N, const double* sigmas,
                                          int32_t C, double* results, int32_t n_threads)
{
    try {
        if (V < 0 || N < 0 || C <= 0) return host_fail(1, "n_centers/n_atoms must be >= 0 and n_channels > 0");
        if (V == 0 || N == 0) return 0;
        if (!centers || !coords || !sigmas || !results) return host_fail(1, "centers/coords/sigmas/results pointer is NULL");
        if (N > 0xFFFFFFF0LL) return host_fail(1, "more than 2^32 atoms");
        mkamd::cpu::calculate_occupancy(centers, V, coords, N, sigmas, C, results, (int)n_threads);
        return 0;
    } catch (const std::bad_alloc&) {
        return host_fail(6, "out of host memory");
    } catch (...) {
        return host_fail(2, "unexpected C++ exception");
    }
}

int mkamd_calculate_occupancy_cpu(const double* centers, int64_t V, const float* coords, int64_t N, const double* sigmas, int32_t C,
                                  double* results)
{
    return mkamd_calculate_occupancy_cpu_threads(centers, V, coords, N, sigmas, C, results, 0);
}

}  // extern "C"
