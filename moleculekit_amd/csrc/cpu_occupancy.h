// cpu_occupancy.h -- the library's HOST implementation of calculate_occupancy ("a CPU function with
// a1's exact contract"), for the GPU-less hosts on which moleculekit users prepare data (built twice: into libmkamd.so, and by
// the plain C++ compiler into libmkamd_host.so -- host_capi.cpp -- which needs no ROCm at all).  Product code: it shares nothing
// with the test suite's checker (which restates the reference's N x V loop) and is never taken silently -- only
// mkamd_calculate_occupancy_cpu / method="CPU" reach it; every GPU entry point still fails loudly without a device.
//
// Contract (moleculekit/occupancy_utils/occupancy_utils.pyx:34-61):
//     results[v,c] = max(results[v,c], 1 - exp(-(sigmas[a,c] / |coords[a] - centers[v]|)^12))   over atoms a with |.|^2 < 25, sigma != 0
// in double, float32 coordinates promoted, strict `<`, x^12 as x3*x3*x3*x3 with x3 = x*x*x, `value > old ? value : old`
// (a NaN is never stored), max-accumulated IN PLACE.  The reference walks all N x V pairs; here the ATOMS are binned into a
// uniform cell list (cell edge >= the 5 A cutoff) and every centre looks at the 27 cells around it -- the same arithmetic per
// pair that is in range, and a maximum does not depend on the order its candidates arrive in, so the result is the
// reference's bit for bit (tests/test_cpu_entry.py compares with the goldens of the real reference: array_equal).
// Centres are independent: host threads take contiguous slices of them (no write is shared).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace mkamd {
namespace cpu {

constexpr double CPU_CUTOFF = 5.0, CPU_CUTOFF2 = 25.0;     // occupancy_utils.pyx:53

struct AtomCells {
    double lo[3] = {0, 0, 0}, inv_h[3] = {0, 0, 0};
    int n[3] = {0, 0, 0};
    std::vector<uint32_t> start;          // [ncell + 1]
    std::vector<double> pos;              // [M, 3] cell-sorted (already promoted to double, as the reference's subtraction does)
    std::vector<uint32_t> atom;           // [M] original index (the sigma row)
};

// bin the atoms that can matter: finite coordinates and at least one channel with sigma != 0
inline bool build_cells(const float* coords, int64_t N, const double* sigmas, int32_t C, AtomCells& cl)
{
    std::vector<uint32_t> use;
    use.reserve((size_t)N);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t a = 0; a < N; ++a) {
        const float* p = coords + 3 * a;
        if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;      // d^2 is NaN / inf: never < 25
        bool any = false;
        for (int32_t c = 0; c < C && !any; ++c) any = sigmas[(size_t)a * C + c] != 0.0;           // (NaN != 0: kept; its value is NaN, never stored)
        if (!any) continue;
        use.push_back((uint32_t)a);
        for (int ax = 0; ax < 3; ++ax) { lo[ax] = std::min(lo[ax], (double)p[ax]); hi[ax] = std::max(hi[ax], (double)p[ax]); }
    }
    if (use.empty()) return false;
    size_t ncell = 1;
    for (int ax = 0; ax < 3; ++ax) {
        // cell edge: the cutoff, or more when the atoms span so much that the grid would not stay small (outliers): <= 256 cells per axis
        const double extent = hi[ax] - lo[ax];
        const double h = std::max(CPU_CUTOFF, extent / 256.0) * (1.0 + 1e-12);
        cl.lo[ax] = lo[ax];
        cl.inv_h[ax] = 1.0 / h;
        cl.n[ax] = (int)std::floor(extent * cl.inv_h[ax]) + 1;
        ncell *= (size_t)cl.n[ax];
    }
    auto cell_of = [&](const float* p) {
        size_t id = 0;
        for (int ax = 0; ax < 3; ++ax) {
            int i = (int)std::floor(((double)p[ax] - cl.lo[ax]) * cl.inv_h[ax]);
            i = i < 0 ? 0 : (i >= cl.n[ax] ? cl.n[ax] - 1 : i);
            id = id * (size_t)cl.n[ax] + (size_t)i;
        }
        return id;
    };
    cl.start.assign(ncell + 1, 0u);
    for (uint32_t a : use) ++cl.start[cell_of(coords + 3 * (size_t)a) + 1];
    for (size_t i = 0; i < ncell; ++i) cl.start[i + 1] += cl.start[i];
    std::vector<uint32_t> cur(cl.start.begin(), cl.start.end() - 1);
    cl.pos.resize(use.size() * 3);
    cl.atom.resize(use.size());
    for (uint32_t a : use) {
        const float* p = coords + 3 * (size_t)a;
        const uint32_t s = cur[cell_of(p)]++;
        cl.pos[3 * (size_t)s] = (double)p[0]; cl.pos[3 * (size_t)s + 1] = (double)p[1]; cl.pos[3 * (size_t)s + 2] = (double)p[2];
        cl.atom[s] = a;
    }
    return true;
}

// centres [v0, v1): every atom of the 27 cells around a centre, the reference's arithmetic per pair -- one rounding per
// operation, never contracted into an fma: the pragma below covers THIS function's body only (clang: libmkamd.so, where this
// header sits in the middle of capi.hip); the stand-alone host library (host_capi.cpp, g++) is built with -ffp-contract=off
inline void centres_slice(const AtomCells& cl, const double* centers, int64_t v0, int64_t v1, const double* sigmas, int32_t C, double* results)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    for (int64_t v = v0; v < v1; ++v) {
        const double cx = centers[3 * v], cy = centers[3 * v + 1], cz = centers[3 * v + 2];
        const double c3[3] = {cx, cy, cz};
        int lo[3], hi[3];
        bool none = false;
        for (int ax = 0; ax < 3; ++ax) {
            // cells an atom within the cutoff of this centre can sit in (a NaN centre fails both comparisons: no cells)
            // (the reach a hair wider than the cutoff: c - 5 is rounded, and an atom exactly that close must not fall off the range)
            const double reach = CPU_CUTOFF + 1e-6 + 1e-12 * std::fabs(c3[ax]);
            const double a = std::floor((c3[ax] - reach - cl.lo[ax]) * cl.inv_h[ax]), b = std::floor((c3[ax] + reach - cl.lo[ax]) * cl.inv_h[ax]);
            if (!(b >= 0.0) || !(a <= (double)(cl.n[ax] - 1))) { none = true; break; }
            lo[ax] = a < 0.0 ? 0 : (int)a;
            hi[ax] = b > (double)(cl.n[ax] - 1) ? cl.n[ax] - 1 : (int)b;
        }
        if (none) continue;
        double* res = results + (size_t)v * C;
        for (int ix = lo[0]; ix <= hi[0]; ++ix)
            for (int iy = lo[1]; iy <= hi[1]; ++iy) {
                const size_t row = ((size_t)ix * cl.n[1] + (size_t)iy) * (size_t)cl.n[2];
                const uint32_t s0 = cl.start[row + (size_t)lo[2]], s1 = cl.start[row + (size_t)hi[2] + 1];     // the z-cells of a row are adjacent
                for (uint32_t s = s0; s < s1; ++s) {
                    const double dx = cl.pos[3 * (size_t)s] - cx, dy = cl.pos[3 * (size_t)s + 1] - cy, dz = cl.pos[3 * (size_t)s + 2] - cz;
                    const double dist2 = dx * dx + dy * dy + dz * dz;
                    if (!(dist2 < CPU_CUTOFF2)) continue;
                    const double root = std::sqrt(dist2);
                    const double* sg = sigmas + (size_t)cl.atom[s] * C;
                    for (int32_t h = 0; h < C; ++h) {
                        if (sg[h] == 0.0) continue;
                        const double x = sg[h] / root;
                        const double x3 = x * x * x;
                        const double x12 = x3 * x3 * x3 * x3;
                        const double value = 1.0 - std::exp(-x12);
                        if (value > res[h]) res[h] = value;
                    }
                }
            }
    }
}

inline int default_threads(int64_t V, int64_t N)
{
    if (const char* e = std::getenv("MKAMD_CPU_THREADS")) { const int n = std::atoi(e); if (n > 0) return n; }
    if ((double)V * (double)std::max<int64_t>(N, 1) < 4e6) return 1;                   // a pocket: thread start-up would dominate
    return (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
}

inline void calculate_occupancy(const double* centers, int64_t V, const float* coords, int64_t N, const double* sigmas, int32_t C,
                                double* results, int n_threads)
{
    AtomCells cl;
    if (V <= 0 || N <= 0 || !build_cells(coords, N, sigmas, C, cl)) return;
    int nt = n_threads > 0 ? n_threads : default_threads(V, N);
    nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, V / 512));
    if (nt <= 1) { centres_slice(cl, centers, 0, V, sigmas, C, results); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const int64_t v0 = V * t / nt, v1 = V * (t + 1) / nt;
        pool.emplace_back([&, v0, v1] { centres_slice(cl, centers, v0, v1, sigmas, C, results); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace cpu
}  // namespace mkamd
