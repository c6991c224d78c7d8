/*
 * oracle/occupancy_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, double precision, single thread) of the reference
 * algorithm for the voxel-descriptor hot path. Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it, and only as the checker / the timed
 * CPU baseline ("kind": "port"). The product path (moleculekit_amd/) never links,
 * imports or falls back to anything in oracle/.
 *
 * Parity is PINNED: oracle_calculate_occupancy is checked against outputs of the real
 * reference (built in a scratch dir from /root/reference and imported in the build
 * container, see tests/golden/make_golden.py) and against the reference's own test
 * fixtures (celecoxib/ledipasvir channel 7) -- tests/test_oracle.py.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/*
 * Follows moleculekit/occupancy_utils/occupancy_utils.pyx:34-61 (calculate_occupancy).
 *   centers f64 [V,3], coords f32 [N,3], sigmas f64 [N,C], results f64 [V,C] (in-place max).
 * Loop order (atoms outer, centres inner), all-double arithmetic with float32 coords
 * promoted, strict `dist2 < 25`, `sigma == 0` skip, x12 = x3*x3*x3*x3, and the
 * `value > old ? value : old` max (so NaN is never stored) are kept as in the reference.
 */
void oracle_calculate_occupancy(const double *centers, int64_t n_centers,
                                const float *coords, int64_t n_atoms,
                                const double *sigmas, int32_t n_channels,
                                double *results)
{
    for (int64_t a = 0; a < n_atoms; ++a) {
        const double ax = (double)coords[3 * a + 0];
        const double ay = (double)coords[3 * a + 1];
        const double az = (double)coords[3 * a + 2];
        const double *sig = sigmas + (size_t)a * n_channels;
        for (int64_t c = 0; c < n_centers; ++c) {
            const double dx = ax - centers[3 * c + 0];
            const double dy = ay - centers[3 * c + 1];
            const double dz = az - centers[3 * c + 2];
            const double dist2 = dx * dx + dy * dy + dz * dz;
            if (dist2 < 25.0) {
                double *res = results + (size_t)c * n_channels;
                for (int32_t h = 0; h < n_channels; ++h) {
                    if (sig[h] == 0.0)
                        continue;
                    const double x = sig[h] / sqrt(dist2);
                    const double x3 = x * x * x;
                    const double x12 = x3 * x3 * x3 * x3;
                    const double value = 1.0 - exp(-x12);
                    if (value > res[h])
                        res[h] = value;
                }
            }
        }
    }
}

/*
 * Periodic (orthorhombic minimum-image) voxelization: calculate_occupancy with the
 * displacement coord-centre wrapped per axis before the cutoff test, i.e. the composition
 * of occupancy_utils.pyx:46-61 with the wrap of distance_utils.pyx:49-52
 * (`d = d - box * round(d / box)`, C round(): half away from zero), evaluated in double
 * (SURVEY.md section 8a "Periodic voxelization"). box f64 [3]; every edge must be > 10.
 * This is an EXTENSION of the reference (getVoxelDescriptors has no periodic handling);
 * it is pinned against 27 shifted calls of the reference kernel (tests/golden/make_golden.py).
 */
void oracle_calculate_occupancy_pbc(const double *centers, int64_t n_centers,
                                    const float *coords, int64_t n_atoms,
                                    const double *sigmas, int32_t n_channels,
                                    const double *box, double *results)
{
    for (int64_t a = 0; a < n_atoms; ++a) {
        const double ax = (double)coords[3 * a + 0];
        const double ay = (double)coords[3 * a + 1];
        const double az = (double)coords[3 * a + 2];
        const double *sig = sigmas + (size_t)a * n_channels;
        for (int64_t c = 0; c < n_centers; ++c) {
            double dx = ax - centers[3 * c + 0];
            double dy = ay - centers[3 * c + 1];
            double dz = az - centers[3 * c + 2];
            dx = dx - box[0] * round(dx / box[0]);
            dy = dy - box[1] * round(dy / box[1]);
            dz = dz - box[2] * round(dz / box[2]);
            const double dist2 = dx * dx + dy * dy + dz * dz;
            if (dist2 < 25.0) {
                double *res = results + (size_t)c * n_channels;
                for (int32_t h = 0; h < n_channels; ++h) {
                    if (sig[h] == 0.0)
                        continue;
                    const double x = sig[h] / sqrt(dist2);
                    const double x3 = x * x * x;
                    const double x12 = x3 * x3 * x3 * x3;
                    const double value = 1.0 - exp(-x12);
                    if (value > res[h])
                        res[h] = value;
                }
            }
        }
    }
}

/*
 * Lattice centres, follows moleculekit/tools/voxeldescriptors.py:125-132 (_getGridCenters)
 * + :245-247 (getCenters: `lattice + bb_min`, reshape (V,3)): centre = fl64(index*res) + bb_min,
 * x slowest / z fastest. bb_min is passed already promoted to double (it is float32 in the
 * bbox branch and float64 in the boxsize branch of the reference).
 */
void oracle_grid_centers(const double *bb_min, const int64_t *nvox, double resolution,
                         double *centers)
{
    size_t o = 0;
    for (int64_t ix = 0; ix < nvox[0]; ++ix)
        for (int64_t iy = 0; iy < nvox[1]; ++iy)
            for (int64_t iz = 0; iz < nvox[2]; ++iz) {
                centers[o++] = (double)ix * resolution + bb_min[0];
                centers[o++] = (double)iy * resolution + bb_min[1];
                centers[o++] = (double)iz * resolution + bb_min[2];
            }
}

/*
 * Minimum-image squared distance, follows moleculekit/distance_utils/distance_utils.pyx:188-206
 * (_dist2; same arithmetic as _dist :34-54). The reference is compiled as C++, where the
 * unqualified round(float) resolves to the float overload, so the whole wrap
 * `d = d - box * round(d / box)` is float32 arithmetic (round: half away from zero).
 * Pinned against the compiled reference's dist_trajectory in tests/golden (min_image vectors).
 */
float oracle_min_image_dist2(const float *c1, const float *c2, const float *box,
                             int diff_chain, int pbc)
{
    float dx = c1[0] - c2[0];
    float dy = c1[1] - c2[1];
    float dz = c1[2] - c2[2];
    if (pbc && diff_chain) {
        dx = dx - box[0] * roundf(dx / box[0]);
        dy = dy - box[1] * roundf(dy / box[1]);
        dz = dz - box[2] * roundf(dz / box[2]);
    }
    return dx * dx + dy * dy + dz * dz;
}

/*
 * The same loop split over host threads (contiguous slices of the centres: no two threads touch the same result row).
 * Only bench.py's cpu_baseline leg uses it, for the "what would all host cores do" figure next to the serial number --
 * the reference itself is serial (no nogil, OpenMP commented out: setup.py:48).
 */
#include <pthread.h>

typedef struct {
    const double *centers; int64_t c0, c1;
    const float *coords; int64_t n_atoms;
    const double *sigmas; int32_t n_channels;
    double *results;
} oracle_slice_t;

static void *oracle_slice_run(void *arg)
{
    const oracle_slice_t *s = (const oracle_slice_t *)arg;
    oracle_calculate_occupancy(s->centers + 3 * s->c0, s->c1 - s->c0, s->coords, s->n_atoms, s->sigmas, s->n_channels,
                               s->results + (size_t)s->c0 * s->n_channels);
    return NULL;
}

int oracle_calculate_occupancy_threads(const double *centers, int64_t n_centers, const float *coords, int64_t n_atoms,
                                       const double *sigmas, int32_t n_channels, double *results, int32_t n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if ((int64_t)n_threads > n_centers) n_threads = (int32_t)(n_centers > 0 ? n_centers : 1);
    pthread_t th[1024];
    oracle_slice_t sl[1024];
    int started = 0;
    for (int t = 0; t < n_threads; ++t) {
        sl[t].centers = centers; sl[t].coords = coords; sl[t].n_atoms = n_atoms; sl[t].sigmas = sigmas;
        sl[t].n_channels = n_channels; sl[t].results = results;
        sl[t].c0 = n_centers * t / n_threads; sl[t].c1 = n_centers * (t + 1) / n_threads;
        if (pthread_create(&th[t], NULL, oracle_slice_run, &sl[t]) != 0) break;
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    for (int t = started; t < n_threads; ++t) oracle_slice_run(&sl[t]);     /* could not spawn: run inline */
    return started;
}
