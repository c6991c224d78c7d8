/*
 * oracle/distance_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float32 arithmetic in the reference's operation order, single thread)
 * of moleculekit/distance_utils/distance_utils.pyx for the SURVEY.md section 8f-1 row.
 * Compiled with -ffp-contract=off so every multiply / add rounds separately, as in the reference
 * build (g++ -O3 for baseline x86-64: no FMA).  Pinned bit-exactly against the compiled reference
 * (tests/golden/distance_*.npz, tests/test_oracle_distance.py).
 *
 * coords layout is the reference's: float32 [n_atoms, 3, n_frames], frame fastest
 * (Molecule.coords, molecule.py:205-232); box float32 [3, n_frames].
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define C3(coords, F, a, k, f) ((coords)[((size_t)(a) * 3 + (k)) * (size_t)(F) + (size_t)(f)])

/* distance_utils.pyx:34-54 (_dist) and :188-206 (_dist2): float32 throughout; the reference is C++,
 * where round(float) is the float overload. */
static float dist2_wrap(float x1, float y1, float z1, float x2, float y2, float z2,
                        float bx, float by, float bz, int wrap)
{
    float dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
    if (wrap) {
        dx = dx - bx * roundf(dx / bx);
        dy = dy - by * roundf(dy / by);
        dz = dz - bz * roundf(dz / bz);
    }
    return dx * dx + dy * dy + dz * dz;
}

/* distance_utils.pyx:126-155 (dist_trajectory); `squared` != 0 returns dist2 (what
 * contacts_trajectory :59-93 thresholds). results float32 [F, npairs]. */
void oracle_dist_trajectory(const float *coords, int64_t n_atoms, int64_t F, const float *box,
                            const uint32_t *sel1, int64_t n1, const uint32_t *sel2, int64_t n2,
                            const uint32_t *chains, int selfdist, int pbc, int squared, float *results,
                            int64_t npairs)
{
    (void)n_atoms;
    for (int64_t f = 0; f < F; ++f) {
        int64_t idx = 0;
        for (int64_t i = 0; i < n1; ++i) {
            const uint32_t a = sel1[i];
            for (int64_t j = selfdist ? i + 1 : 0; j < n2; ++j) {
                const uint32_t b = sel2[j];
                const float d2 = dist2_wrap(C3(coords, F, a, 0, f), C3(coords, F, a, 1, f), C3(coords, F, a, 2, f),
                                            C3(coords, F, b, 0, f), C3(coords, F, b, 1, f), C3(coords, F, b, 2, f),
                                            box[0 * F + f], box[1 * F + f], box[2 * F + f],
                                            pbc && chains[a] != chains[b]);
                results[f * npairs + idx] = squared ? d2 : (float)sqrt((double)d2);
                ++idx;
            }
        }
    }
}

/* distance_utils.pyx:160-183 (_calc_com): sequential float32 accumulation in group order. */
static void calc_com(const float *coords, int64_t F, int64_t f, const int32_t *group, int64_t n,
                     const float *masses, float *com)
{
    float total = 0, cx = 0, cy = 0, cz = 0;
    for (int64_t k = 0; k < n; ++k) {
        const int32_t a = group[k];
        cx += C3(coords, F, a, 0, f) * masses[a];
        cy += C3(coords, F, a, 1, f) * masses[a];
        cz += C3(coords, F, a, 2, f) * masses[a];
        total += masses[a];
    }
    com[0] = cx / total; com[1] = cy / total; com[2] = cz / total;
}

/* distance_utils.pyx:211-281 (dist_trajectory_reduction) and :286-350 (..._pairs, pairs != 0).
 * Groups are CSR lists: atoms int32 [sum], offsets int64 [ng+1]. reduction: 0 closest, 1 COM. */
void oracle_dist_trajectory_reduction(const float *coords, int64_t F, const float *box,
                                      const int32_t *g1_atoms, const int64_t *g1_off, int64_t ng1,
                                      const int32_t *g2_atoms, const int64_t *g2_off, int64_t ng2,
                                      const uint32_t *chains1, const uint32_t *chains2, int selfdist,
                                      int pairs, int pbc, const float *masses, int reduction1,
                                      int reduction2, float *results, int64_t nout)
{
    for (int64_t f = 0; f < F; ++f) {
        int64_t idx = 0;
        const float bx = box[0 * F + f], by = box[1 * F + f], bz = box[2 * F + f];
        for (int64_t a = 0; a < ng1; ++a) {
            const int64_t b0 = pairs ? a : (selfdist ? a + 1 : 0);
            const int64_t b1 = pairs ? a + 1 : ng2;
            float com1[3] = {0, 0, 0};
            if (reduction1 == 1) calc_com(coords, F, f, g1_atoms + g1_off[a], g1_off[a + 1] - g1_off[a], masses, com1);
            for (int64_t b = b0; b < b1; ++b) {
                float com2[3] = {0, 0, 0};
                if (reduction2 == 1) calc_com(coords, F, f, g2_atoms + g2_off[b], g2_off[b + 1] - g2_off[b], masses, com2);
                float mindist = -1;
                const int diff_chain = chains1[a] != chains2[b];
                const int64_t n1 = reduction1 == 1 ? 1 : g1_off[a + 1] - g1_off[a];
                const int64_t n2 = reduction2 == 1 ? 1 : g2_off[b + 1] - g2_off[b];
                for (int64_t i = 0; i < n1; ++i) {
                    float x1, y1, z1;
                    if (reduction1 == 1) { x1 = com1[0]; y1 = com1[1]; z1 = com1[2]; }
                    else { const int32_t at = g1_atoms[g1_off[a] + i]; x1 = C3(coords, F, at, 0, f); y1 = C3(coords, F, at, 1, f); z1 = C3(coords, F, at, 2, f); }
                    for (int64_t j = 0; j < n2; ++j) {
                        float x2, y2, z2;
                        if (reduction2 == 1) { x2 = com2[0]; y2 = com2[1]; z2 = com2[2]; }
                        else { const int32_t at = g2_atoms[g2_off[b] + j]; x2 = C3(coords, F, at, 0, f); y2 = C3(coords, F, at, 1, f); z2 = C3(coords, F, at, 2, f); }
                        const float d2 = dist2_wrap(x1, y1, z1, x2, y2, z2, bx, by, bz, pbc && diff_chain);
                        if (d2 < mindist || mindist < 0) mindist = d2;
                    }
                }
                results[f * nout + idx] = (float)sqrt((double)mindist);
                ++idx;
            }
        }
    }
}

/* distance_utils.pyx:355-383 (cdist): results float32 [n1, n2]; any dimension D. */
void oracle_cdist(const float *c1, int64_t n1, const float *c2, int64_t n2, int64_t D, float *results)
{
    for (int64_t i = 0; i < n1; ++i)
        for (int64_t j = 0; j < n2; ++j) {
            float d2 = 0;
            for (int64_t k = 0; k < D; ++k) {
                const float diff = c1[i * D + k] - c2[j * D + k];
                d2 = d2 + diff * diff;
            }
            results[i * n2 + j] = (float)sqrt((double)d2);
        }
}

/* distance_utils.pyx:388-416 (pdist): condensed upper triangle, float32 [n(n-1)/2]. */
void oracle_pdist(const float *c, int64_t n, int64_t D, float *results)
{
    int64_t ii = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) {
            float d2 = 0;
            for (int64_t k = 0; k < D; ++k) {
                const float diff = c[i * D + k] - c[j * D + k];
                d2 = d2 + diff * diff;
            }
            results[ii++] = (float)sqrt((double)d2);
        }
}
