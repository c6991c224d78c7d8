/*
 * mkamd_distance.h -- C ABI of libmkamd.so, distance_utils row.
 *
 * GPU replacements for moleculekit/distance_utils/distance_utils.pyx:
 *   dist_trajectory                  :126-155     -> mkamd_dist_trajectory_host / _dev
 *   contacts_trajectory              :59-93  }    -> mkamd_contacts_trajectory_host / _dev (thresholded and compacted on the GPU,
 *   get_collisions                   :98-121 }       the reference's (frame, i, j) order; no [frames x pairs] matrix anywhere)
 *   dist_trajectory_reduction        :211-281 }   -> mkamd_dist_reduction_host / _dev (pairs = 0 / 1)
 *   dist_trajectory_reduction_pairs  :286-350 }
 *   cdist                            :355-383     -> mkamd_cdist_host / _dev
 *   pdist                            :388-416     -> mkamd_pdist_host / _dev
 * All float32, BIT-EXACT with the reference (same operation order, one rounding per operation).
 *
 * Layouts (C-contiguous, the reference's):  coords float32 [n_atoms, 3, n_frames] (Molecule.coords),
 * box float32 [3, n_frames] (Molecule.box), results float32 [n_frames, n_pairs].
 * Pair order: i over sel1, j over sel2 (from i+1 when selfdist) -- the reference's loop order.
 * n_frames < 2^30 (MKAMD_EINVAL beyond: a frame's byte offset into a coordinate row is a 32-bit buffer offset in the
 * kernels; the reference's own frame loops are C ints).
 * Host pointers in, host pointers out (copies + kernels + synchronise); status codes as mkamd_voxel.h.
 */
#ifndef MKAMD_DISTANCE_H
#define MKAMD_DISTANCE_H

#include "mkamd_voxel.h"

#ifdef __cplusplus
extern "C" {
#endif

/* number of (i,j) pairs the reference's loops visit (n1*n2, or sum_i max(n2-1-i,0) when selfdist) */
int64_t mkamd_dist_count_pairs(int64_t n1, int64_t n2, int selfdist);

/* dist_trajectory(coords, box, sel1, sel2, digitized_chains, selfdist, pbc, results);  squared != 0
 * stores the squared distance (what contacts_trajectory compares with threshold^2) instead of its sqrt. */
int mkamd_dist_trajectory_host(mkamd_ctx* ctx, const float* coords, int64_t n_atoms, int64_t n_frames,
                               const float* box, const uint32_t* sel1, int64_t n1, const uint32_t* sel2,
                               int64_t n2, const uint32_t* digitized_chains, int selfdist, int pbc,
                               int squared, float* results);
/* same on device pointers (asynchronous on the context's stream) */
int mkamd_dist_trajectory_dev(mkamd_ctx* ctx, const float* d_coords, int64_t n_frames, const float* d_box,
                              const uint32_t* d_sel1, int64_t n1, const uint32_t* d_sel2, int64_t n2,
                              const uint32_t* d_digitized_chains, int selfdist, int pbc, int squared,
                              float* d_results);

/* contacts_trajectory(coords, box, sel1, sel2, digitized_chains, selfdist, pbc, dist_threshold): per frame the atom
 * pairs (a, b) = (sel1[i], sel2[j]) with dist2 <= threshold^2 (float32 compare, distance_utils.pyx:73,82), in the
 * reference's (i, j) loop order.  Counted, prefix-summed and written on the device, chunk of frames by chunk of frames
 * (memory stays bounded whatever n_frames x n_pairs is).
 *   frame_offsets int64 [n_frames + 1] (out): frame f owns pairs [frame_offsets[f], frame_offsets[f+1])
 *   *pairs (out): 2 * frame_offsets[n_frames] uint32 (a0, b0, a1, b1, ...) in memory OWNED BY THE CONTEXT, valid until
 *   the next contacts call on it (NULL when there is no contact).
 * get_collisions (:98-121) is the one-frame, non-periodic case on the concatenation of the two coordinate sets. */
int mkamd_contacts_trajectory_host(mkamd_ctx* ctx, const float* coords, int64_t n_atoms, int64_t n_frames,
                                   const float* box, const uint32_t* sel1, int64_t n1, const uint32_t* sel2,
                                   int64_t n2, const uint32_t* digitized_chains, int selfdist, int pbc,
                                   float dist_threshold, int64_t* frame_offsets, const uint32_t** pairs);

/* dist_trajectory_reduction / dist_trajectory_reduction_pairs.  The reference's vector<vector<int>> groups
 * are passed as CSR: atoms int32 [sum of group sizes], offsets int64 [n_groups + 1].  reduction: 0 closest,
 * 1 centre of mass (masses float32 [n_atoms]).  results float32 [n_frames, n_out], n_out = n_groups1 when
 * pairs else mkamd_dist_count_pairs(n_groups1, n_groups2, selfdist). */
int mkamd_dist_reduction_host(mkamd_ctx* ctx, const float* coords, int64_t n_atoms, int64_t n_frames,
                              const float* box, const int32_t* g1_atoms, const int64_t* g1_offsets,
                              int64_t n_groups1, const int32_t* g2_atoms, const int64_t* g2_offsets,
                              int64_t n_groups2, const uint32_t* digitized_chains1,
                              const uint32_t* digitized_chains2, int selfdist, int pairs, int pbc,
                              const float* masses, int reduction1, int reduction2, float* results);

/* ---- device-resident forms (round 6): device pointers in, device results out, on the context's stream (mkamd_ctx_set_stream)
 * -- for callers that keep the trajectory on the GPU (a decoded XTC chunk, an ML / analysis loop).  Same kernels, same bits as the
 * "_host" forms.  Asynchronous like mkamd_dist_trajectory_dev, except the contact list (its size has to reach the host). ---- */

/* contacts_trajectory (distance_utils.pyx:59-93) on device pointers: frame_offsets is a HOST array int64 [n_frames + 1] (out);
 * *d_pairs (out) points at 2 * frame_offsets[n_frames] uint32 (a0, b0, a1, b1, ...) in DEVICE memory owned by the context, valid
 * until the next contacts call on it (NULL when there is no contact).  The call returns when the list is complete. */
int mkamd_contacts_trajectory_dev(mkamd_ctx* ctx, const float* d_coords, int64_t n_frames, const float* d_box,
                                  const uint32_t* d_sel1, int64_t n1, const uint32_t* d_sel2, int64_t n2,
                                  const uint32_t* d_digitized_chains, int selfdist, int pbc, float dist_threshold,
                                  int64_t* frame_offsets, const uint32_t** d_pairs);
/* dist_trajectory_reduction[_pairs] (:211-350) on device pointers.  n_atoms (rows of d_coords) and n_g1_atoms (length of
 * d_g1_atoms = the value of d_g1_offsets[n_groups1]) are what the host knows about the device arrays: they choose the kernel
 * variant (32-bit row offsets, first-group atoms per wave), never the result.  Indices are NOT range-checked on the device. */
int mkamd_dist_reduction_dev(mkamd_ctx* ctx, const float* d_coords, int64_t n_atoms, int64_t n_frames, const float* d_box,
                             const int32_t* d_g1_atoms, const int64_t* d_g1_offsets, int64_t n_groups1, int64_t n_g1_atoms,
                             const int32_t* d_g2_atoms, const int64_t* d_g2_offsets, int64_t n_groups2,
                             const uint32_t* d_digitized_chains1, const uint32_t* d_digitized_chains2, int selfdist, int pairs,
                             int pbc, const float* d_masses, int reduction1, int reduction2, float* d_results);
/* cdist (:355-383) / pdist (:388-416) on device pointers */
int mkamd_cdist_dev(mkamd_ctx* ctx, const float* d_coords1, int64_t n1, const float* d_coords2, int64_t n2, int32_t dim,
                    float* d_results);
int mkamd_pdist_dev(mkamd_ctx* ctx, const float* d_coords, int64_t n, int32_t dim, float* d_results);

/* Self-test of the kernels' float32 square root.  The reference's sqrtf is correctly rounded; the kernels take roots with
 * one exact-residual correction of x * rsq(x) (8 issue slots; the provable form, v_sqrt_f32 + Tuckerman's test, takes 12 and
 * the kernels are bound by instruction issue).  That this is the correctly rounded root is a property of gfx950's v_rsq_f32,
 * checked rather than proved: this call compares the two forms on the device over EVERY float in [2^-96, inf) -- 1.9e9 values,
 * a few milliseconds -- and returns the number of mismatches (0 on the hardware this library is built for) and the bit pattern
 * of the first one.  Run by the GPU test tier. */
int mkamd_selftest_sqrt(mkamd_ctx* ctx, uint64_t* mismatches, uint32_t* first_bad_bits);

/* Which kernels dist_trajectory may take (default 0: all; the choice depends on the shape of the call): bits of `avoid_mask` --
 * 1 the block-per-frame kernel (rectangular calls with short rows -- the small calls MetricDistance makes: one launch), 2 the row
 * kernel (rectangular calls with rows of >= 64 second atoms), 4 the rectangular tile kernel, 8 the row kernel's 16-byte stores,
 * 16 (not an exclusion) the row kernel wherever it applies, also where the tile kernel is the measured better choice
 * (selfdist always takes the pair-table kernel); 32 (not a kernel): the host entry points upload the whole coordinate array instead
 * of the selected atoms' rows (csrc/host_pack.h); 64: selfdist calls keep the pair-table kernel where the triangular form of the row kernel
 * would be taken (selections of >= 700 atoms up to 32 frames, of >= 1 500 atoms at any frame count).  128: rectangular calls of few frames whose rows are too short for the row kernel and whose first selection is long keep the tile
 * kernel instead of the row kernel with the selections swapped (lanes along the first selection, transposed stores).
 * Calls of at most 32 frames take the row kernel wherever it applies (its lanes
 * run along the second atoms; the other kernels' along frames).  Every kernel produces the same
 * bits; for tests (every kernel over the same shapes) and same-box A-B timing. */
int mkamd_ctx_set_dist_kernels(mkamd_ctx* ctx, int avoid_mask);
/* Which kernel the "closest"/"closest" group reductions take (default 0: k_dist_reduction_closest with 4 or 8 first-group atoms in
 * registers, chosen from the mean size of the first groups -- and, for calls of at most 16 frames (8 when not periodic) in any reduction mode,
 * k_dist_reduction_few, whose lanes run along the second groups instead of the frames; 4 / 8: that many (+ 100: in blocks of four
 * waves instead of eight); -1: the generic kernel the centre-of-mass modes use; -2: the few-frame kernel at any number of frames).
 * Every choice produces the same bits; for tests and same-box A-B timing. */
int mkamd_ctx_set_reduction_block(mkamd_ctx* ctx, int block);
/* Names of the kernels the last dist_trajectory call on this context launched, as a profiler prints them (e.g.
 * "mkamd::k_sel_to_frames + mkamd::k_dist_rows<true, 4, true>"; empty before the first call): what bench.py reports as
 * the distance leg's `roofline.kernel` -- the choice depends on the shape of the call. */
int mkamd_ctx_last_dist_kernel(mkamd_ctx* ctx, char* name, size_t name_cap);

/* cdist(coords1 [n1,D], coords2 [n2,D]) -> results [n1,n2];  pdist(coords [n,D]) -> results [n(n-1)/2] */
int mkamd_cdist_host(mkamd_ctx* ctx, const float* coords1, int64_t n1, const float* coords2, int64_t n2,
                     int32_t dim, float* results);
int mkamd_pdist_host(mkamd_ctx* ctx, const float* coords, int64_t n, int32_t dim, float* results);

#ifdef __cplusplus
}
#endif
#endif /* MKAMD_DISTANCE_H */
