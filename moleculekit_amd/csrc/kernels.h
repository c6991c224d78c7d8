// kernels.h -- hand-written HIP kernels for the voxel-descriptor hot path on MI355X (gfx950).
//
// Replaces moleculekit/occupancy_utils/occupancy_utils.pyx:34-61 (calculate_occupancy), the
// lattice-centre generation of moleculekit/tools/voxeldescriptors.py:125-132,245-247 and, for
// periodic frames, fuses the orthorhombic minimum-image wrap of
// moleculekit/distance_utils/distance_utils.pyx:49-52 into the binning stage.
//
// Formulation:
//   reference :  res[v,c] = max_a { 1 - exp(-(sigma[a,c]/|a-v|)^12) : |a-v|^2 < 25, sigma != 0 }
//   here      :  q[v,c]   = min_a { |a-v|^2 * w[a,c]              : |a-v|^2 < 25 },  w = 1/sigma^2
//                res[v,c] = 1 - exp(-q^-6)            (monotone => max f == f(min q), exact)
//   so the inner loop has no transcendentals; rcp/exp run once per voxel-channel.
//   Entries (atom x channel) that share the same sigma form a CLASS; inside a class
//                min_a d2*w == w * min_a d2   and   "some in-range atom" == (min_a d2 < 25),
//   bit for bit (w > 0, float rounding is monotone), so the cutoff test and the multiply by w also
//   leave the inner loop; with d^2 expanded in x about the tile centre (plane_slope / plane_d2 below) what is left
//   per (voxel, entry) is  half a v_pk_fma and half a v_min3_f32.
//   It is a GATHER: one wave owns a K x 8 x 8 voxel tile, lane = (y,z), K x-planes in registers;
//   candidate atoms come from a uniform cell list, are culled against the tile box, counting-sorted
//   by (channel, class) in LDS and broadcast-read by all 64 lanes.  No atomics on the grid, no
//   zero-fill pass, one 32-byte store per voxel.  MFMA unused (a neighbourhood min-reduction).
//
// Kernels, in launch order (pipeline.h):
//   big items  : k_bin_count, k_prepass_reduce1/2, k_scan_finish, k_bin_fill              (cell lists; the counters
//                clear themselves, pipeline.h)
//   small items: k_prepass_items                                  (the same, one workgroup per item)
//   one molecule per call (<= 1 024 tile waves): k_bin_solo       (the whole pre-pass in one launch: records in the direct
//                layout, the class table kept across calls); big calls that are not pipelined: k_bin_direct (one pass,
//                the chain enqueued behind it as its device-side fall-back)
//   then       : k_voxelize_tiles[_lean|_team]<K,ECAP[,TEAM]>, k_voxelize_items<K>, k_tail<K,ECAP,SigT>   (the grid; dense tiles + fix-up)
//   explicit centres: k_sigma_to_w, k_occupancy_centers;   lattice centres: k_grid_centers.
//
// Coordinates: everything is in VOXEL units relative to the grid origin (voxel i's centre sits at
// integer coordinate i).  Atoms are decomposed IN DOUBLE into (cell index, cell-centre-relative
// float32 offset) so that float32 distances carry ~1e-7 voxel of error (parity 1e-5 needs < 2e-6).
#pragma once
#ifndef MK_DEVICE_API_PROVIDED   // tests/emu provides a host-side SIMT emulation of this API
#include "mk_device.h"
#endif

#include "mk_diagnostics.h"     // MK_DIAG / MK_PHASE_* / MK_BIN_*: all zero / empty in a release build (the only -D knobs of this file)

namespace mkamd {

constexpr int CHG = 8;               // channels per channel-group (one group = one pass of the tile kernel)
constexpr int NCLS = 15;             // distinct sigma values (classes) the sorted path handles per class table (4-bit ids)
constexpr int NSLOT = 16;            // bucket stride per channel (slot 15 is never used)
constexpr int NBUCKET = CHG * NSLOT; // (channel, class) buckets per tile
// LDS entry capacity of a tile (three float arrays, structure of arrays) comes in tiers: LDS per tile is
// what bounds the tile kernel's occupancy, so the leanest tier is the fastest as long as the tiles fit
// it; the host moves up when too many tiles of the previous calls did not (pipeline.h, choose_tier).
constexpr int NTIER = 3;
constexpr int ECAP_TIER[NTIER] = {640, 768, 1024};
constexpr int DENSE_WORDS = 1 + NTIER;   // [0] dense-list length, [1+t] tiles with more entries than tier t holds
constexpr int FEEDBACK_WORDS = NTIER + 4; // host-visible: [t] tiles over tier t, [NTIER] tiles, [NTIER+1] the error flag,
                                          // [NTIER+2] sequence number of the call whose tile kernel has finished (k_tail
                                          // writes it as it starts), [NTIER+3] of the last call whose k_tail changed values
constexpr int FB_TILES_DONE = NTIER + 2, FB_TAIL_WROTE = NTIER + 3;
constexpr int TRAV_BATCH = 4;       // candidate chunks whose loads are in flight together (6 and 8 measured the same)
constexpr int SURV_BATCH = 2;       // survivor chunks gathered together (histogram / placement passes)
constexpr int NXR = 3;               // x-reach sub-buckets: 0 = all K planes, 1 = low half only, 2 = high half only
constexpr int NBUCKET3 = NBUCKET * NXR;
constexpr unsigned CLS_EMPTY = 0xffffffffu;   // empty slot of the class table (never a valid w)
constexpr int CLS_OVERFLOW = NCLS;   // word NCLS of the table buffer: CLS_EMPTY, or 0 once > NCLS classes were seen
constexpr int CLS_TABLE_WORDS = NCLS + 1;

// Everything the kernels need to know about the batch of lattice grids (passed by value).
struct GridDesc {
    int nx, ny, nz;             // voxels per axis (identical for every item of the batch)
    int tnx, tny, tnz, ntiles;  // tiles per axis / per item (tile = K x 8 x 8 voxels)
    int K;                      // x-planes per lane
    int cs_log2, cs;            // cell edge in voxels (power of two >= cutoff radius)
    int h;                      // halo cells on each side of the grid
    int ncx, ncy, ncz, ncell;   // padded cell grid per item
    int cstride;                // cell-array stride per item = ncell + 1 (the extra word is the item's end marker)
    int cls_per_item;           // 1: one sigma-class table per item (k_prepass_items), 0: one for the call
    int rint;                   // integer upper bound of the cutoff radius in voxels
    int C, G;                   // channels, channel groups = ceil(C/8)
    int B;                      // items (molecules / poses / frames)
    int pbc;                    // 1: per-item orthorhombic box given
    int force_general;          // 1: never use the class-sorted path (testing / A-B)
    float R2;                   // cutoff^2 in voxel units  (25 / res^2)
    float R2cull;               // slightly inflated cutoff^2 for tile culling
    double inv_res;             // 1 / voxelsize
    double w_scale;             // voxelsize^2  (w = w_scale / sigma^2 -> q is in A^2/A^2)
    double Rp;                  // image acceptance radius, voxel units (cutoff + margin)
    long long V;                // nx*ny*nz
    unsigned M;                 // capacity of the record arrays (= total atoms x img_cap)
    int img_cap;                // temp / record slots reserved per atom (1 unless periodic)
    int prepass_hurry;          // 1: binning / fill waves raise their issue priority (pipeline.h, run_lattice)
    // exact cut-off decisions for wide sigmas (k_tail's fix-up waves)
    double res;                 // voxelsize
    float w_exact_max;          // an entry with w below this has a value step > 5e-6 at the cutoff (sigma > 1.81 A)
    // tolerance-aware reach (opt-in, mkamd_ctx_set_value_tolerance; 0 = off, the reference's hard 5 A everywhere):
    // an entry contributes less than eps beyond w * d^2 = reach_tau = eps^(-1/6), so its record carries a reach LEVEL
    // (bits 30-31 of the packed cell word) and the tile culls it at reach^2 = (1 - REACH_STEP * level) * cutoff^2
    float reach_tau;
    // direct binning (round 3; k_bin_direct): records live at cell * cell_cap + rank, no scan and no fill pass.
    // direct_words points at the call's direct counters [B * cstride] followed by DIRECT_WORDS control words; nullptr = the
    // call has no direct pass.  Whether the records ARE in that layout is decided on the device (word DIRECT_FAILED).
    int cell_cap;
    unsigned spill_base, spill_cap;   // spill areas: item b owns slots [spill_base + b * spill_cap, + spill_cap), counted in its spare cell counter
    const unsigned* direct_words;
    // counter of cell i = direct_words[DIRECT_HEAD + (i << cnt_shift)]: the one-launch pre-pass of ONE molecule sends 50 000
    // rank atomics to ~1 000 counters, and device-scope atomics on neighbouring words serialise (packed 4 bytes apart they
    // were 9 us of a 13 us kernel) -- k_bin_solo keeps them 32 bytes apart (cnt_shift = 3; 64 and 128 bytes: the same time)
    int cnt_shift;
    // topology reuse (round 5; pipeline.h TopologyDev): != 0 -- every item of the call is one set of coordinates of the SAME
    // molecule of topo_n atoms (the frames of a trajectory), whose per-atom class ids / compact channel words / class table
    // were built once from its sigmas: sigma-side arrays are indexed by the atom's index INSIDE its item
    long long topo_n;
    unsigned topo_wide;         // atoms of that molecule with a sigma wide enough for the exact cut-off fix-up (the handle lists them)
};
enum { DIRECT_FAILED = 0, DIRECT_SPILLED = 1, DIRECT_WORDS = 4, DIRECT_HEAD = 32 /* words in front of the counters (one 128-byte line) */ };
constexpr float REACH_STEP = 0.17f;   // levels 0..3: reach 5.00 / 4.56 / 4.06 / 3.50 A at the 5 A cutoff (H at eps = 1e-6: 3.48 A)

// r2 (a cutoff^2) scaled down to the reach of one record; untouched -- not even multiplied by one -- unless the call
// runs with a value tolerance (a scalar branch around the arithmetic: the exact mode pays a register move at most)
MK_DEV float reach_r2(const GridDesc& g, float r2, int packed_cell)
{
    if (g.reach_tau > 0.f) {
        mk_stay_in_branch();             // (left alone the compiler turns this into five instructions and a select, run always)
        r2 *= 1.f - REACH_STEP * (float)((unsigned)packed_cell >> 30);
    }
    return r2;
}

// is the call's record array in the direct layout (k_bin_direct succeeded)?  One scalar load per wave.
MK_DEV bool direct_layout(const GridDesc& g)
{
    return g.direct_words != nullptr && mk_uniform(g.direct_words[DIRECT_FAILED]) == 0u;
}

// w of a present channel is clamped to a finite value (+inf is the "channel absent" marker): a
// vanishing sigma then still yields 1 exactly on a voxel centre (d = 0) and 0 elsewhere, as
// occupancy_utils.pyx:57-60 does.
constexpr float MK_W_MAX = 3.0e38f;
constexpr double CUTOFF2_A = 25.0;   // occupancy_utils.pyx:53, in A^2
constexpr double CUTOFF_A_KERNEL = 5.0;

enum { MK_ERR_RECORD_OVERFLOW = 1, MK_ERR_BAD_BOX = 2, MK_ERR_TOO_MANY_IMAGES = 4, MK_ERR_TOPOLOGY = 8 /* an item is not topo_n atoms long */ };

// w = voxelsize^2 / sigma^2 of one (atom, channel); +inf when the atom is not in the channel
// (sigma == 0, occupancy_utils.pyx:55-56) or sigma is NaN (the reference never stores NaN).
template <typename SigT>
MK_DEV float sigma_to_w(SigT sigma, double w_scale)
{
    const double s = (double)sigma;
    const float t = (float)(w_scale / (s * s));
    return (s != 0.0 && t == t) ? fminf(t, MK_W_MAX) : mk_inf();
}

// w of the (up to) CHG channels [c0, c0+CHG) of one atom.  An atom almost always carries ONE radius in
// all the channels it has (sigma = radius x mask), so the double-precision division is done once for the
// lane's first usable sigma and only atoms with several distinct sigmas take the per-channel path.
// Values are exactly sigma_to_w's.
// Besides the CHG values it returns the atom's COMPACT channel description, which the binning parks per (atom, group)
// so that the fill pass needs neither the sigmas nor the division again:
//   cw.x = bit pattern of w0 (the lane's first usable sigma; CLS_EMPTY when there is none)
//   cw.y = nibble mask, nibble j = 1 when channel c0+j carries w0; ATOM_MULTI_SIGMA when the atom has several distinct
//          sigmas (rare), which sends the fill pass back to the sigma row.
constexpr unsigned ATOM_MULTI_SIGMA = 0xffffffffu;

// (in two halves -- the loads, then the arithmetic -- so that a caller can have the row in flight beside its other loads)
template <typename SigT>
MK_DEV void load_channel_sigmas(const SigT* __restrict__ row, int c0, int C, SigT (&s)[CHG])
{
    // a full group of channels in a 16-byte aligned row (C = 8, the reference's channel set): 16-byte loads -- as eight
    // separate dwords with the `c0 + j < C` branches between them the wave asked the L1 for four times the lines
    struct alignas(16) Vec16 { SigT v[16 / sizeof(SigT)]; };
    constexpr int PER = 16 / (int)sizeof(SigT);
    if (c0 + CHG <= C && (reinterpret_cast<uintptr_t>(row + c0) & (uintptr_t)15) == 0) {
#pragma unroll
        for (int i = 0; i < CHG / PER; ++i) {
            const Vec16 v = reinterpret_cast<const Vec16*>(row + c0)[i];
#pragma unroll
            for (int j = 0; j < PER; ++j) s[i * PER + j] = v.v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < CHG; ++j) s[j] = (c0 + j < C) ? row[c0 + j] : (SigT)0;
    }
}

template <typename SigT>
MK_DEV uint2 channel_w_of(const SigT (&s)[CHG], double w_scale, float (&w)[CHG])
{
    SigT s0 = (SigT)0;
#pragma unroll
    for (int j = CHG - 1; j >= 0; --j) s0 = (s[j] != (SigT)0) ? s[j] : s0;
    const float w0 = sigma_to_w(s0, w_scale);                    // +inf when the atom has no channel here
    bool other = false;
    unsigned nib = 0u;
#pragma unroll
    for (int j = 0; j < CHG; ++j) {
        w[j] = (s[j] == s0) ? w0 : mk_inf();
        nib |= (s[j] == s0) ? (1u << (4 * j)) : 0u;
        other |= (s[j] != s0) && (s[j] != (SigT)0);
    }
    if (other) {                                                 // rare: several distinct sigmas (or NaN)
#pragma unroll
        for (int j = 0; j < CHG; ++j)
            if ((s[j] != s0) && (s[j] != (SigT)0)) w[j] = sigma_to_w(s[j], w_scale);
    }
    const bool none = !(w0 < mk_inf());
    return make_uint2(none ? CLS_EMPTY : mk_float_bits(w0), other ? ATOM_MULTI_SIGMA : (none ? 0u : nib));
}

template <typename SigT>
MK_DEV uint2 atom_channel_w(const SigT* __restrict__ row, int c0, int C, double w_scale, float (&w)[CHG])
{
    SigT s[CHG];
    load_channel_sigmas(row, c0, C, s);
    return channel_w_of(s, w_scale, w);
}

// Class table: NCLS slots of w bit patterns (CLS_EMPTY = unused), word NCLS = overflow marker.
// ------------------------------------------------------------------------------------------------
// Class discovery: the distinct w values of the batch -> cls_table (<= NCLS, else the overflow word
// is raised and the call takes the general tile path).  Two tiny kernels and NO global atomics (a
// shared table updated by thousands of blocks serialises on one cache line):
//   k_bin_count       : (it reads the sigmas anyway) duplicates removed per wave (ballot / readlane
//                       election) and per block (LDS hash set); each block stores its set (<= 32 values)
//   k_merge_classes   : two levels: 64-row slices -> 64-word sets, then one block -> the final table
// ------------------------------------------------------------------------------------------------
constexpr int CLS_BLOCK_SET = 32;

MK_DEV unsigned class_hash(unsigned bits) { return (bits >> 9) ^ (bits >> 15) ^ (bits >> 21); }

// insert into an LDS open-addressing set of `size` (power of two) slots; false when the set is full
MK_DEV bool lds_set_insert(unsigned* set, unsigned size, unsigned bits)
{
    unsigned h = class_hash(bits) & (size - 1u);
    for (unsigned probe = 0; probe < size; ++probe) {
        if (set[h] == bits) return true;                               // common case: no atomic needed
        const unsigned old = mk_lds_cas(&set[h], CLS_EMPTY, bits);
        if (old == CLS_EMPTY || old == bits) return true;
        h = (h + 1u) & (size - 1u);
    }
    return false;
}

// wave-cooperative registration of up to 8 w bit patterns per lane into the block's LDS set.
// Must be called by ALL lanes of the wave (inactive lanes pass CLS_EMPTY everywhere).
MK_DEV void wave_register_value(unsigned bits, unsigned* s_set, unsigned* s_full)
{
    const int lane = threadIdx.x & (WAVE - 1);
    bool pending = bits != CLS_EMPTY;
    for (;;) {                                                   // wave-uniform: one trip per distinct value
        const unsigned long long mask = mk_ballot(pending);
        if (mask == 0ull) break;
        const int leader = __builtin_ctzll(mask);
        const unsigned lb = mk_readlane(bits, leader);
        if (bits == lb) pending = false;
        if (lane == leader && !lds_set_insert(s_set, CLS_BLOCK_SET, lb)) *s_full = 1u;
    }
}

// `first` = the lane's w0 bits (most atoms carry ONE radius in all their channels: that is their only value),
// `multi` = the lane's atom has several distinct sigmas, its other values are in wb[].
MK_DEV void wave_register_classes(unsigned first, bool multi, const unsigned (&wb)[CHG], unsigned* s_set, unsigned* s_full)
{
    // A value that already sits in its home slot of the block's set needs nothing: one LDS read per lane instead of an
    // election trip per distinct value -- after the block's first wave that is nearly every lane (a value displaced by
    // a collision, or one being inserted by another wave right now, simply takes the election as before).
    const unsigned first0 = first;
    if (first != CLS_EMPTY && s_set[class_hash(first) & (unsigned)(CLS_BLOCK_SET - 1)] == first) first = CLS_EMPTY;
    // the lane's first value goes through one election loop (a handful of trips per wave) ...
    wave_register_value(first, s_set, s_full);
    // ... and only waves holding atoms with SEVERAL distinct sigmas pay for the remaining slots
    if (mk_ballot(multi) != 0ull) {
#pragma unroll
        for (int j = 0; j < CHG; ++j) wave_register_value((multi && wb[j] != first0) ? wb[j] : CLS_EMPTY, s_set, s_full);
    }
}

constexpr unsigned CLS_TOO_MANY = 0xfffffffeu;     // a set that overflowed reports this marker instead of values
constexpr int MERGE_SET = 64;                      // capacity of the merge kernels' LDS set (> NCLS)

// Merge `rows_per_block` sets of `row_words` words each into one 64-word set per block.
// Level 1: grid = ceil(nrows / rows_per_block) blocks over the per-block sets of k_bin_count;
// level 2: one block over level 1's output, which also writes the final class table.
MK_DEV void merge_classes_block(const unsigned* __restrict__ rows, unsigned nrows, unsigned row_words,
                                unsigned rows_per_block, unsigned* __restrict__ out_sets,
                                unsigned* __restrict__ cls_table /* non-null: final level */, unsigned block)
{
    __shared__ unsigned s_set[MERGE_SET];
    __shared__ unsigned s_over;
    if (threadIdx.x < MERGE_SET) s_set[threadIdx.x] = CLS_EMPTY;
    if (threadIdx.x == 0) s_over = 0u;
    mk_block_sync();
    const unsigned r0 = block * rows_per_block;
    const unsigned r1 = r0 + rows_per_block < nrows ? r0 + rows_per_block : nrows;
    const unsigned w0 = r0 * row_words, w1 = r1 * row_words;               // multiples of 4 words
    const uint4* __restrict__ v4 = reinterpret_cast<const uint4*>(rows);
    for (unsigned i = w0 / 4u + threadIdx.x; i < w1 / 4u; i += blockDim.x) {
        const uint4 q4 = v4[i];
        const unsigned vv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v = vv[j];
            if (v == CLS_EMPTY) continue;
            if (v == CLS_TOO_MANY || !lds_set_insert(s_set, (unsigned)MERGE_SET, v)) s_over = 1u;
        }
    }
    mk_block_sync();
    if (cls_table == nullptr) {
        if (threadIdx.x < MERGE_SET)
            out_sets[(size_t)block * MERGE_SET + threadIdx.x] = s_over ? CLS_TOO_MANY : s_set[threadIdx.x];
    } else if (threadIdx.x == 0) {
        unsigned n = 0;
        bool over = s_over != 0u;
        for (int i = 0; i < MERGE_SET; ++i) {
            const unsigned v = s_set[i];
            if (v == CLS_EMPTY) continue;
            if (n < (unsigned)NCLS) cls_table[n] = v;
            ++n;
        }
        if (n > (unsigned)NCLS) over = true;
        for (unsigned i = n; i < (unsigned)NCLS; ++i) cls_table[i] = CLS_EMPTY;
        cls_table[CLS_OVERFLOW] = over ? 0u : CLS_EMPTY;
    }
}

MK_KERNEL(256) void k_merge_classes(const unsigned* __restrict__ rows, unsigned nrows, unsigned row_words,
                                    unsigned rows_per_block, unsigned* __restrict__ out_sets,
                                    unsigned* __restrict__ cls_table /* non-null: final level */)
{
    mk_wave_priority_high();
    merge_classes_block(rows, nrows, row_words, rows_per_block, out_sets, cls_table, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Binning: atoms (and, for periodic items, their images) -> padded uniform cell grid.
//   k_bin_count : one thread per atom.  Decomposes the position IN DOUBLE into (cell, cell-centre-
//                 relative float32 offset), takes its rank inside the cell from ONE returning
//                 atomicAdd on the cell's counter and parks (offset, cell, rank) in a temp record
//                 (img_cap temp slots per atom; unused slots are marked).
//   (exclusive scan of the counts -> cell starts)
//   k_bin_fill  : one thread per temp slot: a pure permutation into the cell-sorted record arrays
//                 (no atomics, nothing recomputed) + the per-channel sigma classes of the atom.
// Record = pos (cell-centre-relative x,y,z as f32 ; packed padded cell coords)
//          + per channel group EITHER 8 class ids (4 bits each, 0 = absent)     [sorted path]
//                              OR two float4 of w (+inf = absent)               [general path]
// ------------------------------------------------------------------------------------------------
constexpr unsigned TMP_UNUSED = 0xffffffffu;

// One atom of the binning: its channels' w (drop test + class discovery), its position decomposed in double, one
// temp record per (periodic) image inside grid+halo with the rank `rank_in_cell(cell)` hands out.  Called by all
// lanes of a wave together (`act` = this lane holds an atom): the class registration is wave-cooperative.
// Rank of each wanted lane's atom inside its cell with ONE global atomic per distinct cell of the wave (called by all
// 64 lanes together).  Atom lists of real systems are spatially coherent -- the atoms of a residue or a water follow
// each other -- and same-address device atomics serialise: with one atomic per atom a spatially ordered cfg2 list
// bins in 139 us against 77 us for a shuffled one.  Up to WAVE_RANK_ROUNDS distinct cells are grouped (leader = lowest
// lane of the group, it adds the group's size; members rank by lane order); lanes left over after that -- a wave of a
// shuffled list holds ~64 distinct cells -- add for themselves as before.  No round waits for an atomic: the bases are
// fetched from the leaders after the loop.
constexpr int WAVE_RANK_ROUNDS = 8;

MK_DEV unsigned wave_rank_in_cell(bool want, unsigned cell, unsigned* __restrict__ cell_count)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = mk_ballot(want);
    int leader_lane = lane;                 // default: every wanted lane is its own group
    unsigned offset = 0u, group = 1u;
    int grouped = 0;                        // lanes saved so far (wave-uniform)
    for (int it = 0; it < WAVE_RANK_ROUNDS && todo != 0ull; ++it) {         // wave-uniform
        const int l0 = mk_ctz64(todo);
        const unsigned c0 = mk_readlane(cell, l0);
        const unsigned long long m = mk_ballot(want && cell == c0) & todo;  // contains l0
        if ((m >> lane) & 1ull) {
            leader_lane = l0;
            offset = (unsigned)mk_popc64(m & below);
            group = (unsigned)mk_popc64(m);
        }
        todo &= ~m;
        grouped += mk_popc64(m) - 1;
        if (it == 1 && grouped == 0) break;      // two singleton groups in a row: a shuffled list, stop looking
    }
    unsigned base = 0u;
    if (want && leader_lane == lane) base = mk_atomic_add(&cell_count[cell], group);
    base = mk_shfl(base, leader_lane);
    return base + offset;
}

// largest b in [lo, B) with atom_offsets[b] <= a
MK_DEV int item_of_atom(const long long* __restrict__ atom_offsets, int B, long long a, int lo)
{
    int hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (atom_offsets[mid] <= a) lo = mid; else hi = mid;
    }
    return lo;
}

// the same for the first and the last atom of a block, without the two chains of dependent loads in the common cases:
// items of equal size (the guess a * B / total is right: two independent loads confirm it) and a block inside one item
MK_DEV void items_of_block(const long long* __restrict__ atom_offsets, int B, long long total_atoms, long long a_first,
                           long long a_last, int& b_lo, int& b_hi)
{
    if (B == 1) { b_lo = b_hi = 0; return; }                       // one item per call (the reference's usage): nothing to look up
    // (a guess, verified below: single precision and v_rcp_f32 -- the double-precision division was a dozen f64 instructions per wave)
    int b = (int)((float)a_first * ((float)B * mk_rcp((float)(total_atoms > 0 ? total_atoms : 1))));
    b = b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
    const long long o0 = atom_offsets[b], o1 = atom_offsets[b + 1];
    long long next = o1;
    if (o0 <= a_first && a_first < o1) b_lo = b;
    else { b_lo = item_of_atom(atom_offsets, B, a_first, 0); next = atom_offsets[b_lo + 1]; }
    b_hi = a_last < next ? b_lo : item_of_atom(atom_offsets, B, a_last, b_lo);
}

// PBC: 0 = open boundaries, 1 = periodic, -1 = decided at run time (g.pbc).  The periodic image loop costs ~20 VGPRs;
// k_bin_count is compiled for both so that the common open-boundary kernel stays small enough to run beside the tile kernel.
// TOPO (round 5, GridDesc::topo_n): the call's items are frames of ONE molecule whose class ids were built once
// (k_topology_ids): `sigmas` then points at those ids (unsigned [topo_n, G]) -- no sigma row, no double-precision division,
// no class discovery; the ids are parked (tmp_cls as unsigned [atoms, G]) for the fill pass.
template <typename SigT, int PBC, bool TOPO = false, class RankOne, class RankWave>
MK_DEV void bin_atom(const GridDesc& g, long long a, bool act, int b_lo, int b_hi, const float* __restrict__ coords,
                     const long long* __restrict__ atom_offsets, const SigT* __restrict__ sigmas,
                     const double* __restrict__ origins, const float* __restrict__ box, const double* __restrict__ affine,
                     float4* __restrict__ tmp_pos, uint2* __restrict__ tmp_idx, uint2* __restrict__ tmp_cls,
                     int* __restrict__ err_flag,
                     bool classes, unsigned* s_set, unsigned* s_full, RankOne&& rank_one, RankWave&& rank_wave)
{
    const bool pbc = PBC < 0 ? (g.pbc != 0) : (PBC != 0);
    bool any = false;
    int b = 0;
    if constexpr (TOPO) {
        if (act) {
            int lo = b_lo, hi = b_hi + 1;                        // (nearly always b_lo == b_hi: nothing is searched)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (atom_offsets[mid] <= a) lo = mid; else hi = mid;
            }
            b = lo;
            const long long first = atom_offsets[b];
            if (atom_offsets[b + 1] - first != g.topo_n) {
                mk_atomic_or(err_flag, MK_ERR_TOPOLOGY);         // not a frame of the molecule the topology was built from
            } else {
                const unsigned* __restrict__ ids = reinterpret_cast<const unsigned*>(sigmas) + (size_t)(a - first) * g.G;
                unsigned* __restrict__ park_ids = reinterpret_cast<unsigned*>(tmp_cls) + (size_t)a * g.G;
                for (int gq = 0; gq < g.G; ++gq) {
                    const unsigned v = ids[gq];
                    any |= v != 0u;
                    mk_tmp_store(&park_ids[gq], v);
                }
            }
        }
    }
    // ---- the atom's channels: w bit patterns (one pass over the sigmas serves the drop test AND the class
    //      discovery); registration is wave-cooperative, so every lane takes part ----
    for (int c0 = 0; !TOPO && c0 < g.C; c0 += CHG) {
        unsigned wb[CHG];
        float w[CHG];
#pragma unroll
        for (int j = 0; j < CHG; ++j) w[j] = mk_inf();
        uint2 cw = make_uint2(CLS_EMPTY, 0u);
        if (act) {
            cw = atom_channel_w(sigmas + (size_t)a * g.C, c0, g.C, g.w_scale, w);
            mk_tmp_store(&tmp_cls[(size_t)a * g.G + (c0 / CHG)], cw);              // parked for the fill pass
        }
        const bool multi = cw.y == ATOM_MULTI_SIGMA;
#pragma unroll
        for (int j = 0; j < CHG; ++j) wb[j] = (w[j] < mk_inf()) ? mk_float_bits(w[j]) : CLS_EMPTY;
        any |= cw.x != CLS_EMPTY;
        if (multi) {
#pragma unroll
            for (int j = 0; j < CHG; ++j) any |= wb[j] != CLS_EMPTY;
        }
        if (classes) wave_register_classes(cw.x, multi, wb, s_set, s_full);
    }

    const size_t t0 = (size_t)(act ? a : 0) * (size_t)g.img_cap;     // this atom's temp slots
    int used = 0;
    // an atom with no usable channel is dropped here (occupancy_utils.pyx:55-56)
    bool drop = !act || !any;
    double p[3] = {0.0, 0.0, 0.0}, Lv[3] = {0.0, 0.0, 0.0};
    int k0[3] = {0, 0, 0}, k1[3] = {0, 0, 0};
    if (!drop) {
        // item of this atom: the largest b in [b_lo, b_hi] with atom_offsets[b] <= a (the caller narrows the range
        // with wave-uniform look-ups: nearly always b_lo == b_hi and nothing is searched per lane)
        if constexpr (!TOPO) {
            int lo = b_lo, hi = b_hi + 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (atom_offsets[mid] <= a) lo = mid; else hi = mid;
            }
            b = lo;
        }
        const int nvox[3] = {g.nx, g.ny, g.nz};
        // fused augmentation (tools/voxeldescriptors.py:78-114 rotateCoordinates, then the astype(float32) of
        // _getOccupancyC :519): x' = M x + t in double, rounded to float32 like the reference pipeline does
        // (asking for the position up front, together with the sigmas, was measured: no gain -- the waves spend 10 % of
        //  their cycles in s_waitcnt, the kernel is not waiting for memory)
        float xyz[3] = {coords[3 * a + 0], coords[3 * a + 1], coords[3 * a + 2]};
        if (affine != nullptr) {
            const double* A = affine + 12 * (size_t)b;
            const double x = (double)xyz[0], y = (double)xyz[1], z = (double)xyz[2];
            xyz[0] = (float)(A[0] * x + A[1] * y + A[2] * z + A[9]);
            xyz[1] = (float)(A[3] * x + A[4] * y + A[5] * z + A[10]);
            xyz[2] = (float)(A[6] * x + A[7] * y + A[8] * z + A[11]);
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            p[ax] = ((double)xyz[ax] - origins[3 * (size_t)b + ax]) * g.inv_res;
            if (pbc) {
                const double L = (double)box[3 * (size_t)b + ax] * g.inv_res;
                if (!(L > 2.0 * (g.Rp - 1e-3))) { mk_atomic_or(err_flag, MK_ERR_BAD_BOX); drop = true; }
                Lv[ax] = L;
                // the range of images, ceil / floor of two QUOTIENTS by the box length: a quotient's integer part only depends
                // on how it is rounded within ~1e-16 of a whole number, so the quotients are formed with a single-precision
                // reciprocal (relative error < 2e-7; |quotient| is a few units) and only a lane within 1e-5 of a whole number
                // takes the double-precision divisions (2 of ~50 000 atoms) -- the same integers, four f64 divisions per
                // atom fewer (round 5: they were a quarter of the periodic binning's instructions)
                const double invL = (double)mk_rcp((float)L);
                const double n0 = -g.Rp - p[ax], n1 = (double)(nvox[ax] - 1) + g.Rp - p[ax];
                double q0 = n0 * invL, q1 = n1 * invL;
                if (!(fabs(q0 - rint(q0)) > 1e-5 * (1.0 + fabs(q0))) || !(fabs(q1 - rint(q1)) > 1e-5 * (1.0 + fabs(q1)))) { q0 = n0 / L; q1 = n1 / L; }
                const double a0 = ceil(q0);
                const double a1 = floor(q1);
                if (a1 - a0 > 64.0) { mk_atomic_or(err_flag, MK_ERR_TOO_MANY_IMAGES); drop = true; }
                k0[ax] = (int)a0; k1[ax] = (int)a1;              // empty range when a1 < a0
            } else {
                if (p[ax] < -g.Rp || p[ax] > (double)(nvox[ax] - 1) + g.Rp) drop = true;
            }
        }
    }
    const double cmid = 0.5 * (double)(g.cs - 1);                     // (the cell edge is a power of two: ldexp divides by it exactly)
    const int nc[3] = {g.ncx, g.ncy, g.ncz};
    // cell and cell-relative offset of one image position q
    auto locate = [&](const double (&q)[3], int (&pc)[3], float (&rel)[3]) {
        bool inside = true;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int ci = (int)floor(ldexp(q[ax] + 0.5, -g.cs_log2));   // unpadded cell index
            pc[ax] = ci + g.h;
            inside &= (pc[ax] >= 0) && (pc[ax] < nc[ax]);
            rel[ax] = (float)(q[ax] - ((double)ci * (double)g.cs + cmid));
        }
        return inside;
    };
    auto park = [&](const int (&pc)[3], const float (&rel)[3], size_t cell, unsigned rank) {
        mk_tmp_store(&tmp_pos[t0 + used], make_float4(rel[0], rel[1], rel[2], mk_int_as_float(pc[0] | (pc[1] << 10) | (pc[2] << 20))));
        mk_tmp_store(&tmp_idx[t0 + used], make_uint2((unsigned)cell, rank));
        ++used;
    };
    if (!pbc) {
        // one position per atom: every lane takes part in the ranking (consecutive atoms of one cell share an atomic)
        int pc[3] = {0, 0, 0};
        float rel[3] = {0.f, 0.f, 0.f};
        const bool want = !drop && locate(p, pc, rel);
        const size_t cell = want ? (size_t)b * g.cstride + ((size_t)pc[0] * g.ncy + pc[1]) * g.ncz + pc[2] : (size_t)0;
        const unsigned rank = rank_wave(want, cell);
        if (want) park(pc, rel, cell, rank);
    } else {
        // periodic: the images p + k L inside grid + halo.  Nearly every atom has exactly one, so the FIRST image of
        // every lane is ranked together (one atomic per distinct cell of the wave, like above) and only further images
        // (grids wider than the box) fall back to one atomic each.
        // (the images are walked with three counters, z fastest -- x slowest, the order of the old loops; an index that
        //  is divided back into (kx, ky, kz) costs three integer divisions per image and a dozen registers)
        const bool none = drop || k1[0] < k0[0] || k1[1] < k0[1] || k1[2] < k0[2];
        int kx = k0[0], ky = k0[1], kz = k0[2];
        bool more = !none;                                           // (kx, ky, kz) is an image not looked at yet
        auto image = [&](int (&pc)[3], float (&rel)[3]) {            // look at the current image, step to the next
            const double q[3] = {p[0] + kx * Lv[0], p[1] + ky * Lv[1], p[2] + kz * Lv[2]};
            if (++kz > k1[2]) { kz = k0[2]; if (++ky > k1[1]) { ky = k0[1]; if (++kx > k1[0]) more = false; } }
            return locate(q, pc, rel);
        };
        int pc[3] = {0, 0, 0};
        float rel[3] = {0.f, 0.f, 0.f};
        bool want = false;
        while (more && !want) want = image(pc, rel);                 // per-lane trip count, no cross-lane work inside
        const size_t cell0 = want ? (size_t)b * g.cstride + ((size_t)pc[0] * g.ncy + pc[1]) * g.ncz + pc[2] : (size_t)0;
        const unsigned rank0 = rank_wave(want, cell0);                // all lanes
        if (want) park(pc, rel, cell0, rank0);
        while (more) {
            if (!image(pc, rel)) continue;
            if (used >= g.img_cap) { mk_atomic_or(err_flag, MK_ERR_RECORD_OVERFLOW); continue; }
            const size_t cell = (size_t)b * g.cstride + ((size_t)pc[0] * g.ncy + pc[1]) * g.ncz + pc[2];
            park(pc, rel, cell, rank_one(cell));
        }
    }
    if (act)
        for (int i = used; i < g.img_cap; ++i) tmp_idx[t0 + i] = make_uint2(TMP_UNUSED, 0u);
}

template <typename SigT, int PBC, bool SHARED = false, bool TOPO = false>
MK_KERNEL(256) void k_bin_count(GridDesc g, const float* __restrict__ coords,
                                const long long* __restrict__ atom_offsets, long long total_atoms,
                                const SigT* __restrict__ sigmas, const double* __restrict__ origins,
                                const float* __restrict__ box, const double* __restrict__ affine,
                                unsigned* __restrict__ cell_count,
                                float4* __restrict__ tmp_pos, uint2* __restrict__ tmp_idx, uint2* __restrict__ tmp_cls,
                                unsigned* __restrict__ block_sets, int* __restrict__ err_flag,
                                unsigned nblk /* blocks of 256 atoms; the grid may be smaller: see below */)
{
    if (direct_layout(g)) return;                        // the direct pass of this call has binned everything (block-uniform)
    if (g.prepass_hurry) mk_wave_priority_high();
    __shared__ unsigned s_set[CLS_BLOCK_SET];
    __shared__ unsigned s_full;
    const bool classes = !TOPO && !g.force_general;           // (a topology call: the classes are the topology's, nothing to discover)
    // One workgroup per 256 atoms -- except behind a direct pass (SHARED), where this kernel is only the fall-back: the
    // launch is then a few thousand workgroups that share the blocks (leaving at once when the pass succeeded costs 5 us,
    // not the 39 us of 50 000 empty workgroups).  A separate instance: as a loop the kernel needs 80 VGPRs, and the one that
    // runs beside the previous call's tile kernel must stay within 48 (tests/test_register_budgets.py).
    for (unsigned lb = blockIdx.x; lb < (SHARED ? nblk : blockIdx.x + 1u); lb += (SHARED ? gridDim.x : 1u)) {
    if (classes) {
        if (threadIdx.x < CLS_BLOCK_SET) s_set[threadIdx.x] = CLS_EMPTY;
        if (threadIdx.x == 0) s_full = 0u;
        mk_block_sync();
    }
    // (block i -> atoms [256 i, ...).  Giving concurrent blocks S different regions of the batch, so that their rank
    //  atomics spread over the whole counter array, was measured -- S = 16 / 64 / 256: k_bin_count 333 -> 382 us at 64, the
    //  in-order step 2.56 -> 2.80 / 2.60 / 2.53 ms: the batch binning is not bound by where its atomics land, unlike the
    //  one-molecule call, see k_bin_solo)
    // the items of the block's first and last atom (block-uniform: scalar loads); one block rarely spans several
    const long long a_first = (long long)lb * blockDim.x;
    const long long a_last = (a_first + blockDim.x < total_atoms ? a_first + blockDim.x : total_atoms) - 1;
    int b_lo, b_hi;
    items_of_block(atom_offsets, g.B, total_atoms, a_first, a_last, b_lo, b_hi);
    const long long a = a_first + threadIdx.x;
    bin_atom<SigT, PBC, TOPO>(g, a, a < total_atoms, b_lo, b_hi, coords, atom_offsets, sigmas, origins, box, affine, tmp_pos, tmp_idx, tmp_cls, err_flag,
                   classes, s_set, &s_full, [&](size_t cell) { return mk_atomic_add(&cell_count[cell], 1u); },
                   [&](bool want, size_t cell) { return wave_rank_in_cell(want, (unsigned)cell, cell_count); });
    if (classes) {
        mk_block_sync();
        if (threadIdx.x < CLS_BLOCK_SET)
            block_sets[(size_t)lb * CLS_BLOCK_SET + threadIdx.x] = s_full ? CLS_TOO_MANY : s_set[threadIdx.x];
        if (SHARED && lb + gridDim.x < nblk) mk_block_sync();         // (the set is re-initialised for the next block)
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct binning (round 3): ONE pass instead of count -> reduce -> scan -> fill.  A cell owns `cell_cap` record slots;
// an atom's rank inside its cell (the same returning atomic as k_bin_count) IS its slot, so the record -- position and
// class ids -- is written in place: no temp records (32 B per atom written and read back), no scan, no permutation
// pass (44 + 20 B per atom instead of 132).  What the pass cannot know it assumes and checks:
//   * the class ids come from the class table the PREVIOUS call on this workspace left behind (the sigma classes of a
//     workload do not change from call to call); an atom whose sigma is not in it, an atom with several distinct sigmas,
//     an overflowed table;
//   * a cell that receives more atoms than it has slots sends the surplus to its item's spill area, which every tile of
//     the item reads as one more candidate run; a full spill area
// -- any of these raises DIRECT_FAILED, and the kernels of the count / scan / fill chain, which are enqueued behind this
// pass in every call and return at once while the word is clear, do the call the old way (and leave the new class
// table).  The tile kernels read the word too (find_candidate_runs).  Open boundaries, one channel group.
// Measured on cfg2 (256 x 50 000 atoms, same box, docs/EXPERIMENTS_r3.md): 390 us against 338 + 189 + 48 for count + fill +
// reductions (pre-pass traffic 0.9-1.2 GB instead of 1.7); the chain behind it is launched as <= 4 096 workgroups that leave
// at once (as one workgroup per block it cost 82 us to launch and leave), and the tile kernel reads 27 cell runs instead
// of 9-16 column runs (+4 %): in-order step 2.41 against 2.55 ms, pipelined 2.28 against 2.27 (nothing).  What bounds the pass
// is its 12.8 M scattered 16 + 4 B record stores, not its arithmetic or its round trips (docs/EXPERIMENTS_r3.md).  With an
// XCD-aware block order (an item's atoms binned by one XCD, as k_bin_fill does) the pass takes 594 us: the item's rank
// atomics then all arrive together.  AUTOMATIC for big calls that are not pipelined (pipeline.h: direct_big); the
// pipelined ones keep the chain, whose cost hides beside the previous call's tile kernel.
// ------------------------------------------------------------------------------------------------
template <typename SigT>
MK_KERNEL(256) void k_bin_direct(GridDesc g, const float* __restrict__ coords, const long long* __restrict__ atom_offsets,
                                 long long total_atoms, const SigT* __restrict__ sigmas, const double* __restrict__ origins,
                                 const double* __restrict__ affine, unsigned* __restrict__ counts /* + DIRECT_WORDS */,
                                 float4* __restrict__ rec_pos, unsigned* __restrict__ rec_cls, uint2* __restrict__ tmp_cls,
                                 const unsigned* __restrict__ cls_table, unsigned* __restrict__ block_sets)
{
    if (g.prepass_hurry) mk_wave_priority_high();
    __shared__ unsigned s_set[CLS_BLOCK_SET];
    MK_BIN_BEGIN();
    // A wave's life here is a chain of memory round trips (75 % of it is s_waitcnt), so the loads are ISSUED together and
    // waited for once: the position first (its address depends on the atom index alone: it flies during the set-up and the
    // item look-up), the origin as soon as the item is known, the sigma row last -- one round trip instead of three.
    // (All of them unconditional, from clamped indices: a load inside `if (act)` is followed by a wait where the branch
    //  ends -- the register allocator puts a copy there.)
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = a < total_atoms;
    const long long a_c = act ? a : total_atoms - 1;                            // (the kernel is not launched without atoms)
    const int lane = threadIdx.x & (WAVE - 1);
    float xyz[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) xyz[ax] = coords[3 * a_c + ax];
    // the class table of the previous call: lane s holds class s (CLS_EMPTY beyond the table's end)
    const unsigned tab_word = cls_table[lane < CLS_TABLE_WORDS ? lane : 0];
    __shared__ unsigned s_full;
    if (threadIdx.x < CLS_BLOCK_SET) s_set[threadIdx.x] = CLS_EMPTY;
    if (threadIdx.x == 0) s_full = 0u;
    mk_block_sync();
    unsigned* const words = counts;
    counts += DIRECT_HEAD;
    const long long a_first = (long long)blockIdx.x * blockDim.x;
    const long long a_last = (a_first + blockDim.x < total_atoms ? a_first + blockDim.x : total_atoms) - 1;
    int b_lo, b_hi;
    items_of_block(atom_offsets, g.B, total_atoms, a_first, a_last, b_lo, b_hi);
    MK_BIN_MARK(0);

    // ---- the atom's item and its origin ----
    int b = b_lo;
    if (b_hi != b_lo) {                                                        // block-uniform: the block straddles items
        int lo = b_lo, hi = b_hi + 1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (atom_offsets[mid] <= a_c) lo = mid; else hi = mid;
        }
        b = lo;
    }
    double org[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) org[ax] = origins[3 * (size_t)b + ax];

    // ---- channels: one radius per atom is the rule; its class id by election (a handful of distinct values per wave) ----
    float w[CHG];
    uint2 cw = atom_channel_w(sigmas + (size_t)a_c * g.C, 0, g.C, g.w_scale, w);
    if (!act) cw = make_uint2(CLS_EMPTY, 0u);
    if (act) tmp_cls[a] = cw;                                                  // (the fix-up waves of k_tail find an atom's sigma here)
    const unsigned tab = lane < CLS_TABLE_WORDS ? tab_word : CLS_EMPTY;
    bool failed = mk_readlane(tab, CLS_OVERFLOW) != CLS_EMPTY;                 // more classes than ids: that call took the general path
    const bool multi = cw.y == ATOM_MULTI_SIGMA;
    const bool any = cw.x != CLS_EMPTY && !multi;
    MK_BIN_MARK(1);
    unsigned wb[CHG];
#pragma unroll
    for (int j = 0; j < CHG; ++j) wb[j] = CLS_EMPTY;
    wave_register_classes(cw.x, false, wb, s_set, &s_full);                    // (the fix-up waves of k_tail read the blocks' sets)
    MK_BIN_MARK(2);
    unsigned id = 0u;
    {
        bool pending = any;
        for (;;) {                                                             // wave-uniform: one trip per distinct value
            const unsigned long long todo = mk_ballot(pending);
            if (todo == 0ull) break;
            const unsigned lb = mk_readlane(cw.x, mk_ctz64(todo));
            const unsigned long long hit = mk_ballot(lane < NCLS && tab == lb);
            const unsigned cid = hit ? (unsigned)mk_ctz64(hit) + 1u : 0u;
            if (pending && cw.x == lb) { id = cid; pending = false; }
            failed = failed || cid == 0u;
        }
    }
    failed = failed || mk_ballot(multi) != 0ull;
    MK_BIN_MARK(3);

    // ---- position -> (cell, cell-relative offset) in double, exactly as bin_atom does ----
    bool want = any;
    int pc[3] = {0, 0, 0};
    float rel[3] = {0.f, 0.f, 0.f};
    if (want) {
        const int nvox[3] = {g.nx, g.ny, g.nz};
        if (affine != nullptr) {
            const double* A = affine + 12 * (size_t)b;
            const double x = (double)xyz[0], y = (double)xyz[1], z = (double)xyz[2];
            xyz[0] = (float)(A[0] * x + A[1] * y + A[2] * z + A[9]);
            xyz[1] = (float)(A[3] * x + A[4] * y + A[5] * z + A[10]);
            xyz[2] = (float)(A[6] * x + A[7] * y + A[8] * z + A[11]);
        }
        const double cmid = 0.5 * (double)(g.cs - 1);
        const int nc[3] = {g.ncx, g.ncy, g.ncz};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double q = ((double)xyz[ax] - org[ax]) * g.inv_res;
            if (q < -g.Rp || q > (double)(nvox[ax] - 1) + g.Rp) want = false;
            const int ci = (int)floor(ldexp(q + 0.5, -g.cs_log2));
            pc[ax] = ci + g.h;
            want = want && (pc[ax] >= 0) && (pc[ax] < nc[ax]);
            rel[ax] = (float)(q - ((double)ci * (double)g.cs + cmid));
        }
    }
    const size_t cell = want ? (size_t)b * g.cstride + ((size_t)pc[0] * g.ncy + pc[1]) * g.ncz + pc[2] : (size_t)0;
    MK_BIN_MARK(4);
    const unsigned rank = wave_rank_in_cell(want, (unsigned)cell << g.cnt_shift, counts);
    MK_BIN_MARK(5);
    if (want) {
        unsigned slot;
        if (rank < (unsigned)g.cell_cap) {
            slot = (unsigned)cell * (unsigned)g.cell_cap + rank;
        } else {                                                               // the cell is full: the item's spill area
            const unsigned sp = mk_atomic_add(&counts[((size_t)b * g.cstride + g.ncell) << g.cnt_shift], 1u);     // (the item's spare counter)
            (void)mk_atomic_add(&words[DIRECT_SPILLED], 1u);                   // statistics
            slot = g.spill_base + (unsigned)b * g.spill_cap + (sp < g.spill_cap ? sp : 0u);
            if (sp >= g.spill_cap) { want = false; failed = true; }
        }
        if (want) {
            int level = 0;
            if (g.reach_tau > 0.f) {                                           // tolerance-aware reach (fill_record's rule)
                const float frac = g.reach_tau / (mk_uint_as_float(cw.x) * g.R2);
                level = (int)fminf(fmaxf(floorf((1.f - frac) * (1.f / REACH_STEP) - 1e-3f), 0.f), 3.f);
            }
            rec_pos[slot] = make_float4(rel[0], rel[1], rel[2], mk_int_as_float(pc[0] | (pc[1] << 10) | (pc[2] << 20) | (level << 30)));
            rec_cls[slot] = id * cw.y;                                         // ids <= 15: no carry between nibbles
        }
    }
    if (mk_ballot(failed) != 0ull && lane == 0) words[DIRECT_FAILED] = 1u;
    MK_BIN_MARK(6);
    mk_block_sync();
    if (threadIdx.x < CLS_BLOCK_SET)
        block_sets[(size_t)blockIdx.x * CLS_BLOCK_SET + threadIdx.x] = s_full ? CLS_TOO_MANY : s_set[threadIdx.x];
    MK_BIN_MARK(7);
}

// ------------------------------------------------------------------------------------------------
// The pre-pass of a SMALL call in one launch (round 3: one molecule per call, the reference's own call pattern --
// voxeldescriptors.py:251-365).  A grid-wide barrier inside one persistent kernel costs 12-25 us at 1 024 workgroups on
// this chip (one device-scope atomic per workgroup serialises at ~12 ns; tools/gridsync.hip) against ~3.3 us for a launch
// boundary, so the way to a short call is FEWER launches, not one: this kernel replaces count + class table / scan + fill
// (26 us for a 50 000-atom item, three boundaries) by the direct layout of k_bin_direct made infallible:
//   * the class table lives ACROSS calls; a w it does not hold yet is inserted on the spot (one CAS per wave and new
//     value: the first call of a workload pays a few hundred of them, later calls none);
//   * when the table is full the overflow word is raised: every record carries its w values too (rec_w), and the tile
//     kernels then take the general path as they do for the chain (k_tail empties the table afterwards, so that the
//     next call starts over);
//   * atoms with several distinct sigmas look every channel up;
//   * the spill area holds every atom of the call, so it cannot overflow.
// Nothing is left to fall back to: no chain is enqueued behind it.  The counters are zero when the call starts and
// k_tail zeroes them again (no memset launch).  Same arithmetic as bin_atom / fill_record: the records are the chain's.
// ------------------------------------------------------------------------------------------------
// class id (1..NCLS, 0 = the table is full) of one w bit pattern per lane (CLS_EMPTY: none).  `tab`: lane s holds what
// this wave knows of table slot s.  All lanes of the wave call it.
MK_DEV unsigned wave_class_id_insert(unsigned bits, unsigned& tab, unsigned* __restrict__ table)
{
    const int lane = threadIdx.x & (WAVE - 1);
    unsigned id = 0u;
    bool pending = bits != CLS_EMPTY;
    for (;;) {                                                                 // wave-uniform: one trip per distinct value
        const unsigned long long todo = mk_ballot(pending);
        if (todo == 0ull) break;
        const unsigned lb = mk_readlane(bits, mk_ctz64(todo));
        const unsigned long long hit = mk_ballot(lane < NCLS && tab == lb);
        unsigned cid = hit ? (unsigned)mk_ctz64(hit) + 1u : 0u;
        if (cid == 0u) {                                                       // not in the table as this wave knows it
            unsigned got = 0u;
            if (lane == 0) {
                for (unsigned sl = 0; sl < (unsigned)NCLS; ++sl) {
                    const unsigned old = mk_atomic_cas(&table[sl], CLS_EMPTY, lb);
                    if (old == CLS_EMPTY || old == lb) { got = sl + 1u; break; }
                }
                if (got == 0u) table[CLS_OVERFLOW] = 0u;                       // more classes than ids: the general path
            }
            cid = mk_readlane(got, 0);
            if (cid != 0u && lane == (int)cid - 1) tab = lb;
        }
        if (pending && bits == lb) { id = cid; pending = false; }
    }
    return id;
}

template <typename SigT>
MK_KERNEL(256) void k_bin_solo(GridDesc g, const float* __restrict__ coords, const long long* __restrict__ atom_offsets,
                               long long total_atoms, const SigT* __restrict__ sigmas, const double* __restrict__ origins,
                               const double* __restrict__ affine, unsigned* __restrict__ counts /* + DIRECT_WORDS */,
                               float4* __restrict__ rec_pos, float4* __restrict__ rec_w /* planes g.M apart */,
                               unsigned* __restrict__ rec_cls, uint2* __restrict__ tmp_cls,
                               unsigned* __restrict__ cls_table, unsigned* __restrict__ block_sets)
{
    __shared__ unsigned s_set[CLS_BLOCK_SET];
    __shared__ unsigned s_full;
    if (threadIdx.x < CLS_BLOCK_SET) s_set[threadIdx.x] = CLS_EMPTY;
    if (threadIdx.x == 0) s_full = 0u;
    mk_block_sync();
    unsigned* const words = counts;
    counts += DIRECT_HEAD;
    const int lane = threadIdx.x & (WAVE - 1);
    const long long a_first = (long long)blockIdx.x * blockDim.x;
    const long long a_last = (a_first + blockDim.x < total_atoms ? a_first + blockDim.x : total_atoms) - 1;
    const long long a = a_first + threadIdx.x;
    const bool act = a < total_atoms;
    // everything the atom needs from memory is asked for up front: the call's length is a chain of round trips
    unsigned tab = lane < CLS_TABLE_WORDS ? cls_table[lane] : CLS_EMPTY;
    float xyz[3] = {0.f, 0.f, 0.f};
    if (act) { xyz[0] = coords[3 * a + 0]; xyz[1] = coords[3 * a + 1]; xyz[2] = coords[3 * a + 2]; }
    const bool one = g.B == 1;                                                 // one molecule per call: no item to look up
    double org[3] = {0.0, 0.0, 0.0};
    if (one) { org[0] = origins[0]; org[1] = origins[1]; org[2] = origins[2]; }
    int b_lo = 0, b_hi = 0;
    if (!one) items_of_block(atom_offsets, g.B, total_atoms, a_first, a_last, b_lo, b_hi);
    float w[CHG];
#pragma unroll
    for (int j = 0; j < CHG; ++j) w[j] = mk_inf();
    uint2 cw = make_uint2(CLS_EMPTY, 0u);
    if (act) {
        cw = atom_channel_w(sigmas + (size_t)a * g.C, 0, g.C, g.w_scale, w);
        tmp_cls[a] = cw;                                                       // (the fix-up waves of k_tail find an atom's sigma here)
    }
    const bool multi = cw.y == ATOM_MULTI_SIGMA;
    unsigned wb[CHG];
    bool any = cw.x != CLS_EMPTY;
#pragma unroll
    for (int j = 0; j < CHG; ++j) {
        wb[j] = (w[j] < mk_inf()) ? mk_float_bits(w[j]) : CLS_EMPTY;
        any |= multi && wb[j] != CLS_EMPTY;
    }
    wave_register_classes(cw.x, multi, wb, s_set, &s_full);                    // (the fix-up waves of k_tail read the blocks' sets)
    // ---- class ids: one look-up per atom is the rule ----
    unsigned ids = wave_class_id_insert(multi ? CLS_EMPTY : cw.x, tab, cls_table) * (multi ? 0u : cw.y);   // ids <= 15: no carry between nibbles
    if (mk_ballot(multi) != 0ull) {                                            // wave-uniform, rare
#pragma unroll
        for (int j = 0; j < CHG; ++j) ids |= wave_class_id_insert(multi ? wb[j] : CLS_EMPTY, tab, cls_table) << (4 * j);
    }

    // ---- position -> (cell, cell-relative offset) in double, exactly as bin_atom does ----
    bool want = any;
    int pc[3] = {0, 0, 0};
    float rel[3] = {0.f, 0.f, 0.f};
    int b = 0;
    if (want) {
        if (!one) {
            int lo = b_lo, hi = b_hi + 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (atom_offsets[mid] <= a) lo = mid; else hi = mid;
            }
            b = lo;
            org[0] = origins[3 * (size_t)b + 0]; org[1] = origins[3 * (size_t)b + 1]; org[2] = origins[3 * (size_t)b + 2];
        }
        const int nvox[3] = {g.nx, g.ny, g.nz};
        if (affine != nullptr) {
            const double* A = affine + 12 * (size_t)b;
            const double x = (double)xyz[0], y = (double)xyz[1], z = (double)xyz[2];
            xyz[0] = (float)(A[0] * x + A[1] * y + A[2] * z + A[9]);
            xyz[1] = (float)(A[3] * x + A[4] * y + A[5] * z + A[10]);
            xyz[2] = (float)(A[6] * x + A[7] * y + A[8] * z + A[11]);
        }
        const double cmid = 0.5 * (double)(g.cs - 1);
        const int nc[3] = {g.ncx, g.ncy, g.ncz};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double q = ((double)xyz[ax] - org[ax]) * g.inv_res;
            if (q < -g.Rp || q > (double)(nvox[ax] - 1) + g.Rp) want = false;
            const int ci = (int)floor(ldexp(q + 0.5, -g.cs_log2));
            pc[ax] = ci + g.h;
            want = want && (pc[ax] >= 0) && (pc[ax] < nc[ax]);
            rel[ax] = (float)(q - ((double)ci * (double)g.cs + cmid));
        }
    }
    const size_t cell = want ? (size_t)b * g.cstride + ((size_t)pc[0] * g.ncy + pc[1]) * g.ncz + pc[2] : (size_t)0;
    const unsigned rank = wave_rank_in_cell(want, (unsigned)cell << g.cnt_shift, counts);
    if (want) {
        unsigned slot;
        if (rank < (unsigned)g.cell_cap) {
            slot = (unsigned)cell * (unsigned)g.cell_cap + rank;
        } else {                                                               // the cell is full: the item's spill area
            const unsigned sp = mk_atomic_add(&counts[((size_t)b * g.cstride + g.ncell) << g.cnt_shift], 1u);     // (the item's spare counter)
            (void)mk_atomic_add(&words[DIRECT_SPILLED], 1u);                   // statistics
            slot = g.spill_base + (unsigned)b * g.spill_cap + sp;              // (it holds every atom of the call)
            want = sp < g.spill_cap;
        }
        if (want) {
            int level = 0;
            if (g.reach_tau > 0.f) {                                           // tolerance-aware reach (fill_record's rule)
                float w_min = mk_inf();
#pragma unroll
                for (int j = 0; j < CHG; ++j) w_min = fminf(w_min, w[j]);
                const float frac = g.reach_tau / (w_min * g.R2);
                level = (int)fminf(fmaxf(floorf((1.f - frac) * (1.f / REACH_STEP) - 1e-3f), 0.f), 3.f);
            }
            rec_pos[slot] = make_float4(rel[0], rel[1], rel[2], mk_int_as_float(pc[0] | (pc[1] << 10) | (pc[2] << 20) | (level << 30)));
            rec_cls[slot] = ids;
            rec_w[slot] = make_float4(w[0], w[1], w[2], w[3]);                // (measured: these two stores cost the call nothing)
            rec_w[(size_t)g.M + slot] = make_float4(w[4], w[5], w[6], w[7]);
        }
    }
    mk_block_sync();
    if (threadIdx.x < CLS_BLOCK_SET)
        block_sets[(size_t)blockIdx.x * CLS_BLOCK_SET + threadIdx.x] = s_full ? CLS_TOO_MANY : s_set[threadIdx.x];
}

// One cell-sorted record: the parked position + the atom's per-channel class ids (or w values on the general path).
// `tab` = the class table in registers (wave-uniform), so that a lookup is NCLS register compares.
// The atom's channels come from the compact description the binning parked (w0 bits + nibble mask): one table look-up
// and one multiply give the 8 class ids; only atoms with several distinct sigmas go back to their sigma row.
template <typename SigT, bool TOPO = false>
MK_DEV void fill_record(const GridDesc& g, size_t t, unsigned slot, const SigT* __restrict__ sigmas,
                        const float4* __restrict__ tmp_pos, const uint2* __restrict__ tmp_cls,
                        float4* __restrict__ rec_pos, float4* __restrict__ rec_w,
                        unsigned* __restrict__ rec_cls, const unsigned (&tab)[NCLS], bool general)
{
    float4 pos = mk_tmp_load(&tmp_pos[t]);
    const size_t a = g.img_cap == 1 ? t : t / (size_t)g.img_cap;
    if constexpr (TOPO) {                                            // the binning parked the topology's class ids: a pure permutation
        rec_pos[slot] = pos;
        const unsigned* __restrict__ ids = reinterpret_cast<const unsigned*>(tmp_cls) + a * (size_t)g.G;
        for (int gq = 0; gq < g.G; ++gq) rec_cls[(size_t)gq * g.M + slot] = mk_tmp_load(&ids[gq]);
        return;
    }
    if (g.reach_tau > 0.f) {                                         // tolerance-aware reach (wave-uniform; off by default)
        // the atom's widest sigma -> the largest reach level whose radius still covers tau / w_min (an entry is worth
        // less than the tolerance beyond it); the level rides in the two spare bits of the packed cell word
        float w_min = mk_inf();
        for (int gq = 0; gq < g.G; ++gq) {
            const uint2 cw = tmp_cls[a * (size_t)g.G + gq];
            if (cw.y != ATOM_MULTI_SIGMA) {
                if (cw.x != CLS_EMPTY) w_min = fminf(w_min, mk_uint_as_float(cw.x));
            } else {
                float w[CHG];
                atom_channel_w(sigmas + a * (size_t)g.C, gq * CHG, g.C, g.w_scale, w);
#pragma unroll
                for (int c = 0; c < CHG; ++c) w_min = fminf(w_min, w[c]);
            }
        }
        const float frac = g.reach_tau / (w_min * g.R2);             // (reach / cutoff)^2 this atom needs (0 for +inf)
        const float lv = floorf((1.f - frac) * (1.f / REACH_STEP) - 1e-3f);
        const int level = w_min < mk_inf() ? (int)fminf(fmaxf(lv, 0.f), 3.f) : 0;
        pos.w = mk_int_as_float(mk_float_as_int(pos.w) | (level << 30));
    }
    rec_pos[slot] = pos;
    auto class_of = [&](unsigned bits) {
        unsigned id = 0;
#pragma unroll
        for (int i = 0; i < NCLS; ++i) id = (tab[i] == bits) ? (unsigned)(i + 1) : id;
        return id;
    };
    for (int gq = 0; gq < g.G; ++gq) {
        const uint2 cw = mk_tmp_load(&tmp_cls[a * (size_t)g.G + gq]);
        if (cw.y != ATOM_MULTI_SIGMA) {
            if (general) {
                float w[CHG];
#pragma unroll
                for (int c = 0; c < CHG; ++c) w[c] = ((cw.y >> (4 * c)) & 1u) ? mk_int_as_float((int)cw.x) : mk_inf();
                rec_w[(size_t)(gq * 2 + 0) * g.M + slot] = make_float4(w[0], w[1], w[2], w[3]);
                rec_w[(size_t)(gq * 2 + 1) * g.M + slot] = make_float4(w[4], w[5], w[6], w[7]);
            } else {
                rec_cls[(size_t)gq * g.M + slot] = class_of(cw.x) * cw.y;      // ids <= 15: no carry between nibbles
            }
            continue;
        }
        float w[CHG];
        atom_channel_w(sigmas + a * (size_t)g.C, gq * CHG, g.C, g.w_scale, w);
        if (general) {
            rec_w[(size_t)(gq * 2 + 0) * g.M + slot] = make_float4(w[0], w[1], w[2], w[3]);
            rec_w[(size_t)(gq * 2 + 1) * g.M + slot] = make_float4(w[4], w[5], w[6], w[7]);
        } else {
            unsigned ids = 0;
            for (int c = 0; c < CHG; ++c)
                if (w[c] < mk_inf()) ids |= class_of(mk_float_bits(w[c])) << (4 * c);
            rec_cls[(size_t)gq * g.M + slot] = ids;
        }
    }
}

template <typename SigT, bool SHARED = false, bool TOPO = false>
__attribute__((amdgpu_num_vgpr(24))) MK_KERNEL(256) void k_bin_fill(GridDesc g, const SigT* __restrict__ sigmas,
                               const unsigned* __restrict__ cell_start,
                               const float4* __restrict__ tmp_pos, const uint2* __restrict__ tmp_idx,
                               const uint2* __restrict__ tmp_cls,
                               float4* __restrict__ rec_pos, float4* __restrict__ rec_w,
                               unsigned* __restrict__ rec_cls, const unsigned* __restrict__ cls_table, unsigned nblk)
{
    if (direct_layout(g)) return;                        // nothing to permute: the direct pass wrote the records in place
    if (g.prepass_hurry) mk_wave_priority_high();
    // XCD-aware order: the dispatcher places block i on XCD i % 8; give each XCD a contiguous eighth of the temp slots =
    // whole items.  The records of an item are a few hundred KB: written by ONE XCD they meet in its L2 and leave as full
    // lines; spread over all eight (consecutive blocks of an item) every 128-byte line leaves in up to eight pieces
    // (WRITE_SIZE 324 MB for 187 MB of records).  Placement is a speed matter only: any block may take any slots.
    // (the grid is one workgroup per 256 slots, or -- SHARED: behind a direct pass, where this kernel is the fall-back -- a few
    //  thousand that share them, a multiple of eight so that a virtual block stays on its XCD; a separate instance: as a loop
    //  the kernel spills under its 48-register cap, tests/test_register_budgets.py)
    const unsigned per_xcd = nblk >> 3;
    for (unsigned vb = blockIdx.x; vb < (SHARED ? nblk : blockIdx.x + 1u); vb += (SHARED ? gridDim.x : 1u)) {
        const unsigned lb = vb < 8u * per_xcd ? (vb & 7u) * per_xcd + (vb >> 3) : vb;
        const size_t t = (size_t)lb * blockDim.x + threadIdx.x;
        if (t >= (size_t)g.M) continue;
        const uint2 ix = mk_tmp_load(&tmp_idx[t]);
        if (ix.x == TMP_UNUSED) continue;
        const bool general = !TOPO && (g.force_general || cls_table[CLS_OVERFLOW] != CLS_EMPTY);
        unsigned tab[NCLS];
#pragma unroll
        for (int i = 0; i < NCLS; ++i) tab[i] = TOPO ? 0u : cls_table[i];
        fill_record<SigT, TOPO>(g, t, cell_start[ix.x] + ix.y, sigmas, tmp_pos, tmp_cls, rec_pos, rec_w, rec_cls, tab, general);
    }
}

// ------------------------------------------------------------------------------------------------
// Topology (round 5): what the pre-pass derives from the SIGMAS alone -- an atom's compact channel words,
// the sigma classes, its class ids, whether any sigma is wide enough for the exact cut-off fix-up -- is the same for every
// frame of a trajectory (the sigmas depend on the topology only, voxeldescriptors.py:332-335).  Built once per molecule and
// voxel size by these two kernels (+ k_merge_classes between them); a call that brings the handle (GridDesc::topo_n) bins
// with k_bin_count<.., TOPO> / k_bin_fill<.., TOPO>: 4 bytes of ids per atom instead of its sigma row, no division, no class
// discovery, no table look-ups -- the records, and therefore the features, are the plain path's bit for bit (class ids are
// labels: a minimum over classes does not depend on how they are numbered).
// ------------------------------------------------------------------------------------------------
template <typename SigT>
MK_KERNEL(256) void k_topology_classes(const SigT* __restrict__ sigmas, long long n, int C, int G, double w_scale,
                                       uint2* __restrict__ cw_out /* [n, G] */, unsigned* __restrict__ block_sets)
{
    __shared__ unsigned s_set[CLS_BLOCK_SET];
    __shared__ unsigned s_full;
    if (threadIdx.x < CLS_BLOCK_SET) s_set[threadIdx.x] = CLS_EMPTY;
    if (threadIdx.x == 0) s_full = 0u;
    mk_block_sync();
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = a < n;
    for (int c0 = 0; c0 < C; c0 += CHG) {                            // (uniform: registration is wave-cooperative)
        unsigned wb[CHG];
        float w[CHG];
#pragma unroll
        for (int j = 0; j < CHG; ++j) w[j] = mk_inf();
        uint2 cw = make_uint2(CLS_EMPTY, 0u);
        if (act) {
            cw = atom_channel_w(sigmas + (size_t)a * C, c0, C, w_scale, w);
            cw_out[(size_t)a * G + (c0 / CHG)] = cw;
        }
#pragma unroll
        for (int j = 0; j < CHG; ++j) wb[j] = (w[j] < mk_inf()) ? mk_float_bits(w[j]) : CLS_EMPTY;
        wave_register_classes(cw.x, cw.y == ATOM_MULTI_SIGMA, wb, s_set, &s_full);
    }
    mk_block_sync();
    if (threadIdx.x < CLS_BLOCK_SET)
        block_sets[(size_t)blockIdx.x * CLS_BLOCK_SET + threadIdx.x] = s_full ? CLS_TOO_MANY : s_set[threadIdx.x];
}

template <typename SigT>
MK_KERNEL(256) void k_topology_ids(const SigT* __restrict__ sigmas, const uint2* __restrict__ cw_in, const unsigned* __restrict__ cls_table,
                                   long long n, int C, int G, double w_scale, float w_exact_max,
                                   unsigned* __restrict__ ids_out /* [n, G] */, int* __restrict__ flags /* [0] |= 1: some sigma is wide; [1]: how many atoms */,
                                   unsigned* __restrict__ wide_list /* [n]: the atoms with a wide sigma, in the order they arrive (the host sorts) */)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    unsigned tab[NCLS];
#pragma unroll
    for (int i = 0; i < NCLS; ++i) tab[i] = cls_table[i];
    auto class_of = [&](unsigned bits) {
        unsigned id = 0;
#pragma unroll
        for (int i = 0; i < NCLS; ++i) id = (tab[i] == bits) ? (unsigned)(i + 1) : id;
        return id;
    };
    bool wide = false;
    for (int gq = 0; gq < G; ++gq) {
        const uint2 cw = cw_in[(size_t)a * G + gq];
        unsigned ids = 0u;
        if (cw.y != ATOM_MULTI_SIGMA) {
            ids = class_of(cw.x) * cw.y;                             // ids <= 15: no carry between nibbles (fill_record's rule)
            wide |= cw.x != CLS_EMPTY && mk_uint_as_float(cw.x) < w_exact_max;
        } else {
            float w[CHG];
            atom_channel_w(sigmas + (size_t)a * C, gq * CHG, C, w_scale, w);
            for (int c = 0; c < CHG; ++c)
                if (w[c] < mk_inf()) { ids |= class_of(mk_float_bits(w[c])) << (4 * c); wide |= w[c] < w_exact_max; }
        }
        ids_out[(size_t)a * G + gq] = ids;
    }
    if (wide) {
        mk_atomic_or(flags, 1);
        wide_list[mk_atomic_add(reinterpret_cast<unsigned*>(flags) + 1, 1u)] = (unsigned)a;
    }
}

// ------------------------------------------------------------------------------------------------
// The whole pre-pass in ONE launch for batches of small items (ligand poses, pockets: up to a few thousand
// atoms per item): one block per item counts its atoms into LDS cell counters (the rank inside the cell comes
// from the LDS atomic -- no global atomics), collects the item's sigma classes in an LDS set, scans the counters
// in place and permutes the temp records into the item's slice [atom_offsets[b] * img_cap, ...) of the record
// arrays.  Replaces memset + k_bin_count + 2 reductions + k_scan_finish + k_bin_fill: for such batches those
// are launch-latency-bound (~5 us per dependent launch).  Class tables are per item here (cls_table[B][16]): an
// item with more than 15 distinct sigmas takes the general tile path on its own.
// ------------------------------------------------------------------------------------------------
constexpr int ITEM_HIST = 8192;                    // most LDS cell counters per item (cells + end marker): 32 KiB;
                                                   // instantiated for 512 / 2048 / 8192 so that small grids stay small

// exclusive scan helper for up to 1024 threads (16 waves)
MK_DEV unsigned block_scan_exclusive16(unsigned v, unsigned* total, unsigned* lds /* >= 16 */)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned t = mk_shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == WAVE - 1) lds[wv] = incl;
    mk_block_sync();
    unsigned woff = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) {
        const unsigned sv = lds[i];
        if (i < wv) woff += sv;
        tot += sv;
    }
    mk_block_sync();
    *total = tot;
    return woff + incl - v;
}

template <typename SigT, int HIST>
MK_KERNEL(1024) void k_prepass_items(GridDesc g, const float* __restrict__ coords,
                                     const long long* __restrict__ atom_offsets,
                                     const void* __restrict__ sigmas_v, const double* __restrict__ origins,
                                     const float* __restrict__ box, const double* __restrict__ affine,
                                     unsigned* __restrict__ cell_start,
                                     float4* __restrict__ tmp_pos, uint2* __restrict__ tmp_idx, uint2* __restrict__ tmp_cls,
                                     float4* __restrict__ rec_pos, float4* __restrict__ rec_w,
                                     unsigned* __restrict__ rec_cls, unsigned* __restrict__ cls_table,
                                     unsigned* __restrict__ dense_words, int* __restrict__ err_flag)
{
    mk_wave_priority_high();
    const SigT* __restrict__ sigmas = static_cast<const SigT*>(sigmas_v);
    __shared__ unsigned s_hist[HIST];
    __shared__ unsigned s_set[CLS_BLOCK_SET];
    __shared__ unsigned s_tab[CLS_TABLE_WORDS];
    __shared__ unsigned s_scan[16];
    __shared__ unsigned s_full;
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int ncell = g.ncell;                                           // <= HIST - 1 (host checks)
    if (b == 0 && tid < DENSE_WORDS) dense_words[tid] = 0u;              // dense-list length + tier statistics
    for (int i = tid; i <= ncell; i += nth) s_hist[i] = 0u;
    if (tid < CLS_BLOCK_SET) s_set[tid] = CLS_EMPTY;
    if (tid == 0) s_full = 0u;
    mk_block_sync();
    const long long a0 = atom_offsets[b], a1 = atom_offsets[b + 1];
    const size_t cell_base = (size_t)b * g.cstride;
    const bool classes = !g.force_general;

    // ---- phase 1: temp records with the rank inside the cell, LDS counts, the item's sigma classes ----
    for (long long base = a0; base < a1; base += nth) {                  // block-uniform trip count
        const long long a = base + tid;
        bin_atom<SigT, -1>(g, a, a < a1, b, b, coords, atom_offsets, sigmas, origins, box, affine, tmp_pos, tmp_idx, tmp_cls, err_flag,
                       classes, s_set, &s_full, [&](size_t cell) { return mk_lds_add(&s_hist[cell - cell_base], 1u); },
                       [&](bool want, size_t cell) { return want ? mk_lds_add(&s_hist[cell - cell_base], 1u) : 0u; });
    }
    mk_block_sync();

    // ---- phase 2: the item's class table; counts -> starts (in place), published as this item's cell_start ----
    if (tid == 0) {
        unsigned n = 0;
        bool over = !classes || s_full != 0u;
        for (int i = 0; i < CLS_BLOCK_SET; ++i) {
            const unsigned v = s_set[i];
            if (v == CLS_EMPTY) continue;
            if (n < (unsigned)NCLS) s_tab[n] = v;
            ++n;
        }
        if (n > (unsigned)NCLS) over = true;
        for (unsigned i = n; i < (unsigned)NCLS; ++i) s_tab[i] = CLS_EMPTY;
        s_tab[CLS_OVERFLOW] = over ? 0u : CLS_EMPTY;
    }
    const unsigned rbase = (unsigned)((unsigned long long)a0 * (unsigned long long)g.img_cap);
    unsigned carry = 0;
    for (int base = 0; base <= ncell; base += nth) {                     // block-uniform
        const int i = base + tid;
        const unsigned v = (i <= ncell) ? s_hist[i] : 0u;
        unsigned tot;
        const unsigned ex = block_scan_exclusive16(v, &tot, s_scan);
        if (i <= ncell) {
            s_hist[i] = carry + ex;
            cell_start[cell_base + i] = rbase + carry + ex;              // [ncell] = the item's end marker
        }
        carry += tot;
    }
    mk_block_sync();
    if (tid < CLS_TABLE_WORDS) cls_table[(size_t)b * CLS_TABLE_WORDS + tid] = s_tab[tid];

    // ---- phase 3: permute the temp records into cell order ----
    const bool general = g.force_general || s_tab[CLS_OVERFLOW] != CLS_EMPTY;
    unsigned tab[NCLS];
#pragma unroll
    for (int i = 0; i < NCLS; ++i) tab[i] = s_tab[i];
    const size_t t0 = (size_t)a0 * (size_t)g.img_cap, t1 = (size_t)a1 * (size_t)g.img_cap;
    for (size_t t = t0 + tid; t < t1; t += nth) {
        const uint2 ix = tmp_idx[t];
        if (ix.x == TMP_UNUSED) continue;
        fill_record<SigT>(g, t, rbase + s_hist[ix.x - (unsigned)cell_base] + ix.y, sigmas, tmp_pos, tmp_cls, rec_pos, rec_w, rec_cls, tab, general);
    }
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan of the cell counts (n values -> n+1 starts).  Three small kernels:
// per-4096-chunk sums, single-block scan of the sums, per-chunk scan + offset.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER_THREAD;   // 4096

// inclusive scan across the 64 lanes of a wave
MK_DEV unsigned wave_scan_inclusive(unsigned v)
{
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned t = mk_shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; *total gets the block sum.
MK_DEV unsigned block_scan_exclusive(unsigned v, unsigned* total, unsigned* lds /* >= 8 */)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const unsigned incl = wave_scan_inclusive(v);
    if (lane == WAVE - 1) lds[wv] = incl;
    mk_block_sync();
    unsigned woff = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) {
        const unsigned s = lds[i];
        if (i < wv) woff += s;
        tot += s;
    }
    mk_block_sync();                     // lds may be reused by the caller's next call
    *total = tot;
    return woff + incl - v;
}

MK_DEV void scan_chunk_sums_block(const unsigned* __restrict__ in, size_t n, unsigned* __restrict__ chunk_sums, unsigned block)
{
    __shared__ unsigned lds[8];
    const size_t base = (size_t)block * SCAN_CHUNK;
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        const size_t i = base + (size_t)j * SCAN_THREADS + threadIdx.x;
        s += (i < n) ? in[i] : 0u;
    }
    unsigned tot;
    (void)block_scan_exclusive(s, &tot, lds);
    if (threadIdx.x == 0) chunk_sums[block] = tot;
}

MK_DEV void scan_sums_inplace_block(unsigned* __restrict__ chunk_sums, unsigned nchunks)
{
    __shared__ unsigned lds[8];
    unsigned carry = 0;
    for (unsigned base = 0; base < nchunks; base += SCAN_THREADS) {
        const unsigned i = base + threadIdx.x;
        const unsigned v = (i < nchunks) ? chunk_sums[i] : 0u;
        unsigned tot;
        const unsigned ex = block_scan_exclusive(v, &tot, lds);
        if (i < nchunks) chunk_sums[i] = carry + ex;
        carry += tot;
    }
}

MK_KERNEL(SCAN_THREADS) void k_scan_chunk_sums(const unsigned* __restrict__ in, size_t n,
                                               unsigned* __restrict__ chunk_sums)
{
    mk_wave_priority_high();
    scan_chunk_sums_block(in, n, chunk_sums, blockIdx.x);
}

MK_KERNEL(SCAN_THREADS) void k_scan_sums_inplace(unsigned* __restrict__ chunk_sums, unsigned nchunks)
{
    mk_wave_priority_high();
    scan_sums_inplace_block(chunk_sums, nchunks);
}

// The lattice pre-pass runs the two independent reductions that follow the binning -- sigma classes (two
// merge levels) and the cell-count scan (chunk sums, scan of the sums) -- side by side in two launches
// instead of five: each launch boundary is ~5 us of dependent latency on a chain that is latency-bound.
//   stage 1: blocks [0, nl1) merge groups of per-block sigma sets, blocks [nl1, nl1+nchunks) sum count chunks
//   stage 2: block 0 merges the level-1 sets into the class table, the other block scans the chunk sums
static_assert(SCAN_THREADS == 256, "the fused pre-pass kernels run both jobs with 256 threads");
MK_KERNEL(256) void k_prepass_reduce1(const unsigned* __restrict__ block_sets, unsigned nblk, unsigned rows_per_block,
                                      unsigned nl1, unsigned* __restrict__ l1sets,
                                      const unsigned* __restrict__ counts, size_t n, unsigned* __restrict__ chunk_sums,
                                      const unsigned* __restrict__ direct_failed /* nullptr, or the call's DIRECT_FAILED word */)
{
    if (direct_failed != nullptr && *direct_failed == 0u) return;     // the direct pass binned everything: the class table stays
    mk_wave_priority_high();
    if (blockIdx.x < nl1)                                              // block-uniform
        merge_classes_block(block_sets, nblk, (unsigned)CLS_BLOCK_SET, rows_per_block, l1sets, nullptr, blockIdx.x);
    else
        scan_chunk_sums_block(counts, n, chunk_sums, blockIdx.x - nl1);
}

MK_KERNEL(256) void k_prepass_reduce2(const unsigned* __restrict__ l1sets, unsigned nl1, unsigned* __restrict__ cls_table,
                                      unsigned* __restrict__ chunk_sums, unsigned nchunks,
                                      const unsigned* __restrict__ direct_failed)
{
    if (direct_failed != nullptr && *direct_failed == 0u) return;
    mk_wave_priority_high();
    if (nl1 != 0u && blockIdx.x == 0)                                  // block-uniform
        merge_classes_block(l1sets, nl1, (unsigned)MERGE_SET, nl1, nullptr, cls_table, 0u);
    else
        scan_sums_inplace_block(chunk_sums, nchunks);
}

// Both reductions after the binning -- sigma classes -> class table, cell counts -> cell starts -- by ONE workgroup, for
// calls small enough that three dependent launches (~5 us each) cost more than the work: one grid per call.
constexpr int SMALL_PREPASS_THREADS = 1024;
constexpr unsigned SMALL_PREPASS_MAX_CELLS = 1u << 13;     // counts one block scans in a few microseconds (8 rounds)
constexpr unsigned SMALL_PREPASS_MAX_BLOCKS = 4096;        // per-block sigma sets one block merges likewise

MK_KERNEL(SMALL_PREPASS_THREADS) void k_prepass_small(const unsigned* __restrict__ block_sets, unsigned nblk,
                                                      unsigned* __restrict__ cls_table, unsigned* __restrict__ counts,
                                                      unsigned n, unsigned* __restrict__ starts /* n+1 */)
{
    mk_wave_priority_high();
    __shared__ unsigned s_scan[16];
    if (nblk != 0u) merge_classes_block(block_sets, nblk, (unsigned)CLS_BLOCK_SET, nblk, nullptr, cls_table, 0u);
    unsigned carry = 0;
    for (unsigned base = 0; base <= n; base += SMALL_PREPASS_THREADS) {    // block-uniform
        const unsigned i = base + threadIdx.x;
        const unsigned v = i < n ? counts[i] : 0u;
        if (i < n) counts[i] = 0u;                                         // the counters leave every call as they entered it: zero
        unsigned tot;
        const unsigned ex = block_scan_exclusive16(v, &tot, s_scan);
        if (i <= n) starts[i] = carry + ex;                                // starts[n] = grand total
        carry += tot;
    }
}

MK_KERNEL(SCAN_THREADS) void k_scan_finish(unsigned* __restrict__ in /* zeroed once read: see run_lattice */, size_t n,
                                           const unsigned* __restrict__ chunk_offsets,
                                           unsigned* __restrict__ out /* n+1 */, const unsigned* __restrict__ direct_failed)
{
    if (direct_failed != nullptr && *direct_failed == 0u) return;
    mk_wave_priority_high();
    __shared__ unsigned lds[8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    unsigned v[SCAN_PER_THREAD];
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        const size_t i = base + j;
        v[j] = (i < n) ? in[i] : 0u;
        if (i < n) in[i] = 0u;
        s += v[j];
    }
    unsigned tot;
    unsigned run = chunk_offsets[blockIdx.x] + block_scan_exclusive(s, &tot, lds);
#pragma unroll
    for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        const size_t i = base + j;
        if (i <= n) out[i] = run;        // out[n] = grand total
        run += v[j];
    }
}

// ------------------------------------------------------------------------------------------------
// Occupancy value from the reduced q = min d^2/sigma^2 :  1 - exp(-q^-6)
// (occupancy_utils.pyx:57-60 with x^12 = (sigma^2/d^2)^6).  q=+inf -> 0, q=0 -> 1.
// For small x = q^-6 the float32 form 1-exp(-x) cancels (absolute error 6e-8 whatever x is), so the
// tail uses the alternating series of -expm1(-x): relative accuracy <= 4e-6 everywhere, which is
// what lets the reference's own np.allclose(rtol=1e-5) style checks pass on float32 results.
// ------------------------------------------------------------------------------------------------
MK_DEV float occupancy_from_q(float q)
{
    const float u = mk_rcp(q);                       // v_rcp_f32, 1 ulp: u^6 carries <= 1e-6 relative error
    const float u3 = u * u * u;
    const float x = u3 * u3;
    const float big = 1.0f - mk_exp2(-1.4426950408889634f * x);
    // x - x^2/2 + x^3/6 for x < 1/64 (truncation x^3/24 < 1.6e-7 relative); above that 1-exp(-x) is
    // relatively accurate to 6e-8/x <= 4e-6
    const float small = x * mk_fma(-x, mk_fma(-x, 0.16666667f, 0.5f), 1.0f);   // (the fmas hipcc contracts to anyway)
    return x < 0.015625f ? small : big;
}

// ------------------------------------------------------------------------------------------------
// THE hot kernel.  One 64-lane wave per K x 8 x 8 voxel tile (lane = (y,z), z fastest as in the
// output layout, K x-planes per lane in registers); blockIdx.y = channel group.
// ------------------------------------------------------------------------------------------------
template <int N> struct IntC { static constexpr int value = N; };   // compile-time int for generic lambdas

struct TileGeom {                     // wave-uniform description of the tile being voxelized
    int b, x0, y0, z0;
    int cx_lo, cx_hi, cy_lo, cy_hi, cz_lo, cz_hi;
    float offx, offy, offz, fcs;
};

// Visit every candidate record of the tile in chunks of 64 (one record per lane).
// The (cell-x, cell-y) columns around the tile each contribute one contiguous run of records (their
// z-cells are adjacent in memory); the runs' bounds are fetched by one lane each, the chunks of all
// runs are numbered 0..T-1 and processed in batches whose loads are all issued up front (the loads
// are L2 / fabric round trips; chunk-at-a-time every chunk would pay ~1 us of exposed latency).
// f(survives, record index, tile-relative x,y,z, class ids, packed cell word) is called by ALL lanes for every chunk.
struct CandChunk {                    // one chunk of 64 candidate records (one per lane), loads in flight
    float4 P;
    unsigned r, ids;
    unsigned code;                    // (run << SURV_OFF_BITS) | offset inside the run: what the survivor list keeps of a record
    bool valid;
};
// Survivor list of the tile kernel: the records that pass the cull, as 16-bit (run, offset) codes in the ~1 KB of LDS the
// tile has to spare.  The cull is then evaluated ONCE, and the histogram and placement passes walk ~290 survivors
// (5 chunks) instead of ~1000 candidates (18 chunks) each; their per-record channel loops run over dense lanes.
constexpr int SURV_CAP = 480;
constexpr int SURV_OFF_BITS = 10;

// The candidate runs of a tile: lane j < ncols holds the record range of cell column j (one load round trip) -- the
// z-cells of a column are adjacent in memory -- trimmed to the cells that can hold an atom within the cutoff of the tile
// box (columns out of reach altogether are dropped).  The candidates are numbered 0..N-1 across the runs (`pre` = a
// run's first number) and visited in chunks of 64 consecutive numbers, so that a chunk is full whatever the runs'
// lengths are.  Found once per tile and shared by every traversal.
struct CandRuns {
    unsigned r0, len, pre, N, T;
    bool codable;                     // every run is short enough for the 16-bit survivor codes
    MK_PHASE_FIELDS
};

template <int K>
MK_DEV CandRuns find_candidate_runs(const GridDesc& g, const TileGeom& tg, const unsigned* __restrict__ cell_start)
{
    const int lane = threadIdx.x & (WAVE - 1);
    CandRuns cr;
    cr.r0 = 0; cr.len = 0;
    if (direct_layout(g)) {
        // direct layout: a cell's records sit at cell * cap (its count in the direct counters), so every CELL around the
        // tile is a run of its own (<= 63 of them, the host checked), and the ITEM's spill area -- records of cells that
        // overflowed their capacity, normally none -- is one more run that every tile of the item looks at: candidates
        // are culled by their distance to the tile, so seeing a record too many is harmless (a record of ANOTHER item
        // would not be: the packed cell word has no room for the item, hence one spill area per item)
        const unsigned* __restrict__ cnt = g.direct_words + DIRECT_HEAD;
        const int nxc = tg.cx_hi - tg.cx_lo + 1, nyc = tg.cy_hi - tg.cy_lo + 1, nzc = tg.cz_hi - tg.cz_lo + 1;
        const int ncells = nxc * nyc * nzc;
        // lane -> cell: at most 4 cells per axis (the rule: an 8-voxel tile edge + 2 x the cutoff over cells of at least the
        // cutoff) are dealt out by the lane's bit fields -- three divisions by run-time numbers cost ~70 instructions per tile
        const bool boxed = nxc <= 4 && nyc <= 4 && nzc <= 4 && ncells < WAVE;            // wave-uniform (lane 63 stays free for the spill run)
        int ix, iy, iz;
        bool holds_cell;
        if (boxed) { ix = lane >> 4; iy = (lane >> 2) & 3; iz = lane & 3; holds_cell = ix < nxc && iy < nyc && iz < nzc; }
        else { ix = lane / (nzc * nyc); iy = (lane / nzc) % nyc; iz = lane % nzc; holds_cell = lane < ncells; }
        if (holds_cell) {
            const int pcz = tg.cz_lo + iz, pcy = tg.cy_lo + iy, pcx = tg.cx_lo + ix;
            const float fcs = (float)g.cs;
            const float cx0 = (float)((pcx - g.h) * g.cs) - 0.5f, cy0 = (float)((pcy - g.h) * g.cs) - 0.5f, cz0 = (float)((pcz - g.h) * g.cs) - 0.5f;
            const float gx = fmaxf(fmaxf((float)tg.x0 - (cx0 + fcs), cx0 - (float)(tg.x0 + K - 1)), 0.f);
            const float gy = fmaxf(fmaxf((float)tg.y0 - (cy0 + fcs), cy0 - (float)(tg.y0 + 7)), 0.f);
            const float gz = fmaxf(fmaxf((float)tg.z0 - (cz0 + fcs), cz0 - (float)(tg.z0 + 7)), 0.f);
            if (gx * gx + gy * gy + gz * gz < g.R2cull) {
                const size_t cell = (size_t)tg.b * g.cstride + ((size_t)pcx * g.ncy + pcy) * g.ncz + pcz;
                const unsigned n = cnt[cell << g.cnt_shift];
                cr.r0 = (unsigned)cell * (unsigned)g.cell_cap;
                cr.len = n < (unsigned)g.cell_cap ? n : (unsigned)g.cell_cap;
            }
        } else if (lane == WAVE - 1) {
            const unsigned n = cnt[((size_t)tg.b * g.cstride + g.ncell) << g.cnt_shift];     // the item's spare counter: its spilled records
            cr.r0 = g.spill_base + (unsigned)tg.b * g.spill_cap;
            cr.len = n < g.spill_cap ? n : g.spill_cap;
        }
    } else {
    const int nyc = tg.cy_hi - tg.cy_lo + 1;
    const int ncols = (tg.cx_hi - tg.cx_lo + 1) * nyc;           // <= 49 (cell edge > half the cutoff radius, plan_lattice)
    if (lane < ncols) {
        const int pcx = tg.cx_lo + lane / nyc, pcy = tg.cy_lo + lane % nyc;
        // a cell holds positions in [c0 - 0.5, c0 + cs - 0.5), c0 = (pc - h) * cs (bin_atom: locate); its gap to the tile's voxel box
        const float fcs = (float)g.cs;
        const float cx0 = (float)((pcx - g.h) * g.cs) - 0.5f, cy0 = (float)((pcy - g.h) * g.cs) - 0.5f;
        const float gx = fmaxf(fmaxf((float)tg.x0 - (cx0 + fcs), cx0 - (float)(tg.x0 + K - 1)), 0.f);
        const float gy = fmaxf(fmaxf((float)tg.y0 - (cy0 + fcs), cy0 - (float)(tg.y0 + 7)), 0.f);
        const float rest = g.R2cull - (gx * gx + gy * gy);
        if (rest > 0.f) {
            const float zg = sqrtf(rest) * 1.0001f + 1e-3f;      // reach along z left for this column (generous)
            const float inv = 1.0f / fcs;
            int ca = (int)floorf(((float)tg.z0 - zg + 0.5f) * inv) + g.h, cb = (int)floorf(((float)(tg.z0 + 7) + zg + 0.5f) * inv) + g.h;
            ca = ca < tg.cz_lo ? tg.cz_lo : ca;
            cb = cb > tg.cz_hi ? tg.cz_hi : cb;
            if (ca <= cb) {
                const size_t cbase = (size_t)tg.b * g.cstride + ((size_t)pcx * g.ncy + pcy) * g.ncz;
                cr.r0 = cell_start[cbase + ca];
                cr.len = cell_start[cbase + cb + 1] - cr.r0;      // z-run of cells is contiguous
            }
        }
    }
    }
    const unsigned incl = wave_scan_inclusive(cr.len);
    cr.pre = incl - cr.len;
    cr.N = mk_readlane(incl, WAVE - 1);
    cr.T = (cr.N + (WAVE - 1)) >> 6;
    cr.codable = mk_ballot(cr.len > (1u << SURV_OFF_BITS)) == 0ull;      // (run index < 64: 6 bits)
    return cr;
}

// issue(t, ch) starts the loads of candidate chunk t; consume(ch, f) makes its records tile-relative, culls them
// against the tile box and calls f(survives, record index, x, y, z, class ids) -- by ALL lanes
struct CandLoader {
    const GridDesc& g;
    const TileGeom& tg;
    const CandRuns& cr;
    const float4* __restrict__ rec_pos;
    const unsigned* __restrict__ rec_cls;
};

template <bool LOAD_CLS>
MK_DEV void cand_issue(const CandLoader& L, unsigned t, CandChunk& ch)
{
    const CandRuns& cr = L.cr;
    const float4* __restrict__ rec_pos = L.rec_pos;
    const unsigned* __restrict__ rec_cls = L.rec_cls;
    {
        const int lane = threadIdx.x & (WAVE - 1);
        ch.r = 0u; ch.valid = false; ch.ids = 0u; ch.code = 0u;
        ch.P = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < cr.T) {                                              // wave-uniform
            // candidate numbers [64 t, 64 t + 64): the runs that overlap them (a handful), each claimed by its lanes
            const unsigned n0 = t << 6, n = n0 + (unsigned)lane;
            unsigned long long over = mk_ballot(cr.len != 0u && cr.pre < n0 + (unsigned)WAVE && cr.pre + cr.len > n0);
            while (over) {                                           // wave-uniform
                const int j = mk_ctz64(over);
                over &= over - 1ull;
                const unsigned pj = mk_readlane(cr.pre, j), lj = mk_readlane(cr.len, j), rj = mk_readlane(cr.r0, j);
                const unsigned off = n - pj;
                const bool mine = off < lj;                          // unsigned: pj <= n < pj + lj  (selects, not a branch:
                ch.r = mine ? rj + off : ch.r;                       //  the masked-region form costs a dozen scalar ops per run)
                ch.code = mine ? (((unsigned)j << SURV_OFF_BITS) | off) : ch.code;
                ch.valid = ch.valid || mine;
            }
            if (ch.valid) {
                ch.P = rec_pos[ch.r];
                if (LOAD_CLS) ch.ids = rec_cls[ch.r];
            }
        }
    }
}

template <int K, class F, bool LOAD_CODE = false>
MK_DEV void cand_consume(const CandLoader& L, const CandChunk& ch, F&& f)
{
    const GridDesc& g = L.g;
    const TileGeom& tg = L.tg;
    {
        constexpr float HX = 0.5f * (float)(K - 1);
        const int pk = mk_float_as_int(ch.P.w);
        // (cell centre - tile centre) is an exact small half-integer; ONE rounding per axis
        const float ex = ch.P.x + ((float)(pk & 1023) * tg.fcs + tg.offx);
        const float ey = ch.P.y + ((float)((pk >> 10) & 1023) * tg.fcs + tg.offy);
        const float ez = ch.P.z + ((float)((pk >> 20) & 1023) * tg.fcs + tg.offz);
        const float gx = fmaxf(fabsf(ex) - HX, 0.f);
        const float gy = fmaxf(fabsf(ey) - 3.5f, 0.f);
        const float gz = fmaxf(fabsf(ez) - 3.5f, 0.f);
        const bool surv = ch.valid && (gx * gx + gy * gy + gz * gz < reach_r2(g, g.R2cull, pk));
        f(surv, LOAD_CODE ? ch.code : ch.r, ex, ey, ez, ch.ids, pk);
    }
}

template <int K, bool LOAD_CLS, int BATCH, class F, bool LOAD_CODE = false>
MK_DEV void for_each_candidate(const GridDesc& g, const TileGeom& tg, const CandRuns& cr,
                               const float4* __restrict__ rec_pos, const unsigned* __restrict__ rec_cls, F&& f,
                               unsigned first_batch = 0u, unsigned batch_stride = 1u)
{
    const CandLoader ld{g, tg, cr, rec_pos, rec_cls};
    const unsigned T = cr.T;
    // BATCH chunks' loads are issued back to back, then the BATCH chunks are processed: the wave pays
    // the L2 / fabric round trip once per batch instead of once per chunk
    // (a team of waves shares one tile: wave w takes the CHUNKS w, w + team, ... -- first_batch / batch_stride -- BATCH of
    //  them in flight together; dealing whole batches left the 3PTB pocket's waves with 0 to 4 chunks each: 28.3 -> 25.9 us
    //  per call.  Six or eight in flight instead of four change nothing for a 64^3 grid's 24-chunk tiles)
    CandChunk ch[BATCH];
    for (unsigned t = first_batch; t < T; t += BATCH * batch_stride) {
        MK_PHASE_NOW(ta_);
#pragma unroll
        for (int k = 0; k < BATCH; ++k) cand_issue<LOAD_CLS>(ld, t + (unsigned)k * batch_stride, ch[k]);
        MK_PHASE_DRAIN();
        MK_PHASE_NOW(tb_);
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (t + (unsigned)k * batch_stride < T) cand_consume<K, F&, LOAD_CODE>(ld, ch[k], f);      // wave-uniform
        MK_PHASE_NOW(tc_);
        MK_PHASE_CAND(cr, ta_, tb_, tc_);                             // (diagnostics builds: flushed at the end of the tile)
    }
}

// visit the present channels of a record's 8 x 4-bit class ids: f(channel, class id 1..15)
template <class F>
MK_DEV void for_each_present_channel(unsigned ids, F&& f)
{
    while (ids) {                                                    // per-lane trip count (<= 8)
        const int sh = (__builtin_ctz(ids)) & ~3;
        f(sh >> 2, (ids >> sh) & 0xfu);
        ids &= ~(0xfu << sh);
    }
}

// DENSE = false: the tile kernel proper.  A class-sorted tile whose entries do not fit the LDS arrays
// is only appended to dense_list (dense_count = its length) and left to the DENSE = true instance,
// which runs afterwards over that list: keeping the rare multi-round code out of this kernel keeps
// its register footprint at 4 waves/SIMD (one kernel holding both needed ~2x the VGPRs).
// x of plane k relative to the tile centre is c_k = k - (K-1)/2.  d^2 of (voxel in plane k, entry) is
//   exact form:  (c_k - ex)^2 + dy^2 + dz^2                                  two instructions per (voxel, entry)
//   fast form :  g_k + c_k^2,  g_k = fma(-2 c_k, ex, D0),  D0 = ex^2 + dy^2 + dz^2   ONE (D0 is shared by the K planes)
// The fast form expands only the x term and only about the tile centre, but D0 is rounded at magnitude ~c_k^2 while the
// pairs that matter have d^2 ~ rho^2 (rho = sigma / voxelsize, where the occupancy is steepest: slope 2.2 in relative
// d^2): value error <= 2.2 * 2^-24 * (c_max^2 / rho^2 + 1).  It is therefore used per sigma CLASS only while
// w = 1/rho^2 <= FAST_W_MAX = 14 / c_max^2 (error budget 2e-6; K = 8: sigma >= 0.94 voxels -- every vdW radius on a
// 1 A grid), other classes take the exact form.  Every path (sorted, dense, general) makes the same choice from the
// same w, so their results stay identical bit for bit.
template <int K> MK_DEV constexpr float plane_x(int k) { return (float)k - 0.5f * (float)(K - 1); }
template <int K> MK_DEV constexpr float plane_slope(int k) { return -2.f * plane_x<K>(k); }
template <int K> MK_DEV constexpr float fast_w_max() { return 14.f / (plane_x<K>(K - 1) * plane_x<K>(K - 1)); }
template <int K> MK_DEV float plane_d2(int k, float gk)
{
    // rounding may leave -1e-7 for a pair at distance 0: callers scale |d2| (a free source modifier), NaN stays NaN
    return gk + plane_x<K>(k) * plane_x<K>(k);
}
// d^2 of one entry against the K planes, per-pair paths (general / dense chunks): the class rule above per entry
template <int K> MK_DEV void entry_d2(float ex, float dyz2, float w, float (&d2)[K])
{
    if (mk_uint_as_float(mk_uniform(mk_float_bits(w))) <= fast_w_max<K>()) {   // the entry is broadcast: a scalar branch
        const float d0 = mk_fma(ex, ex, dyz2);
#pragma unroll
        for (int k = 0; k < K; ++k) d2[k] = plane_d2<K>(k, mk_fma(plane_slope<K>(k), ex, d0));
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float dx = plane_x<K>(k) - ex;
            d2[k] = mk_fma(dx, dx, dyz2);
        }
    }
}

// TEAM > 1 (launches too small to fill the chip: one grid per call, the reference's usage): TEAM waves share one tile --
// each takes every TEAM-th batch of candidate chunks in the two traversals (histogram and placement go through the
// same LDS counters), every TEAM-th PAIR of a sub-bucket's entries against all K planes in the pair loops (splitting the
// planes instead left each wave the whole per-pair set-up for one or two planes' worth of tests: 73 % of the 3PTB call's
// tile kernel), and, after one min-reduction of the waves' accumulators through LDS, K / TEAM planes of the epilogue.
// Minima are order-free and the class flush is monotone, so the bits are the one-wave kernel's.  The arithmetic per
// (voxel, entry) is the one-wave kernel's, bit for bit; what changes is the latency of a tile (~1/3).
template <int K, bool DENSE, int ECAP, int TEAM = 1>
MK_DEV void voxelize_tile(const GridDesc& g, const unsigned lt, const int gq, const unsigned* __restrict__ cell_start,
                          const float4* __restrict__ rec_pos,
                          const float4* __restrict__ rec_w, const unsigned* __restrict__ rec_cls,
                          const unsigned* __restrict__ cls_table, float* __restrict__ out,
                          unsigned* __restrict__ dense_count, unsigned* __restrict__ dense_list)
{
    static_assert(K == 4 || K == 8, "K");
    static_assert(TEAM == 1 || (!DENSE && (K % TEAM == 0 || TEAM % K == 0) && (TEAM & (TEAM - 1)) == 0 && TEAM <= K * CHG / 2),
                  "a team (a power of two of waves) splits the planes -- and, beyond K waves, the channels -- of the epilogue evenly");
    constexpr int KL = K;                                 // planes in this wave's accumulators (all of them)
    constexpr int KE = TEAM <= K ? K / TEAM : 1;          // planes of the epilogue this wave owns: [kb, kb + KE)
    constexpr int CSPLIT = TEAM <= K ? 1 : TEAM / K;      // ... and, for teams of more than K waves, which channels of them:
    constexpr int CE = CHG / CSPLIT;                      //     [cb, cb + CE)
    MK_PHASE_BEGIN();
    // sorted path: entries as structure-of-arrays so that a PAIR of entries is three 8-byte
    // broadcast reads (ds_read_b64: 2 LDS cycles each).  LDS per tile is what bounds occupancy here
    // (measured: 1.25 -> 2 -> 2.75 -> 3.25 waves/SIMD = 0.69 -> 0.48 -> 0.43 -> 0.40 ms on cfg2), hence
    // the lean ~9.3 KiB budget.
    // One object, so that the pair loop addresses x, y and z from ONE register with constant offsets.
    constexpr int ESTRIDE = (ECAP + 2 + 3) & ~3;          // floats per coordinate array (16-byte multiple)
    __shared__ __attribute__((aligned(16))) float sxyz[3 * ESTRIDE];
    float* const sx = sxyz;
    float* const sy = sxyz + ESTRIDE;
    float* const sz = sxyz + 2 * ESTRIDE;
    float4* const ebuf = reinterpret_cast<float4*>(sxyz);   // general path: one chunk's entries of one channel (x,y,z,w)
    // per-tile histogram -> placement cursors -> (after placement) sub-bucket starts again; [NBUCKET3] = end
    __shared__ unsigned bucket[NBUCKET3 + 1];
    // a team keeps the placement cursors in an array of their own: the counts, the cursors and the final table of starts then
    // need two workgroup barriers between them, not four (a barrier of four unevenly loaded waves is ~0.3 us of a 28 us call)
    __shared__ unsigned bucket_cur[TEAM > 1 ? NBUCKET3 + 1 : 1];
    // the hot kernel culls once and keeps the survivors (see SURV_CAP).  The list BORROWS the z array of the entry buffer:
    // it is written by the cull pass, read by the histogram pass, and moved into registers (eight 16-bit codes per lane)
    // before the placement pass starts writing entries -- no LDS of its own, so every tier keeps its occupancy
    constexpr bool SURV_LIST = TEAM == 1 && !DENSE;
    // a team's waves share ONE survivor list: every wave culls its share of the candidate chunks once and appends what
    // survives (tile-relative position, class ids, x-reach) -- the placement pass then walks ~450 survivors split over the
    // team instead of re-culling ~1 500 candidates.  The list borrows the LDS of the end-of-tile reduction (s_red).
    constexpr int SV_CAP = TEAM > 1 ? 1024 : 0;
    constexpr int TEAM_WORDS = TEAM > 1 ? ((CHG * K * WAVE > 4 * SV_CAP + SV_CAP / 4) ? CHG * K * WAVE : 4 * SV_CAP + SV_CAP / 4) : 1;
    __shared__ __attribute__((aligned(16))) unsigned s_team[TEAM_WORDS];
    __shared__ unsigned s_nsv;
    unsigned* const s_red = s_team;
    float* const sv_x = reinterpret_cast<float*>(s_team);
    float* const sv_y = sv_x + SV_CAP;
    float* const sv_z = sv_y + SV_CAP;
    unsigned* const sv_ids = s_team + 3 * SV_CAP;
    unsigned char* const sv_xr = reinterpret_cast<unsigned char*>(s_team + 4 * SV_CAP);
    unsigned short* const s_surv = reinterpret_cast<unsigned short*>(sz);
    static_assert(2 * ESTRIDE >= SURV_CAP, "the survivor codes fit the z array");
    constexpr int SURV_REGS = (SURV_CAP + WAVE - 1) / WAVE;
    unsigned surv_code[SURV_REGS];                          // this lane's codes (survivors lane, lane + 64, ...), for the placement pass

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = TEAM > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int kb = TEAM <= K ? wv * KE : wv / CSPLIT;
    const int cb = (wv % CSPLIT) * CE;
    const bool lead = lane == 0 && wv == 0;               // the one thread of the tile's team that reports to global memory
    // x of this wave's plane j relative to the tile centre (compile-time constants for the one-wave kernel)
    auto pl_x = [&](int j) { return plane_x<K>(j); };
    auto pl_slope = [&](int j) { return -2.f * pl_x(j); };

    TileGeom tg;
    tg.b = (int)(lt / (unsigned)g.ntiles);
    // the sigma classes of the call, or of this tile's item (k_prepass_items builds one table per item): one 64-byte
    // load issued before the geometry arithmetic below, words 0..14 = w bits of the classes, word 15 = overflow marker
    const unsigned* __restrict__ table = cls_table + (g.cls_per_item ? (size_t)tg.b * CLS_TABLE_WORDS : (size_t)0);
    const unsigned table_word = (lane < CLS_TABLE_WORDS) ? table[lane] : CLS_EMPTY;
    {
        int t = (int)(lt - (unsigned)tg.b * (unsigned)g.ntiles);
        const int tz = t % g.tnz; t /= g.tnz;
        const int ty = t % g.tny;
        const int tx = t / g.tny;
        tg.x0 = tx * K; tg.y0 = ty * 8; tg.z0 = tz * 8;
    }
    const int ly = lane >> 3, lz = lane & 7;
    const float Y = (float)ly - 3.5f, Z = (float)lz - 3.5f;
    const mk_f2 Y2 = mk_f2_splat(Y), Z2 = mk_f2_splat(Z);
    constexpr float HX = 0.5f * (float)(K - 1);
    const float R2 = g.R2, INF = mk_inf();
    constexpr unsigned INF_BITS = 0x7f800000u;

    // padded cell ranges that can hold atoms within the cutoff of this tile
    {
        const int h = g.h, csl = g.cs_log2;
        tg.cx_lo = ((tg.x0 - g.rint) >> csl) + h; tg.cx_hi = ((tg.x0 + K - 1 + g.rint) >> csl) + h;
        tg.cy_lo = ((tg.y0 - g.rint) >> csl) + h; tg.cy_hi = ((tg.y0 + 7 + g.rint) >> csl) + h;
        tg.cz_lo = ((tg.z0 - g.rint) >> csl) + h; tg.cz_hi = ((tg.z0 + 7 + g.rint) >> csl) + h;
        tg.cx_lo = tg.cx_lo < 0 ? 0 : tg.cx_lo; tg.cx_hi = tg.cx_hi > g.ncx - 1 ? g.ncx - 1 : tg.cx_hi;
        tg.cy_lo = tg.cy_lo < 0 ? 0 : tg.cy_lo; tg.cy_hi = tg.cy_hi > g.ncy - 1 ? g.ncy - 1 : tg.cy_hi;
        tg.cz_lo = tg.cz_lo < 0 ? 0 : tg.cz_lo; tg.cz_hi = tg.cz_hi > g.ncz - 1 ? g.ncz - 1 : tg.cz_hi;
        // cell centre (voxel coords) minus tile centre, per axis:  (pc-h)*cs + (cs-1)/2 - (x0 + HX)
        const float cmid = 0.5f * (float)(g.cs - 1);
        tg.offx = cmid - (float)(h * g.cs) - ((float)tg.x0 + HX);
        tg.offy = cmid - (float)(h * g.cs) - ((float)tg.y0 + 3.5f);
        tg.offz = cmid - (float)(h * g.cs) - ((float)tg.z0 + 3.5f);
        tg.fcs = (float)g.cs;
    }

    const CandRuns runs = find_candidate_runs<K>(g, tg, cell_start);   // issued early: the loads fly during the set-up below

    // running minima kept as BIT PATTERNS: every candidate value is a non-negative float (or +inf /
    // NaN), for which unsigned-integer order == float order and NaN (0x7fc00000) sorts above +inf,
    // so v_min_u32 / v_min3_u32 are exact NaN-ignoring float minima with no canonicalisation op.
    unsigned q[CHG][KL];
#pragma unroll
    for (int c = 0; c < CHG; ++c)
#pragma unroll
        for (int k = 0; k < KL; ++k) q[c][k] = INF_BITS;

    // "more sigma classes than the table holds": for a call-wide table (one hot line) this is checked up front; a
    // per-item table is a fresh line per item, so the check waits until the first traversal has hidden the load
    bool general = !DENSE && (g.force_general || (!g.cls_per_item && mk_readlane(table_word, CLS_OVERFLOW) != CLS_EMPTY));
    const unsigned* __restrict__ clsp = rec_cls + (size_t)gq * g.M;
    const float4* __restrict__ w0p = rec_w + (size_t)(gq * 2 + 0) * g.M;
    const float4* __restrict__ w1p = rec_w + (size_t)(gq * 2 + 1) * g.M;
    // lane s < NCLS holds the w bits of class s (read back with a uniform-lane register read)
    const unsigned my_class_w = (lane < NCLS) ? table_word : INF_BITS;

    if (!general) {
        MK_PHASE_MARK(0);                                   // prologue
        // ---- traversal 1: cull and histogram the buckets (traversal 2 places; the records are L2-hot then) ----
        // bucket = (channel, class, x-reach): an entry whose x lies more than the cutoff below the
        // middle of the tile cannot reach the upper K/2 planes (those pairs would fail d^2 < 25 on
        // dx^2 alone) and vice versa, so such entries are only run against their half of the planes.
        // (how far along x an entry reaches depends on how far OUTSIDE the tile's y-z square it sits: rx^2 = R^2 - dyz^2;
        //  with the plain cutoff as reach 330 pair tests per voxel, tools/tile_model.py)
        const float R2m = R2 * 1.000002f;
        auto x_reach = [&](float ex, float ey, float ez, int pk) {
            const float dy = mk_max(mk_abs(ey) - 3.5f, 0.f), dz = mk_max(mk_abs(ez) - 3.5f, 0.f);
            const float rx2 = reach_r2(g, R2m, pk) - mk_fma(dy, dy, dz * dz);
            const float lo = 0.5f - ex, hi = ex + 0.5f;               // distance to the nearest plane of the upper / lower half
            return (lo > 0.f && lo * lo > rx2) ? 1 : ((hi > 0.f && hi * hi > rx2) ? 2 : 0);
        };
#pragma unroll
        for (int i = 0; i < NBUCKET3 / WAVE; ++i) bucket[lane + i * WAVE] = 0u;
        if (TEAM > 1 && threadIdx.x == 0) s_nsv = 0u;
        mk_block_sync();
        auto count_entry = [&](bool surv, unsigned, float ex, float ey, float ez, unsigned ids, int pk) {
            const int xr = x_reach(ex, ey, ez, pk);
            for_each_present_channel(surv ? ids : 0u, [&](int c, unsigned id) {
                (void)mk_lds_add(&bucket[(c * NSLOT + (int)id - 1) * NXR + xr], 1u);
            });
        };
        // A team's wave sees every TEAM-th batch of chunks; what survives the cull goes to the team's list (see above)
        auto count_and_list = [&](bool surv, unsigned, float ex, float ey, float ez, unsigned ids, int pk) {
            const int xr = x_reach(ex, ey, ez, pk);
            const unsigned long long m = mk_ballot(surv);
            unsigned base = 0u;
            if (lane == 0 && m != 0ull) base = mk_lds_add(&s_nsv, (unsigned)mk_popc64(m));
            const unsigned pos = mk_readlane(base, 0) + (unsigned)mk_rank_in_mask(m);
            if (surv && pos < (unsigned)SV_CAP) { sv_x[pos] = ex; sv_y[pos] = ey; sv_z[pos] = ez; sv_ids[pos] = ids; sv_xr[pos] = (unsigned char)xr; }
            for_each_present_channel(surv ? ids : 0u, [&](int c, unsigned id) {
                (void)mk_lds_add(&bucket[(c * NSLOT + (int)id - 1) * NXR + xr], 1u);
            });
        };
        // ---- the hot kernel: ONE pass over the candidates (cull + compaction of the survivors' codes), then the
        //      histogram over the survivors only ----
        unsigned nsurv = 0u;                                  // wave-uniform
        bool use_list = false;
        // survivor i of the list: its record again (an L2-hot gather), tile-relative; f(valid, ., x, y, z, class ids)
        auto for_each_survivor = [&](auto from_regs_, auto&& f) {
            constexpr bool FROM_REGS = decltype(from_regs_)::value != 0;
            constexpr int SB = SURV_BATCH;
            static_assert(SURV_REGS % SB == 0, "the batches tile the code registers");
#pragma unroll
            for (int cb = 0; cb < SURV_REGS; cb += SB) {                 // SB chunks' loads in flight
                const unsigned i0 = (unsigned)(cb * WAVE);
                if (i0 >= nsurv) break;                                  // wave-uniform
                float4 P[SB]; unsigned ids[SB]; bool ok[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const unsigned i = i0 + (unsigned)(u * WAVE + lane);
                    ok[u] = i < nsurv;
                    const unsigned code = FROM_REGS ? surv_code[cb + u] : (ok[u] ? (unsigned)s_surv[i] : 0u);
                    const unsigned r = mk_shfl(runs.r0, (int)(code >> SURV_OFF_BITS)) + (code & ((1u << SURV_OFF_BITS) - 1u));
                    P[u] = make_float4(0.f, 0.f, 0.f, 0.f); ids[u] = 0u;
                    if (ok[u]) { P[u] = rec_pos[r]; ids[u] = clsp[r]; }
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    if (i0 + (unsigned)(u * WAVE) >= nsurv) break;        // wave-uniform
                    const int pk = mk_float_as_int(P[u].w);
                    const float ex = P[u].x + ((float)(pk & 1023) * tg.fcs + tg.offx);
                    const float ey = P[u].y + ((float)((pk >> 10) & 1023) * tg.fcs + tg.offy);
                    const float ez = P[u].z + ((float)((pk >> 20) & 1023) * tg.fcs + tg.offz);
                    f(ok[u], 0u, ex, ey, ez, ids[u], pk);
                }
            }
        };
        if constexpr (SURV_LIST) {
            // (worth it when most candidates are rejected, i.e. when there are many: a sparse tile's few chunks go the old way)
            if (!(MK_DIAG & 16) && runs.codable && runs.N > 4u * WAVE) {
                auto keep_survivor = [&](bool surv, unsigned code, float, float, float, unsigned, int) {
                    const unsigned long long m = mk_ballot(surv);
                    const unsigned pos = nsurv + (unsigned)mk_rank_in_mask(m);
                    if (surv && pos < (unsigned)SURV_CAP) s_surv[pos] = (unsigned short)code;
                    nsurv += (unsigned)mk_popc64(m);
                };
                for_each_candidate<K, false, TRAV_BATCH, decltype(keep_survivor)&, true>(g, tg, runs, rec_pos, clsp, keep_survivor);
                use_list = nsurv <= (unsigned)SURV_CAP;
                mk_block_sync();                              // the list is read by other lanes than wrote it
                if (use_list) {
                    if (!(MK_DIAG & 32)) for_each_survivor(IntC<0>{}, count_entry);
#pragma unroll
                    for (int cb = 0; cb < SURV_REGS; ++cb) {              // the codes leave the z array before entries go in
                        const unsigned i = (unsigned)(cb * WAVE + lane);
                        surv_code[cb] = i < nsurv ? (unsigned)s_surv[i] : 0u;
                    }
                }
            }
        }
        if (MK_DIAG & 16) {
        } else if (use_list) {
        } else if (TEAM > 1) {
            for_each_candidate<K, true, TRAV_BATCH>(g, tg, runs, rec_pos, clsp, count_and_list, (unsigned)wv, (unsigned)TEAM);
        } else {
            for_each_candidate<K, true, TRAV_BATCH>(g, tg, runs, rec_pos, clsp, count_entry, (unsigned)wv, (unsigned)TEAM);
        }
        mk_block_sync();
        const unsigned nsv = TEAM > 1 ? mk_uniform(s_nsv) : 0u;             // the same in every wave of the team
        const bool use_sv = TEAM > 1 && nsv <= (unsigned)SV_CAP;
        if (!DENSE && g.cls_per_item && mk_readlane(table_word, CLS_OVERFLOW) != CLS_EMPTY) {
            general = true;                      // this item alone has too many classes (its records carry w, not ids)
        } else {
        MK_PHASE_MARK(1);                                   // traversal 1 (histogram)
        // ---- bucket starts: lane owns groups 2*lane, 2*lane+1 (3 sub-buckets each); every sub-bucket
        //      is padded to an even count so the pair loop never straddles two of them ----
        unsigned cnt[2 * NXR], pad[2 * NXR], start[2 * NXR];
        unsigned mine = 0;
#pragma unroll
        for (int i = 0; i < 2 * NXR; ++i) {
            cnt[i] = bucket[2 * NXR * lane + i];
            pad[i] = (cnt[i] + 1u) & ~1u;
            mine += pad[i];
        }
        const unsigned incl = wave_scan_inclusive(mine);
        const unsigned total = mk_readlane(incl, WAVE - 1);
        {
            unsigned run = incl - mine;
#pragma unroll
            for (int i = 0; i < 2 * NXR; ++i) { start[i] = run; run += pad[i]; }
        }
        if (!DENSE && total > (unsigned)ECAP_TIER[0]) {                      // wave-uniform; statistics for the tier choice
            if (lead) {
#pragma unroll
                for (int t = 0; t < NTIER; ++t)
                    if (total > (unsigned)ECAP_TIER[t]) (void)mk_atomic_add(&dense_count[1 + t], 1u);
            }
        }
        if (!DENSE && total > (unsigned)ECAP) {                              // wave-uniform
            // too dense for one round: hand the tile to the dense instance of this kernel
            if (lead) dense_list[mk_atomic_add(dense_count, 1u)] = lt + (unsigned)gq * ((unsigned)g.B * (unsigned)g.ntiles);
            return;                                                          // (the whole team: `total` is the same in every wave)
        }
        // A team splits the tile's sorted entry SLOTS into TEAM contiguous ranges of whole pairs, one per wave: a wave walks
        // only the classes its range touches (all of a small class, a part of a big one -- clamped in the loops below), so
        // the fixed work of a class (loop set-ups, the flush: ~70 instructions) is paid about once per class and team, and
        // the waves finish together.  (Dealing the pairs of every class round-robin made every wave pay for every class:
        // cfg2's tiles hold 27 classes of ~12 entries, the one-grid call was bound by exactly that; whole small classes
        // dealt round-robin were better but uneven.)  Minima are order-free and the flush is monotone in the class
        // minimum: the bits do not depend on who looked at an entry.
        unsigned my_b = 0u, my_e = 0xffffffffu;              // this wave's slots [my_b, my_e), both even (wave-uniform)
        if (TEAM > 1) {
            const unsigned share = (((total >> 1) + (unsigned)TEAM - 1u) / (unsigned)TEAM) << 1;
            my_b = (unsigned)wv * share;
            my_e = my_b + share;
        }
        const unsigned long long ne0 = mk_ballot((cnt[0] | cnt[1] | cnt[2]) != 0u && (TEAM == 1 || (start[0] < my_e && start[3] > my_b)));
        const unsigned long long ne1 = mk_ballot((cnt[3] | cnt[4] | cnt[5]) != 0u && (TEAM == 1 || (start[3] < my_e && start[5] + pad[5] > my_b)));
        // non-empty classes of channel c (which owns groups 16c..16c+15 = lanes 8c..8c+7, two groups each)
        auto class_bits = [&](int c) {
            const unsigned e = (unsigned)(ne0 >> (8 * c)) & 0xffu, o = (unsigned)(ne1 >> (8 * c)) & 0xffu;
            unsigned bits = 0;
            for (int j = 0; j < 8; ++j) bits |= (((e >> j) & 1u) << (2 * j)) | (((o >> j) & 1u) << (2 * j + 1));
            return bits;
        };
        // ---- one channel, class by class: inner loop = sub, fma, half a min3 per (voxel, entry); the
        //      class flush applies the cutoff to the class minimum and scales by w ----
        auto process_classes = [&](int c, unsigned bits, unsigned (&acc)[KL]) {
            while (bits) {                                                // wave-uniform
                const int cls = __builtin_ctz(bits);
                bits &= bits - 1u;
                const unsigned* bgp = &bucket[(c * NSLOT + cls) * NXR];                           // uniform reads
                const uint4 bg = make_uint4(mk_uniform(bgp[0]), mk_uniform(bgp[1]), mk_uniform(bgp[2]), mk_uniform(bgp[3]));
                const float wcls = mk_uint_as_float(mk_readlane(my_class_w, cls));
                // m[k] = min over the class's entries of g_k = d^2 - c_k^2 (c_k = x of plane k relative to the tile centre):
                // with D0 = ex^2 + dy^2 + dz^2 per (lane, entry), g_k = D0 - 2 c_k ex is ONE fma per (voxel, entry); the
                // plane constant c_k^2 is added to the class minimum at the flush (rounding is monotone: the same bits as
                // adding it per entry).  g_k may be negative (>= -c_k^2), so these minima are float minima (v_min3_f32).
                float m[KL];
#pragma unroll
                for (int k = 0; k < KL; ++k) { m[k] = INF; mk_keep(m[k]); }
                // planes [K0, K1) against the entries of one sub-bucket: start words b0 (this) and b1 (next); the
                // slots are padded to an even count, bit 0 of b0 says that the last slot is padding
                auto run = [&](auto k0_, auto k1_, unsigned b0, unsigned b1) {
                    constexpr int K0 = decltype(k0_)::value, K1 = decltype(k1_)::value;
                    if (MK_DIAG & 1) return;
                    constexpr int J0 = K0, J1 = K1;
                    const unsigned s0 = b0 & ~1u, odd = b0 & 1u;
                    const unsigned npairs = (((b1 & ~1u) - s0) >> 1) - odd;
                    unsigned lo = s0, hi = s0 + 2u * npairs;              // the pairs of the sub-bucket, clamped to this wave's range
                    if (TEAM > 1) { lo = lo > my_b ? lo : my_b; hi = hi < my_e ? hi : my_e; }
                    // (no interleaving: the optimizer would otherwise split m[] into two accumulator sets that
                    //  have to be merged after every one of these short runs -- measured 7 % slower)
                    // (the LDS address is the only induction variable: a trip counter ends up in a VGPR with a carry-out
                    //  compare -- one VALU instruction per trip more, PMC: 8 970 -> 8 739 per tile)
                    if (TEAM == 1 || lo < hi) {
                        const float* e = sxyz + lo;
                        const float* const e_end = sxyz + hi;
#pragma clang loop vectorize(disable) interleave(disable)
                        for (; e != e_end; e += 2) {
                            // a PAIR of entries per trip (s0 is even: 8-byte aligned), both halves of every packed op used
                            // (fetching the next pair one trip ahead was measured: 2-4 % slower, the copies cost more)
                            const mk_f2 px = mk_f2_load(e), py = mk_f2_load(e + ESTRIDE), pz = mk_f2_load(e + 2 * ESTRIDE);
                            const mk_f2 dy = Y2 - py, dz = Z2 - pz;
                            const mk_f2 d0 = mk_f2_fma(px, px, mk_f2_fma(dy, dy, dz * dz));
#pragma unroll
                            for (int k = J0; k < J1; ++k) {
                                const mk_f2 gk = mk_f2_fma(mk_f2_splat(pl_slope(k)), px, d0);
                                m[k] = mk_min3(m[k], gk[0], gk[1]);
                            }
                        }
                    }
                    const unsigned tslot = s0 + 2u * npairs;              // the unpaired last entry's slot
                    if (odd && (TEAM == 1 || (tslot >= my_b && tslot < my_e))) {                  // wave-uniform
                        const float* t = sxyz + tslot;
                        const float ex = t[0], dy = Y - t[ESTRIDE], dz = Z - t[2 * ESTRIDE];
                        const float d0 = mk_fma(ex, ex, mk_fma(dy, dy, dz * dz));
#pragma unroll
                        for (int k = J0; k < J1; ++k) m[k] = mk_min_raw(m[k], mk_fma(pl_slope(k), ex, d0));
                    }
                };
                // exact form for a class of small sigmas (w > FAST_W_MAX, rare): one compact loop, every plane against
                // every entry of the three sub-buckets (the planes an entry cannot reach fail the cutoff anyway)
                auto run_exact = [&](unsigned b0, unsigned b1) {
                    if (MK_DIAG & 1) return;
                    const unsigned s0 = b0 & ~1u;
                    unsigned lo = s0, hi = (b1 & ~1u) - (b0 & 1u);        // the entries of the sub-bucket, clamped to this wave's range
                    if (TEAM > 1) { lo = lo > my_b ? lo : my_b; hi = hi < my_e ? hi : my_e; }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                    for (unsigned i = lo; i < hi; ++i) {
                        const float* ee = sxyz + i;
                        const float px = ee[0], dy = Y - ee[ESTRIDE], dz = Z - ee[2 * ESTRIDE];
                        const float r = mk_fma(dy, dy, dz * dz);
#pragma unroll
                        for (int k = 0; k < KL; ++k) {
                            const float dx = pl_x(k) - px;
                            m[k] = mk_min_raw(m[k], mk_fma(dx, dx, r));
                        }
                    }
                };
                const bool fast = wcls <= fast_w_max<K>();                // wave-uniform
                if (fast) {
                    run(IntC<0>{}, IntC<K>{}, bg.x, bg.y);
                    run(IntC<0>{}, IntC<K / 2>{}, bg.y, bg.z);
                    run(IntC<K / 2>{}, IntC<K>{}, bg.z, bg.w);
                } else {
                    run_exact(bg.x, bg.y);
                    run_exact(bg.y, bg.z);
                    run_exact(bg.z, bg.w);
                }
                // class flush: cutoff on the class minimum (occupancy_utils.pyx:53), then scale by w
                if (!(MK_DIAG & 2))
#pragma unroll
                for (int k = 0; k < KL; ++k) {
                    const float d2 = fast ? m[k] + pl_x(k) * pl_x(k) : m[k];         // (plane_d2: g_k + c_k^2)
                    acc[k] = mk_min_bits(acc[k], d2 < R2 ? mk_abs(d2) * wcls : INF);
                }
            }
        };
        // dense tiles only: a finished channel goes straight to memory (4-byte stores, 32-byte stride)
        auto store_channel = [&](int c, const unsigned (&acc)[KL]) {
            const int y = tg.y0 + ly, z = tg.z0 + lz;
            if (y < g.ny && z < g.nz && gq * CHG + c < g.C) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int x = tg.x0 + k;
                    const size_t vox = (size_t)tg.b * (size_t)g.V + ((size_t)x * g.ny + y) * g.nz + z;
                    if (x < g.nx) out[vox * (size_t)g.C + (size_t)(gq * CHG + c)] = occupancy_from_q(mk_uint_as_float(acc[k]));
                }
            }
        };

        // ---- placement of channels [c0, c1) (`count` padded entries starting at `base` of the scan) ----
        auto place = [&](int c0, int c1, unsigned base, unsigned count) {
            const bool in_round = (lane >> 3) >= c0 && (lane >> 3) < c1;
            unsigned* const cursor = TEAM > 1 ? bucket_cur : bucket;
            if (TEAM == 1) mk_block_sync();                                  // counts read; previous round done
            if (in_round) {
#pragma unroll
                for (int i = 0; i < 2 * NXR; ++i) cursor[2 * NXR * lane + i] = start[i] - base;     // placement cursors
            }
            mk_block_sync();
            // ---- traversal 2: place the entries into their buckets ----
            const unsigned rmask = (c1 == CHG ? 0xffffffffu : ((1u << (4 * c1)) - 1u)) & ~((1u << (4 * c0)) - 1u);
            auto place_at = [&](bool surv, float ex, float ey, float ez, unsigned ids, int xr) {
                for_each_present_channel(surv ? (ids & rmask) : 0u, [&](int c, unsigned id) {
                    const unsigned pos = mk_lds_add(&cursor[(c * NSLOT + (int)id - 1) * NXR + xr], 1u);
                    sx[pos] = ex; sy[pos] = ey; sz[pos] = ez;
                });
            };
            auto place_entry = [&](bool surv, unsigned, float ex, float ey, float ez, unsigned ids, int pk) {
                place_at(surv, ex, ey, ez, ids, x_reach(ex, ey, ez, pk));
            };
            if (MK_DIAG & 64) {
            } else if (use_list) {
                for_each_survivor(IntC<1>{}, place_entry);
            } else if (use_sv) {
                for (unsigned i = (unsigned)(wv * WAVE + lane); i < nsv; i += (unsigned)(TEAM * WAVE))     // per-lane trip count: no collectives inside
                    place_at(true, sv_x[i], sv_y[i], sv_z[i], sv_ids[i], (int)sv_xr[i]);
            } else {
                for_each_candidate<K, true, TRAV_BATCH>(g, tg, runs, rec_pos, clsp, place_entry, (unsigned)wv, (unsigned)TEAM);
            }
            if (TEAM == 1) mk_block_sync();       // (a team writes the table into the COUNTS' array, which every wave has read by now)
            // cursors are dead now: the array becomes the table of sub-bucket starts (even; bit 0 = "odd count, the
            // last slot is padding"; sub-buckets are contiguous, so a group's three ranges are four consecutive words; the word after the
            // last group placed belongs to a channel outside the round, whose counts live in registers)
            if (in_round) {
#pragma unroll
                for (int i = 0; i < 2 * NXR; ++i) bucket[2 * NXR * lane + i] = (start[i] - base) | (cnt[i] & 1u);
            }
            if (lane == 0) bucket[c1 * NSLOT * NXR] = count;
            mk_block_sync();
        };

        if constexpr (!DENSE) {
            // ---- the normal case: one round takes all eight channels; minima stay in q for the epilogue ----
            MK_PHASE_MARK(2);                               // counts -> starts
            if (!(MK_DIAG & 8)) place(0, CHG, 0u, total);
            MK_PHASE_MARK(3);                               // traversal 2 (placement)
            if (!(MK_DIAG & 8))
#pragma unroll
            for (int c = 0; c < CHG; ++c) process_classes(c, class_bits(c), q[c]);
            MK_PHASE_MARK(4);                               // pair loops + class flushes
        } else {
            // ---- dense tile: consecutive channels whose padded entries fit the LDS arrays together are
            //      placed and processed in one round; a channel that does not fit on its own goes chunk
            //      by chunk through LDS with the cutoff and w applied per (voxel, entry) (the general
            //      path's values, bit for bit).  A finished channel is stored on the spot. ----
            int c0 = 0;
            while (c0 < CHG) {                                               // wave-uniform
                const unsigned base = c0 ? mk_readlane(incl, 8 * c0 - 1) : 0u;
                int c1 = c0;
                while (c1 < CHG && mk_readlane(incl, 8 * c1 + 7) - base <= (unsigned)ECAP) ++c1;
                if (c1 == c0) {
                    unsigned m[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) m[k] = INF_BITS;
                    mk_block_sync();
                    for_each_candidate<K, true, 1>(g, tg, runs, rec_pos, clsp,
                        [&](bool surv, unsigned, float ex, float ey, float ez, unsigned ids, int) {
                            const unsigned id = surv ? (ids >> (4 * c0)) & 0xfu : 0u;
                            const float wc = id ? mk_uint_as_float(table[id - 1u]) : INF;
                            const bool has = wc < INF;                       // false for +inf and NaN
                            const unsigned long long mask = mk_ballot(has);
                            if (mask == 0ull) return;                        // wave-uniform
                            const int n = mk_popc64(mask);
                            if (has) ebuf[mk_rank_in_mask(mask)] = make_float4(ex, ey, ez, wc);
                            mk_block_sync();
#pragma clang loop vectorize(disable) interleave(disable)
                            for (int i = 0; i < n; ++i) {
                                const float4 e = ebuf[i];
                                const float dy = Y - e.y, dz = Z - e.z;
                                float d2[K];
                                entry_d2<K>(e.x, mk_fma(dy, dy, dz * dz), e.w, d2);
#pragma unroll
                                for (int k = 0; k < K; ++k)
                                    m[k] = mk_min_bits(m[k], d2[k] < R2 ? mk_abs(d2[k]) * e.w : INF);   // occupancy_utils.pyx:53
                            }
                            mk_block_sync();                                 // ebuf is rewritten next
                        });
                    store_channel(c0, m);
                    ++c0;
                    continue;
                }
                const unsigned count = mk_readlane(incl, 8 * c1 - 1) - base;
                if (count != 0u) place(c0, c1, base, count);
                for (int c = c0; c < c1; ++c) {                               // wave-uniform, not unrolled
                    unsigned acc[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[k] = INF_BITS;
                    if (count != 0u) process_classes(c, class_bits(c), acc);
                    store_channel(c, acc);
                }
                c0 = c1;
            }
            return;                                                          // every channel stored
        }
        }                                                                    // (item not general)
    }

    if (general) {
        // ---- general path (arbitrary per-entry sigma: more sigma classes than the id table holds):
        //      chunk by chunk, per-channel compaction through LDS, cutoff test per (voxel, entry) ----
        mk_block_sync();
        auto body = [&](bool surv, unsigned r, float ex, float ey, float ez, unsigned, int) {
            float4 W0 = make_float4(INF, INF, INF, INF), W1 = W0;
            if (surv) { W0 = w0p[r]; W1 = w1p[r]; }
            const float wch[CHG] = {W0.x, W0.y, W0.z, W0.w, W1.x, W1.y, W1.z, W1.w};
#pragma unroll
            for (int c = 0; c < CHG; ++c) {
                const float wc = wch[c];
                const bool has = surv && (wc < INF);                // false for +inf and NaN
                const unsigned long long mask = mk_ballot(has);
                if (mask == 0ull) continue;                          // wave-uniform
                const int n = mk_popc64(mask);
                if (has) ebuf[mk_rank_in_mask(mask)] = make_float4(ex, ey, ez, wc);
                mk_block_sync();
#pragma clang loop vectorize(disable) interleave(disable)
                for (int i = (TEAM > 1 ? wv : 0); i < n; i += TEAM) {        // (a team's wave: every TEAM-th entry, all planes)
                    const float4 e = ebuf[i];
                    const float dy = Y - e.y, dz = Z - e.z;
                    float d2[KL];
                    entry_d2<K>(e.x, mk_fma(dy, dy, dz * dz), e.w, d2);                  // same fma tree as the sorted path
#pragma unroll
                    for (int k = 0; k < KL; ++k)
                        q[c][k] = mk_min_bits(q[c][k], d2[k] < R2 ? mk_abs(d2[k]) * e.w : INF);   // occupancy_utils.pyx:53
                }
                mk_block_sync();                                     // ebuf is rewritten next
            }
        };
        for_each_candidate<K, false, 1>(g, tg, runs, rec_pos, clsp, body);
    }

    // ---- a team: the waves' accumulators (each saw a quarter of the entries) meet in LDS -- unsigned minima of bit
    //      patterns, as everywhere -- and every wave takes K / TEAM planes of the result through the epilogue ----
    if constexpr (TEAM > 1) {
#pragma unroll
        for (int i = wv; i < CHG * K; i += TEAM) s_red[i * WAVE + lane] = INF_BITS;
        mk_block_sync();
#pragma unroll
        for (int c = 0; c < CHG; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (q[c][k] != INF_BITS) mk_lds_min(&s_red[(c * K + k) * WAVE + lane], q[c][k]);
        mk_block_sync();
#pragma unroll
        for (int c = 0; c < CE; ++c)
#pragma unroll
            for (int k = 0; k < KE; ++k) q[c][k] = s_red[((cb + c) * K + kb + k) * WAVE + lane];   // (this wave's planes and channels move to the front)
    }
    // ---- epilogue: q -> occupancy, one 32-byte store per voxel (z fastest across lanes) ----
    const int y = tg.y0 + ly, z = tg.z0 + lz;
    const bool yz_in = (y < g.ny) && (z < g.nz);
    // the lane's voxel in the wave's first plane, and the (wave-uniform) step to the next x-plane: one 64-bit add per plane
    // instead of the index arithmetic (three 64-bit multiplies: 15 instructions per plane, 9 % of a ligand tile's work)
    const size_t plane_vox = (size_t)g.ny * (size_t)g.nz;
    const size_t vox0 = (size_t)tg.b * (size_t)g.V + (size_t)(tg.x0 + kb) * plane_vox + (size_t)y * (size_t)g.nz + (size_t)z;
#pragma unroll
    for (int k = 0; k < KE; ++k) {
        const int x = tg.x0 + kb + k;
        float f[CHG];
#pragma unroll
        for (int c = 0; c < CE; ++c) f[c] = (MK_DIAG & 4) ? mk_uint_as_float(q[c][k]) : occupancy_from_q(mk_uint_as_float(q[c][k]));
        if (yz_in && x < g.nx) {
            const size_t vox = vox0 + (size_t)k * plane_vox;
            if constexpr (CSPLIT > 1) {                      // a big team: this wave stores CE channels of the voxel (16 or 8 bytes)
                float* o = out + vox * (size_t)g.C + (size_t)gq * CHG + cb;
                if (g.C == CHG) {
                    if constexpr (CE == 4) *reinterpret_cast<float4*>(o) = make_float4(f[0], f[1], f[2], f[3]);
                    else *reinterpret_cast<float2*>(o) = float2{f[0], f[1]};
                } else {
#pragma unroll
                    for (int c = 0; c < CE; ++c)
                        if (gq * CHG + cb + c < g.C) o[c] = f[c];
                }
            } else if (g.C == CHG) {
                float4* o = reinterpret_cast<float4*>(out + vox * CHG);
                // Two non-temporal 16-byte pieces per lane at a 32-byte stride.  ALONE this pattern writes 2.4 TB/s (a store-only
                // kernel of this launch shape: 0.91 ms per 2.15 GB; plain stores 6.4 TB/s, the plane turned through LDS into
                // whole 256-byte rows 5.8 -- tools/store_pattern.hip, profiles/r5_store_pattern.txt), but the kernel needs
                // 1.0 TB/s, spread over its whole length: same-session A/B (round 5) plain stores +0.6 % on cfg2, +1.7 % on cfg1,
                // +1.4 % on cfg4, turned +1.3 / +1.6 / +0.4 % -- not store-bound, and the non-temporal form keeps the L2 for the
                // records its neighbours re-read.  (A kernel with less arithmetic per voxel would want the plain form.)
                mk_store_result<TEAM == 1>(o, make_float4(f[0], f[1], f[2], f[3]));
                mk_store_result<TEAM == 1>(o + 1, make_float4(f[4], f[5], f[6], f[7]));
            } else {
                float* o = out + vox * (size_t)g.C + (size_t)gq * CHG;
#pragma unroll
                for (int c = 0; c < CHG; ++c)
                    if (gq * CHG + c < g.C) o[c] = f[c];
            }
        }
    }
    MK_PHASE_MARK(5);                                       // epilogue
    MK_PHASE_FLUSH(runs);
}

template <int K, int ECAP>
MK_KERNEL(64) void k_voxelize_tiles(GridDesc g, const unsigned* __restrict__ cell_start,
                                    const float4* __restrict__ rec_pos,
                                    const float4* __restrict__ rec_w, const unsigned* __restrict__ rec_cls,
                                    const unsigned* __restrict__ cls_table, float* __restrict__ out,
                                    unsigned* __restrict__ dense_count, unsigned* __restrict__ dense_list)
{
    // XCD-aware order: the dispatcher places block i on XCD i%8; give each XCD a contiguous run of
    // tiles so neighbouring tiles (which share candidate cells) hit the same 4 MiB L2.
    const unsigned per_xcd = gridDim.x >> 3;      // gridDim.x is a multiple of 8 (host pads)
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lt >= (unsigned)g.B * (unsigned)g.ntiles) return;                // whole wave leaves together
    voxelize_tile<K, false, ECAP>(g, lt, (int)blockIdx.y, cell_start, rec_pos, rec_w, rec_cls, cls_table, out, dense_count, dense_list);
}

// Same kernel held to 104 VGPRs (amdgpu_num_vgpr counts register PAIRS on gfx90a+: 52 -> 104): the ~100
// registers it leaves free on every SIMD are what lets the binning pre-pass of the NEXT call run beside it
// (at 122 -> 128 allocated x 4 waves the register file is 100 % taken and the other queue starves until the
// grid drains).  Costs ~3 % of the tile kernel (a dozen cold-path spills), wins 5-13 % on pipelined calls.
template <int K, int ECAP>
__attribute__((amdgpu_num_vgpr(52))) MK_KERNEL(64) void k_voxelize_tiles_lean(GridDesc g, const unsigned* __restrict__ cell_start,
                                    const float4* __restrict__ rec_pos,
                                    const float4* __restrict__ rec_w, const unsigned* __restrict__ rec_cls,
                                    const unsigned* __restrict__ cls_table, float* __restrict__ out,
                                    unsigned* __restrict__ dense_count, unsigned* __restrict__ dense_list)
{
    // XCD-aware order: the dispatcher places block i on XCD i%8; give each XCD a contiguous run of
    // tiles so neighbouring tiles (which share candidate cells) hit the same 4 MiB L2.
    const unsigned per_xcd = gridDim.x >> 3;      // gridDim.x is a multiple of 8 (host pads)
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lt >= (unsigned)g.B * (unsigned)g.ntiles) return;                // whole wave leaves together
    voxelize_tile<K, false, ECAP>(g, lt, (int)blockIdx.y, cell_start, rec_pos, rec_w, rec_cls, cls_table, out, dense_count, dense_list);
}

// TEAM waves per tile (voxelize_tile): for launches of fewer tiles than the chip has SIMDs, where the latency of one
// tile is the latency of the call.
constexpr int TILE_TEAM = 4;
template <int K, int ECAP, int TEAM = TILE_TEAM>
MK_KERNEL(TEAM * 64) void k_voxelize_tiles_team(GridDesc g, const unsigned* __restrict__ cell_start,
                                    const float4* __restrict__ rec_pos,
                                    const float4* __restrict__ rec_w, const unsigned* __restrict__ rec_cls,
                                    const unsigned* __restrict__ cls_table, float* __restrict__ out,
                                    unsigned* __restrict__ dense_count, unsigned* __restrict__ dense_list)
{
    const unsigned per_xcd = gridDim.x >> 3;      // gridDim.x is a multiple of 8 (host pads)
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lt >= (unsigned)g.B * (unsigned)g.ntiles) return;                // the whole team leaves together
    voxelize_tile<K, false, ECAP, TEAM>(g, lt, (int)blockIdx.y, cell_start, rec_pos, rec_w, rec_cls, cls_table, out, dense_count, dense_list);
}

// ------------------------------------------------------------------------------------------------
// Batches of LIGAND-SIZED items (cfg3, cfg5: tens of atoms per item, a few dozen tiles per item).  A tile of such an
// item holds a few dozen entries, and most of what its wave does in k_voxelize_tiles is fixed cost -- candidate runs,
// cull, histogram, scan, placement: four chains of dependent loads / LDS atomics per tile -- spent on nearly the same
// atoms 27 times over.  Here ONE workgroup of four waves takes an item: its entries are sorted by (channel, class) ONCE
// into LDS, then every wave walks tiles of the item and only converts the entries to tile-relative coordinates (entries
// out of the tile's reach become a far-away sentinel: the cull of the tile kernel, bit for bit), runs the pair loops
// and the epilogue.  Same arithmetic per (voxel, entry) as voxelize_tile, so the bits are the same.
// An item that does not fit (more than ITEM_ECAP padded entries, or more sigma classes than the table holds) is done by
// the same waves without the sort: its records chunk by chunk, per-entry w and cutoff (the general path's arithmetic).
// ------------------------------------------------------------------------------------------------
constexpr int ITEM_ECAP = 384;                       // (atom, channel) entries of an item, groups padded to even
constexpr int ITEM_STRIDE = (ITEM_ECAP + 2 + 3) & ~3;
constexpr int ITEM_MAX_RECORDS = 4 * ITEM_ECAP;      // items with more records are not even counted

template <int K>
MK_DEV void voxelize_item_tile(const GridDesc& g, const int b, const int t, const int gq, const float4* s_ent,
                               const unsigned total, const unsigned* s_gstart /* [NBUCKET + 1]: start | odd */,
                               const unsigned* s_cbits /* [CHG]: non-empty classes of a channel */,
                               const unsigned char* s_grp /* [total]: group of an entry, 0xff = padding */, unsigned* cur /* [NBUCKET], this wave's */,
                               const unsigned my_class_w, float* stage, float* __restrict__ out)
{
    const int lane = threadIdx.x & (WAVE - 1);
    constexpr float HX = 0.5f * (float)(K - 1);
    const int tz = t % g.tnz, ty = (t / g.tnz) % g.tny, tx = t / (g.tnz * g.tny);
    const int x0 = tx * K, y0 = ty * 8, z0 = tz * 8;
    const int ly = lane >> 3, lz = lane & 7;
    const float Y = (float)ly - 3.5f, Z = (float)lz - 3.5f;
    const mk_f2 Y2 = mk_f2_splat(Y), Z2 = mk_f2_splat(Z);
    const float R2 = g.R2, INF = mk_inf();
    constexpr unsigned INF_BITS = 0x7f800000u;
    // cell centre (voxel coords) minus tile centre, per axis (as in voxelize_tile)
    const float cmid = 0.5f * (float)(g.cs - 1), fcs = (float)g.cs;
    const float offx = cmid - (float)(g.h * g.cs) - ((float)x0 + HX);
    const float offy = cmid - (float)(g.h * g.cs) - ((float)y0 + 3.5f);
    const float offz = cmid - (float)(g.h * g.cs) - ((float)z0 + 3.5f);
    float* const sx = stage;
    // ---- the item's entries relative to THIS tile (cand_consume's arithmetic and cull); the ones within reach move to
    //      the front of their group's slots (one LDS atomic each; their order inside a group is free: minima) ----
    for (int gi = lane; gi < NBUCKET; gi += WAVE) cur[gi] = 0u;
    mk_wave_sync();
    for (unsigned i = (unsigned)lane; i < total; i += WAVE) {
        const unsigned gi = s_grp[i];
        if (gi == 0xffu) continue;                                        // padding slot
        const float4 P = s_ent[i];
        const int pk = mk_float_as_int(P.w);
        const float ex = P.x + ((float)(pk & 1023) * fcs + offx);
        const float ey = P.y + ((float)((pk >> 10) & 1023) * fcs + offy);
        const float ez = P.z + ((float)((pk >> 20) & 1023) * fcs + offz);
        const float gx = fmaxf(fabsf(ex) - HX, 0.f), gy = fmaxf(fabsf(ey) - 3.5f, 0.f), gz = fmaxf(fabsf(ez) - 3.5f, 0.f);
        if (gx * gx + gy * gy + gz * gz < reach_r2(g, g.R2cull, pk)) {
            const unsigned slot = (s_gstart[gi] & ~1u) + mk_lds_add(&cur[gi], 1u);
            sx[slot] = ex; sx[ITEM_STRIDE + slot] = ey; sx[2 * ITEM_STRIDE + slot] = ez;
        }
    }
    mk_wave_sync();
    unsigned q[CHG][K];
#pragma unroll
    for (int c = 0; c < CHG; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) q[c][k] = INF_BITS;
    unsigned live = 0u;                                                   // channels with an entry within reach (wave-uniform)
#pragma unroll
    for (int c = 0; c < CHG; ++c) {
        unsigned bits = mk_uniform(s_cbits[c]);
        while (bits) {                                                    // wave-uniform
            const int cls = __builtin_ctz(bits);
            bits &= bits - 1u;
            const unsigned n_in = mk_uniform(cur[c * NSLOT + cls]);            // entries of the group within reach of this tile
            if (n_in == 0u) continue;
            live |= 1u << c;
            const unsigned s0 = mk_uniform(s_gstart[c * NSLOT + cls]) & ~1u, odd = n_in & 1u;
            const float wcls = mk_uint_as_float(mk_readlane(my_class_w, cls));
            float m[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { m[k] = INF; mk_keep(m[k]); }
            const bool fast = wcls <= fast_w_max<K>();                    // wave-uniform
            if (fast) {                                                   // (the pair loop of voxelize_tile, all K planes)
                const unsigned npairs = n_in >> 1;
                const float* e = sx + s0;
                const float* const e_end = e + 2u * npairs;
#pragma clang loop vectorize(disable) interleave(disable)
                for (; e != e_end; e += 2) {
                    const mk_f2 px = mk_f2_load(e), py = mk_f2_load(e + ITEM_STRIDE), pz = mk_f2_load(e + 2 * ITEM_STRIDE);
                    const mk_f2 dy = Y2 - py, dz = Z2 - pz;
                    const mk_f2 d0 = mk_f2_fma(px, px, mk_f2_fma(dy, dy, dz * dz));
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const mk_f2 gk = mk_f2_fma(mk_f2_splat(plane_slope<K>(k)), px, d0);
                        m[k] = mk_min3(m[k], gk[0], gk[1]);
                    }
                }
                if (odd) {                                                // wave-uniform: the unpaired last entry
                    const float ex = e[0], dy = Y - e[ITEM_STRIDE], dz = Z - e[2 * ITEM_STRIDE];
                    const float d0 = mk_fma(ex, ex, mk_fma(dy, dy, dz * dz));
#pragma unroll
                    for (int k = 0; k < K; ++k) m[k] = mk_min_raw(m[k], mk_fma(plane_slope<K>(k), ex, d0));
                }
            } else {                                                      // exact form for a class of small sigmas
                const unsigned n = n_in;
                const float* e = sx + s0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                for (unsigned i = 0; i < n; ++i, ++e) {
                    const float px = e[0], dy = Y - e[ITEM_STRIDE], dz = Z - e[2 * ITEM_STRIDE];
                    const float r = mk_fma(dy, dy, dz * dz);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float dx = plane_x<K>(k) - px;
                        m[k] = mk_min_raw(m[k], mk_fma(dx, dx, r));
                    }
                }
            }
            // class flush: cutoff on the class minimum (occupancy_utils.pyx:53), then scale by w
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float d2 = fast ? m[k] + plane_x<K>(k) * plane_x<K>(k) : m[k];
                q[c][k] = mk_min_bits(q[c][k], d2 < R2 ? mk_abs(d2) * wcls : INF);
            }
        }
    }
    // ---- epilogue: q -> occupancy in place (a channel without an entry in reach is all zeros: no rcp / exp for it --
    //      for these sparse tiles the 64 evaluations are a third of the work), one 32-byte store per voxel ----
#pragma unroll
    for (int c = 0; c < CHG; ++c) {
        if (live & (1u << c)) {                                           // wave-uniform
#pragma unroll
            for (int k = 0; k < K; ++k) q[c][k] = mk_float_bits(occupancy_from_q(mk_uint_as_float(q[c][k])));
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) q[c][k] = 0u;
        }
    }
    const int y = y0 + ly, z = z0 + lz;
    const bool yz_in = (y < g.ny) && (z < g.nz);
    const size_t plane_vox = (size_t)g.ny * (size_t)g.nz;                 // (see voxelize_tile: one add per plane)
    const size_t vox0 = (size_t)b * (size_t)g.V + (size_t)x0 * plane_vox + (size_t)y * (size_t)g.nz + (size_t)z;
    // the two half-voxels this lane STORES (see below): voxel s2 * 32 + lane / 2 of the tile's y-z face
    size_t t_vox[2];
    bool t_in[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int vv = s2 * 32 + (lane >> 1);
        const int yy = y0 + (vv >> 3), zz = z0 + (vv & 7);
        t_in[s2] = (yy < g.ny) && (zz < g.nz);
        t_vox[s2] = (size_t)b * (size_t)g.V + (size_t)x0 * plane_vox + (size_t)yy * (size_t)g.nz + (size_t)zz;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int x = x0 + k;
        float f[CHG];
#pragma unroll
        for (int c = 0; c < CHG; ++c) f[c] = mk_uint_as_float(q[c][k]);
        if (g.C == CHG) {
            // A lane holds the 32 bytes of ONE voxel but a store instruction takes 16 per lane: written from where they are,
            // the two instructions of a plane each put every other 16-byte piece into 16 lines, and the L2 has to merge
            // them -- the kernel was store-bound at 4.4 TB/s with its arithmetic half exposed (round 3: 3.29 ms per cfg3 step,
            // 2.06 without the stores, 2.70 with the stores alone).  Turned through 2 KB of the wave's staging area, lane L
            // writes piece L of 1 KB = 32 voxels: 16 consecutive lanes cover one 256-byte z-row, every instruction writes
            // whole lines, and they can go non-temporal: 3.29 -> 2.64 ms (0.555 -> 0.69 of the HBM peak), cfg5 6.55 -> 5.68.
            static_assert(3 * ITEM_STRIDE * sizeof(float) >= 2 * WAVE * sizeof(float4), "the staging area holds a plane");
            float4* const tb = reinterpret_cast<float4*>(stage);
            mk_wave_sync();                                               // (the previous plane has been read)
            tb[2 * lane] = make_float4(f[0], f[1], f[2], f[3]);
            tb[2 * lane + 1] = make_float4(f[4], f[5], f[6], f[7]);
            mk_wave_sync();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float4 v = tb[2 * (s2 * 32 + (lane >> 1)) + (lane & 1)];
                if (t_in[s2] && x < g.nx)
                    mk_store_result<true>(reinterpret_cast<float4*>(out + (t_vox[s2] + (size_t)k * plane_vox) * CHG) + (lane & 1), v);
            }
        } else
        if (yz_in && x < g.nx) {
            const size_t vox = vox0 + (size_t)k * plane_vox;
            if (g.C == CHG) {
                float4* o = reinterpret_cast<float4*>(out + vox * CHG);
                mk_store_result<false>(o, make_float4(f[0], f[1], f[2], f[3]));
                mk_store_result<false>(o + 1, make_float4(f[4], f[5], f[6], f[7]));
            } else {
                float* o = out + vox * (size_t)g.C + (size_t)gq * CHG;
#pragma unroll
                for (int c = 0; c < CHG; ++c)
                    if (gq * CHG + c < g.C) o[c] = f[c];
            }
        }
    }
    mk_wave_sync();                                                       // the staging arrays are rewritten for the next tile
}

// One tile of an item whose entries were NOT sorted (see k_voxelize_items): the item's records in chunks of 64, a
// channel's entries of the chunk compacted into the wave's staging area, per-entry w and cutoff -- voxelize_tile's
// general path over the item's records instead of the tile's candidate cells.
template <int K>
MK_DEV void voxelize_item_tile_unsorted(const GridDesc& g, const int b, const int t, const int gq, const unsigned r0, const unsigned r1,
                                        const float4* __restrict__ rec_pos, const unsigned* __restrict__ clsp,
                                        const float4* __restrict__ w0p /* w1p = w0p + g.M */, const bool carries_w,
                                        const unsigned* __restrict__ table, float* stage, float* __restrict__ out)
{
    const int lane = threadIdx.x & (WAVE - 1);
    constexpr float HX = 0.5f * (float)(K - 1);
    const int tz = t % g.tnz, ty = (t / g.tnz) % g.tny, tx = t / (g.tnz * g.tny);
    const int x0 = tx * K, y0 = ty * 8, z0 = tz * 8;
    const int ly = lane >> 3, lz = lane & 7;
    const float Y = (float)ly - 3.5f, Z = (float)lz - 3.5f;
    const float R2 = g.R2, INF = mk_inf();
    const float cmid = 0.5f * (float)(g.cs - 1), fcs = (float)g.cs;
    const float offx = cmid - (float)(g.h * g.cs) - ((float)x0 + HX);
    const float offy = cmid - (float)(g.h * g.cs) - ((float)y0 + 3.5f);
    const float offz = cmid - (float)(g.h * g.cs) - ((float)z0 + 3.5f);
    float4* const ebuf = reinterpret_cast<float4*>(stage);               // 64 entries of one channel (x, y, z, w)
    static_assert(3 * ITEM_STRIDE * sizeof(float) >= WAVE * sizeof(float4), "the staging area holds one chunk");
    unsigned q[CHG][K];
#pragma unroll
    for (int c = 0; c < CHG; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) q[c][k] = 0x7f800000u;
    for (unsigned base = r0; base < r1; base += WAVE) {                  // wave-uniform
        const unsigned r = base + (unsigned)lane;
        bool surv = r < r1;
        float ex = 0.f, ey = 0.f, ez = 0.f;
        float wch[CHG];
#pragma unroll
        for (int c = 0; c < CHG; ++c) wch[c] = INF;
        if (surv) {
            const float4 P = rec_pos[r];
            const int pk = mk_float_as_int(P.w);
            ex = P.x + ((float)(pk & 1023) * fcs + offx);
            ey = P.y + ((float)((pk >> 10) & 1023) * fcs + offy);
            ez = P.z + ((float)((pk >> 20) & 1023) * fcs + offz);
            const float gx = fmaxf(fabsf(ex) - HX, 0.f), gy = fmaxf(fabsf(ey) - 3.5f, 0.f), gz = fmaxf(fabsf(ez) - 3.5f, 0.f);
            surv = gx * gx + gy * gy + gz * gz < reach_r2(g, g.R2cull, pk);
            if (surv) {
                if (carries_w) {
                    const float4 W0 = w0p[r], W1 = w0p[(size_t)g.M + r];
                    wch[0] = W0.x; wch[1] = W0.y; wch[2] = W0.z; wch[3] = W0.w; wch[4] = W1.x; wch[5] = W1.y; wch[6] = W1.z; wch[7] = W1.w;
                } else {
                    const unsigned ids = clsp[r];
#pragma unroll
                    for (int c = 0; c < CHG; ++c) {
                        const unsigned id = (ids >> (4 * c)) & 0xfu;
                        if (id) wch[c] = mk_uint_as_float(table[id - 1u]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CHG; ++c) {
            const float wc = wch[c];
            const bool has = surv && (wc < INF);                         // false for +inf and NaN
            const unsigned long long mask = mk_ballot(has);
            if (mask == 0ull) continue;                                  // wave-uniform
            const int n = mk_popc64(mask);
            if (has) ebuf[mk_rank_in_mask(mask)] = make_float4(ex, ey, ez, wc);
            mk_wave_sync();
#pragma clang loop vectorize(disable) interleave(disable)
            for (int i = 0; i < n; ++i) {
                const float4 e = ebuf[i];
                const float dy = Y - e.y, dz = Z - e.z;
                float d2[K];
                entry_d2<K>(e.x, mk_fma(dy, dy, dz * dz), e.w, d2);      // same fma tree as the sorted path
#pragma unroll
                for (int k = 0; k < K; ++k)
                    q[c][k] = mk_min_bits(q[c][k], d2[k] < R2 ? mk_abs(d2[k]) * e.w : INF);   // occupancy_utils.pyx:53
            }
            mk_wave_sync();                                              // ebuf is rewritten next
        }
    }
    const int y = y0 + ly, z = z0 + lz;
    const bool yz_in = (y < g.ny) && (z < g.nz);
    const size_t plane_vox = (size_t)g.ny * (size_t)g.nz;
    const size_t vox0 = (size_t)b * (size_t)g.V + (size_t)x0 * plane_vox + (size_t)y * (size_t)g.nz + (size_t)z;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int x = x0 + k;
        float f[CHG];
#pragma unroll
        for (int c = 0; c < CHG; ++c) f[c] = occupancy_from_q(mk_uint_as_float(q[c][k]));
        if (yz_in && x < g.nx) {
            const size_t vox = vox0 + (size_t)k * plane_vox;
            if (g.C == CHG) {
                float4* o = reinterpret_cast<float4*>(out + vox * CHG);
                mk_store_result<false>(o, make_float4(f[0], f[1], f[2], f[3]));
                mk_store_result<false>(o + 1, make_float4(f[4], f[5], f[6], f[7]));
            } else {
                float* o = out + vox * (size_t)g.C + (size_t)gq * CHG;
#pragma unroll
                for (int c = 0; c < CHG; ++c)
                    if (gq * CHG + c < g.C) o[c] = f[c];
            }
        }
    }
}

template <int K>
MK_KERNEL(TILE_TEAM * 64) void k_voxelize_items(GridDesc g, const unsigned* __restrict__ cell_start,
                                                const float4* __restrict__ rec_pos, const float4* __restrict__ rec_w,
                                                const unsigned* __restrict__ rec_cls, const unsigned* __restrict__ cls_table,
                                                float* __restrict__ out, int tiles_per_block /* a multiple of TILE_TEAM */)
{
    __shared__ __attribute__((aligned(16))) float4 s_ent[ITEM_ECAP + 2];
    __shared__ __attribute__((aligned(16))) float s_stage[TILE_TEAM][3 * ITEM_STRIDE];
    __shared__ unsigned s_cnt[NBUCKET];                       // counts, then placement cursors
    __shared__ unsigned s_gstart[NBUCKET + 1];
    __shared__ unsigned s_cbits[CHG];
    __shared__ unsigned s_total;
    __shared__ unsigned s_cur[TILE_TEAM][NBUCKET];            // per wave and tile: entries of a group within reach
    __shared__ unsigned char s_grp[ITEM_ECAP + 2];
    // few items: several workgroups per item, each with a share of its tiles (the sort of a ligand's entries is cheap
    // enough to repeat; a small batch then still fills the chip)
    const int nchunk = (g.ntiles + tiles_per_block - 1) / tiles_per_block;
    const int b = (int)blockIdx.x / nchunk, gq = (int)blockIdx.y;
    const int t_begin = ((int)blockIdx.x - b * nchunk) * tiles_per_block;
    const int t_end = t_begin + tiles_per_block < g.ntiles ? t_begin + tiles_per_block : g.ntiles;
    const int tid = (int)threadIdx.x, lane = tid & (WAVE - 1);
    // the wave's index as a SCALAR: the compiler cannot know that threadIdx.x >> 6 is the same in all lanes, and did a tile's
    // geometry (three integer divisions by run-time sizes, the float offsets) per lane in vector instructions
    const int wv = (int)mk_uniform((unsigned)(tid >> 6));
    const unsigned* __restrict__ table = cls_table + (g.cls_per_item ? (size_t)b * CLS_TABLE_WORDS : (size_t)0);
    const unsigned table_word = (lane < CLS_TABLE_WORDS) ? table[lane] : CLS_EMPTY;
    const unsigned my_class_w = (lane < NCLS) ? table_word : 0x7f800000u;
    const size_t cbase = (size_t)b * (size_t)g.cstride;
    const unsigned r0 = cell_start[cbase], r1 = cell_start[cbase + (size_t)g.ncell];      // the item's records (cell-sorted, contiguous)
    const unsigned* __restrict__ clsp = rec_cls + (size_t)gq * g.M;
    bool fits = !g.force_general && mk_readlane(table_word, CLS_OVERFLOW) == CLS_EMPTY && (r1 - r0) <= (unsigned)ITEM_MAX_RECORDS;
    if (fits) {
        if (tid < NBUCKET) s_cnt[tid] = 0u;
        mk_block_sync();
        for (unsigned r = r0 + (unsigned)tid; r < r1; r += blockDim.x)
            for_each_present_channel(clsp[r], [&](int c, unsigned id) { (void)mk_lds_add(&s_cnt[c * NSLOT + (int)id - 1], 1u); });
        mk_block_sync();
        if (wv == 0) {                                        // group starts: lane owns groups 2 lane, 2 lane + 1, padded to even
            const unsigned c0 = s_cnt[2 * lane], c1 = s_cnt[2 * lane + 1];
            const unsigned p0 = (c0 + 1u) & ~1u, p1 = (c1 + 1u) & ~1u;
            const unsigned incl = wave_scan_inclusive(p0 + p1);
            const unsigned st0 = incl - p0 - p1, st1 = st0 + p0;
            s_gstart[2 * lane] = st0 | (c0 & 1u);
            s_gstart[2 * lane + 1] = st1 | (c1 & 1u);
            s_cnt[2 * lane] = st0; s_cnt[2 * lane + 1] = st1;                // placement cursors
            if (lane == WAVE - 1) { s_gstart[NBUCKET] = incl; s_total = incl; }
            // non-empty classes of channel c = groups 16 c .. 16 c + 15 = lanes 8 c .. 8 c + 7
            const unsigned long long ne0 = mk_ballot(c0 != 0u), ne1 = mk_ballot(c1 != 0u);
            if (lane < CHG) {
                const unsigned e = (unsigned)(ne0 >> (8 * lane)) & 0xffu, o = (unsigned)(ne1 >> (8 * lane)) & 0xffu;
                unsigned bits = 0;
                for (int j = 0; j < 8; ++j) bits |= (((e >> j) & 1u) << (2 * j)) | (((o >> j) & 1u) << (2 * j + 1));
                s_cbits[lane] = bits;
            }
        }
        mk_block_sync();
        fits = s_total <= (unsigned)ITEM_ECAP;                // block-uniform
    }
    if (fits) {
        for (int i = tid; i < ITEM_ECAP + 2; i += (int)blockDim.x) s_grp[i] = 0xffu;
        mk_block_sync();
        for (unsigned r = r0 + (unsigned)tid; r < r1; r += blockDim.x) {
            const float4 P = rec_pos[r];
            for_each_present_channel(clsp[r], [&](int c, unsigned id) {
                const unsigned pos = mk_lds_add(&s_cnt[c * NSLOT + (int)id - 1], 1u);
                s_ent[pos] = P;
                s_grp[pos] = (unsigned char)(c * NSLOT + (int)id - 1);
            });
        }
        mk_block_sync();
        const unsigned total = s_total;
        for (int t = t_begin + wv; t < t_end; t += TILE_TEAM)
            voxelize_item_tile<K>(g, b, t, gq, s_ent, total, s_gstart, s_cbits, s_grp, s_cur[wv], my_class_w, s_stage[wv], out);
        return;
    }
    // the item does not fit: every wave takes tiles as above, but walks the item's records chunk by chunk without sorting
    // them (per-entry w and cutoff: the general path's arithmetic) -- slow, correct for any item, and rare here
    const bool carries_w = g.force_general || mk_readlane(table_word, CLS_OVERFLOW) != CLS_EMPTY;
    for (int t = t_begin + wv; t < t_end; t += TILE_TEAM)
        voxelize_item_tile_unsorted<K>(g, b, t, gq, r0, r1, rec_pos, clsp, rec_w + (size_t)(gq * 2) * g.M, carries_w, table, s_stage[wv], out);
}

// ------------------------------------------------------------------------------------------------
// Exact cut-off decisions.  occupancy_utils.pyx:53 tests d^2 < 25 in DOUBLE; the tile kernels test float32 distances
// that carry ~3e-6 A^2 of error, so a pair within that of the cutoff can land on the wrong side.  What is then at stake is
// the value AT the cutoff, 1 - exp(-(sigma^2/25)^6): below 5e-6 for sigma <= 1.81 A (every H C N O F P S Cl -- inside
// the 1e-5 parity bound together with the 2-3e-6 of float32 noise), but 7.7e-5 for Na, 2e-3 for a user sigma of 3 A.
// So for the WIDE sigmas only (w < g.w_exact_max) every (voxel, channel) with an atom on the cutoff shell is re-evaluated
// after the tile kernels with the reference's own arithmetic in double:
//   * driven by the atoms, not by the voxels: the hot kernels are untouched.  A wave looks at the summary the pre-pass
//     already built for its 256 atoms (the block's sigma set) or its item (the item's class table): no wide sigma, the
//     common case, and it is done;
//   * a wide atom's shell is walked analytically: for every (x, y) voxel column within reach the two z where the sphere
//     |v - atom| = 5 A crosses it, the voxels next to them tested in double against a band of +-2e-4 A^2 (60 x the float32
//     error); 0.006 voxels per atom on average at 1 A, a few dozen when atoms and voxels share a lattice;
//   * each hit is recomputed over ALL atoms of the item: centre = fl64(index * res) + bb_min
//     (voxeldescriptors.py:125-132,245-247), d = float32 coordinate - centre (min-imaged for periodic items,
//     distance_utils.pyx:49-52), strict d^2 < 25, x = sigma / sqrt(d^2), 1 - exp(-x^12), max over atoms
//     (occupancy_utils.pyx:46-61), stored as float32.
// ------------------------------------------------------------------------------------------------
constexpr double EXACT_BAND_REL = 8e-6;      // band = R^2 x this (2e-4 A^2)

// the occupancy of (voxel ix,iy,iz of item b) in nch <= EXACT_CH channels (wave-uniform; their indices packed 16 bits each into `chs`:
// no register array is indexed dynamically) exactly as the reference computes it; all 64 lanes take part.
// Round 6 (profiles/r6_topology_wide_ab.txt): this used to be ONE channel per pass, one atom per lane and round trip to memory -- a
// loop whose `continue`s keep the compiler from overlapping the loads: 469 dependent round trips for a 30 000-atom frame, ~60 us per
// (voxel, channel) on one wave, and k_tail lasts as long as its unluckiest wave: 0.41-0.5 ms per cfg4-sized step for a molecule with
// eight ions.  Now (a) EXACT_BATCH atoms per lane are loaded together (24 loads in flight: 59 round trips per 30 000-atom frame), (b) a float32
// pre-test -- d^2 > 25.5 A^2 leaves at once: an error of 1e-3 A^2 at most against a margin of 0.5, whichever image the float32 rounding
// picks at half a box length, where both images are beyond the cutoff -- puts the ~50 atoms near the voxel on a list in LDS, and only
// those take the double-precision arithmetic with its divisions and read their (strided) sigma row -- one copy of that code, a lane
// per listed atom, (c) all the wide channels of the shell's atom share the pass.
constexpr int EXACT_BATCH = 8, EXACT_CH = 4, EXACT_LIST = 2 * WAVE * EXACT_BATCH;   // (the list holds two whole batches: it is emptied BETWEEN batches)
template <bool B> struct ExactFull { static constexpr bool value = B; };
template <typename SigT>
MK_DEV void exact_recompute(const GridDesc& g, int b, int ix, int iy, int iz, unsigned long long chs /* 16 bits per channel index */, int nch,
                            const float* __restrict__ coords,
                            const long long* __restrict__ atom_offsets, const SigT* __restrict__ sigmas,
                            const double* __restrict__ origins, const float* __restrict__ box,
                            const double* __restrict__ affine, float* __restrict__ out, double* s_best, unsigned* s_near /* [EXACT_LIST] */,
                            long long s_lo = 0, long long s_hi = 0x7fffffffffffffffLL /* the atoms [s_lo, s_hi) of the item only ... */,
                            bool accumulate = false /* ... joined to the stored value by an atomic maximum (k_exact_redo: a wave per slice) */)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const double cx = mk_dadd_rn(mk_dmul_rn((double)ix, g.res), origins[3 * (size_t)b + 0]);
    const double cy = mk_dadd_rn(mk_dmul_rn((double)iy, g.res), origins[3 * (size_t)b + 1]);
    const double cz = mk_dadd_rn(mk_dmul_rn((double)iz, g.res), origins[3 * (size_t)b + 2]);
    double L[3] = {1.0, 1.0, 1.0};
    if (g.pbc) { L[0] = (double)box[3 * (size_t)b]; L[1] = (double)box[3 * (size_t)b + 1]; L[2] = (double)box[3 * (size_t)b + 2]; }
    const long long a_lo = atom_offsets[b], a_end = atom_offsets[b + 1];
    const long long a_hi = a_end - a_lo > s_hi ? a_lo + s_hi : a_end;     // (the end of this wave's slice)
    const long long sshift = g.topo_n ? a_lo : 0;                         // topology calls: the molecule's one sigma matrix
    const float fcx = (float)cx, fcy = (float)cy, fcz = (float)cz;
    const float fL[3] = {(float)L[0], (float)L[1], (float)L[2]};
    const float fiL[3] = {1.0f / fL[0], 1.0f / fL[1], 1.0f / fL[2]};
    double best[EXACT_CH];
#pragma unroll
    for (int k = 0; k < EXACT_CH; ++k) best[k] = 0.0;
    unsigned cnt = 0;                                                      // wave-uniform: atoms waiting in s_near
    // the reference's arithmetic for the atoms of the list (a lane each) -- ONE copy of the double-precision code, outside the unrolled scan
    auto flush = [&]() {
        mk_wave_sync();
        for (unsigned i0 = 0; i0 < cnt; i0 += WAVE) {                      // wave-uniform
            const unsigned i = i0 + (unsigned)lane;
            if (i >= cnt) continue;
            const long long a = a_lo + (long long)s_near[i];
            float xyz[3] = {coords[3 * a + 0], coords[3 * a + 1], coords[3 * a + 2]};
            if (affine != nullptr) {                                       // rounded to float32 like the binning (bin_atom)
                const double* A = affine + 12 * (size_t)b;
                const double x = (double)xyz[0], y = (double)xyz[1], z = (double)xyz[2];
                xyz[0] = (float)(A[0] * x + A[1] * y + A[2] * z + A[9]);
                xyz[1] = (float)(A[3] * x + A[4] * y + A[5] * z + A[10]);
                xyz[2] = (float)(A[6] * x + A[7] * y + A[8] * z + A[11]);
            }
            double dx = (double)xyz[0] - cx, dy = (double)xyz[1] - cy, dz = (double)xyz[2] - cz;
            if (g.pbc) {
                dx -= L[0] * round(dx / L[0]);
                dy -= L[1] * round(dy / L[1]);
                dz -= L[2] * round(dz / L[2]);
            }
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < CUTOFF2_A) {
                const double root = sqrt(d2);
#pragma unroll
                for (int k = 0; k < EXACT_CH; ++k) {
                    if (k >= nch) break;
                    const double sg = (double)sigmas[(size_t)(a - sshift) * g.C + (size_t)((chs >> (16 * k)) & 0xffffull)];
                    if (!(sg != 0.0) || sg != sg) continue;                // occupancy_utils.pyx:55-56 (a NaN is never stored)
                    const double x = sg / root;
                    const double x3 = x * x * x;
                    const double val = 1.0 - exp(-(x3 * x3 * x3 * x3));
                    if (val > best[k]) best[k] = val;
                }
            }
        }
        mk_wave_sync();
        cnt = 0;
    };
    // a batch of WAVE x EXACT_BATCH atoms from `base` on.  FULL: all of them exist -- the loads are ONE address per lane plus constant
    // offsets (no clamp, no 64-bit arithmetic per load: with it the 48 addresses of a batch alone took ~100 registers)
    auto scan = [&](long long base, auto full_) {
        constexpr bool FULL = decltype(full_)::value;
        float X[EXACT_BATCH], Y[EXACT_BATCH], Z[EXACT_BATCH];
        const float* __restrict__ p0 = coords + 3 * (base + lane);
#pragma unroll
        for (int u = 0; u < EXACT_BATCH; ++u) {                            // every load of the batch before the first use
            if constexpr (FULL) {
                X[u] = p0[3 * WAVE * u + 0]; Y[u] = p0[3 * WAVE * u + 1]; Z[u] = p0[3 * WAVE * u + 2];
            } else {
                const long long a = base + (long long)u * WAVE + lane, ai = a < a_hi ? a : a_hi - 1;
                X[u] = coords[3 * ai + 0]; Y[u] = coords[3 * ai + 1]; Z[u] = coords[3 * ai + 2];
            }
        }
#pragma unroll
        for (int u = 0; u < EXACT_BATCH; ++u) {
            const long long a = base + (long long)u * WAVE + lane;
            bool near = FULL || a < a_hi;
            if (affine == nullptr) {                                       // (an augmented call: every atom takes the exact path)
                float fx = X[u] - fcx, fy = Y[u] - fcy, fz = Z[u] - fcz;
                if (g.pbc) { fx -= fL[0] * mk_rint(fx * fiL[0]); fy -= fL[1] * mk_rint(fy * fiL[1]); fz -= fL[2] * mk_rint(fz * fiL[2]); }
                near = near && (fx * fx + fy * fy + fz * fz <= 25.5f);     // far (or NaN: the reference's `d2 < 25` is false too)
            }
            const unsigned long long m = mk_ballot(near);
            if (m == 0ull) continue;                                       // wave-uniform
            if (near) s_near[cnt + (unsigned)mk_rank_in_mask(m)] = (unsigned)(a - a_lo);
            cnt += (unsigned)mk_popc64(m);
        }
        if (cnt > (unsigned)(EXACT_LIST - WAVE * EXACT_BATCH)) flush();   // (not inside the unrolled scan: the batch's 24 registers would live across it)
    };
    long long base = a_lo + s_lo;
    for (; base + (long long)WAVE * EXACT_BATCH <= a_hi; base += (long long)WAVE * EXACT_BATCH) scan(base, ExactFull<true>{});   // wave-uniform
    for (; base < a_hi; base += (long long)WAVE * EXACT_BATCH) scan(base, ExactFull<false>{});
    flush();
#pragma unroll
    for (int k = 0; k < EXACT_CH; ++k) {                                   // wave-uniform
        if (k >= nch) break;
        s_best[lane] = best[k];
        mk_block_sync();
        if (lane == 0) {
            double m = s_best[0];
            for (int j = 1; j < WAVE; ++j) m = s_best[j] > m ? s_best[j] : m;
            const size_t vox = (size_t)b * (size_t)g.V + ((size_t)ix * g.ny + iy) * g.nz + iz;
            float* dst = out + vox * (size_t)g.C + (size_t)((chs >> (16 * k)) & 0xffffull);
            // (occupancies are >= +0: their bit patterns order like the values, so the slices of k_exact_redo meet in an integer maximum)
            if (accumulate) mk_atomic_max(reinterpret_cast<unsigned*>(dst), mk_float_bits((float)m));
            else *dst = (float)m;
        }
        mk_block_sync();
    }
}

// The list of hits a topology call with wide atoms leaves for k_exact_redo: REDO_HEAD words (the count, then "a hit did not fit"), then
// REDO_ENTRY words per hit: item, voxel x / y / z, the channel indices (16 bits each, two words), their number.
enum { REDO_COUNT = 0, REDO_OVERFLOW = 1, REDO_HEAD = 4, REDO_ENTRY = 8 };
constexpr int REDO_SLICE = WAVE * EXACT_BATCH * 4;   // atoms per wave of k_exact_redo (four batches: 2 048)

// The cut-off shell of ONE wide atom `a` of item `b` (all 64 lanes): every (voxel, wide channel) the shell passes is recomputed
// exactly.  `srow` = the atom's row in `sigmas` (a topology call: its index inside the molecule).
template <typename SigT, int LIST_ONLY = 0 /* 1: hits go to the list or, when it is full, raise its overflow word -- no recompute in this code at all */>
MK_DEV void exact_fixup_atom(const GridDesc& g, const int b, const long long a, const long long srow, const float* __restrict__ coords,
                             const long long* __restrict__ atom_offsets, const SigT* __restrict__ sigmas, const double* __restrict__ origins,
                             const float* __restrict__ box, const double* __restrict__ affine, float* __restrict__ out, double* s_best,
                             unsigned* s_near, unsigned* __restrict__ feedback, unsigned seq,
                             unsigned* __restrict__ redo_list = nullptr /* != nullptr: hits are listed for k_exact_redo */, unsigned redo_cap = 0u)
{
    const int lane = threadIdx.x;
    const float wmax = g.w_exact_max;
    const double R = CUTOFF_A_KERNEL / g.res, R2 = R * R, band = R2 * EXACT_BAND_REL, Rb = sqrt(R2 + band);
    const int nvox[3] = {g.nx, g.ny, g.nz};
    // position in voxel units, as the binning sees it
    float xyz[3] = {coords[3 * a + 0], coords[3 * a + 1], coords[3 * a + 2]};
    if (affine != nullptr) {
        const double* A = affine + 12 * (size_t)b;
        const double x = (double)xyz[0], y = (double)xyz[1], z = (double)xyz[2];
        xyz[0] = (float)(A[0] * x + A[1] * y + A[2] * z + A[9]);
        xyz[1] = (float)(A[3] * x + A[4] * y + A[5] * z + A[10]);
        xyz[2] = (float)(A[6] * x + A[7] * y + A[8] * z + A[11]);
    }
    double p[3], Lv[3] = {0.0, 0.0, 0.0};
    int k0[3] = {0, 0, 0}, k1[3] = {0, 0, 0};
    bool skip = false;
    for (int ax = 0; ax < 3; ++ax) {
        p[ax] = ((double)xyz[ax] - origins[3 * (size_t)b + ax]) * g.inv_res;
        if (g.pbc) {
            const double Lx = (double)box[3 * (size_t)b + ax] * g.inv_res;
            if (!(Lx > 2.0 * (g.Rp - 1e-3))) skip = true;          // bad box: the binning has raised the error flag
            Lv[ax] = Lx;
            const double i0 = ceil((-Rb - p[ax]) / Lx), i1 = floor(((double)(nvox[ax] - 1) + Rb - p[ax]) / Lx);
            if (!(i1 - i0 <= 64.0)) skip = true;
            k0[ax] = (int)i0; k1[ax] = (int)i1;
        }
        if (!(p[ax] == p[ax])) skip = true;                        // NaN coordinate
    }
    if (skip) return;
    for (int kx = k0[0]; kx <= k1[0]; ++kx)
    for (int ky = k0[1]; ky <= k1[1]; ++ky)
    for (int kz = k0[2]; kz <= k1[2]; ++kz) {                      // wave-uniform: the periodic images
        const double qx = p[0] + kx * Lv[0], qy = p[1] + ky * Lv[1], qz = p[2] + kz * Lv[2];
        const double fx0 = ceil(qx - Rb), fx1 = floor(qx + Rb), fy0 = ceil(qy - Rb), fy1 = floor(qy + Rb);
        const int x_lo = fx0 > 0.0 ? (int)fx0 : 0, x_hi = fx1 < (double)(g.nx - 1) ? (int)fx1 : g.nx - 1;
        const int y_lo = fy0 > 0.0 ? (int)fy0 : 0, y_hi = fy1 < (double)(g.ny - 1) ? (int)fy1 : g.ny - 1;
        if (x_hi < x_lo || y_hi < y_lo) continue;
        const int nyr = y_hi - y_lo + 1, ncol = (x_hi - x_lo + 1) * nyr;
        for (int cb = 0; cb < ncol; cb += WAVE) {                  // wave-uniform: 64 voxel columns at a time
            const int col = cb + lane;
            const int ix = x_lo + (col < ncol ? col / nyr : 0), iy = y_lo + (col < ncol ? col % nyr : 0);
            const double dx = (double)ix - qx, dy = (double)iy - qy, r2 = R2 - dx * dx - dy * dy;
            const double zr = sqrt(r2 > 0.0 ? r2 : 0.0);
            const long long ia = (long long)floor(qz - zr + 0.5), ib = (long long)floor(qz + zr + 0.5);
            for (int s = 0; s < 6; ++s) {                          // the voxels next to the two crossings
                const long long iz = s < 3 ? ia - 1 + s : ib - 1 + (s - 3);
                const double dz = (double)iz - qz, d2 = dx * dx + dy * dy + dz * dz;
                const bool hit = col < ncol && r2 >= -band && iz >= 0 && iz < (long long)g.nz && (s < 3 || iz > ia + 1) &&
                                 fabs(d2 - R2) <= band;
                unsigned long long hits = mk_ballot(hit);
                while (hits) {                                     // wave-uniform: rare
                    const int hl = mk_ctz64(hits);
                    hits &= hits - 1ull;
                    const int vx = (int)mk_readlane((unsigned)ix, hl), vy = (int)mk_readlane((unsigned)iy, hl);
                    const int vz = (int)mk_readlane((unsigned)(int)iz, hl);
                    if (feedback != nullptr && lane == 0) feedback[FB_TAIL_WROTE] = seq;   // (the host may have read the tile kernel's values already)
                    unsigned long long chs = 0ull;                 // the atom's wide channels, EXACT_CH of them per pass over the item's atoms
                    int nch = 0;
                    for (int c = 0; c < g.C; ++c) {
                        if (sigma_to_w(sigmas[(size_t)srow * g.C + c], g.w_scale) < wmax) { chs |= (unsigned long long)(c & 0xffff) << (16 * nch); ++nch; }
                        if (nch == EXACT_CH || (c == g.C - 1 && nch)) {
                            // A topology call with wide atoms lists the hit for k_exact_redo (the item's atoms in slices, a wave each) and
                            // zeroes the values it will rebuild; a full list: here and now, this wave over all of the item's atoms.
                            bool listed = false;
                            if (redo_list != nullptr) {
                                unsigned slot = 0u;
                                if (lane == 0) slot = mk_atomic_add(&redo_list[REDO_COUNT], 1u);
                                slot = mk_uniform(mk_shfl(slot, 0));
                                if (slot < redo_cap) {
                                    const size_t vox = (size_t)b * (size_t)g.V + ((size_t)vx * g.ny + vy) * g.nz + vz;
                                    if (lane == 0) {
                                        unsigned* e = redo_list + REDO_HEAD + (size_t)slot * REDO_ENTRY;
                                        e[0] = (unsigned)b; e[1] = (unsigned)vx; e[2] = (unsigned)vy; e[3] = (unsigned)vz;
                                        e[4] = (unsigned)(chs & 0xffffffffull); e[5] = (unsigned)(chs >> 32); e[6] = (unsigned)nch; e[7] = 0u;
                                    }
                                    if (lane < nch) out[vox * (size_t)g.C + (size_t)((chs >> (16 * lane)) & 0xffffull)] = 0.f;
                                    listed = true;
                                }
                            }
                            if constexpr (LIST_ONLY) {
                                if (!listed && lane == 0) redo_list[REDO_OVERFLOW] = 1u;      // (k_exact_shells<.., true> walks the shells once more, in place)
                            } else {
                                if (!listed)
                                    exact_recompute<SigT>(g, b, vx, vy, vz, chs, nch, coords, atom_offsets, sigmas, origins, box, affine, out, s_best, s_near);
                            }
                            chs = 0ull; nch = 0;
                        }
                    }
                }
            }
        }
    }
}

// One wave per 256 atoms of the binning (per_item == 0: `summary` = the blocks' sigma sets, CLS_BLOCK_SET words each;
// nullptr = no summary, look at every atom), per item (per_item == 1: `summary` = the items' class tables), or -- a topology
// call, per_item == 2 -- per (item, wide atom of the molecule): `summary` = the handle's list of the g.topo_wide atoms that have
// a wide sigma, job = item * g.topo_wide + position in the list.  (Round 5 gave a topology call ONE job per item: a single wave
// walked all of the item's atoms 64 at a time and took its wide atoms one after the other -- for 30 000-atom frames with a
// few ions ~470 dependent load / ballot rounds on B waves; now the B x n_wide shells are spread over all fix-up waves and
// nobody looks at an atom that is not wide.)
template <typename SigT, int LIST_ONLY = 0>
MK_DEV void exact_fixup_block(const GridDesc& g, const unsigned blk, int per_item, const unsigned* __restrict__ summary,
                              const float* __restrict__ coords, const long long* __restrict__ atom_offsets,
                              long long total_atoms, const SigT* __restrict__ sigmas, const double* __restrict__ origins,
                              const float* __restrict__ box, const double* __restrict__ affine,
                              const uint2* __restrict__ tmp_cls, float* __restrict__ out, double* s_best, unsigned* s_near,
                              unsigned* __restrict__ feedback = nullptr, unsigned seq = 0u,
                              unsigned* __restrict__ redo_list = nullptr, unsigned redo_cap = 0u)
{
    const int lane = threadIdx.x;
    const float wmax = g.w_exact_max;
    auto wide_bits = [&](unsigned bits) { return bits != CLS_EMPTY && mk_uint_as_float(bits) < wmax; };   // NaN: false
    if (per_item == 2) {
        const int b = (int)(blk / g.topo_wide);
        const long long k = (long long)summary[blk % g.topo_wide];         // the atom's index inside the molecule
        const long long a_lo = atom_offsets[b], a_hi = atom_offsets[b + 1];
        // an item that is not the topology's atom count long: the binning has raised MK_ERR_TOPOLOGY; nothing of the handle's
        // is indexed with it (the C API promises a clean MKAMD_EINVAL at the next synchronize, not a read past the handle)
        if (a_hi - a_lo != (long long)g.topo_n || k >= (long long)g.topo_n) return;
        exact_fixup_atom<SigT, LIST_ONLY>(g, b, a_lo + k, k, coords, atom_offsets, sigmas, origins, box, affine, out, s_best, s_near, feedback, seq,
                                          redo_list, redo_cap);
        return;
    }
    // ---- the summary: is there anything wide among this wave's atoms at all? ----
    long long a_lo, a_hi;
    if (per_item) {
        const int b = (int)blk;
        a_lo = atom_offsets[b]; a_hi = atom_offsets[b + 1];
        if (summary != nullptr) {
            const unsigned t = lane < CLS_TABLE_WORDS ? summary[(size_t)b * CLS_TABLE_WORDS + lane] : CLS_EMPTY;
            const bool maybe = lane == CLS_OVERFLOW ? t != CLS_EMPTY : wide_bits(t);   // overflowed table: look at the atoms
            if (mk_ballot(maybe) == 0ull) return;
        }
    } else {
        a_lo = (long long)blk * 256;
        a_hi = a_lo + 256 < total_atoms ? a_lo + 256 : total_atoms;
        if (summary != nullptr) {
            const unsigned t = lane < CLS_BLOCK_SET ? summary[(size_t)blk * CLS_BLOCK_SET + lane] : CLS_EMPTY;
            if (mk_ballot(t == CLS_TOO_MANY || wide_bits(t)) == 0ull) return;
        }
    }
    int b_hint = per_item ? (int)blk : item_of_atom(atom_offsets, g.B, a_lo, 0);
    for (long long base = a_lo; base < a_hi; base += WAVE) {               // wave-uniform
        const long long a_mine = base + lane;
        bool wide = false;
        if (a_mine < a_hi) {
            for (int gq = 0; gq < g.G; ++gq) {
                const uint2 cw = tmp_cls[(size_t)a_mine * g.G + gq];
                if (cw.y == ATOM_MULTI_SIGMA) {
                    for (int c = gq * CHG; c < g.C && c < gq * CHG + CHG; ++c)
                        wide |= sigma_to_w(sigmas[(size_t)a_mine * g.C + c], g.w_scale) < wmax;
                } else wide |= wide_bits(cw.x);
            }
        }
        unsigned long long todo = mk_ballot(wide);
        while (todo) {                                                     // wave-uniform: one wide atom at a time
            const int l = mk_ctz64(todo);
            todo &= todo - 1ull;
            const long long a = base + l;
            const int b = per_item ? (int)blk : item_of_atom(atom_offsets, g.B, a, b_hint);
            b_hint = b;
            exact_fixup_atom<SigT>(g, b, a, a, coords, atom_offsets, sigmas, origins, box, affine, out, s_best, s_near, feedback, seq);
        }
    }
}

// The last launch of a call: (a) the tiles the tile kernel left for the dense instance (usually none), (b) the
// exact cut-off fix-up, which must see a tile's final values -- in ONE launch instead of two (5 us of every small
// call).  `dense_wgs` workgroups take dense tiles; the rest are fix-up waves, and only when there WERE dense tiles do
// they wait for the dense ones to finish.  That wait must not depend on the order workgroups are dispatched in (HIP
// promises none): when there are dense tiles a workgroup's ROLE is not its block index but a ticket it draws as it
// starts running -- the first `dense_wgs` tickets are the dense roles.  Whoever waits therefore drew a later ticket than
// every dense role: those workgroups are already running (or done), hold their resources and cannot be starved by the
// waiters.  Without dense tiles (nearly every call) nobody waits and the block index is the role, no atomic at all.
// Block 0 also mirrors the tier statistics and the error flag to host-visible memory and clears the OTHER copy of the
// dense words (two copies alternate from call to call, so that nothing still reads what is being cleared).
template <int K, int ECAP, typename SigT>
MK_KERNEL(64) void k_tail(GridDesc g, unsigned dense_wgs, const unsigned* __restrict__ cell_start,
                          const float4* __restrict__ rec_pos, const unsigned* __restrict__ rec_cls,
                          const unsigned* __restrict__ cls_table, float* __restrict__ out,
                          unsigned* __restrict__ dense_words, unsigned* __restrict__ other_words,
                          const unsigned* __restrict__ dense_list, unsigned* __restrict__ feedback, const int* __restrict__ err_flag,
                          int per_item, const unsigned* __restrict__ summary, const float* __restrict__ coords,
                          const long long* __restrict__ atom_offsets, long long total_atoms, const SigT* __restrict__ sigmas,
                          const double* __restrict__ origins, const float* __restrict__ box, const double* __restrict__ affine,
                          const uint2* __restrict__ tmp_cls,
                          unsigned* __restrict__ solo_counts /* k_bin_solo's counters + control words, or nullptr */, unsigned solo_n,
                          unsigned* __restrict__ solo_table, unsigned seq /* of this call (small host calls), else 0 */,
                          unsigned fix_jobs /* fix-up jobs (256-atom blocks, or items): the fix-up waves share them */)
{
    __shared__ double s_best[WAVE];
    __shared__ unsigned s_near[EXACT_LIST];
    const unsigned n = dense_words[0], total_tiles = (unsigned)g.B * (unsigned)g.ntiles;
    if (blockIdx.x == 0) {
        if (threadIdx.x <= (unsigned)DENSE_WORDS + 1u) other_words[threadIdx.x] = 0u;  // (+ the done counter and the role tickets)
        if (feedback && threadIdx.x == 0) {                                // host-visible: drives the next calls' tier
            // A small synchronous host call converts / copies the result on the host; it starts as soon as the TILE kernel is
            // done -- this launch starting says so -- instead of waiting for this launch and the stream's completion signal
            // too, and repeats the pass in the rare case that this launch changes values (dense tiles, exact cut-off hits).
            // (system scope: the tile kernel's stores into the mapped result buffer are ordered before the word the host polls)
            mk_threadfence_system();
            if (n != 0u) feedback[FB_TAIL_WROTE] = seq;
            feedback[FB_TILES_DONE] = seq;
#pragma unroll
            for (int t = 0; t < NTIER; ++t) feedback[t] = dense_words[1 + t];
            feedback[NTIER] = total_tiles * (unsigned)g.G;
            feedback[NTIER + 1] = (unsigned)*err_flag;  // mirror of the device-side error flag (set by the binning, long done)
        }
    }
    unsigned role = blockIdx.x;
    if (n != 0u) {                                                         // wave-uniform, rare: roles by order of arrival
        if (threadIdx.x == 0) role = mk_atomic_add(&dense_words[DENSE_WORDS + 1], 1u);
        role = mk_uniform(mk_shfl(role, 0));
    }
    if (role < dense_wgs) {
        for (unsigned i = role; i < n; i += dense_wgs) {                   // wave-uniform
            const unsigned e = dense_list[i];
            voxelize_tile<K, true, ECAP>(g, e % total_tiles, (int)(e / total_tiles), cell_start, rec_pos, nullptr, rec_cls, cls_table, out, nullptr, nullptr);
            mk_block_sync();                                               // LDS arrays are reused by the next tile
        }
        if (n != 0u) {                                                     // somebody may be waiting for these stores
            mk_threadfence();
            mk_block_sync();
            if (threadIdx.x == 0) (void)mk_atomic_add(&dense_words[DENSE_WORDS], 1u);
        }
        return;
    }
    if (n != 0u) {                                                         // wave-uniform, rare
        if (threadIdx.x == 0)
            while (mk_load_relaxed(&dense_words[DENSE_WORDS]) < dense_wgs) mk_sleep();
        mk_block_sync();
        mk_threadfence();
    }
    if (solo_counts != nullptr) {
        // a call binned by k_bin_solo: its counters go back to zero here, after the last reader (the dense tiles, if there
        // were any, are done: see the wait above) -- the next call finds them clean without a memset launch; a class
        // table that overflowed is emptied, so that the next call starts a new one
        const unsigned me = role - dense_wgs, nfix = gridDim.x - dense_wgs;
        for (unsigned i = me * WAVE + threadIdx.x; i < solo_n; i += nfix * WAVE) solo_counts[i] = 0u;
        const unsigned overflow_word = cls_table[CLS_OVERFLOW];
        mk_wave_sync();                                                    // (every lane has read the word before one clears it)
        if (me == 0u && threadIdx.x < (unsigned)CLS_TABLE_WORDS && overflow_word != CLS_EMPTY) solo_table[threadIdx.x] = CLS_EMPTY;
    }
    // (a big batch has tens of thousands of fix-up jobs, nearly all of which end at their summary: a few thousand waves that
    //  take several each start and drain faster than one wave per job -- 18 -> ~6 us per 256-grid step)
    for (unsigned job = role - dense_wgs; job < fix_jobs; job += gridDim.x - dense_wgs)
        exact_fixup_block<SigT>(g, job, per_item, summary, coords, atom_offsets, total_atoms, sigmas, origins, box, affine,
                                tmp_cls, out, s_best, s_near, feedback, seq);
}

// The hits k_tail listed (a topology call with wide atoms), recomputed by MANY waves: job = (hit, slice of REDO_SLICE atoms of the
// item); every wave runs exact_recompute over its slice and joins the stored value (zeroed by k_tail when it listed the hit) with an
// atomic maximum.  Round 6: one wave per hit scanned all 30 000 atoms of a cfg4-sized frame -- 59 dependent round trips, ~40 us -- and
// k_tail lasted as long as the wave with the most hits (300 ions: 317 us per 256-frame step); here a hit is 15 waves of four round trips.
// (The list's counter is zeroed by a launch of its own in front of the next call's hot kernels: a done-ticket that let the last of these
// blocks do it cost 4 096 atomics on one word -- 85 us, measured, where the recomputes themselves take ten.)
template <typename SigT>
MK_KERNEL(64) void k_exact_redo(GridDesc g, unsigned* __restrict__ redo_list, unsigned redo_cap, const float* __restrict__ coords,
                                const long long* __restrict__ atom_offsets, const SigT* __restrict__ sigmas,
                                const double* __restrict__ origins, const float* __restrict__ box, const double* __restrict__ affine,
                                float* __restrict__ out)
{
    __shared__ double s_best[WAVE];
    __shared__ unsigned s_near[EXACT_LIST];
    unsigned count = redo_list[REDO_COUNT];
    count = count < redo_cap ? count : redo_cap;
    const unsigned slices = (unsigned)((g.topo_n + REDO_SLICE - 1) / REDO_SLICE);
    const unsigned long long jobs = (unsigned long long)count * slices;
    for (unsigned long long job = blockIdx.x; job < jobs; job += gridDim.x) {          // block-uniform
        const unsigned* e = redo_list + REDO_HEAD + (size_t)(job / slices) * REDO_ENTRY;
        const long long s_lo = (long long)(job % slices) * REDO_SLICE;
        const unsigned long long chs = (unsigned long long)e[4] | ((unsigned long long)e[5] << 32);
        exact_recompute<SigT>(g, (int)e[0], (int)e[1], (int)e[2], (int)e[3], chs, (int)e[6], coords, atom_offsets, sigmas, origins, box, affine, out,
                              s_best, s_near, s_lo, s_lo + REDO_SLICE, true);
    }
}

// The fix-up jobs of a topology call with wide atoms -- (item, wide atom) shells -- in launches of their own.  Inside k_tail a fix-up
// wave carries the dense-tile role's LDS and the in-place recompute's registers (191: two waves per SIMD), and 76 800 shells of a
// 256-frame batch with 300 ions took 135 us.  INPLACE = false: walk the shells, LIST the hits (k_exact_redo follows) -- no recompute in
// this code; a hit that does not fit the list raises its overflow word.  INPLACE = true, launched behind k_exact_redo: leaves at once
// unless that word is up, else walks every shell once more and recomputes each hit where it stands (the round-3 way: always right).
template <typename SigT, bool INPLACE>
MK_KERNEL(64) void k_exact_shells(GridDesc g, unsigned fix_jobs, const unsigned* __restrict__ summary, const float* __restrict__ coords,
                                  const long long* __restrict__ atom_offsets, long long total_atoms, const SigT* __restrict__ sigmas,
                                  const double* __restrict__ origins, const float* __restrict__ box, const double* __restrict__ affine,
                                  const uint2* __restrict__ tmp_cls, float* __restrict__ out, unsigned* __restrict__ redo_list, unsigned redo_cap)
{
    __shared__ double s_best[WAVE];
    __shared__ unsigned s_near[EXACT_LIST];
    if constexpr (INPLACE) {
        if (redo_list[REDO_OVERFLOW] == 0u) return;                        // block-uniform: nearly always
        for (unsigned job = blockIdx.x; job < fix_jobs; job += gridDim.x)
            exact_fixup_block<SigT, 0>(g, job, 2, summary, coords, atom_offsets, total_atoms, sigmas, origins, box, affine, tmp_cls, out, s_best, s_near);
    } else {
        for (unsigned job = blockIdx.x; job < fix_jobs; job += gridDim.x)
            exact_fixup_block<SigT, 1>(g, job, 2, summary, coords, atom_offsets, total_atoms, sigmas, origins, box, affine, tmp_cls, out, s_best, s_near,
                                       nullptr, 0u, redo_list, redo_cap);
    }
}

MK_KERNEL(64) void k_zero_words(unsigned* __restrict__ p, unsigned n)
{
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

// ------------------------------------------------------------------------------------------------
// Explicit (non-lattice) centres: the exact calculate_occupancy contract for arbitrary `centers`
// (usercenters / direct calls).  Distances in DOUBLE exactly as the reference (float32 coords
// promoted, strict d^2 < 25), min-q reduction in float32.  Brute force O(N*V).
// A workgroup owns 64 centres (lane = centre) and its waves split the ATOMS: the atoms go through LDS in chunks of one
// atom per thread, wave w tests the w-th 64 of a chunk, and the waves' running minima meet in LDS (ds_min_u32 on the
// bit patterns: order-free, so the result does not depend on the split) before the epilogue.  The host picks 4, 8 or 16
// waves so that a small centre list still fills the chip -- with a thread per centre and every thread walking all
// atoms (round 1) a 24^3 grid ran on 54 of the 256 CUs for 100 us.  blockIdx.y = channel group.
// w here is 1/sigma^2 in A^-2 (w_scale = 1).
// ------------------------------------------------------------------------------------------------
constexpr int EXPL_CENTERS = 64;                     // centres per workgroup
constexpr int EXPL_MAX_WAVES = 16;

MK_KERNEL(EXPL_MAX_WAVES * WAVE) void k_occupancy_centers(const double* __restrict__ centers, long long V,
                                                 const float* __restrict__ coords, long long N,
                                                 const float4* __restrict__ w /* [G][2][N] */,
                                                 int C, int use_box, double bx, double by, double bz,
                                                 float* __restrict__ out)
{
    __shared__ float4 s_pos[EXPL_MAX_WAVES * WAVE];
    __shared__ float4 s_w0[EXPL_MAX_WAVES * WAVE];
    __shared__ float4 s_w1[EXPL_MAX_WAVES * WAVE];
    __shared__ unsigned s_q[CHG][EXPL_CENTERS];
    const int gq = blockIdx.y;
    const int nth = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const long long v0 = (long long)blockIdx.x * EXPL_CENTERS, v = v0 + lane;
    const bool active = v < V;
    double cx = 0, cy = 0, cz = 0;
    if (active) { cx = centers[3 * v]; cy = centers[3 * v + 1]; cz = centers[3 * v + 2]; }
    const float INF = mk_inf();
    unsigned q[CHG];
#pragma unroll
    for (int c = 0; c < CHG; ++c) q[c] = 0x7f800000u;
    for (int i = tid; i < CHG * EXPL_CENTERS; i += nth) s_q[i / EXPL_CENTERS][i % EXPL_CENTERS] = 0x7f800000u;
    mk_block_sync();

    for (long long a0 = 0; a0 < N; a0 += nth) {
        const long long a = a0 + tid;
        if (a < N) {
            s_pos[tid] = make_float4(coords[3 * a], coords[3 * a + 1], coords[3 * a + 2], 0.f);
            s_w0[tid] = w[(size_t)(gq * 2 + 0) * N + a];
            s_w1[tid] = w[(size_t)(gq * 2 + 1) * N + a];
        }
        mk_block_sync();
        const long long left = N - a0 - (long long)wv * WAVE;             // this wave's atoms of the chunk
        const int cnt = left < 0 ? 0 : (left < WAVE ? (int)left : WAVE);
        for (int k = 0; k < cnt; ++k) {
            const int i = wv * WAVE + k;
            const float4 p = s_pos[i];
            double dx = (double)p.x - cx, dy = (double)p.y - cy, dz = (double)p.z - cz;
            if (use_box) {                       // distance_utils.pyx:49-52, evaluated in double
                dx -= bx * round(dx / bx);
                dy -= by * round(dy / by);
                dz -= bz * round(dz / bz);
            }
            const double d2 = dx * dx + dy * dy + dz * dz;
            const bool in = d2 < 25.0;           // occupancy_utils.pyx:53
            const float d2f = (float)d2;
            const float4 w0 = s_w0[i], w1 = s_w1[i];
            const float wc[CHG] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int c = 0; c < CHG; ++c)
                q[c] = mk_min_bits(q[c], in ? d2f * wc[c] : INF);     // 0*inf = NaN sorts above +inf
        }
        mk_block_sync();
    }
#pragma unroll
    for (int c = 0; c < CHG; ++c) mk_lds_min(&s_q[c][lane], q[c]);
    mk_block_sync();
    // epilogue: consecutive threads take consecutive channels of a centre (the output is centre-major)
    for (int i = tid; i < CHG * EXPL_CENTERS; i += nth) {
        const int c = i % CHG, l = i / CHG;
        if (v0 + l < V && gq * CHG + c < C)
            out[(size_t)(v0 + l) * C + gq * CHG + c] = occupancy_from_q(mk_uint_as_float(s_q[c][l]));
    }
}

// sigma [N,C] -> w [G][2][N] float4 (1/sigma^2 ; +inf where the channel is absent)
template <typename SigT>
MK_KERNEL(256) void k_sigma_to_w(const SigT* __restrict__ sigmas, long long N, int C, int G,
                                 double w_scale, float4* __restrict__ w)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N) return;
    for (int gq = 0; gq < G; ++gq) {
        float t[CHG];
#pragma unroll
        for (int c = 0; c < CHG; ++c) {
            const int ch = gq * CHG + c;
            t[c] = ch < C ? sigma_to_w(sigmas[(size_t)a * C + ch], w_scale) : mk_inf();
        }
        w[(size_t)(gq * 2 + 0) * N + a] = make_float4(t[0], t[1], t[2], t[3]);
        w[(size_t)(gq * 2 + 1) * N + a] = make_float4(t[4], t[5], t[6], t[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// Trajectory slab -> packed items (batch._stream_voxelize): frames of Molecule.coords
// ([rows = atoms x 3][frames], frame fastest, row pitch `src_pitch` floats) become frame-major items
// ([frames][rows]), scaled on the way (XTC stores nm).  A 64 x 64 tile through LDS (65-float pitch: conflict-free):
// read with lanes along frames, written with lanes along rows -- both sides whole 256-byte segments.  The same kernel
// turns the box lengths ([3][frames]) into [frames][3] (rows = 3).  One float32 multiply per value: torch's
// `x.mul_(scale)` bits.
// ------------------------------------------------------------------------------------------------
// Tiles are dealt to the 8 XCDs in CONTIGUOUS ranges of the destination's memory order (frame tile, then row tile; a 1-D grid
// padded to a multiple of 8: block b runs on XCD b & 7 as its (b >> 3)-th block) -- the lesson of the distance kernels
// (dist_kernels.h, xcd_contiguous_tile): neighbouring 256-byte pieces of a row written from different XCDs' L2s at different
// times reach the HBM at less than half the rate of pieces that leave ONE L2 together.
MK_KERNEL(256) void k_frames_to_items(const float* __restrict__ src, long long rows, long long src_pitch, long long nframes,
                                      float scale, float* __restrict__ dst)
{
    __shared__ float tile[64][65];
    const long long rtiles = (rows + 63) / 64, T = rtiles * ((nframes + 63) / 64), per_xcd = (T + 7) / 8;
    const long long gt = (long long)(blockIdx.x & 7u) * per_xcd + (long long)(blockIdx.x >> 3);
    if (gt >= T) return;                                             // (the whole block)
    const long long f0 = (gt / rtiles) * 64, r0 = (gt % rtiles) * 64;
    const int a = threadIdx.x & 63, b = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long long r = r0 + b + 4 * i, f = f0 + a;
        if (r < rows && f < nframes) tile[b + 4 * i][a] = src[r * src_pitch + f] * scale;
    }
    mk_block_sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long long f = f0 + b + 4 * i, r = r0 + a;
        if (f < nframes && r < rows) dst[f * rows + r] = tile[a][b + 4 * i];
    }
}

// ------------------------------------------------------------------------------------------------
// Lattice centres (voxeldescriptors.py:125-132 + :245-247): centre = fl64(index*res) + bb_min,
// x slowest / z fastest, float64 [V,3].  Same IEEE operations as numpy -> bit-exact.
// ------------------------------------------------------------------------------------------------
MK_KERNEL(256) void k_grid_centers(double ox, double oy, double oz, int nx, int ny, int nz,
                                   double res, double* __restrict__ centers)
{
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long V = (long long)nx * ny * nz;
    if (v >= V) return;
    const int iz = (int)(v % nz);
    const int iy = (int)((v / nz) % ny);
    const int ix = (int)(v / ((long long)nz * ny));
    // keep the two roundings of numpy's multiply-then-add (no FMA contraction)
    centers[3 * v + 0] = mk_dadd_rn(mk_dmul_rn((double)ix, res), ox);
    centers[3 * v + 1] = mk_dadd_rn(mk_dmul_rn((double)iy, res), oy);
    centers[3 * v + 2] = mk_dadd_rn(mk_dmul_rn((double)iz, res), oz);
}

}  // namespace mkamd
