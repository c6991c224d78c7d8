// dist_kernels.h -- HIP kernels for moleculekit/distance_utils/distance_utils.pyx on MI355X (gfx950)
// (dist_trajectory, dist_trajectory_reduction[_pairs], cdist, pdist; the
//  contact / collision lists and squareform are index work on top of these, moleculekit_amd/distance_utils.py).
//
// The reference computes in float32 with a fixed operation order, so these kernels are BIT-EXACT: every
// multiply / add / divide / sqrt is a separately rounded round-to-nearest op (mk_f*_rn, never contracted
// into an FMA), sums run in the reference's order, round() is half-away-from-zero.
//
// Layouts are the reference's: coords float32 [n_atoms, 3, n_frames] (frame fastest, Molecule.coords),
// box float32 [3, n_frames], results float32 [n_frames, n_pairs].  Frames are the coalescing axis of the
// inputs and pairs the one of the output, so the tile kernels compute a 64-frame x 64-pair tile with lanes
// along frames, transpose it through LDS (65-float pitch: conflict-free) and store with lanes along pairs;
// rectangular calls with long rows turn the few selected atoms frame-major instead and write rows directly
// (k_sel_to_frames + k_dist_rows below).
#pragma once
#ifndef MK_DEVICE_API_PROVIDED
#include "mk_device.h"
#endif

namespace mkamd {

// d - b * round(d / b) (distance_utils.pyx:49-51) without the division where that is provably the same: what is needed of
// fl(d / b) is only WHICH integer it rounds to.  q = fl(d * fl(1 / b)) is within 3 x 2^-24 |q| of fl(d / b); unless q sits
// that close to a half-integer (where round() changes its value) both round to the same integer -- and away from the
// half-integers round-half-away (C round) and round-half-even (ONE instruction, v_rndne_f32) agree as well.  With
// r = rndne(q) the distance of q from the nearest half-integer is 0.5 - |q - r| (the subtraction is exact).  The three
// axes share ONE test: the largest |q - r| against the bound of the largest |q| (two v_max3_f32, an fma, a compare).
// Pairs that fail it -- and everything that is not an ordinary number: a zero box makes q infinite, the bound -inf and
// the comparison false -- take the correctly rounded divisions.  (A NaN q is ignored by the maxima; then r is NaN and so
// is the reference's result: 0 / 0, inf / inf or a NaN operand.)
// `ib` = fl(1 / b), computed once per (lane, frame) instead of three IEEE divisions per pair.
// (Round 5 tried the test on the SHIFTED separation instead -- |d - b r| < b (1/2 - 2^-12) per axis and |q| < 512: three compares
//  against per-frame constants, a max3 and a compare where this takes seven instructions; same results, emulator and GPU tests
//  green -- and every periodic kernel ran 10-15 % SLOWER in a same-session probe (200 x 500: 245 us against 213; the open calls
//  beside them unchanged): three compares into scalar pairs and their s_and chain wait on each other, the max3 / fma form does not.
//  Also tried: the shift as ONE fma per axis where |r| <= 2 on every axis -- b r is exact then, so d - b r rounds once either way --
//  for one more compare (`qm < 2.5`) on the test: 3 instructions fewer of ~27, bit-identical (GPU tests green), and within +-1.5 % of
//  this form on every periodic shape of tools/dist_shapes_probe.py, three alternating runs each: not adopted.)
MK_DEV float round_quotient_exact(float d, float b) { return roundf(mk_fdiv_rn(d, b)); }

// distance_utils.pyx:34-54 (_dist) / :188-206 (_dist2)
MK_DEV float dist2_min_image_f32(float x1, float y1, float z1, float x2, float y2, float z2,
                                 float bx, float by, float bz, float ibx, float iby, float ibz, bool wrap)
{
    float dx = mk_fsub_rn(x1, x2), dy = mk_fsub_rn(y1, y2), dz = mk_fsub_rn(z1, z2);
    if (wrap) {
        const float qx = mk_fmul_rn(dx, ibx), qy = mk_fmul_rn(dy, iby), qz = mk_fmul_rn(dz, ibz);
        float rx = mk_rint(qx), ry = mk_rint(qy), rz = mk_rint(qz);
        const float tm = mk_max3(fabsf(qx - rx), fabsf(qy - ry), fabsf(qz - rz));
        const float qm = mk_max3(fabsf(qx), fabsf(qy), fabsf(qz));
        if (!(tm < mk_fma(-3e-7f, qm, 0.5f))) {
            asm volatile("" ::: "memory");                           // a real branch: the divisions must not be computed "just in case"
            rx = round_quotient_exact(dx, bx); ry = round_quotient_exact(dy, by); rz = round_quotient_exact(dz, bz);
        }
        dx = mk_fsub_rn(dx, mk_fmul_rn(bx, rx));
        dy = mk_fsub_rn(dy, mk_fmul_rn(by, ry));
        dz = mk_fsub_rn(dz, mk_fmul_rn(bz, rz));
    }
    return mk_fadd_rn(mk_fadd_rn(mk_fmul_rn(dx, dx), mk_fmul_rn(dy, dy)), mk_fmul_rn(dz, dz));
}

// Two pairs that share their first atom and take no image shift, in packed arithmetic (v_pk_add_f32 / v_pk_mul_f32: two float32
// operations per lane and instruction, each rounded on its own like the scalar ones): the row kernel's non-periodic walk,
// 0.198-0.205 -> 0.193 ms.  (With the image shift packed as well the periodic walk got SLOWER, 0.241-0.247 -> 0.265-0.269 ms:
// rounding, the test and the per-pair selects work on single components, and moving them in and out of register pairs cost
// more than the packed multiplies saved.)
MK_DEV mk_f2 dist2_x2(float x1, float y1, float z1, mk_f2 x2, mk_f2 y2, mk_f2 z2)
{
    const mk_f2 dx = mk_f2_sub_rn(mk_f2_splat(x1), x2), dy = mk_f2_sub_rn(mk_f2_splat(y1), y2), dz = mk_f2_sub_rn(mk_f2_splat(z1), z2);
    return mk_f2_add_rn(mk_f2_add_rn(mk_f2_mul_rn(dx, dx), mk_f2_mul_rn(dy, dy)), mk_f2_mul_rn(dz, dz));
}

// The image integers of SEVERAL pairs behind one wave-uniform test (round 6; the group reductions first, then the pair-table walk):
// risk = max over the pairs of (largest |q - rndne(q)| + 3e-7 largest |q|), accumulated without a branch; a stretch whose risk
// reaches DRC_RISK in any lane is computed once more pair by pair (dist2_min_image_f32: the per-pair test, the divisions).
constexpr float DRC_RISK = 0.4999998f;               // risk below this: every rndne(d * fl(1/b)) is the reference's round(d / b)
                                                     // (the per-pair test is tm < 0.5 - 3e-7 qm, proven bound 1.8e-7 qm; here
                                                     //  tm + 3e-7 qm is rounded once more: 2e-7 of slack)

// d^2 of two atom pairs (first atoms ax/ay/az[0..1], second atom (x2, y2, z2)) -- distance_utils.pyx:188-206
template <bool WR>
MK_DEV mk_f2 dist2_pk(mk_f2 ax, mk_f2 ay, mk_f2 az, float x2, float y2, float z2, float bx, float by, float bz,
                      float ibx, float iby, float ibz, float& risk)
{
    mk_f2 dx = mk_f2_sub_rn(ax, mk_f2_splat(x2)), dy = mk_f2_sub_rn(ay, mk_f2_splat(y2)), dz = mk_f2_sub_rn(az, mk_f2_splat(z2));
    if constexpr (WR) {
        const mk_f2 qx = mk_f2_mul_rn(dx, mk_f2_splat(ibx)), qy = mk_f2_mul_rn(dy, mk_f2_splat(iby)), qz = mk_f2_mul_rn(dz, mk_f2_splat(ibz));
        const mk_f2 rx = mk_f2{mk_rint(qx[0]), mk_rint(qx[1])}, ry = mk_f2{mk_rint(qy[0]), mk_rint(qy[1])}, rz = mk_f2{mk_rint(qz[0]), mk_rint(qz[1])};
        const mk_f2 tx = mk_f2_sub_rn(qx, rx), ty = mk_f2_sub_rn(qy, ry), tz = mk_f2_sub_rn(qz, rz);       // (exact)
        const float s0 = mk_fma(3e-7f, mk_max3_abs_raw(qx[0], qy[0], qz[0]), mk_max3_abs_raw(tx[0], ty[0], tz[0]));
        const float s1 = mk_fma(3e-7f, mk_max3_abs_raw(qx[1], qy[1], qz[1]), mk_max3_abs_raw(tx[1], ty[1], tz[1]));
        risk = mk_max3_raw(risk, s0, s1);
        dx = mk_f2_sub_rn(dx, mk_f2_mul_rn(mk_f2_splat(bx), rx));
        dy = mk_f2_sub_rn(dy, mk_f2_mul_rn(mk_f2_splat(by), ry));
        dz = mk_f2_sub_rn(dz, mk_f2_mul_rn(mk_f2_splat(bz), rz));
    }
    return mk_f2_add_rn(mk_f2_add_rn(mk_f2_mul_rn(dx, dx), mk_f2_mul_rn(dy, dy)), mk_f2_mul_rn(dz, dz));
}

// Self-test of mk_fsqrt_rn_ordinary (mk_device.h: the short correctly-rounded root) against the provable form, over the float bit
// patterns [lo, lo + n): mismatches counted, the first one kept (mkamd_selftest_sqrt).
MK_KERNEL(256) void k_selftest_sqrt(unsigned lo, unsigned long long n, unsigned long long* __restrict__ bad, unsigned* __restrict__ first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0ull;
    unsigned where = 0u;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const unsigned bits = lo + (unsigned)k;
        const float x = mk_uint_as_float(bits);
        if (mk_float_bits(mk_fsqrt_rn_ordinary(x)) != mk_float_bits(mk_fsqrt_rn_tuckerman(x))) { if (!mine) where = bits; ++mine; }
    }
    if (mine) { if (mk_atomic_add64(bad, mine) == 0ull) *first = where; }
}

constexpr int DT = 64;                 // tile edge (frames and pairs)

// Which tile a workgroup of a 1-D launch of 8 * ceil(T / 8) blocks takes, T tiles numbered frame slab-major (all pair tiles of
// slab 0, then slab 1, ...): the dispatcher deals consecutive blocks round-robin to the 8 XCDs, so block b runs on XCD b & 7 as
// its (b >> 3)-th block; XCD c is given the CONTIGUOUS tile range [c * ceil(T / 8), ...).  Every XCD then works through its
// own frame slabs: the 256-byte row segments its blocks write one after the other are neighbours in memory and leave its L2
// as long runs of the same DRAM pages (round 4: dealt tile by tile to all XCDs the result was written at 2.5 TB/s -- a
// store-only build of the kernel took the same 0.32 ms as the real one -- against 6.9 TB/s for a linear fill), and it reads only
// its own slabs' coordinates.  Placement is a matter of speed only: any block -> XCD rule gives the same result.
MK_DEV long long xcd_contiguous_tile(long long T)
{
    const long long per_xcd = (T + 7) / 8;
    const long long g = (long long)(blockIdx.x & 7u) * per_xcd + (long long)(blockIdx.x >> 3);
    return g < T ? g : -1;
}
constexpr int DT_THREADS = 256;

// The second half of every tile kernel here: tile[pair][frame] (lanes ran along frames) goes out with lanes along pairs.
// A wave owns every fourth frame row; a full tile (the common case, block-uniform) reads its 16 values from LDS at
// once and stores them behind one another, an edge tile checks every element.
template <int NW = DT_THREADS / DT /* waves of the block */>
MK_DEV void store_tile_rows(const float (&tile)[DT][DT + 1], long long f0, long long p0, long long F, long long p_end /* pairs >= this are not stored */,
                            long long P /* row pitch of `out` */, float* __restrict__ out)
{
    constexpr int ROWS = DT / NW;
    const int pl = threadIdx.x & (DT - 1), fq = threadIdx.x >> 6;
    float* __restrict__ o = out + (size_t)(f0 + fq) * (size_t)P + (size_t)(p0 + pl);
    const size_t step = (size_t)NW * (size_t)P;
    if (f0 + DT <= F && p0 + DT <= p_end) {
        // a full tile: FOUR store instructions of 16 bytes per lane instead of sixteen of 4, whatever the alignment of its rows
        // (mk_store_f4_dword_aligned: the triangular list of 450 atoms has 101 025 pairs per row -- an odd pitch; any float* is
        // accepted as the result, an offset view is 4-byte aligned only) (the
        // memory pipeline takes a wave's store instructions one by one: round-4 PMC showed the kernel at the same 0.30 ms with
        // and without its image arithmetic and with a third of its loads).  A lane owns four consecutive pairs of one frame;
        // the lanes are dealt so that the 32 lanes the LDS serves together read 32 different banks:
        // lane -> (pair quad q = lane & 7 | (lane >> 5) << 3, frame r = (lane >> 3) & 3 of a group of four).
        const int lane = threadIdx.x & (DT - 1);
        const int q = (lane & 7) | ((lane >> 5) << 3), r = (lane >> 3) & 3;
        constexpr int NIT = DT / 4 / NW;                             // groups of four frames per wave
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int fg = (fq + NW * it) * 4 + r;
            v[it] = make_float4(tile[4 * q][fg], tile[4 * q + 1][fg], tile[4 * q + 2][fg], tile[4 * q + 3][fg]);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int fg = (fq + NW * it) * 4 + r;
            mk_store_f4_dword_aligned(out + (size_t)(f0 + fg) * (size_t)P + (size_t)(p0 + 4 * q), v[it]);
        }
    } else {
        for (int i = 0; i < ROWS; ++i)
            if (f0 + fq + i * NW < F && p0 + pl < p_end) o[(size_t)i * step] = tile[pl][fq + i * NW];
    }
}

// Pair table of dist_trajectory (distance_utils.pyx:144-155): loop order i over sel1, j over sel2 from
// (selfdist ? i+1 : 0).  One thread per (i, j); wrap = pbc && chains differ (:49).
MK_KERNEL(256) void k_build_atom_pairs(const unsigned* __restrict__ sel1, long long n1,
                                       const unsigned* __restrict__ sel2, long long n2,
                                       const unsigned* __restrict__ chains, int selfdist, int pbc,
                                       unsigned* __restrict__ pa, unsigned* __restrict__ pb,
                                       unsigned* __restrict__ wrap)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n2) return;
    for (long long i = blockIdx.y; i < n1; i += gridDim.y) {        // rows beyond the 65535 y-blocks: grid stride
        long long idx;
        if (selfdist) {
            if (j <= i) continue;
            // rows 0..i-1 hold max(n2-1-k, 0) entries each
            const long long full = i < n2 ? i : n2;                 // rows with a positive count
            idx = full * (n2 - 1) - full * (full - 1) / 2 + (j - i - 1);
        } else {
            idx = i * n2 + j;
        }
        const unsigned a = sel1[i], b = sel2[j];
        pa[idx] = a; pb[idx] = b;
        wrap[idx] = (pbc && chains[a] != chains[b]) ? 1u : 0u;
    }
}

// dist_trajectory (distance_utils.pyx:126-155): results[f, p] = sqrt(_dist(...)) (or the square).
// Lanes run along frames, so the atom indices of a pair are wave-uniform: pairs are visited in the reference's
// i-major order and the first atom's coordinates (and the frame's box) stay in registers while i does not
// change -- 3 instead of 6 coordinate loads per distance.
// A wave takes DP_RUN consecutive pairs of the tile.  Their (a, b, wrap) come from ONE load each (lane k holds pair
// k, handed out with readlane), and the coordinates of DP_BATCH pairs are requested together before the first of them
// is used: the first version looked its pair up, waited, loaded its three coordinates, waited -- 32 dependent round
// trips to L2 per wave, which is what bounded it (0.56 ms for 100 000 pairs x 2 048 frames: 1.5 TB/s of stores).
constexpr int DP_RUN = DT / (DT_THREADS / DT);   // 16
template <bool B> struct DistFlag { static constexpr bool value = B; };
constexpr int DP_BATCH = 4;           // (8 was measured too)

// d^2 of the DP_RUN consecutive pairs [pw, pw + DP_RUN) for this lane's frame (byte offset fb = 4 f into a coordinate
// row; the host refuses F >= 2^30): emit(k, d2), k = 0 .. DP_RUN-1, called by all lanes.  Pairs past the end repeat
// the last pair (valid addresses, no divergence) -- the caller drops what they emit.  Needs pw < P.
// emit(k0, d2[DP_BATCH]): a batch at a time.
template <bool MAYWRAP = true /* false: a call with pbc = 0 -- no pair wraps, the flags are not even looked at */, class Emit>
MK_DEV void for_pair_run(const float* __restrict__ coords, long long F, unsigned fb, float bx, float by, float bz,
                         const unsigned* __restrict__ pa, const unsigned* __restrict__ pb, const unsigned* __restrict__ wrap,
                         long long P, long long pw, Emit&& emit)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const long long left = P - pw;                                   // wave-uniform, > 0
    const long long pi = pw + (lane & (DP_RUN - 1)) < P ? pw + (lane & (DP_RUN - 1)) : P - 1;
    // lane k: first atom of pair k, and its second atom with the wrap flag in bit 31 (atom indices are int32)
    unsigned vb_ = pb[pi];
    if constexpr (MAYWRAP) vb_ |= wrap[pi] != 0u ? 0x80000000u : 0u;
    const unsigned va = pa[pi], vb = vb_;
    const unsigned a_first = mk_readlane(va, 0);
    const bool one_a = mk_ballot(va != a_first) == 0ull;              // the usual case in i-major order: one first atom for the run
    // a coordinate row is a wave-uniform base (the atom index sits in a scalar register) plus this lane's frame as a
    // 32-bit byte offset: the addresses cost no vector instructions.  When every row this run touches ends below 4 GiB from
    // the start of the array (wave-uniform; a 30 000-atom, 2 048-frame trajectory is 0.7 GiB) the row is a 32-bit scalar
    // offset from ONE descriptor (mk_load_f32_base_soffset), else a 64-bit base and a descriptor per row.
    const unsigned F4 = (unsigned)F * 4u;                             // (F < 2^30: the host checks)
    const unsigned hi_atom = va > (vb & 0x7fffffffu) ? va : (vb & 0x7fffffffu);
    const bool small_rows = mk_ballot(((unsigned long long)hi_atom * 3ull + 3ull) * (unsigned long long)F4 > 0xffffffffull) == 0ull;
    // Does ANY pair of this run wrap (pbc and different chains, distance_utils.pyx:49)?  Wave-uniform.  A run without one --
    // every call with pbc = False, the common MetricDistance call, and every intra-chain stretch of a periodic one -- takes
    // a body that holds no image arithmetic at all; the reciprocals of the box are only formed (three IEEE divisions: 30
    // vector instructions) for runs that wrap, ONCE per run: left to itself the optimizer sinks them into every batch
    // (round 4: 99 v_rcp_f32 in the kernel, 7.5 instructions per distance).
    const bool run_wraps = MAYWRAP && mk_ballot((vb & 0x80000000u) != 0u) != 0ull;
    float ibx = 0.f, iby = 0.f, ibz = 0.f;
    if (run_wraps) { ibx = mk_fdiv_rn(1.f, bx); iby = mk_fdiv_rn(1.f, by); ibz = mk_fdiv_rn(1.f, bz); }
    mk_keep(ibx); mk_keep(iby); mk_keep(ibz);
    unsigned cur_a = 0xffffffffu;
    float xa = 0.f, ya = 0.f, za = 0.f;
    auto run_body = [&](auto wraps_, auto small_) {
    constexpr bool WR = decltype(wraps_)::value, SMALL = decltype(small_)::value;
    auto at = [&](unsigned atom, int ax) {
        if constexpr (SMALL) return mk_load_f32_base_soffset(coords, (atom * 3u + (unsigned)ax) * F4, fb);
        else return mk_load_f32_uniform_base(coords + ((size_t)atom * 3 + (size_t)ax) * (size_t)F, fb);
    };
#pragma unroll
    for (int k0 = 0; k0 < DP_RUN; k0 += DP_BATCH) {
        if ((long long)k0 >= left) break;                            // wave-uniform
        unsigned a[DP_BATCH], b[DP_BATCH], w[DP_BATCH];
        bool same = true;                                            // wave-uniform: the batch stays with the cached first atom
#pragma unroll
        for (int u = 0; u < DP_BATCH; ++u) {
            a[u] = one_a ? a_first : mk_readlane(va, k0 + u);
            const unsigned bw = mk_readlane(vb, k0 + u);
            b[u] = bw & 0x7fffffffu; w[u] = bw >> 31;
            same &= a[u] == cur_a;
        }
        float B3[DP_BATCH][3], d2[DP_BATCH];
#pragma unroll
        for (int u = 0; u < DP_BATCH; ++u) { B3[u][0] = at(b[u], 0); B3[u][1] = at(b[u], 1); B3[u][2] = at(b[u], 2); }
        if (!same) {
            // a batch with ONE first atom that is not the cached one (the first batch of a run, the first after a row of the pair list
            // ended): that atom is fetched once and the batch goes the way of the cached ones (round 6: it used to load it four times)
            bool one = true;
#pragma unroll
            for (int u = 1; u < DP_BATCH; ++u) one &= a[u] == a[0];
            if (one) { cur_a = a[0]; xa = at(a[0], 0); ya = at(a[0], 1); za = at(a[0], 2); same = true; }
        }
        if (same) {
            // (round 6) a batch whose four pairs ALL wrap (wave-uniform: the flags sit in scalar registers) in packed arithmetic, two
            // pairs per instruction, their image integers behind ONE accumulated test (dist2_pk; b - a instead of a - b: every
            // operation on the way to d^2 is odd or even in the separation, rndne included).  A batch that fails it in any lane
            // -- a separation within 3e-7 of half a box length, an infinite quotient -- and mixed batches go pair by pair.
            bool packed = false;
            if constexpr (WR) {
                static_assert(DP_BATCH == 4, "two packed pairs of second atoms");
                if ((w[0] & w[1] & w[2] & w[3]) != 0u) {
                    float risk = 0.f;
                    const mk_f2 e01 = dist2_pk<true>(mk_f2{B3[0][0], B3[1][0]}, mk_f2{B3[0][1], B3[1][1]}, mk_f2{B3[0][2], B3[1][2]}, xa, ya, za,
                                                     bx, by, bz, ibx, iby, ibz, risk);
                    const mk_f2 e23 = dist2_pk<true>(mk_f2{B3[2][0], B3[3][0]}, mk_f2{B3[2][1], B3[3][1]}, mk_f2{B3[2][2], B3[3][2]}, xa, ya, za,
                                                     bx, by, bz, ibx, iby, ibz, risk);
                    packed = mk_ballot(!(risk < DRC_RISK)) == 0ull;
                    d2[0] = e01[0]; d2[1] = e01[1]; d2[2] = e23[0]; d2[3] = e23[1];
                }
            }
            if (!packed) {
                if constexpr (WR) mk_stay_in_branch();
#pragma unroll
                for (int u = 0; u < DP_BATCH; ++u)
                    d2[u] = dist2_min_image_f32(xa, ya, za, B3[u][0], B3[u][1], B3[u][2], bx, by, bz, ibx, iby, ibz, WR && w[u] != 0u);
            }
        } else {
            float A3[DP_BATCH][3];
#pragma unroll
            for (int u = 0; u < DP_BATCH; ++u) { A3[u][0] = at(a[u], 0); A3[u][1] = at(a[u], 1); A3[u][2] = at(a[u], 2); }
#pragma unroll
            for (int u = 0; u < DP_BATCH; ++u)
                d2[u] = dist2_min_image_f32(A3[u][0], A3[u][1], A3[u][2], B3[u][0], B3[u][1], B3[u][2], bx, by, bz, ibx, iby, ibz, WR && w[u] != 0u);
            cur_a = a[DP_BATCH - 1];
            xa = A3[DP_BATCH - 1][0]; ya = A3[DP_BATCH - 1][1]; za = A3[DP_BATCH - 1][2];
        }
        emit(k0, d2);                                                // the batch's DP_BATCH values at once: emit(first k, d2[DP_BATCH])
    }
    };
    if (small_rows) { if (run_wraps) run_body(DistFlag<true>{}, DistFlag<true>{}); else run_body(DistFlag<false>{}, DistFlag<true>{}); }
    else { if (run_wraps) run_body(DistFlag<true>{}, DistFlag<false>{}); else run_body(DistFlag<false>{}, DistFlag<false>{}); }
}

// (Round 5 tried the kernel WITHOUT the turn through LDS: a lane stores the four roots of a batch -- four consecutive pairs of its
//  frame -- as 16 bytes of out[f, p ..], a store instruction touches 64 rows whose lines fill up over the wave's four batches.  No
//  LDS, no barrier, the same bits -- and 35-55 % SLOWER on every triangular shape (450 x 450: 400 us against 292, 2.1 TB/s; 100 x 100:
//  23.8 against 19.7): sixty-four quarter-lines per store instruction are what the memory pipeline is slowest at.)
// (round 6: PBC is a template parameter -- the periodic and the open call are different kernels to a profiler, as the row and
//  frame kernels' are: `mkamd::k_dist_pairs<true>` / `<false>`; the open one never reads the wrap flags)
template <bool PBC>
MK_KERNEL(DT_THREADS) void k_dist_pairs(const float* __restrict__ coords, long long F,
                                        const float* __restrict__ box, const unsigned* __restrict__ pa,
                                        const unsigned* __restrict__ pb, const unsigned* __restrict__ wrap,
                                        long long P, int squared, float* __restrict__ out)
{
    __shared__ float tile[DT][DT + 1];
    const long long ptiles = (P + DT - 1) / DT, g = xcd_contiguous_tile(ptiles * ((F + DT - 1) / DT));
    if (g < 0) return;                                               // (the whole block: the grid is padded to a multiple of 8)
    const long long f0 = (g / ptiles) * DT, p0 = (g % ptiles) * DT;
    {
        const int fl = threadIdx.x & (DT - 1), pq = threadIdx.x >> 6;
        // frames past the end compute on the last frame (the store phase never reads those tile entries)
        const long long f = f0 + fl < F ? f0 + fl : F - 1;
        const float bx = box[0 * F + f], by = box[1 * F + f], bz = box[2 * F + f];
        const long long pw = p0 + pq * DP_RUN;                       // the wave's first pair
        if (pw < P)
            for_pair_run<PBC>(coords, F, (unsigned)f * 4u, bx, by, bz, pa, pb, wrap, P, pw,
                         [&](int k0, const float (&d2)[DP_BATCH]) {
                             // the batch's roots behind ONE wave-uniform test (mk_sqrt_ordinary: practically always true)
                             const bool ordinary = mk_sqrt_ordinary_all(d2);
                             if (squared) {
#pragma unroll
                                 for (int u = 0; u < DP_BATCH; ++u) tile[pq * DP_RUN + k0 + u][fl] = d2[u];
                             } else if (mk_ballot(!ordinary) == 0ull) {
#pragma unroll
                                 for (int u = 0; u < DP_BATCH; ++u) tile[pq * DP_RUN + k0 + u][fl] = mk_fsqrt_rn_ordinary(d2[u]);
                             } else {
#pragma unroll
                                 for (int u = 0; u < DP_BATCH; ++u) tile[pq * DP_RUN + k0 + u][fl] = mk_fsqrt_rn(d2[u]);
                             }
                         });
    }
    mk_block_sync();
    store_tile_rows(tile, f0, p0, F, P, P, out);
}

// ------------------------------------------------------------------------------------------------
// dist_trajectory WITHOUT selfdist (every sel1 atom against every sel2 atom: pair p = i * n2 + j, distance_utils.pyx:144-155
// with j from 0) -- the common MetricDistance call -- has a rectangular pair table, and k_dist_pairs wastes it: a tile of 64
// consecutive pairs shares its FIRST atom and loads 64 second atoms x 3 rows per tile, 12 bytes from the L2 for every 4 bytes
// it stores (2.6 GB per 0.8 GB result on the bench workload; round-4 PMC: with and without the image arithmetic the kernel took
// the same 0.30 ms -- 92 % and 58 % VALU-busy).  Here a block owns 64 consecutive sel2 atoms x DR_I consecutive sel1 atoms x 64
// frames: a wave keeps the coordinates of ITS 16 second atoms (lane = frame) in 48 registers and walks the first atoms --
// three loads per first atom instead of 51 -- transposing one 64-pair row of the result through LDS per first atom, so the
// stores are k_dist_pairs' (whole 256-byte rows of out[f, i * n2 + j0 ..]).  No pair table is built.  Round 6: the second atoms sit in
// registers as packed pairs and a row's pairs go through dist2_pk behind one accumulated image test (dist2_min_image_f32 where it fails):
// the same bits; no faster -- the kernel is bound by its turn through LDS, not by its arithmetic (profiles/r6_dist_rect_packed_ab.txt).
// ------------------------------------------------------------------------------------------------
constexpr int DR_I = 8;                // first atoms per block (the second atoms' loads are amortised over them)
constexpr int DR_WAVES = 8;            // waves per block: 8 second atoms each -- 24 registers of coordinates, not 48 (sixteen per wave: 129-141
                                       // VGPRs, three waves per SIMD)
constexpr int DR_PW = DT / DR_WAVES;   // second atoms (pairs of a row) per wave

template <bool PBC, bool SMALL>
MK_DEV void dist_rect_block(const float* __restrict__ coords, long long F, const float* __restrict__ box,
                            const unsigned* __restrict__ sel1, long long n1, const unsigned* __restrict__ sel2, long long n2,
                            const unsigned* __restrict__ chains, int squared, float* __restrict__ out, float (&tiles)[2][DT][DT + 1],
                            long long tj /* tile of second atoms */, long long ti /* group of first atoms */, long long tf /* frame slab */)
{
    const int fl = threadIdx.x & (DT - 1), wq = threadIdx.x >> 6;
    const long long f0 = tf * DT, j0 = tj * DT, i0 = ti * DR_I;
    const long long f = f0 + fl < F ? f0 + fl : F - 1;              // frames past the end compute on the last one (never stored)
    const unsigned fb = (unsigned)f * 4u, F4 = (unsigned)F * 4u;
    auto at = [&](unsigned atom, int ax) {
        if constexpr (SMALL) return mk_load_f32_base_soffset(coords, (atom * 3u + (unsigned)ax) * F4, fb);
        else return mk_load_f32_uniform_base(coords + ((size_t)atom * 3 + (size_t)ax) * (size_t)F, fb);
    };
    // this wave's 16 second atoms (past the end: the last one again -- computed, never stored); lane k holds atom k and its chain
    const long long jw = j0 + wq * DR_PW;
    const long long jk = jw + (fl & (DR_PW - 1)) < n2 ? jw + (fl & (DR_PW - 1)) : n2 - 1;
    const unsigned vb = sel2[jk], vcb = PBC ? chains[vb] : 0u;
    // (round 6) the wave's second atoms as packed PAIRS: two pairs of the result per instruction (dist2_pk), the image integers of a
    // row's DR_PW pairs behind one accumulated test
    constexpr int H = DR_PW / 2;
    mk_f2 BX[H], BY[H], BZ[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const unsigned b0 = mk_readlane(vb, 2 * h), b1 = mk_readlane(vb, 2 * h + 1);
        BX[h] = mk_f2{at(b0, 0), at(b1, 0)}; BY[h] = mk_f2{at(b0, 1), at(b1, 1)}; BZ[h] = mk_f2{at(b0, 2), at(b1, 2)};
    }
    float bx = 0.f, by = 0.f, bz = 0.f, ibx = 0.f, iby = 0.f, ibz = 0.f;
    if (PBC) {
        bx = box[0 * F + f]; by = box[1 * F + f]; bz = box[2 * F + f];
        ibx = mk_fdiv_rn(1.f, bx); iby = mk_fdiv_rn(1.f, by); ibz = mk_fdiv_rn(1.f, bz);
    }
    const long long P = n1 * n2;
    const long long ni = n1 - i0 < DR_I ? n1 - i0 : DR_I;            // block-uniform, >= 1
    unsigned a = sel1[i0];
    float xa = at(a, 0), ya = at(a, 1), za = at(a, 2);
    for (long long ii = 0; ii < ni; ++ii) {
        // two tiles alternate: the rows of first atom ii leave tile ii & 1 while ii + 1 is computed into the other one -- ONE
        // barrier per first atom (what orders the reads of tile ii & 1 before its next writes is the barrier of ii + 1)
        float (&tile)[DT][DT + 1] = tiles[ii & 1];
        // the next first atom's coordinates are requested before this one's distances are computed
        const unsigned a_next = sel1[ii + 1 < ni ? i0 + ii + 1 : i0 + ii];
        const float xn = at(a_next, 0), yn = at(a_next, 1), zn = at(a_next, 2);
        // which of the wave's second atoms wrap against this first atom (distance_utils.pyx:49): wave-uniform bits
        unsigned wm = 0u;
        if (PBC) { const unsigned ca = chains[a]; wm = (unsigned)(mk_ballot(vcb != ca) & ((1ull << DR_PW) - 1ull)); }
        float d2[DR_PW];
        bool redo = false;
        float none = 0.f;
        if (PBC && wm == (1u << DR_PW) - 1u) {
            float risk = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const mk_f2 e = dist2_pk<true>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, risk);   // (b - a: d^2 is even in the separation)
                d2[2 * h] = e[0]; d2[2 * h + 1] = e[1];
            }
            redo = mk_ballot(!(risk < DRC_RISK)) != 0ull;            // an image integer may differ from round(d / b) (rare)
        } else if (PBC && wm != 0u) {
            // only some of the row's pairs wrap: both forms, chosen per pair (wave-uniform bits: no branch)
            float risk = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const mk_f2 w2 = dist2_pk<true>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, risk);
                const mk_f2 o2 = dist2_pk<false>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, none);
                d2[2 * h] = ((wm >> (2 * h)) & 1u) ? w2[0] : o2[0]; d2[2 * h + 1] = ((wm >> (2 * h + 1)) & 1u) ? w2[1] : o2[1];
            }
            redo = mk_ballot(!(risk < DRC_RISK)) != 0ull;
        } else {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const mk_f2 e = dist2_pk<false>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, none);
                d2[2 * h] = e[0]; d2[2 * h + 1] = e[1];
            }
        }
        if (redo) {
            // pair by pair with the per-pair test and the correctly rounded divisions behind it, the second atoms loaded again (a loop:
            // sixteen inlined copies of the division path cost the walk its registers), straight into the tile
            mk_stay_in_branch();
#pragma unroll 1
            for (int k = 0; k < DR_PW; ++k) {
                const unsigned bk = mk_readlane(vb, k);
                const float d = dist2_min_image_f32(xa, ya, za, at(bk, 0), at(bk, 1), at(bk, 2), bx, by, bz, ibx, iby, ibz, ((wm >> k) & 1u) != 0u);
                tile[wq * DR_PW + k][fl] = squared ? d : mk_fsqrt_rn(d);
            }
        } else {
            const bool ordinary = mk_sqrt_ordinary_all(d2);
            // the roots of the batch behind ONE wave-uniform test (all values ordinary numbers: practically always)
            if (squared) {
#pragma unroll
                for (int k = 0; k < DR_PW; ++k) tile[wq * DR_PW + k][fl] = d2[k];
            } else if (mk_ballot(!ordinary) == 0ull) {
#pragma unroll
                for (int k = 0; k < DR_PW; ++k) tile[wq * DR_PW + k][fl] = mk_fsqrt_rn_ordinary(d2[k]);
            } else {
#pragma unroll
                for (int k = 0; k < DR_PW; ++k) tile[wq * DR_PW + k][fl] = mk_fsqrt_rn(d2[k]);
            }
        }
        mk_block_sync();
        // the row of 64 pairs (i, j0 .. j0 + 63) of every frame of the slab (pairs past the row's end -- j >= n2 -- belong to the
        // next first atom: the row is clipped at its own end, not at P)
        store_tile_rows<DR_WAVES>(tile, f0, (i0 + ii) * n2 + j0, F, (i0 + ii) * n2 + (j0 + DT <= n2 ? j0 + DT : n2), P, out);
        a = a_next; xa = xn; ya = yn; za = zn;
    }
}

template <bool PBC>
MK_KERNEL(DR_WAVES * WAVE) void k_dist_rect(const float* __restrict__ coords, long long F, const float* __restrict__ box,
                                       const unsigned* __restrict__ sel1, long long n1, const unsigned* __restrict__ sel2, long long n2,
                                       const unsigned* __restrict__ chains, int squared, float* __restrict__ out)
{
    __shared__ float tile[2][DT][DT + 1];
    // every row this BLOCK touches ends below 4 GiB from the start of the array?  Block-uniform: every wave looks at all 64
    // second atoms of the tile and at the block's first atoms (lane l: sel2[j0 + l] and sel1[i0 + l mod DR_I])
    // tiles in the order of the result's memory: slab, then group of first atoms, then tile of second atoms (p = i * n2 + j)
    const long long gx = (n2 + DT - 1) / DT, gy = (n1 + DR_I - 1) / DR_I;
    const long long g = xcd_contiguous_tile(gx * gy * ((F + DT - 1) / DT));
    if (g < 0) return;
    const long long bx = g % gx, by = (g / gx) % gy, bz = g / (gx * gy);
    const int l = threadIdx.x & (DT - 1);
    const long long jl = bx * DT + l, il = by * DR_I + (l & (DR_I - 1));
    const unsigned hb = sel2[jl < n2 ? jl : n2 - 1], ha = sel1[il < n1 ? il : n1 - 1];
    const unsigned hi_atom = hb > ha ? hb : ha;
    const bool small_rows = mk_ballot(((unsigned long long)hi_atom * 3ull + 3ull) * ((unsigned long long)F * 4ull) > 0xffffffffull) == 0ull;
    if (small_rows) dist_rect_block<PBC, true>(coords, F, box, sel1, n1, sel2, n2, chains, squared, out, tile, bx, by, bz);
    else dist_rect_block<PBC, false>(coords, F, box, sel1, n1, sel2, n2, chains, squared, out, tile, bx, by, bz);
}

// ------------------------------------------------------------------------------------------------
// The same rectangular call once more, for rows of the result that are long enough to be written directly (round 4, second
// half).  k_dist_rect reads the coordinates along frames (their fast axis) and must turn every 64 x 64 tile through LDS to
// write along pairs: a barrier per first atom, two LDS accesses per distance, and phases of loading, computing and storing
// that overlap only through other blocks (measured: 0.24-0.28 ms where a store-only build takes 0.157).  But the atoms of
// the two selections are few (n1 + n2 against n1 * n2 pairs): turning THEM frame-major first costs next to nothing
// (k_sel_to_frames: [F][3][n] per selection, 17 MB where the result has 819), and then a WAVE owns a frame: a lane keeps ROWS_J
// second atoms in registers, the first atom of the moment comes in through scalar loads, and the wave writes 256 contiguous
// bytes of out[f, i * n2 + j..] per store -- no LDS, no barrier, nothing but the pair arithmetic between a load and a store.
// Same functions per pair (dist2_min_image_f32, mk_fsqrt_rn): the same bits.
// ------------------------------------------------------------------------------------------------
constexpr int ROWS_CI = 16;    // first atoms a wave walks for its second atoms (their loads are amortised over them)

// The two selections' coordinates, frame-major: T[f][ax][k] = coords[sel[k], ax, f], k < np (np = n rounded up: the pad repeats the
// last atom, so that idle lanes compute on something harmless); cs[k] = chains[sel[k]].  ONE launch for both selections and all
// three axes -- a block turns 64 frames x 64 atoms of one axis through LDS: (F / 64) x (np1 / 64 + np2 / 64) x 3 blocks (as two
// launches of 64 x 64 x 3-axis blocks the 17 MB of the bench leg took 2 x 14.7 us beside a 170 us row kernel: 128-256 blocks
// with sixteen dependent rounds of loads each).
MK_KERNEL(256) void k_sel_to_frames(const float* __restrict__ coords, long long F, const unsigned* __restrict__ sel1, long long n1,
                                    long long np1, const unsigned* __restrict__ sel2, long long n2, long long np2,
                                    const unsigned* __restrict__ chains, float* __restrict__ T1, unsigned* __restrict__ cs1,
                                    float* __restrict__ T2, unsigned* __restrict__ cs2)
{
    __shared__ float tile[DT][DT + 1];
    const int l = threadIdx.x & (DT - 1), wq = threadIdx.x >> 6, ax = (int)blockIdx.z;
    const long long tiles1 = np1 / DT;
    const bool second = (long long)blockIdx.y >= tiles1;             // block-uniform: which selection
    const unsigned* __restrict__ sel = second ? sel2 : sel1;
    const long long n = second ? n2 : n1, np = second ? np2 : np1;
    float* __restrict__ T = second ? T2 : T1;
    const long long f0 = (long long)blockIdx.x * DT, k0 = ((long long)blockIdx.y - (second ? tiles1 : 0)) * DT;
    const long long f = f0 + l < F ? f0 + l : F - 1;
    float v[DT / 4];
#pragma unroll
    for (int r = 0; r < DT / 4; ++r) {                               // sixteen independent loads per lane, one wait
        const long long k = k0 + wq + 4 * r < n ? k0 + wq + 4 * r : n - 1;
        v[r] = coords[((size_t)sel[k] * 3 + (size_t)ax) * (size_t)F + (size_t)f];
    }
#pragma unroll
    for (int r = 0; r < DT / 4; ++r) tile[wq + 4 * r][l] = v[r];
    if (blockIdx.x == 0 && ax == 0 && chains != nullptr && wq == 0) (second ? cs2 : cs1)[k0 + l] = chains[sel[k0 + l < n ? k0 + l : n - 1]];
    mk_block_sync();
#pragma unroll
    for (int r = 0; r < DT / 4; ++r) {
        const long long fr = f0 + wq + 4 * r;
        if (fr < F) T[((size_t)fr * 3 + (size_t)ax) * (size_t)np + (size_t)(k0 + l)] = tile[l][wq + 4 * r];
    }
}

// A wave: frame f, first atoms [ic * ROWS_CI, ...), second atoms of block jb: lane-strided (j = jb * 64 * JPL + lane + 64 * k,
// k < JPL: JPL stores of 256 contiguous bytes per first atom) or (VEC, JPL = 4) four neighbours per lane (j = jb * 256 + 4 * lane
// + k: ONE store of 1 KB per first atom, 16 bytes per lane at whatever alignment the row has).  Wave tasks are numbered in the
// result's memory order (frame, then chunk of first atoms, then block of second atoms) and dealt to the XCDs in contiguous
// ranges, like the tiles of the other kernels.
// TRI (round 6, late): the selfdist form -- only the pairs (i, j > i) exist, and they go out in the reference's condensed order
// (distance_utils.pyx:140-150: row i starts at i (n2 - 1) - i (i - 1) / 2 and holds j = i + 1 ... n2 - 1).  Same first / second atom, same
// arithmetic and wrap rule as the rectangle's pair (i, j): the same bits.  Tasks wholly on or below the diagonal leave at once, rows of a
// task on it store what lies above; P = the condensed row length.  (The pair-table kernel this replaces for few frames and for large
// selections runs its lanes along frames and loads two table entries and six gathered coordinates per pair.)
// SWAPPED (round 6, late): the caller handed the selections over the other way round -- the rows walk the reference's SECOND selection, the lanes
// run along its FIRST (a receptor's 20 000 atoms against a ligand's 40 on one frame: rows of 40 are too short for the lanes, rows of 20 000 are
// not) -- and the pair (row r, lane atom c) goes to out[f, c * n_rows + r], the reference's (first, second) order.  The separation enters with the
// opposite sign; every operation on it is odd (the subtraction, the quotient, both roundings, the shift) and it ends squared: the same bits.
template <bool PBC, int JPL, bool VEC, bool TRI = false, bool SWAPPED = false>
MK_KERNEL(256) void k_dist_rows(const float* __restrict__ T1, long long np1, const unsigned* __restrict__ c1, const float* __restrict__ T2,
                                long long np2, const unsigned* __restrict__ c2, const float* __restrict__ box, long long F, long long n1,
                                long long n2, int squared, float* __restrict__ out, long long P_tri = 0)
{
    static_assert(!VEC || JPL == 4, "four neighbours per lane");
    const long long NJ = (n2 + 64 * JPL - 1) / (64 * JPL), NI = (n1 + ROWS_CI - 1) / ROWS_CI;
    const long long tasks = F * NI * NJ, blocks = (tasks + 3) / 4;
    const long long gb = xcd_contiguous_tile(blocks);
    if (gb < 0) return;
    const long long task = gb * 4 + (long long)mk_uniform(threadIdx.x >> 6);
    if (task >= tasks) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const long long jb = task % NJ, ic = (task / NJ) % NI, f = task / (NJ * NI);
    if constexpr (TRI) {
        if (jb * 64 * JPL + 64 * JPL - 1 <= ic * ROWS_CI) return;   // wave-uniform: no pair of this task lies above the diagonal
    }
    const long long j0 = jb * 64 * JPL + (VEC ? 4 * lane : lane);
    constexpr int JS = VEC ? 1 : 64;                                // distance between a lane's second atoms
    float B[JPL][3];
    unsigned cb[JPL];
    const float* __restrict__ t2 = T2 + (size_t)f * 3 * (size_t)np2;
#pragma unroll
    for (int k = 0; k < JPL; ++k) {
        const long long j = j0 + JS * k;                            // (< np2: the pad)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) B[k][ax] = t2[(size_t)ax * (size_t)np2 + (size_t)j];
        cb[k] = PBC ? c2[j] : 0u;
    }
    float bx = 0.f, by = 0.f, bz = 0.f, ibx = 0.f, iby = 0.f, ibz = 0.f;
    if (PBC) {
        bx = box[0 * F + f]; by = box[1 * F + f]; bz = box[2 * F + f];
        ibx = mk_fdiv_rn(1.f, bx); iby = mk_fdiv_rn(1.f, by); ibz = mk_fdiv_rn(1.f, bz);
    }
    const long long P = TRI ? P_tri : n1 * n2;
    const long long i_begin = ic * ROWS_CI, i_end = i_begin + ROWS_CI < n1 ? i_begin + ROWS_CI : n1;
    const float* __restrict__ t1 = T1 + (size_t)f * 3 * (size_t)np1;
    float* __restrict__ o = out + (size_t)f * (size_t)P + (TRI ? (size_t)0 : (size_t)i_begin * (size_t)n2 + (size_t)j0);
    const bool whole = jb * 64 * JPL + 64 * JPL <= n2;              // wave-uniform: every lane's every pair exists
    for (long long i = i_begin; i < i_end; ++i, o += (TRI ? 0 : n2)) {
        if constexpr (TRI) {
            if (jb * 64 * JPL + 64 * JPL - 1 <= i) break;            // wave-uniform: this row and the ones after it end left of the task
        }
        const float xa = t1[i], ya = t1[(size_t)np1 + (size_t)i], za = t1[2 * (size_t)np1 + (size_t)i];   // wave-uniform: scalar loads
        const unsigned ca = PBC ? c1[i] : 0u;
        float d[JPL];
        if constexpr (JPL >= 2 && !PBC) {
#pragma unroll
            for (int k = 0; k < JPL; k += 2) {                       // two pairs per packed operation
                const mk_f2 d2 = dist2_x2(xa, ya, za, mk_f2{B[k][0], B[k + 1][0]}, mk_f2{B[k][1], B[k + 1][1]}, mk_f2{B[k][2], B[k + 1][2]});
                d[k] = d2[0]; d[k + 1] = d2[1];
            }
        } else {
#pragma unroll
            for (int k = 0; k < JPL; ++k)
                d[k] = dist2_min_image_f32(xa, ya, za, B[k][0], B[k][1], B[k][2], bx, by, bz, ibx, iby, ibz, PBC && cb[k] != ca);
        }
        const bool ordinary = mk_sqrt_ordinary_all(d);
        if (!squared) {
            if (mk_ballot(!ordinary) == 0ull) {
#pragma unroll
                for (int k = 0; k < JPL; ++k) d[k] = mk_fsqrt_rn_ordinary(d[k]);
            } else {
#pragma unroll
                for (int k = 0; k < JPL; ++k) d[k] = mk_fsqrt_rn(d[k]);
            }
        }
        if constexpr (SWAPPED) {
            static_assert(!TRI, "selfdist has no long side to swap to");
            float* __restrict__ cbase = out + (size_t)f * (size_t)P + (size_t)i;                 // out[f, j * n1 + i]: the lane's atom is the reference's first
#pragma unroll
            for (int k = 0; k < JPL; ++k) {
                const long long j = j0 + JS * k;
                if (j < n2) cbase[(size_t)j * (size_t)n1] = d[k];
            }
        } else if constexpr (TRI) {
            // the condensed row of i: element (i, j) at full (n2 - 1) - full (full - 1) / 2 + (j - i - 1), full = min(i, n2)
            const long long full = i < n2 ? i : n2;
            float* __restrict__ r = o + (full * (n2 - 1) - full * (full - 1) / 2 - i - 1);       // (+ j: only j > i is ever formed into an address below)
            if constexpr (VEC) {
                if (j0 > i && (whole || j0 + 3 < n2)) mk_store_f4_dword_aligned(r + j0, make_float4(d[0], d[1], d[2], d[3]));
                else {
#pragma unroll
                    for (int k = 0; k < JPL; ++k)
                        if (j0 + k > i && j0 + k < n2) r[j0 + k] = d[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < JPL; ++k)
                    if (j0 + 64 * k > i && j0 + 64 * k < n2) r[j0 + 64 * k] = d[k];
            }
        } else if constexpr (VEC) {
            // (rows of any length and any float* result: a row starts on 4 bytes only when n2 is not a multiple of four;
            //  non-temporal stores measured: opposite signs for the two modes)
            if (whole || j0 + 3 < n2) mk_store_f4_dword_aligned(o, make_float4(d[0], d[1], d[2], d[3]));
            else {
#pragma unroll
                for (int k = 0; k < JPL; ++k)
                    if (j0 + k < n2) o[k] = d[k];
            }
        } else if (whole) {
#pragma unroll
            for (int k = 0; k < JPL; ++k) o[64 * k] = d[k];
        } else {
#pragma unroll
            for (int k = 0; k < JPL; ++k)
                if (j0 + 64 * k < n2) o[64 * k] = d[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One launch for rectangular calls with SHORT rows (round 5): a BLOCK per frame (or per slice of a frame's pairs).  The atoms of both
// selections are few, so the block stages the frame's copy of them in LDS -- (n1 + n2) x {x, y, z, chain}: strided gathers from the
// reference layout, no turning launch in front -- and then walks the frame's pair list in its MEMORY order, four consecutive pairs
// per lane: the wave writes 1 KB of out[f, p..] per store instruction however short the rows are (the calls MetricDistance usually
// makes, protein C-alphas x ligand atoms: 300 x 30 pairs x 2 048 frames took the rectangular tile kernel 43 us open / 62 us periodic,
// this 34 / 40; 1 000 x 30: 126 / 167 against 127 / 162; profiles/r5_dist_shapes_probe.txt).  No pair table: a lane finds the
// (i, j) of its first pair by one division and steps.  Same functions per pair (dist2_min_image_f32, mk_fsqrt_rn): the same bits
// as every other kernel here.  (The triangular list of selfdist through the same walk -- rows found by a square root and stepping --
// was measured too: 450 x 450 318 us against 235 for the pair-table kernel, 60 x 60 23 us against 12: finding the rows costs more
// than the table's two loads; selfdist keeps the pair-table kernel.)
// ------------------------------------------------------------------------------------------------
constexpr int DF_THREADS = 256, DF_PPL = 4;                  // pairs per lane and step
constexpr int DF_STEP = DF_THREADS * DF_PPL;                 // pairs a block takes per step

// A block takes FR CONSECUTIVE frames (and a slice of their pairs).  The reference layout is frame-fastest, so the FR frames of one
// (atom, axis) are 4 FR contiguous bytes: staged with ONE 16-byte load where FR = 4 and the rows are aligned (F a multiple of four) --
// a quarter of the cache lines a block per single frame asks for (4 096 blocks x 990 lines for the 300 x 30 x 2 048 call).  Measured:
// the staging with four work items in flight per thread took the 1 000 x 30 call from 127 to 112 us (periodic 162 -> 145); the four
// frames per block changed nothing measurable at 300 x 30 (33.4 -> 33.6 us in the probe): what is left of that kernel's 34 us is its
// ~26 instructions per pair (a third of them index arithmetic and LDS addresses), not the lines it fetches.
// I: the type pair numbers are computed in -- unsigned while the frame's list is shorter than 2^30 pairs (64-bit multiplies and
// divisions are a dozen instructions each), long long beyond
template <bool PBC, int CAP /* atoms of both selections the LDS copy holds */, int FR /* frames per block */, typename I>
MK_KERNEL(DF_THREADS) void k_dist_frame(const float* __restrict__ coords, long long F, const float* __restrict__ box,
                                        const unsigned* __restrict__ sel1, long long n1_, const unsigned* __restrict__ sel2, long long n2_,
                                        const unsigned* __restrict__ chains, int squared, long long slices, float* __restrict__ out)
{
    static_assert(FR == 1 || FR == 4, "one frame, or four with 16-byte loads");
    __shared__ float4 s_at[FR][CAP];                                 // {x, y, z, chain id bits} per frame: sel1's atoms, then sel2's
    const long long groups = (F + FR - 1) / FR;
    const long long g = xcd_contiguous_tile(groups * slices);        // (frame group, slice) in the result's memory order, XCD-contiguous
    if (g < 0) return;
    const long long fg = g / slices, sl = g - fg * slices, f0 = fg * FR;
    const int tid = threadIdx.x;
    const I n1 = (I)n1_, n2 = (I)n2_;
    const long long P_ = n1_ * n2_;
    const int na = (int)(n1_ + n2_);
    // staging: work items (atom, axis), FOUR per thread in flight -- the atom indices in one round trip, then the coordinates (and, with
    // the x axis, the chain) in another: a block's time before its first pair is a chain of round trips
    const bool vec = FR == 4 && (F & 3) == 0 && (reinterpret_cast<uintptr_t>(coords) & (uintptr_t)15) == 0;     // block-uniform
    for (int w0 = 0; w0 < 3 * na; w0 += 4 * DF_THREADS) {           // block-uniform
        unsigned a[4];
        float v[4][FR];
        unsigned ch[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + u * DF_THREADS + tid, wc = w < 3 * na ? w : 3 * na - 1, k = wc / 3;
            a[u] = k < (int)n1_ ? sel1[k] : sel2[k - (int)n1_];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + u * DF_THREADS + tid, wc = w < 3 * na ? w : 3 * na - 1, ax = wc - (wc / 3) * 3;
            const float* __restrict__ c = coords + ((size_t)a[u] * 3 + (size_t)ax) * (size_t)F + (size_t)f0;
            if (FR == 4 && vec) {
                const float4 q = *reinterpret_cast<const float4*>(c);
                v[u][0] = q.x; v[u][FR > 1 ? 1 : 0] = q.y; v[u][FR > 2 ? 2 : 0] = q.z; v[u][FR > 3 ? 3 : 0] = q.w;
            } else {
#pragma unroll
                for (int r = 0; r < FR; ++r) v[u][r] = c[f0 + r < F ? r : 0];
            }
            ch[u] = (PBC && ax == 0) ? chains[a[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + u * DF_THREADS + tid;
            if (w < 3 * na) {
                const int k = w / 3, ax = w - k * 3;
#pragma unroll
                for (int r = 0; r < FR; ++r) {
                    float* dst = reinterpret_cast<float*>(&s_at[r][k]);
                    dst[ax] = v[u][r];
                    if (ax == 0) dst[3] = mk_uint_as_float(ch[u]);
                }
            }
        }
    }
    mk_block_sync();
    // this slice's pairs [p_lo, p_hi): whole steps of the block, so that a lane's four pairs start on a multiple of four
    const long long steps = (P_ + DF_STEP - 1) / DF_STEP, per = (steps + slices - 1) / slices;
    const long long p_lo_ = sl * per * DF_STEP, p_end_ = (sl + 1) * per * DF_STEP;
    const I p_lo = (I)(p_lo_ < P_ ? p_lo_ : P_), p_hi = (I)(p_end_ < P_ ? p_end_ : P_);
    // the step from a lane's last pair + 1 to its next first pair, in whole rows and a rest (block-uniform)
    const I adv = (I)(DF_STEP - DF_PPL), adv_rows = adv / n2, adv_rest = adv - adv_rows * n2;
    const I p_first = p_lo + (I)tid * (I)DF_PPL;
    const I i_first = p_first / n2, j_first = p_first - i_first * n2;
    for (int r = 0; r < FR; ++r) {                                   // block-uniform
        const long long f = f0 + r;
        if (f >= F) break;
        float bx = 0.f, by = 0.f, bz = 0.f, ibx = 0.f, iby = 0.f, ibz = 0.f;
        if (PBC) {
            bx = box[0 * F + f]; by = box[1 * F + f]; bz = box[2 * F + f];
            ibx = mk_fdiv_rn(1.f, bx); iby = mk_fdiv_rn(1.f, by); ibz = mk_fdiv_rn(1.f, bz);
        }
        float* __restrict__ row = out + (size_t)f * (size_t)P_;
        const float4* __restrict__ s1 = s_at[r];
        const float4* __restrict__ s2 = s_at[r] + n1_;
        I p = p_first, i = i_first, j = j_first;
        // (the trip count is the BLOCK's: the roots' ballot inside is taken by whole waves; a lane past the end of the list computes
        //  on clamped atoms and stores nothing)
        for (I base = p_lo; base < p_hi; base += (I)DF_STEP, p += (I)DF_STEP) {
            float d[DF_PPL];
#pragma unroll
            for (int k = 0; k < DF_PPL; ++k) {
                const I ic = i < n1 ? i : n1 - 1;                                         // (pairs past the end of the list: the last atom again, never stored)
                const float4 A = s1[ic], B = s2[j];
                const bool wrap = PBC && mk_float_bits(A.w) != mk_float_bits(B.w);       // distance_utils.pyx:49
                d[k] = dist2_min_image_f32(A.x, A.y, A.z, B.x, B.y, B.z, bx, by, bz, ibx, iby, ibz, wrap);
                if (++j >= n2) { ++i; j = 0; }
            }
            const bool ordinary = mk_sqrt_ordinary_all(d);
            if (!squared) {
                if (mk_ballot(!ordinary) == 0ull) {
#pragma unroll
                    for (int k = 0; k < DF_PPL; ++k) d[k] = mk_fsqrt_rn_ordinary(d[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < DF_PPL; ++k) d[k] = mk_fsqrt_rn(d[k]);
                }
            }
            if (p + (I)DF_PPL <= p_hi) {
                mk_store_f4_dword_aligned(row + p, make_float4(d[0], d[1], d[2], d[3]));   // (rows of an odd pitch start on 4 bytes only)
            } else if (p < p_hi) {
#pragma unroll
                for (int k = 0; k < DF_PPL; ++k)
                    if (p + (I)k < p_hi) row[p + (I)k] = d[k];
            }
            // on to pair p + DF_STEP: (i, j) stands at pair p + DF_PPL
            i += adv_rows; j += adv_rest;
            if (j >= n2) { j -= n2; ++i; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// contacts_trajectory / get_collisions (distance_utils.pyx:59-93, :98-121) entirely on the device: the reference
// appends (a, b) inside its (frame, i, j) loops whenever dist2 <= threshold^2 (float32); here the same pairs come out
// in the same order without ever materialising the [frames x pairs] distance matrix:
//   k_contacts_count : the 64-frame x 64-pair tiles of k_dist_pairs; every lane (= frame) tests its wave's 16
//                      consecutive pairs and the block stores ONE count per (tile, frame)
//   k_contacts_scan  : per frame, exclusive prefix of the counts over the pair tiles (+ the frame's total)
//   k_contacts_fill  : the same tiles again; a lane keeps its 16 hits as a bit mask, ranks them behind the tile's
//                      prefix and the lower waves of its block, and writes (a, b) at frame_base + rank: (i, j) order
//                      (round 6: from the 16-bit masks the count pass left -- no distance is computed twice)
// Frames are processed in chunks (host loop, dist_pipeline.h: run_contacts) so that the counters stay within a fixed memory budget.
// (Round 6: these three serve the calls the kernels further down do not take -- rows that fill less than three eighths of their
//  64-wide tiles, i.e. fewer than 24 second atoms; everything else counts with its second atoms in registers: k_contacts_count_rect,
//  k_contacts_count_rect_few, k_contacts_fill_rect.)
// ------------------------------------------------------------------------------------------------
constexpr int CT_RUN = DT / (DT_THREADS / DT);        // consecutive pairs per wave of a tile (16)

// bit k set: pair (p_first + k) of frame f is a contact.  Same traversal as k_dist_pairs (for_pair_run); a padded frame
// (`fin` false) computes on frame 0 and reports nothing.
MK_DEV unsigned contact_mask(const float* __restrict__ coords, long long F, long long f, bool fin, float bx, float by, float bz,
                             const unsigned* __restrict__ pa, const unsigned* __restrict__ pb,
                             const unsigned* __restrict__ wrap, long long P, long long p_first, float thr2)
{
    static_assert(CT_RUN == DP_RUN, "one run per wave");
    if (p_first >= P) return 0u;                                     // wave-uniform
    unsigned mask = 0u;
    for_pair_run(coords, F, fin ? (unsigned)f * 4u : 0u, bx, by, bz, pa, pb, wrap, P, p_first,
                 [&](int k0, const float (&d2)[DP_BATCH]) {                          // distance_utils.pyx:82 / :111 (NaN: no contact)
#pragma unroll
                     for (int u = 0; u < DP_BATCH; ++u) mask |= (d2[u] <= thr2) ? (1u << (k0 + u)) : 0u;
                 });
    const long long left = P - p_first;
    if (left < CT_RUN) mask &= (1u << (unsigned)left) - 1u;
    return fin ? mask : 0u;
}

// blockIdx.x = pair tile, blockIdx.y = 64-frame slab of the chunk [f_begin, f_begin + fc); cnt is [tiles][fc_pad];
// masks is [tiles * 4 runs][fc_pad] (round 6): the 16 contact bits of every (run of 16 pairs, frame), kept for the fill pass --
// which then computes no distance at all (before: every pair twice, 0.56 ms for 200 x 500 pairs x 2 048 frames)
MK_KERNEL(DT_THREADS) void k_contacts_count(const float* __restrict__ coords, long long F, long long f_begin, long long fc,
                                            long long fc_pad, const float* __restrict__ box, const unsigned* __restrict__ pa,
                                            const unsigned* __restrict__ pb, const unsigned* __restrict__ wrap, long long P,
                                            float thr2, unsigned* __restrict__ cnt, unsigned short* __restrict__ masks)
{
    __shared__ unsigned s_c[DT_THREADS / DT][DT];
    const int fl = threadIdx.x & (DT - 1), pq = threadIdx.x >> 6;
    const long long lf = (long long)blockIdx.y * DT + fl, f = f_begin + lf;
    const bool fin = lf < fc;
    const float bx = fin ? box[0 * F + f] : 1.f, by = fin ? box[1 * F + f] : 1.f, bz = fin ? box[2 * F + f] : 1.f;
    const unsigned m = contact_mask(coords, F, f, fin, bx, by, bz, pa, pb, wrap, P, (long long)blockIdx.x * DT + pq * CT_RUN, thr2);
    masks[((size_t)blockIdx.x * (DT_THREADS / DT) + (size_t)pq) * (size_t)fc_pad + (size_t)lf] = (unsigned short)m;
    s_c[pq][fl] = (unsigned)__builtin_popcount(m);
    mk_block_sync();
    if (pq == 0) {
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < DT_THREADS / DT; ++w) t += s_c[w][fl];
        cnt[(size_t)blockIdx.x * fc_pad + lf] = t;                 // padded frames count 0
    }
}

// one block per 64-frame slab: lanes = frames, the block's CS_WAVES waves split the tile range; in-place exclusive prefix over the
// tiles of every frame, totals[frame] = its number of contacts.  (Round 6: sixteen waves instead of four -- with 1 563 tiles a wave
// walked 390 of them twice, one dependent load after the other: 134 us of a 370-us contacts call, profiles/r6_dist_rocprofv3_kernel_stats.csv.)
constexpr int CS_WAVES = 16;
MK_KERNEL(CS_WAVES * WAVE) void k_contacts_scan(unsigned* __restrict__ cnt, long long tiles, long long fc_pad /* row pitch of cnt */,
                                                long long lanes /* frames that exist in a row: <= the pitch, or the pitch itself (padded rows) */,
                                                unsigned long long* __restrict__ totals)
{
    __shared__ unsigned long long s_seg[CS_WAVES][DT];
    const int fl = threadIdx.x & (DT - 1), w = threadIdx.x >> 6;
    constexpr int NW = CS_WAVES;
    const long long lf = (long long)blockIdx.x * DT + fl;
    const long long per = (tiles + NW - 1) / NW, t0 = w * per < tiles ? w * per : tiles, t1 = t0 + per < tiles ? t0 + per : tiles;
    unsigned long long sum = 0;
    const bool on = lf < lanes;
    for (long long t = t0; t < t1; ++t) sum += on ? cnt[(size_t)t * fc_pad + lf] : 0u;
    s_seg[w][fl] = sum;
    mk_block_sync();
    unsigned long long run = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { if (i < w) run += s_seg[i][fl]; total += s_seg[i][fl]; }
    for (long long t = t0; on && t < t1; ++t) {
        const unsigned c = cnt[(size_t)t * fc_pad + lf];
        cnt[(size_t)t * fc_pad + lf] = (unsigned)run;              // per-frame prefix: < 2^32 pairs per frame (host checks P)
        run += c;
    }
    if (w == 0) totals[lf] = total;
}

MK_KERNEL(DT_THREADS) void k_contacts_fill(long long fc, long long fc_pad, const unsigned* __restrict__ pa, const unsigned* __restrict__ pb,
                                           const unsigned short* __restrict__ masks, const unsigned* __restrict__ prefix,
                                           const unsigned long long* __restrict__ frame_base, uint2* __restrict__ out)
{
    __shared__ unsigned s_c[DT_THREADS / DT][DT];
    const int fl = threadIdx.x & (DT - 1), pq = threadIdx.x >> 6;
    const long long lf = (long long)blockIdx.y * DT + fl;
    const bool fin = lf < fc;
    const long long p_first = (long long)blockIdx.x * DT + pq * CT_RUN;
    unsigned m = masks[((size_t)blockIdx.x * (DT_THREADS / DT) + (size_t)pq) * (size_t)fc_pad + (size_t)lf];   // (0 for padded frames and pairs past the end)
    s_c[pq][fl] = (unsigned)__builtin_popcount(m);
    mk_block_sync();
    if (!fin || m == 0u) return;
    unsigned long long pos = frame_base[lf] + prefix[(size_t)blockIdx.x * fc_pad + lf];
    for (int w = 0; w < pq; ++w) pos += s_c[w][fl];
    while (m) {                                                    // ascending pair index = the reference's (i, j) order
        const int k = __builtin_ctz(m);
        m &= m - 1u;
        out[pos++] = make_uint2(pa[p_first + k], pb[p_first + k]);
    }
}

// ------------------------------------------------------------------------------------------------
// The contact lists of a RECTANGULAR call (no selfdist: every sel1 atom against every sel2 atom), round 6.  The pair-table walk
// above loads the second atom of EVERY pair (three 256-byte loads per 64 pair-frames) and spends 33 instructions on a periodic
// pair; its counters are per 64-pair tile, so a scan over 1 600 tiles and a fill pass of as many blocks follow (43 + 46 us of a
// 0.30-ms call for 200 x 500 pairs x 2 048 frames).  Here -- as in k_dist_rect -- a wave keeps ITS sixteen second atoms in 48
// registers (as eight packed pairs: lane = frame) and walks a GROUP of first atoms: the sixteen d^2 of a first atom come from
// eight packed calls (dist2_pk), their image integers behind ONE accumulated test; rows of which only some pairs wrap compute
// both forms and choose per pair.  Nothing goes through LDS, there is no barrier.
//   masks  [(i * JT + jt) * 4 + run][frame]   the 16 contact bits of (first atom i, second atoms jt * 64 + 16 run ..), JT = ceil(n2 / 64);
//                                             zero past a row's end and for padded frames
//   cntg   [group][frame]                     contacts of the group's rows (groups of `ni` consecutive first atoms: contiguous in the
//                                             reference's (i, j) order) -- added up by the waves with atomics; k_contacts_scan then runs over
//                                             the GROUPS (25 instead of 1 600 tiles) and k_contacts_fill_rect finds a mask's place inside
//                                             its group itself.  No pair table is built.
// ------------------------------------------------------------------------------------------------
template <bool PBC, bool SMALL>
MK_DEV void contacts_rect_block(const float* __restrict__ coords, long long F, long long f_begin, long long fc, long long fc_pad,
                                const float* __restrict__ box, const unsigned* __restrict__ sel1, long long n1,
                                const unsigned* __restrict__ sel2, long long n2, const unsigned* __restrict__ chains, float thr2, long long ni, int selfdist,
                                unsigned* __restrict__ cntg, unsigned short* __restrict__ masks, long long g, long long jt, long long JT)
{
    const int fl = threadIdx.x & (DT - 1), pq = threadIdx.x >> 6;
    const long long lf = (long long)blockIdx.y * DT + fl;
    const bool fin = lf < fc;
    const long long f = fin ? f_begin + lf : f_begin;                // a padded frame computes on the chunk's first one and reports nothing
    const unsigned fb = (unsigned)f * 4u, F4 = (unsigned)F * 4u;
    auto at = [&](unsigned atom, int ax) {
        if constexpr (SMALL) return mk_load_f32_base_soffset(coords, (atom * 3u + (unsigned)ax) * F4, fb);
        else return mk_load_f32_uniform_base(coords + ((size_t)atom * 3 + (size_t)ax) * (size_t)F, fb);
    };
    const long long i0 = g * ni, rows = n1 - i0 < ni ? n1 - i0 : ni;            // block-uniform, >= 1
    const long long jw = jt * DT + (long long)pq * CT_RUN;           // the wave's first second atom
    const int nv = n2 - jw >= CT_RUN ? CT_RUN : (n2 - jw > 0 ? (int)(n2 - jw) : 0);     // wave-uniform: how many of its 16 exist
    auto mask_at = [&](long long ii) -> unsigned short& {
        return masks[(((size_t)(i0 + ii) * (size_t)JT + (size_t)jt) * (DT_THREADS / DT) + (size_t)pq) * (size_t)fc_pad + (size_t)lf];
    };
    // selfdist (distance_utils.pyx:76: j from i + 1): a pair exists where j > i -- a run whose last atom is not beyond the group's FIRST row
    // holds none for any of its rows
    if (nv == 0 || (selfdist && jw + CT_RUN - 1 <= i0)) {            // nothing here, said explicitly (the fill pass reads it)
        for (long long ii = 0; ii < rows; ++ii) mask_at(ii) = 0;
        return;
    }
    // lane k holds second atom k and its chain (past the row's end: the last one again -- computed, masked out)
    const long long jk = jw + (fl & (CT_RUN - 1)) < n2 ? jw + (fl & (CT_RUN - 1)) : n2 - 1;
    const unsigned vb = sel2[jk], vcb = PBC ? chains[vb] : 0u;
    constexpr int H = CT_RUN / 2;
    mk_f2 BX[H], BY[H], BZ[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const unsigned b0 = mk_readlane(vb, 2 * h), b1 = mk_readlane(vb, 2 * h + 1);
        BX[h] = mk_f2{at(b0, 0), at(b1, 0)}; BY[h] = mk_f2{at(b0, 1), at(b1, 1)}; BZ[h] = mk_f2{at(b0, 2), at(b1, 2)};
    }
    float bx = 1.f, by = 1.f, bz = 1.f, ibx = 1.f, iby = 1.f, ibz = 1.f;
    if (PBC) {
        bx = box[0 * F + f]; by = box[1 * F + f]; bz = box[2 * F + f];
        ibx = mk_fdiv_rn(1.f, bx); iby = mk_fdiv_rn(1.f, by); ibz = mk_fdiv_rn(1.f, bz);
    }
    const unsigned valid = nv >= CT_RUN ? 0xffffu : (1u << (unsigned)nv) - 1u;
    // the bits of row i that are pairs of a selfdist call: second atoms jw + k with jw + k > i (wave-uniform).  (Rows of a run that straddles
    // the diagonal are computed whole and masked: skipping them -- an `if` around the arithmetic -- cost the walk 80 registers.)
    auto upper = [&](long long i) -> unsigned {
        if (!selfdist) return 0xffffu;
        const long long first = i + 1 - jw;                          // the first bit that counts
        return first <= 0 ? 0xffffu : first >= CT_RUN ? 0u : (0xffffu << (unsigned)first) & 0xffffu;
    };
    unsigned a = sel1[i0], total = 0u, redo_rows = 0u;
    float xa = at(a, 0), ya = at(a, 1), za = at(a, 2);
    for (long long ii = 0; ii < rows; ++ii) {
        // the next first atom's coordinates are requested before this one's distances are computed
        const unsigned a_next = sel1[ii + 1 < rows ? i0 + ii + 1 : i0 + ii];
        const float xn = at(a_next, 0), yn = at(a_next, 1), zn = at(a_next, 2);
        // which of the 16 second atoms wrap against this first atom (pbc and different chains, distance_utils.pyx:49): wave-uniform bits
        unsigned wm = 0u;
        if (PBC) { const unsigned ca = chains[a]; wm = (unsigned)(mk_ballot(vcb != ca) & 0xffffull); }
        unsigned m = 0u;
        bool redo = false;
        float none = 0.f;
        if (PBC && wm == 0xffffu) {
            float risk = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const mk_f2 d2 = dist2_pk<true>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, risk);
                m |= (d2[0] <= thr2 ? 1u << (2 * h) : 0u) | (d2[1] <= thr2 ? 2u << (2 * h) : 0u);         // distance_utils.pyx:82 (NaN: no contact)
                if (h & 1) mk_sched_barrier();                     // (two packed calls at a time: left alone the scheduler interleaves all eight -- 215 registers)
            }
            redo = mk_ballot(!(risk < DRC_RISK)) != 0ull;            // an image integer may differ from round(d / b) (rare)
        } else if (PBC && wm != 0u) {
            // only some of the sixteen wrap (wave-uniform bits): both forms, chosen per pair -- no branch, four packed instructions more
            float risk = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const mk_f2 w2 = dist2_pk<true>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, risk);
                const mk_f2 o2 = dist2_pk<false>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, none);
                const float d0 = ((wm >> (2 * h)) & 1u) ? w2[0] : o2[0], d1 = ((wm >> (2 * h + 1)) & 1u) ? w2[1] : o2[1];
                m |= (d0 <= thr2 ? 1u << (2 * h) : 0u) | (d1 <= thr2 ? 2u << (2 * h) : 0u);
                mk_sched_barrier();
            }
            redo = mk_ballot(!(risk < DRC_RISK)) != 0ull;
        } else {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const mk_f2 d2 = dist2_pk<false>(BX[h], BY[h], BZ[h], xa, ya, za, bx, by, bz, ibx, iby, ibz, none);
                m |= (d2[0] <= thr2 ? 1u << (2 * h) : 0u) | (d2[1] <= thr2 ? 2u << (2 * h) : 0u);
                if ((h & 3) == 3) mk_sched_barrier();
            }
        }
        if (redo) redo_rows |= 1u << (unsigned)ii;                  // (wave-uniform; at most 32 rows per group)
        m = fin ? m & valid & upper(i0 + ii) : 0u;
        mask_at(ii) = (unsigned short)m;
        total += (unsigned)__builtin_popcount(m);
        a = a_next; xa = xn; ya = yn; za = zn;
    }
    while (redo_rows) {
        // rows whose accumulated test failed in some lane, once more: pair by pair with the per-pair test and the correctly rounded
        // divisions behind it, the atoms loaded again.  (AFTER the walk and as loops: inside it, as sixteen inlined pairs, the kernel held
        // 182 registers, as a loop 209 -- two waves per SIMD; out here the walk keeps ~100.)
        mk_stay_in_branch();
        const int ii = __builtin_ctz(redo_rows);
        redo_rows &= redo_rows - 1u;
        const unsigned ar = sel1[i0 + ii];
        const unsigned wm = (unsigned)(mk_ballot(vcb != chains[ar]) & 0xffffull);
        const float xr = at(ar, 0), yr = at(ar, 1), zr = at(ar, 2);
        unsigned m = 0u;
#pragma unroll 1
        for (int k = 0; k < CT_RUN; ++k) {
            const unsigned bk = mk_readlane(vb, k);
            const float d = dist2_min_image_f32(xr, yr, zr, at(bk, 0), at(bk, 1), at(bk, 2), bx, by, bz, ibx, iby, ibz, ((wm >> k) & 1u) != 0u);
            m |= d <= thr2 ? 1u << k : 0u;
        }
        m = fin ? m & valid & upper(i0 + ii) : 0u;
        const unsigned old = mask_at(ii);
        mask_at(ii) = (unsigned short)m;
        total += (unsigned)__builtin_popcount(m) - (unsigned)__builtin_popcount(old);
    }
    if (total) mk_atomic_add(&cntg[(size_t)g * (size_t)fc_pad + (size_t)lf], total);
}

// blockIdx.x = (group of `ni` first atoms) * JT + tile of 64 second atoms, blockIdx.y = 64-frame slab of the chunk; cntg (zeroed by the
// host sequence) is [groups][fc_pad]
template <bool PBC>
MK_KERNEL(DT_THREADS) void k_contacts_count_rect(const float* __restrict__ coords, long long F, long long f_begin, long long fc, long long fc_pad,
                                                 const float* __restrict__ box, const unsigned* __restrict__ sel1, long long n1,
                                                 const unsigned* __restrict__ sel2, long long n2, const unsigned* __restrict__ chains,
                                                 float thr2, long long ni, int selfdist, unsigned* __restrict__ cntg, unsigned short* __restrict__ masks)
{
    // (the tiles of a group ROTATED by the group's number: consecutive blocks go round-robin to the 8 XCDs, and with JT a multiple of 8 XCD k
    //  would be handed tile k of every group -- in a selfdist call, where the low tiles of most groups lie below the diagonal and are not computed,
    //  XCD 7 then had 512 full blocks and XCD 0 sixty: 289 us for 450 x 450 x 2 048 where an even deal takes 180)
    const long long JT = (n2 + DT - 1) / DT, g = (long long)blockIdx.x / JT, jt = ((long long)blockIdx.x % JT + g) % JT;
    // every row this BLOCK touches ends below 4 GiB from the start of the array?  (block-uniform: every wave looks at the tile's 64
    // second atoms and at first atoms of the group, lane l at the l-th of them as far as they go)
    const int l = threadIdx.x & (DT - 1);
    const long long jl = jt * DT + l;
    unsigned hi_atom = sel2[jl < n2 ? jl : n2 - 1];
    for (long long il = g * ni + l; il < n1 && il < (g + 1) * ni; il += DT) { const unsigned ha = sel1[il]; hi_atom = ha > hi_atom ? ha : hi_atom; }
    const bool small_rows = mk_ballot(((unsigned long long)hi_atom * 3ull + 3ull) * ((unsigned long long)F * 4ull) > 0xffffffffull) == 0ull;
    if (small_rows) contacts_rect_block<PBC, true>(coords, F, f_begin, fc, fc_pad, box, sel1, n1, sel2, n2, chains, thr2, ni, selfdist, cntg, masks, g, jt, JT);
    else contacts_rect_block<PBC, false>(coords, F, f_begin, fc, fc_pad, box, sel1, n1, sel2, n2, chains, thr2, ni, selfdist, cntg, masks, g, jt, JT);
}

// The count pass for calls of FEW frames (get_collisions: one; a single structure's contact list): with lanes along frames such a call
// uses one lane in 64.  Here lanes run along the SECOND ATOMS of a row tile: a wave keeps its 64 second atoms of ONE frame in three
// registers and walks the group's first atoms (their coordinates arrive through scalar loads); a row's 64 contact bits are one ballot =
// the tile's four 16-bit masks, in the layout of k_contacts_count_rect -- scan and fill do not know the difference.  Per pair
// dist2_min_image_f32 (the pair's own image test: lanes are different pairs here).
// blockIdx.x = group * JT + tile (rotated as above), blockIdx.y = frame of the chunk; one wave per block.
template <bool PBC>
MK_KERNEL(WAVE) void k_contacts_count_rect_few(const float* __restrict__ coords, long long F, long long f_begin, long long fc_pad,
                                               const float* __restrict__ box, const unsigned* __restrict__ sel1, long long n1,
                                               const unsigned* __restrict__ sel2, long long n2, const unsigned* __restrict__ chains,
                                               float thr2, long long ni, int selfdist, unsigned* __restrict__ cntg, unsigned short* __restrict__ masks)
{
    const long long JT = (n2 + DT - 1) / DT, g = (long long)blockIdx.x / JT, jt = ((long long)blockIdx.x % JT + g) % JT;
    const long long lf = blockIdx.y, f = f_begin + lf;
    const int l = threadIdx.x & (WAVE - 1);
    const long long j = jt * DT + l;
    const bool has = j < n2;
    const unsigned b = sel2[has ? j : n2 - 1];
    const float x2 = coords[((size_t)b * 3 + 0) * (size_t)F + (size_t)f], y2 = coords[((size_t)b * 3 + 1) * (size_t)F + (size_t)f],
                z2 = coords[((size_t)b * 3 + 2) * (size_t)F + (size_t)f];
    const unsigned cb = PBC ? chains[b] : 0u;
    float bx = 1.f, by = 1.f, bz = 1.f, ibx = 1.f, iby = 1.f, ibz = 1.f;
    if (PBC) {
        bx = box[0 * F + f]; by = box[1 * F + f]; bz = box[2 * F + f];
        ibx = mk_fdiv_rn(1.f, bx); iby = mk_fdiv_rn(1.f, by); ibz = mk_fdiv_rn(1.f, bz);
    }
    const long long i0 = g * ni, rows = n1 - i0 < ni ? n1 - i0 : ni;
    unsigned total = 0u;
    for (long long ii = 0; ii < rows; ++ii) {
        const long long i = i0 + ii;
        const unsigned a = sel1[i];                                  // wave-uniform: scalar loads
        const float xa = coords[((size_t)a * 3 + 0) * (size_t)F + (size_t)f], ya = coords[((size_t)a * 3 + 1) * (size_t)F + (size_t)f],
                    za = coords[((size_t)a * 3 + 2) * (size_t)F + (size_t)f];
        const bool wrap = PBC && cb != chains[a];                    // distance_utils.pyx:49
        const float d2 = dist2_min_image_f32(xa, ya, za, x2, y2, z2, bx, by, bz, ibx, iby, ibz, wrap);
        const unsigned long long m = mk_ballot(has && d2 <= thr2 && (!selfdist || j > i));       // :82 (NaN: no contact); :76 (j from i + 1)
        if (l < DT_THREADS / DT)
            masks[(((size_t)i * (size_t)JT + (size_t)jt) * (DT_THREADS / DT) + (size_t)l) * (size_t)fc_pad + (size_t)lf] = (unsigned short)((m >> (16 * l)) & 0xffffull);
        total += (unsigned)mk_popc64(m);
    }
    if (l == 0 && total) mk_atomic_add(&cntg[(size_t)g * (size_t)fc_pad + (size_t)lf], total);
}

// The fill pass of a rectangular call: blockIdx.x = group of first atoms, blockIdx.y = 64-frame slab; lanes = frames.  The group's masks of a
// frame -- rows x JT * 4 runs, in the reference's (i, j) order -- are split among the block's CF_WAVES waves: a wave counts its stretch,
// the stretches' counts meet in LDS, then it walks the stretch again and writes (a, b) from
//   frame_base[frame] + gprefix[group][frame] (k_contacts_scan over the groups) + the stretches before its own.
// The pair behind bit k of run r of row i is (sel1[i], sel2[16 r + k]).
constexpr int CF_WAVES = 16;
constexpr int CF_CHUNK = 16;              // masks a lane has in flight
constexpr int CF_SEL2 = 4096;             // second atoms a block stages in LDS (16 KB)
MK_KERNEL(CF_WAVES * WAVE) void k_contacts_fill_rect(long long fc, long long fc_pad /* row pitch of masks / gprefix: >= fc */, const unsigned* __restrict__ sel1, long long n1,
                                                     const unsigned* __restrict__ sel2, long long n2, long long ni,
                                                     const unsigned short* __restrict__ masks, const unsigned* __restrict__ gprefix,
                                                     const unsigned long long* __restrict__ frame_base, uint2* __restrict__ out)
{
    __shared__ unsigned s_seg[CF_WAVES][DT];
    __shared__ unsigned s_sel2[CF_SEL2];
    const int fl = threadIdx.x & (DT - 1), w = threadIdx.x >> 6;
    const long long lf = (long long)blockIdx.y * DT + fl, g = blockIdx.x;
    const long long R4 = ((n2 + DT - 1) / DT) * (DT_THREADS / DT);   // runs per row, padded to whole tiles
    const long long i0 = g * ni, rows = n1 - i0 < ni ? n1 - i0 : ni, M = rows * R4;
    const long long per = (M + CF_WAVES - 1) / CF_WAVES, q0 = w * per < M ? w * per : M, q1 = q0 + per < M ? q0 + per : M;
    const unsigned short* __restrict__ mrow = masks + (size_t)i0 * (size_t)R4 * (size_t)fc_pad + (size_t)lf;   // mask q of this frame: mrow[q * fc_pad]
    // the atoms behind the bits: the group's first atoms in the lanes of a register (at most 32 rows), the second atoms in LDS -- as
    // loads from memory inside the walk every contact cost a round trip to the L2 before its store (38 us for 1.5 M contacts)
    const bool staged = n2 <= CF_SEL2;                               // block-uniform
    if (staged) for (long long j = threadIdx.x; j < n2; j += CF_WAVES * WAVE) s_sel2[j] = sel2[j];
    const unsigned va = sel1[i0 + (fl < rows ? fl : 0)];
    // CF_CHUNK masks at a time, their loads issued together; the first chunk stays in registers for the second walk
    const bool on = lf < fc;                                         // (rows of few-frame calls are not padded to 64 frames: lanes past the end load nothing)
    unsigned sum = 0u;
    unsigned first[CF_CHUNK];
    for (long long q = q0; q < q1; q += CF_CHUNK) {
        unsigned t[CF_CHUNK];
#pragma unroll
        for (int u = 0; u < CF_CHUNK; ++u) t[u] = (on && q + u < q1) ? (unsigned)mrow[(size_t)(q + u) * (size_t)fc_pad] : 0u;
#pragma unroll
        for (int u = 0; u < CF_CHUNK; ++u) sum += (unsigned)__builtin_popcount(t[u]);
        if (q == q0) {
#pragma unroll
            for (int u = 0; u < CF_CHUNK; ++u) first[u] = t[u];
        }
    }
    s_seg[w][fl] = sum;
    mk_block_sync();
    const bool live = lf < fc && sum != 0u;                          // (no lane leaves: the row's atom is handed out by readlane below)
    if (mk_ballot(live) == 0ull) return;                             // wave-uniform
    unsigned long long pos = live ? frame_base[lf] + gprefix[(size_t)g * (size_t)fc_pad + (size_t)lf] : 0ull;
    for (int v = 0; v < w; ++v) pos += s_seg[v][fl];
    long long i = i0 + q0 / R4, r = q0 % R4;
    for (long long q = q0; q < q1; q += CF_CHUNK) {
        unsigned t[CF_CHUNK];
        if (q == q0) {
#pragma unroll
            for (int u = 0; u < CF_CHUNK; ++u) t[u] = first[u];
        } else {
#pragma unroll
            for (int u = 0; u < CF_CHUNK; ++u) t[u] = (on && q + u < q1) ? (unsigned)mrow[(size_t)(q + u) * (size_t)fc_pad] : 0u;
        }
#pragma unroll
        for (int u = 0; u < CF_CHUNK; ++u) {
            if (q + u < q1) {                                          // wave-uniform
                const unsigned a = mk_readlane(va, (int)(i - i0));
                unsigned m = live ? t[u] : 0u;
                while (m) {                                            // ascending j = the reference's (i, j) order
                    const int k = __builtin_ctz(m);
                    m &= m - 1u;
                    const long long j = r * CT_RUN + k;
                    out[pos++] = make_uint2(a, staged ? s_sel2[j] : sel2[j]);
                }
                if (++r == R4) { r = 0; ++i; }
            }
        }
    }
}

// Centre of mass of every group in every frame (distance_utils.pyx:160-183): sequential float32
// accumulation in group order.  com has the coords layout [n_groups, 3, F]; lanes along frames.
MK_KERNEL(256) void k_group_com(const float* __restrict__ coords, long long F,
                                const int* __restrict__ g_atoms, const long long* __restrict__ g_off,
                                long long ng, const float* __restrict__ masses, float* __restrict__ com)
{
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    for (long long g = blockIdx.y; g < ng; g += gridDim.y) {        // grid stride over the groups
        float total = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
        for (long long k = g_off[g]; k < g_off[g + 1]; ++k) {
            const size_t a = (size_t)g_atoms[k];
            const float m = masses[a];
            cx = mk_fadd_rn(cx, mk_fmul_rn(coords[(a * 3 + 0) * F + f], m));
            cy = mk_fadd_rn(cy, mk_fmul_rn(coords[(a * 3 + 1) * F + f], m));
            cz = mk_fadd_rn(cz, mk_fmul_rn(coords[(a * 3 + 2) * F + f], m));
            total = mk_fadd_rn(total, m);
        }
        com[((size_t)g * 3 + 0) * F + f] = mk_fdiv_rn(cx, total);
        com[((size_t)g * 3 + 1) * F + f] = mk_fdiv_rn(cy, total);
        com[((size_t)g * 3 + 2) * F + f] = mk_fdiv_rn(cz, total);
    }
}

// Group-pair table of dist_trajectory_reduction (:240-281) / _pairs (:312-350).
MK_KERNEL(256) void k_build_group_pairs(long long ng1, long long ng2, const unsigned* __restrict__ chains1,
                                        const unsigned* __restrict__ chains2, int selfdist, int pairs, int pbc,
                                        unsigned* __restrict__ ga, unsigned* __restrict__ gb,
                                        unsigned* __restrict__ wrap)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ng2) return;
    for (long long i = blockIdx.y; i < ng1; i += gridDim.y) {       // grid stride over the first groups
        long long idx;
        if (pairs) {
            if (j != i) continue;
            idx = i;
        } else if (selfdist) {
            if (j <= i) continue;
            const long long full = i < ng2 ? i : ng2;
            idx = full * (ng2 - 1) - full * (full - 1) / 2 + (j - i - 1);
        } else {
            idx = i * ng2 + j;
        }
        ga[idx] = (unsigned)i; gb[idx] = (unsigned)j;
        wrap[idx] = (pbc && chains1[i] != chains2[j]) ? 1u : 0u;
    }
}

// dist_trajectory_reduction[_pairs]: per (frame, group pair) the minimum squared distance over the atom
// pairs (reduction 0, "closest") or between centres of mass (reduction 1; c1/c2 then point at COM arrays
// and every group counts as one pseudo-atom indexed by its group id).  The reference's update rule
// `if dist2 < mindist or mindist < 0` (mindist starts at -1) is kept verbatim, NaN behaviour included.
// Lanes run along frames; a wave takes every fourth group pair of the tile.  Everything about a group pair is
// wave-uniform (group ids, atom lists: scalar loads), the coordinate rows are scalar bases + the lane's frame offset
// (mk_load_f32_uniform_base), and the second group's atoms go four at a time: twelve loads in flight, then the four
// distances, then the reference's update in ITS order.  (First version: one pair at a time, every load waited for, a
// 64-bit multiply-add per lane and load for the address.)
constexpr int DR_BATCH = 4;

MK_KERNEL(DT_THREADS) void k_dist_reduction(const float* __restrict__ c1, const float* __restrict__ c2,
                                            long long F, const float* __restrict__ box,
                                            const int* __restrict__ g1_atoms, const long long* __restrict__ g1_off,
                                            const int* __restrict__ g2_atoms, const long long* __restrict__ g2_off,
                                            int com1, int com2, const unsigned* __restrict__ ga,
                                            const unsigned* __restrict__ gb, const unsigned* __restrict__ wrap,
                                            long long P, float* __restrict__ out)
{
    __shared__ float tile[DT][DT + 1];
    const long long ptiles = (P + DT - 1) / DT, gt = xcd_contiguous_tile(ptiles * ((F + DT - 1) / DT));   // (as k_dist_pairs)
    if (gt < 0) return;
    const long long f0 = (gt / ptiles) * DT, p0 = (gt % ptiles) * DT;
    {
        const int fl = threadIdx.x & (DT - 1);
        const int pq = (int)mk_uniform(threadIdx.x >> 6);            // the wave's index, as a scalar
        // frames past the end compute on the last frame (the store phase never reads those tile entries)
        const long long f = f0 + fl < F ? f0 + fl : F - 1;
        const unsigned fb = (unsigned)f * 4u;                        // the host refuses F >= 2^30
        const float bx = box[0 * F + f], by = box[1 * F + f], bz = box[2 * F + f];
        const float ibx = mk_fdiv_rn(1.f, bx), iby = mk_fdiv_rn(1.f, by), ibz = mk_fdiv_rn(1.f, bz);
        auto at = [&](const float* __restrict__ c, size_t atom, int ax) {
            return mk_load_f32_uniform_base(c + (atom * 3 + (size_t)ax) * (size_t)F, fb);
        };
        for (int pp = pq; pp < DT; pp += DT_THREADS / DT) {
            const long long p = p0 + pp;
            if (p >= P) break;                                       // wave-uniform
            const long long a = ga[p], b = gb[p];
            const bool w = wrap[p] != 0u;
            const long long i0 = com1 ? a : g1_off[a], i1 = com1 ? a + 1 : g1_off[a + 1];
            const long long j0 = com2 ? b : g2_off[b], j1 = com2 ? b + 1 : g2_off[b + 1];
            float mindist = -1.f;
            for (long long i = i0; i < i1; ++i) {
                const size_t at1 = com1 ? (size_t)i : (size_t)g1_atoms[i];
                const float x1 = at(c1, at1, 0), y1 = at(c1, at1, 1), z1 = at(c1, at1, 2);
                long long j = j0;
                for (; j + DR_BATCH <= j1; j += DR_BATCH) {
                    float B3[DR_BATCH][3], d2[DR_BATCH];
#pragma unroll
                    for (int u = 0; u < DR_BATCH; ++u) {
                        const size_t at2 = com2 ? (size_t)(j + u) : (size_t)g2_atoms[j + u];
                        B3[u][0] = at(c2, at2, 0); B3[u][1] = at(c2, at2, 1); B3[u][2] = at(c2, at2, 2);
                    }
#pragma unroll
                    for (int u = 0; u < DR_BATCH; ++u)
                        d2[u] = dist2_min_image_f32(x1, y1, z1, B3[u][0], B3[u][1], B3[u][2], bx, by, bz, ibx, iby, ibz, w);
#pragma unroll
                    for (int u = 0; u < DR_BATCH; ++u)
                        if (d2[u] < mindist || mindist < 0.f) mindist = d2[u];
                }
                for (; j < j1; ++j) {
                    const size_t at2 = com2 ? (size_t)j : (size_t)g2_atoms[j];
                    const float d2 = dist2_min_image_f32(x1, y1, z1, at(c2, at2, 0), at(c2, at2, 1), at(c2, at2, 2),
                                                         bx, by, bz, ibx, iby, ibz, w);
                    if (d2 < mindist || mindist < 0.f) mindist = d2;
                }
            }
            tile[pp][fl] = mk_fsqrt_rn(mindist);
        }
    }
    mk_block_sync();
    store_tile_rows(tile, f0, p0, F, P, P, out);
}

// dist_trajectory_reduction of FEW frames (distance_utils.pyx:211-281; round 6): the residue-contact map of ONE structure, or of
// a handful of frames.  The kernels above and below run their lanes along frames -- one frame is one lane in 64, and a call of
// 1 to 64 frames costs what 64 frames cost (200 groups of 15 atoms: 0.14 ms whatever F <= 64 is).  Here the lanes run along the
// SECOND groups: a block is (first group a, 256 consecutive second groups, frame f).
//  * The atoms of group a (frame f) are staged in LDS once per block, DRF_CAP at a time (larger groups take several passes), and
//    read back as broadcasts (every lane the same address): nothing about the first group is loaded per atom pair.
//  * A lane walks the atoms of ITS second group (per-lane trip counts; the index of the atom after next and the coordinates of
//    the next one are loaded while this one's pairs are computed).  The first group's atoms lie in LDS two by two, as the packed
//    operands of dist2_pk (the arithmetic of k_dist_reduction_closest below: v_pk_add / v_pk_mul, separately rounded, the image
//    integers behind an ACCUMULATED exactness test): 17 instead of 35 instructions per wrapped pair.  Whether a group pair wraps
//    is per lane here (pbc and different chains): a wave in which any pair wraps walks the wrapping arithmetic with 1 / box = 0
//    in the lanes whose pair does not; the risk is accumulated per (lane, second atom), and a row that fails it is redone with
//    dist2_min_image_f32 and its correctly rounded divisions.  (First version of this kernel: every pair through
//    dist2_min_image_f32 -- 32 us per 200-residue map of one frame, 101 us for 16 frames; profiles/r6_reduction_few_probe.txt.)
//  * The reference's update `if dist2 < mindist or mindist < 0` depends on the order of the pairs only through WHICH pair is first
//    (a NaN there stays; afterwards it is a minimum that ignores NaN): v_min3_f32 over all pairs plus the NaN-ness of (first atom
//    of a, first atom of b) -- component 0 of the first packed pair of the lane's first second atom, as in the kernel below.
//  * No pair table: the result index is computed (the reference's order; selfdist rows are the b > a part), lanes are
//    consecutive results of one frame's row: coalesced stores.
// The centre-of-mass modes come through the same kernel like they do through k_dist_reduction (c1 / c2 are then the COM arrays,
// a group is the one pseudo-atom of its own index).  The pairs mode (a result per group, not per group pair) has no second
// axis to put lanes on and stays with the kernels whose lanes are frames.
constexpr int DRF_THREADS = 256;
constexpr int DRF_CAP = 256;                         // first-group atoms staged per pass
constexpr int DRF_MAX_FRAMES = 16;                   // periodic calls of up to this many frames take this kernel, open ones of up to half as many
                                                     // (measured: profiles/r6_reduction_few_probe.txt)

MK_KERNEL(DRF_THREADS) void k_dist_reduction_few(const float* __restrict__ c1, const float* __restrict__ c2, long long F,
                                                 const float* __restrict__ box,
                                                 const int* __restrict__ g1_atoms, const long long* __restrict__ g1_off, long long ng1,
                                                 const int* __restrict__ g2_atoms, const long long* __restrict__ g2_off, long long ng2,
                                                 int com1, int com2, const unsigned* __restrict__ chains1,
                                                 const unsigned* __restrict__ chains2, int selfdist, int pbc, long long P,
                                                 float* __restrict__ out)
{
    __shared__ float4 s_xy[DRF_CAP / 2];                             // {x, x', y, y'} of two first-group atoms side by side (packed operands)
    __shared__ float2 s_z[DRF_CAP / 2];                              // {z, z'}
    const long long b_lo = (long long)blockIdx.x * DRF_THREADS, b = b_lo + threadIdx.x;
    for (long long f = blockIdx.z; f < F; f += gridDim.z) {          // (grid strides: block-uniform trip counts, the barriers stay uniform)
        const float bx = box[0 * F + f], by = box[1 * F + f], bz = box[2 * F + f];
        const float ibx = mk_fdiv_rn(1.f, bx), iby = mk_fdiv_rn(1.f, by), ibz = mk_fdiv_rn(1.f, bz);
        auto row1 = [&](size_t atom, int ax) { return c1[(atom * 3 + (size_t)ax) * (size_t)F + (size_t)f]; };
        auto row2 = [&](size_t atom, int ax) { return c2[(atom * 3 + (size_t)ax) * (size_t)F + (size_t)f]; };
        for (long long a = blockIdx.y; a < ng1; a += gridDim.y) {
            if (selfdist && b_lo + DRF_THREADS - 1 <= a) continue;   // block-uniform: no second group of this block lies behind a
            const bool live = b < ng2 && (!selfdist || b > a);
            const long long i0 = com1 ? a : g1_off[a], i1 = com1 ? a + 1 : g1_off[a + 1];
            long long j0 = 0, j1 = 0;
            bool w = false;
            if (live) {
                j0 = com2 ? b : g2_off[b]; j1 = com2 ? b + 1 : g2_off[b + 1];
                w = pbc && chains1[a] != chains2[b];
            }
            auto atom1 = [&](long long i) { return com1 ? (size_t)i : (size_t)g1_atoms[i]; };
            auto atom2 = [&](long long j) { return com2 ? (size_t)j : (size_t)g2_atoms[j]; };
            float acc = -1.f;                                        // the reference's `mindist = -1` (:252)
            for (long long ib = i0; ib < i1; ib += DRF_CAP) {        // block-uniform
                const int n = (int)(i1 - ib < DRF_CAP ? i1 - ib : DRF_CAP), nh = (n + 1) / 2;
                mk_block_sync();                                     // the previous pass (or first group) has been read
                for (int k = (int)threadIdx.x; k < nh; k += DRF_THREADS) {
                    const size_t e0 = atom1(ib + 2 * k), e1 = atom1(ib + (2 * k + 1 < n ? 2 * k + 1 : n - 1));   // (padding: the last atom once more)
                    s_xy[k] = make_float4(row1(e0, 0), row1(e1, 0), row1(e0, 1), row1(e1, 1));
                    s_z[k] = make_float2(row1(e0, 2), row1(e1, 2));
                }
                mk_block_sync();
                const bool any_wraps = mk_ballot(w) != 0ull;         // wave-uniform (every lane of the wave is here)
                if (j1 <= j0) continue;                              // (per lane, after the barriers) an empty second group: -1 stays
                // One walk for the whole wave: if ANY of its group pairs wraps, every lane takes the wrapping arithmetic -- a lane whose
                // pair does not wrap with a zero in place of 1 / box (quotient 0, image integer 0, shift b * 0 = 0: the separation
                // unchanged).  (Two walks -- wrapping lanes, then the others -- doubled the loads and the LDS reads of nearly every wave of
                // a multi-chain call: 59 against 32 us per 200-residue map.)
                const float lbx = w ? ibx : 0.f, lby = w ? iby : 0.f, lbz = w ? ibz : 0.f;
                float m = mk_inf(), first = 0.f;
                auto exact_row = [&](float x2, float y2, float z2, bool is_first, float& mj) {     // one second atom, pair by pair
                    mj = mk_inf();
                    for (int k = 0; k < n; ++k) {
                        const float4 A = s_xy[k >> 1]; const float2 Zp = s_z[k >> 1];
                        const float d2 = (k & 1) ? dist2_min_image_f32(A.y, A.w, Zp.y, x2, y2, z2, bx, by, bz, ibx, iby, ibz, w)
                                                 : dist2_min_image_f32(A.x, A.z, Zp.x, x2, y2, z2, bx, by, bz, ibx, iby, ibz, w);
                        if (k == 0 && is_first) first = d2;
                        mj = mk_min_raw(mj, d2);
                    }
                };
                auto sweep = [&](auto wraps_) {
                    constexpr bool WR = decltype(wraps_)::value;
                    size_t at_next = atom2(j0), at_after = j0 + 1 < j1 ? atom2(j0 + 1) : 0;
                    float nx = row2(at_next, 0), ny = row2(at_next, 1), nz = row2(at_next, 2);
                    for (long long j = j0; j < j1; ++j) {
                        const float x2 = nx, y2 = ny, z2 = nz;
                        if (j + 1 < j1) {                            // the next atom's coordinates and the index after it: in flight during this atom's pairs
                            at_next = at_after;
                            if (j + 2 < j1) at_after = atom2(j + 2);
                            nx = row2(at_next, 0); ny = row2(at_next, 1); nz = row2(at_next, 2);
                        }
                        float mj = mk_inf(), risk = 0.f;             // this second atom's minimum and its accumulated image-integer risk
                        int k = 0;
                        if (j == j0) {                               // component 0 of the first packed pair is the reference's first pair
                            const float4 A = s_xy[0]; const float2 Zp = s_z[0];
                            const mk_f2 d2 = dist2_pk<WR>(mk_f2{A.x, A.y}, mk_f2{A.z, A.w}, mk_f2{Zp.x, Zp.y}, x2, y2, z2, bx, by, bz, lbx, lby, lbz, risk);
                            first = d2[0];
                            mj = mk_min3_raw(mj, d2[0], d2[1]);
                            k = 1;
                        }
                        for (; k + 2 <= nh; k += 2) {                // two packed pairs (four atom pairs) per step: four LDS reads in flight
                            const float4 A = s_xy[k], B = s_xy[k + 1]; const float2 Za = s_z[k], Zb = s_z[k + 1];
                            const mk_f2 d2 = dist2_pk<WR>(mk_f2{A.x, A.y}, mk_f2{A.z, A.w}, mk_f2{Za.x, Za.y}, x2, y2, z2, bx, by, bz, lbx, lby, lbz, risk);
                            const mk_f2 e2 = dist2_pk<WR>(mk_f2{B.x, B.y}, mk_f2{B.z, B.w}, mk_f2{Zb.x, Zb.y}, x2, y2, z2, bx, by, bz, lbx, lby, lbz, risk);
                            mj = mk_min3_raw(mj, d2[0], d2[1]);
                            mj = mk_min3_raw(mj, e2[0], e2[1]);
                        }
                        if (k < nh) {
                            const float4 A = s_xy[k]; const float2 Zp = s_z[k];
                            const mk_f2 d2 = dist2_pk<WR>(mk_f2{A.x, A.y}, mk_f2{A.z, A.w}, mk_f2{Zp.x, Zp.y}, x2, y2, z2, bx, by, bz, lbx, lby, lbz, risk);
                            mj = mk_min3_raw(mj, d2[0], d2[1]);
                        }
                        if constexpr (WR) {
                            // an image integer of this atom's pairs may differ from the reference's round(d / b) (a separation within 3e-7 of half
                            // a box length, an infinite quotient): once more, pair by pair, with the per-pair test and the divisions behind it (rare)
                            if (!(risk < DRC_RISK)) exact_row(x2, y2, z2, j == j0, mj);
                        }
                        m = mk_min_raw(m, mj);
                    }
                };
                if (any_wraps) sweep(DistFlag<true>{}); else sweep(DistFlag<false>{});
                if (any_wraps && !w && (first != first || !(m < mk_inf()))) {
                    // a lane that rode along without wrapping and met something that is not a number: 0 * inf is NaN where the reference,
                    // which does not form the quotient for this pair at all, keeps the infinite separation -- its pairs once more, as they are
                    m = mk_inf();
                    for (long long j = j0; j < j1; ++j) {
                        const size_t c = atom2(j);
                        float mj;
                        exact_row(row2(c, 0), row2(c, 1), row2(c, 2), j == j0, mj);
                        m = mk_min_raw(m, mj);
                    }
                }
                // `if dist2 < mindist or mindist < 0` over the passes: the first pass's first pair decides about NaN, later passes join the minimum
                if (acc < 0.f) acc = (first != first) ? first : m;
                else if (acc == acc) acc = mk_min_raw(acc, m);
            }
            if (live) {
                long long idx;
                if (selfdist) {
                    const long long full = a < ng2 ? a : ng2;
                    idx = full * (ng2 - 1) - full * (full - 1) / 2 + (b - a - 1);
                } else {
                    idx = a * ng2 + b;
                }
                out[(size_t)f * (size_t)P + (size_t)idx] = mk_fsqrt_rn(acc);                  // ONE root per (frame, group pair) (:276)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dist_trajectory_reduction[_pairs] with the "closest" reduction on BOTH sides (distance_utils.pyx:211-281, :286-350 with
// reduction1 = reduction2 = 0): the residue-contact maps of MetricDistance -- for every (frame, group pair) the smallest d^2
// over |g1| x |g2| atom pairs, then ONE root.  Round 6.  (The kernel above re-loaded every atom of every atom pair -- three
// loads and ~50 instructions per pair, 2.9 ms for 200 groups of 15 atoms x 512 frames; it stays for the centre-of-mass modes,
// whose group pairs are one or |g| atom pairs.)
//
//  * Lanes run along frames; a wave takes DT / NW CONSECUTIVE group pairs of its tile (the kernel above: every fourth), which
//    in the reference's g1-major order share their first group.  I atoms of that group stay in registers as I/2 packed pairs
//    (mk_f2: x, y, z of two atoms side by side) for the whole stretch; the atoms of the second groups stream past them, one
//    load per (atom, axis) for I atom pairs, a coalesced 256-byte row segment each (frames are the fastest axis of `coords`).
//  * The arithmetic is packed (v_pk_add_f32 / v_pk_mul_f32: two separately rounded float32 operations per lane and
//    instruction -- the reference's roundings): separation, quotient, image shift, squares: 13 packed instructions per TWO
//    pairs, plus three v_rndne_f32 per pair.  Nothing is selected per pair: whether a group pair wraps (pbc and different
//    chains, :225-229) is wave-uniform.
//  * The exactness test of the image integers (round_quotient_exact above) is ACCUMULATED: risk = max over the pairs of
//    (largest |q - rndne(q)| + 3e-7 largest |q|) -- two v_max3 with |x| modifiers, an fma and half a v_max3 per pair, no branch.
//    One wave-uniform test per (block of first atoms, second group); a stretch that fails it (a separation within 3e-7 of half
//    a box length, an infinite quotient) is redone pair by pair with dist2_min_image_f32 and its correctly rounded divisions.
//  * The reference's update `if dist2 < mindist or mindist < 0` (mindist = -1 at first) keeps the FIRST pair's d^2 whatever it
//    is and then every smaller one: the result is NaN when the first pair's is, else the minimum over the pairs that are not
//    NaN.  That is v_min3_f32 (NaN operands are ignored) over all pairs in any order plus the first pair's NaN-ness -- which is
//    component 0 of the first packed pair of the first second atom; the order of the pairs is free, and so is their grouping.
//    Padding slots (a group whose size is not a multiple of I) repeat the group's last atom: the same d^2 once more.
//  * A group with more than I atoms takes several passes over its second groups; the partial minima wait in the wave's own
//    column of the LDS tile (-1 = nothing yet, as the reference starts), where the root is taken at the end.
// ------------------------------------------------------------------------------------------------
constexpr int DRC_WAVES = 8;                         // waves per block: 8 consecutive group pairs each (round 6, measured: 4 waves x 16 pairs
                                                     // left the chip's last round of blocks 43 % full on 19 900 pairs x 512 frames)
template <int I /* first-group atoms in registers: 4 or 8 */, bool SMALL /* every coordinate row ends below 4 GiB: one descriptor */,
          int NW = DRC_WAVES /* waves per block; DT / NW consecutive group pairs per wave */>
MK_KERNEL(NW * WAVE) void k_dist_reduction_closest(const float* __restrict__ coords, long long F, const float* __restrict__ box,
                                                    const int* __restrict__ g1_atoms, const long long* __restrict__ g1_off,
                                                    const int* __restrict__ g2_atoms, const long long* __restrict__ g2_off,
                                                    const unsigned* __restrict__ ga, const unsigned* __restrict__ gb,
                                                    const unsigned* __restrict__ wrap, long long P, float* __restrict__ out)
{
    static_assert(I == 4 || I == 8, "packed pairs of first atoms");
    static_assert(NW == 4 || NW == 8, "waves per block");
    constexpr int H = I / 2, DRC_RUN = DT / NW;
    __shared__ float tile[DT][DT + 1];
    const long long ptiles = (P + DT - 1) / DT, gt = xcd_contiguous_tile(ptiles * ((F + DT - 1) / DT));   // (as k_dist_pairs)
    if (gt < 0) return;
    const long long f0 = (gt / ptiles) * DT, p0 = (gt % ptiles) * DT;
    {
        const int fl = threadIdx.x & (DT - 1);
        const int pq = (int)mk_uniform(threadIdx.x >> 6);            // the wave's index, as a scalar
        const long long f = f0 + fl < F ? f0 + fl : F - 1;           // frames past the end compute on the last frame (never stored)
        const unsigned fb = (unsigned)f * 4u;                        // (the host refuses F >= 2^30)
        const unsigned F4 = (unsigned)F * 4u;
        const float bx = box[0 * F + f], by = box[1 * F + f], bz = box[2 * F + f];
        float ibx = mk_fdiv_rn(1.f, bx), iby = mk_fdiv_rn(1.f, by), ibz = mk_fdiv_rn(1.f, bz);
        mk_keep(ibx); mk_keep(iby); mk_keep(ibz);
        auto at = [&](unsigned atom, int ax) {
            if constexpr (SMALL) return mk_load_f32_base_soffset(coords, (atom * 3u + (unsigned)ax) * F4, fb);
            else return mk_load_f32_uniform_base(coords + ((size_t)atom * 3 + (size_t)ax) * (size_t)F, fb);
        };
        const long long pw = p0 + (long long)pq * DRC_RUN;           // the wave's first group pair
        const int nrun = pw >= P ? 0 : (P - pw < DRC_RUN ? (int)(P - pw) : DRC_RUN);
        for (int k = 0; k < nrun;) {
            // a stretch [k, ke) of group pairs that share their first group
            const unsigned a = ga[pw + k];
            int ke = k + 1;
            while (ke < nrun && ga[pw + ke] == a) ++ke;
            const long long i0 = g1_off[a], i1 = g1_off[a + 1];
            for (int kk = k; kk < ke; ++kk) tile[pq * DRC_RUN + kk][fl] = -1.f;      // the reference's `mindist = -1` (:252)
            for (long long ib = i0; ib < i1; ib += I) {
                mk_f2 ax[H], ay[H], az[H];
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    const long long e0 = ib + 2 * u < i1 ? ib + 2 * u : i1 - 1, e1 = ib + 2 * u + 1 < i1 ? ib + 2 * u + 1 : i1 - 1;
                    const unsigned a0 = (unsigned)g1_atoms[e0], a1 = (unsigned)g1_atoms[e1];
                    ax[u] = mk_f2{at(a0, 0), at(a1, 0)}; ay[u] = mk_f2{at(a0, 1), at(a1, 1)}; az[u] = mk_f2{at(a0, 2), at(a1, 2)};
                }
                const int valid = i1 - ib < I ? (int)(i1 - ib) : I;   // first atoms of this block that are not padding
                for (int kk = k; kk < ke; ++kk) {
                    const unsigned b = gb[pw + kk];
                    const bool w = wrap[pw + kk] != 0u;
                    const long long j0 = g2_off[b], j1 = g2_off[b + 1];
                    if (j1 <= j0) continue;                          // an empty second group: -1 stays (sqrt(-1), as the reference)
                    float m[H], first = 0.f, risk = 0.f;
#pragma unroll
                    for (int u = 0; u < H; ++u) m[u] = mk_inf();
                    auto sweep = [&](auto wraps_) {
                        constexpr bool WR = decltype(wraps_)::value;
                        long long j = j0;
                        {   // the first second atom alone: component 0 of its first packed pair is the reference's first pair
                            const unsigned c = (unsigned)g2_atoms[j];
                            const float x2 = at(c, 0), y2 = at(c, 1), z2 = at(c, 2);
#pragma unroll
                            for (int u = 0; u < H; ++u) {
                                const mk_f2 d2 = dist2_pk<WR>(ax[u], ay[u], az[u], x2, y2, z2, bx, by, bz, ibx, iby, ibz, risk);
                                if (u == 0) first = d2[0];
                                m[u] = mk_min3_raw(m[u], d2[0], d2[1]);
                            }
                            ++j;
                        }
                        // two second atoms per step, the NEXT step's six loads issued before this step's arithmetic (round 6, PMC: with
                        // the loads waited for where they were issued a third of the wave cycles were spent in s_waitcnt, 82 % VALU-busy)
                        // (the two atoms' coordinates side by side in register PAIRS, X = {x of the first, x of the second}: a packed
                        //  instruction broadcasts either half; with six single registers the compiler paired a current value with a
                        //  NEXT one and every use of the pair waited for the next step's load)
                        mk_f2 X = mk_f2_splat(0.f), Y = X, Z = X;
                        if (j + 2 <= j1) {
                            const unsigned c0 = (unsigned)g2_atoms[j], c1 = (unsigned)g2_atoms[j + 1];
                            X = mk_f2{at(c0, 0), at(c1, 0)}; Y = mk_f2{at(c0, 1), at(c1, 1)}; Z = mk_f2{at(c0, 2), at(c1, 2)};
                        }
                        for (; j + 2 <= j1; j += 2) {
                            mk_keep(X); mk_keep(Y); mk_keep(Z);      // (pairs of the SAME step, complete here: see above)
                            // (past the end: the last two atoms once more -- valid addresses, the values are not used)
                            const long long jn = j + 4 <= j1 ? j + 2 : j;
                            const unsigned n0 = (unsigned)g2_atoms[jn], n1 = (unsigned)g2_atoms[jn + 1];
                            const mk_f2 nX = mk_f2{at(n0, 0), at(n1, 0)}, nY = mk_f2{at(n0, 1), at(n1, 1)}, nZ = mk_f2{at(n0, 2), at(n1, 2)};
#pragma unroll
                            for (int u = 0; u < H; ++u) {
                                const mk_f2 d2 = dist2_pk<WR>(ax[u], ay[u], az[u], X[0], Y[0], Z[0], bx, by, bz, ibx, iby, ibz, risk);
                                const mk_f2 e2 = dist2_pk<WR>(ax[u], ay[u], az[u], X[1], Y[1], Z[1], bx, by, bz, ibx, iby, ibz, risk);
                                m[u] = mk_min3_raw(m[u], d2[0], d2[1]);
                                m[u] = mk_min3_raw(m[u], e2[0], e2[1]);
                            }
                            X = nX; Y = nY; Z = nZ;
                        }
                        if (j < j1) {
                            const unsigned c = (unsigned)g2_atoms[j];
                            const float x2 = at(c, 0), y2 = at(c, 1), z2 = at(c, 2);
#pragma unroll
                            for (int u = 0; u < H; ++u) {
                                const mk_f2 d2 = dist2_pk<WR>(ax[u], ay[u], az[u], x2, y2, z2, bx, by, bz, ibx, iby, ibz, risk);
                                m[u] = mk_min3_raw(m[u], d2[0], d2[1]);
                            }
                        }
                    };
                    if (w) sweep(DistFlag<true>{}); else sweep(DistFlag<false>{});
                    float mm = m[0];
#pragma unroll
                    for (int u = 1; u < H; ++u) mm = mk_min_raw(mm, m[u]);
                    if (w && mk_ballot(!(risk < DRC_RISK)) != 0ull) {
                        // an image integer of this stretch may differ from the reference's round(d / b): once more, pair by pair,
                        // with the per-pair test and the correctly rounded divisions behind it (wave-uniform, rare)
                        mk_stay_in_branch();
                        mm = mk_inf();
                        for (int u = 0; u < valid; ++u) {
                            const unsigned a1 = (unsigned)g1_atoms[ib + u];                // (loaded again: no dynamic register index)
                            const float x1 = at(a1, 0), y1 = at(a1, 1), z1 = at(a1, 2);
                            for (long long j = j0; j < j1; ++j) {
                                const unsigned c = (unsigned)g2_atoms[j];
                                const float d2 = dist2_min_image_f32(x1, y1, z1, at(c, 0), at(c, 1), at(c, 2), bx, by, bz, ibx, iby, ibz, true);
                                if (u == 0 && j == j0) first = d2;
                                mm = mk_min_raw(mm, d2);
                            }
                        }
                    }
                    float& acc = tile[pq * DRC_RUN + kk][fl];
                    const float old = acc;
                    if (old < 0.f) acc = (first != first) ? first : mm;          // the first block: the first pair decides about NaN
                    else if (old == old) acc = mk_min_raw(old, mm);              // (a NaN stays: `dist2 < NaN` never holds)
                }
            }
            for (int kk = k; kk < ke; ++kk) {
                float& acc = tile[pq * DRC_RUN + kk][fl];
                acc = mk_fsqrt_rn(acc);                                           // ONE root per (frame, group pair) (:276)
            }
            k = ke;
        }
    }
    mk_block_sync();
    store_tile_rows<NW>(tile, f0, p0, F, P, P, out);
}

// cdist (distance_utils.pyx:355-383): results[i, j]; any dimension D; lanes along j.
MK_KERNEL(256) void k_cdist(const float* __restrict__ c1, long long n1, const float* __restrict__ c2,
                            long long n2, int D, float* __restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n2) return;
    for (long long i = blockIdx.y; i < n1; i += gridDim.y) {        // grid stride over the rows
        float d2 = 0.f;
        for (int k = 0; k < D; ++k) {
            const float diff = mk_fsub_rn(c1[i * D + k], c2[j * D + k]);
            d2 = mk_fadd_rn(d2, mk_fmul_rn(diff, diff));
        }
        out[i * n2 + j] = mk_fsqrt_rn(d2);
    }
}

// pdist (distance_utils.pyx:388-416): condensed upper triangle, row i holds n-1-i entries.
MK_KERNEL(256) void k_pdist(const float* __restrict__ c, long long n, int D, float* __restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    for (long long i = blockIdx.y; i < n && i < j; i += gridDim.y) { // grid stride over the rows (upper triangle: i < j)
        float d2 = 0.f;
        for (int k = 0; k < D; ++k) {
            const float diff = mk_fsub_rn(c[i * D + k], c[j * D + k]);
            d2 = mk_fadd_rn(d2, mk_fmul_rn(diff, diff));
        }
        out[i * (n - 1) - i * (i - 1) / 2 + (j - i - 1)] = mk_fsqrt_rn(d2);
    }
}

// cdist / pdist for the dimensions people call them with (D = 2, 3: coordinates) -- round 6: the kernels above re-load the second point
// for every row and store four bytes per lane: 0.26 / 0.15 of the HBM roofline (8 192^2 x 3 / 16 384 points).  Here a lane keeps FOUR
// consecutive second points in registers and walks CD_ROWS first points (scalar loads: the row is block-uniform), one 16-byte store
// per row and lane (1 KB per wave and row, any alignment: a result row of n2 floats starts wherever it starts).  Same operation order
// per pair as the generic kernels (sequential `dist2 += diff * diff` from 0), the same roots: the same bits.
constexpr int CD_ROWS = 16;            // first points per block
constexpr int CD_JPL = 4;              // second points per lane

template <int D>
MK_DEV void cd_load_points(const float* __restrict__ c2, long long n2, long long j0, float (&p)[CD_JPL][D])
{
#pragma unroll
    for (int u = 0; u < CD_JPL; ++u) {
        const long long j = j0 + u < n2 ? j0 + u : n2 - 1;          // (past the end: the last point once more, never stored)
#pragma unroll
        for (int k = 0; k < D; ++k) p[u][k] = c2[j * D + k];
    }
}

template <int D>
MK_DEV void cd_row(const float* __restrict__ a /* the first point: block-uniform */, const float (&p)[CD_JPL][D], float (&r)[CD_JPL])
{
    float d2[CD_JPL];
#pragma unroll
    for (int u = 0; u < CD_JPL; ++u) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float diff = mk_fsub_rn(a[k], p[u][k]);
            acc = mk_fadd_rn(acc, mk_fmul_rn(diff, diff));
        }
        d2[u] = acc;
    }
    if (mk_ballot(!mk_sqrt_ordinary_all(d2)) == 0ull) {              // (practically always: one wave-uniform test per row)
#pragma unroll
        for (int u = 0; u < CD_JPL; ++u) r[u] = mk_fsqrt_rn_ordinary(d2[u]);
    } else {
#pragma unroll
        for (int u = 0; u < CD_JPL; ++u) r[u] = mk_fsqrt_rn(d2[u]);
    }
}

template <int D>
MK_KERNEL(256) void k_cdist_rows(const float* __restrict__ c1, long long n1, const float* __restrict__ c2, long long n2, float* __restrict__ out)
{
    const long long j0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * CD_JPL;
    const long long i0 = (long long)blockIdx.y * CD_ROWS, i1 = i0 + CD_ROWS < n1 ? i0 + CD_ROWS : n1;
    const long long jw = ((long long)blockIdx.x * blockDim.x + (long long)(threadIdx.x & ~(WAVE - 1))) * CD_JPL;   // the wave's first second point
    if (jw >= n2) return;                                            // wave-uniform (a partly covered wave stays whole: cd_row votes)
    float p[CD_JPL][D];
    cd_load_points<D>(c2, n2, j0 < n2 ? j0 : n2 - 1, p);
    for (long long i = i0; i < i1; ++i) {                            // block-uniform
        float r[CD_JPL];
        cd_row<D>(c1 + i * D, p, r);
        float* __restrict__ o = out + i * n2 + j0;
        if (j0 + CD_JPL <= n2) mk_store_f4_dword_aligned(o, make_float4(r[0], r[1], r[2], r[3]));
        else
#pragma unroll
            for (int u = 0; u < CD_JPL; ++u) if (j0 + u < n2) o[u] = r[u];
    }
}

// the condensed upper triangle: row i holds the n - 1 - i pairs (i, j > i) from offset i (n - 1) - i (i - 1) / 2 on.
// A row starts wherever the rows before it end, so four consecutive pairs of a lane would be stored at any 4-byte offset (first version:
// 0.41 of the roofline against cdist's 0.73 -- every store instruction of a wave straddles one line more and leaves two of them partly
// written).  Here a lane's four pairs are chosen PER ROW so that their 16 bytes are 16-byte aligned in the result: with s = (row offset)
// mod 4 -- the same for the whole block -- lane q takes the second points 4 q - s .. 4 q - s + 3; it keeps the seven points
// 4 q - 3 .. 4 q + 3 in registers and a block-uniform switch picks the four.
template <int D, int S>
MK_DEV void pd_row_shifted(const float* __restrict__ a, const float (&p7)[CD_JPL + 3][D], float (&r)[CD_JPL])
{
    float p[CD_JPL][D];
#pragma unroll
    for (int u = 0; u < CD_JPL; ++u)
#pragma unroll
        for (int k = 0; k < D; ++k) p[u][k] = p7[u + 3 - S][k];
    cd_row<D>(a, p, r);
}

template <int D>
MK_KERNEL(256) void k_pdist_rows(const float* __restrict__ c, long long n, float* __restrict__ out)
{
    const long long qb = (long long)blockIdx.x * blockDim.x;                                           // the block's first quad of second points
    const long long i0 = (long long)blockIdx.y * CD_ROWS, i1 = i0 + CD_ROWS < n ? i0 + CD_ROWS : n;
    if ((qb + blockDim.x) * CD_JPL - 1 <= i0) return;                // block-uniform: entirely on or below the diagonal
    const long long qw = qb + (long long)(threadIdx.x & ~(WAVE - 1));                                  // the wave's first quad
    if (qw * CD_JPL - 3 >= n) return;                                // wave-uniform
    const long long q = qb + threadIdx.x;
    float p7[CD_JPL + 3][D];
#pragma unroll
    for (int u = 0; u < CD_JPL + 3; ++u) {
        long long j = q * CD_JPL - 3 + u;
        j = j < 0 ? 0 : (j < n ? j : n - 1);                          // (outside the list: a valid point, never stored)
#pragma unroll
        for (int k = 0; k < D; ++k) p7[u][k] = c[j * D + k];
    }
    for (long long i = i0; i < i1; ++i) {                            // block-uniform
        const long long row = i * (n - 1) - i * (i - 1) / 2 - i - 1; // out[row + j] is the pair (i, j)
        const int s = (int)(((row % 4) + 4) % 4);                    // block-uniform: (row + 4 q - s) is a multiple of 4
        const long long j0 = q * CD_JPL - s;
        float r[CD_JPL];
        switch (s) {
        case 0: pd_row_shifted<D, 0>(c + i * D, p7, r); break;
        case 1: pd_row_shifted<D, 1>(c + i * D, p7, r); break;
        case 2: pd_row_shifted<D, 2>(c + i * D, p7, r); break;
        default: pd_row_shifted<D, 3>(c + i * D, p7, r); break;
        }
        if (j0 > i && j0 + CD_JPL <= n) *reinterpret_cast<float4*>(out + row + j0) = make_float4(r[0], r[1], r[2], r[3]);   // (16-byte aligned when `out` is)
        else
#pragma unroll
            for (int u = 0; u < CD_JPL; ++u) if (j0 + u > i && j0 + u < n) out[row + j0 + u] = r[u];
    }
}

}  // namespace mkamd
