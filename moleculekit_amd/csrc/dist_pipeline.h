// dist_pipeline.h -- launch sequences for dist_kernels.h (same backend concept as pipeline.h).
#pragma once
#include "dist_kernels.h"
#include "pipeline.h"

#include <vector>

namespace mkamd {

inline long long count_pairs(long long n1, long long n2, int selfdist)
{
    if (!selfdist) return n1 * n2;
    long long s = 0;
    for (long long i = 0; i < n1; ++i) s += (n2 - 1 - i) > 0 ? (n2 - 1 - i) : 0;
    return s;
}

// Which rectangular calls (no selfdist) take the row kernel (k_sel_to_frames + k_dist_rows, dist_kernels.h), and with how many
// second atoms per lane: the largest of 4 / 2 / 1 that keeps >= 80 % of a wave's lanes on real pairs; 0 = the tile kernel
// (short rows, or selections whose frame-major copy would be more than a quarter of the result).
inline int dist_rows_jpl(long long n1, long long n2, long long F)
{
    int jpl = 0;
    for (int c : {4, 2, 1})
        if (!jpl && (double)n2 / (double)(ceil_div(n2, 64 * c) * 64 * c) >= 0.8) jpl = c;
    if (!jpl) return 0;
    const long long np1 = ceil_div(n1, DT) * DT, np2 = ceil_div(n2, 64 * jpl) * 64 * jpl;
    const long long tasks = F * ceil_div(n1, ROWS_CI) * ceil_div(n2, 64 * jpl);
    const bool fits = (np1 + np2) * 3 * 4 <= n1 * n2 && tasks / 4 + 8 <= 0x7ffffff0LL && (np1 + np2) / DT <= 65535 &&
                      ceil_div(F, DT) <= 0x7fffffffLL;
    return fits ? jpl : 0;
}

// dist_trajectory on device pointers (coords [N,3,F], box [3,F], sel/chains uint32) -> out [F, P]
// `avoid`: kernels NOT to take (tests walk every kernel over the same shapes; A-B timing) -- the choice below is made among the rest
// selfdist through the triangular row kernel (measured against the pair-table kernel per frame count and selection size, profiles/r6_dist_self_tri_probe.txt:
// 450 atoms: the pair table at every frame count (12.6 against 14.2 us at one frame); 1 000: the rows up to 32 frames (15 against 35 us at one, 26 against 37 at
// 32; even at 64); 2 000 and 5 000: the rows at every frame count (5 000 atoms: 27 against 624 us at one frame, 767 against 1 621 at 64))
constexpr long long TRI_FEW_FRAMES = 32, TRI_MANY = 700;   // calls of up to 32 frames with selections of at least 700 atoms ...
constexpr long long TRI_BIG = 1500;                        // ... and calls of any length whose selections are at least this long
enum { DIST_AVOID_FRAME = 1, DIST_AVOID_ROWS = 2, DIST_AVOID_RECT = 4, DIST_AVOID_VEC = 8, DIST_PREFER_ROWS = 16, DIST_NO_PACKING = 32 /* capi.hip: host calls */,
       DIST_AVOID_SELF_ROWS = 64 /* selfdist calls of few frames keep the pair-table kernel */,
       DIST_AVOID_SWAPPED = 128 /* short-row calls of few frames keep the tile kernel */ };
template <class BE>
int run_dist_trajectory(BE& be, const float* coords, long long F, const float* box, const unsigned* sel1, long long n1,
                        const unsigned* sel2, long long n2, const unsigned* chains, int selfdist, int pbc, int squared,
                        float* out, std::string& err, int avoid = 0)
{
    if (F < 0 || n1 < 0 || n2 < 0) { err = "negative size"; return ST_EINVAL; }
    const long long P = count_pairs(n1, n2, selfdist);
    if (F == 0 || P == 0) return ST_OK;
    if (P > 0x7fffffffLL * 32) { err = "too many pairs"; return ST_EINVAL; }
    if (F > 0x3fffffffLL) { err = "too many frames (>= 2^30)"; return ST_EINVAL; }
    int st;
    // selfdist through the row kernel's triangular form (k_dist_rows<.., TRI>: lanes along the second atoms, the condensed order written directly) --
    // on ONE structure or a handful of frames, where the pair-table kernel's lanes (frames) are starved (5 000 atoms of one frame: 0.63 ms), and
    // for large selections at any frame count, where its table loads and gathers cost more than the half tiles the triangle wastes
    // (profiles/r6_dist_few_frames_probe.txt)
    const bool tri_rows = selfdist && !(avoid & (DIST_AVOID_SELF_ROWS | DIST_AVOID_ROWS)) && n1 >= 128 && n2 >= 128 && dist_rows_jpl(n1, n2, F) >= 1 &&
                          ((F <= TRI_FEW_FRAMES && n1 >= TRI_MANY && n2 >= TRI_MANY) || (n1 >= TRI_BIG && n2 >= TRI_BIG) || (avoid & DIST_PREFER_ROWS));
    // every sel1 atom against every sel2 atom (no selfdist: the common MetricDistance call): the rectangular kernel -- no pair
    // table, the second atoms of a tile stay in registers while the block walks DR_I first atoms (dist_kernels.h)
    // (both tile kernels: a 1-D grid padded to a multiple of 8, every XCD a contiguous range of tiles -- xcd_contiguous_tile)
    auto padded8 = [](long long tiles) { return (unsigned)(((tiles + 7) / 8) * 8); };
    const bool no_frame = (avoid & DIST_AVOID_FRAME) != 0, no_rows = (avoid & DIST_AVOID_ROWS) != 0, no_rect = (avoid & DIST_AVOID_RECT) != 0;
    // One launch for rectangular calls whose rows are too short for the row kernel (k_dist_frame: a block per frame stages both
    // selections in LDS and walks the pair list in memory order).  (selfdist keeps the pair-table kernel: dist_kernels.h)
    // Which of the three rectangular kernels (round 6, tools/dist_shapes_probe.py over 13 shapes, profiles/r6_dist_crossover.txt):
    //  * the row kernel pays its frame-major pre-pass (13 us) and, with ONE second atom per lane, 4-byte stores: with two or four per lane
    //    it wins wherever it applies; with one only on rows of >= 128 atoms of a result of >= 256 MB (150 x 300 x 2 048: 92 us against the
    //    tile kernel's 119) -- below that the tile kernel does (300 x 60: 37 against 48 us, 5 000 x 60: 0.91 against 1.24 ms);
    //  * rows too short for it go to the block-per-frame kernel (300 x 30: 33 against 36 us), except open calls with full 64-wide tiles and
    //    a small result, where the tile kernel is 15-27 % faster (300 x 100: 75 against 88 us, 100 x 100: 27 against 36).
    // DIST_PREFER_ROWS (tests): the row kernel wherever it applies, as in rounds 4-5.
    const int jpl_any = ((!selfdist || tri_rows) && !no_rows) ? dist_rows_jpl(n1, n2, F) : 0;
    const bool small_result = (double)F * (double)n1 * (double)n2 * 4.0 < 268435456.0;
    // Calls of FEW frames (one structure, a handful of poses): the tile kernel and the block-per-frame kernel run their lanes / blocks along frames
    // (3 000 x 300 atoms, one frame: 37 us in the tile kernel whatever F <= 64 is); the row kernel's lanes run along the second atoms
    // (profiles/r6_dist_few_frames_probe.txt)
    const bool few_frames = F <= 32;
    const bool rows_ok = jpl_any >= 2 || (jpl_any == 1 && ((avoid & DIST_PREFER_ROWS) || few_frames || tri_rows || (n2 >= 128 && !small_result)));
    // few frames, rows too short for the row kernel, a long first selection that the block-per-frame kernel cannot stage (a receptor x a ligand on one
    // frame): the row kernel the other way round, lanes along the FIRST selection, transposed stores (profiles/r6_dist_short_rows_probe.txt: 20 000 x 40
    // atoms of one frame 45 / 86 us in the tile kernel, 11 / 15 for the transposed shape)
    // (results of up to 8 MB: the transposed stores are 4 bytes per 4 n2 -- 20 000 x 40 atoms x EIGHT frames, 26 MB: 90 us this way, 47 in the tile kernel)
    const bool swapped = !selfdist && !no_rows && few_frames && jpl_any == 0 && !(avoid & DIST_AVOID_SWAPPED) && n1 + n2 > 4096 && dist_rows_jpl(n2, n1, F) >= 1 &&
                         (double)F * (double)n1 * (double)n2 * 4.0 <= 8388608.0;
    const int rows_jpl = swapped ? dist_rows_jpl(n2, n1, F) : rows_ok ? jpl_any : 0;
    const bool rect_first = !selfdist && !no_rect && ((jpl_any >= 1 && !rows_ok) || (jpl_any == 0 && !pbc && n2 >= DT && small_result && !no_rows));
    if (!no_frame && !selfdist && rows_jpl == 0 && !rect_first && n1 + n2 <= 4096 && F * 64 <= 0x7ffffff0LL) {
        // four consecutive frames per block while both selections fit 32 KB of LDS (they share the cache lines of their atoms' rows),
        // one beyond; slices of the pair list so that ~1 280 blocks exist, of at least four steps each (measured with 512 ... 4 096:
        // 300 x 30 x 2 048 is flat, 300 x 60 64-65 -> 59-60 us against 2 048 blocks, 1 000 x 30 +25 % beyond 2 048;
        // profiles/r5_dist_frame_blocks.txt)
        const bool four = n1 + n2 <= 512;
        const long long groups = four ? ceil_div(F, 4) : F;
        long long slices = ceil_div(1280, groups);                   // (32 KB of LDS per block: 5 blocks per CU x 256 CUs are resident together)
        const long long most = (P + 4 * DF_STEP - 1) / (4 * DF_STEP);
        slices = slices < 1 ? 1 : (slices > most ? most : slices);
        if (slices > 64) slices = 64;
        const dim3 grid(padded8(groups * slices)), block(DF_THREADS);
        auto go = [&](auto kern) { return be.launch(kern, grid, block, coords, F, box, sel1, n1, sel2, n2, chains, squared, slices, out); };
        const bool big_p = P > 0x3fffffffLL;
        char nm[96];
        snprintf(nm, sizeof nm, "mkamd::k_dist_frame<%s, %d, %d, %s>", pbc ? "true" : "false", four ? 512 : 4096, four ? 4 : 1, big_p ? "long long" : "unsigned int");
        be.note_dist_kernel(nm);
        if (pbc) {
            if (big_p) return four ? go(k_dist_frame<true, 512, 4, long long>) : go(k_dist_frame<true, 4096, 1, long long>);
            return four ? go(k_dist_frame<true, 512, 4, unsigned>) : go(k_dist_frame<true, 4096, 1, unsigned>);
        }
        if (big_p) return four ? go(k_dist_frame<false, 512, 4, long long>) : go(k_dist_frame<false, 4096, 1, long long>);
        return four ? go(k_dist_frame<false, 512, 4, unsigned>) : go(k_dist_frame<false, 4096, 1, unsigned>);
    }
    if ((!selfdist || tri_rows) && !no_rows) {
        // rows of >= 64 second atoms are written directly by a wave per frame, from selections turned frame-major first
        const int jpl = rows_jpl;
        if (jpl) {
            // (swapped: the kernel's rows are the reference's second selection, its lanes the first -- see k_dist_rows)
            const unsigned *rsel = swapped ? sel2 : sel1, *lsel = swapped ? sel1 : sel2;
            const long long rn = swapped ? n2 : n1, ln = swapped ? n1 : n2;
            const long long np1 = ceil_div(rn, DT) * DT, np2 = ceil_div(ln, 64 * jpl) * 64 * jpl;
            const long long tasks = F * ceil_div(rn, ROWS_CI) * ceil_div(ln, 64 * jpl);
            void *t1 = nullptr, *t2 = nullptr, *cs1 = nullptr, *cs2 = nullptr;
            if ((st = be.ensure(WS_D_COM1, (size_t)F * 3 * (size_t)np1 * 4, &t1, 0))) return st;
            if ((st = be.ensure(WS_D_COM2, (size_t)F * 3 * (size_t)np2 * 4, &t2, 0))) return st;
            if ((st = be.ensure(WS_D_PA, (size_t)np1 * 4, &cs1, 0))) return st;
            if ((st = be.ensure(WS_D_PB, (size_t)np2 * 4, &cs2, 0))) return st;
            const unsigned* ch = pbc ? chains : nullptr;
            if ((st = be.launch(k_sel_to_frames, dim3((unsigned)ceil_div(F, DT), (unsigned)((np1 + np2) / DT), 3u), dim3(256), coords, F, rsel, rn, np1,
                                lsel, ln, np2, ch, (float*)t1, (unsigned*)cs1, (float*)t2, (unsigned*)cs2))) return st;
            const dim3 grid(padded8(ceil_div(tasks, 4))), block(256);
            auto go = [&](auto kernel) {
                return be.launch(kernel, grid, block, (const float*)t1, np1, (const unsigned*)cs1, (const float*)t2, np2, (const unsigned*)cs2, box, F,
                                 rn, ln, squared, out, P);
            };
            const bool no_vec = (avoid & DIST_AVOID_VEC) != 0;
            const bool vec = jpl == 4 && !no_vec;      // (16-byte stores at 4-byte alignment: rows of any length, any float* result)
            {
                char nm[96];
                snprintf(nm, sizeof nm, "mkamd::k_sel_to_frames + mkamd::k_dist_rows<%s, %d, %s%s>", pbc ? "true" : "false", jpl, vec ? "true" : "false",
                         selfdist ? ", true" : swapped ? ", false, true" : "");
                be.note_dist_kernel(nm);
            }
            if (swapped) {
                if (pbc) return vec ? go(k_dist_rows<true, 4, true, false, true>) : jpl == 4 ? go(k_dist_rows<true, 4, false, false, true>)
                                    : jpl == 2 ? go(k_dist_rows<true, 2, false, false, true>) : go(k_dist_rows<true, 1, false, false, true>);
                return vec ? go(k_dist_rows<false, 4, true, false, true>) : jpl == 4 ? go(k_dist_rows<false, 4, false, false, true>)
                           : jpl == 2 ? go(k_dist_rows<false, 2, false, false, true>) : go(k_dist_rows<false, 1, false, false, true>);
            }
            if (selfdist) {
                if (pbc) return vec ? go(k_dist_rows<true, 4, true, true>) : jpl == 4 ? go(k_dist_rows<true, 4, false, true>) : jpl == 2 ? go(k_dist_rows<true, 2, false, true>)
                                                                                                                                    : go(k_dist_rows<true, 1, false, true>);
                return vec ? go(k_dist_rows<false, 4, true, true>) : jpl == 4 ? go(k_dist_rows<false, 4, false, true>) : jpl == 2 ? go(k_dist_rows<false, 2, false, true>)
                                                                                                                             : go(k_dist_rows<false, 1, false, true>);
            }
            if (pbc) return vec ? go(k_dist_rows<true, 4, true>) : jpl == 4 ? go(k_dist_rows<true, 4, false>) : jpl == 2 ? go(k_dist_rows<true, 2, false>)
                                                                                                                          : go(k_dist_rows<true, 1, false>);
            return vec ? go(k_dist_rows<false, 4, true>) : jpl == 4 ? go(k_dist_rows<false, 4, false>) : jpl == 2 ? go(k_dist_rows<false, 2, false>)
                                                                                                                   : go(k_dist_rows<false, 1, false>);
        }
    }
    if (!selfdist && !no_rect) {
        const long long tiles = ceil_div(n2, DT) * ceil_div(n1, DR_I) * ceil_div(F, DT);
        if (tiles <= 0x7ffffff0LL) be.note_dist_kernel(pbc ? "mkamd::k_dist_rect<true>" : "mkamd::k_dist_rect<false>");
        if (tiles <= 0x7ffffff0LL)
            return pbc ? be.launch(k_dist_rect<true>, dim3(padded8(tiles)), dim3(DR_WAVES * WAVE), coords, F, box, sel1, n1, sel2, n2, chains, squared, out)
                       : be.launch(k_dist_rect<false>, dim3(padded8(tiles)), dim3(DR_WAVES * WAVE), coords, F, box, sel1, n1, sel2, n2, chains, squared, out);
    }
    void *pa = nullptr, *pb = nullptr, *wr = nullptr;
    if ((st = be.ensure(WS_D_PA, (size_t)P * 4, &pa, 0))) return st;
    if ((st = be.ensure(WS_D_PB, (size_t)P * 4, &pb, 0))) return st;
    if ((st = be.ensure(WS_D_WRAP, (size_t)P * 4, &wr, 0))) return st;
    if ((st = be.launch(k_build_atom_pairs, dim3((unsigned)ceil_div(n2, 256), (unsigned)std::min<long long>(n1, 65535)), dim3(256), sel1, n1, sel2, n2,
                        chains, selfdist, pbc, (unsigned*)pa, (unsigned*)pb, (unsigned*)wr))) return st;
    if (ceil_div(P, DT) * ceil_div(F, DT) > 0x7ffffff0LL) { err = "too many tiles (pairs x frames / 4096 >= 2^31)"; return ST_EINVAL; }
    be.note_dist_kernel(pbc ? "mkamd::k_build_atom_pairs + mkamd::k_dist_pairs<true>" : "mkamd::k_build_atom_pairs + mkamd::k_dist_pairs<false>");
    const dim3 pgrid(padded8(ceil_div(P, DT) * ceil_div(F, DT)));
    return pbc ? be.launch(k_dist_pairs<true>, pgrid, dim3(DT_THREADS), coords, F, box, (const unsigned*)pa, (const unsigned*)pb, (const unsigned*)wr, P, squared, out)
               : be.launch(k_dist_pairs<false>, pgrid, dim3(DT_THREADS), coords, F, box, (const unsigned*)pa, (const unsigned*)pb, (const unsigned*)wr, P, squared, out);
}

// 4 or 8 first-group atoms per pass of k_dist_reduction_closest from the group sizes themselves (host offsets): the padded slot
// totals, 8 unless it pads more than 6 % worse
inline int reduction_block_for(const long long* g1_off, long long ng1)
{
    long long s4 = 0, s8 = 0;
    for (long long g = 0; g < ng1; ++g) {
        const long long n = g1_off[g + 1] - g1_off[g];
        s4 += (n + 3) / 4 * 4; s8 += (n + 7) / 8 * 8;
    }
    return s8 * 100 <= s4 * 106 ? 8 : 4;
}

// dist_trajectory_reduction[_pairs] on device pointers; groups as CSR (atoms int32, offsets int64)
// `n_atoms` (rows of coords) and `n_g1_atoms` (length of g1_atoms) are what the HOST knows about arrays that live on the device:
// they choose the kernel variant (32-bit row offsets; how many first-group atoms a wave keeps in registers), never the result.
// `closest_block`: 0 = choose, 4 / 8 = that many first-group atoms in registers (tests, A-B timing; + 100: blocks of four waves), -1 = the generic kernel.
// `few_frames`: 0 = calls of up to DRF_MAX_FRAMES frames (half as many when not periodic) take k_dist_reduction_few, 1 = every call does (but the pairs mode), -1 = none.
template <class BE>
int run_dist_reduction(BE& be, const float* coords, long long n_atoms, long long F, const float* box, const int* g1_atoms,
                       const long long* g1_off, long long ng1, long long n_g1_atoms, const int* g2_atoms, const long long* g2_off,
                       long long ng2, const unsigned* chains1, const unsigned* chains2, int selfdist, int pairs, int pbc,
                       const float* masses, int reduction1, int reduction2, float* out, std::string& err, int closest_block = 0,
                       int few_frames = 0)
{
    if (F < 0 || ng1 < 0 || ng2 < 0) { err = "negative size"; return ST_EINVAL; }
    if (pairs && ng1 != ng2) { err = "pairs mode needs the same number of groups on both sides"; return ST_EINVAL; }
    if ((reduction1 | reduction2) & ~1) { err = "reduction must be 0 (closest) or 1 (com)"; return ST_EINVAL; }
    const long long P = pairs ? ng1 : count_pairs(ng1, ng2, selfdist);
    if (F == 0 || P == 0) return ST_OK;
    if (F > 0x3fffffffLL) { err = "too many frames (>= 2^30)"; return ST_EINVAL; }
    void *ga = nullptr, *gb = nullptr, *wr = nullptr, *com1 = nullptr, *com2 = nullptr;
    int st;
    // few frames (one structure's residue-contact map): lanes along the second groups, no pair table (k_dist_reduction_few)
    const bool few = !pairs && ceil_div(ng2, DRF_THREADS) <= 0x7fffffffLL && (few_frames > 0 || (few_frames == 0 && F <= (pbc ? DRF_MAX_FRAMES : DRF_MAX_FRAMES / 2)));
    if (!few) {
        if ((st = be.ensure(WS_D_PA, (size_t)P * 4, &ga, 0))) return st;
        if ((st = be.ensure(WS_D_PB, (size_t)P * 4, &gb, 0))) return st;
        if ((st = be.ensure(WS_D_WRAP, (size_t)P * 4, &wr, 0))) return st;
        if ((st = be.launch(k_build_group_pairs, dim3((unsigned)ceil_div(ng2, 256), (unsigned)std::min<long long>(ng1, 65535)), dim3(256), ng1, ng2, chains1,
                            chains2, selfdist, pairs, pbc, (unsigned*)ga, (unsigned*)gb, (unsigned*)wr))) return st;
    }
    const float *c1 = coords, *c2 = coords;
    if (reduction1 == 1) {
        if ((st = be.ensure(WS_D_COM1, (size_t)ng1 * 3 * F * 4, &com1, 0))) return st;
        if ((st = be.launch(k_group_com, dim3((unsigned)ceil_div(F, 256), (unsigned)std::min<long long>(ng1, 65535)), dim3(256), coords, F, g1_atoms, g1_off,
                            ng1, masses, (float*)com1))) return st;
        c1 = (const float*)com1;
    }
    if (reduction2 == 1) {
        if ((st = be.ensure(WS_D_COM2, (size_t)ng2 * 3 * F * 4, &com2, 0))) return st;
        if ((st = be.launch(k_group_com, dim3((unsigned)ceil_div(F, 256), (unsigned)std::min<long long>(ng2, 65535)), dim3(256), coords, F, g2_atoms, g2_off,
                            ng2, masses, (float*)com2))) return st;
        c2 = (const float*)com2;
    }
    if (few)
        return be.launch(k_dist_reduction_few, dim3((unsigned)ceil_div(ng2, DRF_THREADS), (unsigned)std::min<long long>(ng1, 65535),
                                                    (unsigned)std::min<long long>(F, 65535)), dim3(DRF_THREADS), c1, c2, F, box, g1_atoms, g1_off,
                         ng1, g2_atoms, g2_off, ng2, reduction1, reduction2, chains1, chains2, selfdist, pbc, P, out);
    if (ceil_div(P, DT) * ceil_div(F, DT) > 0x7ffffff0LL) { err = "too many tiles (groups pairs x frames / 4096 >= 2^31)"; return ST_EINVAL; }
    const dim3 grid((unsigned)(((ceil_div(P, DT) * ceil_div(F, DT) + 7) / 8) * 8));
    if (reduction1 == 0 && reduction2 == 0 && closest_block >= 0) {
        // closest atom pair of two atom lists -- the residue-contact maps: first-group atoms in registers, packed arithmetic
        // (k_dist_reduction_closest).  Eight atoms per pass when the first groups are large enough to fill them.
        const int blk = closest_block % 100;                         // (+100: four waves of 16 pairs per block -- A-B timing)
        // 4 or 8 first-group atoms per pass: whichever pads the first groups less, 8 on a tie (fewer loads per atom pair: measured
        // 0.42 against 0.54 ms on open pairs, 1.12 against 1.18 ms on periodic ones, 200 groups of 15; groups of 9: 0.56 against 0.67 ms
        // for 4 -- profiles/r6_reduction_probe.txt).  Only the MEAN group size is known here (the offsets live on the device); a
        // caller that has the sizes passes its choice (mkamd_dist_reduction_host does: reduction_block_for).
        const long long mean = ng1 > 0 ? (n_g1_atoms + ng1 - 1) / ng1 : 1;
        const bool eight = blk ? blk == 8 : ceil_div(mean, 8) * 8 * 100 <= ceil_div(mean, 4) * 4 * 106;
        const bool four_waves = closest_block >= 100;       // (sixteen waves of 4 pairs were measured too: 1.31 against 1.12 ms)
        const bool small_rows = (unsigned long long)n_atoms * 3ull * (unsigned long long)F * 4ull <= 0xffffffffull;
        auto go = [&](auto kern, int nw) {
            return be.launch(kern, grid, dim3((unsigned)(nw * WAVE)), coords, F, box, g1_atoms, g1_off, g2_atoms, g2_off, (const unsigned*)ga,
                             (const unsigned*)gb, (const unsigned*)wr, P, out);
        };
        if (four_waves) {
            if (eight) return small_rows ? go(k_dist_reduction_closest<8, true, 4>, 4) : go(k_dist_reduction_closest<8, false, 4>, 4);
            return small_rows ? go(k_dist_reduction_closest<4, true, 4>, 4) : go(k_dist_reduction_closest<4, false, 4>, 4);
        }
        if (eight) return small_rows ? go(k_dist_reduction_closest<8, true>, DRC_WAVES) : go(k_dist_reduction_closest<8, false>, DRC_WAVES);
        return small_rows ? go(k_dist_reduction_closest<4, true>, DRC_WAVES) : go(k_dist_reduction_closest<4, false>, DRC_WAVES);
    }
    return be.launch(k_dist_reduction, grid, dim3(DT_THREADS), c1, c2, F, box,
                     g1_atoms, g1_off, g2_atoms, g2_off, reduction1, reduction2, (const unsigned*)ga, (const unsigned*)gb,
                     (const unsigned*)wr, P, out);
}

// contacts_trajectory / get_collisions on device pointers (dist_kernels.h: count -> scan -> fill per chunk of frames).
// Besides the usual backend concept the backend provides
//   int to_host(void* dst, const void* src_dev, size_t bytes)     copy out and wait for it (and for the kernels before it)
//   int to_device(void* dst_dev, const void* src, size_t bytes)   copy in (stream-ordered)
// `budget_bytes` bounds the per-(pair tile, frame) counters: it decides how many frames go in one chunk.
// frame_offsets [F+1] is a host-side result (the counts have to reach the host to size the list); the (a, b) pairs of a chunk are
// written by the fill kernel where `sink.reserve(n, &dst)` says and then handed over with `sink.commit(dst, n)`:
//   HostPairSink    a chunk lands in the WS_H_OUT workspace and is copied behind the host vector (the "_host" entry point)
//   DevicePairSink  chunks are written behind one another in ONE device buffer that grows by copying (the "_dev" entry point:
//                   the list never leaves the device; one chunk -- a 256 MB counter budget covers 42 000 frames of 100 000
//                   pairs -- is written in place).  The backend provides grow_keep(slot, bytes, keep_bytes, &ptr).
template <class BE>
struct HostPairSink {
    BE& be;
    std::vector<unsigned>& pairs;
    int reserve(size_t n, void** dst) { return be.ensure(WS_H_OUT, n * 8, dst, 0); }
    int commit(void* dst, size_t n)
    {
        const size_t old = pairs.size();
        pairs.resize(old + n * 2);
        return be.to_host(pairs.data() + old, dst, n * 8);
    }
};
template <class BE>
struct DevicePairSink {
    BE& be;
    size_t size = 0;                       // pairs written so far
    void* base = nullptr;
    int reserve(size_t n, void** dst)
    {
        const int st = be.grow_keep(WS_D_CONTACTS, (size + n) * 8, size * 8, &base);
        *dst = static_cast<char*>(base) + size * 8;
        return st;
    }
    int commit(void*, size_t n) { size += n; return 0; }
};
enum { CONTACTS_AVOID_RECT = 1, CONTACTS_AVOID_FEW = 2 };
constexpr long long CONTACTS_FEW_FRAMES = 16;        // calls of at most this many frames count with lanes along the second atoms      // `avoid`: the tests walk both walks over the same shapes; bits 8-15: rows per group of the rectangular walk
template <class BE, class Sink>
int run_contacts(BE& be, const float* coords, long long F, const float* box, const unsigned* sel1, long long n1,
                 const unsigned* sel2, long long n2, const unsigned* chains, int selfdist, int pbc, float dist_threshold,
                 size_t budget_bytes, long long* frame_offsets, Sink&& sink, std::string& err, int avoid = 0)
{
    for (long long f = 0; f <= F; ++f) frame_offsets[f] = 0;
    if (F < 0 || n1 < 0 || n2 < 0) { err = "negative size"; return ST_EINVAL; }
    const long long P = count_pairs(n1, n2, selfdist);
    if (F == 0 || P == 0) return ST_OK;
    if (P >= 0xffffffffLL) { err = "too many atom pairs (>= 2^32); split the selections"; return ST_EINVAL; }
    if (F > 0x3fffffffLL) { err = "too many frames (>= 2^30)"; return ST_EINVAL; }
    void *pa = nullptr, *pb = nullptr, *wr = nullptr, *cnt = nullptr, *tot = nullptr, *base = nullptr, *dout = nullptr, *msk = nullptr;
    int st;
    // Calls whose rows fill at least three eighths of their 64-wide row tiles: second atoms in registers, no pair table
    // (k_contacts_count_rect; a 30-atom ligand: 30 of 64; a single ion: the pair-table walk).  selfdist as well: the rows' bits with
    // j <= i do not count, runs wholly below the diagonal are not computed -- calculate_contacts with sel1 == sel2 (distance.py:364)
    // over a whole protein is millions of pairs, and the pair-table form scans their 64-pair tiles one after the other per 64-frame slab
    const long long JT = ceil_div(n2, DT);
    const bool rect = !(avoid & CONTACTS_AVOID_RECT) && n2 * 8 >= JT * DT * 3 && n1 * JT <= 0x7ffffff0LL;
    if (!rect) {
        if ((st = be.ensure(WS_D_PA, (size_t)P * 4, &pa, 0))) return st;
        if ((st = be.ensure(WS_D_PB, (size_t)P * 4, &pb, 0))) return st;
        if ((st = be.ensure(WS_D_WRAP, (size_t)P * 4, &wr, 0))) return st;
        if ((st = be.launch(k_build_atom_pairs, dim3((unsigned)ceil_div(n2, 256), (unsigned)std::min<long long>(n1, 65535)), dim3(256), sel1, n1, sel2, n2,
                            chains, selfdist, pbc, (unsigned*)pa, (unsigned*)pb, (unsigned*)wr))) return st;
    }
    // frames per chunk: the per-(tile, frame) counters stay within the budget whatever the number of pairs
    // (a rectangular call: counters per GROUP of `ni` first atoms, masks per row tile.  A block of the count kernel walks the ni rows of a
    //  group for its 64 second atoms: the more rows, the better its start -- 48 loads per wave, three divisions, ~0.6 of a row's work -- is
    //  amortised; the fewer, the smaller the last, partly filled round of blocks (four blocks per CU are resident at 121 registers).
    //  ni <= 32 that minimises rounds x (ni + 0.6); measured on 200 x 500 x 2 048: 8 rows 129 us, 25 rows -- two full rounds -- ... )
    const long long slabs_all = ceil_div(F, DT);
    const bool few = rect && F <= CONTACTS_FEW_FRAMES && !(avoid & CONTACTS_AVOID_FEW);
    const long long ni_forced = std::min(32, (avoid >> 8) & 0xff);   // (the tests walk group sizes the small cases would never get; <= 32: a bit per row)
    long long ni = 0;
    if (rect) {
        // (few frames -- get_collisions has one: lanes along the second atoms instead of the frames, k_contacts_count_rect_few; its blocks
        //  are single waves, thirty-two to a CU, one per (group, tile, FRAME))
        const long long slots = (few ? 32LL : 4LL) * std::max(1, be.compute_units());
        double best = 0.0;
        for (long long c = 1; c <= std::min<long long>(32, n1); ++c) {
            long long blocks = ceil_div(n1, c) * JT;                 // per slab (or frame); selfdist: only the tiles that reach beyond a group's first row compute
            if (selfdist) {
                blocks = 0;
                for (long long g = 0; g * c < n1; ++g) blocks += std::max<long long>(0, JT - (g * c < DT - 1 ? 0 : (g * c - (DT - 1)) / DT + 1));
            }
            const double cost = (double)ceil_div(std::max<long long>(1, blocks) * (few ? F : slabs_all), slots) * ((double)c + 0.6);
            if (ni == 0 || cost <= best) { best = cost; ni = c; }     // (ties: the larger group)
        }
        if (ni_forced) ni = ni_forced;
    }
    const long long groups = rect ? ceil_div(n1, ni) : 0;
    const long long tiles = rect ? groups : ceil_div(P, DT);         // what k_contacts_scan runs over
    const long long mask_rows = rect ? n1 * JT * (DT_THREADS / DT) : tiles * (DT_THREADS / DT);
    long long chunk = ((long long)budget_bytes / (tiles * 4 + mask_rows * 2)) / DT * DT;      // 4 B per counter + 2 B per run of 16 pairs, per frame
    chunk = std::max<long long>(DT, std::min<long long>(chunk, (F + DT - 1) / DT * DT));
    chunk = std::min<long long>(chunk, 65535LL * DT);
    const float thr2 = dist_threshold * dist_threshold;              // `float dist_threshold` squared in float (:73)
    if ((st = be.ensure(WS_D_CNT, (size_t)tiles * chunk * 4, &cnt, 0))) return st;
    if ((st = be.ensure(WS_D_MASK, (size_t)mask_rows * chunk * 2, &msk, 0))) return st;
    if ((st = be.ensure(WS_D_TOT, (size_t)chunk * 8, &tot, 0))) return st;
    if ((st = be.ensure(WS_D_BASE, (size_t)chunk * 8, &base, 0))) return st;
    std::vector<unsigned long long> totals((size_t)chunk), bases((size_t)chunk);
    for (long long f0 = 0; f0 < F; f0 += chunk) {
        const long long fc = std::min<long long>(chunk, F - f0), fc_pad = (fc + DT - 1) / DT * DT;
        const dim3 grid((unsigned)tiles, (unsigned)(fc_pad / DT));
        // (the rows of the counters and masks: 64-frame slabs -- lanes are frames -- except in a call of few frames, whose rows hold just its frames:
        //  get_collisions of 60 000 x 3 000 atoms would otherwise read and write 1.4 GB of masks for 22 MB of bits)
        const long long pitch = few ? fc : fc_pad;
        if (rect) {
            const dim3 cgrid((unsigned)(groups * JT), (unsigned)(fc_pad / DT));
            if ((st = be.fill(cnt, 0, (size_t)groups * (size_t)pitch * 4))) return st;
            if (few) {
                const dim3 fgrid((unsigned)(groups * JT), (unsigned)fc);
                st = pbc ? be.launch(k_contacts_count_rect_few<true>, fgrid, dim3(WAVE), coords, F, f0, pitch, box, sel1, n1, sel2, n2, chains, thr2, ni, selfdist,
                                     (unsigned*)cnt, (unsigned short*)msk)
                         : be.launch(k_contacts_count_rect_few<false>, fgrid, dim3(WAVE), coords, F, f0, pitch, box, sel1, n1, sel2, n2, chains, thr2, ni, selfdist,
                                     (unsigned*)cnt, (unsigned short*)msk);
            } else
            st = pbc ? be.launch(k_contacts_count_rect<true>, cgrid, dim3(DT_THREADS), coords, F, f0, fc, fc_pad, box, sel1, n1, sel2, n2, chains, thr2, ni, selfdist,
                                 (unsigned*)cnt, (unsigned short*)msk)
                     : be.launch(k_contacts_count_rect<false>, cgrid, dim3(DT_THREADS), coords, F, f0, fc, fc_pad, box, sel1, n1, sel2, n2, chains, thr2, ni, selfdist,
                                 (unsigned*)cnt, (unsigned short*)msk);
            if (st) return st;
        } else if ((st = be.launch(k_contacts_count, grid, dim3(DT_THREADS), coords, F, f0, fc, fc_pad, box, (const unsigned*)pa, (const unsigned*)pb,
                                   (const unsigned*)wr, P, thr2, (unsigned*)cnt, (unsigned short*)msk))) return st;
        if ((st = be.launch(k_contacts_scan, dim3((unsigned)(fc_pad / DT)), dim3(CS_WAVES * WAVE), (unsigned*)cnt, tiles, pitch, pitch,
                            (unsigned long long*)tot))) return st;
        if ((st = be.to_host(totals.data(), tot, (size_t)fc_pad * 8))) return st;
        unsigned long long run = 0;
        for (long long i = 0; i < fc_pad; ++i) { bases[(size_t)i] = run; run += i < fc ? totals[(size_t)i] : 0ull; }
        for (long long i = 0; i < fc; ++i) frame_offsets[f0 + i + 1] = frame_offsets[f0 + i] + (long long)totals[(size_t)i];
        if (run == 0) continue;
        if ((st = sink.reserve((size_t)run, &dout))) return st;
        if ((st = be.to_device(base, bases.data(), (size_t)fc_pad * 8))) return st;
        if (rect) {
            if ((st = be.launch(k_contacts_fill_rect, grid, dim3(CF_WAVES * WAVE), fc, pitch, sel1, n1, sel2, n2, ni, (const unsigned short*)msk,
                                (const unsigned*)cnt, (const unsigned long long*)base, (uint2*)dout))) return st;
        } else if ((st = be.launch(k_contacts_fill, grid, dim3(DT_THREADS), fc, fc_pad, (const unsigned*)pa, (const unsigned*)pb, (const unsigned short*)msk,
                                   (const unsigned*)cnt, (const unsigned long long*)base, (uint2*)dout))) return st;
        if ((st = sink.commit(dout, (size_t)run))) return st;
    }
    return ST_OK;
}

template <class BE>
int run_cdist(BE& be, const float* c1, long long n1, const float* c2, long long n2, int D, float* out, std::string& err)
{
    if (n1 < 0 || n2 < 0 || D < 0) { err = "negative size"; return ST_EINVAL; }
    if (n1 == 0 || n2 == 0) return ST_OK;
    if ((D == 2 || D == 3) && ceil_div(n1, CD_ROWS) <= 65535) {     // coordinates: four second points per lane in registers, 16-byte stores
        const dim3 grid((unsigned)ceil_div(n2, 256 * CD_JPL), (unsigned)ceil_div(n1, CD_ROWS));
        return D == 3 ? be.launch(k_cdist_rows<3>, grid, dim3(256), c1, n1, c2, n2, out) : be.launch(k_cdist_rows<2>, grid, dim3(256), c1, n1, c2, n2, out);
    }
    return be.launch(k_cdist, dim3((unsigned)ceil_div(n2, 256), (unsigned)std::min<long long>(n1, 65535)), dim3(256), c1, n1, c2, n2, D, out);
}

template <class BE>
int run_pdist(BE& be, const float* c, long long n, int D, float* out, std::string& err)
{
    if (n < 0 || D < 0) { err = "negative size"; return ST_EINVAL; }
    if (n < 2) return ST_OK;
    if ((D == 2 || D == 3) && ceil_div(n, CD_ROWS) <= 65535 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {   // (a result that is 16-byte aligned: the row kernel's stores are)
        const dim3 grid((unsigned)ceil_div(n + 3, 256 * CD_JPL), (unsigned)ceil_div(n, CD_ROWS));
        return D == 3 ? be.launch(k_pdist_rows<3>, grid, dim3(256), c, n, out) : be.launch(k_pdist_rows<2>, grid, dim3(256), c, n, out);
    }
    return be.launch(k_pdist, dim3((unsigned)ceil_div(n, 256), (unsigned)std::min<long long>(n, 65535)), dim3(256), c, n, D, out);
}

}  // namespace mkamd
