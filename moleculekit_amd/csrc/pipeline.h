// pipeline.h -- host-side planning and launch sequences for the kernels in kernels.h.
//
// Written against a small "backend" concept so the exact same planning / launch code drives
//   * the HIP backend in capi.hip (the product: MI355X, one stream per context), and
//   * the host-thread SIMT emulation under tests/emu (test infrastructure, CPU-only boxes).
//
// Backend concept:
//   int  ensure(int slot, size_t bytes, void** ptr)     grow-only workspace buffer
//   int  fill(void* p, int byte, size_t bytes)          async memset on the stream
//   bool pipelining_possible()                           big calls may run their pre-pass beside the previous call's tile kernel
//   int  launch(kernel, dim3 grid, dim3 block, args...) async launch on the stream
//   void hot_begin() / hot_end()                        bracket the tile kernel (event timing)
//   int  acquire_set(bool pipelined)                    pick a workspace set (double-buffered); when
//                                                       pipelined, following launches go to an internal
//                                                       stream that may run beside the previous call's
//                                                       tile kernel
//   void prepass_done(int set)                          back to the caller's stream, which waits for the pre-pass
//   void tile_done(int set)                             the set may be reused once the tile kernel has finished
// ensure() takes the workspace set as its last argument.
#pragma once
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <cstdlib>

namespace mkamd {

constexpr int SOLO_CNT_SHIFT = 3;   // k_bin_solo's cell counters 32 bytes apart (GridDesc::cnt_shift); 16 and 128 bytes measured the same

enum Status { ST_OK = 0, ST_EINVAL = 1, ST_EHIP = 2, ST_ENODEV = 3, ST_EOVERFLOW = 4, ST_EBOX = 5 };

enum WsSlot {
    WS_CELL_COUNT = 0, WS_CELL_START, WS_SCAN_CHUNKS, WS_REC_POS, WS_REC_W, WS_REC_CLS, WS_CLS_TABLE, WS_CLS_BLOCKS, WS_CLS_L1, WS_TMP_POS, WS_TMP_IDX, WS_TMP_CLS, WS_DENSE_LIST, WS_ERR, WS_W_EXPLICIT, WS_DENSE_WORDS, WS_DIRECT_COUNT, WS_REDO_LIST,
    // staging for the "_host" entry points
    WS_H_COORDS, WS_H_SIGMAS, WS_H_OFFSETS, WS_H_ORIGINS, WS_H_BOX, WS_H_OUT, WS_H_CENTERS, WS_H_STAGE,
    // distance_utils row (dist_pipeline.h)
    WS_D_PA, WS_D_PB, WS_D_WRAP, WS_D_COM1, WS_D_COM2, WS_D_SEL1, WS_D_SEL2, WS_D_CHAINS, WS_D_CHAINS2, WS_D_G1A, WS_D_G1O,
    WS_D_G2A, WS_D_G2O, WS_D_MASS, WS_D_CNT, WS_D_TOT, WS_D_BASE, WS_D_CONTACTS, WS_D_MASK,
    WS_NSLOTS
};

constexpr double CUTOFF_A = 5.0;            // occupancy_utils.pyx:53 (d^2 < 25)

// A molecule's topology (round 5; kernels.h "Topology"): everything the pre-pass derives from the sigmas alone, on the device.
// Built once (run_topology_build); a lattice call that brings it voxelizes items that are each ONE set of coordinates of that
// molecule -- the frames of a trajectory -- without touching sigmas, classes or table look-ups again.
struct TopologyDev {
    long long n = 0;                        // atoms of the molecule
    int C = 0, G = 0, sigmas_f64 = 0;
    double voxelsize = 0.0;                 // w = voxelsize^2 / sigma^2: the handle is for one voxel size
    const unsigned* ids = nullptr;          // [n, G] class ids (8 x 4 bits per word)
    const uint2* cw = nullptr;              // [n, G] compact channel words (what k_tail's fix-up waves look at first)
    const void* sigmas = nullptr;           // [n, C] the library's own copy (the exact fix-up recomputes from it)
    const unsigned* table = nullptr;        // [CLS_TABLE_WORDS] the class table
    bool overflow = false;                  // more than NCLS distinct sigmas: no class ids (creation reports it, calls are refused)
    bool wide = false;                      // some sigma is wide enough for the exact cut-off fix-up (GridDesc::w_exact_max)
    const unsigned* wide_list = nullptr;    // [n_wide] the atoms that have one, ascending (k_tail's fix-up jobs of a topology call: item x wide atom)
    unsigned n_wide = 0;
};

struct LatticeProblem {
    int B = 0;
    long long total_atoms = 0;
    int C = 0;
    int sigmas_f64 = 0;
    int nvox[3] = {0, 0, 0};
    double voxelsize = 1.0;
    int pbc = 0;
    int max_images = 1;
    int tile_k = 0;                         // 0 = auto
    int force_general = 0;                  // 1 = never use the class-sorted path
    int lds_tier = -1;                      // -1 = adaptive (choose_tier), else the ECAP_TIER index to use
    int prepass_mode = -1;                  // -1 = automatic, 0 = multi-kernel chain, 1 = one-launch per-item pre-pass (if it fits)
    int fine_cells = 0;                     // 1 = half-cutoff cells (A-B benchmarking, see plan_lattice)
    double value_tol = 0.0;                 // > 0: entries may be dropped where they are worth less than this (tolerance-aware reach)
    int direct = -1;                        // direct binning (k_bin_direct): 1 = whenever the geometry allows; -1 (automatic) and 0 = the chain
    int cell_cap = 0;                       // record slots per cell of the direct layout (0 = 128; tests shrink it to see cells spill)
    unsigned spill_cap = 0;                 // slots of an item's spill area in the direct layout (0 = max(1024, an eighth of the average item))
    unsigned seq = 0;                       // != 0: k_tail reports this number in the host-visible feedback words as it starts (FB_TILES_DONE)
    int tile_team = -1;                     // -1 = automatic (a team of waves per tile when the launch is tiny), 0 = never, 1 = always (4, 8, 16: with that many waves)
    int tile_items = -1;                    // -1 = automatic (a workgroup per item for batches of ligand-sized items), 0 = never, 1 = always
    int exact_redo_list = 0;                // 0 = a topology call with wide atoms hands its exact cut-off hits to k_exact_redo, -1 = k_tail recomputes them in place;
                                            // n > 0 (tests): as 0 with a list of n hits
    // device pointers
    const float* coords = nullptr;
    const long long* atom_offsets = nullptr;
    const void* sigmas = nullptr;
    const double* origins = nullptr;
    const float* box = nullptr;
    const double* affine = nullptr;        // optional [B,12]: rotation (row-major 3x3) + translation per item
    float* out = nullptr;
    const TopologyDev* topo = nullptr;      // every item is topo->n atoms of that molecule (checked on the device: MK_ERR_TOPOLOGY)
};

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Fill the GridDesc for a batch; returns ST_OK or ST_EINVAL with a message.
inline int plan_lattice(const LatticeProblem& P, GridDesc& g, std::string& err)
{
    char buf[256];
    if (P.B < 0 || P.total_atoms < 0 || P.C <= 0) { err = "n_items/total_atoms must be >= 0 and n_channels > 0"; return ST_EINVAL; }
    if (!(P.voxelsize > 0.0) || !std::isfinite(P.voxelsize)) { err = "voxelsize must be a positive finite number"; return ST_EINVAL; }
    for (int ax = 0; ax < 3; ++ax)
        if (P.nvox[ax] < 0) { err = "nvoxels must be >= 0"; return ST_EINVAL; }
    g = GridDesc{};
    g.nx = P.nvox[0]; g.ny = P.nvox[1]; g.nz = P.nvox[2];
    g.V = (long long)g.nx * g.ny * g.nz;
    g.C = P.C; g.G = ceil_div(P.C, CHG);
    g.B = P.B; g.pbc = P.pbc ? 1 : 0; g.force_general = P.force_general ? 1 : 0;
    g.inv_res = 1.0 / P.voxelsize;
    g.w_scale = P.voxelsize * P.voxelsize;
    const double R = CUTOFF_A / P.voxelsize;                 // cutoff in voxel units
    g.R2 = (float)(R * R);
    g.R2cull = (float)(R * R * 1.0002 + 1e-3);
    g.res = P.voxelsize;
    // value step at the cutoff = 1-exp(-(1/(R2 w))^6) > 5e-6  <=>  R2 w < (5e-6)^(-1/6) = 7.647  (sigma > 1.81 A)
    g.w_exact_max = (float)(7.647 / (R * R));
    // 1 - exp(-t^-6) < eps beyond t = w d^2 = eps^(-1/6); off (0) unless the caller opted in.  Capped at 1e-5: the parity bound
    g.cell_cap = 0; g.spill_base = 0u; g.spill_cap = 0u; g.direct_words = nullptr; g.cnt_shift = 0;
    g.reach_tau = (P.value_tol > 0.0) ? (float)std::pow(std::min(P.value_tol, 1e-5), -1.0 / 6.0) : 0.f;
    g.Rp = R + 1e-3;
    g.rint = (int)std::ceil(R);
    if (g.rint > 512) { err = "voxelsize too small (cutoff spans > 512 voxels)"; return ST_EINVAL; }
    // cell edge: the power of two >= the cutoff radius in voxels (1 A grid: 8), so that a tile looks at <= 3 x 3 x 3 cells.
    // `fine_cells` halves it (<= 7 x 7 cell columns, one per lane): the candidates of a tile then hug its rounded box
    // better (at 1 A 7 200 A^3 instead of 27 x 8^3 = 13 824 A^3, against the 3 986 A^3 that survive the exact cull) --
    // measured round 2: VALU instructions of the tile kernel -2.4 %, scalar ones +10 %, time +1.5 % on cfg2 and +5 % on
    // cfg3 (eight times the cell counters), so it is not the default
    g.cs_log2 = 3;
    while ((1 << g.cs_log2) < g.rint && g.cs_log2 < 9) ++g.cs_log2;
    if (g.cs_log2 > 2 && P.fine_cells) --g.cs_log2;
    g.cs = 1 << g.cs_log2;
    g.h = ceil_div((long long)g.rint + 1, g.cs);
    g.ncx = ceil_div(g.nx, g.cs) + 2 * g.h;
    g.ncy = ceil_div(g.ny, g.cs) + 2 * g.h;
    g.ncz = ceil_div(g.nz, g.cs) + 2 * g.h;
    if (g.ncx > 1023 || g.ncy > 1023 || g.ncz > 1023) { err = "grid too large (more than 1023 cells per axis)"; return ST_EINVAL; }
    const long long ncell = (long long)g.ncx * g.ncy * g.ncz;
    if ((ncell + 1) * (long long)(P.B > 0 ? P.B : 1) > 0xFFFF0000LL) {
        snprintf(buf, sizeof buf, "batch too large: %lld cells x %d items exceeds 2^32; split the batch", ncell, P.B);
        err = buf; return ST_EINVAL;
    }
    g.ncell = (int)ncell;
    g.cstride = (int)ncell + 1;
    g.cls_per_item = 0;
    g.prepass_hurry = 1;

    // tile depth: K=8 unless the x extent pads badly or the launch would be too small to fill 256 CUs
    int K = P.tile_k;
    if (K != 4 && K != 8) {
        const long long pad8 = (long long)ceil_div(g.nx, 8) * 8, pad4 = (long long)ceil_div(g.nx, 4) * 4;
        const long long tiles8 = (long long)ceil_div(g.nx, 8) * ceil_div(g.ny, 8) * ceil_div(g.nz, 8) * P.B * g.G;
        K = (pad4 < pad8 || tiles8 < 2048) ? 4 : 8;
    }
    g.K = K;
    g.tnx = ceil_div(g.nx, K); g.tny = ceil_div(g.ny, 8); g.tnz = ceil_div(g.nz, 8);
    const long long ntiles = (long long)g.tnx * g.tny * g.tnz;
    if (ntiles * (long long)(P.B > 0 ? P.B : 1) > 0x7FFFFF00LL) { err = "batch too large: more than 2^31 tiles; split the batch"; return ST_EINVAL; }
    g.ntiles = (int)ntiles;

    long long images = 1;
    if (g.pbc) {
        if (P.max_images < 1) { err = "max_images_per_atom must be >= 1 for periodic items"; return ST_EINVAL; }
        images = P.max_images;
    }
    const long long M = P.total_atoms * images;
    if (M > 0xFFFF0000LL) { err = "batch too large: more than 2^32 atom records; split the batch"; return ST_EINVAL; }
    g.M = (unsigned)(M > 0 ? M : 1);
    g.img_cap = (int)images;
    return ST_OK;
}

// upper bound on periodic images of one atom inside grid+halo, from host-known boxes [B,3] (A)
inline int max_images_from_boxes(const float* box, int B, const int* nvox, double voxelsize, std::string& err)
{
    long long worst = 1;
    for (int b = 0; b < B; ++b) {
        long long m = 1;
        for (int ax = 0; ax < 3; ++ax) {
            const double L = (double)box[3 * b + ax];
            if (!(L > 2.0 * CUTOFF_A)) { err = "periodic box edges must be > 10 A (2 x cutoff)"; return -1; }
            const double span = (double)(nvox[ax] > 0 ? nvox[ax] - 1 : 0) * voxelsize + 2.0 * CUTOFF_A + 2e-3 * voxelsize;
            m *= (long long)std::floor(span / L) + 1;
        }
        if (m > worst) worst = m;
    }
    if (worst > 4096) { err = "periodic box much smaller than the grid (more than 4096 images per atom)"; return -1; }
    return (int)worst;
}

template <class BE>
int run_scan(BE& be, unsigned* counts /* cleared by the last kernel */, size_t n, unsigned* starts /* n+1 */, int set = 0)
{
    const size_t nchunks = (n + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
    void* chunks = nullptr;
    int st = be.ensure(WS_SCAN_CHUNKS, nchunks * sizeof(unsigned), &chunks, set);
    if (st) return st;
    if ((st = be.launch(k_scan_chunk_sums, dim3((unsigned)nchunks), dim3(SCAN_THREADS), (const unsigned*)counts, n, (unsigned*)chunks))) return st;
    if ((st = be.launch(k_scan_sums_inplace, dim3(1), dim3(SCAN_THREADS), (unsigned*)chunks, (unsigned)nchunks))) return st;
    return be.launch(k_scan_finish, dim3((unsigned)nchunks), dim3(SCAN_THREADS), counts, n, (const unsigned*)chunks, starts, (const unsigned*)nullptr);
}

// LDS tier of the tile kernel: forced (0..NTIER-1), or the leanest tier that at most 5 % of the tiles of
// the most recent finished call overflowed (feedback = {tiles over tier 0, 1, 2, tiles} written by the
// dense kernel into host-visible memory; stale or zero feedback only costs speed, never correctness:
// all tiers and the dense path produce bit-identical values).
inline int choose_tier(int forced, const volatile unsigned* feedback)
{
    if (forced >= 0) return forced < NTIER ? forced : NTIER - 1;
    if (!feedback) return 0;
    const unsigned tiles = feedback[NTIER];
    if (tiles == 0) return 0;
    int tier = 0;
    while (tier < NTIER - 1 && (unsigned long long)feedback[tier] * 20ull > tiles) ++tier;
    return tier;
}

enum TileFlavour { TILES_PLAIN = 0, TILES_LEAN = 1, TILES_TEAM = 2, TILES_ITEMS = 3 };

constexpr unsigned REDO_BLOCKS = 4096;       // waves of k_exact_redo (they share the jobs)
constexpr unsigned SHELL_BLOCKS = 16384;     // waves of k_exact_shells (they share the (item, wide atom) jobs)
constexpr unsigned REDO_CAP = 32768;         // hits the list holds (1 MB); a call with more walks its shells once more and recomputes in place (k_exact_shells<.., true>)
// what the call's last launch (k_tail: dense tiles + exact cut-off fix-up) needs besides the tile kernel's arguments
struct TailArgs {
    unsigned dense_wgs = 0, fix_waves = 0, fix_jobs = 0;
    unsigned* other_words = nullptr;
    int per_item = 0;
    const unsigned* summary = nullptr;
    const LatticeProblem* P = nullptr;
    const void* tcls = nullptr;
    int team_waves = 0;                     // waves per tile of the team kernel (0 = by the number of tiles; 4, 8, 16: K = 4 only)
    unsigned* solo_counts = nullptr;        // a call binned by k_bin_solo: its counters (+ control words), zeroed by k_tail
    unsigned solo_n = 0;
    const void* sigmas = nullptr;           // != nullptr: the sigma matrix k_tail recomputes from (a topology call: the handle's copy)
    int sigmas_f64 = 0;
    unsigned* redo_list = nullptr;          // a topology call with wide atoms: k_tail lists its cut-off hits here, k_exact_redo recomputes them
    unsigned redo_cap = 0;
};

template <int K, int T, class BE>
int launch_tiles_tier(BE& be, int flavour, dim3 tgrid, const TailArgs& ta, const GridDesc& g, void* start, void* rpos, void* rw, void* rcls,
                      void* ctab, float* out, unsigned* dcount, void* dlist, void* eflag)
{
    constexpr int E = ECAP_TIER[T];
    int st;
    const bool lean = flavour == TILES_LEAN;
    if (flavour == TILES_ITEMS) {     // batches of ligand-sized items: a workgroup per item, its entries sorted once
        // one workgroup per item when there are enough items to fill the chip (4 096 waves), else several per item
        const long long want_blocks = 1024;
        long long nchunk = (want_blocks + g.B - 1) / g.B;
        const long long max_chunk = (g.ntiles + TILE_TEAM - 1) / TILE_TEAM;
        nchunk = nchunk < 1 ? 1 : (nchunk > max_chunk ? max_chunk : nchunk);
        int tpb = (int)((g.ntiles + nchunk - 1) / nchunk);
        tpb = ((tpb + TILE_TEAM - 1) / TILE_TEAM) * TILE_TEAM;
        const unsigned blocks_per_item = (unsigned)((g.ntiles + tpb - 1) / tpb);
        st = be.launch(k_voxelize_items<K>, dim3((unsigned)g.B * blocks_per_item, (unsigned)g.G), dim3(WAVE * TILE_TEAM), g, (const unsigned*)start,
                       (const float4*)rpos, (const float4*)rw, (const unsigned*)rcls, (const unsigned*)ctab, out, tpb);
    } else if (flavour == TILES_TEAM) {      // a handful of tiles (one grid per call): a team of waves per tile --
        // four when that fills the chip (a 64^3 grid: 1 024 tiles), more for fewer tiles (a pocket: 54 tiles of K = 4)
        auto go = [&](auto kern, int team) {
            return be.launch(kern, tgrid, dim3((unsigned)(WAVE * team)), g, (const unsigned*)start, (const float4*)rpos, (const float4*)rw,
                             (const unsigned*)rcls, (const unsigned*)ctab, out, dcount, (unsigned*)dlist);
        };
        const unsigned long long tw = (unsigned long long)g.B * (unsigned)g.ntiles * (unsigned)g.G;
        // (the 3PTB pocket, 54 tiles: 30.0 us per call with 4 waves per tile, 27.8 with 8, 28.3 with 16; a cfg2 grid, 1 024 tiles:
        //  39.9 with 4, 46.6 with 8)
        const int team = ta.team_waves > 0 ? ta.team_waves : (K == 4 && tw <= 256ull) ? 8 : TILE_TEAM;
        if constexpr (K == 4) {
            st = team == 16 ? go(k_voxelize_tiles_team<K, E, 16>, 16) : team == 8 ? go(k_voxelize_tiles_team<K, E, 8>, 8) : go(k_voxelize_tiles_team<K, E, TILE_TEAM>, TILE_TEAM);
        } else {
            st = go(k_voxelize_tiles_team<K, E, TILE_TEAM>, TILE_TEAM);
        }
    } else if constexpr (T <= 1) {    // the biggest tier is LDS-bound to < 3 waves/SIMD anyway: no lean instance of it
        st = lean ? be.launch(k_voxelize_tiles_lean<K, E>, tgrid, dim3(WAVE), g, (const unsigned*)start, (const float4*)rpos, (const float4*)rw,
                              (const unsigned*)rcls, (const unsigned*)ctab, out, dcount, (unsigned*)dlist)
                  : be.launch(k_voxelize_tiles<K, E>, tgrid, dim3(WAVE), g, (const unsigned*)start, (const float4*)rpos, (const float4*)rw,
                              (const unsigned*)rcls, (const unsigned*)ctab, out, dcount, (unsigned*)dlist);
    } else {
        st = be.launch(k_voxelize_tiles<K, E>, tgrid, dim3(WAVE), g, (const unsigned*)start, (const float4*)rpos, (const float4*)rw,
                       (const unsigned*)rcls, (const unsigned*)ctab, out, dcount, (unsigned*)dlist);
    }
    // the tiles left behind (usually none), the statistics for the next call and the exact cut-off fix-up: one launch
    if (!st && ta.dense_wgs + ta.fix_waves != 0u) {
        const LatticeProblem& P = *ta.P;
        // a topology call with wide atoms: k_tail keeps its dense tiles and bookkeeping, the shells of the wide atoms run in a launch of
        // their own (k_exact_shells: waves without the dense role's LDS footprint), their hits in a third (k_exact_redo)
        const bool split = ta.redo_list != nullptr;
        const unsigned tail_fix_waves = split ? 1u : ta.fix_waves, tail_fix_jobs = split ? 0u : ta.fix_jobs;   // (one wave stays for k_tail's housekeeping)
        auto tail = [&](auto kern, auto* sig) {
            return be.launch(kern, dim3(ta.dense_wgs + tail_fix_waves), dim3(WAVE), g, ta.dense_wgs, (const unsigned*)start, (const float4*)rpos,
                             (const unsigned*)rcls, (const unsigned*)ctab, out, dcount, ta.other_words, (const unsigned*)dlist,
                             g.force_general ? (unsigned*)nullptr : be.feedback_dev(), (const int*)eflag, ta.per_item, ta.summary, P.coords,
                             P.atom_offsets, P.total_atoms, sig, P.origins, P.box, P.affine, (const uint2*)ta.tcls, ta.solo_counts, ta.solo_n,
                             (unsigned*)ctab, g.force_general ? 0u : P.seq, tail_fix_jobs);
        };
        const void* sig = ta.sigmas ? ta.sigmas : P.sigmas;
        const int sig64 = ta.sigmas ? ta.sigmas_f64 : P.sigmas_f64;
        st = sig64 ? tail(k_tail<K, E, double>, (const double*)sig) : tail(k_tail<K, E, float>, (const float*)sig);
        if (!st && split) {
            auto shells = [&](auto kern, auto* sg) {
                return be.launch(kern, dim3(ta.fix_jobs < SHELL_BLOCKS ? ta.fix_jobs : SHELL_BLOCKS), dim3(WAVE), g, ta.fix_jobs, ta.summary, P.coords, P.atom_offsets,
                                 P.total_atoms, sg, P.origins, P.box, P.affine, (const uint2*)ta.tcls, out, ta.redo_list, ta.redo_cap);
            };
            st = sig64 ? shells(k_exact_shells<double, false>, (const double*)sig) : shells(k_exact_shells<float, false>, (const float*)sig);
        }
        if (!st && split) {
            // the listed hits, a wave per (hit, slice of the item's atoms); blocks that find the list empty leave at once
            auto redo = [&](auto kern, auto* sg) {
                return be.launch(kern, dim3(REDO_BLOCKS), dim3(WAVE), g, ta.redo_list, ta.redo_cap, P.coords, P.atom_offsets, sg, P.origins, P.box, P.affine, out);
            };
            st = sig64 ? redo(k_exact_redo<double>, (const double*)sig) : redo(k_exact_redo<float>, (const float*)sig);
        }
        if (!st && split) {
            // the list was full (REDO_CAP hits in one call)?  Then every shell once more, recomputed in place; else these blocks leave at once
            auto again = [&](auto kern, auto* sg) {
                return be.launch(kern, dim3(ta.fix_jobs < 8192u ? ta.fix_jobs : 8192u), dim3(WAVE), g, ta.fix_jobs, ta.summary, P.coords, P.atom_offsets,
                                 P.total_atoms, sg, P.origins, P.box, P.affine, (const uint2*)ta.tcls, out, ta.redo_list, ta.redo_cap);
            };
            st = sig64 ? again(k_exact_shells<double, true>, (const double*)sig) : again(k_exact_shells<float, true>, (const float*)sig);
        }
    }
    return st;
}

template <int K, class BE>
int launch_tiles(BE& be, int tier, int flavour, dim3 tgrid, const TailArgs& ta, const GridDesc& g, void* start, void* rpos, void* rw, void* rcls,
                 void* ctab, float* out, unsigned* dcount, void* dlist, void* eflag)
{
    switch (tier) {
    case 0: return launch_tiles_tier<K, 0>(be, flavour, tgrid, ta, g, start, rpos, rw, rcls, ctab, out, dcount, dlist, eflag);
    case 1: return launch_tiles_tier<K, 1>(be, flavour, tgrid, ta, g, start, rpos, rw, rcls, ctab, out, dcount, dlist, eflag);
    default: return launch_tiles_tier<K, 2>(be, flavour, tgrid, ta, g, start, rpos, rw, rcls, ctab, out, dcount, dlist, eflag);
    }
}

// What a backend remembers about the cell-counter buffer of one workspace set (run_lattice): which allocation it is and
// how many of its leading bytes are known to be zero between calls.
// The dense words (dense-tile list length, tier statistics, k_tail's done counter) live in a buffer of their own, in TWO
// copies that alternate from call to call: a call's last launch clears the copy the NEXT call will use.
// dptr / dclean: the same for the counters of the one-launch pre-pass (k_bin_solo; k_tail zeroes them again); tptr: the
// class-table buffer that holds a valid table (k_bin_solo keeps its table across calls: a new buffer starts empty).
struct CounterState { void* ptr = nullptr; size_t clean = 0; void* wptr = nullptr; bool wclean = false; unsigned parity = 0;
                      void* dptr = nullptr; size_t dclean = 0; void* tptr = nullptr; };
constexpr int DENSE_SET_WORDS = DENSE_WORDS + 2;     // + the done counter of k_tail's dense blocks + its role tickets

// The lattice hot path: bin -> scan -> fill -> tile kernel.  All pointers in P are device pointers.
template <class BE>
int run_lattice(BE& be, const LatticeProblem& P, std::string& err)
{
    GridDesc g;
    int st = plan_lattice(P, g, err);
    if (st) return st;
    if (P.B == 0 || g.V == 0) return ST_OK;

    // Big batches are software-pipelined across calls: the pre-pass (latency / atomic bound) of this call
    // runs on an internal stream beside the tile kernel (VALU bound) of the previous call, on the other
    // workspace set.  Small calls stay in order on the caller's stream (the hand-over costs ~20 us).
    // small items (up to a few thousand atoms, cell grid within the LDS counters): the one-launch per-item pre-pass
    // (short enough that overlapping it with the previous call's tile kernel does not pay: in order, set 0)
    // (when the call can be pipelined -- the caller opted in and the batch is big -- items of more than ~1 000 atoms go to
    //  the kernel chain: its pre-pass then hides behind the previous call's tile kernel, the one-launch one never does;
    //  cfg1 x 4096 = 1 639 atoms per item: 2.00 -> 1.93 ms per step; 60-atom items lose 4 % that way)
    // a topology call (P.topo): the chain with the TOPO binning kernels; what they do not cover -- the general path, the
    // tolerance-aware reach (it needs every atom's smallest w at fill time) -- is refused, not approximated
    const bool topo = P.topo != nullptr;
    if (topo) {
        if (P.topo->overflow || g.force_general || g.reach_tau > 0.f) {
            err = "a topology call takes the class-sorted path only: not with more than 15 distinct sigmas, force_general or a value tolerance (use the plain entry point)";
            return ST_EINVAL;
        }
        if (P.topo->C != P.C || P.topo->voxelsize != P.voxelsize || P.topo->n <= 0 || P.total_atoms != (long long)P.B * P.topo->n) {
            err = "the topology was built for another channel count / voxel size, or the call is not n_items x its atom count long";
            return ST_EINVAL;
        }
    }
    g.topo_n = topo ? P.topo->n : 0;
    g.topo_wide = topo ? P.topo->n_wide : 0u;
    const bool chain_pays = be.pipelining_possible() && P.total_atoms >= 200000 && P.total_atoms > 1024LL * (long long)g.B;
    const unsigned total_tiles = (unsigned)g.B * (unsigned)g.ntiles;
    // fewer tile waves than the chip has SIMDs (one or two 64^3 grids, a pocket): a team of waves per tile
    const bool team = P.tile_team > 0 || (P.tile_team < 0 && (unsigned long long)total_tiles * (unsigned)g.G <= 1024ull);
    // direct layouts (k_bin_direct, k_bin_solo): open boundaries, one channel group, tiles that see at most 63 cells
    auto span = [&](int width) {                     // most cells a tile of `width` voxels (aligned to it) sees along one axis
        int best = 0;
        for (int x0 = 0; x0 < (g.cs > width ? g.cs : width); x0 += width) {
            const int n = ((x0 + width - 1 + g.rint) >> g.cs_log2) - ((x0 - g.rint) >> g.cs_log2) + 1;
            best = n > best ? n : best;
        }
        return best;
    };
    static const int env_cap = [] { const char* e = std::getenv("MKAMD_CELL_CAP"); return e ? std::atoi(e) : 0; }();     // A-B knob
    const int direct_cap = P.cell_cap > 0 ? P.cell_cap : (env_cap > 0 ? env_cap : 128);
    const bool direct_geom = !g.pbc && g.G == 1 && !g.force_general && P.total_atoms > 0 && direct_cap <= (1 << SURV_OFF_BITS) &&
                             span(g.K) * span(8) * span(8) <= WAVE - 1;
    // a SMALL call (the team regime: one molecule per call) takes the one-launch pre-pass k_bin_solo, unless the caller
    // chose a pre-pass (prepass_mode) or it is a ligand-sized call of the workgroup-per-item tile kernel; direct == 2
    // forces it for any size (tests)
    const bool items_sized = P.tile_items != 0 && P.total_atoms <= 96LL * (long long)g.B && g.ntiles <= 512;
    const bool solo = !topo && direct_geom && (unsigned long long)P.total_atoms * (unsigned)g.B <= (1ull << 22) &&
                      (P.direct == 2 || (P.direct != 0 && P.prepass_mode < 0 && team && !(items_sized && P.tile_team <= 0 && g.ncell + 1 <= ITEM_HIST)));
    const bool per_item = !topo && !solo && (g.ncell + 1 <= ITEM_HIST) && P.prepass_mode != 0 &&
                          (P.prepass_mode == 1 || (P.total_atoms <= 4096LL * (long long)g.B && !chain_pays));
    g.cls_per_item = per_item ? 1 : 0;
    const int set = be.acquire_set(P.total_atoms >= 200000 && !per_item && !solo);
    // Issue priority of the binning / fill waves that run beside the previous call's tile kernel.  Raised (s_setprio 3)
    // they take issue slots from the tile waves whenever they are ready; left at 0 they live on the slots the tile
    // kernel leaves idle, which is cheaper (cfg2 +2..4 % at 16..512 grids per step) as long as the chain still
    // finishes inside the tile kernel.  It does when the tile kernel has enough work per atom: measured on cfg2's atoms
    // over smaller grids, the chain is late (-8 %) at 2.2 voxels per atom, in time from 2.8 on; the periodic binning
    // (one wave per SIMD beside the tile kernel, not two) needs the raised priority up to cfg4's 3.7 at least.
    g.prepass_hurry = !(be.set_is_pipelined(set) && !g.pbc && (double)g.B * (double)g.V >= 4.0 * (double)P.total_atoms) ? 1 : 0;
    const size_t ncells = (size_t)g.B * (size_t)g.cstride;
    void *count = nullptr, *start = nullptr, *rpos = nullptr, *rw = nullptr, *rcls = nullptr, *ctab = nullptr, *eflag = nullptr;
    void *tpos = nullptr, *tidx = nullptr, *tcls = nullptr;
    // (when the counters do need a memset its size is a multiple of 256 bytes: an odd tail costs the runtime a second
    //  fill kernel)
    const size_t count_bytes = (ncells * sizeof(unsigned) + 255) & ~(size_t)255;
    void* dwords_all = nullptr;
    if ((st = be.ensure(WS_DENSE_WORDS, 2 * DENSE_SET_WORDS * sizeof(unsigned), &dwords_all, set))) return st;
    if ((st = be.ensure(WS_CELL_COUNT, count_bytes, &count, set))) return st;
    if ((st = be.ensure(WS_CELL_START, (ncells + 1) * sizeof(unsigned), &start, set))) return st;
    // direct binning (k_bin_direct: opt-in, big calls, the chain as its fall-back; k_bin_solo: small calls, nothing behind it)
    void* dcnt = nullptr;
    size_t mrec = (size_t)g.M, dbytes = 0;
    CounterState& cs = be.counter_state(set);
    {
        // spill slots PER ITEM (k_bin_solo: every atom of the call, so that it cannot run out)
        const unsigned spill = solo ? (unsigned)P.total_atoms
                                    : P.spill_cap > 0 ? P.spill_cap : (unsigned)std::max<long long>(1024, P.total_atoms / (8LL * g.B));
        const unsigned long long slots = (unsigned long long)ncells * (unsigned)direct_cap + (unsigned long long)spill * (unsigned)g.B;
        // k_bin_direct for a big call: whenever asked for (1), and by itself (-1) when the call is NOT pipelined -- in order
        // the one-pass form is 3 % faster (the class table of the previous call on the workspace serves; the chain behind
        // it leaves at once), beside the previous call's tile kernel it gains nothing
        const bool direct_big = !topo && !per_item && direct_geom && (P.direct == 1 || (P.direct < 0 && !be.set_is_pipelined(set) && P.total_atoms >= 200000));
        const bool direct = (solo || direct_big) && slots <= 0xFFFF0000ull;
        if (solo && !direct) { err = "internal: the one-launch pre-pass does not fit its record slots"; return ST_EINVAL; }
        if (direct) {
            g.cell_cap = direct_cap; g.spill_base = (unsigned)(ncells * (size_t)direct_cap); g.spill_cap = spill;
            mrec = std::max<size_t>(mrec, (size_t)slots);
            g.cnt_shift = (solo && ncells <= (1u << 16)) ? SOLO_CNT_SHIFT : 0;        // small calls: the counters spread out (see GridDesc)
            dbytes = (((size_t)DIRECT_HEAD + (ncells << g.cnt_shift)) * sizeof(unsigned) + 255) & ~(size_t)255;
            if ((st = be.ensure(WS_DIRECT_COUNT, dbytes, &dcnt, set))) return st;
            if (cs.dptr != dcnt) { cs.dptr = dcnt; cs.dclean = 0; }
            // the direct counters and the control words of this call: zero -- k_tail leaves them so after a solo call
            if (cs.dclean < dbytes && (st = be.fill(dcnt, 0, dbytes))) return st;
            cs.dclean = 0;                                  // (vouched for again once this call has been enqueued in full)
            g.direct_words = (const unsigned*)dcnt;
        }
    }
    if ((st = be.ensure(WS_REC_POS, mrec * sizeof(float4), &rpos, set))) return st;
    if ((st = be.ensure(WS_REC_W, (solo ? mrec : (size_t)g.M) * sizeof(float4) * 2 * g.G, &rw, set))) return st;
    if ((st = be.ensure(WS_REC_CLS, (g.G == 1 ? mrec : (size_t)g.M) * sizeof(unsigned) * g.G, &rcls, set))) return st;
    if ((st = be.ensure(WS_CLS_TABLE, (per_item ? (size_t)g.B : (size_t)1) * CLS_TABLE_WORDS * sizeof(unsigned), &ctab, set))) return st;
    if ((st = be.ensure(WS_ERR, sizeof(int), &eflag, 0))) return st;
    if ((st = be.ensure(WS_TMP_POS, (size_t)g.M * sizeof(float4), &tpos, set))) return st;
    if ((st = be.ensure(WS_TMP_IDX, (size_t)g.M * sizeof(uint2), &tidx, set))) return st;
    if ((st = be.ensure(WS_TMP_CLS, (size_t)(P.total_atoms > 0 ? P.total_atoms : 1) * g.G * sizeof(uint2), &tcls, set))) return st;

    if (solo) {
        g.M = (unsigned)mrec;                       // the record arrays' plane stride (rec_w) is the direct layout's slot count
        if (cs.tptr != ctab) {                      // a table buffer nobody has written yet: k_bin_solo starts from an empty table
            if ((st = be.fill(ctab, 0xff, CLS_TABLE_WORDS * sizeof(unsigned)))) return st;
        }
    }
    void* const table_before = cs.tptr;
    cs.tptr = nullptr;                              // (a call that fails half-way leaves no table behind)
    if (cs.ptr != count) { cs.ptr = count; cs.clean = 0; }
    size_t clean_after = cs.clean;
    cs.clean = 0;                                   // nothing is vouched for until this call has been enqueued in full
    if (cs.wptr != dwords_all || !cs.wclean) {      // new buffer, or a call that failed half-way: both copies from scratch
        if ((st = be.fill(dwords_all, 0, 2 * DENSE_SET_WORDS * sizeof(unsigned)))) return st;
        cs.wptr = dwords_all;
    }
    cs.wclean = false;
    unsigned* const dcount = (unsigned*)dwords_all + cs.parity * DENSE_SET_WORDS;
    unsigned* const dother = (unsigned*)dwords_all + (cs.parity ^ 1u) * DENSE_SET_WORDS;

    const unsigned* fix_summary = nullptr;         // what a fix-up wave of k_tail looks at first (see exact_fixup_block)
    unsigned fix_waves = 0;
    if (per_item) {
        fix_summary = g.force_general ? nullptr : (const unsigned*)ctab;
        fix_waves = (unsigned)g.B;
        unsigned* dwords = dcount;
        auto go = [&](auto kern) {
            // few items: big blocks (latency of the one item matters); many items: small blocks (they fill the chip) --
            // down to ONE wave per item for ligand-sized items (a block's time is a chain of latencies whatever its
            // size, and four times as many blocks are resident: cfg3's 32 768 items 343 -> ~90 us)
            const long long avg = P.total_atoms / (long long)g.B;
            const unsigned threads = (g.B < 512 && avg > 256) ? 1024u : (g.B >= 2048 && avg <= 64) ? 64u : (g.B >= 2048 && avg <= 128) ? 128u : 256u;
            return be.launch(kern, dim3((unsigned)g.B), dim3(threads), g, P.coords, P.atom_offsets, P.sigmas, P.origins, P.box, P.affine,
                             (unsigned*)start, (float4*)tpos, (uint2*)tidx, (uint2*)tcls, (float4*)rpos, (float4*)rw, (unsigned*)rcls,
                             (unsigned*)ctab, dwords, (int*)eflag);
        };
        const int need = g.ncell + 1;
        if (P.sigmas_f64) st = need <= 512 ? go(k_prepass_items<double, 512>) : need <= 2048 ? go(k_prepass_items<double, 2048>) : go(k_prepass_items<double, ITEM_HIST>);
        else              st = need <= 512 ? go(k_prepass_items<float, 512>) : need <= 2048 ? go(k_prepass_items<float, 2048>) : go(k_prepass_items<float, ITEM_HIST>);
        if (st) return st;
    } else {
        // The counters are zero when a call starts and every call leaves them zero (the scan kernels clear what they
        // read): the memset -- a launch of its own, 6 us of a one-grid call -- is only
        // needed for bytes no call has vouched for yet (a new or grown buffer, a call that failed half-way).
        if (clean_after < count_bytes) {
            if ((st = be.fill(count, 0, count_bytes))) return st;
            clean_after = count_bytes;
        }
        const dim3 ablk(256), agrid((unsigned)ceil_div(P.total_atoms > 0 ? P.total_atoms : 1, 256));
        const unsigned nblk = agrid.x, nfblk = (unsigned)ceil_div((long long)g.M, 256);
        // behind a direct pass the chain is a fall-back that usually leaves at once: a few thousand workgroups that share
        // the blocks instead of one each (k_bin_count, k_bin_fill)
        const bool fallback_only = g.direct_words != nullptr && !solo;
        const dim3 cgrid(fallback_only && nblk > 4096u ? 4096u : nblk);
        const dim3 fgrid(fallback_only && nfblk > 4096u ? 4096u : nfblk);
        const unsigned rows_per_block = 128;
        const unsigned nl1 = (nblk + rows_per_block - 1) / rows_per_block;
        void *bsets = nullptr, *l1sets = nullptr;
        if ((st = be.ensure(WS_CLS_BLOCKS, (size_t)nblk * CLS_BLOCK_SET * sizeof(unsigned), &bsets, set))) return st;
        if ((st = be.ensure(WS_CLS_L1, (size_t)nl1 * MERGE_SET * sizeof(unsigned), &l1sets, set))) return st;
        fix_summary = g.force_general ? nullptr : (const unsigned*)bsets;
        fix_waves = P.total_atoms > 0 ? nblk : 0u;
        if (topo) {
            // fix-up jobs per (item, WIDE atom of the molecule) -- the handle lists them -- or none at all.  (Round 5: one job per item;
            // a wave then walked all of a 30 000-atom frame 64 atoms at a time and took its wide atoms one after the other.)
            fix_summary = P.topo->wide_list;
            const unsigned long long jobs = (unsigned long long)g.B * P.topo->n_wide;
            if (jobs > 0xffffffffull) { err = "too many (item, wide atom) fix-up jobs (>= 2^32): split the batch"; return ST_EINVAL; }
            fix_waves = (unsigned)jobs;
        }
        const unsigned* dfail = g.direct_words ? g.direct_words + DIRECT_FAILED : nullptr;
        if (solo) {
            st = P.sigmas_f64 ? be.launch(k_bin_solo<double>, agrid, ablk, g, P.coords, P.atom_offsets, P.total_atoms, (const double*)P.sigmas, P.origins,
                                          P.affine, (unsigned*)dcnt, (float4*)rpos, (float4*)rw, (unsigned*)rcls, (uint2*)tcls, (unsigned*)ctab, (unsigned*)bsets)
                              : be.launch(k_bin_solo<float>, agrid, ablk, g, P.coords, P.atom_offsets, P.total_atoms, (const float*)P.sigmas, P.origins,
                                          P.affine, (unsigned*)dcnt, (float4*)rpos, (float4*)rw, (unsigned*)rcls, (uint2*)tcls, (unsigned*)ctab, (unsigned*)bsets);
            if (st) return st;
        } else {
        if (g.direct_words) {
            // the one-pass form first; the chain below is enqueued behind it and returns at once unless the pass gave up
            st = P.sigmas_f64 ? be.launch(k_bin_direct<double>, agrid, ablk, g, P.coords, P.atom_offsets, P.total_atoms, (const double*)P.sigmas, P.origins,
                                          P.affine, (unsigned*)dcnt, (float4*)rpos, (unsigned*)rcls, (uint2*)tcls, (const unsigned*)ctab, (unsigned*)bsets)
                              : be.launch(k_bin_direct<float>, agrid, ablk, g, P.coords, P.atom_offsets, P.total_atoms, (const float*)P.sigmas, P.origins,
                                          P.affine, (unsigned*)dcnt, (float4*)rpos, (unsigned*)rcls, (uint2*)tcls, (const unsigned*)ctab, (unsigned*)bsets);
            if (st) return st;
        }
        if (P.total_atoms > 0) {
            auto bin = [&](auto kern, auto* sig) {
                return be.launch(kern, cgrid, ablk, g, P.coords, P.atom_offsets, P.total_atoms, sig, P.origins, P.box, P.affine,
                                 (unsigned*)count, (float4*)tpos, (uint2*)tidx, (uint2*)tcls, (unsigned*)bsets, (int*)eflag, nblk);
            };
            if (topo) {                                      // (the ids stand where the sigmas would: bin_atom<.., TOPO>)
                auto tbin = [&](auto kern) {
                    return be.launch(kern, cgrid, ablk, g, P.coords, P.atom_offsets, P.total_atoms, (const float*)P.topo->ids, P.origins, P.box, P.affine,
                                     (unsigned*)count, (float4*)tpos, (uint2*)tidx, (uint2*)tcls, (unsigned*)bsets, (int*)eflag, nblk);
                };
                st = g.pbc ? tbin(k_bin_count<float, 1, false, true>) : tbin(k_bin_count<float, 0, false, true>);
            } else if (fallback_only) {                             // (open boundaries: the direct layouts have no periodic form)
                st = P.sigmas_f64 ? bin(k_bin_count<double, 0, true>, (const double*)P.sigmas) : bin(k_bin_count<float, 0, true>, (const float*)P.sigmas);
            } else if (P.sigmas_f64) st = g.pbc ? bin(k_bin_count<double, 1>, (const double*)P.sigmas) : bin(k_bin_count<double, 0>, (const double*)P.sigmas);
            else              st = g.pbc ? bin(k_bin_count<float, 1>, (const float*)P.sigmas) : bin(k_bin_count<float, 0>, (const float*)P.sigmas);
            if (st) return st;
        }
        // sigma classes (per-block sets -> class table) and the scan of the cell counts, fused two launches deep
        const bool do_classes = P.total_atoms > 0 && !g.force_general && !topo;
        if (!do_classes && !topo && (st = be.fill(ctab, 0xff, CLS_TABLE_WORDS * sizeof(unsigned)))) return st;   // nothing to register
        if (!g.direct_words && ncells <= SMALL_PREPASS_MAX_CELLS && nblk <= SMALL_PREPASS_MAX_BLOCKS) {
            // a small call (one grid): one launch instead of three dependent ones
            if ((st = be.launch(k_prepass_small, dim3(1), dim3(SMALL_PREPASS_THREADS), (const unsigned*)bsets, do_classes ? nblk : 0u,
                                (unsigned*)ctab, (unsigned*)count, (unsigned)ncells, (unsigned*)start))) return st;
        } else {
            const size_t nchunks = (ncells + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
            void* chunks = nullptr;
            if ((st = be.ensure(WS_SCAN_CHUNKS, nchunks * sizeof(unsigned), &chunks, set))) return st;
            const unsigned nl1_eff = do_classes ? nl1 : 0u;
            if ((st = be.launch(k_prepass_reduce1, dim3(nl1_eff + (unsigned)nchunks), dim3(256), (const unsigned*)bsets, nblk, rows_per_block,
                                nl1_eff, (unsigned*)l1sets, (const unsigned*)count, ncells, (unsigned*)chunks, dfail))) return st;
            if ((st = be.launch(k_prepass_reduce2, dim3(do_classes ? 2u : 1u), dim3(256), (const unsigned*)l1sets, nl1_eff, (unsigned*)ctab,
                                (unsigned*)chunks, (unsigned)nchunks, dfail))) return st;
            if ((st = be.launch(k_scan_finish, dim3((unsigned)nchunks), dim3(SCAN_THREADS), (unsigned*)count, ncells,
                                (const unsigned*)chunks, (unsigned*)start, dfail))) return st;
        }
        if (P.total_atoms > 0) {
            // (FOUR temp slots per thread with their loads in flight together -- for calls that run alone on the chip, where the
            //  48-register budget does not apply -- were measured: 261 us against 212, the pass is bound by its scattered
            //  stores, not by the round trips in front of them)
            auto fill = [&](auto kern, auto* sig) {
                return be.launch(kern, fgrid, ablk, g, sig, (const unsigned*)start, (const float4*)tpos, (const uint2*)tidx, (const uint2*)tcls,
                                 (float4*)rpos, (float4*)rw, (unsigned*)rcls, (const unsigned*)ctab, nfblk);
            };
            if (topo) st = fill(k_bin_fill<float, false, true>, (const float*)nullptr);
            else if (fallback_only) st = P.sigmas_f64 ? fill(k_bin_fill<double, true>, (const double*)P.sigmas) : fill(k_bin_fill<float, true>, (const float*)P.sigmas);
            else               st = P.sigmas_f64 ? fill(k_bin_fill<double>, (const double*)P.sigmas) : fill(k_bin_fill<float>, (const float*)P.sigmas);
            if (st) return st;
        }
        }                                           // (not solo)
    }
    be.prepass_done(set);

    const dim3 tgrid(((total_tiles + 7u) / 8u) * 8u, (unsigned)g.G);
    if ((unsigned long long)total_tiles * (unsigned)g.G > 0xFFFF0000ull) { err = "batch too large: more than 2^32 tiles x channel groups; split the batch"; return ST_EINVAL; }
    void* dlist = nullptr;
    if ((st = be.ensure(WS_DENSE_LIST, (size_t)total_tiles * g.G * sizeof(unsigned), &dlist, set))) return st;
    const int tier = choose_tier(P.lds_tier, be.feedback_host());
    int flavour = team ? TILES_TEAM : (be.set_is_pipelined(set) ? TILES_LEAN : TILES_PLAIN);   // lean: leave registers for the next call's pre-pass
    // many ligand-sized items (cfg3, cfg5): a workgroup per item sorts its entries once for all its tiles
    if (P.tile_items != 0 && ((P.tile_items > 0 && !team) || (P.tile_items < 0 && P.tile_team <= 0 && per_item && P.total_atoms <= 96LL * (long long)g.B && g.ntiles <= 512)))
        flavour = TILES_ITEMS;
    TailArgs ta;
    // (the general path has no dense tiles; its fix-up waves still run, and its statistics stay what they were)
    ta.dense_wgs = g.force_general ? 0u : (total_tiles * (unsigned)g.G < 4096u ? total_tiles * (unsigned)g.G : 4096u);
    ta.fix_jobs = fix_waves;
    ta.fix_waves = fix_waves < 8192u ? fix_waves : 8192u;         // (the fix-up waves share the jobs: see k_tail)
    ta.other_words = dother; ta.per_item = topo ? 2 : per_item ? 1 : 0; ta.summary = fix_summary; ta.P = &P; ta.tcls = topo ? (const void*)P.topo->cw : tcls;
    if (topo) { ctab = const_cast<unsigned*>(P.topo->table); ta.sigmas = P.topo->sigmas; ta.sigmas_f64 = P.topo->sigmas_f64; }
    ta.team_waves = (P.tile_team == 4 || P.tile_team == 8 || P.tile_team == 16) ? P.tile_team : 0;
    if (solo) { ta.solo_counts = (unsigned*)dcnt; ta.solo_n = (unsigned)(DIRECT_HEAD + (ncells << g.cnt_shift)); }
    if (topo && g.topo_wide != 0u && P.seq == 0u && !g.force_general && P.exact_redo_list >= 0) {
        // a trajectory of a molecule with wide sigmas (ions): the exact recomputes of k_tail's hits are spread over many waves (k_exact_redo)
        void* rl = nullptr;
        if ((st = be.ensure(WS_REDO_LIST, (size_t)(REDO_HEAD + (size_t)REDO_CAP * REDO_ENTRY) * sizeof(unsigned), &rl, set))) return st;
        // (the list's counter back to zero: queued behind the previous call's k_exact_redo on this stream, in front of this call's hot kernels)
        if ((st = be.launch(k_zero_words, dim3(1), dim3(WAVE), (unsigned*)rl, (unsigned)REDO_HEAD))) return st;
        ta.redo_list = (unsigned*)rl;
        ta.redo_cap = P.exact_redo_list > 0 && (unsigned)P.exact_redo_list < REDO_CAP ? (unsigned)P.exact_redo_list : REDO_CAP;   // (a tiny list: tests of the overflow pass)
    }
    be.hot_begin(flavour, g.K, ECAP_TIER[tier]);
    st = g.K == 8 ? launch_tiles<8>(be, tier, flavour, tgrid, ta, g, start, rpos, rw, rcls, ctab, P.out, dcount, dlist, eflag)
                  : launch_tiles<4>(be, tier, flavour, tgrid, ta, g, start, rpos, rw, rcls, ctab, P.out, dcount, dlist, eflag);
    be.hot_end();
    if (!st) {
        cs.clean = clean_after;
        cs.wclean = true;
        cs.tptr = topo ? table_before : ctab;       // every pre-pass leaves a whole table in the buffer (a topology call: untouched)
        if (solo) cs.dclean = dbytes;               // k_tail has zeroed them
        if (ta.dense_wgs + ta.fix_waves != 0u) cs.parity ^= 1u;       // k_tail has cleared the other copy: the next call's
    }
    be.note_error_flag_mirrored(!st && !g.force_general && ta.dense_wgs != 0u && be.feedback_dev() != nullptr);
    be.note_tail_reports(!st && !g.force_general && ta.dense_wgs + ta.fix_waves != 0u && be.feedback_dev() != nullptr && P.seq != 0u);
    be.tile_done(set);
    return st;
}

// Build a molecule's topology into caller-provided device buffers (cw [n, G] uint2, ids [n, G], table [CLS_TABLE_WORDS],
// flags [1] int, zeroed by the caller); `d_sigmas` is the library's own copy.  The caller reads table[CLS_OVERFLOW] and
// flags[0] back once the stream has drained.  Workspace: the class-set slots of set 0.
template <class BE>
int run_topology_build(BE& be, const void* d_sigmas, int sigmas_f64, long long n, int C, double voxelsize, uint2* cw, unsigned* ids,
                       unsigned* table, int* flags /* 2 words, zeroed */, unsigned* wide_list /* [n] */, std::string& err)
{
    if (n <= 0 || C <= 0) { err = "a topology needs n_atoms > 0 and n_channels > 0"; return ST_EINVAL; }
    if (!(voxelsize > 0.0) || !std::isfinite(voxelsize)) { err = "voxelsize must be a positive finite number"; return ST_EINVAL; }
    const int G = ceil_div(C, CHG);
    const double w_scale = voxelsize * voxelsize, R = CUTOFF_A / voxelsize;
    const float w_exact_max = (float)(7.647 / (R * R));                     // plan_lattice's rule
    const unsigned nblk = (unsigned)ceil_div(n, 256), rows_per_block = 128, nl1 = (nblk + rows_per_block - 1) / rows_per_block;
    void *bsets = nullptr, *l1sets = nullptr;
    int st;
    if ((st = be.ensure(WS_CLS_BLOCKS, (size_t)nblk * CLS_BLOCK_SET * sizeof(unsigned), &bsets, 0))) return st;
    if ((st = be.ensure(WS_CLS_L1, (size_t)nl1 * MERGE_SET * sizeof(unsigned), &l1sets, 0))) return st;
    st = sigmas_f64 ? be.launch(k_topology_classes<double>, dim3(nblk), dim3(256), (const double*)d_sigmas, n, C, G, w_scale, cw, (unsigned*)bsets)
                    : be.launch(k_topology_classes<float>, dim3(nblk), dim3(256), (const float*)d_sigmas, n, C, G, w_scale, cw, (unsigned*)bsets);
    if (st) return st;
    if ((st = be.launch(k_merge_classes, dim3(nl1), dim3(256), (const unsigned*)bsets, nblk, (unsigned)CLS_BLOCK_SET, rows_per_block,
                        (unsigned*)l1sets, (unsigned*)nullptr))) return st;
    if ((st = be.launch(k_merge_classes, dim3(1), dim3(256), (const unsigned*)l1sets, nl1, (unsigned)MERGE_SET, nl1, (unsigned*)nullptr, table))) return st;
    return sigmas_f64 ? be.launch(k_topology_ids<double>, dim3(nblk), dim3(256), (const double*)d_sigmas, (const uint2*)cw, (const unsigned*)table, n, C, G,
                                  w_scale, w_exact_max, ids, flags, wide_list)
                      : be.launch(k_topology_ids<float>, dim3(nblk), dim3(256), (const float*)d_sigmas, (const uint2*)cw, (const unsigned*)table, n, C, G,
                                  w_scale, w_exact_max, ids, flags, wide_list);
}

// Explicit centres: sigma -> w, then the brute-force double-precision kernel.
template <class BE>
int run_centers(BE& be, const double* d_centers, long long V, const float* d_coords, long long N,
                const void* d_sigmas, int sigmas_f64, int C, const double* box_host, float* d_out,
                std::string& err)
{
    if (V < 0 || N < 0 || C <= 0) { err = "n_centers/n_atoms must be >= 0 and n_channels > 0"; return ST_EINVAL; }
    if (V == 0) return ST_OK;
    if (box_host)
        for (int ax = 0; ax < 3; ++ax)
            if (!(box_host[ax] > 2.0 * CUTOFF_A)) { err = "periodic box edges must be > 10 A (2 x cutoff)"; return ST_EBOX; }
    const int G = ceil_div(C, CHG);
    void* w = nullptr;
    int st = be.ensure(WS_W_EXPLICIT, (size_t)(N > 0 ? N : 1) * sizeof(float4) * 2 * G, &w, 0);
    if (st) return st;
    if (N > 0) {
        const dim3 blk(256), grid((unsigned)ceil_div(N, 256));
        st = sigmas_f64 ? be.launch(k_sigma_to_w<double>, grid, blk, (const double*)d_sigmas, N, C, G, 1.0, (float4*)w)
                        : be.launch(k_sigma_to_w<float>, grid, blk, (const float*)d_sigmas, N, C, G, 1.0, (float4*)w);
        if (st) return st;
    }
    // 64 centres per workgroup; its waves split the atoms: as many (4, 8, 16) as it takes to put ~4 waves on every SIMD
    const long long wgs = ceil_div(V, EXPL_CENTERS) * G;
    int waves = 4;
    while (waves < EXPL_MAX_WAVES && wgs * waves < 4096) waves *= 2;
    const dim3 grid((unsigned)ceil_div(V, EXPL_CENTERS), (unsigned)G), blk((unsigned)(waves * WAVE));
    return be.launch(k_occupancy_centers, grid, blk, d_centers, V, d_coords, N, (const float4*)w, C,
                     box_host ? 1 : 0, box_host ? box_host[0] : 0.0, box_host ? box_host[1] : 0.0,
                     box_host ? box_host[2] : 0.0, d_out);
}

template <class BE>
int run_grid_centers(BE& be, const double* bb_min, const int* nvox, double voxelsize, double* d_centers,
                     std::string& err)
{
    if (nvox[0] < 0 || nvox[1] < 0 || nvox[2] < 0) { err = "nvoxels must be >= 0"; return ST_EINVAL; }
    const long long V = (long long)nvox[0] * nvox[1] * nvox[2];
    if (V == 0) return ST_OK;
    return be.launch(k_grid_centers, dim3((unsigned)ceil_div(V, 256)), dim3(256), bb_min[0], bb_min[1],
                     bb_min[2], nvox[0], nvox[1], nvox[2], voxelsize, d_centers);
}

// Is this centre list a getCenters lattice (x slowest, z fastest, one positive step)?  The test of
// moleculekit_amd/voxeldescriptors.py::_recognise_lattice_numpy in two passes over the array instead of a dozen numpy
// temporaries: axis lengths from the first place z (then y, at stride nz) stops increasing, one common positive step,
// then every centre against fl64(index * step) + centre 0 with the tolerance 1e-9 * max(1, max |c|).  Host code.
inline bool lattice_from_centers(const double* c, long long V, double* bb_min, int* nvoxels, double* voxelsize)
{
    if (!c || !bb_min || !nvoxels || !voxelsize || V < 2) return false;
    long long nz = V;
    for (long long i = 1; i < V; ++i) if (c[3 * i + 2] <= c[3 * (i - 1) + 2]) { nz = i; break; }
    if (V % nz) return false;
    const long long rows = V / nz;
    long long ny = rows;
    for (long long i = 1; i < rows; ++i) if (c[3 * (i * nz) + 1] <= c[3 * ((i - 1) * nz) + 1]) { ny = i; break; }
    if (rows % ny) return false;
    const long long nx = rows / ny;
    if (nx > 0x7fffffff || ny > 0x7fffffff || nz > 0x7fffffff) return false;
    double steps[3]; int ns = 0;
    if (nz > 1) steps[ns++] = c[3 * 1 + 2] - c[2];
    if (ny > 1) steps[ns++] = c[3 * nz + 1] - c[1];
    if (nx > 1) steps[ns++] = c[3 * (nz * ny) + 0] - c[0];
    if (ns == 0) return false;
    for (int i = 0; i < ns; ++i) if (!(steps[i] > 0.0)) return false;
    const double vs = steps[0];
    const double stol = 1e-9 * std::max(1.0, std::fabs(vs));
    for (int i = 0; i < ns; ++i) if (std::fabs(steps[i] - vs) > stol) return false;
    double maxabs = 0.0;
    for (long long i = 0; i < 3 * V; ++i) {
        const double a = std::fabs(c[i]);
        if (!(a <= maxabs)) { if (a != a) return false; maxabs = a; }           // a NaN centre is no lattice
    }
    const double tol = 1e-9 * std::max(1.0, maxabs);
    const double o[3] = {c[0], c[1], c[2]};
    const double* q = c;
    for (long long ix = 0; ix < nx; ++ix) {
        const double ex = (double)ix * vs + o[0];
        for (long long iy = 0; iy < ny; ++iy) {
            const double ey = (double)iy * vs + o[1];
            for (long long iz = 0; iz < nz; ++iz, q += 3) {
                const double ez = (double)iz * vs + o[2];
                if (std::fabs(ex - q[0]) > tol || std::fabs(ey - q[1]) > tol || std::fabs(ez - q[2]) > tol) return false;
            }
        }
    }
    bb_min[0] = o[0]; bb_min[1] = o[1]; bb_min[2] = o[2];
    nvoxels[0] = (int)nx; nvoxels[1] = (int)ny; nvoxels[2] = (int)nz;
    *voxelsize = vs;
    return true;
}

// mkamd_calculate_occupancy's choice between the tiled lattice kernels and the pairwise double-precision kernel
// (occupancy_utils.pyx:34-61 takes any centre list; its one caller passes a lattice).  Only "not a lattice" and the
// lattice plan's own refusal of a geometry (ST_EINVAL: more than 1023 cells per axis, a cutoff of more than 512 voxels,
// more than 2^31 tiles) reach the pairwise kernel; every other status of the lattice path -- a HIP error, a failed
// allocation -- is the call's status.
template <class Lattice, class Pairwise>
inline int route_calculate_occupancy(bool is_lattice, Lattice&& lattice, Pairwise&& pairwise)
{
    if (is_lattice) {
        const int st = lattice();
        if (st != ST_EINVAL) return st;
    }
    return pairwise();
}

}  // namespace mkamd
