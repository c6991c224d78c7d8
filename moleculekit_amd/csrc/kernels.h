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
//   It is a GATHER: one wave owns a K x 8 x 8 voxel tile, lane = (y,z), K x-planes in registers,
//   candidate atoms come from a uniform cell list, are culled against the tile box, compacted per
//   channel through LDS and broadcast-read by all 64 lanes.  No atomics on the grid, no zero-fill
//   pass, one 32-byte store per voxel.  MFMA unused (neighbourhood min-reduction, not a contraction).
//
// Coordinates: everything is in VOXEL units relative to the grid origin (voxel i's centre sits at
// integer coordinate i).  Atoms are decomposed IN DOUBLE into (cell index, cell-centre-relative
// float32 offset) so that float32 distances carry ~1e-7 voxel of error (parity 1e-5 needs < 2e-6).
#pragma once
#ifndef MK_DEVICE_API_PROVIDED   // tests/emu provides a host-side SIMT emulation of this API
#include "mk_device.h"
#endif

namespace mkamd {

constexpr int CHG = 8;          // channels per channel-group (one group = one pass of the tile kernel)

// Everything the kernels need to know about the batch of lattice grids (passed by value).
struct GridDesc {
    int nx, ny, nz;             // voxels per axis (identical for every item of the batch)
    int tnx, tny, tnz, ntiles;  // tiles per axis / per item (tile = K x 8 x 8 voxels)
    int K;                      // x-planes per lane
    int cs_log2, cs;            // cell edge in voxels (power of two >= cutoff radius)
    int h;                      // halo cells on each side of the grid
    int ncx, ncy, ncz, ncell;   // padded cell grid per item
    int rint;                   // integer upper bound of the cutoff radius in voxels
    int C, G;                   // channels, channel groups = ceil(C/8)
    int B;                      // items (molecules / poses / frames)
    int pbc;                    // 1: per-item orthorhombic box given
    float R2;                   // cutoff^2 in voxel units  (25 / res^2)
    float R2cull;               // slightly inflated cutoff^2 for tile culling
    double inv_res;             // 1 / voxelsize
    double w_scale;             // voxelsize^2  (w = w_scale / sigma^2 -> q is in A^2/A^2)
    double Rp;                  // image acceptance radius, voxel units (cutoff + margin)
    long long V;                // nx*ny*nz
    unsigned M;                 // capacity of the record arrays
};

// w of a present channel is clamped to a finite value (+inf is the "channel absent" marker): a
// vanishing sigma then still yields 1 exactly on a voxel centre (d = 0) and 0 elsewhere, as
// occupancy_utils.pyx:57-60 does.
constexpr float MK_W_MAX = 3.0e38f;

enum { MK_ERR_RECORD_OVERFLOW = 1, MK_ERR_BAD_BOX = 2, MK_ERR_TOO_MANY_IMAGES = 4 };

// ------------------------------------------------------------------------------------------------
// Binning: atoms (and, for periodic items, their images) -> padded uniform cell grid.
// PHASE 0 counts, PHASE 1 fills the cell-sorted record arrays (counts are walked back to zero).
// One thread per atom; one global atomic per atom-image.
// Record = pos (cell-centre-relative x,y,z as f32 ; packed padded cell coords)
//          + per channel group two float4 of w = voxelsize^2 / sigma^2   (+inf: channel absent)
// ------------------------------------------------------------------------------------------------
template <int PHASE, typename SigT>
MK_KERNEL(256) void k_bin_atoms(GridDesc g, const float* __restrict__ coords,
                                const long long* __restrict__ atom_offsets, long long total_atoms,
                                const SigT* __restrict__ sigmas, const double* __restrict__ origins,
                                const float* __restrict__ box, unsigned* __restrict__ cell_count,
                                const unsigned* __restrict__ cell_start, float4* __restrict__ rec_pos,
                                float4* __restrict__ rec_w, int* __restrict__ err_flag)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= total_atoms) return;

    // item of this atom: largest b with atom_offsets[b] <= a
    int lo = 0, hi = g.B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (atom_offsets[mid] <= a) lo = mid; else hi = mid;
    }
    const int b = lo;

    // w per channel; an atom with no usable channel is dropped here (occupancy_utils.pyx:55-56)
    const SigT* sg = sigmas + (size_t)a * g.C;
    bool any = false;
    for (int c = 0; c < g.C; ++c) {
        const double s = (double)sg[c];
        const float w = (float)(g.w_scale / (s * s));
        any |= (s != 0.0) && (w == w);
    }
    if (!any) return;

    double p[3], Lv[3] = {0.0, 0.0, 0.0};
    int k0[3] = {0, 0, 0}, k1[3] = {0, 0, 0};
    const int nvox[3] = {g.nx, g.ny, g.nz};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        p[ax] = ((double)coords[3 * a + ax] - origins[3 * (size_t)b + ax]) * g.inv_res;
        if (g.pbc) {
            const double L = (double)box[3 * (size_t)b + ax] * g.inv_res;
            if (!(L > 2.0 * (g.Rp - 1e-3))) { mk_atomic_or(err_flag, MK_ERR_BAD_BOX); return; }
            Lv[ax] = L;
            const double a0 = ceil((-g.Rp - p[ax]) / L);
            const double a1 = floor(((double)(nvox[ax] - 1) + g.Rp - p[ax]) / L);
            if (a1 - a0 > 64.0) { mk_atomic_or(err_flag, MK_ERR_TOO_MANY_IMAGES); return; }
            k0[ax] = (int)a0; k1[ax] = (int)a1;          // empty range when a1 < a0
        } else {
            if (p[ax] < -g.Rp || p[ax] > (double)(nvox[ax] - 1) + g.Rp) return;
        }
    }

    const double inv_cs = 1.0 / (double)g.cs;
    const double cmid = 0.5 * (double)(g.cs - 1);
    for (int kx = k0[0]; kx <= k1[0]; ++kx)
        for (int ky = k0[1]; ky <= k1[1]; ++ky)
            for (int kz = k0[2]; kz <= k1[2]; ++kz) {
                const double q[3] = {p[0] + kx * Lv[0], p[1] + ky * Lv[1], p[2] + kz * Lv[2]};
                int pc[3];
                float rel[3];
                bool inside = true;
                const int nc[3] = {g.ncx, g.ncy, g.ncz};
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const int ci = (int)floor((q[ax] + 0.5) * inv_cs);       // unpadded cell index
                    pc[ax] = ci + g.h;
                    inside &= (pc[ax] >= 0) && (pc[ax] < nc[ax]);
                    rel[ax] = (float)(q[ax] - ((double)ci * (double)g.cs + cmid));
                }
                if (!inside) continue;
                const size_t cell = (size_t)b * g.ncell + ((size_t)pc[0] * g.ncy + pc[1]) * g.ncz + pc[2];
                if (PHASE == 0) {
                    mk_atomic_add(&cell_count[cell], 1u);
                } else {
                    const unsigned slot = cell_start[cell] + mk_atomic_sub(&cell_count[cell], 1u) - 1u;
                    if (slot >= g.M) { mk_atomic_or(err_flag, MK_ERR_RECORD_OVERFLOW); continue; }
                    rec_pos[slot] = make_float4(rel[0], rel[1], rel[2],
                                                mk_int_as_float(pc[0] | (pc[1] << 10) | (pc[2] << 20)));
                    for (int gq = 0; gq < g.G; ++gq) {
                        float w[CHG];
#pragma unroll
                        for (int c = 0; c < CHG; ++c) {
                            const int ch = gq * CHG + c;
                            float wc = mk_inf();
                            if (ch < g.C) {
                                const double s = (double)sg[ch];
                                const float t = (float)(g.w_scale / (s * s));
                                if (s != 0.0 && t == t) wc = fminf(t, MK_W_MAX);   // NaN sigma: absent
                            }
                            w[c] = wc;
                        }
                        rec_w[(size_t)(gq * 2 + 0) * g.M + slot] = make_float4(w[0], w[1], w[2], w[3]);
                        rec_w[(size_t)(gq * 2 + 1) * g.M + slot] = make_float4(w[4], w[5], w[6], w[7]);
                    }
                }
            }
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan of the cell counts (n values -> n+1 starts).  Three small kernels:
// per-4096-chunk sums, single-block scan of the sums, per-chunk scan + offset.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER_THREAD;   // 4096

// exclusive scan of one value per thread across a 256-thread block; *total gets the block sum.
MK_DEV unsigned block_scan_exclusive(unsigned v, unsigned* total, unsigned* lds /* >= 8 */)
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
        const unsigned s = lds[i];
        if (i < wv) woff += s;
        tot += s;
    }
    mk_block_sync();                     // lds may be reused by the caller's next call
    *total = tot;
    return woff + incl - v;
}

MK_KERNEL(SCAN_THREADS) void k_scan_chunk_sums(const unsigned* __restrict__ in, size_t n,
                                               unsigned* __restrict__ chunk_sums)
{
    __shared__ unsigned lds[8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        const size_t i = base + (size_t)j * SCAN_THREADS + threadIdx.x;
        s += (i < n) ? in[i] : 0u;
    }
    unsigned tot;
    (void)block_scan_exclusive(s, &tot, lds);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}

MK_KERNEL(SCAN_THREADS) void k_scan_sums_inplace(unsigned* __restrict__ chunk_sums, unsigned nchunks)
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

MK_KERNEL(SCAN_THREADS) void k_scan_finish(const unsigned* __restrict__ in, size_t n,
                                           const unsigned* __restrict__ chunk_offsets,
                                           unsigned* __restrict__ out /* n+1 */)
{
    __shared__ unsigned lds[8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    unsigned v[SCAN_PER_THREAD];
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        const size_t i = base + j;
        v[j] = (i < n) ? in[i] : 0u;
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
// ------------------------------------------------------------------------------------------------
MK_DEV float occupancy_from_q(float q)
{
    const float u = mk_rcp_refined(q);
    const float u3 = u * u * u;
    const float u6 = u3 * u3;
    return 1.0f - mk_exp2(-1.4426950408889634f * u6);
}

// ------------------------------------------------------------------------------------------------
// THE hot kernel.  One 64-lane wave per K x 8 x 8 voxel tile (lane = (y,z), z fastest as in the
// output layout, K x-planes per lane in registers); blockIdx.y = channel group.
// ------------------------------------------------------------------------------------------------
template <int K>
MK_KERNEL(64) void k_voxelize_tiles(GridDesc g, const unsigned* __restrict__ cell_start,
                                    const float4* __restrict__ rec_pos,
                                    const float4* __restrict__ rec_w, float* __restrict__ out)
{
    static_assert(K == 4 || K == 8, "K");
    __shared__ float4 ebuf[WAVE + 1];             // per-channel compacted entries (x,y,z,w) + pad

    const int lane = threadIdx.x;
    // XCD-aware order: the dispatcher places block i on XCD i%8; give each XCD a contiguous run of
    // tiles so neighbouring tiles (which share candidate cells) hit the same 4 MiB L2.
    const unsigned per_xcd = gridDim.x >> 3;      // gridDim.x is a multiple of 8 (host pads)
    const unsigned lt = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const unsigned total_tiles = (unsigned)g.B * (unsigned)g.ntiles;
    if (lt >= total_tiles) return;                // whole wave leaves together
    const int gq = blockIdx.y;

    const int b = (int)(lt / (unsigned)g.ntiles);
    int t = (int)(lt - (unsigned)b * (unsigned)g.ntiles);
    const int tz = t % g.tnz; t /= g.tnz;
    const int ty = t % g.tny;
    const int tx = t / g.tny;
    const int x0 = tx * K, y0 = ty * 8, z0 = tz * 8;

    const int ly = lane >> 3, lz = lane & 7;
    const float Y = (float)ly - 3.5f, Z = (float)lz - 3.5f;
    constexpr float HX = 0.5f * (float)(K - 1);
    const float R2 = g.R2, R2cull = g.R2cull, INF = mk_inf();

    // running minima kept as BIT PATTERNS: every candidate value is a non-negative float (or +inf /
    // NaN), for which unsigned-integer order == float order and NaN (0x7fc00000) sorts above +inf,
    // so v_min_u32 is an exact NaN-ignoring float min with no canonicalisation op in front of it.
    unsigned q[CHG][K];
#pragma unroll
    for (int c = 0; c < CHG; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) q[c][k] = 0x7f800000u;

    // padded cell ranges that can hold atoms within the cutoff of this tile
    const int h = g.h, csl = g.cs_log2;
    int cx_lo = ((x0 - g.rint) >> csl) + h, cx_hi = ((x0 + K - 1 + g.rint) >> csl) + h;
    int cy_lo = ((y0 - g.rint) >> csl) + h, cy_hi = ((y0 + 7 + g.rint) >> csl) + h;
    int cz_lo = ((z0 - g.rint) >> csl) + h, cz_hi = ((z0 + 7 + g.rint) >> csl) + h;
    cx_lo = cx_lo < 0 ? 0 : cx_lo; cx_hi = cx_hi > g.ncx - 1 ? g.ncx - 1 : cx_hi;
    cy_lo = cy_lo < 0 ? 0 : cy_lo; cy_hi = cy_hi > g.ncy - 1 ? g.ncy - 1 : cy_hi;
    cz_lo = cz_lo < 0 ? 0 : cz_lo; cz_hi = cz_hi > g.ncz - 1 ? g.ncz - 1 : cz_hi;

    // cell centre (voxel coords) minus tile centre, per axis:  (pc-h)*cs + (cs-1)/2 - (x0 + HX)
    const float cmid = 0.5f * (float)(g.cs - 1);
    const float offx = cmid - (float)(h * g.cs) - ((float)x0 + HX);
    const float offy = cmid - (float)(h * g.cs) - ((float)y0 + 3.5f);
    const float offz = cmid - (float)(h * g.cs) - ((float)z0 + 3.5f);
    const float fcs = (float)g.cs;

    const float4* __restrict__ w0p = rec_w + (size_t)(gq * 2 + 0) * g.M;
    const float4* __restrict__ w1p = rec_w + (size_t)(gq * 2 + 1) * g.M;

    for (int pcx = cx_lo; pcx <= cx_hi; ++pcx)
        for (int pcy = cy_lo; pcy <= cy_hi; ++pcy) {
            const size_t cbase = (size_t)b * g.ncell + ((size_t)pcx * g.ncy + pcy) * g.ncz;
            const unsigned r0 = cell_start[cbase + cz_lo];
            const unsigned r1 = cell_start[cbase + cz_hi + 1];     // z-run of cells is contiguous
            for (unsigned rr = r0; rr < r1; rr += WAVE) {
                const unsigned r = rr + lane;
                const bool valid = r < r1;
                // ---- stage: each lane takes one candidate record, makes it tile-relative, culls ----
                float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) P = rec_pos[r];
                const int pk = mk_float_as_int(P.w);
                // (cell centre - tile centre) is an exact small half-integer; ONE rounding per axis
                const float ex = P.x + ((float)(pk & 1023) * fcs + offx);
                const float ey = P.y + ((float)((pk >> 10) & 1023) * fcs + offy);
                const float ez = P.z + ((float)((pk >> 20) & 1023) * fcs + offz);
                const float gx = fmaxf(fabsf(ex) - HX, 0.f);
                const float gy = fmaxf(fabsf(ey) - 3.5f, 0.f);
                const float gz = fmaxf(fabsf(ez) - 3.5f, 0.f);
                const bool surv = valid && (gx * gx + gy * gy + gz * gz < R2cull);
                float4 W0 = make_float4(INF, INF, INF, INF), W1 = W0;
                if (surv) { W0 = w0p[r]; W1 = w1p[r]; }
                const float wv[CHG] = {W0.x, W0.y, W0.z, W0.w, W1.x, W1.y, W1.z, W1.w};

#pragma unroll
                for (int c = 0; c < CHG; ++c) {
                    const float wc = wv[c];
                    const bool has = surv && (wc < INF);                // false for +inf and NaN
                    const unsigned long long mask = mk_ballot(has);
                    if (mask == 0ull) continue;                          // wave-uniform
                    const int n = mk_popc64(mask);
                    // ---- compact this channel's entries through LDS ----
                    if (has) ebuf[mk_rank_in_mask(mask)] = make_float4(ex, ey, ez, wc);
                    mk_block_sync();
                    // ---- every lane visits every entry (LDS broadcast read, next entry prefetched) ----
                    float4 e = ebuf[0];
                    for (int i = 0; i < n; ++i) {
                        const float4 en = ebuf[i + 1];                   // slot n is padding
                        const float dy = Y - e.y, dz = Z - e.z;
                        const float dyz2 = dy * dy + dz * dz;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const float dx = ((float)k - HX) - e.x;
                            const float d2 = dx * dx + dyz2;
                            const float s = d2 * e.w;
                            const float t = d2 < R2 ? s : INF;           // occupancy_utils.pyx:53
                            q[c][k] = mk_min_bits(q[c][k], t);
                        }
                        e = en;
                    }
                    mk_block_sync();                                     // ebuf is rewritten next
                }
            }
        }

    // ---- epilogue: q -> occupancy, one 32-byte store per voxel (z fastest across lanes) ----
    const int y = y0 + ly, z = z0 + lz;
    const bool yz_in = (y < g.ny) && (z < g.nz);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int x = x0 + k;
        float f[CHG];
#pragma unroll
        for (int c = 0; c < CHG; ++c)
            f[c] = occupancy_from_q(mk_uint_as_float(q[c][k]));
        if (yz_in && x < g.nx) {
            const size_t vox = (size_t)b * (size_t)g.V + ((size_t)x * g.ny + y) * g.nz + z;
            if (g.C == CHG) {
                float4* o = reinterpret_cast<float4*>(out + vox * CHG);
                o[0] = make_float4(f[0], f[1], f[2], f[3]);
                o[1] = make_float4(f[4], f[5], f[6], f[7]);
            } else {
                float* o = out + vox * (size_t)g.C + (size_t)gq * CHG;
#pragma unroll
                for (int c = 0; c < CHG; ++c)
                    if (gq * CHG + c < g.C) o[c] = f[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Explicit (non-lattice) centres: the exact calculate_occupancy contract for arbitrary `centers`
// (usercenters / direct calls).  Distances in DOUBLE exactly as the reference (float32 coords
// promoted, strict d^2 < 25), min-q reduction in float32.  Brute force O(N*V): thread per centre,
// atoms staged through LDS in chunks of 256.  blockIdx.y = channel group.
// w here is 1/sigma^2 in A^-2 (w_scale = 1).
// ------------------------------------------------------------------------------------------------
constexpr int EXPL_THREADS = 256;

MK_KERNEL(EXPL_THREADS) void k_occupancy_centers(const double* __restrict__ centers, long long V,
                                                 const float* __restrict__ coords, long long N,
                                                 const float4* __restrict__ w /* [G][2][N] */,
                                                 int C, int use_box, double bx, double by, double bz,
                                                 float* __restrict__ out)
{
    __shared__ float4 s_pos[EXPL_THREADS];
    __shared__ float4 s_w0[EXPL_THREADS];
    __shared__ float4 s_w1[EXPL_THREADS];
    const int gq = blockIdx.y;
    const long long v = (long long)blockIdx.x * EXPL_THREADS + threadIdx.x;
    const bool active = v < V;
    double cx = 0, cy = 0, cz = 0;
    if (active) { cx = centers[3 * v]; cy = centers[3 * v + 1]; cz = centers[3 * v + 2]; }
    const float INF = mk_inf();
    float q[CHG];
#pragma unroll
    for (int c = 0; c < CHG; ++c) q[c] = INF;

    for (long long a0 = 0; a0 < N; a0 += EXPL_THREADS) {
        const long long a = a0 + threadIdx.x;
        if (a < N) {
            s_pos[threadIdx.x] = make_float4(coords[3 * a], coords[3 * a + 1], coords[3 * a + 2], 0.f);
            s_w0[threadIdx.x] = w[(size_t)(gq * 2 + 0) * N + a];
            s_w1[threadIdx.x] = w[(size_t)(gq * 2 + 1) * N + a];
        }
        mk_block_sync();
        const int cnt = (N - a0) < EXPL_THREADS ? (int)(N - a0) : EXPL_THREADS;
        for (int i = 0; i < cnt; ++i) {
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
            const float wv[CHG] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int c = 0; c < CHG; ++c) {
                const float s = in ? d2f * wv[c] : INF;     // 0*inf = NaN is ignored by mk_min
                q[c] = mk_min(q[c], s);
            }
        }
        mk_block_sync();
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < CHG; ++c)
            if (gq * CHG + c < C) out[(size_t)v * C + gq * CHG + c] = occupancy_from_q(q[c]);
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
            float wc = mk_inf();
            if (ch < C) {
                const double s = (double)sigmas[(size_t)a * C + ch];
                const float x = (float)(w_scale / (s * s));
                if (s != 0.0 && x == x) wc = fminf(x, MK_W_MAX);       // NaN sigma: absent
            }
            t[c] = wc;
        }
        w[(size_t)(gq * 2 + 0) * N + a] = make_float4(t[0], t[1], t[2], t[3]);
        w[(size_t)(gq * 2 + 1) * N + a] = make_float4(t[4], t[5], t[6], t[7]);
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
