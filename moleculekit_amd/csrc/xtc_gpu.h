// xtc_gpu.h -- XTC coordinate decompression ON THE DEVICE (round 4; SURVEY.md section 8f-4, "trajectory feeding").
//
// The host decoder (xtc_reader.h) feeds a cfg4-shaped trajectory at 28 k frames/s on the ~16 cores an MI355X box grants; the
// voxelizer takes 240 k.  Frames are independent records, but the bit stream INSIDE a frame is serial: where an atom's bits
// start depends on the flag and run length after every full-precision atom before it.  A first version gave a lane a frame and
// had it decode everything (the state machine of xtc_reader.h::decode_frame, which restates xdrfile.cpp:749-983): ~300
// instructions per atom at the one-instruction-per-4-5-cycles of a wave alone on its SIMD -- 28 ms for a 30 000-atom frame
// whatever the number of frames (PMC: docs/EXPERIMENTS_r4.md section 7).  What is serial is only the WALK, so the work is split:
//
//   k_xtc_scan    a lane per frame walks its stream from LDS windows the wave refills together: per GROUP (a full-precision
//                 atom and the run of small atoms that follows it) it reads the 1 + 5 flag / run bits, writes a record
//                 {bit position, first output atom, smallidx, run length} and skips the rest -- no triple is decoded
//   k_xtc_expand  a thread per group: the mixed-radix numbers, the deltas, the output order (the first small atom goes out
//                 before the full one), the float32 operations of the host path ((float)int * (1 / precision), then * scale),
//                 frame-major ([frame][atom][3]: the packed items the voxelizer takes -- no transposition)
//
//   bytes   the frames' records, as in the file, from a 4-byte-aligned file offset (device copy, XTC_PAD bytes longer than the records)
//   desc    one XtcFrameDesc per frame, parsed from the record headers on the host (xtc_reader.h::frame_desc)
//   status  per frame: 0 ok, 1 corrupt stream (overrun, table index out of range, more atoms than announced),
//           2 outside what the device path takes: a number of more than 64 bits (ranges of > ~2 million quanta per axis), a
//           frame of >= 2^21 atoms or >= 512 MB (the host decoder takes such files)
#pragma once
#ifndef MK_DEVICE_API_PROVIDED
#include "mk_device.h"
#endif

namespace mkamd {

struct XtcFrameDesc {                    // 64 bytes
    unsigned long long data_off;         // of the bit stream (or of the plain floats) inside `bytes`; a multiple of 4
    unsigned nbytes;                     // of the bit stream
    int smallidx;
    int lo[3];
    unsigned range[3];
    float inv_precision;
    int triple_bits;                     // bits of a full coordinate's mixed-radix number; 0 = three bit fields (`field_bits`)
    int field_bits[3];
    int raw;                             // <= 9 atoms: plain big-endian floats
};
static_assert(sizeof(XtcFrameDesc) == 64, "layout shared with the host");

MK_DEV_CONST int XTC_MAGIC[73] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406,
                                      512, 645, 812, 1024, 1290, 1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642,
                                      26007, 32768, 41285, 52015, 65536, 82570, 104031, 131072, 165140, 208063, 262144, 330280, 416127,
                                      524287, 660561, 832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304, 5284491,
                                      6658042, 8388607, 10568983, 13316085, 16777216};
constexpr int XTC_FIRST = 9, XTC_NMAGIC = 73;

constexpr int XTC_PAD = 1024;            // bytes the caller's byte buffer extends past the last record (windows and refills read
                                         // ahead of what they use)

struct XtcBits {                         // MSB-first bit reader over 32-bit big-endian words, from any bit position (k_xtc_expand)
    const unsigned* wp;
    unsigned long long acc;              // next bits, left-aligned
    int nacc;
    MK_DEVFN void start(const unsigned char* base, unsigned bitpos)
    {
        wp = reinterpret_cast<const unsigned*>(base) + (bitpos >> 5);
        const int sh = (int)(bitpos & 31u);
        acc = (unsigned long long)__builtin_bswap32(*wp++) << (32 + sh);
        nacc = 32 - sh;
    }
    MK_DEVFN unsigned get(int n)       // 0 <= n <= 32
    {
        if (nacc < n) {
            acc |= (unsigned long long)__builtin_bswap32(*wp++) << (32 - nacc);
            nacc += 32;
        }
        const unsigned r = (unsigned)((acc >> 1) >> (63 - n));
        acc <<= n;
        nacc -= n;
        return r;
    }
};

// x = q * r + rem for r < 2^25 and q < 2^48 (the quotients of the format's mixed-radix numbers): the quotient estimated in
// double (53 bits: within one of the true one) and corrected -- the 64-bit integer division of this target is a ~100-instruction
// routine, and there are two per atom
MK_DEV unsigned long long xtc_divmod(unsigned long long x, unsigned r, double rinv, unsigned& rem)
{
    unsigned long long q = (unsigned long long)((double)x * rinv);
    long long d = (long long)(x - q * (unsigned long long)r);
    while (d < 0) { --q; d += (long long)r; }
    while (d >= (long long)r) { ++q; d -= (long long)r; }
    rem = (unsigned)d;
    return q;
}

// the same for x < 2^32: the conversions are one instruction each way, the product x * (1/r) is within one of the quotient
// (x is exact in double, 1/r is off by 2^-53 of itself), and the correction is two selects.  Most of the stream's numbers are
// the small atoms' (smallidx bits, ~20-30) and the second division of a full one -- this path is what the lane's latency is made of.
MK_DEV unsigned xtc_divmod32(unsigned x, unsigned r, double rinv, unsigned& rem)
{
    unsigned q = (unsigned)((double)x * rinv);
    int d = (int)(x - q * r);
    if (d < 0) { --q; d += (int)r; }
    if (d >= (int)r) { ++q; d -= (int)r; }
    rem = (unsigned)d;
    return q;
}

// three values packed as one mixed-radix number of nbits (<= 64) bits whose bytes come least-significant first
MK_DEV void xtc_triple(XtcBits& b, int nbits, unsigned r1, unsigned r2, double r1inv, double r2inv, int (&out)[3])
{
    int nfull = (nbits - 1) >> 3;
    const int top = nbits - 8 * nfull;
    unsigned rem2, rem1;
    if (nbits <= 32) {
        unsigned x = 0u;
        if (nfull) x = __builtin_bswap32(b.get(8 * nfull) << (32 - 8 * nfull));
        x |= b.get(top) << (8 * nfull);
        const unsigned q2 = xtc_divmod32(x, r2, r2inv, rem2);
        const unsigned q1 = xtc_divmod32(q2, r1, r1inv, rem1);
        out[2] = (int)rem2; out[1] = (int)rem1; out[0] = (int)q1;
        return;
    }
    unsigned long long x = 0ull;
    int shift = 0;
    while (nfull >= 4) { x |= (unsigned long long)__builtin_bswap32(b.get(32)) << shift; shift += 32; nfull -= 4; }
    if (nfull) { x |= (unsigned long long)__builtin_bswap32(b.get(8 * nfull) << (32 - 8 * nfull)) << shift; shift += 8 * nfull; }
    x |= (unsigned long long)b.get(top) << shift;
    const unsigned long long q2 = xtc_divmod(x, r2, r2inv, rem2);
    unsigned long long q1;
    if ((q2 >> 32) == 0ull) q1 = xtc_divmod32((unsigned)q2, r1, r1inv, rem1);
    else q1 = xtc_divmod(q2, r1, r1inv, rem1);
    out[2] = (int)rem2; out[1] = (int)rem1; out[0] = (int)(unsigned)(q1 & 0xffffffffull);
}

// ---- pass 1: the walk ----
constexpr int XS_WIN = 1024;             // bytes of a lane's window of its stream (512: +8 % on a stream without runs, +14 % on 3PTB's -- a refill costs the wave ~3 000 cycles)
constexpr int XS_ROW = XS_WIN + 8;       // its LDS row (the word after the window may be read; its bits are never used)
constexpr int XS_LW = XS_WIN / 4 / WAVE; // words of a row a lane loads in a refill
constexpr int XS_BATCH = 32 / XS_LW;     // rows refilled per batch of loads in flight
constexpr int XS_SPEC = 8;               // groups looked at together (below)
static_assert(XS_WIN % (4 * WAVE) == 0 && XS_BATCH >= 1 && WAVE % XS_BATCH == 0, "window: whole words per lane");

struct XtcGroup { unsigned pos, what; }; // what = first output atom (21 bits) | smallidx of the run << 21 | small atoms << 28

MK_KERNEL(64) void k_xtc_scan(const unsigned char* __restrict__ bytes, const XtcFrameDesc* __restrict__ desc, long long nframes,
                              long long natoms, float scale, float* __restrict__ out, XtcGroup* __restrict__ groups,
                              int* __restrict__ ngroups, int* __restrict__ status)
{
    __shared__ unsigned win[WAVE * XS_ROW / 4];
    // this wave is a latency chain on a SIMD it usually shares with the voxelizer's waves (the feed decodes chunk k+1 beside
    // chunk k's tile kernel): first in line at the issue arbiter (beside cfg4 steps: 9.5 ms per 2 048 frames without, 7.7 alone)
    mk_setprio_high();
    const int lane = (int)(threadIdx.x & (WAVE - 1));
    const long long f = (long long)blockIdx.x * WAVE + lane;
    bool live = f < nframes;
    XtcFrameDesc d = {};
    if (live) d = desc[f];
    int st = 0;
    if (live && d.raw) {                                             // <= 9 atoms: plain big-endian floats
        const unsigned* p = reinterpret_cast<const unsigned*>(bytes + d.data_off);
        float* __restrict__ o = out + (size_t)f * (size_t)natoms * 3;
        for (long long a = 0; a < natoms * 3; ++a) o[a] = mk_fmul_rn(mk_uint_as_float(__builtin_bswap32(p[a])), scale);
        status[f] = 0; ngroups[f] = 0;
        live = false; d.data_off = 0ull;
    }
    const bool mine = live;
    int smallidx = d.smallidx;
    if (live) {
        if (d.triple_bits > 64 || natoms >= (1ll << 21) || d.nbytes >= (1u << 29) - 4u) { st = 2; live = false; }
        else if (smallidx < XTC_FIRST || smallidx >= XTC_NMAGIC) { st = 1; live = false; }
    }
    const unsigned full_bits = d.triple_bits ? (unsigned)d.triple_bits : (unsigned)(d.field_bits[0] + d.field_bits[1] + d.field_bits[2]);
    const unsigned long long total_bits = (unsigned long long)((d.nbytes + 3u) / 4u) * 32ull;
    XtcGroup* __restrict__ grp = groups + (size_t)(f < nframes ? f : 0) * (size_t)(natoms + XS_SPEC);
    unsigned pos = 0u;
    int w = 0, g = 0, run = 0;
    bool together = false;                                           // wave-uniform: look at XS_SPEC groups together (below)
    while (mk_ballot(live) != 0ull) {
        // the wave refills every lane's window: row l <- XS_WIN bytes of frame l's stream from the word its position is in
        const unsigned wbase = pos >> 5;
        const unsigned long long src = (unsigned long long)(uintptr_t)(bytes + d.data_off) + 4ull * wbase;
        const unsigned src_lo = (unsigned)src, src_hi = (unsigned)(src >> 32);
        mk_block_sync();                                             // the rows are no longer being read
        for (int l0 = 0; l0 < WAVE; l0 += XS_BATCH) {
            unsigned v[XS_BATCH][XS_LW];
#pragma unroll
            for (int j = 0; j < XS_BATCH; ++j) {
                const unsigned long long a = (unsigned long long)mk_readlane(src_lo, l0 + j) | ((unsigned long long)mk_readlane(src_hi, l0 + j) << 32);
                const unsigned* __restrict__ p = reinterpret_cast<const unsigned*>((uintptr_t)a);
#pragma unroll
                for (int i = 0; i < XS_LW; ++i) v[j][i] = p[XS_LW * lane + i];
            }
#pragma unroll
            for (int j = 0; j < XS_BATCH; ++j)
#pragma unroll
                for (int i = 0; i < XS_LW; ++i) win[(l0 + j) * (XS_ROW / 4) + XS_LW * lane + i] = v[j][i];
        }
        mk_block_sync();
        // the walk inside the window.  A wave alone on its SIMD issues an instruction every ~5-8 cycles and pays every divergent
        // branch in exec-mask bookkeeping (the first version of this loop: ~100 instructions and 820 cycles per group), so the
        // checks are accumulated, not branched on, and positions are 32-bit offsets from the window's start.  And what is
        // serial is less than it looks: a flag is only SET where the run length or the small-number table changes -- between
        // two flags every group has the same length (stride below), so the flag bits of the next XS_SPEC groups are read
        // TOGETHER (independent LDS reads: one latency), the groups before the first set flag are taken in one step, and the
        // one-group step with all its checks only runs where a flag, the window's end or an error stops them.  That pays where
        // flags are rare (this package's writer sets none: 6.3 -> 2.9 ms per 30 000-atom frame; 4RWS with its water: 18 % of the
        // groups) and costs a second LDS round trip per group where they are not -- on a protein the reference's writer
        // adapts its small-number table at nearly every group (3PTB: 95 %; 0.24 -> 0.69 ms) --, so the wave votes after every
        // window on what its frames' flags were like.
        int n_clear = 0, n_set = 0;
        const unsigned* row = &win[lane * (XS_ROW / 4)];
        const unsigned long long wbit = (unsigned long long)wbase << 5;
        const unsigned tot = (unsigned)(total_bits - wbit < 0x7fff0000ull ? total_bits - wbit : 0x7fff0000ull);  // (wbit <= pos <= total_bits)
        unsigned rel = pos & 31u;
        // one group, whatever it is (the caller has checked that its flag and run bits are inside the window)
        auto one_group = [&]() {
            const unsigned hdr = rel + full_bits;                                // where the flag bit is
            const unsigned long long two = ((unsigned long long)__builtin_bswap32(row[hdr >> 5]) << 32) | __builtin_bswap32(row[(hdr >> 5) + 1]);
            const unsigned v = (unsigned)(two >> (58u - (hdr & 31u))) & 63u;     // flag, then the five run bits
            const bool flag = (v & 32u) != 0u;
            n_set += flag ? 1 : 0;
            n_clear += flag ? 0 : 1;
            const int r5 = (int)(v & 31u), m3 = r5 - 3 * ((r5 * 171) >> 9);      // r5 % 3
            run = flag ? r5 - m3 : run;                                          // (a run length stays until the next flag)
            const int step = flag ? m3 - 1 : 0;                                  // -1 / 0 / +1
            const int nsmall = (run * 171) >> 9;                                 // run / 3
            const unsigned next = hdr + (flag ? 6u : 1u) + (unsigned)(nsmall * smallidx);
            const bool e_end = hdr + 1u > tot;
            const bool e_wide = nsmall != 0 && smallidx > 64;
            const bool e_more = w + 1 + nsmall > (int)natoms || next > tot;
            grp[g] = XtcGroup{(unsigned)wbit + rel, (unsigned)w | ((unsigned)smallidx << 21) | ((unsigned)nsmall << 28)};  // (g <= w < natoms)
            ++g;
            rel = next;
            w += 1 + nsmall;
            smallidx += step;
            const bool e_idx = (unsigned)(smallidx - XTC_FIRST) >= (unsigned)(XTC_NMAGIC - XTC_FIRST);
            st = e_end ? 1 : (e_wide ? 2 : ((e_more || e_idx) ? 1 : 0));
            live = st == 0 && w != (int)natoms;
        };
        // (two loops, not one with the step above under a condition: with both in one loop the compiler's exec-mask
        // bookkeeping made the one-by-one walk 80 % slower -- 3PTB 0.23 -> 0.42 ms)
        if (!together) {
            while (live && rel + full_bits + 6u <= XS_WIN * 8u) one_group();
        } else {
            while (live && rel + full_bits + 6u <= XS_WIN * 8u) {
                const int ns = (run * 171) >> 9;                                 // small atoms per group until a flag says otherwise (run / 3)
                const unsigned tail = 1u + (unsigned)(ns * smallidx), stride = full_bits + tail;
                const int per = 1 + ns;
                // (every word is read before anything is decided, and nothing below branches: eight independent LDS reads, one wait)
                unsigned wd[XS_SPEC];
#pragma unroll
                for (int j = 0; j < XS_SPEC; ++j) {
                    const unsigned k = (rel + full_bits + (unsigned)j * stride) >> 5;
                    wd[j] = row[k < (unsigned)(XS_WIN / 4) ? k : (unsigned)(XS_WIN / 4)];
                }
                bool open = !(ns != 0 && smallidx > 64);
                int np = 0;
#pragma unroll
                for (int j = 0; j < XS_SPEC; ++j) {
                    const unsigned h = rel + full_bits + (unsigned)j * stride;   // group j's flag bit, if groups 0..j-1 have none
                    const bool ok = (h + 6u <= XS_WIN * 8u) & (((__builtin_bswap32(wd[j]) >> (31u - (h & 31u))) & 1u) == 0u) & (h + tail <= tot) &
                                    (w + (j + 1) * per <= (int)natoms);
                    open = open & ok;
                    np += open ? 1 : 0;
                    // (written whether taken or not: the frame's records have XS_SPEC of slack, and what is not taken is
                    // overwritten by the next step or lies beyond the frame's count)
                    grp[g + j] = XtcGroup{(unsigned)wbit + h - full_bits, (unsigned)(w + j * per) | ((unsigned)smallidx << 21) | ((unsigned)ns << 28)};
                }
                g += np;
                n_clear += np;
                rel += (unsigned)np * stride;
                w += np * per;
                if (w == (int)natoms) live = false;
                if (np < XS_SPEC && live && rel + full_bits + 6u <= XS_WIN * 8u) one_group();
            }
        }
        pos = (unsigned)wbit + rel;
        // a lane that is done -- or dead: a corrupt frame's last step may have put `rel` up to a group's length past the end of
        // its stream -- keeps taking part in the refills (the wave refills all 64 rows while any lane is live): from the start
        // of the byte buffer, so that nothing is read beyond the MKAMD_XTC_PAD bytes the header asks for behind the last record
        if (!live) { pos = 0u; d.data_off = 0ull; }
        const unsigned long long voters = mk_ballot(live), ayes = mk_ballot(live && n_clear >= 3 * n_set);
        together = voters != 0ull && 2 * mk_popc64(ayes) >= mk_popc64(voters);
    }
    if (mine) { status[f] = st; ngroups[f] = st ? 0 : g; }
}

// ---- pass 2: the numbers ----
MK_KERNEL(256) void k_xtc_expand(const unsigned char* __restrict__ bytes, const XtcFrameDesc* __restrict__ desc, long long frame0,
                                 long long natoms, float scale, float* __restrict__ out, const XtcGroup* __restrict__ groups,
                                 const int* __restrict__ ngroups, int blocks_per_frame)
{
    const long long f = frame0 + (long long)(blockIdx.x / (unsigned)blocks_per_frame);
    const int g = (int)(blockIdx.x % (unsigned)blocks_per_frame) * 256 + (int)threadIdx.x;
    if (g >= ngroups[f]) return;
    const XtcFrameDesc& d = desc[f];
    const XtcGroup rec = groups[(size_t)f * (size_t)(natoms + XS_SPEC) + g];
    const long long w = (long long)(rec.what & 0x1FFFFFu);
    const int sidx = (int)((rec.what >> 21) & 127u), nsmall = (int)(rec.what >> 28);
    float* __restrict__ o = out + (size_t)f * (size_t)natoms * 3;
    const float inv_precision = d.inv_precision;
    auto put = [&](long long a, const int (&c)[3]) {
        // the host path's float32 operations: (float)int * inv_precision (xtc_reader.h), then the nm -> Angstrom scale
        o[3 * a + 0] = mk_fmul_rn(mk_fmul_rn((float)c[0], inv_precision), scale);
        o[3 * a + 1] = mk_fmul_rn(mk_fmul_rn((float)c[1], inv_precision), scale);
        o[3 * a + 2] = mk_fmul_rn(mk_fmul_rn((float)c[2], inv_precision), scale);
    };
    XtcBits b;
    b.start(bytes + d.data_off, rec.pos);
    int cur[3];
    if (d.triple_bits == 0) { cur[0] = (int)b.get(d.field_bits[0]); cur[1] = (int)b.get(d.field_bits[1]); cur[2] = (int)b.get(d.field_bits[2]); }
    else xtc_triple(b, d.triple_bits, d.range[1], d.range[2], 1.0 / (double)d.range[1], 1.0 / (double)d.range[2], cur);
    cur[0] = (int)((unsigned)cur[0] + (unsigned)d.lo[0]); cur[1] = (int)((unsigned)cur[1] + (unsigned)d.lo[1]); cur[2] = (int)((unsigned)cur[2] + (unsigned)d.lo[2]);
    if (nsmall == 0) { put(w, cur); return; }
    if (b.get(1) == 1u) b.get(5);
    const unsigned radix = (unsigned)XTC_MAGIC[sidx];
    const unsigned smallnum = radix / 2u;
    const double inv = 1.0 / (double)radix;
    int prev[3] = {cur[0], cur[1], cur[2]};
    for (int k = 0; k < nsmall; ++k) {
        int nxt[3];
        xtc_triple(b, sidx, radix, radix, inv, inv, nxt);
        nxt[0] = (int)((unsigned)nxt[0] + (unsigned)prev[0] - smallnum);
        nxt[1] = (int)((unsigned)nxt[1] + (unsigned)prev[1] - smallnum);
        nxt[2] = (int)((unsigned)nxt[2] + (unsigned)prev[2] - smallnum);
        if (k == 0) { put(w, nxt); put(w + 1, prev); }               // the first small atom goes out BEFORE the full-precision one
        else put(w + 1 + k, nxt);
        prev[0] = nxt[0]; prev[1] = nxt[1]; prev[2] = nxt[2];
    }
}

}  // namespace mkamd
