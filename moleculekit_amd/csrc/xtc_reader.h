// xtc_reader.h -- host-side decoder for GROMACS XTC trajectories ("trajectory feeding").
//
// Replaces, for reading, moleculekit/fileformats/xtc/src/{xdrfile.cpp: xdrfile_decompress_coord_float :749-983,
// xdrfile_xtc.cpp: xtc_header/xtc_coord :17-72, xtc_src.cpp: the frame index + xtc_read_new / xtc_read_frame}:
// same files in, the same float32 bits out (coords [natoms, 3, nframes] frame-fastest in nm, box vectors
// [3, 3, nframes], time, step).  The format is the published xdrfile / libxdrfile "xdr3dfcoord" scheme:
//
//   frame  := magic(1995) natoms step time(f32) box(9 f32) natoms                    -- big-endian XDR words
//             natoms <= 9 ?  3*natoms f32
//                         :  precision(f32) minint[3] maxint[3] smallidx nbytes  bitstream(nbytes, padded to 4)
//   bitstream, MSB first.  Per atom a triple of non-negative ints relative to minint, packed either as three bit
//   fields (when a range needs more than 24 bits) or as ONE mixed-radix number x0*(s1*s2) + x1*s2 + x2 whose bytes
//   are stored least-significant first; then a flag bit and, if set, 5 bits that change the run length (number
//   of following atoms coded as small offsets from their predecessor, radix MAGIC[smallidx] per axis) and move
//   smallidx one step down / up.  The first small atom of a run is swapped with its predecessor on output
//   (water oxygens behind their hydrogens).  A run length persists until a flag changes it.
//
// Own structure: memory-mapped file, frame index from the record lengths, a 64-bit accumulator bit reader, the
// mixed-radix number held in an unsigned __int128 (at most 72 bits) instead of a byte-array long division, and
// frames decoded in parallel on host threads (they are independent records).  No HIP in here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mkamd {
namespace xtc {

// radix table of the format ("magicints" of the xdrfile specification); entries below FIRST are unused zeros
constexpr int MAGIC[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406,
                         512, 645, 812, 1024, 1290, 1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642,
                         26007, 32768, 41285, 52015, 65536, 82570, 104031, 131072, 165140, 208063, 262144, 330280, 416127,
                         524287, 660561, 832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304, 5284491,
                         6658042, 8388607, 10568983, 13316085, 16777216};
constexpr int FIRST = 9;
constexpr int NMAGIC = (int)(sizeof(MAGIC) / sizeof(MAGIC[0]));
constexpr int32_t FRAME_MAGIC = 1995;

enum Status { OK = 0, E_OPEN = 1, E_FORMAT = 2, E_RANGE = 3 };

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    long long ident[7] = {0, 0, 0, 0, 0, 0, 0};   // device, inode, size, mtime and ctime (s, ns): what the frame index is remembered by
    ~Mapped()
    {
        if (p) munmap(const_cast<uint8_t*>(p), n);
        if (fd >= 0) close(fd);
    }
    bool open_file(const char* path)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        ident[0] = (long long)st.st_dev; ident[1] = (long long)st.st_ino; ident[2] = (long long)st.st_size;
        ident[3] = (long long)st.st_mtim.tv_sec; ident[4] = (long long)st.st_mtim.tv_nsec;
        ident[5] = (long long)st.st_ctim.tv_sec; ident[6] = (long long)st.st_ctim.tv_nsec;
        if (n == 0) return true;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { p = nullptr; return false; }
        p = (const uint8_t*)m;
        return true;
    }
};

inline uint32_t be32(const uint8_t* q) { return ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3]; }
inline int32_t be_i32(const uint8_t* q) { return (int32_t)be32(q); }
inline float be_f32(const uint8_t* q)
{
    const uint32_t u = be32(q);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// byte offsets of the frames; natoms of the first frame.  A truncated last record ends the list.
inline int index_frames(const Mapped& m, std::vector<size_t>& offs, int64_t& natoms)
{
    offs.clear();
    natoms = 0;
    size_t p = 0;
    while (p + 16 + 36 + 4 <= m.n) {
        if (be_i32(m.p + p) != FRAME_MAGIC) return offs.empty() ? E_FORMAT : OK;
        const int32_t na = be_i32(m.p + p + 4);
        if (na < 0) return E_FORMAT;
        if (offs.empty()) natoms = na;
        size_t q = p + 16 + 36 + 4;
        if (na <= 9) {
            q += (size_t)12 * (size_t)na;
        } else {
            if (q + 36 > m.n) break;
            const int32_t nbytes = be_i32(m.p + q + 32);
            if (nbytes < 0) return E_FORMAT;
            q += 36 + (((size_t)nbytes + 3) / 4) * 4;
        }
        if (q > m.n) break;
        offs.push_back(p);
        p = q;
    }
    return OK;
}

// The frame index of the file last read, kept for the next call on the same file (same device, inode, size,
// modification and status-change time): a streaming reader asks for a few hundred frames per call, and walking the record headers of the
// WHOLE file each time -- a page of the mapping per frame -- made a long trajectory quadratic (10 000 frames: as long
// as decoding the chunk itself).
struct FrameIndex { long long ident[7]; std::vector<size_t> offs; int64_t natoms; };

inline int index_frames_cached(const Mapped& m, std::shared_ptr<const FrameIndex>& out)
{
    static std::mutex mu;
    static std::shared_ptr<const FrameIndex> last;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (last && std::memcmp(last->ident, m.ident, sizeof m.ident) == 0) { out = last; return OK; }
    }
    auto idx = std::make_shared<FrameIndex>();
    std::memcpy(idx->ident, m.ident, sizeof m.ident);
    const int st = index_frames(m, idx->offs, idx->natoms);
    if (st != OK) return st;
    out = idx;
    std::lock_guard<std::mutex> lk(mu);
    last = idx;
    return OK;
}

struct BitReader {                       // MSB-first bit stream over [p, end)
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;                    // the next bits of the stream, left-aligned (bit 63 = next bit); the rest is zero
    int nacc = 0;                        // how many of them are valid
    bool overrun = false;
    // at least 56 valid bits afterwards (as long as the stream has them): eight bytes at a time, byte-swapped, while they last
    void refill()
    {
        if (p + 8 <= end) {
            uint64_t v;
            std::memcpy(&v, p, 8);
            v = __builtin_bswap64(v);
            const int take = (63 - nacc) >> 3;          // whole bytes that fit
            acc |= v >> nacc;
            p += take;
            nacc += take * 8;
            acc &= ~0ull << (64 - nacc);                // (the bits of the byte that did not fit whole come with the next refill)
        } else {
            while (nacc <= 56 && p < end) { acc |= (uint64_t)*p++ << (56 - nacc); nacc += 8; }
        }
    }
    uint64_t get64(int nbits)            // 0 <= nbits <= 56
    {
        if (nbits == 0) return 0;
        if (nacc < nbits) {
            refill();
            if (nacc < nbits) { overrun = true; nacc = nbits; }     // past the end: zeros, reported by the caller
        }
        const uint64_t v = acc >> (64 - nbits);
        acc <<= nbits;
        nacc -= nbits;
        return v;
    }
    uint32_t get(int nbits) { return (uint32_t)get64(nbits); }     // 0 <= nbits <= 32
    // three values packed as one mixed-radix number of `nbits` bits whose bytes come least-significant first
    // (the byte-array long division of xdrfile.cpp:545-600 as integer division: 32-bit when the number fits -- every run of
    //  small differences --, 64-bit for full coordinates of ordinary boxes, 128-bit beyond: three ranges of 2^24)
    void get_triple(int nbits, const uint32_t (&radix)[3], int32_t (&out)[3])
    {
        const int nfull = (nbits - 1) >> 3, top = nbits - 8 * nfull;      // whole bytes, then the top 1..8 bits (nbits >= 1)
        if (nbits <= 56) {
            // the whole bytes arrive most-significant-bit first but are the number's LOW bytes in little-endian order
            uint64_t x = 0;
            if (nfull) {
                const uint64_t be = get64(8 * nfull);                        // byte 0 of the number in the top byte of `be`'s field
                x = __builtin_bswap64(be << (64 - 8 * nfull));
            }
            x |= get64(top) << (8 * nfull);
            if (nbits <= 32) {
                uint32_t y = (uint32_t)x;
                const uint32_t q2 = y / radix[2]; out[2] = (int32_t)(y - q2 * radix[2]); y = q2;
                const uint32_t q1 = y / radix[1]; out[1] = (int32_t)(y - q1 * radix[1]);
                out[0] = (int32_t)q1;
            } else {
                const uint64_t q2 = x / radix[2]; out[2] = (int32_t)(uint32_t)(x - q2 * radix[2]);
                const uint64_t q1 = q2 / radix[1]; out[1] = (int32_t)(uint32_t)(q2 - q1 * radix[1]);
                out[0] = (int32_t)(uint32_t)(q1 & 0xffffffffu);
            }
            return;
        }
        unsigned __int128 x = 0;
        int shift = 0, left = nbits;
        while (left > 8) { x |= (unsigned __int128)get(8) << shift; shift += 8; left -= 8; }
        if (left > 0) x |= (unsigned __int128)get(left) << shift;
        out[2] = (int32_t)(uint32_t)(x % radix[2]); x /= radix[2];
        out[1] = (int32_t)(uint32_t)(x % radix[1]); x /= radix[1];
        out[0] = (int32_t)(uint32_t)(x & 0xffffffffu);
    }
};

inline int bits_for(uint32_t size)       // bits needed for values 0 .. size (as the format counts them)
{
    int nb = 0;
    uint64_t lim = 1;
    while ((uint64_t)size >= lim && nb < 32) { ++nb; lim <<= 1; }
    return nb;
}

inline int bits_for_product(const uint32_t (&s)[3])   // bits of (s0 * s1 * s2), the way the format rounds it
{
    // the product as little-endian bytes; bits = 8 * (bytes - 1) + bits of the top byte
    unsigned __int128 prod = (unsigned __int128)s[0] * s[1] * s[2];
    int nbytes = 1;
    unsigned __int128 t = prod;
    while ((t >> 8) != 0) { t >>= 8; ++nbytes; }
    const uint32_t top = (uint32_t)t;
    int nb = 0;
    uint32_t lim = 1;
    while (top >= lim) { ++nb; lim *= 2; }
    return nb + (nbytes - 1) * 8;
}

// Decode the frame at `rec` into column `col` of the frame-fastest outputs (F columns).
// `coords` is addressed as coords[(atom * 3 + axis) * cstride + ccol] (cstride = F, ccol = col writes the final
// array directly; the threaded reader passes a small per-thread block instead and transposes it afterwards).
inline int decode_frame(const Mapped& m, size_t rec, int64_t natoms, int64_t F, int64_t col, float* coords, int64_t cstride,
                        int64_t ccol, float* box, float* time, int32_t* step)
{
    const uint8_t* q = m.p + rec;
    if (be_i32(q) != FRAME_MAGIC) return E_FORMAT;
    if ((int64_t)be_i32(q + 4) != natoms) return E_FORMAT;
    step[col] = be_i32(q + 8);
    time[col] = be_f32(q + 12);
    for (int i = 0; i < 9; ++i) box[(size_t)i * F + col] = be_f32(q + 16 + 4 * i);
    if ((int64_t)be_i32(q + 52) != natoms) return E_FORMAT;
    q += 56;
    auto put = [&](int64_t atom, int axis, float v) { coords[((size_t)atom * 3 + axis) * (size_t)cstride + (size_t)ccol] = v; };
    if (natoms <= 9) {
        for (int64_t a = 0; a < natoms; ++a)
            for (int d = 0; d < 3; ++d) put(a, d, be_f32(q + 4 * (3 * a + d)));
        return OK;
    }
    const float precision = be_f32(q);
    int32_t lo[3], hi[3];
    for (int d = 0; d < 3; ++d) { lo[d] = be_i32(q + 4 + 4 * d); hi[d] = be_i32(q + 16 + 4 * d); }
    int smallidx = be_i32(q + 28);
    const int32_t nbytes = be_i32(q + 32);
    q += 36;
    if (nbytes < 0 || q + nbytes > m.p + m.n) return E_FORMAT;
    uint32_t range[3];
    for (int d = 0; d < 3; ++d) range[d] = (uint32_t)hi[d] - (uint32_t)lo[d] + 1u;
    if (!range[0] || !range[1] || !range[2]) return E_FORMAT;
    int field_bits[3] = {0, 0, 0}, triple_bits = 0;
    const bool wide = (range[0] | range[1] | range[2]) > 0xffffffu;
    if (wide) for (int d = 0; d < 3; ++d) field_bits[d] = bits_for(range[d]);
    else triple_bits = bits_for_product(range);
    if (smallidx < FIRST || smallidx >= NMAGIC) return E_FORMAT;
    int smaller = MAGIC[std::max(FIRST, smallidx - 1)] / 2;
    int smallnum = MAGIC[smallidx] / 2;
    uint32_t small_radix[3] = {(uint32_t)MAGIC[smallidx], (uint32_t)MAGIC[smallidx], (uint32_t)MAGIC[smallidx]};

    const float inv_precision = (float)(1.0 / (double)precision);
    BitReader br{q, q + nbytes};
    int64_t i = 0;          // atoms read
    int64_t w = 0;          // atoms written
    int run = 0;
    auto emit = [&](const int32_t (&c)[3]) -> bool {
        if (w >= natoms) return false;
        for (int d = 0; d < 3; ++d) put(w, d, (float)c[d] * inv_precision);
        ++w;
        return true;
    };
    while (i < natoms) {
        int32_t cur[3];
        if (wide) { for (int d = 0; d < 3; ++d) cur[d] = (int32_t)br.get(field_bits[d]); }
        else br.get_triple(triple_bits, range, cur);
        ++i;
        for (int d = 0; d < 3; ++d) cur[d] = (int32_t)((uint32_t)cur[d] + (uint32_t)lo[d]);
        int32_t prev[3] = {cur[0], cur[1], cur[2]};
        int step_idx = 0;
        if (br.get(1) == 1u) {
            run = (int)br.get(5);
            step_idx = run % 3;
            run -= step_idx;
            --step_idx;                                     // -1 / 0 / +1
        }
        if (run > 0) {
            for (int k = 0; k < run; k += 3) {
                int32_t nxt[3];
                br.get_triple(smallidx, small_radix, nxt);
                ++i;
                for (int d = 0; d < 3; ++d) nxt[d] = (int32_t)((uint32_t)nxt[d] + (uint32_t)prev[d] - (uint32_t)smallnum);
                if (k == 0) {
                    // the first small atom goes out BEFORE the full-precision one it was coded against ...
                    if (!emit(nxt)) return E_FORMAT;
                    if (!emit(prev)) return E_FORMAT;
                } else {
                    if (!emit(nxt)) return E_FORMAT;
                }
                for (int d = 0; d < 3; ++d) prev[d] = nxt[d];   // ... and every small atom is the reference of the next
            }
        } else {
            if (!emit(cur)) return E_FORMAT;
        }
        smallidx += step_idx;
        if (smallidx < FIRST || smallidx >= NMAGIC) return E_FORMAT;
        if (step_idx < 0) {
            smallnum = smaller;
            smaller = smallidx > FIRST ? MAGIC[smallidx - 1] / 2 : 0;
        } else if (step_idx > 0) {
            smaller = smallnum;
            smallnum = MAGIC[smallidx] / 2;
        }
        small_radix[0] = small_radix[1] = small_radix[2] = (uint32_t)MAGIC[smallidx];
        if (small_radix[0] == 0u || br.overrun) return E_FORMAT;
    }
    return (w == natoms) ? OK : E_FORMAT;
}

// natoms / nframes of a file
inline int info(const char* path, int64_t& natoms, int64_t& nframes, std::string& err)
{
    Mapped m;
    if (!m.open_file(path)) { err = std::string("cannot open ") + path; return E_OPEN; }
    std::shared_ptr<const FrameIndex> idx;
    const int st = index_frames_cached(m, idx);
    if (st != OK) { err = "not an XTC file (bad magic number)"; return st; }
    natoms = idx->natoms;
    nframes = (int64_t)idx->offs.size();
    return OK;
}

// The caller's coordinate array is usually FRESH memory (np.zeros / np.empty: pages nobody has touched).  The decode
// threads write it a cache line per row at a time -- every thread into every part of the array -- and their first-touch
// page faults then fight over the same page tables: measured on the 256-core host of an MI355X box (round-3 thread-scaling probe, docs/EXPERIMENTS_r3.md),
// 2 400 frames x 4 507 atoms, 16 threads: 11.9 k frames/s into untouched pages (ONE thread: 18 k) against 241 k once the
// pages exist.  So before the decode every thread touches a CONTIGUOUS slice of the array, one byte per page (zeros:
// each element is overwritten by the decode anyway), after asking for transparent huge pages on the 2 MiB-aligned part.
inline void prefault_output(float* p, size_t n_floats, int nthreads)
{
    const size_t bytes = n_floats * sizeof(float);
    if (nthreads <= 1 || bytes < ((size_t)8 << 20)) return;
    const size_t page = 4096, huge = (size_t)2 << 20;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p), lo = (a + huge - 1) & ~(uintptr_t)(huge - 1), hi = (a + bytes) & ~(uintptr_t)(huge - 1);
#ifdef MADV_HUGEPAGE
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);    // best effort
#endif
    volatile char* c = reinterpret_cast<volatile char*>(p);
    auto touch = [&](int t) {
        size_t b0 = bytes / (size_t)nthreads * (size_t)t, b1 = t == nthreads - 1 ? bytes : bytes / (size_t)nthreads * (size_t)(t + 1);
        for (size_t off = b0; off < b1; off += page) c[off] = 0;
        if (b1 > b0) c[b1 - 1] = 0;
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(touch, t);
    for (auto& th : pool) th.join();
}

// Decode `nsel` frames (indices `sel`, or 0..nsel-1 when sel == nullptr) into frame-fastest arrays of width nsel.
inline int read(const char* path, const int64_t* sel, int64_t nsel, int64_t natoms_expected, float* coords, float* box,
                float* time, int32_t* step, int nthreads, std::string& err)
{
    Mapped m;
    if (!m.open_file(path)) { err = std::string("cannot open ") + path; return E_OPEN; }
    std::shared_ptr<const FrameIndex> idx;
    int st = index_frames_cached(m, idx);
    if (st != OK) { err = "not an XTC file (bad magic number)"; return st; }
    const std::vector<size_t>& offs = idx->offs;
    const int64_t natoms = idx->natoms;
    if (natoms != natoms_expected) { err = "atom count of the file differs from the buffers'"; return E_RANGE; }
    for (int64_t j = 0; j < nsel; ++j) {
        const int64_t f = sel ? sel[j] : j;
        if (f < 0 || f >= (int64_t)offs.size()) { err = "frame index out of range"; return E_RANGE; }
    }
    if (nthreads <= 0) nthreads = (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency()));
    nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(nsel, 1));
    std::vector<int> status((size_t)nthreads, OK);
    // The output is frame-fastest ([natoms, 3, nsel]): one frame is a column with a stride of nsel floats.  Each thread
    // therefore decodes FB consecutive columns into a [3*natoms][FB] block of its own (FB floats = one cache line per
    // row) and copies the block row by row into place, instead of scattering single floats a page apart.
    // (narrower blocks -- 8 or 4 frames -- so that a streamed chunk of 256 frames keeps 64 threads busy instead of 16 were
    //  measured: 82 k -> 44 k frames/s end to end; a caller that wants more threads per chunk asks for bigger chunks)
    constexpr int64_t FB = 16;
    const int64_t nblocks = (nsel + FB - 1) / FB;
    // (round 4: every frame of a block is decoded into a CONTIGUOUS row of 3 * natoms floats -- the decoder's writes are
    //  sequential, a 30 000-atom frame stays in the core's L2 -- and the block is transposed into place afterwards, FB
    //  sequential read streams against one line per row; decoding straight into a [3 * natoms][FB] block touched a cache
    //  line per value: 5.8 MB per thread and block for such a frame)
    auto work = [&](int t) {
        const size_t row = (size_t)natoms * 3;
        std::vector<float> blk(row * FB);
        for (int64_t bidx = t; bidx < nblocks; bidx += nthreads) {
            const int64_t j0 = bidx * FB, nb = std::min<int64_t>(FB, nsel - j0);
            for (int64_t k = 0; k < nb; ++k) {
                const int64_t f = sel ? sel[j0 + k] : j0 + k;
                const int s = decode_frame(m, offs[(size_t)f], natoms, nsel, j0 + k, blk.data() + (size_t)k * row, 1, 0, box, time, step);
                if (s != OK) { status[(size_t)t] = s; return; }
            }
            const float* const b0 = blk.data();
            if (nb == FB) {
                for (size_t r = 0; r < row; ++r) {
                    float* __restrict__ d = coords + r * (size_t)nsel + (size_t)j0;
#pragma GCC unroll 16
                    for (int64_t k = 0; k < FB; ++k) d[k] = b0[(size_t)k * row + r];
                }
            } else {
                for (size_t r = 0; r < row; ++r)
                    for (int64_t k = 0; k < nb; ++k) coords[r * (size_t)nsel + (size_t)j0 + (size_t)k] = b0[(size_t)k * row + r];
            }
        }
    };
    nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(nblocks, 1));
    prefault_output(coords, (size_t)natoms * 3 * (size_t)nsel, nthreads);
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (int s : status)
        if (s != OK) { err = "corrupt XTC frame"; return s; }
    return OK;
}

}  // namespace xtc
}  // namespace mkamd
