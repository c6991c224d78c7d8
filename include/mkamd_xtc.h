/* mkamd_xtc.h -- C ABI of the XTC trajectory decoder in libmkamd.so ("trajectory feeding").
 *
 * Host-side, no GPU involved: what the reference does in moleculekit/fileformats/xtc (xtc.pyx: read_xtc :34-53,
 * read_xtc_frames :57-81, get_xtc_natoms / get_xtc_nframes; src/xdrfile.cpp: xdrfile_decompress_coord_float :749-983;
 * src/xtc_src.cpp: xtc_read_new :195-262, xtc_read_frame :471-624).  Same float32 bits out.  Frames are decoded in
 * parallel on host threads (they are independent records); the reference decodes them one by one.
 *
 * Status: 0 = ok; non-zero = error, message via mkamd_last_error() (mkamd_voxel.h).
 */
#ifndef MKAMD_XTC_H
#define MKAMD_XTC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Atom count and number of (complete) frames of an XTC file  -- get_xtc_natoms / get_xtc_nframes (xtc.pyx:14-31). */
int mkamd_xtc_info(const char* path, int64_t* n_atoms, int64_t* n_frames);

/* Decode frames into caller-owned arrays with the reference's layouts (frame index fastest):
 *   coords f32 [n_atoms, 3, n_sel] in nm, box f32 [3, 3, n_sel] (box vectors, nm), time f32 [n_sel] (ps), step i32 [n_sel].
 * frames: n_sel frame indices (any order, repeats allowed), or NULL for frames 0 .. n_sel-1
 * (read_xtc = NULL with n_sel = n_frames; read_xtc_frames = an index list).  n_atoms must match the file.
 * n_threads: host threads to decode with, 0 = automatic (up to 32). */
int mkamd_xtc_read(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, float* coords, float* box,
                   float* time, int32_t* step, int32_t n_threads);

/* Decoding ON THE DEVICE (round 4; csrc/xtc_gpu.h).  Frames are independent records; inside a frame only the WALK of the bit
 * stream is serial (where an atom's bits start depends on the flag / run length after every full-precision atom before it:
 * xdrfile.cpp:749-983).  So a GPU lane walks a frame and records where each group -- a full atom and the run of small atoms
 * after it -- starts, then a thread per group decodes the numbers; the host only parses the record headers and copies the
 * records' bytes.
 *   mkamd_xtc_chunk_desc   headers of the selected frames -> one 64-byte descriptor per frame (desc_out: n_sel x 64 bytes, the
 *                          layout of mkamd::XtcFrameDesc in csrc/xtc_gpu.h; data offsets relative to *byte_lo), the byte range
 *                          [*byte_lo, *byte_hi) of the file that holds their records, and what the host path returns besides the
 *                          coordinates: box vectors f32 [3,3,n_sel] (nm), time f32 [n_sel] (ps), step i32 [n_sel]
 *   mkamd_xtc_copy_bytes   that byte range into caller memory (pinned staging), by a few host threads (pread)
 *   mkamd_xtc_byte_range   (round 6) the byte range of the selection from the frame index alone -- a streaming reader copies the
 *                          bytes FIRST and then parses the headers out of its copy:
 *   mkamd_xtc_chunk_desc_mem   mkamd_xtc_chunk_desc from a host copy of the file's bytes [bytes_lo, bytes_hi) (the same results and
 *                          checks; a header read through a fresh mapping of the file costs a page fault: 2 ms per 2 048 frames)
 *   mkamd_xtc_decode_work_bytes   size of the device work buffer a decode of n_frames x n_atoms needs (8 bytes per atom: the
 *                          group records)
 *   mkamd_xtc_decode_dev   (needs mkamd_voxel.h's context) d_bytes / d_desc = device copies of the two, d_bytes
 *                          MKAMD_XTC_PAD bytes LONGER than the range (the kernels read ahead of what they use) and 4-byte
 *                          aligned: coordinates float32 [n_frames, n_atoms, 3] -- frame-major, the voxelizer's packed items --
 *                          times `scale` (10 = nm -> Angstrom), with the float32 operations of the host path ((float)int *
 *                          (1 / precision), then * scale: the same bits); d_status[f] = 0 ok, 1 corrupt stream, 2 outside what
 *                          the device path takes (a mixed-radix number of more than 64 bits, a frame of >= 2^21 atoms or
 *                          >= 512 MB: take the host decoder for such a file; that frame's coordinates are not written).
 *                          d_work: 8-byte aligned, only used during the call's kernels.  Asynchronous on `hip_stream`. */
#define MKAMD_XTC_PAD 1024
int mkamd_xtc_chunk_desc(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, void* desc_out,
                         int64_t* byte_lo, int64_t* byte_hi, float* box, float* time, int32_t* step);
int mkamd_xtc_copy_bytes(const char* path, int64_t byte_lo, int64_t byte_hi, void* dst, int32_t n_threads);
int mkamd_xtc_byte_range(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, int64_t* byte_lo, int64_t* byte_hi);
int mkamd_xtc_chunk_desc_mem(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, const void* bytes, int64_t bytes_lo,
                             int64_t bytes_hi, void* desc_out, int64_t* byte_lo, int64_t* byte_hi, float* box, float* time, int32_t* step);
uint64_t mkamd_xtc_decode_work_bytes(int64_t n_frames, int64_t n_atoms);
struct mkamd_ctx;
int mkamd_xtc_decode_dev(struct mkamd_ctx* ctx, void* hip_stream, const void* d_bytes, const void* d_desc, int64_t n_frames,
                         int64_t n_atoms, float scale, float* d_xyz, int32_t* d_status, void* d_work, uint64_t work_bytes);

#ifdef __cplusplus
}
#endif
#endif
