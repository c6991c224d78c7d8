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

#ifdef __cplusplus
}
#endif
#endif
