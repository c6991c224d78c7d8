/*
 * mkamd_voxel.h -- C ABI of libmkamd.so: MI355X (gfx950) voxel-descriptor kernels.
 *
 * This is the drop-in boundary for ONE hot path of Acellera/moleculekit:
 *   moleculekit.occupancy_utils.calculate_occupancy          (occupancy_utils/occupancy_utils.pyx:34-61)
 *   called from tools.voxeldescriptors._getOccupancyC         (tools/voxeldescriptors.py:515-533)
 *   under tools.voxeldescriptors.getVoxelDescriptors          (tools/voxeldescriptors.py:251-365)
 *   with lattice centres from getCenters/_getGridCenters      (tools/voxeldescriptors.py:125-132,197-248)
 *   and, for periodic frames, the orthorhombic minimum image  (distance_utils/distance_utils.pyx:49-52).
 *
 * Plain pointers and sizes only (no torch / numpy types).  Every function returns an int status
 * (MKAMD_OK == 0) and never throws (every entry point is a function-try-block); mkamd_last_error() gives the message for the calling thread.
 * All device work is enqueued on the context's HIP stream.  "_host" entry points take host
 * pointers, copy in/out and synchronise; "_dev" entry points take device pointers, are
 * asynchronous and leave results resident in HBM.
 *
 * Array layouts (all C-contiguous):
 *   coords   float32 [N,3]          (Molecule.coords[:, :, frame], molecule.py:205-232)
 *   sigmas   float64|float32 [N,C]  per-atom per-channel radius, 0 = atom not in channel
 *                                   (voxeldescriptors.py:332-335)
 *   centers  float64 [V,3]
 *   features float32 [V,C]          voxel-major / channel-minor, V flattened x slowest, z fastest
 *                                   (same order as the reference's float64 [V,C], voxeldescriptors.py:531)
 */
#ifndef MKAMD_VOXEL_H
#define MKAMD_VOXEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKAMD_OK 0
#define MKAMD_EINVAL 1    /* bad argument (message says which) */
#define MKAMD_EHIP 2      /* HIP runtime error */
#define MKAMD_ENODEV 3    /* no usable GPU */
#define MKAMD_EOVERFLOW 4 /* more periodic images than max_images_per_atom allowed */
#define MKAMD_EBOX 5      /* periodic box edge <= 2 x cutoff (10 A) or too many images */
#define MKAMD_ENOMEM 6    /* host allocation failed inside the library (nothing C++ ever crosses this boundary) */

typedef struct mkamd_ctx mkamd_ctx;

/* library / device --------------------------------------------------------------------------- */
const char* mkamd_version(void);
const char* mkamd_last_error(void);
int mkamd_device_count(int* count);

/* One context per device (and per host thread that wants to drive it concurrently). Owns the
 * workspace (cell lists, staging buffers) and a HIP stream. */
int mkamd_ctx_create(int device, mkamd_ctx** ctx);
int mkamd_ctx_destroy(mkamd_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL is the legacy
 * default stream (a valid choice); (void*)-1 restores the context's own stream. */
int mkamd_ctx_set_stream(mkamd_ctx* ctx, void* hip_stream);
/* Give up a host call that was begun (mkamd_voxelize_lattice_host_begin) and will not be ended -- the caller lost its second
 * half to an exception, say: the call's kernels are drained, its result is dropped, the context takes every entry point
 * again.  No-op without a pending call.  (The Python binding calls it when the function `voxelize_lattice_begin` returned is
 * garbage-collected without having been called, so that a dropped `end` cannot block a shared default context.) */
int mkamd_ctx_abandon_pending(mkamd_ctx* ctx);
/* Wait for the stream and report asynchronous errors of "_dev" calls (MKAMD_EOVERFLOW/EBOX). */
int mkamd_ctx_synchronize(mkamd_ctx* ctx);
/* Non-blocking form for streaming callers: reports (and clears) an asynchronous error of a "_dev" lattice call that
 * has ALREADY finished -- the device-side flag is mirrored into pinned host memory by every call, so the clean path is
 * one host read.  An error still in flight is reported by a later poll or by mkamd_ctx_synchronize. */
int mkamd_ctx_poll_errors(mkamd_ctx* ctx);
/* name (<= len bytes), compute units, HBM bytes, gcn arch string e.g. "gfx950..." */
int mkamd_ctx_device_info(mkamd_ctx* ctx, char* name, size_t len, int* compute_units,
                          uint64_t* hbm_bytes, char* arch, size_t arch_len);
/* Tile depth K (x-planes per lane) of the lattice kernel: 0 = automatic, 4 or 8. */
int mkamd_ctx_set_tile_k(mkamd_ctx* ctx, int k);
/* 1: always take the general tile-kernel path (per-pair cutoff test, arbitrary per-entry sigma)
 * instead of the class-sorted one; results are bit-identical (testing / A-B benchmarking). */
int mkamd_ctx_set_force_general(mkamd_ctx* ctx, int on);
/* LDS entry capacity of a tile of the lattice kernel: tier 0/1/2 = 640/768/1024 entries (leaner = more
 * waves per CU = faster, as long as the tiles fit; tiles that do not are finished by a second, multi-round
 * kernel).  -1 (default) = adaptive: the leanest tier that at most 5 % of the tiles of the previous calls
 * on this context overflowed.  Results are bit-identical whatever the tier. */
int mkamd_ctx_set_lds_tier(mkamd_ctx* ctx, int tier);
/* Binning pre-pass of the lattice path: 0 = the multi-kernel chain (any batch), 1 = one launch with one
 * workgroup per item (cell grid of an item must fit 8191 LDS counters; meant for items of up to a few thousand
 * atoms: ligand poses, pockets), -1 (default) = per-item when it fits and the batch averages <= 4096 atoms per
 * item.  Results are bit-identical either way. */
int mkamd_ctx_set_prepass_mode(mkamd_ctx* ctx, int mode);
/* Cell edge of the binning: 0 (default) = the power of two >= the cutoff radius in voxels, 1 = half of that (fewer
 * candidates per tile to cull, eight times the cell counters; measured slightly slower -- kept for A-B benchmarking).
 * Values agree to float32 noise (the cell-relative offsets are rounded at a different magnitude), both within the
 * 1e-5 parity bound. */
int mkamd_ctx_set_fine_cells(mkamd_ctx* ctx, int on);
/* Binning of open-boundary calls of one channel group in the DIRECT layout (records written in place at
 * cell * capacity + rank; surplus of a full cell in the item's spill area).  Results are bit-identical in every mode.
 *  -1 (default): SMALL calls (the reference's pattern: one molecule per call; at most 1 024 tile waves) take the
 *      one-launch pre-pass -- the class table lives across the calls of the context and is extended on the spot, atoms
 *      with several sigmas and more classes than ids are handled in place, nothing is enqueued behind it: three launches
 *      per call instead of five; BIG calls that are not pipelined (mkamd_ctx_set_pipelining) take the one-pass form
 *      k_bin_direct -- class ids from the table the previous call on the workspace left, the count / scan / fill chain
 *      enqueued behind it as a device-side fallback (a few thousand workgroups that leave at once unless the pass gave up:
 *      a sigma the table lacks, an atom with several sigmas, a full spill area): cfg2, 256 grids in order 2.41-2.43 against
 *      2.47-2.48 ms, pre-pass traffic 0.9 GB instead of 1.7; pipelined calls keep the chain (the one-pass form gains
 *      nothing beside the previous call's tile kernel);
 *   0: the chain (or the per-item pre-pass) always;
 *   1: the one-pass form for every big call, pipelined or not;
 *   2: the one-launch pre-pass for calls of any size whose geometry allows it (tests). */
int mkamd_ctx_set_direct_binning(mkamd_ctx* ctx, int mode);
/* Tolerance-aware reach (opt-in; 0 = off, the default: the reference's hard 5 A cutoff for every atom,
 * occupancy_utils.pyx:53).  An (atom, channel) entry is worth 1 - exp(-(sigma/r)^12) < eps beyond r = sigma * eps^(-1/12)
 * (hydrogens, sigma 1.1 A: 3.48 A at eps = 1e-6), and the channel value is a maximum over entries, so dropping an
 * entry wherever it is worth less than eps moves no value by more than eps.  With eps > 0 every atom is culled per
 * TILE at min(5 A, that radius) (rounded up to one of four levels); voxels of a tile the atom still reaches see it at
 * any distance below 5 A as before.  Results stay within eps (+ the float32 noise of the exact mode, <= 3.7e-6 seen) of the
 * reference; they are no longer bit-identical between tilings / kernels.  eps in [0, 1e-5]. */
int mkamd_ctx_set_value_tolerance(mkamd_ctx* ctx, double eps);
/* Waves per tile of the lattice kernel: 0 = one (throughput: big batches), 1 = a team that shares the tile's candidate
 * traversal (one survivor list) and splits its sorted entries into one range per wave (latency: one grid per call, the
 * reference's own usage) -- four waves, eight for launches of up to 256 tiles of depth 4; 4 / 8 / 16 = a team of exactly
 * that many (8, 16: depth-4 tiles only); -1 (default) = a team when the whole launch has fewer tiles than the chip has
 * SIMDs (ligand-sized items take the workgroup-per-item kernel below instead, whatever the batch size).  Results are
 * bit-identical either way. */
int mkamd_ctx_set_tile_team(mkamd_ctx* ctx, int mode);
/* A workgroup per ITEM instead of a wave per tile: 1 = always (when no team is used), 0 = never, -1 (default) = for
 * ligand-sized items (<= 96 atoms on average) on the per-item pre-pass, any batch size: the item's entries are sorted
 * once for all its tiles (once per chunk of its tiles while the batch is too small to fill the chip) instead of once
 * per tile.  Results are bit-identical either way. */
int mkamd_ctx_set_tile_items(mkamd_ctx* ctx, int mode);
/* The exact cut-off fix-up of a TOPOLOGY call whose molecule has wide sigmas (ions: sigma > 1.81 A): 0 (default) = the call's last
 * launch lists the (voxel, channel) values to re-decide and one more launch recomputes them with many waves per value (the item's
 * atoms in slices of 2 048, joined by an atomic maximum); -1 = they are recomputed inside the last launch, one wave per value over
 * all of the item's atoms (rounds 3-5; what every other kind of call does).  Results are bit-identical either way. */
int mkamd_ctx_set_exact_redo(mkamd_ctx* ctx, int mode);
/* Opt-in software pipelining ACROSS calls of mkamd_voxelize_lattice_dev (off by default): the binning
 * pre-pass of a call (latency / atomic bound) runs on an internal stream beside the tile kernel (VALU bound)
 * of the previous call, on a second workspace set.  Results still appear in order on the context's stream.
 * Contract while it is on: the inputs of a call must not be produced by work enqueued on the context's
 * stream AFTER the previous voxelize call (the pre-pass is only ordered after everything before that call's
 * tile kernel) and must stay untouched until the call's features have been consumed. */
int mkamd_ctx_set_pipelining(mkamd_ctx* ctx, int on);
/* The same pipelining for ONE call, by promise (what the package's own streaming drivers use -- batch.iterVoxelize*,
 * distributed.ShardedVoxelizer -- so the context-wide knob above is not needed to get it): a one-shot statement about
 * the NEXT mkamd_voxelize_lattice(_aug)_dev call on this context.  Its inputs (coords, offsets, sigmas, origins, box,
 * affine) are complete once `hip_event` (a hipEvent_t recorded on ANY stream of this device) has completed -- NULL: they
 * are complete already and were not produced by work enqueued on the context's stream after the previous voxelize
 * call, e.g. a resident shard -- and they stay untouched until the call's features have been consumed.  The call's
 * pre-pass waits for the event on whichever stream runs it and may then run beside the previous call's tile kernel.
 * The promise is consumed by that call, pipelined or not (a small call runs in order and still waits for the event).  Host
 * entry points ("_host", mkamd_calculate_occupancy) neither honour nor consume it: a promise made on a shared context
 * survives another caller's host call made before the device call it is about. */
int mkamd_ctx_promise_inputs(mkamd_ctx* ctx, void* hip_event);
/* Drop a promise that no call has consumed (a driver that stops between the promise and its call). */
int mkamd_ctx_withdraw_promise(mkamd_ctx* ctx);
/* How many lattice calls of this context have run their pre-pass beside a previous call's tile kernel so far (tests
 * and benchmarks check with it that a driver really is pipelined). */
int mkamd_ctx_pipelined_calls(mkamd_ctx* ctx, int64_t* n);
/* Name of the tile kernel the last lattice call launched, as a profiler prints it ("mkamd::k_voxelize_tiles_lean<8, 640>";
 * the team kernel without its team size; empty before the first call): what bench.py reports as `roofline.kernel`. */
int mkamd_ctx_last_tile_kernel(mkamd_ctx* ctx, char* name, size_t name_cap);
/* Per-kernel timing of the tile kernel with HIP events on the context's stream (bench.py's
 * roofline leg): enable, run, then read back the accumulated time and launch count (resets). */
int mkamd_ctx_enable_kernel_timing(mkamd_ctx* ctx, int enable);
int mkamd_ctx_read_kernel_timing(mkamd_ctx* ctx, double* total_ms, int64_t* launches);
/* The shader clock the device sustains UNDER a given load (bench.py reports it next to the roofline: boxes run the same kernel
 * -- the same cycle count -- at 2.0 to 2.2 GHz).  Enqueues ONE wave on `hip_stream` (the caller's: a stream other than the one
 * the load runs on) that spins for `microseconds` of the fixed 100 MHz reference counter (s_memrealtime) and stores
 * d_ticks2[0] = shader clock ticks (s_memtime) and d_ticks2[1] = reference ticks that passed meanwhile (device memory, two
 * uint64): clock = d_ticks2[0] / d_ticks2[1] x 100 MHz.  Asynchronous; nothing of the context is touched. */
int mkamd_clock_probe_dev(mkamd_ctx* ctx, void* hip_stream, int64_t microseconds, uint64_t* d_ticks2);

/* (1) calculate_occupancy, exact reference contract -------------------------------------------
 * Replaces occupancy_utils.pyx:34-61: for every centre/channel
 *     results[v,c] = max(results[v,c], max_a{ 1-exp(-(sigmas[a,c]/|coords[a]-centers[v]|)^12) :
 *                                               |.|^2 < 25, sigmas[a,c] != 0 })
 * max-accumulating IN PLACE into the caller's float64 results (caller zero-fills, voxeldescriptors.py:531).
 * Centres that form a getCenters lattice (what the reference's only caller passes, voxeldescriptors.py:356) are
 * recognised and take the tiled lattice kernels of (2), values within 1e-5; any other centre list takes the pairwise
 * kernel of (3): distances in double, values float32-accurate (<= 1e-6 abs). */
int mkamd_calculate_occupancy(mkamd_ctx* ctx, const double* centers, int64_t n_centers,
                              const float* coords, int64_t n_atoms, const double* sigmas,
                              int32_t n_channels, double* results);

/* (1b) the same contract on the HOST, no context and no GPU (the reference's `method="C"` dispatch,
 * tools/voxeldescriptors.py:355-360, lands in a CPU loop): the library's own double-precision implementation -- atoms in a
 * uniform cell list, every centre against the 27 cells around it, the reference's arithmetic per pair (float32 coordinates
 * promoted, strict d^2 < 25, x^12 as x3*x3*x3*x3, value > old), centres split over host threads.  A maximum does not
 * depend on the order of its candidates: results are the reference's bit for bit.  EXPLICIT only: no GPU entry point falls
 * back to it (without a device they fail with MKAMD_ENODEV); the Python package reaches it through method="CPU" /
 * occupancy_utils.calculate_occupancy_cpu.  `_threads`: n_threads <= 0 = automatic (MKAMD_CPU_THREADS, else one thread for
 * small calls and up to 32 otherwise). */
int mkamd_calculate_occupancy_cpu(const double* centers, int64_t n_centers, const float* coords, int64_t n_atoms,
                                  const double* sigmas, int32_t n_channels, double* results);
int mkamd_calculate_occupancy_cpu_threads(const double* centers, int64_t n_centers, const float* coords, int64_t n_atoms,
                                          const double* sigmas, int32_t n_channels, double* results, int32_t n_threads);

/* (2) explicit (arbitrary) centres, float32 output, optional orthorhombic box (double[3], A;
 * NULL = not periodic).  sigmas_are_f64: 1 -> const double*, 0 -> const float*. */
int mkamd_occupancy_centers_host(mkamd_ctx* ctx, const double* centers, int64_t n_centers,
                                 const float* coords, int64_t n_atoms, const void* sigmas,
                                 int sigmas_are_f64, int32_t n_channels, const double* box,
                                 float* features);
int mkamd_occupancy_centers_dev(mkamd_ctx* ctx, const double* d_centers, int64_t n_centers,
                                const float* d_coords, int64_t n_atoms, const void* d_sigmas,
                                int sigmas_are_f64, int32_t n_channels, const double* box_host,
                                float* d_features);

/* (3) lattice grids, batched: the hot path ------------------------------------------------------
 * B independent items (molecules / poses / trajectory frames) packed back to back:
 *   coords        float32 [sumN,3]
 *   atom_offsets  int64   [B+1]      item b owns atoms [atom_offsets[b], atom_offsets[b+1])
 *   sigmas        [sumN,C]
 *   origins       float64 [B,3]      bb_min of item b: voxel (0,0,0)'s centre (getCenters :234-243;
 *                                    NO half-voxel shift, voxel i sits at origin + i*voxelsize)
 *   nvoxels       int32   [3]        grid size shared by the batch (getCenters :236-242)
 *   box           float32 [B,3] or NULL: per-item orthorhombic box (Molecule.box[:, frame]); when
 *                                    given, coord-centre is minimum-imaged (every edge must be > 10 A)
 *   features      float32 [B,V,C]    V = nx*ny*nz
 * max_images_per_atom bounds the periodic images of one atom that can fall inside the grid+halo
 * (1 when every box edge >= grid extent + 10 A; ignored without box).  Pass 0 to let the "_host"
 * variant compute it from the boxes. */
int mkamd_voxelize_lattice_host(mkamd_ctx* ctx, int32_t n_items, const float* coords,
                                const int64_t* atom_offsets, const void* sigmas,
                                int sigmas_are_f64, int32_t n_channels, const double* origins,
                                const int32_t* nvoxels, double voxelsize, const float* box,
                                int32_t max_images_per_atom, float* features);
/* the same with float64 features [B,V,C] -- the dtype the reference's _getOccupancyC returns (voxeldescriptors.py:531);
 * values are the float32 results widened (in the pass that takes them out of the pinned result buffer) */
int mkamd_voxelize_lattice_host_f64(mkamd_ctx* ctx, int32_t n_items, const float* coords,
                                    const int64_t* atom_offsets, const void* sigmas,
                                    int sigmas_are_f64, int32_t n_channels, const double* origins,
                                    const int32_t* nvoxels, double voxelsize, const float* box,
                                    int32_t max_images_per_atom, double* features);
/* The same call in two halves, for a caller with host work of its own to do while the device computes (the drop-in
 * getVoxelDescriptors copies its cached voxel centres there, voxeldescriptors.py:245-247): `begin` checks the arguments,
 * ships the inputs and enqueues the kernels (the input arrays must stay valid until `end`); `end` waits and writes the
 * result into ONE of the two arrays (the other NULL) of `n_values` elements -- which must be the n_items * n_voxels *
 * n_channels values the pending call produced (else MKAMD_EINVAL, nothing is written, the call is abandoned).  One call at
 * a time per context; a `begin` that is never ended is abandoned by the next `begin` (or host call) or by
 * mkamd_ctx_abandon_pending; between the two halves only queries, mkamd_grid_centers_*, mkamd_copy_to_host,
 * mkamd_frames_to_items_dev and mkamd_xtc_decode_dev are accepted on the context -- any other entry point would regrow or
 * overwrite what `end` hands back and returns MKAMD_EINVAL (it does not guess whether the pending call is still wanted). */
int mkamd_voxelize_lattice_host_begin(mkamd_ctx* ctx, int32_t n_items, const float* coords,
                                      const int64_t* atom_offsets, const void* sigmas,
                                      int sigmas_are_f64, int32_t n_channels, const double* origins,
                                      const int32_t* nvoxels, double voxelsize, const float* box,
                                      int32_t max_images_per_atom);
int mkamd_voxelize_lattice_host_end(mkamd_ctx* ctx, float* features, double* features_f64, uint64_t n_values);
int mkamd_voxelize_lattice_dev(mkamd_ctx* ctx, int32_t n_items, const float* d_coords,
                               const int64_t* d_atom_offsets, int64_t total_atoms,
                               const void* d_sigmas, int sigmas_are_f64, int32_t n_channels,
                               const double* d_origins, const int32_t* nvoxels, double voxelsize,
                               const float* d_box, int32_t max_images_per_atom,
                               float* d_features);

/* (3b) the same with a fused per-item rigid transform (data augmentation): affine float64 [B,12] = row-major
 * 3x3 matrix M followed by a translation t; every atom of item b is voxelized at float32(M x + t) -- what
 * `rotateCoordinates` (voxeldescriptors.py:78-114) followed by getVoxelDescriptors(usercoords=...) computes. NULL = none. */
int mkamd_voxelize_lattice_aug_dev(mkamd_ctx* ctx, int32_t n_items, const float* d_coords,
                                   const int64_t* d_atom_offsets, int64_t total_atoms,
                                   const void* d_sigmas, int sigmas_are_f64, int32_t n_channels,
                                   const double* d_origins, const int32_t* nvoxels, double voxelsize,
                                   const float* d_box, int32_t max_images_per_atom,
                                   const double* d_affine, float* d_features);

/* (3c) TOPOLOGY REUSE for trajectory-shaped calls (round 5).  The sigmas depend on the topology only (voxeldescriptors.py:332-335:
 * vdW radius x channel mask), so for the frames of a trajectory everything the pre-pass derives from them -- an atom's channel
 * words, the sigma classes and the class table, its class ids, whether any sigma needs the exact cut-off fix-up -- is the same
 * in every call.  A handle holds it on the device (built once: three small launches and one read-back; the library keeps its
 * own copy of the sigmas, so the caller's may change or go away); mkamd_voxelize_lattice_topo_dev voxelizes `n_items` sets
 * of coordinates of that molecule -- atom_offsets[b] = b * n_atoms: every item the topology's atom count long, CHECKED on the
 * device (an item of another length raises MKAMD_EINVAL at the next mkamd_ctx_synchronize / _poll_errors) -- with 4 bytes of
 * ids per atom where the plain call reads the sigma row, divides in double and discovers the classes again.  Features are the
 * plain call's BIT FOR BIT.  The handle is for one voxel size and channel count (the call's must match: MKAMD_EINVAL);
 * molecules with more than 15 distinct sigma values have no class ids to reuse (creation fails with MKAMD_EINVAL: use the plain
 * call), and the A-B modes force_general / value tolerance are refused for topology calls.  Promises
 * (mkamd_ctx_promise_inputs) apply as for the plain call.  What it is for: batch.iterVoxelizeTrajectory / iterVoxelizeXTC /
 * ShardedVoxelizer(shared_sigmas=True) use it by construction. */
typedef struct mkamd_topology mkamd_topology;
int mkamd_topology_create_dev(mkamd_ctx* ctx, const void* d_sigmas, int sigmas_are_f64, int64_t n_atoms, int32_t n_channels,
                              double voxelsize, mkamd_topology** topology);
int mkamd_topology_create_host(mkamd_ctx* ctx, const void* sigmas, int sigmas_are_f64, int64_t n_atoms, int32_t n_channels,
                               double voxelsize, mkamd_topology** topology);
/* Waits for the context's streams (calls that read the handle), then frees it.  ctx may be NULL (the whole device is drained). */
int mkamd_topology_destroy(mkamd_ctx* ctx, mkamd_topology* topology);
int mkamd_topology_info(const mkamd_topology* topology, int64_t* n_atoms, int32_t* n_channels, double* voxelsize, int32_t* has_wide_sigmas);
int mkamd_voxelize_lattice_topo_dev(mkamd_ctx* ctx, int32_t n_items, const float* d_coords, const int64_t* d_atom_offsets,
                                    int64_t total_atoms, const mkamd_topology* topology, const double* d_origins,
                                    const int32_t* nvoxels, double voxelsize, const float* d_box, int32_t max_images_per_atom,
                                    const double* d_affine, float* d_features);

/* (4) lattice centres (voxeldescriptors.py:125-132 + :245-247), float64 [V,3], bit-exact with the
 * reference's numpy arithmetic: centre = fl64(index*voxelsize) + bb_min. */
int mkamd_grid_centers_host(mkamd_ctx* ctx, const double* bb_min, const int32_t* nvoxels,
                            double voxelsize, double* centers);
int mkamd_grid_centers_dev(mkamd_ctx* ctx, const double* bb_min, const int32_t* nvoxels,
                           double voxelsize, double* d_centers);

/* Host helper: touch `bytes` of a buffer the caller is about to have filled (zeros, one byte per page, by host threads in
 * contiguous slices, after a transparent-huge-page hint) -- what the _host entry points do themselves for results of
 * 32 MB and more; for callers that fill a fresh array chunk by chunk through the _dev entry points.  No-op below 8 MB. */
int mkamd_prefault(void* buffer, uint64_t bytes);
/* Synchronous device -> host copy on the context's stream (hipMemcpyAsync + wait): what the _host entry points use for
 * their results, for callers of the _dev entry points that collect chunks into a host array. */
int mkamd_copy_to_host(mkamd_ctx* ctx, void* host_dst, const void* device_src, uint64_t bytes);
/* Asynchronous device -> device copy on the context's stream: how a caller keeps a context-owned device result (the contact
 * list of mkamd_contacts_trajectory_dev) past the next call that reuses its memory. */
int mkamd_copy_dev(mkamd_ctx* ctx, void* device_dst, const void* device_src, uint64_t bytes);
/* Trajectory slab -> packed items, on the device, on a stream of the CALLER's choice (a copy stream: batch._stream_voxelize
 * prepares chunk k+1 there while chunk k is voxelized): `d_src` holds `rows` rows (atoms x 3 of Molecule.coords, or the 3 box
 * lengths) of `src_pitch` floats each, frame fastest; frames [0, n_frames) of it are written frame-major,
 * d_dst[f * rows + r] = d_src[r * src_pitch + f] * scale (one float32 multiply: nm -> Angstrom for XTC input, 1 otherwise).
 * Replaces reader-side work of readers.py:1848-1859 (the x10 of XTCread) and the host transpose of a [N,3,F] slab. */
int mkamd_frames_to_items_dev(mkamd_ctx* ctx, void* hip_stream, const float* d_src, int64_t rows, int64_t src_pitch,
                              int64_t n_frames, float scale, float* d_dst);

/* (5) the inverse, on the host (no context, no GPU): is `centers` float64 [V,3] a getCenters lattice -- bb_min +
 * fl64(index * voxelsize), x slowest / z fastest (voxeldescriptors.py:125-132, :245-247) -- to 1e-9 A?  Returns 1 and
 * fills bb_min[3], nvoxels[3], voxelsize, else 0 (also for NaNs and fewer than two centres).  What the drop-in
 * `_getOccupancyC` asks of the `usercenters` it is handed, before it chooses between (2) and (3). */
int mkamd_lattice_from_centers(const double* centers, int64_t n_centers, double* bb_min, int32_t* nvoxels, double* voxelsize);

#ifdef __cplusplus
}
#endif
#endif /* MKAMD_VOXEL_H */
