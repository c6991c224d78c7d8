// mk_diagnostics.h -- the ONLY place where the product kernels can be changed by a -D flag.
//
// A RELEASE build (moleculekit_amd/_build.py, __graft_entry__.build()) defines none of the macros below: MK_DIAG is 0,
// every MK_PHASE_* / MK_BIN_* hook expands to nothing, and kernels.h holds no other conditional code.  A DIAGNOSTICS build
// must say so with -DMKAMD_DIAGNOSTICS_BUILD (tools/build_variant.sh does; _build.py never does): without it any of the
// knobs is a compile error, and with it mkamd_version() carries the word DIAGNOSTICS, which moleculekit_amd/_lib.py
// refuses to load unless MKAMD_ALLOW_DIAGNOSTICS=1 -- a library that returns wrong numbers on purpose cannot be picked
// up by accident.
//
//   MK_DIAG=<bits>     compile parts of voxelize_tile OUT to time what they cost (WRONG VALUES by construction):
//                      1 pair loops, 2 class flushes, 4 epilogue arithmetic, 8 placement + all class work,
//                      16 cull / histogram traversal, 32 the histogram pass over the survivor list, 64 the placement
//                      traversal (profiles/r4_tile_time_map.txt, profiles/r5_tile_instruction_map.txt)
//   MK_PHASE_TIMERS    cycle counters around the phases of a tile wave (tools/phase_timers.py); with MK_BIN_TIMERS the
//                      sections of k_bin_direct instead (tools/bin_timers.py).  Values stay right, timing does not.
// Experiments that were measured and ruled out (single-entry groups evaluated directly, the rolled channel loop, the
// trip-cut timing build, the kernel without its general path, LDS padding, the two-traversal kernel of round 1) are in
// docs/EXPERIMENTS_r3.md / _r4.md and in the history of kernels.h, not in the product source.
#pragma once

#if (defined(MK_DIAG) || defined(MK_PHASE_TIMERS) || defined(MK_BIN_TIMERS)) && !defined(MKAMD_DIAGNOSTICS_BUILD)
#error "MK_DIAG / MK_PHASE_TIMERS / MK_BIN_TIMERS change the product kernels: diagnostics builds must pass -DMKAMD_DIAGNOSTICS_BUILD (tools/build_variant.sh)"
#endif

#ifndef MK_DIAG
#define MK_DIAG 0
#endif
// MK_AB=<bits>: same-box A/B of two CORRECT code paths while an experiment is open (values stay right; the bits and what
// they select are listed where they are used).  0 in a release build; a diagnostics build like the others.
#if defined(MK_AB) && !defined(MKAMD_DIAGNOSTICS_BUILD)
#error "MK_AB selects experimental code paths: diagnostics builds only (-DMKAMD_DIAGNOSTICS_BUILD, tools/build_variant.sh)"
#endif
#ifndef MK_AB
#define MK_AB 0
#endif

#ifdef MK_PHASE_TIMERS
namespace mkamd { __device__ unsigned long long g_phase_cycles[8]; }
#define MK_PHASE_FIELDS mutable unsigned long long wait_ = 0, proc_ = 0;
#define MK_PHASE_NOW(var) const unsigned long long var = __builtin_readcyclecounter()
#define MK_PHASE_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MK_PHASE_CAND(cr, ta, tb, tc) do { (cr).wait_ += (tb) - (ta); (cr).proc_ += (tc) - (tb); } while (0)
#define MK_PHASE_FLUSH(runs) do { if (!DENSE && threadIdx.x == 0) { atomicAdd(&g_phase_cycles[6], (runs).wait_); atomicAdd(&g_phase_cycles[7], (runs).proc_); } } while (0)
#else
#define MK_PHASE_FIELDS
#define MK_PHASE_NOW(var) do {} while (0)
#define MK_PHASE_DRAIN() do {} while (0)
#define MK_PHASE_CAND(cr, ta, tb, tc) do {} while (0)
#define MK_PHASE_FLUSH(runs) do {} while (0)
#endif

#if defined(MK_PHASE_TIMERS) && !defined(MK_BIN_TIMERS)
#define MK_PHASE_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if (!DENSE && threadIdx.x == 0) atomicAdd(&g_phase_cycles[i], now_ - phase_t_); phase_t_ = now_; } while (0)
#define MK_PHASE_BEGIN() unsigned long long phase_t_ = __builtin_readcyclecounter()
#else
#define MK_PHASE_MARK(i) do {} while (0)
#define MK_PHASE_BEGIN() do {} while (0)
#endif

#if defined(MK_PHASE_TIMERS) && defined(MK_BIN_TIMERS)      // the sections of k_bin_direct (wave 0 of every 64th block: same-address atomics serialise)
#define MK_BIN_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) atomicAdd(&g_phase_cycles[i], now_ - bin_t_); bin_t_ = now_; } while (0)
#define MK_BIN_BEGIN() unsigned long long bin_t_ = __builtin_readcyclecounter()
#else
#define MK_BIN_MARK(i) do {} while (0)
#define MK_BIN_BEGIN() do {} while (0)
#endif

#ifdef MKAMD_DIAGNOSTICS_BUILD
#define MKAMD_BUILD_KIND " DIAGNOSTICS"
#else
#define MKAMD_BUILD_KIND ""
#endif
