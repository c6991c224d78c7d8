// mk_device.h -- gfx950 (CDNA4) device primitives used by kernels.h.
//
// Thin named wrappers over the wave64 / LDS / VALU builtins so that the kernels read in the
// domain's terms.  wave = 64 lanes everywhere (MI355X); nothing here is portable on purpose.
//
// (tests/emu/ holds a host-thread SIMT emulation that shadows this header so the SAME kernel
//  source can be exercised on a CPU-only box by the test-suite; it is test infrastructure and
//  is never built into libmkamd.so.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MK_DEV __device__ __forceinline__
#define MK_KERNEL(bounds) __global__ __launch_bounds__(bounds)
#define MK_KERNEL_OCC(bounds, waves) __global__ __launch_bounds__(bounds, waves)   // at least `waves` waves per SIMD: a register budget of 512 / waves
#define MK_DEVFN __device__                       // a member function of a device-side struct
#define MK_DEV_CONST __device__ const             // a table in device memory

typedef float v2f __attribute__((ext_vector_type(2)));   // -> v_pk_{add,mul,fma}_f32

namespace mkamd {

constexpr int WAVE = 64;

MK_DEV float mk_inf() { return __builtin_inff(); }

// 64-bit lane mask of `pred` over the wave.
MK_DEV unsigned long long mk_ballot(bool pred) { return __ballot(pred); }

// number of set bits of `mask` below this lane (v_mbcnt_lo/hi).
MK_DEV int mk_rank_in_mask(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

MK_DEV int mk_popc64(unsigned long long m) { return __popcll(m); }
MK_DEV int mk_clz64(unsigned long long m) { return __builtin_clzll(m); }      // m != 0
MK_DEV int mk_ctz64(unsigned long long m) { return __builtin_ctzll(m); }      // m != 0

MK_DEV float mk_rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32 (1 ulp); rcp(inf)=0, rcp(0)=inf

MK_DEV float mk_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32

MK_DEV float mk_min(float a, float b) { return __builtin_fminf(a, b); }  // v_min_f32: NaN-ignoring
MK_DEV float mk_abs(float a) { return __builtin_fabsf(a); }             // |x| source modifier
// v_min_f32 as the hardware does it, without the canonicalising v_max_f32 x, x, x the compiler puts in front of fminf when it
// cannot see where an operand comes from (a running minimum carried through a loop): the operands here are never signalling NaNs
MK_DEV float mk_min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// ... and v_min3_f32 / v_max3_f32 the same way (round 6: the running minima / maxima of the group-reduction kernel's inner loop)
MK_DEV float mk_min3_raw(float m, float a, float b) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }
MK_DEV float mk_max3_raw(float m, float a, float b) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }
// max3 of the ABSOLUTE values (the |x| source modifiers of a VOP3 instruction: no separate v_and)
MK_DEV float mk_max3_abs_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// keeps what follows inside its (wave-uniform) branch: a volatile asm is never speculated, so the branch is not if-converted
MK_DEV void mk_stay_in_branch() { asm volatile(""); }
// nothing is scheduled across this point (a bound on how many independent chains the scheduler interleaves -- and keeps in registers)
MK_DEV void mk_sched_barrier() { __builtin_amdgcn_sched_barrier(0); }
// the value stays in its register from here on: the compiler forgets that it is a constant it could build again (it
// re-materialised the eight +inf of an accumulator set in front of every loop that uses them)
MK_DEV void mk_keep(float& x) { asm("" : "+v"(x)); }
MK_DEV float mk_max(float a, float b) { return __builtin_fmaxf(a, b); }  // v_max_f32: NaN-ignoring
MK_DEV float mk_min3(float m, float a, float b) { return __builtin_fminf(__builtin_fminf(a, b), m); }   // v_min3_f32
// min of a running bit pattern with a non-negative float (v_min_u32; see k_voxelize_tiles)
MK_DEV unsigned mk_min_bits(unsigned q, float t)
{
    const unsigned b = __float_as_uint(t);
    return b < q ? b : q;
}
MK_DEV float mk_uint_as_float(unsigned u) { return __uint_as_float(u); }
MK_DEV unsigned mk_float_bits(float f) { return __float_as_uint(f); }
// v is the same in every lane (e.g. loaded from a wave-uniform LDS address): move it to a scalar register so
// that loop counters / branches on it are scalar
MK_DEV unsigned mk_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// two floats processed by one packed instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: IEEE results,
// identical to the scalar ops)
typedef float mk_f2 __attribute__((ext_vector_type(2)));
MK_DEV mk_f2 mk_f2_splat(float v) { return mk_f2{v, v}; }
// the two components sit in ONE aligned register pair from here on (the register allocator otherwise splits a pair whose halves it
// can track separately and re-pairs the halves as it likes -- e.g. a value in use with one that is still being loaded)
MK_DEV void mk_keep(mk_f2& x) { asm("" : "+v"(x)); }
MK_DEV mk_f2 mk_f2_fma(mk_f2 a, mk_f2 b, mk_f2 c) { return __builtin_elementwise_fma(a, b, c); }
// packed arithmetic with the reference's roundings: a multiply and an add / subtract per component, each rounded on its own --
// never contracted into an fma (v_pk_mul_f32 / v_pk_add_f32: two float32 operations per lane and instruction)
MK_DEV mk_f2 mk_f2_mul_rn(mk_f2 a, mk_f2 b)
{
#pragma clang fp contract(off)
    return a * b;
}
MK_DEV mk_f2 mk_f2_add_rn(mk_f2 a, mk_f2 b)
{
#pragma clang fp contract(off)
    return a + b;
}
MK_DEV mk_f2 mk_f2_sub_rn(mk_f2 a, mk_f2 b)
{
#pragma clang fp contract(off)
    return a - b;
}
MK_DEV mk_f2 mk_f2_load(const float* p8) { return *reinterpret_cast<const mk_f2*>(p8); }   // 8-byte aligned
// running bit-pattern minimum with TWO non-negative floats (v_min3_u32)
MK_DEV unsigned mk_min3_bits(unsigned m, float a, float b)
{
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const unsigned t = ua < ub ? ua : ub;
    return t < m ? t : m;
}

// device-scope fence: this thread's earlier writes are visible to every CU before its later ones (and vice versa for reads)
MK_DEV void mk_threadfence() { __threadfence(); }
MK_DEV void mk_sched_fence() { __builtin_amdgcn_sched_barrier(0); }     // the instruction scheduler moves nothing across this point
MK_DEV void mk_threadfence_system() { __threadfence_system(); }   // release / acquire at system scope (host-visible memory)
MK_DEV void mk_sleep() { __builtin_amdgcn_s_sleep(8); }

// the lanes of ONE wave have all passed this point and see each other's LDS writes (a wave's LDS operations complete in
// order: what is needed is that the compiler keeps them in order) -- for waves of a workgroup that run independently
MK_DEV void mk_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier (for the 64-thread tile kernel this is a single-wave s_barrier).
MK_DEV void mk_block_sync() { __syncthreads(); }

MK_DEV unsigned mk_atomic_add(unsigned* p, unsigned v) { return atomicAdd(p, v); }
MK_DEV unsigned long long mk_atomic_add64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
MK_DEV unsigned mk_atomic_sub(unsigned* p, unsigned v) { return atomicSub(p, v); }
MK_DEV void mk_atomic_or(int* p, int v) { atomicOr(p, v); }

MK_DEV unsigned mk_atomic_cas(unsigned* p, unsigned expect, unsigned val) { return atomicCAS(p, expect, val); }
MK_DEV void mk_atomic_min(unsigned* p, unsigned v) { atomicMin(p, v); }
MK_DEV void mk_atomic_max(unsigned* p, unsigned v) { atomicMax(p, v); }
// device-scope relaxed load (bypasses this CU's L1: sees other CUs' atomics)
MK_DEV unsigned mk_load_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
MK_DEV unsigned mk_lds_cas(unsigned* p, unsigned expect, unsigned val) { return atomicCAS(p, expect, val); }  // ds_cmpst_rtn
MK_DEV float mk_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }                        // v_fma_f32, never split
// LDS atomic add, returns the old value (ds_add_rtn_u32)
MK_DEV unsigned mk_lds_add(unsigned* p, unsigned v) { return atomicAdd(p, v); }
MK_DEV void mk_lds_min(unsigned* p, unsigned v) { (void)atomicMin(p, v); }                                   // ds_min_u32
// instruction-issue priority of this wave on its SIMD (s_setprio 0..3): the small latency-bound pre-pass
// kernels raise it so that they are not starved of issue slots by the VALU-saturating tile kernel of the
// previous call when the two overlap
MK_DEV void mk_wave_priority_high() { __builtin_amdgcn_s_setprio(3); }
// A 16-byte piece of the result: written once, never read back by the kernels.  STREAM = a non-temporal store (`nt`): the
// 2 GB a 256-grid step writes then do not push the candidate records (read ~27 times each) out of the L2 -- the VALU-bound
// wave-per-tile kernel gains 1.5-2 % on cfg2 (2.204 -> 2.160 ms), 8 % on the 3PTB batch, 6 % on cfg4 (in a team of waves per
// tile, one grid per call, it changes nothing: plain stores there).  NOT for the store-bound workgroup-per-item kernel: its
// 16-byte pieces are merged into full lines by the L2, and non-temporal they reach the HBM one by one (3.28 -> 5.69 ms).
// The temp records the binning writes once and the fill pass reads once: non-temporal both ways (coalesced, full lines) --
// the in-order pre-pass gains 1 % (step 2.50 -> 2.47 ms), the overlapped one nothing.
typedef float mk_v4f_ __attribute__((ext_vector_type(4)));
typedef unsigned mk_v2u_ __attribute__((ext_vector_type(2)));
// Sixteen bytes to an address that is only known to be 4-byte aligned (a result row whose pitch is an odd number of floats, an
// offset view): ONE global_store_dwordx4 -- the compiler emits it for a 4-byte-aligned 16-byte copy on amdhsa targets, where the
// kernel driver runs every queue in the unaligned access mode (SH_MEM_CONFIG.ALIGNMENT_MODE) -- instead of four dword stores.
MK_DEV void mk_store_f4_dword_aligned(float* p, float4 v) { __builtin_memcpy(p, &v, 16); }
MK_DEV void mk_tmp_store(float4* p, float4 v) { __builtin_nontemporal_store(mk_v4f_{v.x, v.y, v.z, v.w}, reinterpret_cast<mk_v4f_*>(p)); }
MK_DEV void mk_tmp_store(uint2* p, uint2 v) { __builtin_nontemporal_store(mk_v2u_{v.x, v.y}, reinterpret_cast<mk_v2u_*>(p)); }
MK_DEV float4 mk_tmp_load(const float4* p) { const mk_v4f_ v = __builtin_nontemporal_load(reinterpret_cast<const mk_v4f_*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
MK_DEV uint2 mk_tmp_load(const uint2* p) { const mk_v2u_ v = __builtin_nontemporal_load(reinterpret_cast<const mk_v2u_*>(p)); return make_uint2(v.x, v.y); }
MK_DEV void mk_tmp_store(unsigned* p, unsigned v) { __builtin_nontemporal_store(v, p); }
MK_DEV unsigned mk_tmp_load(const unsigned* p) { return __builtin_nontemporal_load(p); }
template <bool STREAM>
MK_DEV void mk_store_result(float4* p, float4 v)
{
    if constexpr (STREAM) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
    } else {
        *p = v;
    }
}
// value of `v` in lane `lane` (wave-uniform index) -> SGPR (v_readlane_b32)
MK_DEV void mk_setprio_high() { __builtin_amdgcn_s_setprio(3); }   // this wave first at the issue arbiter of its SIMD
MK_DEV unsigned mk_readlane(unsigned v, int lane) { return (unsigned)__builtin_amdgcn_readlane((int)v, lane); }

MK_DEV unsigned mk_shfl_up(unsigned v, int delta) { return __shfl_up(v, delta, WAVE); }
MK_DEV unsigned mk_shfl_down(unsigned v, int delta) { return __shfl_down(v, delta, WAVE); }
// value of v in lane `src` (per-lane source: ds_bpermute_b32)
MK_DEV unsigned mk_shfl(unsigned v, int src) { return (unsigned)__shfl((int)v, src, WAVE); }

// separately rounded double multiply / add (never contracted into an FMA)
MK_DEV double mk_dmul_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
MK_DEV double mk_dadd_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}

// float32 ops with exactly one rounding each and no FMA contraction (distance_utils parity is bit-exact)
MK_DEV float mk_fadd_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
MK_DEV float mk_fsub_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
MK_DEV float mk_fmul_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
MK_DEV float mk_fdiv_rn(float a, float b) { return __fdiv_rn(a, b); }     // IEEE correctly rounded
// *(float*)((char*)base + byte_offset) for a WAVE-UNIFORM global base: a raw buffer load whose descriptor holds the base
// in scalar registers (buffer_load_dword v, v_offset, s[rsrc], 0 offen) -- the address costs NO vector instruction.
// (Left alone, the optimizer folds the lane's offset into the 64-bit address arithmetic and redoes it per lane: five
//  vector instructions a load; a global_load with the base pinned to scalar registers still pays one.)
MK_DEV float mk_load_f32_uniform_base(const float* base, unsigned byte_offset)
{
    const unsigned long long b = (unsigned long long)reinterpret_cast<uintptr_t>(base);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    void* p = reinterpret_cast<void*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    // gfx9 raw buffer: stride 0, num_records = 2^32 - 1 bytes (no range to enforce: the callers bound the offset), DATA_FORMAT 32
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)0xffffffffu, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_offset, 0, 0));
}
// The same with the wave-uniform part of the address as a 32-bit SCALAR byte offset from one base: the descriptor is built once
// (the base never changes), a row costs one s_mul / s_add instead of a 64-bit multiply-add and a fresh descriptor (~11 scalar
// instructions per load: k_dist_pairs issued 600 of them per wave, as many as its vector instructions, on the ONE scalar unit
// the four SIMDs of a CU share).  The caller guarantees uniform_byte_offset + byte_offset < 2^32.
MK_DEV float mk_load_f32_base_soffset(const float* base, unsigned uniform_byte_offset, unsigned byte_offset)
{
    const unsigned long long b = (unsigned long long)reinterpret_cast<uintptr_t>(base);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    void* p = reinterpret_cast<void*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)0xffffffffu, 0x00020000);
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)uniform_byte_offset);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_offset, (int)so, 0));
}
MK_DEV float mk_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // v_max3_f32 (NaN operands are ignored)
MK_DEV float mk_rint(float a) { return __builtin_rintf(a); }              // round half to even: v_rndne_f32
// IEEE correctly rounded sqrt.  (hipcc lowers __fsqrt_rn / sqrtf in this build to a bare v_sqrt_f32, which
// is only accurate to 1 ulp -- measured: 15 % of results off in the last bit.)  v_sqrt_f32 is within 1 ulp, so
// the correctly rounded value is s or one of its neighbours; the sign of the exactly computed (FMA)
// residuals x - s*s_down and x - s*s_up tells on which side of the two rounding midpoints x lies.
// The two halves of mk_fsqrt_rn for callers that take MANY roots: `mk_sqrt_ordinary(x)` per value (or-ed over a batch, ONE ballot and
// ONE wave-uniform branch for the batch), then `mk_fsqrt_rn_ordinary` on every value -- straight-line code the scheduler can
// interleave across the batch (with a branch per root every pair of the distance kernels was a basic block of its own: a
// serial chain of ~30 dependent instructions, five s_nop and three branches per distance).
MK_DEV bool mk_sqrt_ordinary(float x) { return (__float_as_uint(x) - 0x0F800000u) < (0x7F800000u - 0x0F800000u); }   // in [2^-96, inf)
// ... and of a whole batch at once: bit patterns of non-negative floats order like the values, everything else (a sign bit, NaN) is
// a LARGER unsigned number than +inf -- so "all in [2^-96, inf)" is the smallest pattern >= 2^-96's and the largest < inf's: a
// v_min3_u32 / v_max3_u32 tree and two compares (round 5: and-ing four single tests cost the compiler ~20 instructions per batch of
// four -- compare, 0/1, shift, or --, a fifth of a non-periodic pair's instructions in the block-per-frame kernel).
template <int N>
MK_DEV bool mk_sqrt_ordinary_all(const float (&x)[N])
{
    unsigned lo = __float_as_uint(x[0]), hi = lo;
#pragma unroll
    for (int i = 1; i < N; ++i) {
        const unsigned b = __float_as_uint(x[i]);
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
    }
    return lo >= 0x0F800000u && hi < 0x7F800000u;
}
// (the provable form: v_sqrt_f32 + the Tuckerman correction, 12 issue slots; what the fast form below is verified against)
MK_DEV float mk_fsqrt_rn_tuckerman(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_down = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_down = __builtin_fmaf(-s_down, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    const float y = (r_down <= 0.0f) ? s_down : s;
    return (r_up > 0.0f) ? s_up : y;
}
// Round 5: the distance kernels are bound by VALU issue and a third of their instructions per pair was this root.  One
// Newton-style correction of x * rsq(x) with the EXACT residual -- y0 = v_rsq_f32(x), s = x y0, r = fma(-s, s, x),
// root = fma(r, y0 / 2, s) -- costs 8 issue slots instead of 12, and on gfx950 it returns the correctly rounded root for EVERY
// float in [2^-96, inf): compared with the form above over all 1 879 048 192 of them (tools/sqrt_exact.hip,
// profiles/r5_sqrt_exact.txt; the library repeats the comparison on demand: mkamd_selftest_sqrt, run by the GPU test tier).
// It is a property of this chip's v_rsq_f32, not a theorem: the error of the correction term (~2^-46 of the root) is larger
// than the closest a root can come to a rounding boundary (2^-50) -- hence the exhaustive check.
MK_DEV float mk_fsqrt_rn_ordinary(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    const float s = x * y0;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, 0.5f * y0, s);
}
MK_DEV float mk_fsqrt_rn(float x)
{
    // every lane of the wave holds an ordinary number in [2^-96, inf) -- practically always: the short form alone
    if (__builtin_amdgcn_ballot_w64(!((__float_as_uint(x) - 0x0F800000u) < (0x7F800000u - 0x0F800000u))) == 0ull)
        return mk_fsqrt_rn_ordinary(x);
    // keep v_sqrt_f32 away from denormal inputs / results: scale tiny x by 2^64 (exact), result by 2^-32
    const bool tiny = x < 0x1.0p-96f;
    const float xs = tiny ? x * 0x1.0p+64f : x;
    float s = __builtin_amdgcn_sqrtf(xs);
    const float s_down = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_down = __builtin_fmaf(-s_down, s, xs);
    const float r_up = __builtin_fmaf(-s_up, s, xs);
    float y = (r_down <= 0.0f) ? s_down : s;
    y = (r_up > 0.0f) ? s_up : y;
    y = tiny ? y * 0x1.0p-32f : y;
    // 0, +inf, NaN and negative inputs: the hardware result is already the IEEE one
    return (xs > 0.0f && xs < __builtin_inff()) ? y : s;
}

MK_DEV float mk_int_as_float(int i) { return __int_as_float(i); }
MK_DEV int mk_float_as_int(float f) { return __float_as_int(f); }

}  // namespace mkamd
