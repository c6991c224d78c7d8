// tests/emu/emu_device.h -- TEST INFRASTRUCTURE ONLY (never part of libmkamd.so).
//
// A tiny host-side SIMT emulation that lets the test-suite run the REAL kernel source
// (moleculekit_amd/csrc/kernels.h + pipeline.h) on a CPU-only box: every GPU thread of a
// workgroup is a fiber (own stack, hand-written register switch) inside one OS thread; wave collectives (ballot, shuffles) and
// workgroup barriers are rendezvous points at which a fiber yields until its 64-lane wave /
// its whole block has arrived.  Workgroups run one after the other.  It exists to catch
// indexing / tiling / binning logic errors before a GPU run -- it says nothing about
// performance and it is not a fallback: the product library has no CPU compute path.
//
// It provides exactly the names the product's mk_device.h provides (and shadows that header via
// MK_DEVICE_API_PROVIDED).
#pragma once
#define MK_DEVICE_API_PROVIDED 1

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define MK_DEV static inline
#define MK_KERNEL(bounds)
#define MK_KERNEL_OCC(bounds, waves)
#define MK_DEVFN
#define MK_DEV_CONST static const
#define __shared__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct float2 { float x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

// "registers" of the fiber that is currently running
struct EmuIdx { unsigned x, y, z; };
static EmuIdx threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 256 * 1024;

// Fiber switch: the callee-saved registers and the stack pointer, nothing else (x86-64 SysV).  ucontext's swapcontext saves
// and restores the signal mask with a system call on every switch -- two thirds of the emulation's run time.
extern "C" void mkamd_emu_switch(void** save_sp, void* load_sp);
#if defined(__x86_64__)
asm(R"(
    .text
    .globl mkamd_emu_switch
    .type mkamd_emu_switch,@function
mkamd_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size mkamd_emu_switch, .-mkamd_emu_switch
)");
#else
#error "the test emulator's fiber switch is written for x86-64"
#endif

struct BlockState {
    int nthreads = 0;
    int cur = 0;                       // running lane
    void* sched = nullptr;             // saved stack pointers
    void* ctx[MAX_THREADS];
    bool done[MAX_THREADS];
    char* stacks = nullptr;
    // rendezvous state: index 0..15 = waves, 16 = whole block
    unsigned gen[17];
    int arrived[17];
    unsigned long long ballot_acc[16];
    unsigned long long ballot_res[16];
    unsigned xchg[MAX_THREADS];
    void (*body)(void*) = nullptr;
    void* body_arg = nullptr;
};
static BlockState g_blk;

static inline void yield_to_scheduler()
{
    const int me = g_blk.cur;
    mkamd_emu_switch(&g_blk.ctx[me], g_blk.sched);
    threadIdx.x = (unsigned)me;        // restored by the scheduler too; belt and braces
}

// wait until `count` fibers of group `grp` have arrived
static inline void rendezvous(int grp, int count)
{
    const unsigned my_gen = g_blk.gen[grp];
    if (++g_blk.arrived[grp] == count) {
        g_blk.arrived[grp] = 0;
        ++g_blk.gen[grp];
        return;
    }
    while (g_blk.gen[grp] == my_gen) yield_to_scheduler();
}

static void fiber_entry()
{
    g_blk.body(g_blk.body_arg);
    g_blk.done[g_blk.cur] = true;
    mkamd_emu_switch(&g_blk.ctx[g_blk.cur], g_blk.sched);      // never resumed
    abort();
}

template <class F>
static void run_block(int nthreads, F& f)
{
    if (nthreads > MAX_THREADS || (nthreads % 64) != 0) { fprintf(stderr, "emu: bad block size %d\n", nthreads); abort(); }
    if (!g_blk.stacks) g_blk.stacks = (char*)malloc(STACK_BYTES * MAX_THREADS);
    g_blk.nthreads = nthreads;
    memset(g_blk.gen, 0, sizeof g_blk.gen);
    memset(g_blk.arrived, 0, sizeof g_blk.arrived);
    memset(g_blk.ballot_acc, 0, sizeof g_blk.ballot_acc);
    g_blk.body = [](void* p) { (*(F*)p)(); };
    g_blk.body_arg = &f;
    for (int t = 0; t < nthreads; ++t) {
        g_blk.done[t] = false;
        // a fresh fiber: what mkamd_emu_switch pops (six registers, then `ret` into fiber_entry with the stack as after a call)
        uintptr_t top = ((uintptr_t)(g_blk.stacks + (size_t)(t + 1) * STACK_BYTES)) & ~(uintptr_t)15;
        uint64_t* sp = (uint64_t*)top;
        *--sp = 0;                                   // (return address of fiber_entry: it never returns)
        *--sp = (uint64_t)(uintptr_t)&fiber_entry;
        for (int r = 0; r < 6; ++r) *--sp = 0;
        g_blk.ctx[t] = sp;
    }
    int remaining = nthreads;
    long spins = 0;
    while (remaining > 0) {
        int progressed = 0;
        for (int t = 0; t < nthreads; ++t) {
            if (g_blk.done[t]) continue;
            g_blk.cur = t;
            threadIdx.x = (unsigned)t; threadIdx.y = threadIdx.z = 0;
            mkamd_emu_switch(&g_blk.sched, g_blk.ctx[t]);
            if (g_blk.done[t]) { --remaining; ++progressed; }
        }
        // a block whose live fibers all wait on lanes that already exited would spin forever
        if (!progressed && ++spins > 100000000L) { fprintf(stderr, "emu: deadlock (divergent collective?)\n"); abort(); }
    }
}

template <class K, class... A>
static void launch(K kernel, dim3 grid, dim3 block, A... args)
{
    gridDim = EmuIdx{grid.x, grid.y, grid.z};
    blockDim = EmuIdx{block.x, block.y, block.z};
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = EmuIdx{bx, by, bz};
                auto body = [&]() { kernel(args...); };
                run_block((int)block.x, body);
            }
}

}  // namespace emu

namespace mkamd {

constexpr int WAVE = 64;

MK_DEV float mk_inf() { return INFINITY; }

MK_DEV unsigned long long mk_ballot(bool pred)
{
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if (pred) emu::g_blk.ballot_acc[wv] |= 1ull << lane;
    // last arriver publishes the result and clears the accumulator
    if (emu::g_blk.arrived[wv] == WAVE - 1) { emu::g_blk.ballot_res[wv] = emu::g_blk.ballot_acc[wv]; emu::g_blk.ballot_acc[wv] = 0; }
    emu::rendezvous(wv, WAVE);
    const unsigned long long r = emu::g_blk.ballot_res[wv];
    emu::rendezvous(wv, WAVE);          // nobody starts the next ballot before everyone has read
    return r;
}
MK_DEV int mk_rank_in_mask(unsigned long long mask)
{
    const int lane = (int)threadIdx.x & 63;
    return __builtin_popcountll(mask & ((1ull << lane) - 1ull));
}
MK_DEV int mk_popc64(unsigned long long m) { return __builtin_popcountll(m); }
MK_DEV int mk_clz64(unsigned long long m) { return __builtin_clzll(m); }
MK_DEV int mk_ctz64(unsigned long long m) { return __builtin_ctzll(m); }
MK_DEV float mk_rcp(float x) { return 1.0f / x; }
MK_DEV float mk_exp2(float x) { return exp2f(x); }
MK_DEV float mk_min(float a, float b) { return fminf(a, b); }
MK_DEV float mk_min_raw(float a, float b) { return fminf(a, b); }
MK_DEV float mk_min3_raw(float m, float a, float b) { return fminf(fminf(a, b), m); }
MK_DEV float mk_max3_raw(float m, float a, float b) { return fmaxf(fmaxf(a, b), m); }
MK_DEV float mk_max3_abs_raw(float a, float b, float c) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)); }
MK_DEV void mk_keep(float&) {}
MK_DEV void mk_stay_in_branch() {}
MK_DEV void mk_sched_barrier() {}
MK_DEV unsigned mk_float_bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
MK_DEV float mk_abs(float a) { return fabsf(a); }
MK_DEV float mk_max(float a, float b) { return fmaxf(a, b); }
MK_DEV float mk_min3(float m, float a, float b) { return fminf(fminf(a, b), m); }
MK_DEV void mk_threadfence() {}
MK_DEV void mk_sched_fence() {}
MK_DEV void mk_threadfence_system() {}
MK_DEV void mk_sleep() {}
MK_DEV unsigned mk_uniform(unsigned v) { return v; }
typedef float mk_f2 __attribute__((vector_size(8)));
MK_DEV mk_f2 mk_f2_splat(float v) { return mk_f2{v, v}; }
MK_DEV void mk_keep(mk_f2&) {}
MK_DEV mk_f2 mk_f2_fma(mk_f2 a, mk_f2 b, mk_f2 c) { return mk_f2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }
MK_DEV mk_f2 mk_f2_mul_rn(mk_f2 a, mk_f2 b) { volatile float x = a[0] * b[0], y = a[1] * b[1]; return mk_f2{x, y}; }
MK_DEV mk_f2 mk_f2_add_rn(mk_f2 a, mk_f2 b) { volatile float x = a[0] + b[0], y = a[1] + b[1]; return mk_f2{x, y}; }
MK_DEV mk_f2 mk_f2_sub_rn(mk_f2 a, mk_f2 b) { volatile float x = a[0] - b[0], y = a[1] - b[1]; return mk_f2{x, y}; }
MK_DEV mk_f2 mk_f2_load(const float* p8) { mk_f2 v; memcpy(&v, p8, 8); return v; }
MK_DEV unsigned mk_min3_bits(unsigned m, float a, float b)
{
    const unsigned ua = mk_float_bits(a), ub = mk_float_bits(b);
    const unsigned t = ua < ub ? ua : ub;
    return t < m ? t : m;
}
MK_DEV unsigned mk_min_bits(unsigned q, float t) { const unsigned b = mk_float_bits(t); return b < q ? b : q; }
MK_DEV float mk_uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
MK_DEV void mk_block_sync() { emu::rendezvous(16, emu::g_blk.nthreads); }
MK_DEV void mk_wave_sync() { emu::rendezvous((int)threadIdx.x >> 6, 64); }
MK_DEV unsigned mk_atomic_add(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
MK_DEV unsigned long long mk_atomic_add64(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
MK_DEV unsigned mk_atomic_sub(unsigned* p, unsigned v) { const unsigned o = *p; *p = o - v; return o; }
MK_DEV void mk_atomic_or(int* p, int v) { *p |= v; }
MK_DEV unsigned mk_atomic_cas(unsigned* p, unsigned expect, unsigned val) { const unsigned o = *p; if (o == expect) *p = val; return o; }
MK_DEV void mk_atomic_min(unsigned* p, unsigned v) { if (v < *p) *p = v; }
MK_DEV void mk_atomic_max(unsigned* p, unsigned v) { if (v > *p) *p = v; }
MK_DEV unsigned mk_load_relaxed(unsigned* p) { return *p; }
MK_DEV unsigned mk_lds_cas(unsigned* p, unsigned expect, unsigned val) { const unsigned o = *p; if (o == expect) *p = val; return o; }
MK_DEV float mk_fma(float a, float b, float c) { return fmaf(a, b, c); }
MK_DEV unsigned mk_lds_add(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
MK_DEV void mk_lds_min(unsigned* p, unsigned v) { if (v < *p) *p = v; }
MK_DEV void mk_wave_priority_high() {}
template <bool STREAM> MK_DEV void mk_store_result(float4* p, float4 v) { *p = v; }
MK_DEV void mk_store_f4_dword_aligned(float* p, float4 v) { memcpy(p, &v, 16); }
MK_DEV void mk_tmp_store(float4* p, float4 v) { *p = v; }
MK_DEV void mk_tmp_store(uint2* p, uint2 v) { *p = v; }
MK_DEV float4 mk_tmp_load(const float4* p) { return *p; }
MK_DEV uint2 mk_tmp_load(const uint2* p) { return *p; }
MK_DEV void mk_tmp_store(unsigned* p, unsigned v) { *p = v; }
MK_DEV unsigned mk_tmp_load(const unsigned* p) { return *p; }
MK_DEV void mk_setprio_high() {}
MK_DEV unsigned mk_readlane(unsigned v, int lane)
{
    const int wv = (int)threadIdx.x >> 6;
    emu::g_blk.xchg[threadIdx.x] = v;
    emu::rendezvous(wv, WAVE);
    const unsigned r = emu::g_blk.xchg[(wv << 6) + lane];
    emu::rendezvous(wv, WAVE);
    return r;
}
MK_DEV unsigned mk_shfl_up(unsigned v, int delta)
{
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    emu::g_blk.xchg[threadIdx.x] = v;
    emu::rendezvous(wv, WAVE);
    const unsigned r = lane >= delta ? emu::g_blk.xchg[threadIdx.x - delta] : v;
    emu::rendezvous(wv, WAVE);
    return r;
}
MK_DEV unsigned mk_shfl(unsigned v, int src)
{
    const int wv = (int)threadIdx.x >> 6;
    emu::g_blk.xchg[threadIdx.x] = v;
    emu::rendezvous(wv, WAVE);
    const unsigned r = emu::g_blk.xchg[(wv << 6) + (src & 63)];
    emu::rendezvous(wv, WAVE);
    return r;
}
MK_DEV unsigned mk_shfl_down(unsigned v, int delta)
{
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    emu::g_blk.xchg[threadIdx.x] = v;
    emu::rendezvous(wv, WAVE);
    const unsigned r = lane + delta < WAVE ? emu::g_blk.xchg[threadIdx.x + delta] : v;
    emu::rendezvous(wv, WAVE);
    return r;
}
MK_DEV double mk_dmul_rn(double a, double b) { volatile double r = a * b; return r; }
MK_DEV double mk_dadd_rn(double a, double b) { volatile double r = a + b; return r; }
MK_DEV float mk_fadd_rn(float a, float b) { volatile float r = a + b; return r; }
MK_DEV float mk_fsub_rn(float a, float b) { volatile float r = a - b; return r; }
MK_DEV float mk_fmul_rn(float a, float b) { volatile float r = a * b; return r; }
MK_DEV float mk_fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
MK_DEV float mk_fsqrt_rn(float a) { return sqrtf(a); }
MK_DEV bool mk_sqrt_ordinary(float x) { return x >= 0x1.0p-96f && x < INFINITY; }
template <int N>
MK_DEV bool mk_sqrt_ordinary_all(const float (&x)[N])
{
    bool all = true;
    for (int i = 0; i < N; ++i) all = all && mk_sqrt_ordinary(x[i]);
    return all;
}
MK_DEV float mk_fsqrt_rn_ordinary(float a) { return sqrtf(a); }
MK_DEV float mk_fsqrt_rn_tuckerman(float a) { return sqrtf(a); }
MK_DEV float mk_load_f32_uniform_base(const float* base, unsigned byte_offset)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_offset);
}
MK_DEV float mk_load_f32_base_soffset(const float* base, unsigned uniform_byte_offset, unsigned byte_offset)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)uniform_byte_offset + (size_t)byte_offset);
}
MK_DEV float mk_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
MK_DEV float mk_rint(float a) { return nearbyintf(a); }              // round half to even (default rounding mode)
MK_DEV float mk_int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
MK_DEV int mk_float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }

}  // namespace mkamd
