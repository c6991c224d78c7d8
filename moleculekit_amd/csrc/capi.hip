// capi.hip -- extern "C" boundary of libmkamd.so (see include/mkamd_voxel.h) and the HIP backend
// that drives the launch sequences of pipeline.h on an MI355X.  One context = one device, one
// stream, one grow-only workspace.  Without a GPU every context entry point fails with MKAMD_ENODEV / MKAMD_EHIP; the one
// host implementation (mkamd_calculate_occupancy_cpu, cpu_occupancy.h) is a separate, explicit entry point nothing here falls back to.
#include "../../include/mkamd_voxel.h"
#include "../../include/mkamd_distance.h"
#include "../../include/mkamd_xtc.h"
#include "xtc_reader.h"
#include <atomic>
#include "pipeline.h"
#include "dist_pipeline.h"
#include "host_pack.h"
#include "xtc_gpu.h"
#include "cpu_occupancy.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace mkamd;

// The message of the calling thread's last failure: a fixed buffer, so that reporting an error can never throw
// (an std::string here could raise bad_alloc on the way out of an extern "C" function).
static thread_local char g_last_error[512] = "";

static int fail(int code, const char* msg) noexcept
{
    snprintf(g_last_error, sizeof g_last_error, "%s", msg ? msg : "");
    return code;
}
static int fail(int code, const std::string& msg) noexcept { return fail(code, msg.c_str()); }

static int hip_fail(hipError_t e, const char* what) noexcept
{
    snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, hipGetErrorString(e));
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? MKAMD_ENODEV : MKAMD_EHIP;
}

// Every extern "C" entry point is a function-try-block closed by this: nothing C++ (bad_alloc from a vector / string,
// anything a dependency throws) crosses the C boundary; it comes back as a status + message like every other failure.
#define MK_API_CATCH                                                                              \
    catch (const std::bad_alloc&) { return fail(MKAMD_ENOMEM, "out of host memory"); }            \
    catch (const std::exception& e) { return fail(MKAMD_EHIP, e.what()); }                        \
    catch (...) { return fail(MKAMD_EHIP, "unexpected C++ exception"); }

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

struct mkamd_ctx {
    int device = 0;
    int n_cus = 256;                       // compute units of the device (launch plans that count wave slots: dist_pipeline.h)
    int compute_units() const { return n_cus; }
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;          // stream the launches of the current call go to (main or side)
    hipStream_t main_stream = nullptr;     // the caller-visible stream
    hipStream_t side_stream = nullptr;     // internal: pre-pass of pipelined calls
    hipEvent_t ev_host_done = nullptr;     // end of a small synchronous host call's launches (voxelize_lattice_host_begin_impl)
    hipEvent_t ev_inputs = nullptr;        // main -> side: the call's inputs are ready
    hipEvent_t ev_pre_done[2] = {};        // side -> main: workspace set s is filled
    hipEvent_t ev_tile_done[2] = {};       // main -> side: the tile kernel that read set s has finished
    hipEvent_t inputs_marker = nullptr;    // the event the next pipelined pre-pass waits for (ev_inputs or a timing event)
    hipEvent_t tile_marker[2] = {};        // ditto per workspace set (ev_tile_done[s] or a timing event)
    hipEvent_t last_hot_end = nullptr;
    bool tile_pending[2] = {false, false};
    int next_set = 0;
    bool in_pipelined_prepass = false;
    bool pipelining = false;               // opt-in (mkamd_ctx_set_pipelining)
    bool promise = false;                  // one-shot: the NEXT lattice call's inputs are promised (mkamd_ctx_promise_inputs)
    hipEvent_t promise_event = nullptr;    // ... complete once this event has completed (nullptr: complete already)
    long long n_pipelined = 0;             // lattice calls whose pre-pass went to the side stream (mkamd_ctx_pipelined_calls)
    bool have_pre_tile_event = false;
    void* bufs[2 * WS_NSLOTS] = {};        // two workspace sets (set 1 only used by pipelined calls)
    CounterState counters[2];              // what is known about each set's cell counters between calls
    size_t caps[2 * WS_NSLOTS] = {};
    int tile_k = 0;
    int force_general = 0;
    int lds_tier = -1;                     // -1 = adaptive
    int prepass_mode = -1;                 // -1 = automatic
    int tile_items = -1;                   // -1 = automatic
    int exact_redo = 0;                    // 0 = automatic (mkamd_ctx_set_exact_redo), -1 = never
    int tile_team = -1;                    // -1 = automatic
    int fine_cells = 0;                    // 1 = half-cutoff cells (A-B benchmarking)
    int direct = -1;                       // direct binning: -1 automatic, 0 never, 1 whenever possible (mkamd_ctx_set_direct_binning)
    double value_tol = 0.0;                // tolerance-aware reach (mkamd_ctx_set_value_tolerance); 0 = the hard 5 A cutoff everywhere
    unsigned* fb_host = nullptr;           // pinned, device-visible: tier statistics of the last finished call
    unsigned* fb_dev = nullptr;
    bool err_mirrored = false;             // fb_host[NTIER+1] holds the error flag as of the last lattice call
    void* stage_host = nullptr;            // pinned staging for the inputs of small _host calls (one H2D copy)
    void* stage_host_dev = nullptr;        // device-side address of the same memory (mapped): tiny inputs are read in place
    size_t stage_cap = 0;
    std::vector<uint32_t> contacts_host;   // result of the last mkamd_contacts_trajectory_host call (owned here)
    std::vector<float> packed_coords;      // the selected atoms' rows of a host distance call (host_pack.h), kept between calls
    std::vector<float> f32_stage;          // host staging of big float64-out results
    struct PendingHostCall { bool active = false; size_t out_bytes = 0; bool mapped_out = false; unsigned seq = 0; void* dout = nullptr; hipEvent_t done = nullptr; };
    PendingHostCall pending;               // a host call between its begin and its end (voxelize_lattice_host_begin_impl)
    void* out_host = nullptr;              // pinned, device-mapped result buffer of small _host calls (no D2H copy)
    void* out_host_dev = nullptr;          // its device-side address
    // tile-kernel timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
    hipEvent_t cur0 = nullptr, cur1 = nullptr;
    int launch_status = 0;

    // ---- backend concept (pipeline.h) ----
    CounterState& counter_state(int set) { return counters[set & 1]; }
    const volatile unsigned* feedback_host() const { return fb_host; }
    unsigned* feedback_dev() const { return fb_dev; }
    void note_error_flag_mirrored(bool yes) { err_mirrored = yes; }
    void note_tail_reports(bool yes) { tail_reports = yes; }
    int dist_avoid = 0;                    // kernels dist_trajectory must not take (mkamd_ctx_set_dist_kernels: tests, A-B timing)
    int reduction_block = 0;               // k_dist_reduction_closest's first-group atoms in registers (mkamd_ctx_set_reduction_block)
    char last_dist_kernel[96] = "";        // what the last dist_trajectory call launched (mkamd_ctx_last_dist_kernel)
    void note_dist_kernel(const char* name) { snprintf(last_dist_kernel, sizeof last_dist_kernel, "%s", name); }
    void note_dist_kernel_append(const char* more) { strncat(last_dist_kernel, more, sizeof last_dist_kernel - strlen(last_dist_kernel) - 1); }
    bool tail_reports = false;             // the last lattice call's k_tail writes FB_TILES_DONE / FB_TAIL_WROTE with seq_next
    unsigned seq_next = 0, seq_counter = 0; // sequence number handed to the next lattice call (0 = none)
    int ensure(int slot, size_t bytes, void** ptr, int set = 0)
    {
        slot += set * WS_NSLOTS;
        if (bytes == 0) bytes = 16;
        if (caps[slot] < bytes) {
            if (bufs[slot]) {
                HIP_TRY(hipStreamSynchronize(main_stream));  // buffer may still be in use
                if (side_stream) HIP_TRY(hipStreamSynchronize(side_stream));
                HIP_TRY(hipFree(bufs[slot]));
                bufs[slot] = nullptr; caps[slot] = 0;
            }
            const size_t want = bytes + bytes / 4 + 256;     // grow-only, 25 % slack
            hipError_t e = hipMalloc(&bufs[slot], want);
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(workspace)");
            caps[slot] = want;
        }
        *ptr = bufs[slot];
        return 0;
    }
    // a workspace buffer that grows WITH its first keep_bytes (the device-resident contact list: chunks are appended)
    int grow_keep(int slot, size_t bytes, size_t keep_bytes, void** ptr)
    {
        if (bytes == 0) bytes = 16;
        if (caps[slot] < bytes) {
            void* fresh = nullptr;
            const size_t want = bytes + bytes / 2 + 256;
            hipError_t e = hipMalloc(&fresh, want);
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(contact list)");
            if (bufs[slot] && keep_bytes) {
                e = hipMemcpyAsync(fresh, bufs[slot], keep_bytes, hipMemcpyDeviceToDevice, stream);
                if (e != hipSuccess) { (void)hipFree(fresh); return hip_fail(e, "hipMemcpyAsync(contact list)"); }
            }
            if (bufs[slot]) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipFree(bufs[slot]));
            }
            bufs[slot] = fresh; caps[slot] = want;
        }
        *ptr = bufs[slot];
        return 0;
    }
    int fill(void* p, int byte, size_t bytes)
    {
        HIP_TRY(hipMemsetAsync(p, byte, bytes, stream));
        return 0;
    }
    int to_host(void* dst, const void* src_dev, size_t bytes)
    {
        HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return 0;
    }
    int to_device(void* dst_dev, const void* src, size_t bytes)
    {
        HIP_TRY(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, stream));
        return 0;
    }
    template <class... KA, class... A>
    int launch(void (*kernel)(KA...), dim3 grid, dim3 block, A... args)
    {
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // ---- cross-call software pipeline (pipeline.h: acquire_set / prepass_done / tile_done) ----
    // Every event record / stream wait below is checked: a failed one would silently remove the ordering the results
    // depend on, so on failure the context drains both streams (which restores a trivially correct order), turns
    // pipelining off for good (`pipeline_broken`) and the call carries on in order on the caller's stream.
    bool pipeline_broken = false;
    bool sync_ok(hipError_t e)
    {
        if (e == hipSuccess) return true;
        (void)hipGetLastError();
        pipeline_broken = true;
        if (side_stream) (void)hipStreamSynchronize(side_stream);
        (void)hipStreamSynchronize(main_stream);
        tile_pending[0] = tile_pending[1] = false;
        have_pre_tile_event = false;
        return false;
    }
    int acquire_set(bool big_enough)
    {
        in_pipelined_prepass = false;
        // a promised call (mkamd_ctx_promise_inputs) waits for the event its inputs come with, on whichever stream runs its
        // pre-pass; should that wait fail, the streams have been drained and the event is awaited on the host
        auto await_promise = [&](hipStream_t s) {
            if (!promise_event) return;
            if (!sync_ok(hipStreamWaitEvent(s, promise_event, 0))) (void)hipEventSynchronize(promise_event);
        };
        if (!big_enough || !(pipelining || promise) || !side_stream || pipeline_broken) {
            // in-order call on the caller's stream, set 0 (any earlier pipelined call has already made the
            // main stream wait for its pre-pass; its tile kernel is ahead of us on the same stream)
            await_promise(main_stream);
            have_pre_tile_event = false;
            return 0;
        }
        const int set = next_set;
        // Inputs must not depend on work enqueued after the PREVIOUS call's tile kernel (the opt-in contract of
        // mkamd_ctx_set_pipelining): the side stream is ordered after everything before that launch only.
        if (!have_pre_tile_event) {
            if (!sync_ok(hipEventRecord(ev_inputs, main_stream))) { await_promise(main_stream); return 0; }
            inputs_marker = ev_inputs;
        }
        if (!sync_ok(hipStreamWaitEvent(side_stream, inputs_marker, 0))) { await_promise(main_stream); return 0; }
        if (tile_pending[set] && !sync_ok(hipStreamWaitEvent(side_stream, tile_marker[set], 0))) { await_promise(main_stream); return 0; }
        if (promise_event && !sync_ok(hipStreamWaitEvent(side_stream, promise_event, 0))) { (void)hipEventSynchronize(promise_event); return 0; }
        next_set ^= 1;
        stream = side_stream;
        in_pipelined_prepass = true;
        ++n_pipelined;
        return set;
    }
    bool set_is_pipelined(int) const { return in_pipelined_prepass; }
    bool pipelining_possible() const { return (pipelining || promise) && side_stream && !pipeline_broken; }
    void prepass_done(int set)
    {
        if (!in_pipelined_prepass) return;
        stream = main_stream;
        // on failure sync_ok() has drained the side stream: the pre-pass is complete, the tile kernel may follow
        if (sync_ok(hipEventRecord(ev_pre_done[set], side_stream)) && sync_ok(hipStreamWaitEvent(main_stream, ev_pre_done[set], 0)))
            have_pre_tile_event = true;                               // hot_begin() records the marker
    }
    void tile_done(int set)
    {
        if (!side_stream) return;
        // Only a PIPELINED call needs the marker (the pre-pass that reuses its workspace set two calls later waits for
        // it).  An in-order call is covered by stream order: the next pipelined pre-pass first waits for a marker recorded
        // on the main stream after everything enqueued so far (acquire_set, !have_pre_tile_event).  The event is a barrier
        // packet of its own -- ~2 us of every one-molecule call on a context that has ever had pipelining switched on.
        const bool piped = in_pipelined_prepass;
        in_pipelined_prepass = false;
        if (pipeline_broken || !piped) { last_hot_end = nullptr; return; }
        if (last_hot_end) { tile_marker[set] = last_hot_end; tile_pending[set] = true; }   // the timing event sits at the same place
        else if (sync_ok(hipEventRecord(ev_tile_done[set], main_stream))) { tile_marker[set] = ev_tile_done[set]; tile_pending[set] = true; }
        last_hot_end = nullptr;
    }
    // Events are barrier packets the command processor takes ~5 us each to retire, and they sit between one
    // call's tile kernel and the next one's: the timing events double as the pipeline's markers when both exist.
    int last_flavour = -1, last_K = 0, last_ecap = 0;      // the tile kernel of the last lattice call (mkamd_ctx_last_tile_kernel)
    void hot_begin(int flavour, int K, int ecap)
    {
        last_flavour = flavour; last_K = K; last_ecap = ecap;
        last_hot_end = nullptr;
        if (timing) {
            std::pair<hipEvent_t, hipEvent_t> ev;
            bool ok = true;
            if (!ev_free.empty()) { ev = ev_free.back(); ev_free.pop_back(); }
            else ok = hipEventCreate(&ev.first) == hipSuccess && hipEventCreate(&ev.second) == hipSuccess;
            if (ok && hipEventRecord(ev.first, stream) == hipSuccess) {
                cur0 = ev.first; cur1 = ev.second;
                if (in_pipelined_prepass) inputs_marker = cur0;       // "everything before this call's tile kernel"
                return;
            }
            (void)hipGetLastError();                                  // no timing for this launch; the marker below still goes in
        }
        if (in_pipelined_prepass) {
            if (sync_ok(hipEventRecord(ev_inputs, main_stream))) inputs_marker = ev_inputs;
        }
    }
    void hot_end()
    {
        if (!timing || !cur0) return;
        if (hipEventRecord(cur1, stream) == hipSuccess) {
            ev_used.emplace_back(cur0, cur1);
            last_hot_end = cur1;
        } else {
            (void)hipGetLastError();
            ev_free.emplace_back(cur0, cur1);
        }
        cur0 = cur1 = nullptr;
    }
};

// A molecule's topology on the device (include/mkamd_voxel.h, mkamd_topology_create_*): one allocation, owned here.
struct mkamd_topology {
    int device = 0;
    void* mem = nullptr;                   // sigmas copy | cw | ids | table | flags
    mkamd::TopologyDev dev;
};

// `pending_ok`: the entry point may run between the two halves of a host call (mkamd_voxelize_lattice_host_begin / _end): the
// halves themselves, and what touches neither the pending call's result buffer nor the workspace it reads (queries, the
// centre generator, copies out).  Every other entry point would regrow or overwrite what `end` is about to hand back --
// it is refused, not guessed at.
static int check_ctx(mkamd_ctx* ctx, bool pending_ok = false)
{
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (!pending_ok && ctx->pending.active)
        return fail(MKAMD_EINVAL, "a host call begun on this context (mkamd_voxelize_lattice_host_begin) has not been ended: call _end (or mkamd_ctx_abandon_pending) first");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->stream = ctx->main_stream;           // a failed call may have left the side stream selected
    return 0;
}

// read + clear the device-side error flag (after the stream has been synchronised)
static int collect_async_errors(mkamd_ctx* ctx)
{
    if (!ctx->bufs[WS_ERR]) return 0;
    // the last lattice call's dense kernel mirrored the flag into pinned host memory: no copy on the clean path
    if (ctx->err_mirrored && ctx->fb_host && ((volatile unsigned*)ctx->fb_host)[NTIER + 1] == 0u) return 0;
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, ctx->bufs[WS_ERR], sizeof(int), hipMemcpyDeviceToHost));
    if (flag == 0) return 0;
    HIP_TRY(hipMemset(ctx->bufs[WS_ERR], 0, sizeof(int)));
    if (flag & MK_ERR_TOPOLOGY) return fail(MKAMD_EINVAL, "a topology call was given an item that is not the topology's atom count long; results are incomplete");
    if (flag & MK_ERR_BAD_BOX) return fail(MKAMD_EBOX, "periodic box edges must be > 10 A (2 x cutoff)");
    if (flag & MK_ERR_TOO_MANY_IMAGES) return fail(MKAMD_EBOX, "periodic box much smaller than the grid (too many images)");
    return fail(MKAMD_EOVERFLOW, "more periodic images than max_images_per_atom allowed; results are incomplete");
}

static int ensure_err_flag(mkamd_ctx* ctx)
{
    if (ctx->bufs[WS_ERR]) return 0;
    void* p = nullptr;
    int st = ctx->ensure(WS_ERR, sizeof(int), &p);
    if (st) return st;
    HIP_TRY(hipMemsetAsync(p, 0, sizeof(int), ctx->stream));
    return 0;
}

// One wave that spins for `ref_ticks` of the fixed-frequency reference counter (s_memrealtime) and reports how far the SHADER clock
// counter (s_memtime) got meanwhile: launched on a stream of the caller's beside the work whose clock is asked for.
__global__ void k_clock_probe(unsigned long long ref_ticks, unsigned long long* __restrict__ out)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
    unsigned long long r;
    unsigned spins = 0u;                                             // (bounded whatever the counter does: a probe must not be able to hang a queue)
    do { __builtin_amdgcn_s_sleep(16); r = __builtin_amdgcn_s_memrealtime(); } while (r - r0 < ref_ticks && ++spins < (1u << 25));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[0] = t1 - t0;
    out[1] = r - r0;
}


extern "C" {

#ifndef MKAMD_SRC_HASH            // _build.py: sha256 over the sources this library was compiled from (first 16 hex digits)
#define MKAMD_SRC_HASH "unstamped"
#endif
// (the hash is what ties a measurement to a build: profiles/*_pmc_counters.json carry it, bench.py refuses counters of another build)
const char* mkamd_version(void) { return "moleculekit_amd 0.4.0 (gfx950, HIP) src " MKAMD_SRC_HASH MKAMD_BUILD_KIND; }

const char* mkamd_last_error(void) { return g_last_error; }

int mkamd_device_count(int* count)
try {
    if (!count) return fail(MKAMD_EINVAL, "count is NULL");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    *count = n;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_create(int device, mkamd_ctx** out)
try {
    if (!out) return fail(MKAMD_EINVAL, "ctx out-pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MKAMD_ENODEV, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= n) return fail(MKAMD_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    mkamd_ctx* c = new mkamd_ctx();
    c->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cus = cus; }
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    c->stream = c->main_stream = c->own_stream;
    // the pre-pass kernels are small and latency-bound: give their queue the highest priority so that they
    // get wave slots as the (huge) tile grid of the previous call retires waves, instead of waiting for it to drain
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    bool ok = hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
        ok = hipEventCreateWithFlags(&c->ev_pre_done[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_tile_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) c->side_stream = nullptr;          // fall back to a single in-order stream
    // tier statistics come back through pinned host memory the dense kernel writes directly (no copy, no sync)
    if (hipHostMalloc((void**)&c->fb_host, FEEDBACK_WORDS * sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
        for (int i = 0; i < FEEDBACK_WORDS; ++i) c->fb_host[i] = 0u;
        if (hipHostGetDevicePointer((void**)&c->fb_dev, c->fb_host, 0) != hipSuccess) c->fb_dev = nullptr;
    } else {
        c->fb_host = nullptr;                   // no feedback: the leanest tier is always used
    }
    if (!c->fb_dev && c->fb_host) { (void)hipHostFree(c->fb_host); c->fb_host = nullptr; }
    *out = c;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_destroy(mkamd_ctx* ctx)
try {
    if (!ctx) return MKAMD_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (int i = 0; i < 2 * WS_NSLOTS; ++i)
        if (ctx->bufs[i]) (void)hipFree(ctx->bufs[i]);
    for (auto& ev : ctx->ev_used) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : ctx->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_host_done) (void)hipEventDestroy(ctx->ev_host_done);
    if (ctx->ev_inputs) (void)hipEventDestroy(ctx->ev_inputs);
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_pre_done[i]) (void)hipEventDestroy(ctx->ev_pre_done[i]);
        if (ctx->ev_tile_done[i]) (void)hipEventDestroy(ctx->ev_tile_done[i]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->fb_host) (void)hipHostFree(ctx->fb_host);
    if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
    if (ctx->out_host) (void)hipHostFree(ctx->out_host);
    delete ctx;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_stream(mkamd_ctx* ctx, void* hip_stream)
try {
    int st = check_ctx(ctx);
    if (st) return st;
    // NULL is a real stream (the legacy default stream, which is what torch.cuda.current_stream() usually is);
    // (void*)-1 selects the context's own stream again
    hipStream_t next = hip_stream == (void*)-1 ? ctx->own_stream : (hipStream_t)hip_stream;
    if (next == ctx->main_stream) return MKAMD_OK;
    HIP_TRY(hipStreamSynchronize(ctx->main_stream)); // workspace is shared between the streams
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    ctx->tile_pending[0] = ctx->tile_pending[1] = false;
    ctx->have_pre_tile_event = false;         // the next pipelined pre-pass takes its marker from the NEW stream
    ctx->stream = ctx->main_stream = next;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_abandon_pending(mkamd_ctx* ctx)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (!ctx->pending.active) return MKAMD_OK;
    // the kernels of the abandoned call still write the result buffer and read the workspace: drained before anything reuses them
    ctx->pending.active = false;
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    HIP_TRY(hipStreamSynchronize(ctx->main_stream));
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_synchronize(mkamd_ctx* ctx)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    HIP_TRY(hipStreamSynchronize(ctx->main_stream));
    return collect_async_errors(ctx);
} MK_API_CATCH

int mkamd_ctx_poll_errors(mkamd_ctx* ctx)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    // the dense pass of every lattice call mirrors the device-side flag into pinned host memory: reading it costs
    // nothing and needs no synchronisation; only a raised flag pays for the drain + the precise report
    if (!ctx->fb_host || ((volatile unsigned*)ctx->fb_host)[NTIER + 1] == 0u) return MKAMD_OK;
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    HIP_TRY(hipStreamSynchronize(ctx->main_stream));
    ctx->err_mirrored = false;
    st = collect_async_errors(ctx);
    ((volatile unsigned*)ctx->fb_host)[NTIER + 1] = 0u;
    return st;
} MK_API_CATCH

int mkamd_ctx_device_info(mkamd_ctx* ctx, char* name, size_t len, int* compute_units,
                          uint64_t* hbm_bytes, char* arch, size_t arch_len)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, ctx->device));
    if (name && len) { strncpy(name, p.name, len - 1); name[len - 1] = 0; }
    if (arch && arch_len) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_tile_k(mkamd_ctx* ctx, int k)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (k != 0 && k != 4 && k != 8) return fail(MKAMD_EINVAL, "tile K must be 0 (auto), 4 or 8");
    ctx->tile_k = k;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_lds_tier(mkamd_ctx* ctx, int tier)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (tier < -1 || tier >= NTIER) return fail(MKAMD_EINVAL, "LDS tier must be -1 (adaptive), 0, 1 or 2");
    ctx->lds_tier = tier;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_prepass_mode(mkamd_ctx* ctx, int mode)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (mode < -1 || mode > 1) return fail(MKAMD_EINVAL, "pre-pass mode must be -1 (automatic), 0 (kernel chain) or 1 (per-item)");
    ctx->prepass_mode = mode;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_tile_team(mkamd_ctx* ctx, int mode)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (mode < -1 || (mode > 1 && mode != 4 && mode != 8 && mode != 16))
        return fail(MKAMD_EINVAL, "tile team mode must be -1 (automatic), 0 (one wave per tile), 1 (a team of waves per tile) or 4 / 8 / 16 (a team of that many waves)");
    ctx->tile_team = mode;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_tile_items(mkamd_ctx* ctx, int mode)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (mode < -1 || mode > 1) return fail(MKAMD_EINVAL, "tile items mode must be -1 (automatic), 0 (tiles on their own) or 1 (a workgroup per item)");
    ctx->tile_items = mode;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_exact_redo(mkamd_ctx* ctx, int mode)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (mode != 0 && mode != -1) return fail(MKAMD_EINVAL, "exact redo mode must be 0 (automatic) or -1 (recompute inside the last launch)");
    ctx->exact_redo = mode;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_fine_cells(mkamd_ctx* ctx, int on)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    ctx->fine_cells = on != 0;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_direct_binning(mkamd_ctx* ctx, int mode)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (mode < -1 || mode > 2) return fail(MKAMD_EINVAL, "direct binning mode must be -1 (automatic), 0 (never), 1 (whenever possible) or 2 (the one-launch pre-pass whenever possible)");
    ctx->direct = mode;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_value_tolerance(mkamd_ctx* ctx, double eps)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (!(eps >= 0.0) || eps > 1e-5) return fail(MKAMD_EINVAL, "value tolerance must be in [0, 1e-5] (0 = off)");
    ctx->value_tol = eps;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_force_general(mkamd_ctx* ctx, int on)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    ctx->force_general = on != 0;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_pipelining(mkamd_ctx* ctx, int on)
try {
    int st = check_ctx(ctx);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->main_stream));
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    ctx->pipelining = on != 0;
    ctx->pipeline_broken = false;             // both streams are drained: a fresh start
    ctx->have_pre_tile_event = false;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_promise_inputs(mkamd_ctx* ctx, void* hip_event)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    ctx->promise = true;
    ctx->promise_event = (hipEvent_t)hip_event;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_pipelined_calls(mkamd_ctx* ctx, int64_t* n)
try {
    if (!ctx || !n) return fail(MKAMD_EINVAL, "ctx / n is NULL");
    *n = (int64_t)ctx->n_pipelined;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_last_tile_kernel(mkamd_ctx* ctx, char* name, size_t name_cap)
try {
    if (!ctx || !name || name_cap == 0) return fail(MKAMD_EINVAL, "ctx / name is NULL");
    static const char* const base[] = {"k_voxelize_tiles", "k_voxelize_tiles_lean", "k_voxelize_tiles_team", "k_voxelize_items"};
    if (ctx->last_flavour < 0 || ctx->last_flavour > 3) snprintf(name, name_cap, "%s", "");
    else if (ctx->last_flavour == 3) snprintf(name, name_cap, "mkamd::%s<%d>", base[3], ctx->last_K);
    else snprintf(name, name_cap, "mkamd::%s<%d, %d>", base[ctx->last_flavour], ctx->last_K, ctx->last_ecap);
    return MKAMD_OK;
} MK_API_CATCH

// the short correctly-rounded square root of the distance kernels against the provable form, over every float it is used for
int mkamd_selftest_sqrt(mkamd_ctx* ctx, uint64_t* mismatches, uint32_t* first_bad_bits)
try {
    int st = check_ctx(ctx);
    if (st) return st;
    if (!mismatches) return fail(MKAMD_EINVAL, "mismatches pointer is NULL");
    void* w = nullptr;
    if ((st = ctx->ensure(WS_D_TOT, 16, &w, 0))) return st;
    HIP_TRY(hipMemsetAsync(w, 0, 16, ctx->stream));
    const unsigned lo = 0x0F800000u;                                // 2^-96: below it mk_fsqrt_rn takes its scaled branch
    const unsigned long long n = 0x7F800000ull - lo;
    if ((st = ctx->launch(k_selftest_sqrt, dim3(65536), dim3(256), lo, n, (unsigned long long*)w, (unsigned*)((char*)w + 8)))) return st;
    unsigned long long host[2] = {0ull, 0ull};
    HIP_TRY(hipMemcpyAsync(host, w, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *mismatches = (uint64_t)host[0];
    if (first_bad_bits) *first_bad_bits = (uint32_t)(host[1] & 0xffffffffull);
    return MKAMD_OK;
} MK_API_CATCH

// Every distance kernel takes its roots with mk_fsqrt_rn_ordinary, whose correct rounding is a PROPERTY OF THIS CHIP's v_rsq_f32 (checked
// exhaustively, not proved: mk_device.h).  So the exhaustive comparison runs once per process and device, in front of the first
// distance call (a few milliseconds), and a device -- another stepping, firmware or compiler lowering of the reciprocal square root --
// on which it finds a mismatch gets no distances at all rather than wrong last bits.  -1 unknown, 1 good, 0 bad.
static std::atomic<int> g_sqrt_verdict[64];
static struct SqrtVerdictInit { SqrtVerdictInit() { for (auto& v : g_sqrt_verdict) v.store(-1); } } g_sqrt_verdict_init;

static int dist_entry(mkamd_ctx* ctx)
{
    int st = check_ctx(ctx);
    if (st) return st;
    const int d = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    int v = g_sqrt_verdict[d].load();
    if (v < 0) {
        uint64_t bad = 0;
        uint32_t first = 0;
        if ((st = mkamd_selftest_sqrt(ctx, &bad, &first))) return st;
        v = bad == 0 ? 1 : 0;
        g_sqrt_verdict[d].store(v);
        if (!v) {
            char msg[200];
            snprintf(msg, sizeof msg, "float32 square-root self-test failed on this device (%llu mismatches, first at bits 0x%08x): "
                     "the distance kernels would not be bit-exact here", (unsigned long long)bad, first);
            return fail(MKAMD_EHIP, msg);
        }
    }
    if (!v) return fail(MKAMD_EHIP, "float32 square-root self-test failed on this device: the distance kernels would not be bit-exact here");
    return MKAMD_OK;
}

int mkamd_clock_probe_dev(mkamd_ctx* ctx, void* hip_stream, int64_t microseconds, uint64_t* d_ticks2)
try {
    if (!ctx || !d_ticks2) return fail(MKAMD_EINVAL, "ctx / result pointer is NULL");
    if (microseconds < 1 || microseconds > 5000000) return fail(MKAMD_EINVAL, "clock probe: 1 us .. 5 s");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, (unsigned long long)microseconds * 100ull,
                       (unsigned long long*)d_ticks2);
    HIP_TRY(hipGetLastError());
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_dist_kernels(mkamd_ctx* ctx, int avoid_mask)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (avoid_mask < 0 || avoid_mask > 255) return fail(MKAMD_EINVAL, "avoid mask: bits 1 (block-per-frame kernel), 2 (row kernel), 4 (rectangular tile kernel), 8 (16-byte row stores), 16 (the row kernel wherever it applies), 32 (host calls upload the whole coordinate array), 64 (selfdist calls keep the pair-table kernel), 128 (short-row calls of few frames keep the tile kernel)");
    ctx->dist_avoid = avoid_mask;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_set_reduction_block(mkamd_ctx* ctx, int block)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    if (block != 0 && block != 4 && block != 8 && block != -1 && block != -2 && block != 104 && block != 108)
        return fail(MKAMD_EINVAL, "reduction block: 0 (choose), 4, 8 (+ 100: blocks of four waves), -1 (the generic kernel) or -2 (the few-frame kernel)");
    ctx->reduction_block = block;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_last_dist_kernel(mkamd_ctx* ctx, char* name, size_t name_cap)
try {
    if (!ctx || !name || name_cap == 0) return fail(MKAMD_EINVAL, "ctx / name is NULL");
    snprintf(name, name_cap, "%s", ctx->last_dist_kernel);
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_withdraw_promise(mkamd_ctx* ctx)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    ctx->promise = false;
    ctx->promise_event = nullptr;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_enable_kernel_timing(mkamd_ctx* ctx, int enable)
try {
    if (!ctx) return fail(MKAMD_EINVAL, "ctx is NULL");
    ctx->timing = enable != 0;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_ctx_read_kernel_timing(mkamd_ctx* ctx, double* total_ms, int64_t* launches)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    int64_t n = 0;
    for (auto& ev : ctx->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { tot += ms; ++n; }
        ctx->ev_free.push_back(ev);
    }
    ctx->ev_used.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return MKAMD_OK;
} MK_API_CATCH

// A big result usually lands in FRESH host memory (np.empty / np.zeros: pages nobody has touched), and the copy out of the
// device then faults them in one by one on one thread (measured: 34 ms for 537 MB against 13 ms into pages that exist).
// Called after the kernels are enqueued and before the copy: host threads touch the destination in contiguous slices
// while the GPU works (xtc_reader.h, prefault_output: it writes zeros, so only for buffers the copy overwrites entirely).
static void prefault_big_result(void* dst, size_t bytes)
{
    if (!dst || bytes < ((size_t)32 << 20)) return;
    const int nt = (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
    mkamd::xtc::prefault_output(static_cast<float*>(dst), bytes / sizeof(float), nt);
}

// ---------------------------------------------------------------------------------------------
// explicit centres
// ---------------------------------------------------------------------------------------------
int mkamd_occupancy_centers_dev(mkamd_ctx* ctx, const double* d_centers, int64_t n_centers,
                                const float* d_coords, int64_t n_atoms, const void* d_sigmas,
                                int sigmas_are_f64, int32_t n_channels, const double* box_host,
                                float* d_features)
try {
    int st = check_ctx(ctx);
    if (st) return st;
    if (n_centers > 0 && (!d_centers || !d_features)) return fail(MKAMD_EINVAL, "centers/features pointer is NULL");
    if (n_atoms > 0 && (!d_coords || !d_sigmas)) return fail(MKAMD_EINVAL, "coords/sigmas pointer is NULL");
    std::string err;
    st = run_centers(*ctx, d_centers, n_centers, d_coords, n_atoms, d_sigmas, sigmas_are_f64, n_channels,
                     box_host, d_features, err);
    if (st && !err.empty()) return fail(st, err);
    return st;
} MK_API_CATCH

int mkamd_occupancy_centers_host(mkamd_ctx* ctx, const double* centers, int64_t V, const float* coords,
                                 int64_t N, const void* sigmas, int sigmas_are_f64, int32_t C,
                                 const double* box, float* features)
try {
    int st = check_ctx(ctx);
    if (st) return st;
    if (V < 0 || N < 0 || C <= 0) return fail(MKAMD_EINVAL, "n_centers/n_atoms must be >= 0 and n_channels > 0");
    if (V == 0) return MKAMD_OK;
    if (!centers || !features) return fail(MKAMD_EINVAL, "centers/features pointer is NULL");
    if (N > 0 && (!coords || !sigmas)) return fail(MKAMD_EINVAL, "coords/sigmas pointer is NULL");
    const size_t sz = sigmas_are_f64 ? 8 : 4;
    void *dc = nullptr, *dx = nullptr, *ds = nullptr, *dout = nullptr;
    if ((st = ctx->ensure(WS_H_CENTERS, (size_t)V * 24, &dc))) return st;
    if ((st = ctx->ensure(WS_H_COORDS, (size_t)N * 12, &dx))) return st;
    if ((st = ctx->ensure(WS_H_SIGMAS, (size_t)N * C * sz, &ds))) return st;
    if ((st = ctx->ensure(WS_H_OUT, (size_t)V * C * 4, &dout))) return st;
    HIP_TRY(hipMemcpyAsync(dc, centers, (size_t)V * 24, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) {
        HIP_TRY(hipMemcpyAsync(dx, coords, (size_t)N * 12, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ds, sigmas, (size_t)N * C * sz, hipMemcpyHostToDevice, ctx->stream));
    }
    st = mkamd_occupancy_centers_dev(ctx, (const double*)dc, V, (const float*)dx, N, ds, sigmas_are_f64, C, box, (float*)dout);
    if (st) return st;
    prefault_big_result(features, (size_t)V * C * 4);
    HIP_TRY(hipMemcpyAsync(features, dout, (size_t)V * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

static int voxelize_lattice_host_begin_impl(mkamd_ctx* ctx, int32_t B, const float* coords, const int64_t* atom_offsets,
                                            const void* sigmas, int sigmas_are_f64, int32_t C, const double* origins,
                                            const int32_t* nvoxels, double voxelsize, const float* box, int32_t max_images);
static int voxelize_lattice_host_end_impl(mkamd_ctx* ctx, float* features, double* features64, double* max_into);

int mkamd_calculate_occupancy(mkamd_ctx* ctx, const double* centers, int64_t V, const float* coords,
                              int64_t N, const double* sigmas, int32_t C, double* results)
try {
    if (V < 0 || N < 0 || C <= 0) return fail(MKAMD_EINVAL, "n_centers/n_atoms must be >= 0 and n_channels > 0");
    if (V == 0 || N == 0) return ctx ? MKAMD_OK : fail(MKAMD_EINVAL, "ctx is NULL");
    if (!results) return fail(MKAMD_EINVAL, "results pointer is NULL");
    int st0 = check_ctx(ctx);
    if (st0) return st0;
    std::vector<float>& tmp = ctx->f32_stage;               // context-owned: no fresh pages to fault in on every call
    // The reference's only caller hands this function a getCenters LATTICE (voxeldescriptors.py:356 via _getOccupancyC):
    // recognised (two passes over the centres, host) it takes the tiled lattice kernels -- microseconds where the
    // pairwise kernel below tests N x V pairs in double.
    // A status other than "this geometry is not for the tiled kernels" (MKAMD_EINVAL: every argument has been checked
    // above, so what is left are the capacity limits of the lattice plan) goes back to the caller -- a HIP error or a
    // failed allocation is not retried on a kernel a hundred times slower (route_calculate_occupancy, pipeline.h).
    double bb_min[3], vs = 0.0;
    int32_t nv[3];
    const bool is_lattice = mkamd::lattice_from_centers(centers, (long long)V, bb_min, nv, &vs);
    bool lattice_done = false;
    const int st = mkamd::route_calculate_occupancy(is_lattice,
        [&] { const int64_t offs[2] = {0, N};                  // (the in-place maximum is taken by the call's second half)
              const int s1 = voxelize_lattice_host_begin_impl(ctx, 1, coords, offs, sigmas, 1, C, bb_min, nv, vs, nullptr, 0);
              if (s1) return s1;
              lattice_done = true;
              return voxelize_lattice_host_end_impl(ctx, nullptr, nullptr, results); },
        [&] { tmp.resize((size_t)V * C);
              return mkamd_occupancy_centers_host(ctx, centers, V, coords, N, sigmas, 1, C, nullptr, tmp.data()); });
    if (st) return st;
    if (lattice_done) return MKAMD_OK;
    // in-place max-accumulate, `value > old ? value : old` as occupancy_utils.pyx:61
    const size_t nvals = (size_t)V * C;
    for (size_t i = 0; i < nvals; ++i) {
        const double v = (double)tmp[i];
        if (v > results[i]) results[i] = v;
    }
    return MKAMD_OK;
} MK_API_CATCH

// calculate_occupancy on the HOST: the library's own double-precision implementation (cpu_occupancy.h), no
// context, no device.  Explicit only: nothing in the library or the package takes it on its own.
int mkamd_calculate_occupancy_cpu_threads(const double* centers, int64_t V, const float* coords, int64_t N, const double* sigmas,
                                          int32_t C, double* results, int32_t n_threads)
try {
    if (V < 0 || N < 0 || C <= 0) return fail(MKAMD_EINVAL, "n_centers/n_atoms must be >= 0 and n_channels > 0");
    if (V == 0 || N == 0) return MKAMD_OK;
    if (!centers || !coords || !sigmas || !results) return fail(MKAMD_EINVAL, "centers/coords/sigmas/results pointer is NULL");
    if (N > 0xFFFFFFF0LL) return fail(MKAMD_EINVAL, "more than 2^32 atoms");
    mkamd::cpu::calculate_occupancy(centers, V, coords, N, sigmas, C, results, (int)n_threads);
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_calculate_occupancy_cpu(const double* centers, int64_t V, const float* coords, int64_t N, const double* sigmas,
                                  int32_t C, double* results)
try {
    return mkamd_calculate_occupancy_cpu_threads(centers, V, coords, N, sigmas, C, results, 0);
} MK_API_CATCH

// ---------------------------------------------------------------------------------------------
// lattice grids (the hot path)
// ---------------------------------------------------------------------------------------------
int mkamd_voxelize_lattice_dev(mkamd_ctx* ctx, int32_t B, const float* d_coords,
                               const int64_t* d_atom_offsets, int64_t total_atoms, const void* d_sigmas,
                               int sigmas_are_f64, int32_t C, const double* d_origins,
                               const int32_t* nvoxels, double voxelsize, const float* d_box,
                               int32_t max_images, float* d_features)
try {
    return mkamd_voxelize_lattice_aug_dev(ctx, B, d_coords, d_atom_offsets, total_atoms, d_sigmas, sigmas_are_f64, C,
                                          d_origins, nvoxels, voxelsize, d_box, max_images, nullptr, d_features);
} MK_API_CATCH

static int voxelize_lattice_dev_impl(mkamd_ctx* ctx, int32_t B, const float* d_coords,
                                     const int64_t* d_atom_offsets, int64_t total_atoms, const void* d_sigmas,
                                     int sigmas_are_f64, int32_t C, const double* d_origins,
                                     const int32_t* nvoxels, double voxelsize, const float* d_box,
                                     int32_t max_images, const double* d_affine, float* d_features, const mkamd_topology* topo)
{
    int st = check_ctx(ctx);
    if (st) return st;
    if (!nvoxels) return fail(MKAMD_EINVAL, "nvoxels pointer is NULL");
    if (B > 0 && (!d_atom_offsets || !d_origins || !d_features)) return fail(MKAMD_EINVAL, "atom_offsets/origins/features pointer is NULL");
    if (total_atoms > 0 && (!d_coords || (!d_sigmas && !topo))) return fail(MKAMD_EINVAL, "coords/sigmas pointer is NULL");
    if (topo && topo->device != ctx->device) return fail(MKAMD_EINVAL, "the topology lives on another device than the context");
    if ((st = ensure_err_flag(ctx))) return st;
    LatticeProblem P;
    P.B = B; P.total_atoms = total_atoms; P.C = C; P.sigmas_f64 = sigmas_are_f64;
    P.nvox[0] = nvoxels[0]; P.nvox[1] = nvoxels[1]; P.nvox[2] = nvoxels[2];
    P.voxelsize = voxelsize; P.pbc = d_box ? 1 : 0; P.max_images = d_box ? max_images : 1;
    P.tile_k = ctx->tile_k; P.force_general = ctx->force_general; P.lds_tier = ctx->lds_tier; P.prepass_mode = ctx->prepass_mode; P.tile_team = ctx->tile_team; P.tile_items = ctx->tile_items; P.exact_redo_list = ctx->exact_redo; P.fine_cells = ctx->fine_cells; P.value_tol = ctx->value_tol; P.direct = ctx->direct; P.seq = ctx->seq_next; ctx->seq_next = 0u;
    P.coords = d_coords; P.atom_offsets = (const long long*)d_atom_offsets; P.sigmas = d_sigmas;
    P.origins = d_origins; P.box = d_box; P.affine = d_affine; P.out = d_features;
    P.topo = topo ? &topo->dev : nullptr;
    std::string err;
    // a promise (mkamd_ctx_promise_inputs) is about the next call of THIS entry point made from outside the library: a host
    // call reaches here through voxelize_lattice_host_begin_impl, which has set the promise aside (its inputs were uploaded
    // on the main stream just now, behind the marker a pipelined pre-pass waits for)
    st = run_lattice(*ctx, P, err);
    ctx->promise = false; ctx->promise_event = nullptr;       // one call's worth
    if (st && !err.empty()) return fail(st, err);
    return st;
}

int mkamd_voxelize_lattice_aug_dev(mkamd_ctx* ctx, int32_t B, const float* d_coords,
                                   const int64_t* d_atom_offsets, int64_t total_atoms, const void* d_sigmas,
                                   int sigmas_are_f64, int32_t C, const double* d_origins,
                                   const int32_t* nvoxels, double voxelsize, const float* d_box,
                                   int32_t max_images, const double* d_affine, float* d_features)
try {
    return voxelize_lattice_dev_impl(ctx, B, d_coords, d_atom_offsets, total_atoms, d_sigmas, sigmas_are_f64, C, d_origins, nvoxels, voxelsize,
                                     d_box, max_images, d_affine, d_features, nullptr);
} MK_API_CATCH

// ---- topology reuse: the frames of a trajectory share everything the pre-pass derives from the sigmas ----
int mkamd_topology_create_dev(mkamd_ctx* ctx, const void* d_sigmas, int sigmas_are_f64, int64_t n_atoms, int32_t C, double voxelsize,
                              mkamd_topology** out)
try {
    if (!out) return fail(MKAMD_EINVAL, "topology out-pointer is NULL");
    *out = nullptr;
    int st = check_ctx(ctx);
    if (st) return st;
    if (n_atoms <= 0 || C <= 0 || !d_sigmas) return fail(MKAMD_EINVAL, "a topology needs n_atoms > 0, n_channels > 0 and the sigmas");
    if (n_atoms > 0x7fffffffLL) return fail(MKAMD_EINVAL, "a topology holds at most 2^31 atoms");
    const int G = ceil_div(C, CHG);
    const size_t a256 = 255, sig_bytes = (size_t)n_atoms * C * (sigmas_are_f64 ? 8 : 4);
    const size_t o_cw = (sig_bytes + a256) & ~a256, o_ids = (o_cw + (size_t)n_atoms * G * sizeof(uint2) + a256) & ~a256,
                 o_tab = (o_ids + (size_t)n_atoms * G * sizeof(unsigned) + a256) & ~a256, o_flags = o_tab + 256, o_wide = o_flags + 256,
                 total = o_wide + (size_t)n_atoms * sizeof(unsigned);
    mkamd_topology* t = new mkamd_topology();
    t->device = ctx->device;
    hipError_t e = hipMalloc(&t->mem, total);
    if (e != hipSuccess) { delete t; return hip_fail(e, "hipMalloc(topology)"); }
    char* m = (char*)t->mem;
    auto drop = [&](int code) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(t->mem); delete t; return code; };
    if ((e = hipMemcpyAsync(m, d_sigmas, sig_bytes, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess) return drop(hip_fail(e, "hipMemcpyAsync(topology sigmas)"));
    if ((e = hipMemsetAsync(m + o_flags, 0, 256, ctx->stream)) != hipSuccess) return drop(hip_fail(e, "hipMemsetAsync(topology flags)"));
    std::string err;
    st = run_topology_build(*ctx, m, sigmas_are_f64, (long long)n_atoms, C, voxelsize, (uint2*)(m + o_cw), (unsigned*)(m + o_ids),
                            (unsigned*)(m + o_tab), (int*)(m + o_flags), (unsigned*)(m + o_wide), err);
    if (st) return drop(err.empty() ? st : fail(st, err));
    unsigned table[CLS_TABLE_WORDS];
    int flags2[2] = {0, 0};
    if ((e = hipMemcpyAsync(table, m + o_tab, sizeof table, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(flags2, m + o_flags, sizeof flags2, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return drop(hip_fail(e, "topology read-back"));
    const int flags = flags2[0];
    const unsigned n_wide = (unsigned)flags2[1];
    if (n_wide > 1u) {
        // the list arrives in the order the lanes' atomics did: sorted here, once, so that a handle is the same whenever it is built
        std::vector<unsigned> wl(n_wide);
        if ((e = hipMemcpyAsync(wl.data(), m + o_wide, (size_t)n_wide * 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return drop(hip_fail(e, "topology read-back (wide atoms)"));
        std::sort(wl.begin(), wl.end());
        if ((e = hipMemcpyAsync(m + o_wide, wl.data(), (size_t)n_wide * 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return drop(hip_fail(e, "topology write-back (wide atoms)"));
    }
    if (table[CLS_OVERFLOW] != CLS_EMPTY)
        return drop(fail(MKAMD_EINVAL, "more than 15 distinct sigma values: no class ids to reuse (use the plain entry point)"));
    t->dev.n = (long long)n_atoms; t->dev.C = C; t->dev.G = G; t->dev.sigmas_f64 = sigmas_are_f64; t->dev.voxelsize = voxelsize;
    t->dev.sigmas = m; t->dev.cw = (const uint2*)(m + o_cw); t->dev.ids = (const unsigned*)(m + o_ids); t->dev.table = (const unsigned*)(m + o_tab);
    t->dev.overflow = false; t->dev.wide = (flags & 1) != 0;
    t->dev.wide_list = (const unsigned*)(m + o_wide); t->dev.n_wide = n_wide;
    *out = t;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_topology_create_host(mkamd_ctx* ctx, const void* sigmas, int sigmas_are_f64, int64_t n_atoms, int32_t C, double voxelsize,
                               mkamd_topology** out)
try {
    if (!out) return fail(MKAMD_EINVAL, "topology out-pointer is NULL");
    *out = nullptr;
    int st = check_ctx(ctx);
    if (st) return st;
    if (n_atoms <= 0 || C <= 0 || !sigmas) return fail(MKAMD_EINVAL, "a topology needs n_atoms > 0, n_channels > 0 and the sigmas");
    void* ds = nullptr;
    const size_t bytes = (size_t)n_atoms * C * (sigmas_are_f64 ? 8 : 4);
    if ((st = ctx->ensure(WS_H_SIGMAS, bytes, &ds))) return st;
    HIP_TRY(hipMemcpyAsync(ds, sigmas, bytes, hipMemcpyHostToDevice, ctx->stream));
    return mkamd_topology_create_dev(ctx, ds, sigmas_are_f64, n_atoms, C, voxelsize, out);
} MK_API_CATCH

int mkamd_topology_destroy(mkamd_ctx* ctx, mkamd_topology* topo)
try {
    if (!topo) return MKAMD_OK;
    if (ctx) {                                   // calls that read the handle may still be in flight on the context's streams
        (void)hipSetDevice(ctx->device);
        if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamSynchronize(ctx->main_stream);
    } else {
        (void)hipSetDevice(topo->device);
        (void)hipDeviceSynchronize();
    }
    if (topo->mem) (void)hipFree(topo->mem);
    delete topo;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_topology_info(const mkamd_topology* topo, int64_t* n_atoms, int32_t* n_channels, double* voxelsize, int32_t* has_wide_sigmas)
try {
    if (!topo) return fail(MKAMD_EINVAL, "topology is NULL");
    if (n_atoms) *n_atoms = (int64_t)topo->dev.n;
    if (n_channels) *n_channels = topo->dev.C;
    if (voxelsize) *voxelsize = topo->dev.voxelsize;
    if (has_wide_sigmas) *has_wide_sigmas = topo->dev.wide ? 1 : 0;
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_voxelize_lattice_topo_dev(mkamd_ctx* ctx, int32_t B, const float* d_coords, const int64_t* d_atom_offsets, int64_t total_atoms,
                                    const mkamd_topology* topo, const double* d_origins, const int32_t* nvoxels, double voxelsize,
                                    const float* d_box, int32_t max_images, const double* d_affine, float* d_features)
try {
    if (!topo) return fail(MKAMD_EINVAL, "topology is NULL");
    return voxelize_lattice_dev_impl(ctx, B, d_coords, d_atom_offsets, total_atoms, nullptr, topo->dev.sigmas_f64, topo->dev.C, d_origins, nvoxels,
                                     voxelsize, d_box, max_images, d_affine, d_features, topo);
} MK_API_CATCH

// End of a small synchronous call (tens of microseconds of GPU work): poll the stream instead of blocking on it -- the
// blocking wait's wake-up costs more than the polls; a call that is still running after ~2 000 polls blocks after all.
// The runtime answers a stream query with a marker packet of its own when the stream's last command is a kernel -- and that
// packet then takes its ~5-10 us to retire while the caller polls.  A small host call therefore records an event right
// behind its last launch (it retires while the host is busy with the result) and asks the EVENT.
static hipError_t wait_for_small_call(hipStream_t s, hipEvent_t done = nullptr)
{
    for (int i = 0; i < 2000; ++i) {
        const hipError_t e = done ? hipEventQuery(done) : hipStreamQuery(s);
        if (e != hipErrorNotReady) {
            if (i != 0 && e == hipSuccess) (void)hipGetLastError();    // "not ready" is an answer, not an error to find later
            return e;
        }
    }
    (void)hipGetLastError();
    return done ? hipEventSynchronize(done) : hipStreamSynchronize(s);
}

// float32 -> float64 over the result of a call (the drop-in path returns the reference's float64 [V, C]): the baseline
// x86-64 build converts two values per instruction; where the host has AVX2 (checked at run time) eight
__attribute__((target("avx2"))) static void widen_avx2(const float* __restrict__ src, double* __restrict__ dst, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}
// AVX-512 hosts: 16 values per trip, the source (device-written pinned memory: it comes from DRAM) prefetched ahead
__attribute__((target("avx512f"))) static void widen_avx512(const float* __restrict__ src, double* __restrict__ dst, size_t n)
{
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        __builtin_prefetch(src + i + 512, 0, 0);
        typedef float v16f __attribute__((vector_size(64), aligned(4)));
        typedef double v8d __attribute__((vector_size(64), aligned(8)));
        typedef float v8f __attribute__((vector_size(32)));
        const v16f v = *(const v16f*)(src + i);
        const v8f lo = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(v, v, 8, 9, 10, 11, 12, 13, 14, 15);
        *(v8d*)(dst + i) = __builtin_convertvector(lo, v8d);
        *(v8d*)(dst + i + 8) = __builtin_convertvector(hi, v8d);
    }
    for (; i < n; ++i) dst[i] = (double)src[i];
}
static void widen(const float* __restrict__ src, double* __restrict__ dst, size_t n)
{
    static const bool avx512 = __builtin_cpu_supports("avx512f") && getenv("MKAMD_NO_AVX512") == nullptr;
    if (avx512) { widen_avx512(src, dst, n); return; }
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { widen_avx2(src, dst, n); return; }
    for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}

// `features64` != NULL: the caller wants the reference's float64 [B,V,C] (voxeldescriptors.py:531); the widening is
// done here, in the pass that takes the results out of the pinned buffer anyway, instead of in a second pass in numpy

// A host call in two halves: `begin` checks, ships the inputs and enqueues the kernels; `end` waits and takes the result
// out.  The one-piece entry points run them back to back; the drop-in getVoxelDescriptors does its own host work (the copy
// of the cached voxel centres, 330 KB for a 24^3 grid) between the two, while the device computes.
static int voxelize_lattice_host_begin_impl(mkamd_ctx* ctx, int32_t B, const float* coords, const int64_t* atom_offsets,
                                            const void* sigmas, int sigmas_are_f64, int32_t C, const double* origins,
                                            const int32_t* nvoxels, double voxelsize, const float* box, int32_t max_images)
{
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (ctx->pending.active) {                                      // a begin without its end: that call is abandoned
        (void)hipStreamSynchronize(ctx->stream);
        ctx->pending.active = false;
    }
    if (B < 0 || C <= 0) return fail(MKAMD_EINVAL, "n_items must be >= 0 and n_channels > 0");
    if (!nvoxels) return fail(MKAMD_EINVAL, "nvoxels pointer is NULL");
    if (nvoxels[0] < 0 || nvoxels[1] < 0 || nvoxels[2] < 0) return fail(MKAMD_EINVAL, "nvoxels must be >= 0");
    const long long V = (long long)nvoxels[0] * nvoxels[1] * nvoxels[2];
    if (B == 0 || V == 0) { ctx->pending = mkamd_ctx::PendingHostCall{}; ctx->pending.active = true; return MKAMD_OK; }
    if (!atom_offsets || !origins) return fail(MKAMD_EINVAL, "atom_offsets/origins pointer is NULL");
    if (atom_offsets[0] != 0) return fail(MKAMD_EINVAL, "atom_offsets[0] must be 0");
    for (int b = 0; b < B; ++b)
        if (atom_offsets[b + 1] < atom_offsets[b]) return fail(MKAMD_EINVAL, "atom_offsets must be non-decreasing");
    const int64_t N = atom_offsets[B];
    if (N > 0 && (!coords || !sigmas)) return fail(MKAMD_EINVAL, "coords/sigmas pointer is NULL");
    if (box && max_images <= 0) {
        std::string err;
        const int nv[3] = {nvoxels[0], nvoxels[1], nvoxels[2]};
        max_images = max_images_from_boxes(box, B, nv, voxelsize, err);
        if (max_images < 0) return fail(MKAMD_EBOX, err);
    }
    const size_t sz = sigmas_are_f64 ? 8 : 4;
    const size_t out_bytes = (size_t)B * (size_t)V * (size_t)C * 4;
    void *dx = nullptr, *ds = nullptr, *doff = nullptr, *dorg = nullptr, *dbox = nullptr, *dout = nullptr;
    // small results (the drop-in path: one molecule per call): the kernels store straight into pinned host memory
    // mapped into the device's address space -- the stores cross PCIe while the kernel runs, and the copy engine's
    // start-up latency (~15 us before a ~13 us copy on the 3PTB grid) disappears
    constexpr size_t OUT_HOST_BYTES = (size_t)1 << 20;
    bool mapped_out = out_bytes <= OUT_HOST_BYTES;
    if (mapped_out && !ctx->out_host) {
        if (hipHostMalloc(&ctx->out_host, OUT_HOST_BYTES, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&ctx->out_host_dev, ctx->out_host, 0) != hipSuccess) {
            if (ctx->out_host) (void)hipHostFree(ctx->out_host);
            ctx->out_host = nullptr;
            (void)hipGetLastError();
            mapped_out = false;
        }
    }
    if (mapped_out) dout = ctx->out_host_dev;
    else if ((st = ctx->ensure(WS_H_OUT, out_bytes, &dout))) return st;
    // small calls (the drop-in path: one molecule per call) are latency-bound: pack the inputs into one pinned
    // buffer and ship them with ONE asynchronous copy instead of five staged pageable ones
    const size_t a16 = 15;
    const size_t o_x = 0, o_s = (o_x + (size_t)N * 12 + a16) & ~a16, o_off = (o_s + (size_t)N * C * sz + a16) & ~a16,
                 o_org = (o_off + (size_t)(B + 1) * 8 + a16) & ~a16, o_box = (o_org + (size_t)B * 24 + a16) & ~a16,
                 packed_bytes = o_box + (box ? (size_t)B * 12 : 0);
    constexpr size_t STAGE_BYTES = (size_t)1 << 20;
    bool packed = packed_bytes <= STAGE_BYTES;
    if (packed && !ctx->stage_host) {
        if (hipHostMalloc(&ctx->stage_host, STAGE_BYTES, hipHostMallocMapped) == hipSuccess) {
            ctx->stage_cap = STAGE_BYTES;
            if (hipHostGetDevicePointer(&ctx->stage_host_dev, ctx->stage_host, 0) != hipSuccess) { ctx->stage_host_dev = nullptr; (void)hipGetLastError(); }
        } else { ctx->stage_host = nullptr; packed = false; }
    }
    if (packed) {
        void* dstage = nullptr;
        if ((st = ctx->ensure(WS_H_STAGE, packed_bytes, &dstage))) return st;
        char* h = (char*)ctx->stage_host;       // free again: every _host call ends with a stream synchronize
        if (N > 0) { memcpy(h + o_x, coords, (size_t)N * 12); memcpy(h + o_s, sigmas, (size_t)N * C * sz); }
        memcpy(h + o_off, atom_offsets, (size_t)(B + 1) * 8);
        memcpy(h + o_org, origins, (size_t)B * 24);
        if (box) memcpy(h + o_box, box, (size_t)B * 12);
        // up to 256 KiB the kernels read the packed inputs in place over PCIe (each byte is read once, by the binning):
        // cheaper than waking the copy engine (~8 us + ~8 us before the first kernel starts on the 3PTB call)
        if (ctx->stage_host_dev != nullptr && packed_bytes <= ((size_t)256 << 10)) dstage = ctx->stage_host_dev;
        else HIP_TRY(hipMemcpyAsync(dstage, h, packed_bytes, hipMemcpyHostToDevice, ctx->stream));
        dx = (char*)dstage + o_x; ds = (char*)dstage + o_s; doff = (char*)dstage + o_off; dorg = (char*)dstage + o_org;
        dbox = (char*)dstage + o_box;
    } else {
        if ((st = ctx->ensure(WS_H_COORDS, (size_t)N * 12, &dx))) return st;
        if ((st = ctx->ensure(WS_H_SIGMAS, (size_t)N * C * sz, &ds))) return st;
        if ((st = ctx->ensure(WS_H_OFFSETS, (size_t)(B + 1) * 8, &doff))) return st;
        if ((st = ctx->ensure(WS_H_ORIGINS, (size_t)B * 24, &dorg))) return st;
        if (box && (st = ctx->ensure(WS_H_BOX, (size_t)B * 12, &dbox))) return st;
        if (N > 0) {
            HIP_TRY(hipMemcpyAsync(dx, coords, (size_t)N * 12, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(ds, sigmas, (size_t)N * C * sz, hipMemcpyHostToDevice, ctx->stream));
        }
        HIP_TRY(hipMemcpyAsync(doff, atom_offsets, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dorg, origins, (size_t)B * 24, hipMemcpyHostToDevice, ctx->stream));
        if (box) HIP_TRY(hipMemcpyAsync(dbox, box, (size_t)B * 12, hipMemcpyHostToDevice, ctx->stream));
    }
    // the per-item pre-pass gives every item ONE workgroup: here the offsets are visible, so a ragged batch whose
    // average is small but which holds a huge item is kept on the kernel chain (automatic mode only)
    long long biggest = 0;
    for (int b = 0; b < B; ++b) biggest = std::max<long long>(biggest, atom_offsets[b + 1] - atom_offsets[b]);
    const int saved_mode = ctx->prepass_mode;
    if (saved_mode < 0 && biggest > 16384) ctx->prepass_mode = 0;
    // small results: the host pass over them (copy / float64 widening) starts when the TILE kernel is done (k_tail says so in
    // the host-visible feedback words as it starts), not when the stream is -- see FB_TILES_DONE
    unsigned seq = 0u;
    if (mapped_out && ctx->fb_host != nullptr) { if (++ctx->seq_counter == 0u) ctx->seq_counter = 1u; seq = ctx->seq_counter; }
    ctx->seq_next = seq;
    ctx->tail_reports = false;
    // an outstanding promise is about the next DEVICE call of whoever made it, not about this host call (whose inputs went up
    // on the main stream a moment ago: a pre-pass on the side stream would race the upload): set aside, handed back after
    const bool saved_promise = ctx->promise;
    const hipEvent_t saved_promise_event = ctx->promise_event;
    ctx->promise = false; ctx->promise_event = nullptr;
    st = mkamd_voxelize_lattice_dev(ctx, B, (const float*)dx, (const int64_t*)doff, N, ds, sigmas_are_f64, C,
                                    (const double*)dorg, nvoxels, voxelsize, box ? (const float*)dbox : nullptr,
                                    max_images, (float*)dout);
    ctx->promise = saved_promise; ctx->promise_event = saved_promise_event;
    ctx->prepass_mode = saved_mode;
    ctx->seq_next = 0u;
    if (st) return st;
    ctx->pending.active = true; ctx->pending.out_bytes = out_bytes; ctx->pending.mapped_out = mapped_out; ctx->pending.seq = seq;
    ctx->pending.dout = dout;
    ctx->pending.done = nullptr;
    if (mapped_out) {
        if (!ctx->ev_host_done && hipEventCreateWithFlags(&ctx->ev_host_done, hipEventDisableTiming) != hipSuccess) { ctx->ev_host_done = nullptr; (void)hipGetLastError(); }
        if (ctx->ev_host_done && hipEventRecord(ctx->ev_host_done, ctx->stream) == hipSuccess) ctx->pending.done = ctx->ev_host_done;
        else (void)hipGetLastError();
    }
    return MKAMD_OK;
}

static int voxelize_lattice_host_end_impl(mkamd_ctx* ctx, float* features, double* features64, double* max_into)
{
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (!ctx->pending.active) return fail(MKAMD_EINVAL, "no host call was begun on this context");
    const size_t out_bytes = ctx->pending.out_bytes;
    const bool mapped_out = ctx->pending.mapped_out;
    const unsigned seq = ctx->pending.seq;
    void* const dout = ctx->pending.dout;
    const hipEvent_t done = ctx->pending.done;
    ctx->pending.active = false;
    if (out_bytes == 0) return MKAMD_OK;                            // no items / no voxels
    if (!features && !features64 && !max_into) { (void)hipStreamSynchronize(ctx->stream); return fail(MKAMD_EINVAL, "features pointer is NULL"); }
    const size_t nvals = out_bytes / 4;
    if (max_into) {
        // calculate_occupancy's contract (occupancy_utils.pyx:61): results[i] = max(results[i], value) -- straight out of the
        // mapped result buffer, once the whole call is done (an early pass could not be repeated: a maximum does not undo)
        const float* src = (const float*)ctx->out_host;
        if (!mapped_out) {
            ctx->f32_stage.resize(nvals);
            HIP_TRY(hipMemcpyAsync(ctx->f32_stage.data(), dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            src = ctx->f32_stage.data();
        }
        HIP_TRY(mapped_out ? wait_for_small_call(ctx->stream, done) : hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < nvals; ++i) {
            const double v = (double)src[i];
            if (v > max_into[i]) max_into[i] = v;
        }
        return collect_async_errors(ctx);
    }
    // wait for "tile kernel done" (a read of host memory per poll; gives up after ~1 ms: the stream wait below covers it)
    bool early = false;
    static const bool no_early = [] { const char* e = std::getenv("MKAMD_NO_EARLY_PASS"); return e && e[0] == '1'; }();   // debugging switch
    if (seq != 0u && ctx->tail_reports && !no_early) {
        const volatile unsigned* fb = ctx->fb_host;
        for (int i = 0; i < 400000 && !early; ++i) early = fb[FB_TILES_DONE] == seq;
        // the result buffer is coherent, uncached host memory the tile kernel wrote before k_tail (same stream) released its
        // system-scope fence and raised the word: nothing of the result may be read ahead of the poll
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!mapped_out) prefault_big_result(features64 ? (void*)features64 : (void*)features, features64 ? out_bytes * 2 : out_bytes);
    if (features64) {
        const float* src = (const float*)ctx->out_host;
        if (!mapped_out) {                                    // big result: through a host staging vector
            ctx->f32_stage.resize(nvals);
            HIP_TRY(hipMemcpyAsync(ctx->f32_stage.data(), dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            src = ctx->f32_stage.data();
        }
        if (early) {
            widen(src, features64, nvals);                          // ... while k_tail runs and the stream signals its completion
            HIP_TRY(wait_for_small_call(ctx->stream, done));
            if (((const volatile unsigned*)ctx->fb_host)[FB_TAIL_WROTE] == seq) widen(src, features64, nvals);     // k_tail changed values (rare): once more
        } else {
            HIP_TRY(mapped_out ? wait_for_small_call(ctx->stream, done) : hipStreamSynchronize(ctx->stream));
            widen(src, features64, nvals);
        }
        st = collect_async_errors(ctx);
        return st;
    }
    if (!mapped_out) HIP_TRY(hipMemcpyAsync(features, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (early) {
        memcpy(features, ctx->out_host, out_bytes);
        HIP_TRY(wait_for_small_call(ctx->stream, done));
        if (((const volatile unsigned*)ctx->fb_host)[FB_TAIL_WROTE] == seq) memcpy(features, ctx->out_host, out_bytes);
        return collect_async_errors(ctx);
    }
    HIP_TRY(mapped_out ? wait_for_small_call(ctx->stream, done) : hipStreamSynchronize(ctx->stream));
    if (mapped_out) memcpy(features, ctx->out_host, out_bytes);
    return collect_async_errors(ctx);
}

static int voxelize_lattice_host_impl(mkamd_ctx* ctx, int32_t B, const float* coords, const int64_t* atom_offsets,
                                      const void* sigmas, int sigmas_are_f64, int32_t C, const double* origins,
                                      const int32_t* nvoxels, double voxelsize, const float* box,
                                      int32_t max_images, float* features, double* features64)
{
    if (ctx && B > 0 && nvoxels && nvoxels[0] > 0 && nvoxels[1] > 0 && nvoxels[2] > 0 && !features && !features64)
        return fail(MKAMD_EINVAL, "atom_offsets/origins/features pointer is NULL");
    const int st = voxelize_lattice_host_begin_impl(ctx, B, coords, atom_offsets, sigmas, sigmas_are_f64, C, origins, nvoxels, voxelsize, box, max_images);
    if (st) return st;
    return voxelize_lattice_host_end_impl(ctx, features, features64, nullptr);
}

int mkamd_voxelize_lattice_host_begin(mkamd_ctx* ctx, int32_t B, const float* coords, const int64_t* atom_offsets,
                                      const void* sigmas, int sigmas_are_f64, int32_t C, const double* origins,
                                      const int32_t* nvoxels, double voxelsize, const float* box, int32_t max_images)
try {
    return voxelize_lattice_host_begin_impl(ctx, B, coords, atom_offsets, sigmas, sigmas_are_f64, C, origins, nvoxels, voxelsize, box, max_images);
} MK_API_CATCH

int mkamd_voxelize_lattice_host_end(mkamd_ctx* ctx, float* features, double* features_f64, uint64_t n_values)
try {
    if (features && features_f64) return fail(MKAMD_EINVAL, "pass ONE result array: float32 or float64");
    // the array the caller brings must hold exactly what the pending call produced: the library writes B*V*C values into it
    if (ctx && ctx->pending.active && ctx->pending.out_bytes / 4 != (size_t)n_values) {
        (void)hipStreamSynchronize(ctx->main_stream);
        ctx->pending.active = false;
        return fail(MKAMD_EINVAL, "the result array does not hold n_items * n_voxels * n_channels values (the pending call is abandoned)");
    }
    return voxelize_lattice_host_end_impl(ctx, features, features_f64, nullptr);
} MK_API_CATCH

int mkamd_voxelize_lattice_host(mkamd_ctx* ctx, int32_t B, const float* coords, const int64_t* atom_offsets,
                                const void* sigmas, int sigmas_are_f64, int32_t C, const double* origins,
                                const int32_t* nvoxels, double voxelsize, const float* box,
                                int32_t max_images, float* features)
try {
    return voxelize_lattice_host_impl(ctx, B, coords, atom_offsets, sigmas, sigmas_are_f64, C, origins, nvoxels, voxelsize, box,
                                      max_images, features, nullptr);
} MK_API_CATCH

int mkamd_voxelize_lattice_host_f64(mkamd_ctx* ctx, int32_t B, const float* coords, const int64_t* atom_offsets,
                                    const void* sigmas, int sigmas_are_f64, int32_t C, const double* origins,
                                    const int32_t* nvoxels, double voxelsize, const float* box,
                                    int32_t max_images, double* features)
try {
    return voxelize_lattice_host_impl(ctx, B, coords, atom_offsets, sigmas, sigmas_are_f64, C, origins, nvoxels, voxelsize, box,
                                      max_images, nullptr, features);
} MK_API_CATCH

// ---------------------------------------------------------------------------------------------
// lattice centres
// ---------------------------------------------------------------------------------------------
int mkamd_grid_centers_dev(mkamd_ctx* ctx, const double* bb_min, const int32_t* nvoxels, double voxelsize,
                           double* d_centers)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (!bb_min || !nvoxels) return fail(MKAMD_EINVAL, "bb_min/nvoxels pointer is NULL");
    const int nv[3] = {nvoxels[0], nvoxels[1], nvoxels[2]};
    std::string err;
    st = run_grid_centers(*ctx, bb_min, nv, voxelsize, d_centers, err);
    if (st && !err.empty()) return fail(st, err);
    return st;
} MK_API_CATCH

int mkamd_grid_centers_host(mkamd_ctx* ctx, const double* bb_min, const int32_t* nvoxels, double voxelsize,
                            double* centers)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (!bb_min || !nvoxels) return fail(MKAMD_EINVAL, "bb_min/nvoxels pointer is NULL");
    if (nvoxels[0] < 0 || nvoxels[1] < 0 || nvoxels[2] < 0) return fail(MKAMD_EINVAL, "nvoxels must be >= 0");
    const long long V = (long long)nvoxels[0] * nvoxels[1] * nvoxels[2];
    if (V == 0) return MKAMD_OK;
    if (!centers) return fail(MKAMD_EINVAL, "centers pointer is NULL");
    void* dc = nullptr;
    if ((st = ctx->ensure(WS_H_CENTERS, (size_t)V * 24, &dc))) return st;
    if ((st = mkamd_grid_centers_dev(ctx, bb_min, nvoxels, voxelsize, (double*)dc))) return st;
    HIP_TRY(hipMemcpyAsync(centers, dc, (size_t)V * 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

extern "C" int mkamd_frames_to_items_dev(mkamd_ctx* ctx, void* hip_stream, const float* d_src, int64_t rows, int64_t src_pitch,
                                         int64_t n_frames, float scale, float* d_dst)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (rows < 0 || n_frames < 0 || src_pitch < n_frames) return fail(MKAMD_EINVAL, "rows / n_frames must be >= 0 and src_pitch >= n_frames");
    if (rows == 0 || n_frames == 0) return MKAMD_OK;
    if (!d_src || !d_dst) return fail(MKAMD_EINVAL, "NULL pointer");
    const long long tiles = ((n_frames + 63) / 64) * ((rows + 63) / 64);
    if (tiles > 0x7ffffff0LL) return fail(MKAMD_EINVAL, "too many tiles (rows x frames / 4096 >= 2^31)");
    hipLaunchKernelGGL(mkamd::k_frames_to_items, dim3((unsigned)(((tiles + 7) / 8) * 8)), dim3(256), 0, (hipStream_t)hip_stream,
                       d_src, (long long)rows, (long long)src_pitch, (long long)n_frames, scale, d_dst);
    HIP_TRY(hipGetLastError());
    return MKAMD_OK;
} MK_API_CATCH

extern "C" int mkamd_copy_to_host(mkamd_ctx* ctx, void* host_dst, const void* device_src, uint64_t bytes)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (bytes == 0) return MKAMD_OK;
    if (!host_dst || !device_src) return fail(MKAMD_EINVAL, "NULL pointer");
    HIP_TRY(hipMemcpyAsync(host_dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

extern "C" int mkamd_copy_dev(mkamd_ctx* ctx, void* device_dst, const void* device_src, uint64_t bytes)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (bytes == 0) return MKAMD_OK;
    if (!device_dst || !device_src) return fail(MKAMD_EINVAL, "NULL pointer");
    HIP_TRY(hipMemcpyAsync(device_dst, device_src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

extern "C" int mkamd_prefault(void* buffer, uint64_t bytes)
try {
    if (!buffer) return MKAMD_OK;
    const int nt = (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
    mkamd::xtc::prefault_output(static_cast<float*>(buffer), (size_t)bytes / sizeof(float), nt);
    return MKAMD_OK;
} catch (...) { return MKAMD_OK; }

// The lattice test of moleculekit_amd/voxeldescriptors.py::_recognise_lattice_numpy in two passes over the array instead
// of a dozen numpy temporaries: axis lengths from the first place z (then y, at stride nz) stops increasing, one common
// positive step, then every centre against fl64(index * step) + centre 0 with the tolerance 1e-9 * max(1, max |c|).
extern "C" int mkamd_lattice_from_centers(const double* c, int64_t V, double* bb_min, int32_t* nvoxels, double* voxelsize)
try {
    return mkamd::lattice_from_centers(c, (long long)V, bb_min, nvoxels, voxelsize) ? 1 : 0;
} catch (...) { return 0; }

// ---------------------------------------------------------------------------------------------
// distance_utils row (include/mkamd_distance.h)
// ---------------------------------------------------------------------------------------------
static int upload(mkamd_ctx* ctx, int slot, const void* src, size_t bytes, void** dst)
try {
    int st = ctx->ensure(slot, bytes, dst, 0);
    if (st) return st;
    if (bytes) HIP_TRY(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
} MK_API_CATCH

int64_t mkamd_dist_count_pairs(int64_t n1, int64_t n2, int selfdist) { return count_pairs(n1, n2, selfdist); }

int mkamd_dist_trajectory_dev(mkamd_ctx* ctx, const float* d_coords, int64_t F, const float* d_box,
                              const uint32_t* d_sel1, int64_t n1, const uint32_t* d_sel2, int64_t n2,
                              const uint32_t* d_chains, int selfdist, int pbc, int squared, float* d_results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    std::string err;
    st = run_dist_trajectory(*ctx, d_coords, F, d_box, d_sel1, n1, d_sel2, n2, d_chains, selfdist, pbc, squared, d_results, err, ctx->dist_avoid);
    if (st && !err.empty()) return fail(st, err);
    return st;
} MK_API_CATCH

int mkamd_dist_trajectory_host(mkamd_ctx* ctx, const float* coords, int64_t N, int64_t F, const float* box,
                               const uint32_t* sel1, int64_t n1, const uint32_t* sel2, int64_t n2,
                               const uint32_t* chains, int selfdist, int pbc, int squared, float* results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (N < 0 || F < 0 || n1 < 0 || n2 < 0) return fail(MKAMD_EINVAL, "negative size");
    const int64_t P = count_pairs(n1, n2, selfdist);
    if (F == 0 || P == 0) return MKAMD_OK;
    if (!coords || !box || !sel1 || !sel2 || !chains || !results) return fail(MKAMD_EINVAL, "NULL pointer");
    for (int64_t i = 0; i < n1; ++i) if (sel1[i] >= (uint64_t)N) return fail(MKAMD_EINVAL, "sel1 index out of range");
    for (int64_t i = 0; i < n2; ++i) if (sel2[i] >= (uint64_t)N) return fail(MKAMD_EINVAL, "sel2 index out of range");
    void *dc, *db, *d1, *d2, *dch, *dout;
    // only the selected atoms' rows go up when they are few (host_pack.h); the selections and chain ids in the packed numbering
    mkamd::PackedAtoms pk;
    pk.collect(sel1, n1); pk.collect(sel2, n2);
    if (!(ctx->dist_avoid & 32) && pk.finish(coords, N, F, ctx->packed_coords)) {
        const std::vector<uint32_t> p1 = pk.remap(sel1, n1), p2 = pk.remap(sel2, n2), pc = pk.gather(chains);
        if ((st = upload(ctx, WS_H_COORDS, ctx->packed_coords.data(), (size_t)pk.size() * 3 * F * 4, &dc))) return st;
        if ((st = upload(ctx, WS_D_SEL1, p1.data(), (size_t)n1 * 4, &d1))) return st;
        if ((st = upload(ctx, WS_D_SEL2, p2.data(), (size_t)n2 * 4, &d2))) return st;
        if ((st = upload(ctx, WS_D_CHAINS, pc.data(), (size_t)pk.size() * 4, &dch))) return st;
        HIP_TRY(hipStreamSynchronize(ctx->stream));                  // (the temporaries above are read by then)
    } else {
        if ((st = upload(ctx, WS_H_COORDS, coords, (size_t)N * 3 * F * 4, &dc))) return st;
        if ((st = upload(ctx, WS_D_SEL1, sel1, (size_t)n1 * 4, &d1))) return st;
        if ((st = upload(ctx, WS_D_SEL2, sel2, (size_t)n2 * 4, &d2))) return st;
        if ((st = upload(ctx, WS_D_CHAINS, chains, (size_t)N * 4, &dch))) return st;
    }
    if ((st = upload(ctx, WS_H_BOX, box, (size_t)3 * F * 4, &db))) return st;
    if ((st = ctx->ensure(WS_H_OUT, (size_t)F * P * 4, &dout, 0))) return st;
    st = mkamd_dist_trajectory_dev(ctx, (const float*)dc, F, (const float*)db, (const uint32_t*)d1, n1, (const uint32_t*)d2, n2,
                                   (const uint32_t*)dch, selfdist, pbc, squared, (float*)dout);
    if (st) return st;
    prefault_big_result(results, (size_t)F * P * 4);
    HIP_TRY(hipMemcpyAsync(results, dout, (size_t)F * P * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

// contacts_trajectory / get_collisions on the device (dist_kernels.h: count -> scan -> fill per chunk of frames)
int mkamd_contacts_trajectory_host(mkamd_ctx* ctx, const float* coords, int64_t N, int64_t F, const float* box,
                                   const uint32_t* sel1, int64_t n1, const uint32_t* sel2, int64_t n2,
                                   const uint32_t* chains, int selfdist, int pbc, float dist_threshold,
                                   int64_t* frame_offsets, const uint32_t** pairs)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (N < 0 || F < 0 || n1 < 0 || n2 < 0) return fail(MKAMD_EINVAL, "negative size");
    if (!frame_offsets || !pairs) return fail(MKAMD_EINVAL, "frame_offsets/pairs pointer is NULL");
    ctx->contacts_host.clear();
    *pairs = nullptr;
    for (int64_t f = 0; f <= F; ++f) frame_offsets[f] = 0;
    const int64_t P = count_pairs(n1, n2, selfdist);
    if (F == 0 || P == 0) return MKAMD_OK;
    if (!coords || !box || !sel1 || !sel2 || !chains) return fail(MKAMD_EINVAL, "NULL pointer");
    for (int64_t i = 0; i < n1; ++i) if (sel1[i] >= (uint64_t)N) return fail(MKAMD_EINVAL, "sel1 index out of range");
    for (int64_t i = 0; i < n2; ++i) if (sel2[i] >= (uint64_t)N) return fail(MKAMD_EINVAL, "sel2 index out of range");
    void *dc, *db, *d1, *d2, *dch;
    mkamd::PackedAtoms pk;                                           // (host_pack.h: only the selected atoms' rows go up when they are few)
    pk.collect(sel1, n1); pk.collect(sel2, n2);
    if (!(ctx->dist_avoid & 32) && pk.finish(coords, N, F, ctx->packed_coords)) {
        const std::vector<uint32_t> p1 = pk.remap(sel1, n1), p2 = pk.remap(sel2, n2), pc = pk.gather(chains);
        if ((st = upload(ctx, WS_H_COORDS, ctx->packed_coords.data(), (size_t)pk.size() * 3 * F * 4, &dc))) return st;
        if ((st = upload(ctx, WS_D_SEL1, p1.data(), (size_t)n1 * 4, &d1))) return st;
        if ((st = upload(ctx, WS_D_SEL2, p2.data(), (size_t)n2 * 4, &d2))) return st;
        if ((st = upload(ctx, WS_D_CHAINS, pc.data(), (size_t)pk.size() * 4, &dch))) return st;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    } else {
        if ((st = upload(ctx, WS_H_COORDS, coords, (size_t)N * 3 * F * 4, &dc))) return st;
        if ((st = upload(ctx, WS_D_SEL1, sel1, (size_t)n1 * 4, &d1))) return st;
        if ((st = upload(ctx, WS_D_SEL2, sel2, (size_t)n2 * 4, &d2))) return st;
        if ((st = upload(ctx, WS_D_CHAINS, chains, (size_t)N * 4, &dch))) return st;
    }
    if ((st = upload(ctx, WS_H_BOX, box, (size_t)3 * F * 4, &db))) return st;
    std::string err;
    static_assert(sizeof(long long) == sizeof(int64_t), "frame offsets are int64");
    st = run_contacts(*ctx, (const float*)dc, (long long)F, (const float*)db, (const unsigned*)d1, (long long)n1, (const unsigned*)d2,
                      (long long)n2, (const unsigned*)dch, selfdist, pbc, dist_threshold, (size_t)256 << 20,
                      (long long*)frame_offsets, HostPairSink<mkamd_ctx>{*ctx, ctx->contacts_host}, err);
    if (st) return err.empty() ? st : fail(st, err);
    if (ctx->contacts_host.empty()) return MKAMD_OK;
    if (pk.on) pk.unpack_in_place(ctx->contacts_host.data(), ctx->contacts_host.size());     // the list names atoms of the caller's array
    *pairs = ctx->contacts_host.data();
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_dist_reduction_host(mkamd_ctx* ctx, const float* coords, int64_t N, int64_t F, const float* box,
                              const int32_t* g1_atoms, const int64_t* g1_off, int64_t ng1, const int32_t* g2_atoms,
                              const int64_t* g2_off, int64_t ng2, const uint32_t* chains1, const uint32_t* chains2,
                              int selfdist, int pairs, int pbc, const float* masses, int reduction1, int reduction2,
                              float* results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (N < 0 || F < 0 || ng1 < 0 || ng2 < 0) return fail(MKAMD_EINVAL, "negative size");
    if (pairs && ng1 != ng2) return fail(MKAMD_EINVAL, "pairs mode needs the same number of groups on both sides");
    const int64_t P = pairs ? ng1 : count_pairs(ng1, ng2, selfdist);
    if (F == 0 || P == 0) return MKAMD_OK;
    if (!coords || !box || !g1_off || !g2_off || !chains1 || !chains2 || !results) return fail(MKAMD_EINVAL, "NULL pointer");
    if ((reduction1 == 1 || reduction2 == 1) && !masses) return fail(MKAMD_EINVAL, "masses are required for the com reduction");
    for (int64_t g = 0; g < ng1; ++g) if (g1_off[g + 1] <= g1_off[g]) return fail(MKAMD_EINVAL, "empty group in groups1");
    for (int64_t g = 0; g < ng2; ++g) if (g2_off[g + 1] <= g2_off[g]) return fail(MKAMD_EINVAL, "empty group in groups2");
    for (int64_t k = 0; k < g1_off[ng1]; ++k) if (g1_atoms[k] < 0 || g1_atoms[k] >= N) return fail(MKAMD_EINVAL, "groups1 atom index out of range");
    for (int64_t k = 0; k < g2_off[ng2]; ++k) if (g2_atoms[k] < 0 || g2_atoms[k] >= N) return fail(MKAMD_EINVAL, "groups2 atom index out of range");
    void *dc, *db, *a1, *o1, *a2, *o2, *c1, *c2, *dm = nullptr, *dout;
    int64_t n_up = N;                                                // atoms of the array that goes up
    mkamd::PackedAtoms pk;                                           // (host_pack.h: only the groups' atoms when they are few)
    pk.collect(g1_atoms, g1_off[ng1]); pk.collect(g2_atoms, g2_off[ng2]);
    if (!(ctx->dist_avoid & 32) && pk.finish(coords, N, F, ctx->packed_coords)) {
        const std::vector<int32_t> p1 = pk.remap(g1_atoms, g1_off[ng1]), p2 = pk.remap(g2_atoms, g2_off[ng2]);
        n_up = pk.size();
        if ((st = upload(ctx, WS_H_COORDS, ctx->packed_coords.data(), (size_t)n_up * 3 * F * 4, &dc))) return st;
        if ((st = upload(ctx, WS_D_G1A, p1.data(), (size_t)g1_off[ng1] * 4, &a1))) return st;
        if ((st = upload(ctx, WS_D_G2A, p2.data(), (size_t)g2_off[ng2] * 4, &a2))) return st;
        if (masses) {
            const std::vector<float> pm = pk.gather(masses);
            if ((st = upload(ctx, WS_D_MASS, pm.data(), (size_t)n_up * 4, &dm))) return st;
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));                  // (the temporaries above are read by then)
    } else {
        if ((st = upload(ctx, WS_H_COORDS, coords, (size_t)N * 3 * F * 4, &dc))) return st;
        if ((st = upload(ctx, WS_D_G1A, g1_atoms, (size_t)g1_off[ng1] * 4, &a1))) return st;
        if ((st = upload(ctx, WS_D_G2A, g2_atoms, (size_t)g2_off[ng2] * 4, &a2))) return st;
        if (masses && (st = upload(ctx, WS_D_MASS, masses, (size_t)N * 4, &dm))) return st;
    }
    if ((st = upload(ctx, WS_H_BOX, box, (size_t)3 * F * 4, &db))) return st;
    if ((st = upload(ctx, WS_D_G1O, g1_off, (size_t)(ng1 + 1) * 8, &o1))) return st;
    if ((st = upload(ctx, WS_D_G2O, g2_off, (size_t)(ng2 + 1) * 8, &o2))) return st;
    if ((st = upload(ctx, WS_D_CHAINS, chains1, (size_t)ng1 * 4, &c1))) return st;
    if ((st = upload(ctx, WS_D_CHAINS2, chains2, (size_t)ng2 * 4, &c2))) return st;
    if ((st = ctx->ensure(WS_H_OUT, (size_t)F * P * 4, &dout, 0))) return st;
    std::string err;
    st = run_dist_reduction(*ctx, (const float*)dc, n_up, F, (const float*)db, (const int*)a1, (const long long*)o1, ng1, g1_off[ng1], (const int*)a2,
                            (const long long*)o2, ng2, (const unsigned*)c1, (const unsigned*)c2, selfdist, pairs, pbc,
                            (const float*)dm, reduction1, reduction2, (float*)dout, err,
                            ctx->reduction_block ? ctx->reduction_block : reduction_block_for((const long long*)g1_off, ng1),
                            ctx->reduction_block == -2 ? 1 : ctx->reduction_block ? -1 : 0);
    if (st) return err.empty() ? st : fail(st, err);
    prefault_big_result(results, (size_t)F * P * 4);
    HIP_TRY(hipMemcpyAsync(results, dout, (size_t)F * P * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

// ---- device-resident forms (round 6): coordinates, selections / groups and results stay on the GPU; asynchronous on the
// context's stream except for the contact list, whose size has to reach the host ----
int mkamd_contacts_trajectory_dev(mkamd_ctx* ctx, const float* d_coords, int64_t F, const float* d_box, const uint32_t* d_sel1,
                                  int64_t n1, const uint32_t* d_sel2, int64_t n2, const uint32_t* d_chains, int selfdist, int pbc,
                                  float dist_threshold, int64_t* frame_offsets, const uint32_t** d_pairs)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (F < 0 || n1 < 0 || n2 < 0) return fail(MKAMD_EINVAL, "negative size");
    if (!frame_offsets || !d_pairs) return fail(MKAMD_EINVAL, "frame_offsets/d_pairs pointer is NULL");
    *d_pairs = nullptr;
    for (int64_t f = 0; f <= F; ++f) frame_offsets[f] = 0;
    if (F == 0 || count_pairs(n1, n2, selfdist) == 0) return MKAMD_OK;
    if (!d_coords || !d_box || !d_sel1 || !d_sel2 || !d_chains) return fail(MKAMD_EINVAL, "NULL pointer");
    std::string err;
    DevicePairSink<mkamd_ctx> sink{*ctx};
    st = run_contacts(*ctx, d_coords, (long long)F, d_box, d_sel1, (long long)n1, d_sel2, (long long)n2, d_chains, selfdist, pbc,
                      dist_threshold, (size_t)256 << 20, (long long*)frame_offsets, sink, err);
    if (st) return err.empty() ? st : fail(st, err);
    if (sink.size) *d_pairs = static_cast<const uint32_t*>(sink.base);
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_dist_reduction_dev(mkamd_ctx* ctx, const float* d_coords, int64_t N, int64_t F, const float* d_box,
                             const int32_t* d_g1_atoms, const int64_t* d_g1_off, int64_t ng1, int64_t n_g1_atoms,
                             const int32_t* d_g2_atoms, const int64_t* d_g2_off, int64_t ng2, const uint32_t* d_chains1,
                             const uint32_t* d_chains2, int selfdist, int pairs, int pbc, const float* d_masses, int reduction1,
                             int reduction2, float* d_results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (N < 0 || F < 0 || ng1 < 0 || ng2 < 0 || n_g1_atoms < 0) return fail(MKAMD_EINVAL, "negative size");
    if (pairs && ng1 != ng2) return fail(MKAMD_EINVAL, "pairs mode needs the same number of groups on both sides");
    const int64_t P = pairs ? ng1 : count_pairs(ng1, ng2, selfdist);
    if (F == 0 || P == 0) return MKAMD_OK;
    if (!d_coords || !d_box || !d_g1_atoms || !d_g1_off || !d_g2_atoms || !d_g2_off || !d_chains1 || !d_chains2 || !d_results)
        return fail(MKAMD_EINVAL, "NULL pointer");
    if ((reduction1 == 1 || reduction2 == 1) && !d_masses) return fail(MKAMD_EINVAL, "masses are required for the com reduction");
    std::string err;
    static_assert(sizeof(long long) == sizeof(int64_t), "group offsets are int64");
    st = run_dist_reduction(*ctx, d_coords, N, F, d_box, (const int*)d_g1_atoms, (const long long*)d_g1_off, ng1, n_g1_atoms,
                            (const int*)d_g2_atoms, (const long long*)d_g2_off, ng2, d_chains1, d_chains2, selfdist, pairs, pbc, d_masses,
                            reduction1, reduction2, d_results, err, ctx->reduction_block,
                            ctx->reduction_block == -2 ? 1 : ctx->reduction_block ? -1 : 0);
    if (st) return err.empty() ? st : fail(st, err);
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_cdist_dev(mkamd_ctx* ctx, const float* d_c1, int64_t n1, const float* d_c2, int64_t n2, int32_t D, float* d_results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (n1 < 0 || n2 < 0 || D < 0) return fail(MKAMD_EINVAL, "negative size");
    if (n1 == 0 || n2 == 0) return MKAMD_OK;
    if (!d_c1 || !d_c2 || !d_results) return fail(MKAMD_EINVAL, "NULL pointer");
    std::string err;
    st = run_cdist(*ctx, d_c1, n1, d_c2, n2, D, d_results, err);
    if (st) return err.empty() ? st : fail(st, err);
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_pdist_dev(mkamd_ctx* ctx, const float* d_c, int64_t n, int32_t D, float* d_results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (n < 0 || D < 0) return fail(MKAMD_EINVAL, "negative size");
    if (n < 2) return MKAMD_OK;
    if (!d_c || !d_results) return fail(MKAMD_EINVAL, "NULL pointer");
    std::string err;
    st = run_pdist(*ctx, d_c, n, D, d_results, err);
    if (st) return err.empty() ? st : fail(st, err);
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_cdist_host(mkamd_ctx* ctx, const float* c1, int64_t n1, const float* c2, int64_t n2, int32_t D, float* results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (n1 < 0 || n2 < 0 || D < 0) return fail(MKAMD_EINVAL, "negative size");
    if (n1 == 0 || n2 == 0) return MKAMD_OK;
    if (!c1 || !c2 || !results) return fail(MKAMD_EINVAL, "NULL pointer");
    void *d1, *d2, *dout;
    if ((st = upload(ctx, WS_H_COORDS, c1, (size_t)n1 * D * 4, &d1))) return st;
    if ((st = upload(ctx, WS_H_SIGMAS, c2, (size_t)n2 * D * 4, &d2))) return st;
    if ((st = ctx->ensure(WS_H_OUT, (size_t)n1 * n2 * 4, &dout, 0))) return st;
    std::string err;
    st = run_cdist(*ctx, (const float*)d1, n1, (const float*)d2, n2, D, (float*)dout, err);
    if (st) return err.empty() ? st : fail(st, err);
    prefault_big_result(results, (size_t)n1 * n2 * 4);
    HIP_TRY(hipMemcpyAsync(results, dout, (size_t)n1 * n2 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

int mkamd_pdist_host(mkamd_ctx* ctx, const float* c, int64_t n, int32_t D, float* results)
try {
    int st = dist_entry(ctx);
    if (st) return st;
    if (n < 0 || D < 0) return fail(MKAMD_EINVAL, "negative size");
    if (n < 2) return MKAMD_OK;
    if (!c || !results) return fail(MKAMD_EINVAL, "NULL pointer");
    void *d1, *dout;
    if ((st = upload(ctx, WS_H_COORDS, c, (size_t)n * D * 4, &d1))) return st;
    if ((st = ctx->ensure(WS_H_OUT, (size_t)n * (n - 1) / 2 * 4, &dout, 0))) return st;
    std::string err;
    st = run_pdist(*ctx, (const float*)d1, n, D, (float*)dout, err);
    if (st) return err.empty() ? st : fail(st, err);
    prefault_big_result(results, (size_t)n * (n - 1) / 2 * 4);
    HIP_TRY(hipMemcpyAsync(results, dout, (size_t)n * (n - 1) / 2 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MKAMD_OK;
} MK_API_CATCH

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// XTC decoding (host side, include/mkamd_xtc.h)
// ---------------------------------------------------------------------------------------------
extern "C" int mkamd_xtc_info(const char* path, int64_t* n_atoms, int64_t* n_frames)
try {
    if (!path || !n_atoms || !n_frames) return fail(MKAMD_EINVAL, "path/n_atoms/n_frames pointer is NULL");
    std::string err;
    int64_t na = 0, nf = 0;
    const int st = mkamd::xtc::info(path, na, nf, err);
    if (st) return fail(MKAMD_EINVAL, err);
    *n_atoms = na; *n_frames = nf;
    return MKAMD_OK;
} MK_API_CATCH

extern "C" int mkamd_xtc_read(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, float* coords, float* box,
                              float* time, int32_t* step, int32_t n_threads)
try {
    if (!path) return fail(MKAMD_EINVAL, "path is NULL");
    if (n_sel < 0 || n_atoms < 0) return fail(MKAMD_EINVAL, "n_sel and n_atoms must be >= 0");
    if (n_sel == 0) return MKAMD_OK;
    if (!coords || !box || !time || !step) return fail(MKAMD_EINVAL, "coords/box/time/step pointer is NULL");
    std::string err;
    const int st = mkamd::xtc::read(path, frames, n_sel, n_atoms, coords, box, time, step, (int)n_threads, err);
    if (st) return fail(MKAMD_EINVAL, err);
    return MKAMD_OK;
} MK_API_CATCH

// ---- XTC decoding on the device (xtc_gpu.h): the host parses the record headers and copies the records' bytes; a lane walks a frame,
// a thread decodes a group ----
// The record headers of the selected frames, read from `base` (= the file's bytes from offset base_off up to limit): descriptors for the
// device decoder (data offsets relative to the lowest selected record), boxes, times, steps, the byte range [lo, hi) of the selection
static int xtc_parse_headers(const uint8_t* base, size_t base_off, size_t limit, size_t file_size, const mkamd::xtc::FrameIndex& idx,
                             const int64_t* frames, int64_t n_sel, int64_t n_atoms, void* desc_out, int64_t* byte_lo, int64_t* byte_hi,
                             float* box, float* time, int32_t* step)
{
    using namespace mkamd::xtc;
    mkamd::XtcFrameDesc* D = (mkamd::XtcFrameDesc*)desc_out;
    size_t lo = (size_t)-1, hi = 0;
    std::vector<size_t> rec((size_t)n_sel), end((size_t)n_sel);
    for (int64_t j = 0; j < n_sel; ++j) {
        const int64_t f = frames ? frames[j] : j;
        if (f < 0 || f >= (int64_t)idx.offs.size()) return fail(MKAMD_EINVAL, "frame index out of range");
        const size_t r = idx.offs[(size_t)f];
        if (r < base_off || r + 92 > limit) return fail(MKAMD_EINVAL, "frame outside the bytes handed over");
        const uint8_t* q = base + (r - base_off);
        if (be_i32(q) != FRAME_MAGIC || (int64_t)be_i32(q + 4) != n_atoms || (int64_t)be_i32(q + 52) != n_atoms) return fail(MKAMD_EINVAL, "corrupt XTC frame");
        step[j] = be_i32(q + 8);
        time[j] = be_f32(q + 12);
        for (int i = 0; i < 9; ++i) box[(size_t)i * (size_t)n_sel + (size_t)j] = be_f32(q + 16 + 4 * i);
        mkamd::XtcFrameDesc d{};
        size_t data = r + 56, e;
        if (n_atoms <= 9) {
            d.raw = 1; d.nbytes = (unsigned)(12 * n_atoms);
            e = data + (size_t)12 * (size_t)n_atoms;
        } else {
            const uint8_t* h = base + (data - base_off);
            const float precision = be_f32(h);
            int32_t hi3[3];
            for (int k = 0; k < 3; ++k) { d.lo[k] = be_i32(h + 4 + 4 * k); hi3[k] = be_i32(h + 16 + 4 * k); }
            d.smallidx = be_i32(h + 28);
            const int32_t nbytes = be_i32(h + 32);
            data += 36;
            if (nbytes < 0 || data + (size_t)nbytes > file_size) return fail(MKAMD_EINVAL, "corrupt XTC frame");
            d.nbytes = (unsigned)nbytes;
            for (int k = 0; k < 3; ++k) d.range[k] = (uint32_t)hi3[k] - (uint32_t)d.lo[k] + 1u;
            if (!d.range[0] || !d.range[1] || !d.range[2]) return fail(MKAMD_EINVAL, "corrupt XTC frame");
            if ((d.range[0] | d.range[1] | d.range[2]) > 0xffffffu) { d.triple_bits = 0; for (int k = 0; k < 3; ++k) d.field_bits[k] = bits_for(d.range[k]); }
            else d.triple_bits = bits_for_product(d.range);
            d.inv_precision = (float)(1.0 / (double)precision);            // as decode_frame
            e = data + (((size_t)nbytes + 3) / 4) * 4;
        }
        rec[(size_t)j] = data; end[(size_t)j] = e;
        lo = std::min(lo, r); hi = std::max(hi, e);
        D[j] = d;
    }
    if (hi > file_size) hi = file_size;
    for (int64_t j = 0; j < n_sel; ++j) D[j].data_off = (unsigned long long)(rec[(size_t)j] - lo);
    *byte_lo = (int64_t)lo; *byte_hi = (int64_t)hi;
    return MKAMD_OK;
}

extern "C" int mkamd_xtc_chunk_desc(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, void* desc_out,
                                    int64_t* byte_lo, int64_t* byte_hi, float* box, float* time, int32_t* step)
try {
    using namespace mkamd::xtc;
    if (!path) return fail(MKAMD_EINVAL, "path is NULL");
    if (n_sel <= 0 || n_atoms < 0) return fail(MKAMD_EINVAL, "n_sel must be > 0 and n_atoms >= 0");
    if (!desc_out || !byte_lo || !byte_hi || !box || !time || !step) return fail(MKAMD_EINVAL, "NULL pointer");
    Mapped m;
    if (!m.open_file(path)) return fail(MKAMD_EINVAL, std::string("cannot open ") + path);
    std::shared_ptr<const FrameIndex> idx;
    if (index_frames_cached(m, idx) != OK) return fail(MKAMD_EINVAL, "not an XTC file (bad magic number)");
    if (idx->natoms != n_atoms) return fail(MKAMD_EINVAL, "atom count of the file differs from the buffers'");
    return xtc_parse_headers(m.p, 0, m.n, m.n, *idx, frames, n_sel, n_atoms, desc_out, byte_lo, byte_hi, box, time, step);
} MK_API_CATCH

// The byte range [lo, hi) of the selected frames' records from the frame index alone (no record is touched): what a streaming reader
// copies FIRST (mkamd_xtc_copy_bytes) -- the headers are then parsed out of that copy (mkamd_xtc_chunk_desc_mem) instead of out of the
// file: a fresh mapping takes a page fault per header, 2 ms per 2 048 frames beside a 4-ms copy (round 6)
extern "C" int mkamd_xtc_byte_range(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, int64_t* byte_lo, int64_t* byte_hi)
try {
    using namespace mkamd::xtc;
    if (!path || !byte_lo || !byte_hi) return fail(MKAMD_EINVAL, "NULL pointer");
    if (n_sel <= 0 || n_atoms < 0) return fail(MKAMD_EINVAL, "n_sel must be > 0 and n_atoms >= 0");
    Mapped m;
    if (!m.open_file(path)) return fail(MKAMD_EINVAL, std::string("cannot open ") + path);
    std::shared_ptr<const FrameIndex> idx;
    if (index_frames_cached(m, idx) != OK) return fail(MKAMD_EINVAL, "not an XTC file (bad magic number)");
    if (idx->natoms != n_atoms) return fail(MKAMD_EINVAL, "atom count of the file differs from the buffers'");
    size_t lo = (size_t)-1, hi = 0;
    for (int64_t j = 0; j < n_sel; ++j) {
        const int64_t f = frames ? frames[j] : j;
        if (f < 0 || f >= (int64_t)idx->offs.size()) return fail(MKAMD_EINVAL, "frame index out of range");
        const size_t r = idx->offs[(size_t)f], e = (size_t)f + 1 < idx->offs.size() ? idx->offs[(size_t)f + 1] : m.n;   // (records lie behind one another)
        lo = std::min(lo, r); hi = std::max(hi, e);
    }
    *byte_lo = (int64_t)lo; *byte_hi = (int64_t)hi;
    return MKAMD_OK;
} MK_API_CATCH

// mkamd_xtc_chunk_desc from a COPY of the file's bytes [bytes_lo, bytes_hi) (host memory, e.g. the pinned staging buffer the records were
// just copied into); same results, the same checks; byte_lo / byte_hi come out as mkamd_xtc_chunk_desc gives them (byte_lo >= bytes_lo)
extern "C" int mkamd_xtc_chunk_desc_mem(const char* path, const int64_t* frames, int64_t n_sel, int64_t n_atoms, const void* bytes, int64_t bytes_lo,
                                        int64_t bytes_hi, void* desc_out, int64_t* byte_lo, int64_t* byte_hi, float* box, float* time, int32_t* step)
try {
    using namespace mkamd::xtc;
    if (!path || !bytes) return fail(MKAMD_EINVAL, "NULL pointer");
    if (n_sel <= 0 || n_atoms < 0) return fail(MKAMD_EINVAL, "n_sel must be > 0 and n_atoms >= 0");
    if (!desc_out || !byte_lo || !byte_hi || !box || !time || !step) return fail(MKAMD_EINVAL, "NULL pointer");
    Mapped m;
    if (!m.open_file(path)) return fail(MKAMD_EINVAL, std::string("cannot open ") + path);
    std::shared_ptr<const FrameIndex> idx;
    if (index_frames_cached(m, idx) != OK) return fail(MKAMD_EINVAL, "not an XTC file (bad magic number)");
    if (idx->natoms != n_atoms) return fail(MKAMD_EINVAL, "atom count of the file differs from the buffers'");
    if (bytes_lo < 0 || bytes_hi < bytes_lo || (size_t)bytes_hi > m.n) return fail(MKAMD_EINVAL, "byte range outside the file");
    return xtc_parse_headers(static_cast<const uint8_t*>(bytes), (size_t)bytes_lo, (size_t)bytes_hi, m.n, *idx, frames, n_sel, n_atoms, desc_out, byte_lo, byte_hi,
                             box, time, step);
} MK_API_CATCH

// Bytes [lo, hi) of the file into dst (pinned staging memory), by threads that pread() their slices: the page cache is copied straight
// into the destination.  (Until round 6 the threads copied out of a fresh mapping: 73 000 page faults per 300 MB chunk and an munmap of
// every touched page afterwards.)
extern "C" int mkamd_xtc_copy_bytes(const char* path, int64_t lo, int64_t hi, void* dst, int32_t n_threads)
try {
    if (!path || !dst) return fail(MKAMD_EINVAL, "NULL pointer");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(MKAMD_EINVAL, std::string("cannot open ") + path);
    struct stat stt;
    if (fstat(fd, &stt) != 0) { ::close(fd); return fail(MKAMD_EINVAL, std::string("cannot stat ") + path); }
    if (lo < 0 || hi < lo || (size_t)hi > (size_t)stt.st_size) { ::close(fd); return fail(MKAMD_EINVAL, "byte range outside the file"); }
    const size_t n = (size_t)(hi - lo);
    int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    nt = (int)std::min<size_t>((size_t)nt, std::max<size_t>(n >> 22, 1));         // >= 4 MB per thread
    std::atomic<int> bad{0};
    auto part = [&](int t) {
        size_t a = n / (size_t)nt * (size_t)t;
        const size_t b = t == nt - 1 ? n : n / (size_t)nt * (size_t)(t + 1);
        while (a < b) {
            const ssize_t got = ::pread(fd, (char*)dst + a, std::min<size_t>(b - a, (size_t)8 << 20), (off_t)((size_t)lo + a));
            if (got <= 0) { if (got < 0 && errno == EINTR) continue; bad.store(1); return; }
            a += (size_t)got;
        }
    };
    if (nt == 1) part(0);
    else { std::vector<std::thread> pool; for (int t = 0; t < nt; ++t) pool.emplace_back(part, t); for (auto& th : pool) th.join(); }
    ::close(fd);
    if (bad.load()) return fail(MKAMD_EINVAL, std::string("short read from ") + path);
    return MKAMD_OK;
} MK_API_CATCH

static uint64_t xtc_work_groups_offset(int64_t n_frames) { return ((uint64_t)n_frames * 4u + 255u) & ~(uint64_t)255u; }

extern "C" uint64_t mkamd_xtc_decode_work_bytes(int64_t n_frames, int64_t n_atoms)
{
    if (n_frames <= 0 || n_atoms < 0) return 0;
    return xtc_work_groups_offset(n_frames) + (uint64_t)n_frames * (uint64_t)(n_atoms + mkamd::XS_SPEC) * sizeof(mkamd::XtcGroup);
}

extern "C" int mkamd_xtc_decode_dev(mkamd_ctx* ctx, void* hip_stream, const void* d_bytes, const void* d_desc, int64_t n_frames,
                                    int64_t n_atoms, float scale, float* d_xyz, int32_t* d_status, void* d_work, uint64_t work_bytes)
try {
    int st = check_ctx(ctx, true);
    if (st) return st;
    if (n_frames < 0 || n_atoms < 0) return fail(MKAMD_EINVAL, "n_frames / n_atoms must be >= 0");
    if (n_frames == 0) return MKAMD_OK;
    if (!d_bytes || !d_desc || !d_xyz || !d_status || !d_work) return fail(MKAMD_EINVAL, "NULL pointer");
    if (work_bytes < mkamd_xtc_decode_work_bytes(n_frames, n_atoms))
        return fail(MKAMD_EINVAL, "work buffer smaller than mkamd_xtc_decode_work_bytes(n_frames, n_atoms)");
    if (((uintptr_t)d_work & 7u) || ((uintptr_t)d_bytes & 3u)) return fail(MKAMD_EINVAL, "d_work must be 8-byte, d_bytes 4-byte aligned");
    hipStream_t s = (hipStream_t)hip_stream;
    int* ngroups = static_cast<int*>(d_work);
    mkamd::XtcGroup* groups = reinterpret_cast<mkamd::XtcGroup*>(static_cast<unsigned char*>(d_work) + xtc_work_groups_offset(n_frames));
    const unsigned char* bytes = static_cast<const unsigned char*>(d_bytes);
    const mkamd::XtcFrameDesc* desc = static_cast<const mkamd::XtcFrameDesc*>(d_desc);
    hipLaunchKernelGGL(mkamd::k_xtc_scan, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, s, bytes, desc, (long long)n_frames,
                       (long long)n_atoms, scale, d_xyz, groups, ngroups, d_status);
    HIP_TRY(hipGetLastError());
    if (n_atoms >= (1ll << 21)) return MKAMD_OK;                     // (every frame got status 2; nothing to expand)
    const int64_t bpf = (n_atoms + 255) / 256;
    if (bpf == 0) return MKAMD_OK;
    const int64_t per_launch = std::max<int64_t>(1, (int64_t)0x7fffffff / bpf);
    for (int64_t f0 = 0; f0 < n_frames; f0 += per_launch) {
        const int64_t nf = std::min(per_launch, n_frames - f0);
        hipLaunchKernelGGL(mkamd::k_xtc_expand, dim3((unsigned)(nf * bpf)), dim3(256), 0, s, bytes, desc, (long long)f0, (long long)n_atoms,
                           scale, d_xyz, groups, ngroups, (int)bpf);
        HIP_TRY(hipGetLastError());
    }
    return MKAMD_OK;
} MK_API_CATCH

#ifdef MK_PHASE_TIMERS
// diagnostics build only (mk_diagnostics.h; tools/phase_timers.py): read and clear the per-phase cycle sums of the tile kernel
extern "C" int mkamd_debug_phase_cycles(unsigned long long* out8)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mkamd::g_phase_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[8] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(mkamd::g_phase_cycles), z, sizeof z) == hipSuccess ? 0 : 1;
}
#endif
