// tests/emu/emu_capi.cpp -- TEST INFRASTRUCTURE ONLY.
// Drives the product's launch sequences (moleculekit_amd/csrc/pipeline.h) and kernels
// (kernels.h) through the host SIMT emulation of emu_device.h.  Same planning code, same kernel
// source, host memory instead of HBM.  Built into tests/emu/libmkamd_emu.so by tests/emu_build.py.
#include "emu_device.h"
#include "../../moleculekit_amd/csrc/pipeline.h"
#include "../../moleculekit_amd/csrc/dist_pipeline.h"
#include "../../moleculekit_amd/csrc/xtc_gpu.h"
#include "../../moleculekit_amd/csrc/host_pack.h"

#include <string>
#include <algorithm>
#include <vector>

using namespace mkamd;

namespace {

struct EmuBackend {
    void* bufs[WS_NSLOTS] = {};
    size_t caps[WS_NSLOTS] = {};
    unsigned feedback[FEEDBACK_WORDS] = {};
    CounterState counters[2];
    int fills = 0;                      // memsets of the cell counters (the tests count them)
    CounterState& counter_state(int set) { return counters[set & 1]; }
    void note_error_flag_mirrored(bool) {}
    void note_tail_reports(bool) {}
    void note_dist_kernel(const char* name);                // (kept for the tests: which kernels run_dist_trajectory chose)
    void note_dist_kernel_append(const char*) {}
    int compute_units() const { return 256; }
    const volatile unsigned* feedback_host() const { return feedback; }
    unsigned* feedback_dev() { return feedback; }
    ~EmuBackend() { for (void* p : bufs) free(p); }
    int ensure(int slot, size_t bytes, void** ptr, int = 0)
    {
        if (bytes == 0) bytes = 16;
        if (caps[slot] < bytes) {
            free(bufs[slot]);
            bufs[slot] = malloc(bytes);
            memset(bufs[slot], 0xCD, bytes);          // poison: catch reads of unwritten workspace
            caps[slot] = bytes;
        }
        *ptr = bufs[slot];
        return 0;
    }
    int grow_keep(int slot, size_t bytes, size_t keep_bytes, void** ptr)       // (as mkamd_ctx::grow_keep: exact sizes here, so that every chunk grows)
    {
        if (bytes == 0) bytes = 16;
        if (caps[slot] < bytes) {
            void* fresh = malloc(bytes);
            memset(fresh, 0xCD, bytes);
            if (bufs[slot] && keep_bytes) memcpy(fresh, bufs[slot], keep_bytes);
            free(bufs[slot]);
            bufs[slot] = fresh; caps[slot] = bytes;
        }
        *ptr = bufs[slot];
        return 0;
    }
    int fill(void* p, int byte, size_t bytes) { memset(p, byte, bytes); ++fills; return 0; }
    int to_host(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
    int to_device(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
    template <class... KA, class... A>
    int launch(void (*kernel)(KA...), dim3 grid, dim3 block, A... args)
    {
        emu::launch(kernel, grid, block, args...);
        return 0;
    }
    void hot_begin(int = 0, int = 0, int = 0) {}
    void hot_end() {}
    int acquire_set(bool) { return 0; }
    bool set_is_pipelined(int) const { return false; }
    bool pipelining_possible() const { return false; }
    void prepass_done(int) {}
    void tile_done(int) {}
};

thread_local std::string g_err;

}  // namespace

extern "C" {

const char* emu_last_error(void) { return g_err.c_str(); }
static std::string g_last_dist_kernel;
const char* emu_last_dist_kernel(void) { return g_last_dist_kernel.c_str(); }
}
void EmuBackend::note_dist_kernel(const char* name) { g_last_dist_kernel = name; }
extern "C" {

// mirrors mkamd_voxelize_lattice_host (pointers are host pointers; features is poisoned first)
int emu_voxelize_lattice(int B, const float* coords, const long long* atom_offsets, const void* sigmas,
                         int sigmas_f64, int C, const double* origins, const int* nvox, double voxelsize,
                         const float* box, int max_images, int tile_k, int force_general, const double* affine, float* features,
                         int* err_flag_out, int lds_tier, unsigned* feedback_io /* NTIER+1: in = previous call's, out = this call's */,
                         int prepass_mode, int tile_team, int fine_cells, int repeat /* calls on ONE backend */, int* fills_out, int tile_items,
                         double value_tol, int direct, int cell_cap, unsigned spill_cap, unsigned* direct_words_out /* [DIRECT_WORDS] of the last call */)
{
    EmuBackend be;
    void* eflag = nullptr;
    be.ensure(WS_ERR, sizeof(int), &eflag);
    *(int*)eflag = 0;
    LatticeProblem P;
    P.B = B; P.total_atoms = B > 0 ? atom_offsets[B] : 0; P.C = C; P.sigmas_f64 = sigmas_f64;
    P.nvox[0] = nvox[0]; P.nvox[1] = nvox[1]; P.nvox[2] = nvox[2];
    P.voxelsize = voxelsize; P.pbc = box ? 1 : 0; P.tile_k = tile_k; P.force_general = force_general; P.lds_tier = lds_tier; P.prepass_mode = prepass_mode; P.tile_team = tile_team; P.fine_cells = fine_cells; P.tile_items = tile_items; P.value_tol = value_tol; P.direct = direct; P.cell_cap = cell_cap; P.spill_cap = spill_cap;
    if (feedback_io) for (int i = 0; i <= NTIER; ++i) be.feedback[i] = feedback_io[i];
    if (box && max_images <= 0) {
        max_images = max_images_from_boxes(box, B, nvox, voxelsize, g_err);
        if (max_images < 0) return ST_EBOX;
    }
    P.max_images = box ? max_images : 1;
    P.coords = coords; P.atom_offsets = atom_offsets; P.sigmas = sigmas; P.origins = origins;
    P.box = box; P.affine = affine; P.out = features;
    const size_t nout = (size_t)B * nvox[0] * nvox[1] * nvox[2] * C;
    int st = 0;
    for (int r = 0; r < (repeat > 0 ? repeat : 1) && !st; ++r) {     // later rounds run on the counters the earlier ones left
        for (size_t i = 0; i < nout; ++i) features[i] = -123.0f;
        st = run_lattice(be, P, g_err);
    }
    if (fills_out) *fills_out = be.fills;
    if (direct_words_out) {
        // what the direct pass of the LAST call reported (all ones when the call had none)
        const bool had = be.bufs[WS_DIRECT_COUNT] != nullptr;
        for (int i = 0; i < DIRECT_WORDS; ++i) direct_words_out[i] = had ? ((const unsigned*)be.bufs[WS_DIRECT_COUNT])[i] : 0xffffffffu;
    }
    if (err_flag_out) *err_flag_out = *(int*)be.bufs[WS_ERR];
    if (feedback_io) for (int i = 0; i <= NTIER; ++i) feedback_io[i] = be.feedback[i];
    return st;
}

// mirrors mkamd_topology_create_host + mkamd_voxelize_lattice_topo_dev (capi.hip): the topology of ONE molecule (sigmas [n, C]) built
// by the product's kernels, then B sets of coordinates of it through run_lattice with P.topo.  *wide_out: the handle's wide flag.
int emu_voxelize_lattice_topo(int B, const float* coords, const long long* atom_offsets, const void* sigmas, int sigmas_f64, long long n, int C,
                              const double* origins, const int* nvox, double voxelsize, const float* box, int max_images, int tile_k,
                              const double* affine, float* features, int* err_flag_out, int* wide_out, int repeat,
                              int exact_redo /* 0: the hits of wide atoms go to k_exact_redo (the default), -1: recomputed inside k_tail */)
{
    EmuBackend be;
    void* eflag = nullptr;
    be.ensure(WS_ERR, sizeof(int), &eflag);
    *(int*)eflag = 0;
    const int G = ceil_div(C, CHG);
    std::vector<uint2> cw((size_t)n * G);
    std::vector<unsigned> ids((size_t)n * G, 0xCDCDCDCDu), table(CLS_TABLE_WORDS, 0xCDCDCDCDu);
    int flags2[2] = {0, 0};
    std::vector<unsigned> wide_list((size_t)n, 0xCDCDCDCDu);
    int st = run_topology_build(be, sigmas, sigmas_f64, n, C, voxelsize, cw.data(), ids.data(), table.data(), flags2, wide_list.data(), g_err);
    if (st) return st;
    const int flags = flags2[0];
    std::sort(wide_list.begin(), wide_list.begin() + flags2[1]);     // (as mkamd_topology_create_dev does)
    TopologyDev T;
    T.n = n; T.C = C; T.G = G; T.sigmas_f64 = sigmas_f64; T.voxelsize = voxelsize; T.ids = ids.data(); T.cw = cw.data(); T.sigmas = sigmas;
    T.table = table.data(); T.overflow = table[CLS_OVERFLOW] != CLS_EMPTY; T.wide = (flags & 1) != 0;
    T.wide_list = wide_list.data(); T.n_wide = (unsigned)flags2[1];
    if (wide_out) *wide_out = T.wide ? 1 : 0;
    LatticeProblem P;
    P.B = B; P.total_atoms = B > 0 ? atom_offsets[B] : 0; P.C = C; P.sigmas_f64 = sigmas_f64;
    P.nvox[0] = nvox[0]; P.nvox[1] = nvox[1]; P.nvox[2] = nvox[2];
    P.voxelsize = voxelsize; P.pbc = box ? 1 : 0; P.tile_k = tile_k;
    if (box && max_images <= 0) {
        max_images = max_images_from_boxes(box, B, nvox, voxelsize, g_err);
        if (max_images < 0) return ST_EBOX;
    }
    P.max_images = box ? max_images : 1;
    P.coords = coords; P.atom_offsets = atom_offsets; P.sigmas = nullptr; P.origins = origins;
    P.box = box; P.affine = affine; P.out = features; P.topo = &T; P.exact_redo_list = exact_redo;
    const size_t nout = (size_t)B * nvox[0] * nvox[1] * nvox[2] * C;
    for (int r = 0; r < (repeat > 0 ? repeat : 1) && !st; ++r) {
        for (size_t i = 0; i < nout; ++i) features[i] = -123.0f;
        st = run_lattice(be, P, g_err);
    }
    if (err_flag_out) *err_flag_out = *(int*)be.bufs[WS_ERR];
    return st;
}

int emu_occupancy_centers(const double* centers, long long V, const float* coords, long long N,
                          const void* sigmas, int sigmas_f64, int C, const double* box, float* features)
{
    EmuBackend be;
    for (long long i = 0; i < V * C; ++i) features[i] = -123.0f;
    return run_centers(be, centers, V, coords, N, sigmas, sigmas_f64, C, box, features, g_err);
}

// mirrors mkamd_calculate_occupancy (capi.hip): the same lattice recogniser and the same routing rule (pipeline.h), the
// emulated kernels behind them.  inject_lattice_status != 0 stands in for a failure inside the lattice path (a HIP error,
// an allocation failure): the tests check that it comes back instead of being retried on the pairwise kernel.
// *route_out: 1 = tiled lattice kernels served the call, 2 = pairwise kernel, 0 = neither (an error came back).
int emu_calculate_occupancy(const double* centers, long long V, const float* coords, long long N, const double* sigmas,
                            int C, double* results, int inject_lattice_status, int* route_out)
{
    if (V <= 0 || N <= 0) return ST_OK;
    std::vector<float> tmp((size_t)V * C, -123.0f);
    double bb_min[3], vs = 0.0;
    int nv[3];
    int route = 0;
    const bool is_lattice = lattice_from_centers(centers, V, bb_min, nv, &vs);
    const int st = route_calculate_occupancy(is_lattice,
        [&] {
            if (inject_lattice_status) return inject_lattice_status;
            const long long offs[2] = {0, N};
            const int r = emu_voxelize_lattice(1, coords, offs, sigmas, 1, C, bb_min, nv, vs, nullptr, 0, 0, 0, nullptr, tmp.data(),
                                               nullptr, -1, nullptr, -1, -1, 0, 1, nullptr, -1, 0.0, 0, 0, 0u, nullptr);
            if (!r) route = 1;
            return r;
        },
        [&] {
            const int r = emu_occupancy_centers(centers, V, coords, N, sigmas, 1, C, nullptr, tmp.data());
            if (!r) route = 2;
            return r;
        });
    if (route_out) *route_out = st ? 0 : route;
    if (st) return st;
    for (size_t i = 0; i < tmp.size(); ++i) {
        const double v = (double)tmp[i];
        if (v > results[i]) results[i] = v;          // occupancy_utils.pyx:61
    }
    return ST_OK;
}

int emu_choose_tier(int forced, const unsigned* feedback) { return choose_tier(forced, feedback); }

int emu_grid_centers(const double* bb_min, const int* nvox, double voxelsize, double* centers)
{
    EmuBackend be;
    return run_grid_centers(be, bb_min, nvox, voxelsize, centers, g_err);
}

int emu_exclusive_scan(const unsigned* in, long long n, unsigned* out /* n+1 */)
{
    EmuBackend be;
    std::vector<unsigned> counts(in, in + n);                // the scan clears its input (the counters are self-cleaning)
    counts.push_back(0u);
    return run_scan(be, counts.data(), (size_t)n, out);
}

// planning only: lets the tests inspect the GridDesc the product would use
int emu_plan(int B, long long total_atoms, int C, const int* nvox, double voxelsize, int pbc, int max_images,
             int tile_k, int* out_ints /* 16 */)
{
    LatticeProblem P;
    P.B = B; P.total_atoms = total_atoms; P.C = C; P.nvox[0] = nvox[0]; P.nvox[1] = nvox[1]; P.nvox[2] = nvox[2];
    P.voxelsize = voxelsize; P.pbc = pbc; P.max_images = max_images; P.tile_k = tile_k;
    GridDesc g;
    const int st = plan_lattice(P, g, g_err);
    if (st) return st;
    const int v[16] = {g.K, g.tnx, g.tny, g.tnz, g.ntiles, g.cs, g.h, g.ncx, g.ncy, g.ncz, g.ncell, g.rint, g.G, (int)g.M, 0, 0};
    memcpy(out_ints, v, sizeof v);
    return 0;
}

// ---- distance_utils row: same launch sequences on host memory ----
int emu_dist_trajectory(const float* coords, long long F, const float* box, const unsigned* sel1, long long n1,
                        const unsigned* sel2, long long n2, const unsigned* chains, int selfdist, int pbc, int squared,
                        float* out, int avoid /* DIST_AVOID_* bits: kernels not to take */)
{
    EmuBackend be;
    return run_dist_trajectory(be, coords, F, box, sel1, n1, sel2, n2, chains, selfdist, pbc, squared, out, g_err, avoid);
}

int emu_dist_reduction(const float* coords, long long F, const float* box, const int* g1a, const long long* g1o, long long ng1,
                       const int* g2a, const long long* g2o, long long ng2, const unsigned* ch1, const unsigned* ch2,
                       int selfdist, int pairs, int pbc, const float* masses, int r1, int r2, float* out, long long n_atoms,
                       int closest_block /* 0 choose, 4 / 8, -1 the generic kernel, -2 the few-frame kernel (as mkamd_ctx_set_reduction_block) */)
{
    EmuBackend be;
    return run_dist_reduction(be, coords, n_atoms, F, box, g1a, g1o, ng1, g1o[ng1], g2a, g2o, ng2, ch1, ch2, selfdist, pairs, pbc, masses, r1, r2, out,
                              g_err, closest_block, closest_block == -2 ? 1 : closest_block ? -1 : 0);
}

// contacts_trajectory: frame_offsets [F+1]; pairs_out (capacity 2*cap uint32) gets the (a, b) pairs; returns the count in *n_out
int emu_contacts(const float* coords, long long F, const float* box, const unsigned* sel1, long long n1, const unsigned* sel2,
                 long long n2, const unsigned* chains, int selfdist, int pbc, float threshold, long long budget_bytes,
                 long long* frame_offsets, unsigned* pairs_out, long long cap, long long* n_out, int device_sink, int avoid /* CONTACTS_AVOID_* */)
{
    EmuBackend be;
    if (device_sink) {                                               // the "_dev" entry point's sink: one buffer that grows by copying
        DevicePairSink<EmuBackend> sink{be};
        const int st = run_contacts(be, coords, F, box, sel1, n1, sel2, n2, chains, selfdist, pbc, threshold, (size_t)budget_bytes,
                                    frame_offsets, sink, g_err, avoid);
        *n_out = (long long)sink.size;
        if (!st && (long long)sink.size <= cap && sink.size) memcpy(pairs_out, sink.base, sink.size * 2 * sizeof(unsigned));
        return st;
    }
    std::vector<unsigned> pairs;
    const int st = run_contacts(be, coords, F, box, sel1, n1, sel2, n2, chains, selfdist, pbc, threshold, (size_t)budget_bytes,
                                frame_offsets, HostPairSink<EmuBackend>{be, pairs}, g_err, avoid);
    *n_out = (long long)(pairs.size() / 2);
    if (!st && (long long)(pairs.size() / 2) <= cap) memcpy(pairs_out, pairs.data(), pairs.size() * sizeof(unsigned));
    return st;
}

// csrc/host_pack.h as the host entry points use it: returns 1 when the call would upload the packed rows (then out_coords [M,3,F], out_uniq [M],
// out_remap [n] = the selection in the packed numbering are filled), 0 when the array goes up as it is (out_uniq / *M still say which atoms)
int emu_pack_atoms(const float* coords, long long N, long long F, const unsigned* sel, long long n, float* out_coords, unsigned* out_uniq,
                   unsigned* out_remap, long long* M)
{
    PackedAtoms pk;
    pk.collect(sel, n);
    std::vector<float> buf;
    const bool on = pk.finish(coords, N, F, buf);
    *M = pk.size();
    for (long long k = 0; k < pk.size(); ++k) out_uniq[k] = pk.uniq[(size_t)k];
    if (!on) return 0;
    memcpy(out_coords, buf.data(), (size_t)pk.size() * 3 * (size_t)F * sizeof(float));
    const std::vector<unsigned> r = pk.remap(sel, n);
    for (long long i = 0; i < n; ++i) out_remap[i] = r[(size_t)i];
    return 1;
}

void emu_unpack_atoms(const unsigned* uniq, long long M, unsigned* atoms, long long n)
{
    PackedAtoms pk;
    pk.uniq.assign(uniq, uniq + M);
    pk.unpack_in_place(atoms, (size_t)n);
}

int emu_cdist(const float* c1, long long n1, const float* c2, long long n2, int D, float* out)
{
    EmuBackend be;
    return run_cdist(be, c1, n1, c2, n2, D, out, g_err);
}

int emu_pdist(const float* c, long long n, int D, float* out)
{
    EmuBackend be;
    return run_pdist(be, c, n, D, out, g_err);
}

// mirrors mkamd_xtc_decode_dev (csrc/capi.hip): the two passes of xtc_gpu.h on host memory; the work buffer is poisoned first
int emu_xtc_decode(const unsigned char* bytes, const void* desc, long long n_frames, long long n_atoms, float scale, float* xyz, int* status)
{
    if (n_frames <= 0) return 0;
    std::vector<int> ngroups((size_t)n_frames, -1);
    std::vector<XtcGroup> groups((size_t)n_frames * (size_t)(n_atoms + XS_SPEC), XtcGroup{0xCDCDCDCDu, 0xCDCDCDCDu});
    const XtcFrameDesc* D = static_cast<const XtcFrameDesc*>(desc);
    emu::launch(k_xtc_scan, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), bytes, D, n_frames, n_atoms, scale, xyz, groups.data(), ngroups.data(), status);
    if (n_atoms >= (1ll << 21)) return 0;
    const long long bpf = (n_atoms + 255) / 256;
    if (bpf == 0) return 0;
    emu::launch(k_xtc_expand, dim3((unsigned)(n_frames * bpf)), dim3(256), bytes, D, 0ll, n_atoms, scale, xyz, (const XtcGroup*)groups.data(),
                (const int*)ngroups.data(), (int)bpf);
    return 0;
}

// which kernel a rectangular dist_trajectory call takes (dist_pipeline.h): second atoms per lane of the row kernel, 0 = the tile kernel
int emu_dist_rows_jpl(long long n1, long long n2, long long F) { return dist_rows_jpl(n1, n2, F); }

}  // extern "C"
