// gdyn_capi.hip -- host side of libgdyn: the C-ABI of include/gdyn.h over the HIP kernels.
//
// One handle = one HIP device + one stream + R replicas of an N-bead system resident in HBM.
// gd_run() enqueues   [list build] + K x k_step   segments with no host round trip; the
// Verlet skin is VERIFIED on the device (every step checks each bead's displacement since
// the build) and a violated chunk is rolled back and re-run with a shorter interval, so
// the fixed build cadence never changes results (only the summation order of pair terms).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/gdyn.h"
#include <hip/hip_fp16.h>

#include "gdyn_types.h"
#include "gdyn_once.hpp"
#ifdef GD_DEV
#include "gdyn_dev.h"
#endif

// ------------------------------------------------------------------ errors

static thread_local char g_err[1024];
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(call)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return fail(GD_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define GDCHK(call)              \
    do {                         \
        int rc_ = (call);        \
        if (rc_ != GD_OK) return rc_; \
    } while (0)

extern "C" const char *gd_last_error(void) { return g_err; }

// Experiment hooks (environment variables, debug prints, the kernel micro-benchmark) exist in developer builds only
// (make dev -> libgdyn_dev.so, -DGD_DEV); the product library reads no environment variable.
#ifdef GD_DEV
static const char *dev_env(const char *name) { return getenv(name); }
#else
static const char *dev_env(const char *) { return nullptr; }
#endif
extern "C" const char *gd_backend_name(void) { return "hip"; }

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t resize(size_t count, bool zero = true)
    {
        if (count == n && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        n = count;
        if (zero) { e = hipMemset(p, 0, count * sizeof(T)); if (e == hipSuccess) e = hipDeviceSynchronize(); }
        return e;
    }
};

// ------------------------------------------------------------------ system

struct Bond { uint32_t i, j; int type; };
struct BendRange { uint32_t start, end; double energy; int per_bead; };
struct PointSource { int kind; double k, b, p[3]; std::vector<uint8_t> mask; };
struct DynSet { bool used = false; gd_bond_params p; std::vector<uint32_t> pairs; };

struct gd_system {
    uint32_t N = 0, R = 0, Np = 0, nblk = 0;
    int device = 0, box_kind = 0;
    double box[3] = {0, 0, 0};
    hipStream_t stream = nullptr;

    // host model
    std::vector<double> a, b, mob, bend;
    bool has_pair = false; gd_pair_softcore pair{};
    std::vector<gd_bond_params> btypes; std::vector<int> bterm;
    std::vector<Bond> bonds;
    DynSet dyn[4];
    std::vector<BendRange> bends;
    std::vector<PointSource> psrc;
    bool has_wall = false; gd_wall wall{};
    bool has_scaling = false; double bs_init = 1, bs_tau = 1, bo_init = 1, bo_tau = 1;
    std::vector<DevCtx> hctx;      // host mirror of the device context

    bool topo_dirty = true, list_valid = false, ctx_dirty = true;
    bool has_bend = false, has_bonds = false;
    uint32_t WB = 0, W = 0, ncell_cap = 0;
    uint32_t list_W = 0;           // row width the list in use was built with (W may change for the next build)
    uint32_t tiled_off = 0;        // why the tiled path is off: 0 on, 1 a tile overflowed (dense transient: retried later), 2 by design
    uint32_t tiled_wait = 0, tiled_backoff = 8;     // accepted chunks since / until the next retry of the tiled path
    int pcur = 0, ccur = 0;
    uint32_t kernel_path = 0;      // 0 auto, 1 generic, 2 tiled
    bool packed_ab = false, tiled_ok = true, list_tiled = false;
    bool w_packed = false;         // pos.w of the current positions holds the packed (a,b) factors (set by a build, reset by gd_set_positions / a new topology)
    bool has_inner = false; gd_inner_sphere inner{};
    bool has_softcore_bonds = false;
    bool bonds_premixed = false;   // every bond parameter record is unmixed (AB mixing resolved per bond by finalize_topology)
    bool bonds_all_scaled = false; // every bond parameter record has scale_by_bond_scale set
    uint32_t sw_n = 0; double sw_eps = 0, sw_decay = 1, sw_cut = 0;     // droplet attraction (gd_set_pair_softwell)
    DevBuf<unsigned> sw_targets; DevBuf<double> sw_esum;
    float *h_stage = nullptr;      // pinned host staging for snapshot downloads (R*N*3 floats)
    char *h_chunk = nullptr;       // pinned host block for the per-chunk readback (flags, contexts, list counts): copies into pageable
                                   // memory are staged by the runtime and cost ~20 us each
    uint32_t tile_hold = 0;        // chunks to stay in the larger tile class after an overflow
    uint32_t last_need_t = 0;      // largest tile of the last build that reported one (entries)
    uint32_t list_tile_cap = 0;    // tile capacity the current list was built with (fixes its entry encoding and LDS need)
    uint32_t cpb = 1, tile_cap = 3312;

    // tuning / cadence
    double skin = 0.75;   // relative to the pair cutoff; 0.65..0.8 are within 3% of each other on S-genome-30k, smaller tiles leave more LDS margin
    // Width by tile class (class_skin): the wider list (0.9: a third fewer builds) is used wherever its largest tile still fits the
    // three-block LDS class -- a rule on the state, not on measured times, so a given state always selects the same width.
    bool skin_fixed = false;       // the caller chose a skin (gd_tuning.skin > 0): keep it
    uint32_t skin_streak = 0, skin_hold = 0;
    double skin_next = 0;          // width the next list build moves to (the list in use serves out its interval; 0: none pending)
    double skin_dense_from = 0;    // > 0: the width was narrowed because a build met a dense state (dense_guard); the width to return to
    uint32_t last_need_w = 0;      // longest list (entries, padded) the last build reported
    uint32_t ncell_seen = 0;       // largest cell grid of the last build that reported one (sizes k_scan's launch)
    uint32_t dense_budget = 0;     // dense_guard: longest list (entries) the memory budget admits
    bool dense_by_tile = false;    // the width was narrowed because the largest tile did not fit the LDS (handle_overflow): returns by tile size
    bool all_near = false;         // single-class lists (near radius = list radius): a build met a far class beyond the tiled record's
                                   // 504 entries; two classes again once the longest list is below that
    uint32_t K = 4, adapt = 1;
    uint32_t K_bad = 0, K_bad_ttl = 0;   // interval that violated the skin recently: stay below it for a while
    uint32_t steps_since_build = 0;
    float rv = 0;
    uint64_t rebuilds = 0, rollbacks = 0;
    uint32_t n_bond_types = 0;
    std::vector<unsigned long long> lcount;
    gd_timing timing{};

    // device: static (bead order)
    DevBuf<float2> ab_o; DevBuf<float> mob_o; DevBuf<float4> bendE_o; DevBuf<unsigned char> psmask_o, bdeg_o;
    DevBuf<unsigned> badj_o; DevBuf<int4> chain_o; DevBuf<BondType> btab;
    // device: per slot
    DevBuf<float4> pos[2], xb, fout, snap;
    DevBuf<unsigned> orig[2], slot_of, rank, members, cell_cnt, cell_start, nbr, meta, badj, flags;
    DevBuf<float> bbox_enc, bbox_w;      // box of the last build's positions (two halves: read / written), k_scatter's per-wave partials
    int bbox_cur = 0;              // half of bbox_enc the next build reads (the other one is accumulated by it)
    bool bbox_valid = false;       // open boxes: bbox_enc[bbox_cur] holds the bounding box of the positions the last build sorted
    DevBuf<unsigned short> nbr16; DevBuf<TileDesc> tiles;
    DevBuf<float4> rec_x0; DevBuf<uint2> rec_mo; DevBuf<unsigned char> len_prev;
    // Ragged rows of the tiled lists (BuildParams): per-wave row table, what every bead needed at the last build, the pool's cursor.
    // nbr16 IS the pool: pool KiB = nbr16.n / 512 entries.
    DevBuf<uint2> wtab, rqueue; DevBuf<unsigned short> need_prev; DevBuf<unsigned> pool;
    bool need_valid = false;       // need_prev describes the state about to be listed well enough to predict row widths from it
    float need_rv = 0; bool need_all_near = false;      // list radius / class mode need_prev was counted at
    uint32_t pool_used = 0;        // KiB the last build took (its cursor's final value: the need, when the pool was full)
    uint32_t repairs = 0;          // k_step waves the last build read back had to repair (diagnostics)
    uint32_t repair_wide = 0;      // > 0: accepted chunks still to run with a repair block for EVERY wave (a build queued more than GD_REPAIR_GRID)
    DevBuf<float> bbox;
    DevBuf<float2> ab; DevBuf<float> mobs; DevBuf<float4> bendE; DevBuf<int4> chain;
    float mob_uniform = -1.f;
    DevBuf<GridP> grid; DevBuf<DevCtx> ctx[2]; DevBuf<float4> react_part[2]; DevBuf<double> epart;   // react_part ping-pongs with ctx
    DevBuf<unsigned long long> lcount_d; DevBuf<float> noise;
    DevBuf<unsigned long long> seeds_d;     // gd_run_desc.replica_seeds of the run in progress
    DevBuf<unsigned> dmax;          // [R] largest squared displacement since the list build (k_step keeps it; zeroed by the build)
    float rn = 0;                   // near-class radius of the tiled list in use
    double a2_ema = 0;              // running mean of (largest displacement)^2 per step of an interval (interval adaptation; 0: none yet)
    double last_dt = 0, last_kT = -1;
    int last_flags = 0;             // flags of the last gd_run (the look-ahead of a list built between runs, gd_search_pairs)
    bool search_list = false;       // the list in use was built by gd_search_pairs at a radius beyond the force list's
    // Skin selection by measured cost (per workload): a few candidate widths are each run for a few verified chunks once the
    // rebuild interval has settled, the device time per step decides (gd_run, tune_skin).  Results do not depend on the skin
    // (verified lists + rollback), only the cost does.
    struct SkinTuner {
        bool enabled = false, done = false;     // opt-in: gd_tuning.auto_skin
        std::vector<double> cand, cost;
        size_t idx = 0;
        int settle = 0, measured = 0, wait = 4, rounds = 0;      // wait: accepted chunks before the (next) sweep may start
        uint32_t K_ref = 0;                                      // rebuild interval when the last sweep ended
        uint32_t cap_ref = 0;                                    // tile class when the last sweep ended
        double acc_ms = 0; uint64_t acc_steps = 0;
    } tuner;
    double pend_dt = 0; int pend_flags = 0;      // timestep and flags of the run that left its last callback pending (GD_RUN_DEFER_CALLBACK)
    double near_frac = 0.65;        // near-class radius = cutoff + near_frac x (list radius - cutoff)
    // gd_search_pairs: device output, counters, and the cached result of the last call
    DevBuf<uint2> sp_out; DevBuf<unsigned long long> sp_count; std::vector<uint2> sp_host;
    bool sp_valid = false; uint32_t sp_r = 0; double sp_dcut = 0; uint64_t sp_serial = 0;
    // gd_contacts_*: per-replica count tables (ContactTab), the pair buffer of an update, dump buffers
    DevBuf<unsigned long long> ct_words; DevBuf<unsigned> ct_distinct; size_t ct_cap = 0;
    DevBuf<uint2> ct_pairs; DevBuf<unsigned long long> ct_count; std::vector<unsigned> ct_distinct_h;
    DevBuf<unsigned long long> ct_ck[2]; DevBuf<unsigned> ct_cv[2], ct_n; DevBuf<char> ct_tmp;
    uint64_t state_serial = 1;     // bumped by everything that changes positions or the model (invalidates the cache)
    uint64_t verified_serial = 0;  // == state_serial: the last gd_run ended on an accepted chunk, i.e. the resident list was verified for the
                                   // positions and the cutoff an observation now sees (no bead beyond the skin margin): energies need no build
    int ocur = 0;   // which orig[] buffer is current
    // Compensated positions (small-dt / T = 0 runs, k_step's p.comp): fp32 residuals by bead index, so that the position of a bead is
    // pos + lo.  gd_set_positions fills them from the fp64 input, a compensated run keeps them, any other run invalidates them.
    DevBuf<float4> lo, snap_lo;
    bool lo_valid = false;         // lo describes the current positions (else: taken as zero)
    bool comp_last = false;        // the last gd_run stepped with the compensated update (diagnostics)
    double mob_max = 1.0;
    std::vector<hipEvent_t> events;
    ~gd_system()
    {
        for (auto e : events) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static int valid_pq(int p, int q) { return (p == 2 || p == 4 || p == 6 || p == 8 || p == 12) && q >= 1 && q <= 4; }

extern "C" int gd_abi_version(void) { return GD_ABI_VERSION; }

extern "C" int gd_create_abi(int abi_version, const gd_desc *d, gd_system **out)
{
    if (abi_version != GD_ABI_VERSION)
        return fail(GD_EINVAL, "gd_create: the caller was built against gdyn.h ABI version %d, this library implements %d", abi_version, GD_ABI_VERSION);
    if (!d || !out) return fail(GD_EINVAL, "gd_create: NULL argument");
    if (d->n_beads == 0 || d->n_replicas == 0) return fail(GD_EINVAL, "gd_create: n_beads and n_replicas must be > 0");
    if (d->n_beads > GD_ADJ_MASK) return fail(GD_EINVAL, "gd_create: n_beads exceeds %u", GD_ADJ_MASK);
    if (d->box_kind != GD_BOX_OPEN && d->box_kind != GD_BOX_PERIODIC) return fail(GD_EINVAL, "gd_create: bad box_kind");
    if (d->box_kind == GD_BOX_PERIODIC)
        for (int k = 0; k < 3; k++) if (!(d->box[k] > 0)) return fail(GD_EINVAL, "gd_create: periodic box needs positive periods");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GD_ENODEVICE, "gd_create: no HIP device visible (libgdyn has no CPU fallback)");
    if (d->device < 0 || d->device >= ndev) return fail(GD_ENODEVICE, "gd_create: device %d not in [0,%d)", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));
    {   // per-device set-up (LDS opt-in of the kernels): once per device ordinal, complete before any handle on it exists
        static gd::DeviceOnce<> once;
        const int rc = once.run(d->device, [](int) { return (int)gd_kernels_init_device(); });
        if (rc != (int)hipSuccess)
            return fail(GD_EHIP, "gd_create: kernel set-up on device %d failed: %s", d->device, rc < 0 ? "device ordinal beyond the guard's table" : hipGetErrorString((hipError_t)rc));
    }
    gd_system *s = new (std::nothrow) gd_system();
    if (!s) return fail(GD_ENOMEM, "gd_create: out of host memory");
    s->N = d->n_beads; s->R = d->n_replicas; s->device = d->device; s->box_kind = d->box_kind;
    memcpy(s->box, d->box, sizeof s->box);
    s->nblk = (s->N + GD_BLOCK - 1) / GD_BLOCK;
    s->Np = s->nblk * GD_BLOCK;
    s->cpb = (s->nblk + GD_XCDS - 1) / GD_XCDS;
    if (s->R % GD_XCDS == 0 && !dev_env("GDYN_SLICE_MAP")) s->cpb = 0;      // whole replicas per XCD (see block_map)
    s->a.assign(s->N, 0.0); s->b.assign(s->N, 0.0); s->mob.assign(s->N, 1.0); s->bend.assign(s->N, 0.0);
    s->hctx.assign(s->R, DevCtx{});
    for (auto &c : s->hctx) { c.bead_scale = 1; c.bond_scale = 1; }   // wall_semiaxes {0,0,0} until a wall is set (simulation_context.hpp:16)
    s->lcount.assign(2 * (size_t)s->R, 0ull);      // per replica: directed entries, then the near entries (in fours) of tiled lists
    s->ncell_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(8ull * s->N, 4096ull), 262144ull);
    if (const char *e = dev_env("GDYN_NEAR_FRAC")) s->near_frac = atof(e);
    if (const char *e = dev_env("GDYN_SKIN")) { s->skin = atof(e); s->skin_fixed = true; }
    if (dev_env("GDYN_AUTO_SKIN")) s->tuner.enabled = true;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; return fail(GD_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    const size_t RNp = (size_t)s->R * s->Np, RN = (size_t)s->R * s->N;
    bool ok = true;
    for (int k = 0; k < 2; k++) {
        ok = ok && s->pos[k].resize(RNp) == hipSuccess && s->orig[k].resize(RNp) == hipSuccess && s->ctx[k].resize(s->R) == hipSuccess &&
             s->react_part[k].resize((size_t)s->R * s->nblk) == hipSuccess;
    }
    ok = ok && s->xb.resize(RNp) == hipSuccess && s->slot_of.resize(RN) == hipSuccess && s->bbox_enc.resize((size_t)2 * s->R * 6) == hipSuccess && s->bbox_w.resize((size_t)s->R * s->nblk * (GD_BLOCK / 64) * 6) == hipSuccess &&
         s->rank.resize(RNp) == hipSuccess && s->members.resize(RNp) == hipSuccess && s->cell_cnt.resize((size_t)s->R * (s->ncell_cap + 1)) == hipSuccess &&
         s->cell_start.resize((size_t)s->R * (s->ncell_cap + 1)) == hipSuccess && s->meta.resize(RNp) == hipSuccess &&
         s->flags.resize((size_t)s->R * GD_NFLAGS) == hipSuccess && s->bbox.resize((size_t)s->R * s->nblk * 6) == hipSuccess &&
         s->ab.resize(RNp) == hipSuccess && s->mobs.resize(RNp) == hipSuccess && s->grid.resize(s->R) == hipSuccess &&
         s->epart.resize((size_t)s->R * s->nblk) == hipSuccess &&
         s->lcount_d.resize(2 * (size_t)s->R) == hipSuccess && s->dmax.resize((size_t)s->R * GD_DMAX_STRIDE) == hipSuccess && s->fout.resize(RN) == hipSuccess && s->snap.resize(RN) == hipSuccess &&
         s->tiles.resize((size_t)s->R * s->nblk) == hipSuccess &&
         s->rec_x0.resize(RNp) == hipSuccess && s->rec_mo.resize(RNp) == hipSuccess && s->len_prev.resize((size_t)s->R * s->N) == hipSuccess &&
         s->wtab.resize(RNp / 64) == hipSuccess && s->need_prev.resize((size_t)s->R * s->N) == hipSuccess && s->pool.resize(4) == hipSuccess && s->rqueue.resize(RNp / 64) == hipSuccess &&
         s->lo.resize(RN) == hipSuccess;
    s->lo_valid = ok;      // (positions and residuals all zero)
    if (!ok) { delete s; return fail(GD_ENOMEM, "gd_create: device allocation failed (%zu slots)", RNp); }
    gd_launch_identity(s->orig[0].p, s->slot_of.p, s->N, s->Np, s->R, s->stream);
    if (hipStreamSynchronize(s->stream) != hipSuccess) { delete s; return fail(GD_EHIP, "gd_create: identity kernel failed"); }
    *out = s;
    return GD_OK;
}

extern "C" int gd_destroy(gd_system *s)
{
    if (!s) return GD_OK;
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->h_chunk) (void)hipHostFree(s->h_chunk);
    delete s;
    return GD_OK;
}

// ---------------------------------------------------------------- context

static int upload_ctx(gd_system *s)
{
    if (!s->ctx_dirty) return GD_OK;
    HIPCHK(hipMemcpyAsync(s->ctx[s->ccur].p, s->hctx.data(), s->R * sizeof(DevCtx), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->ctx_dirty = false;
    return GD_OK;
}
static int download_ctx(gd_system *s)
{
    HIPCHK(hipMemcpyAsync(s->hctx.data(), s->ctx[s->ccur].p, s->R * sizeof(DevCtx), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return GD_OK;
}

// ---------------------------------------------------------- model setters

extern "C" int gd_set_positions(gd_system *s, const double *xyz)
{
    if (!s || !xyz) return fail(GD_EINVAL, "gd_set_positions: NULL argument");
    HIPCHK(hipSetDevice(s->device));
    const size_t RN = (size_t)s->R * s->N;
    std::vector<float4> h(RN), hl(RN);
    for (size_t i = 0; i < RN; i++) {
        if (!std::isfinite(xyz[3 * i]) || !std::isfinite(xyz[3 * i + 1]) || !std::isfinite(xyz[3 * i + 2]))
            return fail(GD_EINVAL, "gd_set_positions: non-finite coordinate at %zu", 3 * i);
        h[i] = make_float4((float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2], 0.f);
        // what the fp32 coordinate drops of the fp64 input: the residual of the compensated update starts from it
        hl[i] = make_float4((float)(xyz[3 * i] - (double)h[i].x), (float)(xyz[3 * i + 1] - (double)h[i].y), (float)(xyz[3 * i + 2] - (double)h[i].z), 0.f);
    }
    HIPCHK(hipMemcpy2DAsync(s->pos[s->pcur].p, (size_t)s->Np * sizeof(float4), h.data(), (size_t)s->N * sizeof(float4),
                            (size_t)s->N * sizeof(float4), s->R, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->lo.p, hl.data(), RN * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    s->lo_valid = true;
    gd_launch_identity(s->orig[s->ocur].p, s->slot_of.p, s->N, s->Np, s->R, s->stream);
    HIPCHK(hipStreamSynchronize(s->stream));
    s->list_valid = false; s->state_serial++; s->w_packed = false; s->bbox_valid = false;
    s->need_valid = false;      // (positions from the caller: what the beads needed before says nothing about their lists now)
    return GD_OK;
}

// Snapshot download: gather to bead order and pack xyz on the device, one copy into a pinned staging buffer that
// lives with the handle (a fresh pageable buffer per call costs page faults and a slower copy).
static int fetch_xyz(gd_system *s, const float **out, int quantize)
{
    HIPCHK(hipSetDevice(s->device));
    const size_t n3 = (size_t)s->R * s->N * 3;
    if (!s->h_stage) HIPCHK(hipHostMalloc((void **)&s->h_stage, n3 * sizeof(float), hipHostMallocDefault));
    gd_launch_gather_xyz(s->pos[s->pcur].p, s->slot_of.p, (float *)s->fout.p, s->N, s->Np, s->R, quantize, s->stream);   // fout: R*N float4 >= n3 floats
    HIPCHK(hipMemcpyAsync(s->h_stage, s->fout.p, n3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *out = s->h_stage;
    return GD_OK;
}

extern "C" int gd_get_positions(gd_system *s, double *xyz)
{
    if (!s || !xyz) return fail(GD_EINVAL, "gd_get_positions: NULL argument");
    const float *h = nullptr;
    GDCHK(fetch_xyz(s, &h, 0));
    const size_t n3 = (size_t)s->R * s->N * 3;
    for (size_t i = 0; i < n3; i++) xyz[i] = h[i];
    if (s->lo_valid) {      // compensated positions: the fp64 boundary gets pos + lo
        std::vector<float4> hl(n3 / 3);
        HIPCHK(hipMemcpy(hl.data(), s->lo.p, hl.size() * sizeof(float4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < hl.size(); i++) { xyz[3 * i] += (double)hl[i].x; xyz[3 * i + 1] += (double)hl[i].y; xyz[3 * i + 2] += (double)hl[i].z; }
    }
    return GD_OK;
}

extern "C" int gd_get_positions_f32(gd_system *s, float *xyz, int quantize)
{
    if (!s || !xyz) return fail(GD_EINVAL, "gd_get_positions_f32: NULL argument");
    const float *h = nullptr;
    GDCHK(fetch_xyz(s, &h, quantize));
    memcpy(xyz, h, (size_t)s->R * s->N * 3 * sizeof(float));
    return GD_OK;
}

extern "C" int gd_set_bead_params(gd_system *s, const double *a, const double *b, const double *mob, const double *bend)
{
    if (!s) return fail(GD_EINVAL, "gd_set_bead_params: NULL system");
    if (mob) for (uint32_t i = 0; i < s->N; i++) if (!(mob[i] >= 0)) return fail(GD_EINVAL, "gd_set_bead_params: negative mobility at %u", i);
    if (a) s->a.assign(a, a + s->N);
    if (b) s->b.assign(b, b + s->N);
    if (mob) s->mob.assign(mob, mob + s->N);
    if (bend) s->bend.assign(bend, bend + s->N);
    s->topo_dirty = true;
    return GD_OK;
}

extern "C" int gd_set_pair_softcore(gd_system *s, const gd_pair_softcore *p)
{
    if (!s || !p) return fail(GD_EINVAL, "gd_set_pair_softcore: NULL argument");
    if (!valid_pq(p->p_a, p->q_a) || !valid_pq(p->p_b, p->q_b)) return fail(GD_EINVAL, "gd_set_pair_softcore: unsupported softcore powers");
    if (p->sigma_a < 0 || p->sigma_b < 0) return fail(GD_EINVAL, "gd_set_pair_softcore: negative diameter");
    s->pair = *p; s->has_pair = true; s->list_valid = false;
    return GD_OK;
}

static int check_bond_params(const gd_bond_params *p)
{
    if (p->kind < GD_POT_HARMONIC || p->kind > GD_POT_SOFTCORE) return fail(GD_EINVAL, "bond params: bad kind %d", p->kind);
    if (p->kind == GD_POT_SOFTCORE && !valid_pq(p->p, p->q)) return fail(GD_EINVAL, "bond params: unsupported softcore powers");
    return GD_OK;
}

static int add_bond_type(gd_system *s, const gd_bond_params *p, int term)
{
    for (size_t i = 0; i < s->btypes.size(); i++)
        if (!memcmp(&s->btypes[i], p, sizeof *p) && s->bterm[i] == term) return (int)i;
    if (s->btypes.size() >= GD_MAX_BOND_TYPES) return -1;
    s->btypes.push_back(*p); s->bterm.push_back(term);
    return (int)s->btypes.size() - 1;
}

extern "C" int gd_add_bond_range(gd_system *s, const gd_bond_params *p, uint32_t start, uint32_t end, uint32_t stride)
{
    if (!s || !p) return fail(GD_EINVAL, "gd_add_bond_range: NULL argument");
    GDCHK(check_bond_params(p));
    if (start > end || end > s->N) return fail(GD_EINVAL, "gd_add_bond_range: range [%u,%u) outside [0,%u)", start, end, s->N);
    if (stride < 1 || stride > 2) return fail(GD_EINVAL, "gd_add_bond_range: stride must be 1 or 2");
    const int t = add_bond_type(s, p, GD_TERM_BOND);
    if (t < 0) return fail(GD_EINVAL, "gd_add_bond_range: too many bond parameter sets");
    for (uint32_t i = start; i + stride < end; i++) s->bonds.push_back({i, i + stride, t});
    s->topo_dirty = true;
    return GD_OK;
}

extern "C" int gd_add_bond_pairs(gd_system *s, const gd_bond_params *p, const uint32_t *pairs, uint32_t n)
{
    if (!s || !p || (n && !pairs)) return fail(GD_EINVAL, "gd_add_bond_pairs: NULL argument");
    GDCHK(check_bond_params(p));
    for (uint32_t k = 0; k < n; k++)
        if (pairs[2 * k] >= s->N || pairs[2 * k + 1] >= s->N || pairs[2 * k] == pairs[2 * k + 1])
            return fail(GD_EINVAL, "gd_add_bond_pairs: bad pair %u (%u,%u)", k, pairs[2 * k], pairs[2 * k + 1]);
    const int t = add_bond_type(s, p, GD_TERM_BOND);
    if (t < 0) return fail(GD_EINVAL, "gd_add_bond_pairs: too many bond parameter sets");
    for (uint32_t k = 0; k < n; k++) s->bonds.push_back({pairs[2 * k], pairs[2 * k + 1], t});
    s->topo_dirty = true;
    return GD_OK;
}

extern "C" int gd_set_dynamic_pairs(gd_system *s, uint32_t slot, const gd_bond_params *p, const uint32_t *pairs, uint32_t n)
{
    if (!s || !p || (n && !pairs)) return fail(GD_EINVAL, "gd_set_dynamic_pairs: NULL argument");
    if (slot >= 4) return fail(GD_EINVAL, "gd_set_dynamic_pairs: slot %u out of range", slot);
    GDCHK(check_bond_params(p));
    for (uint32_t k = 0; k < n; k++)
        if (pairs[2 * k] >= s->N || pairs[2 * k + 1] >= s->N || pairs[2 * k] == pairs[2 * k + 1])
            return fail(GD_EINVAL, "gd_set_dynamic_pairs: bad pair %u", k);
    s->dyn[slot].used = true; s->dyn[slot].p = *p;
    s->dyn[slot].pairs.assign(pairs, pairs + 2 * (size_t)n);
    s->topo_dirty = true;
    return GD_OK;
}

extern "C" int gd_add_bending_range(gd_system *s, uint32_t start, uint32_t end, double energy, int per_bead)
{
    if (!s) return fail(GD_EINVAL, "gd_add_bending_range: NULL system");
    if (start > end || end > s->N) return fail(GD_EINVAL, "gd_add_bending_range: range outside [0,N)");
    s->bends.push_back({start, end, energy, per_bead});
    s->topo_dirty = true;
    return GD_OK;
}

extern "C" int gd_add_point_source(gd_system *s, int kind, double k, double b, const double point[3], const uint32_t *targets, uint32_t nt)
{
    if (!s || !point) return fail(GD_EINVAL, "gd_add_point_source: NULL argument");
    if (kind != GD_POT_HARMONIC && kind != GD_POT_SEMISPRING && kind != GD_POT_SPRING) return fail(GD_EINVAL, "gd_add_point_source: unsupported kind");
    if (s->psrc.size() >= GD_MAX_POINT_SOURCES) return fail(GD_EINVAL, "gd_add_point_source: at most %d sources", GD_MAX_POINT_SOURCES);
    PointSource ps; ps.kind = kind; ps.k = k; ps.b = b; memcpy(ps.p, point, sizeof ps.p);
    if (targets) {
        for (uint32_t i = 0; i < nt; i++) if (targets[i] >= s->N) return fail(GD_EINVAL, "gd_add_point_source: target %u out of range", targets[i]);
        ps.mask.assign(s->N, 0);
        for (uint32_t i = 0; i < nt; i++) ps.mask[targets[i]] = 1;
    }
    s->psrc.push_back(std::move(ps));
    s->topo_dirty = true;
    return GD_OK;
}

extern "C" int gd_set_ellipsoid_wall(gd_system *s, const gd_wall *w)
{
    if (!s || !w) return fail(GD_EINVAL, "gd_set_ellipsoid_wall: NULL argument");
    if (!valid_pq(w->p_a, w->q_a) || !valid_pq(w->p_b, w->q_b)) return fail(GD_EINVAL, "gd_set_ellipsoid_wall: unsupported softcore powers");
    for (int k = 0; k < 3; k++) if (!(w->init_semiaxes[k] > 0)) return fail(GD_EINVAL, "gd_set_ellipsoid_wall: semiaxes must be positive");
    s->wall = *w; s->has_wall = true;
    for (auto &c : s->hctx) memcpy(c.semi, w->init_semiaxes, sizeof c.semi);
    s->ctx_dirty = true;
    return GD_OK;
}

extern "C" int gd_set_pair_softwell(gd_system *s, double energy, double decay, double cutoff, const uint32_t *targets, uint32_t n)
{
    if (!s || (n && !targets)) return fail(GD_EINVAL, "gd_set_pair_softwell: NULL argument");
    if (n > 4096) return fail(GD_EINVAL, "gd_set_pair_softwell: at most 4096 targets");
    if (n && (!(decay > 0) || !(cutoff > 0))) return fail(GD_EINVAL, "gd_set_pair_softwell: decay and cutoff must be positive");
    for (uint32_t k = 0; k < n; k++) if (targets[k] >= s->N) return fail(GD_EINVAL, "gd_set_pair_softwell: target %u out of range", k);
    {   // set_neighbor_targets takes a set of particles: a repeated index would make two threads update one bead
        std::vector<uint32_t> t(targets, targets + n);
        std::sort(t.begin(), t.end());
        for (uint32_t k = 1; k < n; k++) if (t[k] == t[k - 1]) return fail(GD_EINVAL, "gd_set_pair_softwell: target %u listed twice", t[k]);
    }
    HIPCHK(hipSetDevice(s->device));
    s->sw_n = 0;
    if (n) {
        HIPCHK(s->sw_targets.resize(n, false));
        HIPCHK(hipMemcpy(s->sw_targets.p, targets, n * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(s->sw_esum.resize(s->R));
        s->sw_n = n; s->sw_eps = energy; s->sw_decay = decay; s->sw_cut = cutoff;
    }
    return GD_OK;
}

// the droplet term of the step / force / energy evaluation that `p` describes (same positions, same buffers)
static void launch_softwell(gd_system *s, const StepParams &p, int mode)
{
    SoftwellP q;
    memset(&q, 0, sizeof q);
    q.pos_in = p.pos_in; q.pos_out = p.pos_out; q.fout = s->fout.p; q.esum = s->sw_esum.p;
    q.slot_of = s->slot_of.p; q.targets = s->sw_targets.p;
    q.mob_o = s->mob_uniform >= 0.f ? nullptr : s->mob_o.p; q.mob_uniform = s->mob_uniform; q.dt = p.dt;
    q.lo = p.lo; q.comp = p.comp;      // (a compensated step keeps the droplet's share of mu F dt in the residuals too)
    q.eps = (float)s->sw_eps; q.inv_d2 = (float)(1.0 / (s->sw_decay * s->sw_decay)); q.rc2 = (float)(s->sw_cut * s->sw_cut);
    q.N = s->N; q.Np = s->Np; q.R = s->R; q.M = s->sw_n;
    q.periodic = s->box_kind == GD_BOX_PERIODIC;
    for (int k = 0; k < 3; k++) { q.box[k] = (float)s->box[k]; q.inv_box[k] = s->box[k] > 0 ? (float)(1.0 / s->box[k]) : 0.f; }
    gd_launch_softwell(q, mode, s->stream);
}

extern "C" int gd_set_inner_sphere_wall(gd_system *s, const gd_inner_sphere *w)
{
    if (!s || !w) return fail(GD_EINVAL, "gd_set_inner_sphere_wall: NULL argument");
    if (!valid_pq(w->p_a, w->q_a) || !valid_pq(w->p_b, w->q_b)) return fail(GD_EINVAL, "gd_set_inner_sphere_wall: unsupported softcore powers");
    if (!(w->radius > 0)) return fail(GD_EINVAL, "gd_set_inner_sphere_wall: radius must be positive");
    s->inner = *w; s->has_inner = true;
    return GD_OK;
}

extern "C" int gd_set_scaling(gd_system *s, double bi, double bt, double oi, double ot)
{
    if (!s) return fail(GD_EINVAL, "gd_set_scaling: NULL system");
    if (!(bt > 0) || !(ot > 0) || !(bi > 0) || !(oi > 0)) return fail(GD_EINVAL, "gd_set_scaling: init and tau must be positive");
    s->has_scaling = true; s->bs_init = bi; s->bs_tau = bt; s->bo_init = oi; s->bo_tau = ot;
    for (auto &c : s->hctx) { c.bead_scale = bi; c.bond_scale = oi; }
    s->ctx_dirty = true; s->list_valid = false;
    return GD_OK;
}

extern "C" int gd_get_context(gd_system *s, uint32_t r, gd_context *o)
{
    if (!s || !o) return fail(GD_EINVAL, "gd_get_context: NULL argument");
    if (r >= s->R) return fail(GD_EINVAL, "gd_get_context: replica out of range");
    memset(o, 0, sizeof *o);
    const DevCtx &c = s->hctx[r];
    o->step = c.step; o->time = c.time; o->bead_scale = c.bead_scale; o->bond_scale = c.bond_scale;
    memcpy(o->semiaxes, c.semi, sizeof c.semi); memcpy(o->axial_reaction, c.react, sizeof c.react);
    o->list_entries = s->lcount[r]; o->rebuilds = s->rebuilds; o->rollbacks = s->rollbacks;
    o->rebuild_interval = s->K; o->list_radius = s->rv;
    o->list_path = !s->list_valid && s->rebuilds == 0 ? 0u : (s->list_tiled ? 2u : 1u);
    o->callback_pending = c.pending ? 1u : 0u;
    o->tile_capacity = (s->list_valid && s->list_tiled) ? s->list_tile_cap : 0u;
    o->compensated = s->comp_last ? 1u : 0u;
    o->largest_tile = (s->list_valid && s->list_tiled) ? s->last_need_t : 0u;
    o->row_repairs = s->list_tiled ? s->repairs : 0u;
    o->near_entries = s->list_tiled ? s->lcount[(size_t)s->R + r] : 0ull;
    o->list_bytes = !s->list_valid ? 0ull : s->list_tiled ? 1024ull * s->pool_used : (uint64_t)s->list_W * s->R * s->Np * 4ull;
    return GD_OK;
}

extern "C" int gd_begin_phase(gd_system *s, const double *semi)
{
    if (!s) return fail(GD_EINVAL, "gd_begin_phase: NULL system");
    for (uint32_t r = 0; r < s->R; r++) {
        DevCtx &c = s->hctx[r];
        c.step = 0; c.time = 0; c.pending = 0;
        if (s->has_scaling) { c.bead_scale = s->bs_init; c.bond_scale = s->bo_init; }
        if (semi) memcpy(c.semi, semi + 3 * r, sizeof c.semi);
    }
    s->ctx_dirty = true; s->list_valid = false;
    return GD_OK;
}

extern "C" int gd_set_context(gd_system *s, uint32_t r, int64_t step, double bead_scale, double bond_scale, const double semi[3])
{
    if (!s) return fail(GD_EINVAL, "gd_set_context: NULL system");
    if (r >= s->R) return fail(GD_EINVAL, "gd_set_context: replica out of range");
    if (!(bead_scale > 0) || !(bond_scale > 0)) return fail(GD_EINVAL, "gd_set_context: scales must be positive");
    DevCtx &c = s->hctx[r];
    c.step = step; c.bead_scale = bead_scale; c.bond_scale = bond_scale; c.pending = 0;
    if (semi) memcpy(c.semi, semi, sizeof c.semi);
    s->ctx_dirty = true; s->list_valid = false;
    return GD_OK;
}

extern "C" int gd_set_tuning(gd_system *s, const gd_tuning *t)
{
    if (!s || !t) return fail(GD_EINVAL, "gd_set_tuning: NULL argument");
    if (t->kernel_path > 2) return fail(GD_EINVAL, "gd_set_tuning: kernel_path must be 0..2");      // (validated before any state changes)
    if (t->near_fraction < 0 || t->near_fraction > 1) return fail(GD_EINVAL, "gd_set_tuning: near_fraction must be in [0,1]");
    HIPCHK(hipSetDevice(s->device));
    if (t->skin > 0) { s->skin = t->skin; s->skin_fixed = true; }
    else if (t->skin < 0) { s->skin = 0.75; s->skin_fixed = false; s->skin_dense_from = 0; s->dense_by_tile = false; }      // back to the library's own choice
    s->skin_streak = 0; s->skin_hold = 0; s->skin_next = 0;
    if (t->rebuild_interval > 0) s->K = t->rebuild_interval;
    s->tuner = gd_system::SkinTuner{};
    s->tuner.enabled = (t->auto_skin != 0 || dev_env("GDYN_AUTO_SKIN")) && t->adapt_interval != 0;          // (a fixed cadence: nothing to select for)
    if (t->near_fraction > 0) s->near_frac = t->near_fraction;
    s->a2_ema = 0;
    s->adapt = t->adapt_interval;
    if (t->list_width > 0 && t->list_width != s->W) { (void)s->nbr.resize(0); s->W = t->list_width; }
    s->kernel_path = t->kernel_path; s->tiled_ok = true; s->tiled_off = 0;
    s->list_valid = false;
    return GD_OK;
}
extern "C" int gd_get_timing(gd_system *s, gd_timing *o) { if (!s || !o) return fail(GD_EINVAL, "gd_get_timing: NULL"); *o = s->timing; return GD_OK; }
extern "C" int gd_get_stream(gd_system *s, void **st) { if (!s || !st) return fail(GD_EINVAL, "gd_get_stream: NULL"); *st = (void *)s->stream; return GD_OK; }

// --------------------------------------------------------------- topology

static float pair_cutoff(const gd_system *s)
{
    double m = 0;
    if (s->has_pair) {
        if (s->pair.eps_a != 0 && s->pair.sigma_a > m) m = s->pair.sigma_a;
        if (s->pair.eps_b != 0 && s->pair.sigma_b > m) m = s->pair.sigma_b;
    }
    return (float)m;
}

// Flatten the host model into the bead-order device tables (bond adjacency in ELL form,
// bending energies of the three triplets of each bead, point-source masks).
static int finalize_topology(gd_system *s)
{
    if (!s->topo_dirty) return GD_OK;
    HIPCHK(hipSetDevice(s->device));
    const uint32_t N = s->N;
    // bond types: static + dynamic sets
    std::vector<gd_bond_params> types = s->btypes;
    std::vector<int> terms = s->bterm;
    std::vector<Bond> all = s->bonds;
    for (int d = 0; d < 4; d++) if (s->dyn[d].used) {
        if (types.size() >= GD_MAX_BOND_TYPES) return fail(GD_EINVAL, "too many bond parameter sets");
        types.push_back(s->dyn[d].p); terms.push_back(GD_TERM_DYNAMIC);
        const int t = (int)types.size() - 1;
        for (size_t k = 0; k + 1 < s->dyn[d].pairs.size(); k += 2) all.push_back({s->dyn[d].pairs[k], s->dyn[d].pairs[k + 1], t});
    }
    // AB-mixed bond sets (K = a Ka + b Kb, l = a la + b lb with a = (a_i + a_j)/2, b likewise: simulation_driver_forcefield.cc:58-88)
    // have few distinct (a, b) per set -- the bead types are a handful of values -- so every bond gets the index of its own,
    // already mixed, parameter record and the kernels skip the per-bond mixing arithmetic.  More records than the table holds:
    // the sets stay as given and the kernels mix at run time.
    s->bonds_premixed = false;
    {
        std::vector<gd_bond_params> t2; std::vector<int> term2; std::vector<Bond> all2 = all;
        bool fits = true, any_mixed = false;
        for (auto &b : all2) {
            gd_bond_params q = types[b.type];
            if (q.mix) {
                any_mixed = true;
                const double a = 0.5 * (s->a[b.i] + s->a[b.j]), bb = 0.5 * (s->b[b.i] + s->b[b.j]);
                q.k_a = a * q.k_a + bb * q.k_b; q.l_a = a * q.l_a + bb * q.l_b; q.k_b = 0; q.l_b = 0; q.mix = 0;
            }
            int found = -1;
            for (size_t k = 0; k < t2.size() && found < 0; k++)
                if (!memcmp(&t2[k], &q, sizeof q) && term2[k] == terms[b.type]) found = (int)k;
            if (found < 0) {
                if (t2.size() >= GD_MAX_BOND_TYPES) { fits = false; break; }
                t2.push_back(q); term2.push_back(terms[b.type]); found = (int)t2.size() - 1;
            }
            b.type = found;
        }
        if (fits && any_mixed) { types.swap(t2); terms.swap(term2); all.swap(all2); s->bonds_premixed = true; }
        else if (!any_mixed) s->bonds_premixed = true;      // nothing to mix at run time either way
    }
    std::vector<unsigned> deg(N, 0);
    for (auto &b : all) { deg[b.i]++; deg[b.j]++; }
    uint32_t WB = 0;
    for (auto v : deg) WB = std::max(WB, v);
    if (WB > 255) return fail(GD_EINVAL, "a bead has %u bonds (max 255)", WB);
    std::vector<unsigned> adj((size_t)std::max(WB, 1u) * N, 0u);
    std::vector<unsigned char> dg(N, 0);
    for (auto &b : all) {
        adj[(size_t)dg[b.i] * N + b.i] = b.j | ((unsigned)b.type << GD_ADJ_SHIFT); dg[b.i]++;
        adj[(size_t)dg[b.j] * N + b.j] = b.i | ((unsigned)b.type << GD_ADJ_SHIFT); dg[b.j]++;
    }
    std::vector<BondType> bt(std::max<size_t>(types.size(), 1));
    s->has_softcore_bonds = false;
    for (size_t i = 0; i < types.size(); i++) {
        const gd_bond_params &p = types[i];
        const bool harmonic = p.kind == GD_POT_HARMONIC;     // U = K r^2 / 2 is the spring with rest length 0
        bt[i] = BondType{(float)p.k_a, (float)p.k_b, harmonic ? 0.f : (float)p.l_a, harmonic ? 0.f : (float)p.l_b,
                         (p.mix ? 1 : 0) | (p.scale_by_bond_scale ? 2 : 0) | (p.minimum_image ? 4 : 0) | (terms[i] << 8),
                         p.kind == GD_POT_SEMISPRING ? 0.f : -3.0e38f, p.kind, p.p | (p.q << 8)};
        if (p.kind == GD_POT_SOFTCORE) s->has_softcore_bonds = true;
    }
    s->bonds_all_scaled = !types.empty();
    for (auto &p : types) if (!p.scale_by_bond_scale) s->bonds_all_scaled = false;
    // bending: energy of the triplet starting at each bead
    std::vector<double> tE(N, 0.0);
    for (auto &br : s->bends)
        for (uint32_t i = br.start; i + 2 < br.end; i++) tE[i] += br.per_bead ? s->bend[i + 1] : br.energy;
    std::vector<float4> bendE(N);
    std::vector<int4> chain(N);
    bool has_bend = false;
    for (uint32_t j = 0; j < N; j++) {
        const float el = j >= 2 ? (float)tE[j - 2] : 0.f, em = j >= 1 ? (float)tE[j - 1] : 0.f, ef = (float)tE[j];
        bendE[j] = make_float4(el, em, ef, 0.f);
        if (el != 0.f || em != 0.f || ef != 0.f) has_bend = true;
        int4 c = make_int4(-1, -1, -1, -1);
        if (el != 0.f) { c.x = (int)j - 2; c.y = (int)j - 1; }
        if (em != 0.f) { c.y = (int)j - 1; c.z = (int)j + 1; }
        if (ef != 0.f) { c.z = (int)j + 1; c.w = (int)j + 2; }
        chain[j] = c;
    }
    std::vector<unsigned char> psm(N, 0);
    for (size_t q = 0; q < s->psrc.size(); q++)
        for (uint32_t i = 0; i < N; i++) if (s->psrc[q].mask.empty() || s->psrc[q].mask[i]) psm[i] |= (unsigned char)(1u << q);
    std::vector<float2> ab(N);
    std::vector<float> mob(N);
    bool packable = true;
    for (uint32_t i = 0; i < N; i++) {
        ab[i] = make_float2((float)s->a[i], (float)s->b[i]); mob[i] = (float)s->mob[i];
        // (a,b) ride in pos.w as two fp16 when that is exact (0, .5, 1, 5 ... are)
        if ((double)__half2float(__float2half_rn(ab[i].x)) != s->a[i] || (double)__half2float(__float2half_rn(ab[i].y)) != s->b[i]) packable = false;
    }
    s->packed_ab = packable;

    HIPCHK(s->ab_o.resize(N)); HIPCHK(s->mob_o.resize(N)); HIPCHK(s->bendE_o.resize(N)); HIPCHK(s->psmask_o.resize(N));
    HIPCHK(s->bdeg_o.resize(N)); HIPCHK(s->badj_o.resize(adj.size())); HIPCHK(s->chain_o.resize(N)); HIPCHK(s->btab.resize(GD_MAX_BOND_TYPES));   /* always the full table: k_step stages it with one DMA piece */
    HIPCHK(hipMemcpy(s->ab_o.p, ab.data(), N * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->mob_o.p, mob.data(), N * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->bendE_o.p, bendE.data(), N * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->psmask_o.p, psm.data(), N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->bdeg_o.p, dg.data(), N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->badj_o.p, adj.data(), adj.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->chain_o.p, chain.data(), N * sizeof(int4), hipMemcpyHostToDevice));
    if (bt.size() > GD_MAX_BOND_TYPES) return fail(GD_EINVAL, "too many bond types");
    HIPCHK(hipMemcpy(s->btab.p, bt.data(), bt.size() * sizeof(BondType), hipMemcpyHostToDevice));
    s->n_bond_types = (uint32_t)bt.size();
    const size_t RNp = (size_t)s->R * s->Np;
    const uint32_t WBp = std::max((WB + 3u) & ~3u, 4u);      // adjacency width in entries, chunks of 4
    if (WBp != s->WB || !s->badj.p) { HIPCHK(s->badj.resize((size_t)WBp * RNp)); }
    if (has_bend && !s->chain.p) { HIPCHK(s->chain.resize(RNp)); HIPCHK(s->bendE.resize(RNp)); }
    s->WB = WBp; s->has_bend = has_bend; s->has_bonds = !all.empty();
    s->mob_uniform = (float)s->mob[0];
    for (uint32_t i = 1; i < N; i++) if ((float)s->mob[i] != s->mob_uniform) { s->mob_uniform = -1.f; break; }
    s->mob_max = 0;
    for (uint32_t i = 0; i < N; i++) s->mob_max = std::max(s->mob_max, s->mob[i]);
    s->topo_dirty = false; s->list_valid = false; s->w_packed = false;
    return GD_OK;
}

// ------------------------------------------------------------ list builds

static double scale_at(const gd_system *s, double init, double tau, double time) { (void)s; return 1.0 - (1.0 - init) * std::exp(-time / tau); }

// bead_scale bound over the next `ahead` steps (monotone in time)
static double bead_scale_bound(const gd_system *s, const gd_run_desc *run, uint32_t ahead)
{
    double m = 0;
    for (auto &c : s->hctx) {
        m = std::max(m, c.bead_scale);
        if (run && (run->flags & GD_RUN_UPDATE_SCALES) && s->has_scaling)
            m = std::max(m, scale_at(s, s->bs_init, s->bs_tau, (double)(c.step + ahead + 1) * run->timestep));
    }
    return m;
}

static void fill_common(gd_system *s, StepParams &p)
{
    memset(&p, 0, sizeof p);
    p.N = s->N; p.Np = s->Np; p.R = s->R; p.nblk = s->nblk; p.stride = (size_t)s->R * s->Np;
    p.periodic = s->box_kind == GD_BOX_PERIODIC;
    for (int k = 0; k < 3; k++) { p.box[k] = (float)s->box[k]; p.inv_box[k] = s->box[k] > 0 ? (float)(1.0 / s->box[k]) : 0.f; }
    p.pos_in = s->pos[s->pcur].p; p.pos_out = s->pos[s->pcur ^ 1].p; p.xb = s->xb.p; p.orig = s->orig[s->ocur].p;
    p.ab = s->ab.p; p.mob = s->mobs.p; p.bendE = s->bendE.p; p.mob_uniform = s->mob_uniform; p.WB = s->WB;
    p.nbr = s->nbr.p; p.nbr16 = s->nbr16.p; p.wtab = s->wtab.p; p.tiles = s->tiles.p; p.tiled = s->list_tiled ? 1 : 0; p.packed_ab = s->packed_ab ? 1 : 0;
    p.cpb = s->cpb; p.tile_cap = s->list_tiled ? s->list_tile_cap : s->tile_cap;   // as at the build of the list in use
    p.pk = (s->has_pair && s->pair.p_a == 2 && s->pair.q_a == 3 && s->pair.p_b == 8 && s->pair.q_b == 3) ? (s->pair.mix ? 1 : 2) : 0;
    p.meta = s->meta.p; p.rec_x0 = s->rec_x0.p; p.rec_mo = s->rec_mo.p; p.W = s->list_W; p.badj = s->badj.p; p.chain = s->chain.p;
    p.ctx_in = s->ctx[s->ccur].p; p.ctx_out = s->ctx[s->ccur ^ 1].p; p.flags = s->flags.p;
    // wall-reaction partials ping-pong with the context: a launch reads the previous step's partials while its blocks
    // write this step's (one buffer would let late blocks read a mix of two steps)
    p.react_in = s->react_part[s->ccur].p; p.react_out = s->react_part[s->ccur ^ 1].p;
    if (s->has_pair) {
        const gd_pair_softcore &q = s->pair;
        p.pair = PairP{(float)q.eps_a, (float)q.sigma_a, (float)q.eps_b, (float)q.sigma_b, q.p_a, q.q_a, q.p_b, q.q_b,
                       q.mix, q.scale_by_bead_scale, pair_cutoff(s) > 0 ? 1 : 0, pair_cutoff(s)};
    }
    if (s->has_wall) {
        const gd_wall &w = s->wall;
        p.wall.eps_a = (float)w.eps_a; p.wall.sigma_a = (float)w.sigma_a; p.wall.eps_b = (float)w.eps_b; p.wall.sigma_b = (float)w.sigma_b;
        p.wall.p_a = w.p_a; p.wall.q_a = w.q_a; p.wall.p_b = w.p_b; p.wall.q_b = w.q_b;
        p.wall.wall_a = (float)w.wall_a_factor; p.wall.wall_b = (float)w.wall_b_factor; p.wall.scaled = w.scale_by_bead_scale;
        p.wall.enabled = 1; p.wall.packing_spring = (float)w.packing_spring;
        p.wall.fast2383 = (w.p_a == 2 && w.q_a == 3 && w.p_b == 8 && w.q_b == 3) ? 1 : 0;
        for (int k = 0; k < 3; k++) p.wall.spring[k] = w.semiaxes_spring[k];
        p.wall.mobility = w.mobility;
    }
    if (s->has_inner) {
        const gd_inner_sphere &w = s->inner;
        p.wall.inner_enabled = 1; p.wall.in_radius = (float)w.radius;
        p.wall.in_eps_a = (float)w.eps_a; p.wall.in_sigma_a = (float)w.sigma_a; p.wall.in_eps_b = (float)w.eps_b; p.wall.in_sigma_b = (float)w.sigma_b;
        p.wall.in_p_a = w.p_a; p.wall.in_q_a = w.q_a; p.wall.in_p_b = w.p_b; p.wall.in_q_b = w.q_b;
        p.wall.in_wall_a = (float)w.wall_a_factor; p.wall.in_wall_b = (float)w.wall_b_factor; p.wall.in_spring = (float)w.spring;
    }
    p.scaling = ScaleP{s->has_scaling ? 1 : 0, 0, s->bs_init, s->bs_tau, s->bo_init, s->bo_tau, 0.0, 0.0};
    p.btab = s->btab.p; p.nbt = (int)s->n_bond_types; p.has_softcore_bonds = s->has_softcore_bonds ? 1 : 0;
    p.bonds_premixed = s->bonds_premixed ? 1 : 0; p.bonds_all_scaled = s->bonds_all_scaled ? 1 : 0;
    p.nps = (int)s->psrc.size();
    for (int q = 0; q < p.nps; q++) {
        p.ps[q].kind = s->psrc[q].kind; p.ps[q].k = (float)s->psrc[q].k; p.ps[q].b = (float)s->psrc[q].b;
        for (int k = 0; k < 3; k++) p.ps[q].p[k] = (float)s->psrc[q].p[k];
    }
    p.has_bend = s->has_bend; p.has_bonds = s->has_bonds;
    p.rv = s->rv; p.rn = s->sw_n ? 0.f : s->rn; p.dmax = s->dmax.p; p.term_mask = GD_TERM_ALL;      // (the droplet kernel moves beads after k_step has bounded their displacement: both list classes then)
    if (dev_env("GDYN_FORCE_FAR")) p.rn = 0.f;      // (timing experiments: the far class in every step)
    if (dev_env("GDYN_FORCE_NEAR")) p.rn = 1e3f;    // (timing experiments with gd_debug_bench only: never the far class -- wrong forces late in an interval)
    p.fout = s->fout.p; p.epart = s->epart.p;
    p.lo = s->lo.p; p.comp = 0;
}

// Enqueue one list build (counting sort into slot order + ELL fill) with radius rv.
static bool want_tiled(const gd_system *s)
{
    return s->kernel_path != 1 && s->tiled_ok && s->packed_ab && s->W <= GD_TILED_MAX_W;      // open and periodic boxes alike
}

static int enqueue_build(gd_system *s, float rv, bool with_list, bool allow_tiled = true)
{
    const bool tiled = with_list && allow_tiled && want_tiled(s);
    if (with_list) {
        if (s->W == 0) s->W = 96;
        s->W = (s->W + GD_UNROLL - 1) & ~(GD_UNROLL - 1);
        // generic lists: uniform rows of W entries (chunked wave-interleaved layout, k_step), grown on demand -- with an eighth to spare
        // once the rows are long -- and given back when a dense transient has passed.  Tiled lists: ragged rows from a pool, below.
        const size_t need = (size_t)s->W * s->R * s->Np;
        const size_t grow = s->W >= 512 ? need + need / 8 : need;
        if (!tiled && (s->nbr.n < need || s->nbr.n > 4 * need)) HIPCHK(s->nbr.resize(grow, false));
        s->list_W = s->W;
    }
    BuildParams b;
    memset(&b, 0, sizeof b);
    b.N = s->N; b.Np = s->Np; b.R = s->R; b.nblk = s->nblk; b.stride = (size_t)s->R * s->Np;
    b.periodic = s->box_kind == GD_BOX_PERIODIC;
    for (int k = 0; k < 3; k++) { b.box[k] = (float)s->box[k]; b.inv_box[k] = s->box[k] > 0 ? (float)(1.0 / s->box[k]) : 0.f; }
    b.rv = rv; b.ncell_cap = s->ncell_cap; b.dmax = s->dmax.p;
    b.kx = b.periodic ? 1 : 2;
    if (const char *e = dev_env("GDYN_KX")) b.kx = b.periodic ? 1 : std::max(1, atoi(e));      // (experiments: cells per list radius in x)
    b.scan_segments = std::min((s->ncell_seen + s->ncell_seen / 4 + 8191u) / 8192u, (s->ncell_cap + 8191u) / 8192u);      // (0 before the first build: one block per replica)
    {   // near-class radius: the (look-ahead) cutoff the list radius was derived from, plus a share of the skin
        const float cutb = rv - (float)(pair_cutoff(s) * s->skin);
        b.rn = (cutb > 0.f && cutb < rv && !s->all_near) ? cutb + (float)s->near_frac * (rv - cutb) : rv;
    }
    b.pos_in = s->pos[s->pcur].p; b.pos_out = s->pos[s->pcur ^ 1].p; b.xb = s->xb.p;
    b.orig_in = s->orig[s->ocur].p; b.orig_out = s->orig[s->ocur ^ 1].p; b.slot_of = s->slot_of.p;
    b.rank = s->rank.p; b.members = s->members.p; b.cell_cnt = s->cell_cnt.p; b.cell_start = s->cell_start.p;
    b.bbox = s->bbox.p; b.grid = s->grid.p;
    b.bbox_cur = s->bbox_enc.p + (size_t)s->bbox_cur * s->R * 6; b.bbox_next = s->bbox_enc.p + (size_t)(s->bbox_cur ^ 1) * s->R * 6;
    b.warm = (b.periodic || s->bbox_valid) ? 1 : 0; b.bbox_w = s->bbox_w.p;
    b.ab_o = s->ab_o.p; b.mob_o = s->mob_o.p; b.bendE_o = s->bendE_o.p; b.psmask_o = s->psmask_o.p;
    b.badj_o = s->badj_o.p; b.bdeg_o = s->bdeg_o.p; b.chain_o = s->has_bend ? s->chain_o.p : nullptr; b.WB = s->WB;
    b.ab = s->ab.p; b.mob = s->mobs.p; b.bendE = s->bendE.p; b.badj = s->badj.p; b.has_bend = s->has_bend ? 1 : 0;
    b.mob_is_uniform = s->mob_uniform >= 0.f ? 1 : 0;
    b.chain = s->chain.p; b.nbr = (with_list && !tiled) ? s->nbr.p : nullptr; b.nbr16 = tiled ? s->nbr16.p : nullptr;
    b.meta = s->meta.p; b.rec_x0 = s->rec_x0.p; b.rec_mo = s->rec_mo.p; b.len_prev = s->len_prev.p; b.W = s->W; b.tiles = s->tiles.p; b.tiled = tiled ? 1 : 0;
    b.packed_ab = s->packed_ab ? 1 : 0; b.cpb = s->cpb; b.tile_cap = s->tile_cap;
    b.w_valid = (s->packed_ab && s->w_packed) ? 1 : 0;
    b.flags = s->flags.p; b.lcount = s->lcount_d.p; b.dbg = (unsigned long long *)s->fout.p;
    b.wtab = s->wtab.p; b.need_prev = s->need_prev.p; b.pool = s->pool.p; b.rqueue = s->rqueue.p; b.rq_cap = (unsigned)s->rqueue.n; b.rq_grid = s->repair_wide > 0 ? b.rq_cap : std::min(GD_REPAIR_GRID, b.rq_cap);
    if (tiled) {
        // Ragged rows (BuildParams): every k_step wave's rows are as wide as its longest list, predicted from what each bead needed at
        // the build before (no history -- first build, positions from the caller, another list radius or class mode: W entries per
        // bead) and repaired inside k_fill where a list outgrows the prediction.  The pool (nbr16) is sized from the use of the last
        // build with an eighth + a KiB per wave to spare (the use is read back with every chunk; the builds in between grow with the
        // lists); a pool that turns out too small is flagged, its cursor has counted the need, and the chunk is rolled back.
        const size_t waves = (size_t)s->R * s->Np / 64;
        const bool predict = s->need_valid && s->need_all_near == s->all_near && s->need_rv > 0 && std::fabs(rv / s->need_rv - 1.f) <= 0.02f;
        auto pool_kib = [&]() { return (size_t)(s->nbr16.n / 512); };
        const size_t used = predict ? std::max<size_t>(s->pool_used, waves) : std::max<size_t>(s->pool_used, waves * (s->W / 8));
        const size_t want = used + used / 8 + 2 * waves;
        if (pool_kib() < want || pool_kib() > 2 * want + 4 * waves) HIPCHK(s->nbr16.resize((want + want / 16) * 512, false));      // (not preserved: the list in it is about to be rebuilt)
        if (dev_env("GDYN_DEBUG") && dev_env("GDYN_DEBUG")[0] == '2')
            fprintf(stderr, "[gdyn] build %llu: %s, rows used %u KiB, pool %zu KiB, rv %.4f\n", (unsigned long long)s->rebuilds, predict ? "predicted" : "no history", s->pool_used, pool_kib(), rv);
        if (!predict) s->pool_used = (uint32_t)std::min<size_t>(used, 0xffffffffu);      // (the guess stands in until a chunk's readback brings the real use)
        b.predict = predict ? 1 : 0; b.nbr16 = s->nbr16.p; b.pool_cap = (unsigned)std::min<size_t>(pool_kib(), 0xffffffffu);
        s->need_valid = true; s->need_rv = rv; s->need_all_near = s->all_near;
    }
    gd_launch_build(b, s->stream);
    s->bbox_cur ^= 1; s->bbox_valid = tiled;      // (the box of the positions this build sorted, reduced by k_tiles: the next build's grid)
    s->list_tiled = tiled; s->list_tile_cap = s->tile_cap;
    s->w_packed = s->packed_ab;
    s->pcur ^= 1; s->ocur ^= 1;
    s->rv = rv; s->rn = b.rn; s->steps_since_build = 0; s->rebuilds++;
    s->timing.rebuild_launches++;
    return GD_OK;
}

static int read_flags(gd_system *s, std::vector<unsigned> &f)
{
    HIPCHK(hipGetLastError());
    f.resize((size_t)s->R * GD_NFLAGS);
    HIPCHK(hipMemcpyAsync(f.data(), s->flags.p, f.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
    unsigned used[4] = {0u, 0u, 0u, 0u};
    HIPCHK(hipMemcpyAsync(used, s->pool.p, sizeof used, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->list_tiled && used[0] > 0) { s->pool_used = std::max(used[0], used[1]); s->repairs = used[2]; }
    return GD_OK;
}
static int clear_flags(gd_system *s)
{
    HIPCHK(hipMemsetAsync(s->flags.p, 0, (size_t)s->R * GD_NFLAGS * sizeof(unsigned), s->stream));
    HIPCHK(hipMemsetAsync(s->pool.p + 1, 0, 2 * sizeof(unsigned), s->stream));      // the row pool's largest use and repair count of the builds to come
    return GD_OK;
}

// Tile capacities (float4 entries) at which k_step still fits 3, 2, 1 blocks into the 160 KB of LDS of a CU
// (1.2 KB static LDS per block on top of the tile).
static unsigned pick_tile_cap(unsigned need)
{
    // LDS is granted in 1280-byte granules (measured with 1184 B of static LDS: 3264 entries fit 3 blocks, 3318 do not; 720 B now)
    static const std::vector<unsigned> caps = [] {      // (initialised once, thread-safe; experiment hook of developer builds: GDYN_TILE_CAPS=a,b,c)
        std::vector<unsigned> c = {3312u, 4080u, 5072u, 8192u};      // (4080: the largest tile with byte-offset list entries)
        if (const char *e = dev_env("GDYN_TILE_CAPS")) { c.clear(); for (const char *q = e; *q;) { c.push_back((unsigned)strtoul(q, (char **)&q, 10)); if (*q == ',') q++; } }
        return c;
    }();
    for (unsigned c : caps) if (need <= c) return c;
    return need;     // > 8192: the caller falls back to the generic path
}

// A build that meets a dense state -- the spline-refined start of the pipeline is a globule in which some beads have 1 500
// neighbours inside the default list radius.  Tiled lists have ragged rows: the pool holds what the waves need, not the longest list
// times every bead (uniform rows took 18-20 GB at 128 x 30 000 beads there; the pool a fifth of it), so this guard is a safety net:
// only when the rows exceed a sixteenth of the device memory is the list width narrowed so that they fit (lists grow with the cube of
// the radius; at least a skin of 0.15 x cutoff), and class_skin returns to the width it left once the longest list, scaled back, fits
// again.  Not with a caller-chosen skin.
static void dense_guard(gd_system *s, unsigned need_w)
{
    if (s->skin_fixed || need_w <= 512u || !(s->rv > 0)) return;
    const double cut = pair_cutoff(s);
    if (!(cut > 0)) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return;
    // what the rows of the build take: tiled lists the pool's use (ragged rows: the sum of what the waves need), generic lists uniform
    // rows of the longest list at four bytes per entry
    const double rows = (double)s->R * (double)s->Np, budget_b = (double)(total_b / 16);
    const double bytes = s->list_tiled ? 1024.0 * (double)s->pool_used : (double)need_w * 4.0 * rows;
    if (bytes <= budget_b) return;
    const double shrink = 0.9 * budget_b / bytes;                 // lists grow with the cube of the radius
    s->dense_budget = (uint32_t)std::max(64.0, (double)need_w * budget_b / bytes);
    const double sc = s->rv / cut - s->skin;                       // bead-scale part of the radius the build used
    const double r_new = s->rv * std::cbrt(shrink);
    const double skin_new = std::max(0.15, r_new / cut - sc);
    if (skin_new < s->skin - 1e-9) {
        if (!(s->skin_dense_from > 0)) s->skin_dense_from = s->skin;
        s->skin = skin_new; s->skin_next = 0; s->a2_ema = 0;
        if (s->adapt) s->K = std::max(1u, std::min(s->K, 4u));      // (a caller-fixed interval stays the caller's)
        if (!s->list_tiled) s->W = std::max(64u, s->dense_budget & ~7u);      // (the narrowed list is predicted at 0.9 of the budget; a miss is one more exactly sized build)
        if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] dense state (longest list %u, rows %.1f GB): skin %.3f\n", need_w, bytes / 1e9, skin_new);
    }
}

// (class_skin, below: the handle runs at the wider of its two list widths)
static bool class_skin_wide(const gd_system *s)
{
    return !s->skin_fixed && s->adapt && !s->tuner.enabled && !(s->skin_dense_from > 0) && !s->sw_n && s->skin >= 0.9 - 1e-9 && s->skin <= 0.9 + 1e-9;
}

// React to list-width / tile-capacity overflow flags: widen the list, enlarge the LDS tile or
// fall back to the generic path. Returns true when a build has to be redone.
static bool handle_overflow(gd_system *s, const std::vector<unsigned> &f)
{
    unsigned need_w = 0, need_t = 0; bool over = false, tover = false, class_over = false;
    for (uint32_t r = 0; r < s->R; r++) {
        over |= f[r * GD_NFLAGS + GD_FLAG_OVERFLOW] != 0; need_w = std::max(need_w, f[r * GD_NFLAGS + GD_FLAG_NEED_W]);
        class_over |= (f[r * GD_NFLAGS + GD_FLAG_OVERFLOW] & 2u) != 0;
        tover |= f[r * GD_NFLAGS + GD_FLAG_TILE_OVERFLOW] != 0; need_t = std::max(need_t, f[r * GD_NFLAGS + GD_FLAG_NEED_TILE]);
    }
    if (need_t > 0 && need_t < (1u << 20)) s->last_need_t = need_t;
    if (need_w > 0) s->last_need_w = need_w;
    {
        unsigned nc = 0;
        for (uint32_t r = 0; r < s->R; r++) nc = std::max(nc, f[r * GD_NFLAGS + GD_FLAG_NCELL]);
        if (nc > 0) s->ncell_seen = nc;
    }
    if (!tover && !over && s->list_tiled && need_t > 0) {
        // size the LDS tile to what the builds actually need (more resident blocks per CU)
        // LDS capacity is a step function of the tile size: k_step keeps 3 / 2 / 1 blocks (6 / 4 / 2 waves per SIMD)
        // resident per CU up to these tile capacities, so the capacity is always the largest one of its occupancy class
        // (the tiles of consecutive builds differ by a few entries; an overflow costs one rolled-back chunk and then
        // keeps the larger class for a while, so the margin for the smaller class can be thin)
        unsigned want = pick_tile_cap(need_t + 24);
        if (want < s->tile_cap && s->tile_hold > 0) { s->tile_hold--; want = s->tile_cap; }
        // at the width class_skin selected the tiles are about to leave the three-block class: class_skin (called after this) takes
        // the narrower list back at the next build, where they fit it -- no detour through the two-block class
        if (want > 3312u && s->tile_cap <= 3312u && class_skin_wide(s)) want = s->tile_cap;
        if (want != s->tile_cap && want <= 8192u) {
            if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] tile capacity %u -> %u (largest tile %u)\n", s->tile_cap, want, need_t);
            s->tile_cap = want;
        }
    }
    if (tover) {
        const unsigned cap = pick_tile_cap(need_t + need_t / 32 + 32);
        // 128 KB dynamic + static part < 160 KB of LDS per CU (one resident block per CU at the largest class: still ahead of the
        // generic path's global gathers -- S-1kb-250k x 16: 471 us per step generic, 239 us tiled at two blocks per CU)
        const unsigned cap_max = 8192u;
        if (cap <= cap_max) { s->tile_cap = cap; s->tile_hold = 4; }
        else {
            // Too dense for one tile at this list width (the spline-refined start of the pipeline: a globule whose core holds thousands
            // of beads per list sphere).  A tile is the block's own cells plus their neighbour cells: it shrinks about with the square of
            // the list radius -- so the width is narrowed until the largest tile fits the LDS (at least a skin of 0.15 x cutoff; builds
            // get more frequent, but the state stays on the tiled path: global-gather lists cost 2-2.4 x per step there and uniform rows
            // of the longest list), and class_skin returns to the width it left once the largest tile, scaled back, fits again.  Only
            // when that is not enough -- or the caller pinned the width -- the generic path takes over (retried later with back-off).
            const double cut = pair_cutoff(s);
            bool narrowed = false;
            if (!s->skin_fixed && cut > 0 && s->rv > 0 && need_t < (1u << 20)) {
                const double sc = s->rv / cut - s->skin;
                const double ratio = std::min(0.97, std::max(0.5, std::sqrt(0.85 * (double)cap_max / (double)need_t)));
                const double skin_new = std::max(0.15, s->rv * ratio / cut - sc);
                if (skin_new < s->skin - 1e-9) {
                    if (!(s->skin_dense_from > 0)) s->skin_dense_from = s->skin;
                    s->skin = skin_new; s->skin_next = 0; s->a2_ema = 0; s->dense_by_tile = true;
                    if (s->adapt) s->K = std::max(1u, std::min(s->K, 4u));
                    s->tile_cap = cap_max; s->tile_hold = 4;
                    narrowed = true;
                    if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] dense state (largest tile %u): skin %.3f\n", need_t, skin_new);
                }
            }
            if (!narrowed) { s->tiled_ok = false; s->tiled_off = 1; }
        }
    }
    if (over) {
        // Tiled lists (ragged rows): the pool was full -- its cursor counted on, pool_used is the need and the next build sizes the pool
        // from it -- or a build queued more waves for repair than the repair launch has blocks: the next chunks launch one per wave
        // (a list that outgrows its predicted row is repaired on the device and never gets here).  Generic lists (uniform
        // rows): the overflowing build has counted the longest list exactly (a row keeps counting past its width): the next build
        // gets that width with 6 % to spare.
        {
            unsigned bits = 0;
            for (uint32_t r = 0; r < s->R; r++) bits |= f[r * GD_NFLAGS + GD_FLAG_OVERFLOW];
            if (s->list_tiled && (bits & 4u) && (size_t)s->pool_used <= s->nbr16.n / 512) s->repair_wide = 16;      // (bit 4 with room in the pool: the repair queue)
        }
        if (!s->list_tiled) {
            unsigned w = need_w + need_w / 16 + 8;
            s->W = std::max(w, s->W + 8);
        }
        if (class_over) {
            // a class beyond its field of the tiled record: single-class lists while that is the far class; a near class beyond
            // 8 184 entries is beyond tiled rows
            if (!s->all_near && need_w <= GD_TILED_MAX_NEAR) s->all_near = true;
            else s->W = std::max(s->W, GD_TILED_MAX_W + 8u);
        }
        dense_guard(s, need_w);
    }
    else if (!tover && need_w > 0) {
        if (s->list_tiled && (size_t)s->pool_used * 1024u > ((size_t)4 << 30)) dense_guard(s, need_w);      // (rows of several GB: within the budget?)
        if (s->all_near && need_w <= GD_TILED_MAX_FAR) s->all_near = false;      // (no far class can overflow its field any more; from the next build)
        // the longest list is reported by every build: generic lists give the row width back when a dense transient has passed
        const unsigned want_w = std::max(64u, (need_w + need_w / 4 + 16 + GD_UNROLL - 1) & ~(GD_UNROLL - 1));
        if (!s->list_tiled && 2 * want_w <= s->W) s->W = want_w;       // (takes effect at the next build; the list in use keeps list_W)
        else if (s->list_tiled && s->W > GD_TILED_MAX_W && need_w <= GD_TILED_MAX_NEAR) s->W = 96;      // (a near class beyond the tiled record has passed)
    }
    if (tover && dev_env("GDYN_DEBUG")) {
        std::vector<GridP> gp(s->R);
        (void)hipMemcpy(gp.data(), s->grid.p, s->R * sizeof(GridP), hipMemcpyDeviceToHost);
        for (uint32_t r = 0; r < std::min(s->R, 3u); r++)
            fprintf(stderr, "[gdyn] grid r%u: nc %d %d %d ncell %d org %g %g %g inv %g flags need_t %u\n", r, gp[r].nc[0], gp[r].nc[1], gp[r].nc[2],
                    gp[r].ncell, gp[r].org[0], gp[r].org[1], gp[r].org[2], gp[r].inv[0], f[r * GD_NFLAGS + GD_FLAG_NEED_TILE]);
    }
    if ((over || tover) && dev_env("GDYN_DEBUG"))
    {
        unsigned bits = 0;
        for (uint32_t r = 0; r < s->R; r++) bits |= f[r * GD_NFLAGS + GD_FLAG_OVERFLOW];
        fprintf(stderr, "[gdyn] overflow: list %d (bits %u, need %u -> W %u; rows %u KiB of a pool of %zu), tile %d (need %u -> cap %u, tiled_ok %d)\n", (int)over, bits,
                need_w, s->W, s->pool_used, (size_t)(s->nbr16.n / 512), (int)tover, need_t, s->tile_cap, (int)s->tiled_ok);
    }
    return over || tover;
}

// Synchronous build used outside gd_run: grows the list width until nothing overflows.
static int build_now(gd_system *s, float rv, bool with_list, bool allow_tiled = true, float rv_min = 0.f)
{
    const double skin0 = s->skin;
    for (int attempt = 0; attempt < 10; attempt++) {
        GDCHK(clear_flags(s));
        // (a retry after the dense guard has narrowed the width builds at the narrowed radius -- not below what the caller needs
        // the list to cover, rv_min: a pair search at a contact distance beyond the force cutoff)
        const float rv_try = std::max(rv_min, rv - (float)(pair_cutoff(s) * (skin0 - s->skin)));
        GDCHK(enqueue_build(s, rv_try, with_list, allow_tiled));
        std::vector<unsigned> f;
        GDCHK(read_flags(s, f));
        if (!handle_overflow(s, f)) {
            HIPCHK(hipMemcpy(s->lcount.data(), s->lcount_d.p, 2 * (size_t)s->R * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            GDCHK(clear_flags(s));
            return GD_OK;
        }
    }
    return fail(GD_ENOMEM, "neighbour list width did not converge (W=%u)", s->W);
}

static int prepare(gd_system *s)
{
    HIPCHK(hipSetDevice(s->device));
    GDCHK(finalize_topology(s));
    GDCHK(upload_ctx(s));
    return GD_OK;
}

static float list_radius(gd_system *s, const gd_run_desc *run, uint32_t ahead)
{
    const float cut = pair_cutoff(s);
    if (!(cut > 0)) return 1.0f;
    const double sc = s->pair.scale_by_bead_scale ? bead_scale_bound(s, run, ahead) : 1.0;
    // the skin is an absolute width, `skin` x the NOMINAL cutoff: with a scaled-down cutoff (bead_scale < 1 early in the
    // interphase run, simulation_driver_forcefield.cc:47-49) the displacement budget (rv - cutoff) / 2, and with it the
    // rebuild interval, stays what it is at full scale
    return (float)(cut * (sc + s->skin));
}

// The width selected by tile class (class_skin) takes over at a list build: interval and radius change together
static uint32_t interval_for_skin(const gd_system *s, double skin);
static void take_pending_skin(gd_system *s)
{
    if (!(s->skin_next > 0)) return;
    // the interval follows the square of the skin (diffusive displacements) from the interval the handle had adapted to at the old width --
    // a tenth off on the way up -- and not beyond what the measured displacement rate admits: the rate alone, averaged over a few
    // intervals of a state that had just changed its regime, put S-genome-30k without second bonds at 23 steps where 19-20 hold
    // (one rolled-back chunk in the timed steps of that line)
    const double ratio = s->skin > 0 ? s->skin_next / s->skin : 1.0;
    const uint32_t k_scaled = (uint32_t)std::max(1.0, std::floor((double)s->K * ratio * ratio * (ratio > 1.0 ? 0.9 : 1.0)));
    s->skin = s->skin_next; s->skin_next = 0;
    s->K = std::min(interval_for_skin(s, s->skin), std::max(k_scaled, 1u)); s->K_bad_ttl = 0;
}

static int ensure_fresh_list(gd_system *s)
{
    if (s->list_valid && (s->steps_since_build == 0 || s->verified_serial == s->state_serial)) return GD_OK;
    take_pending_skin(s);
    GDCHK(build_now(s, list_radius(s, nullptr, 0), pair_cutoff(s) > 0));
    s->list_valid = true; s->search_list = false;
    return GD_OK;
}

// ---------------------------------------------------------------- stepping

static hipEvent_t get_event(gd_system *s, size_t i)
{
    // (timing events only: no system-scope fence when one is recorded -- the cache write-back and invalidation of the default event
    // idle the device for ~5 us between the kernels on either side; what the host reads of a chunk it reads behind hipStreamSynchronize)
    while (s->events.size() <= i) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr; s->events.push_back(e); }
    return s->events[i];
}

// The state updates a GD_RUN_DEFER_CALLBACK run left pending: one k_ctx launch with that run's timestep and flags (the wall-reaction
// partials of its last step are still the current ones), then the host mirror.
static int apply_pending(gd_system *s)
{
    bool any = false;
    for (auto &c : s->hctx) any |= c.pending != 0;
    if (!any) return GD_OK;
    HIPCHK(hipSetDevice(s->device));
    GDCHK(upload_ctx(s));
    StepParams p;
    fill_common(s, p);
    p.dt_d = s->pend_dt; p.dt = (float)s->pend_dt; p.run_flags = s->pend_flags;
    bool common = s->has_scaling && (s->pend_flags & GD_RUN_UPDATE_SCALES);
    for (uint32_t r = 1; r < s->R && common; r++) common = s->hctx[r].step == s->hctx[0].step && s->hctx[r].pending == s->hctx[0].pending;
    if (common) {      // same libm exp as a callback applied inside gd_run (see host_scales there)
        const double time = (double)(s->hctx[0].step + 1) * s->pend_dt;
        p.scaling.from_host = 1;
        p.scaling.bead_next = scale_at(s, s->bs_init, s->bs_tau, time);
        p.scaling.bond_next = scale_at(s, s->bo_init, s->bo_tau, time);
    }
    gd_launch_finalize(p, 0, s->stream);
    s->ccur ^= 1;
    GDCHK(download_ctx(s));
    s->state_serial++;
    return GD_OK;
}

// Interval the skin admits at the measured displacement rate (the adaptation formula of gd_run)
static uint32_t interval_for_skin(const gd_system *s, double skin)
{
    if (!(s->a2_ema > 0)) return s->K;
    const double lim = 0.5 * pair_cutoff(s) * skin, k = 0.90 * lim * 0.90 * lim / s->a2_ema;
    return (uint32_t)std::max(1.0, std::min(200.0, std::floor(k)));
}

// One accepted chunk of `steps` steps took `ms` on the device.  Candidates: the width in use, 1.2 x it (fewer builds: what a small
// launch-latency-bound system wants) and 0.7 / 0.5 / 0.35 of it (shorter lists and smaller LDS tiles against more frequent builds;
// a periodic box whose tiles did not fit may fit them at a smaller width).  Each candidate: one chunk to settle (tile class, list width, interval), three measured.  The width the sweep
// started from is left only for a gain of 6 % or more (chunk times scatter by a few per cent).
static void tune_skin(gd_system *s, double ms, int64_t steps, bool full_interval, bool rolled_back)
{
    auto &t = s->tuner;
    if (!t.enabled || s->sw_n) return;
    if (s->skin_dense_from > 0 || s->skin_next > 0) return;      // a dense transient at its narrow width (dense_guard): the selection starts once it has passed
    if (rolled_back) { t.settle = std::max(t.settle, 1); t.acc_ms = 0; t.acc_steps = 0; t.measured = 0; return; }
    if (!full_interval || !(s->a2_ema > 0)) return;
    if (t.done) {      // conditions drift (a relaxation, a growing bead scale): look again, around the width in use, once the
                       // rebuild interval -- the displacement rate -- has moved by a third since the last sweep
        // ... or the tiles have outgrown the class the width was selected in (fewer resident blocks per CU: another width may
        // fit the smaller class)
        if (t.wait > 0) t.wait--;
        if (t.cap_ref == 0 && t.wait <= 45 && s->list_tiled) t.cap_ref = std::max(s->list_tile_cap, 3312u);
        const double k = (double)s->K, k0 = (double)std::max(t.K_ref, 1u);
        const bool outgrown = s->list_tiled && t.cap_ref > 0 && s->list_tile_cap > t.cap_ref;
        if (!(outgrown && t.wait <= 40) && (t.wait > 0 || (k < 1.33 * k0 && k0 < 1.33 * k))) return;
        t.done = false; t.cand.clear();
    }
    if (t.cand.empty()) {
        if (t.wait > 0 && t.rounds == 0) { t.wait--; return; }
        if (t.rounds == 0) t.cand = {s->skin, std::min(1.2 * s->skin, 1.0), 0.7 * s->skin, 0.5 * s->skin, 0.35 * s->skin};
        else t.cand = {s->skin, std::min(1.2 * s->skin, 1.0), 0.85 * s->skin, 0.7 * s->skin};      // (finer steps around the width in use)
        t.cost.assign(t.cand.size(), 0.0);
        t.idx = 0; t.settle = 0; t.measured = 0; t.acc_ms = 0; t.acc_steps = 0; t.rounds++;
    }
    if (t.settle > 0) { t.settle--; return; }
    t.acc_ms += ms; t.acc_steps += (uint64_t)steps; t.measured++;
    if (t.measured < 3) return;
    t.cost[t.idx] = t.acc_ms / (double)t.acc_steps;
    if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] skin %.3f: %.4f ms per step (K %u, %s, W %u, tile %u)\n", t.cand[t.idx], t.cost[t.idx], s->K,
                                       s->list_tiled ? "tiled" : "generic", s->list_W, s->list_tile_cap);
    size_t next = t.idx + 1;
    while (next < t.cand.size() && (t.cand[next] == s->skin || interval_for_skin(s, t.cand[next]) < 2)) next++;
    if (next >= t.cand.size()) {      // sweep complete: the cheapest, but the width the sweep started from unless the gain is 6 % or more
        size_t best = 0;
        for (size_t k = 1; k < t.cand.size(); k++) if (t.cost[k] > 0 && t.cost[k] < 0.94 * t.cost[0] && t.cost[k] < t.cost[best]) best = k;
        next = best; t.done = true; t.wait = 50; t.K_ref = interval_for_skin(s, t.cand[best]);
        t.cap_ref = 0;      // (taken a few chunks on, once the selected width has found its class)
    }
    if (t.cand[next] != s->skin) {
        // the tile class for the new width: tiles scale about with the square of the list radius (rows of cells x their
        // neighbour rows); sized from the last build's largest tile so that the candidate is not measured in a class it does
        // not need (or found by overflow and rollback)
        if (s->last_need_t > 0 && s->rv > 0) {
            const double cut = pair_cutoff(s), r0 = cut * (1.0 + s->skin), r1 = cut * (1.0 + t.cand[next]);
            const unsigned est = (unsigned)(1.08 * s->last_need_t * (r1 / r0) * (r1 / r0)) + 32u;
            s->tile_cap = std::min(pick_tile_cap(est), 8192u); s->tile_hold = 0;
        }
        {      // (the interval as in take_pending_skin: from the one adapted to at the width in use, not beyond the measured rate)
            const double ratio = s->skin > 0 ? t.cand[next] / s->skin : 1.0;
            const uint32_t k_scaled = (uint32_t)std::max(1.0, std::floor((double)s->K * ratio * ratio * (ratio > 1.0 ? 0.9 : 1.0)));
            s->skin = t.cand[next];
            s->K = std::min(interval_for_skin(s, s->skin), k_scaled); s->K_bad_ttl = 0;
        }
        s->list_valid = false;
        if (s->kernel_path != 1 && s->packed_ab) { s->tiled_ok = true; s->tiled_off = 0; }      // smaller tiles may fit now
    }
    t.idx = next; t.settle = 1; t.measured = 0; t.acc_ms = 0; t.acc_steps = 0;
}

// Whether a run steps with the compensated position update (k_step's p.comp).  The increment of a step is mu F dt + sigma xi with
// sigma = sqrt(2 mu kT dt); once sigma is within a few dozen ulp of an fp32 coordinate -- always at T = 0 -- the rounding of x + dx is no
// longer small against what a step adds, and summed over a run it is a systematic loss (simulation_fine_sampling: T = 0, dt = 1e-7; an
// fp32 coordinate of 3 ... 8 has an ulp of 2.4e-7 ... 4.8e-7).  Criterion: sigma < 64 ulp(X), X the coordinate range (largest wall
// semiaxis / box period; 16 when the model has neither).  GD_RUN_COMPENSATED / GD_RUN_UNCOMPENSATED override it.
static bool want_compensated(const gd_system *s, const gd_run_desc *run)
{
    if (run->flags & GD_RUN_UNCOMPENSATED) return false;
    if (run->flags & GD_RUN_COMPENSATED) return true;
    double X = 0;
    if (s->has_wall) for (auto &c : s->hctx) for (int k = 0; k < 3; k++) X = std::max(X, c.semi[k]);
    if (s->box_kind == GD_BOX_PERIODIC) for (int k = 0; k < 3; k++) X = std::max(X, s->box[k]);
    if (!(X > 0)) X = 16.0;
    const double ulp = std::ldexp(1.0, std::ilogb(X) - 23), sigma = std::sqrt(2.0 * s->mob_max * run->temperature * run->timestep);
    return sigma < 64.0 * ulp;
}

// List width by tile class.  On S-genome-30k x 128 the steady-state cost is flat to 1 % over skins 0.7 ... 0.85 and 2.5 % lower at
// 0.9 ... 0.95 (interval 20-22 instead of 14: a third fewer builds; `tools/skin_sweep.sh`, profiles/r04_skin_sweep.txt) -- as long as
// the largest tile stays inside the three-block LDS class (3 312 entries); one class up every block loses a third of its occupancy
// (S-genome-62k at 0.9: -15 %).  So: the handle starts at 0.75 and moves to 0.9 once the largest tile of the builds, scaled to the
// wider list (the halo part of a tile grows with the square of the list radius), has fitted the class for three accepted
// chunks in a row; it moves back when the largest tile of a build at 0.9 comes within 24 entries of the class (the next build is
// already at 0.75: no build in the two-block class in between), and waits 64 chunks before it looks again.  The rule reads
// the state only (tile sizes are cell counts), never a clock: the same state selects the same width.  Not with a caller-chosen skin
// (gd_tuning.skin), not while the timing-based selection (gd_tuning.auto_skin) is on.
static void class_skin(gd_system *s, const gd_run_desc *run)
{
    if (s->skin_dense_from > 0 && !s->skin_fixed) {
        // back from the narrow width of a dense state (dense_guard) once the longest list, scaled to the width it left, is short
        // again; the timing-based selection, when enabled, starts from there
        const unsigned need_w = s->last_need_w;
        const double cut = pair_cutoff(s), sc0 = s->rv / cut - s->skin, ratio = (sc0 + s->skin_dense_from) / (sc0 + s->skin);
        // (narrowed for the memory of the rows: back when the longest list, scaled with the cube of the radius, fits the budget again;
        // narrowed for the LDS tile: back -- in steps of at most a quarter of the width, the densest tile decides -- when the largest
        // tile, scaled with the square of the radius, fits 0.85 of the LDS)
        if (s->dense_by_tile) {
            if (s->list_tiled && s->last_need_t > 0 && !(s->skin_next > 0)) {
                const double target = std::min(s->skin_dense_from, s->skin * 1.35 + 0.02);
                const double rt = (sc0 + target) / (sc0 + s->skin);
                if ((double)s->last_need_t * rt * rt <= 0.85 * 8192.0) {      // (a decondensing globule: the tiles shrink from build to build; a miss costs one rolled-back chunk)
                    s->skin_next = target;
                    if (target >= s->skin_dense_from - 1e-9) { s->skin_dense_from = 0; s->dense_by_tile = false; }
                    if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] dense state eases (largest tile %u): skin %.3f at the next build\n", s->last_need_t, s->skin_next);
                }
            }
            return;
        }
        if (need_w > 0 && (double)need_w * ratio * ratio * ratio <= 0.8 * (double)s->dense_budget && !(s->skin_next > 0)) {
            s->skin_next = s->skin_dense_from; s->skin_dense_from = 0;
            { const bool on = s->tuner.enabled; s->tuner = gd_system::SkinTuner{}; s->tuner.enabled = on; }      // (a fresh selection from the default width)
            if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] dense state has passed (longest list %u): skin %.3f at the next build\n", need_w, s->skin_next);
        }
        return;
    }
    if (s->skin_fixed || !s->adapt || s->tuner.enabled || !s->list_tiled || !s->last_need_t || s->sw_n) return;
    const double lo = 0.75, hi = 0.9;
    const double sc = s->pair.scale_by_bead_scale ? bead_scale_bound(s, run, s->K) : 1.0;
    auto move_to = [&](double skin) {      // takes effect at the next build (take_pending_skin): the list in use stays valid until then
        s->skin_next = skin; s->skin_streak = 0;
        if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] list width by tile class: skin %.2f at the next build (largest tile %u)\n", skin, s->last_need_t);
    };
    if (s->skin_next > 0) return;
    if (s->skin < hi - 1e-9) {
        if (s->skin_hold > 0) { s->skin_hold--; return; }
        const double ratio = (sc + hi) / (sc + s->skin);
        // a tile = the block's own slots under three (dz) planes + the halo rows around them: only the halo grows with the cell
        // cross-section (measured on S-genome-30k: 2 930 entries at 0.75, 3 074 at 0.9)
        const double own = 3.0 * GD_BLOCK, est = own + std::max(0.0, (double)s->last_need_t - own) * ratio * ratio + 24.0;
        if (est <= 3312.0 - 24.0 && s->list_tile_cap <= 3312u) { if (++s->skin_streak >= 3) move_to(hi); }
        else s->skin_streak = 0;
    } else if (s->skin <= hi + 1e-9 && (s->list_tile_cap > 3312u || s->last_need_t + 24u > 3312u)) { move_to(lo); s->skin_hold = 64; }
}

extern "C" int gd_apply_callback(gd_system *s)
{
    if (!s) return fail(GD_EINVAL, "gd_apply_callback: NULL system");
    return apply_pending(s);
}

extern "C" int gd_run(gd_system *s, const gd_run_desc *run)
{
    if (!s || !run) return fail(GD_EINVAL, "gd_run: NULL argument");
    if (run->spacestep != 0) return fail(GD_EUNSUPPORTED, "gd_run: spacestep != 0 (adaptive timestep) is not supported");
    if (run->steps < 0 || !(run->timestep > 0) || run->temperature < 0) return fail(GD_EINVAL, "gd_run: bad steps/timestep/temperature");
    if (run->noise_mode < 0 || run->noise_mode > GD_NOISE_HOST)
        return fail(run->noise_mode == 3 ? GD_EUNSUPPORTED : GD_EINVAL, "gd_run: noise_mode %d not available on the device", run->noise_mode);
    if (run->noise_mode == GD_NOISE_HOST && !run->host_noise) return fail(GD_EINVAL, "gd_run: host noise requested without array");
    if ((run->flags & GD_RUN_WALL_DYNAMICS) && !s->has_wall) return fail(GD_ESTATE, "gd_run: wall dynamics requested without a wall");
    if ((run->flags & GD_RUN_UPDATE_SCALES) && !s->has_scaling) return fail(GD_ESTATE, "gd_run: scale updates requested without gd_set_scaling");
    GDCHK(prepare(s));
    GDCHK(apply_pending(s));
    s->state_serial++;
    if (run->timestep != s->last_dt || run->temperature != s->last_kT) { s->a2_ema = 0; s->last_dt = run->timestep; s->last_kT = run->temperature; }   // another regime: measure afresh
    // The wall or the scales start (or stop) moving with this run -- a relaxation is followed by the production phase: the
    // displacement statistics the interval was adapted on are those of the other regime, and the first intervals of the new one used
    // to end in a rolled-back chunk every few runs (bench.py: flags 0 for the relaxation, wall dynamics + scale updates after it).
    // A fifth off the interval until complete intervals of the new regime have been measured.
    if (s->adapt && s->K > 4 && ((run->flags ^ s->last_flags) & (GD_RUN_WALL_DYNAMICS | GD_RUN_UPDATE_SCALES)) != 0 && s->rebuilds > 0)
        s->K -= s->K / 5;
    s->last_flags = run->flags;
    const bool with_list = pair_cutoff(s) > 0;
    const size_t RN = (size_t)s->R * s->N;
    memset(&s->timing, 0, sizeof s->timing);

    // injected noise lives on the device as float (R,N,3) per step
    const bool host_noise = run->noise_mode == GD_NOISE_HOST && run->temperature > 0;
    if (host_noise) {
        const size_t n = (size_t)run->steps * RN * 3;
        std::vector<float> h(n);
        for (size_t i = 0; i < n; i++) h[i] = (float)run->host_noise[i];
        HIPCHK(s->noise.resize(n, false));
        HIPCHK(hipMemcpy(s->noise.p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }

    const bool comp = run->steps > 0 && want_compensated(s, run);
    if (comp) {
        if (!s->lo_valid) { HIPCHK(hipMemsetAsync(s->lo.p, 0, RN * sizeof(float4), s->stream)); s->lo_valid = true; }
        HIPCHK(s->snap_lo.resize(RN, false));
    }
    if (run->steps > 0) s->comp_last = comp;
    if (run->steps > 0 && !comp) s->lo_valid = false;      // the positions are about to move without their residuals (set here, not on the way
                                                          // out: an early error return must not leave residuals that describe other positions)

    if (run->replica_seeds) {
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "seed width");
        HIPCHK(s->seeds_d.resize(s->R, false));
        HIPCHK(hipMemcpy(s->seeds_d.p, run->replica_seeds, s->R * sizeof(uint64_t), hipMemcpyHostToDevice));
    }

    int64_t done = 0;
    int chunk_retries = 0;     // consecutive rollbacks of the chunk in progress
    float last_dmax2 = 0;      // largest bound, over the replicas, of the squared displacement since the build of the positions the last accepted chunk WROTE
    while (done < run->steps) {
        // ---- one verified chunk
        // (while the skin sweep is measuring candidates the chunks are shorter -- four rebuild intervals -- so that a sweep costs
        // a few thousand steps, not tens of thousands)
        const bool sweeping = s->tuner.enabled && !s->tuner.done && !s->tuner.cand.empty();
        const int64_t chunk = std::min<int64_t>(run->steps - done, sweeping ? std::min<int64_t>(128, std::max<int64_t>(32, 4ll * s->K))
                                                                            : std::min<int64_t>(256, std::max<int64_t>(32, 12ll * s->K)));
        // snapshot for rollback: positions in bead order + context
        gd_launch_gather_positions(s->pos[s->pcur].p, s->slot_of.p, s->snap.p, s->N, s->Np, s->R, 0, s->stream);
        if (comp) HIPCHK(hipMemcpyAsync(s->snap_lo.p, s->lo.p, RN * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
        const std::vector<DevCtx> snap_ctx = s->hctx;
        const bool snap_w_packed = s->w_packed;      // (the snapshot's w is what the positions carried at this point)
        GDCHK(clear_flags(s));

        StepParams p;
        // The scales a callback sets are pure functions of the step index (simulation_driver_interphase.cc:42-43): when every
        // replica is at the same step (the usual case) the host evaluates them and passes them with the launch
        bool common_step = s->has_scaling && (run->flags & GD_RUN_UPDATE_SCALES);
        const long long step0 = s->hctx[0].step;
        for (uint32_t r = 1; r < s->R && common_step; r++) common_step = s->hctx[r].step == step0;
        auto host_scales = [&](StepParams &q, int64_t launch) {      // launch: index within the chunk of the launch that applies the callback
            if (!common_step) return;
            const double time = (double)(step0 + launch) * run->timestep;
            q.scaling.from_host = 1;
            q.scaling.bead_next = scale_at(s, s->bs_init, s->bs_tau, time);
            q.scaling.bond_next = scale_at(s, s->bo_init, s->bo_tau, time);
        };
        size_t nev = 0;
        float step_ms = 0, build_ms = 0;
        // Spans of step launches and of builds share their boundary events: the end of one is the start of the next (an event in the
        // stream costs the device ~5 us between two kernels -- kernel trace of one 30 000-bead replica: 0.3 us between two steps,
        // 10-12 us across the two events that used to separate a build from the steps on either side)
        std::vector<std::pair<size_t, int>> spans;   // index of the span's end event (its start: the event before it), kind (0 step, 1 build)
        hipEvent_t ev_begin = get_event(s, nev++);
        HIPCHK(hipEventRecord(ev_begin, s->stream));
        int64_t k = 0;
        bool full_interval = false;      // the chunk contains the last step of a complete K-step interval
        // (an interval that runs on a contact-search list has a wider skin than the force lists: its displacement is no measure
        // for the interval of those)
        bool on_search_list = s->list_valid && s->search_list;
        while (k < chunk) {
            if (!s->list_valid || s->steps_since_build >= s->K) {
                take_pending_skin(s);
                hipEvent_t e1 = get_event(s, nev++);
                GDCHK(enqueue_build(s, list_radius(s, run, (uint32_t)(k + s->K)), with_list));
                s->search_list = false;
                HIPCHK(hipEventRecord(e1, s->stream));
                spans.push_back({nev - 1, 1});
                s->list_valid = true;
            }
            const int64_t n = std::min<int64_t>((int64_t)s->K - s->steps_since_build, chunk - k);
            hipEvent_t e1 = get_event(s, nev++);
            for (int64_t q = 0; q < n; q++) {
                fill_common(s, p);
                p.dt_d = run->timestep; p.dt = (float)run->timestep; p.kT = (float)run->temperature; p.seed = run->seed;
                p.seeds = run->replica_seeds ? s->seeds_d.p : nullptr;
                p.noise_mode = run->noise_mode; p.run_flags = run->flags; p.comp = comp ? 1 : 0;
                p.host_noise = host_noise ? s->noise.p + (size_t)(done + k + q) * RN * 3 : nullptr;
                host_scales(p, k + q);
                // the interval adaptation needs the displacement at K steps since the build: recorded at the last force
                // evaluation of a COMPLETE interval only (a chunk that ends mid-interval records nothing and adapts nothing)
                p.record_disp = (s->steps_since_build + (uint32_t)q + 1u == s->K);
                full_interval |= p.record_disp != 0;
                gd_launch_step(p, GD_MODE_STEP, s->stream);
                if (s->sw_n) launch_softwell(s, p, 0);
                s->pcur ^= 1; s->ccur ^= 1;
            }
            HIPCHK(hipEventRecord(e1, s->stream));
            spans.push_back({nev - 1, 0});
            s->timing.step_launches += (uint64_t)n;
            s->steps_since_build += (uint32_t)n;
            k += n;
        }
        // apply the callback of the last step (unless the caller wants to observe the state its callback sees first), then
        // check the chunk
        const bool defer = (run->flags & GD_RUN_DEFER_CALLBACK) && done + chunk == run->steps;
        if (!defer) {
            fill_common(s, p);
            p.dt_d = run->timestep; p.dt = (float)run->timestep; p.run_flags = run->flags;
            host_scales(p, chunk);
            gd_launch_finalize(p, 0, s->stream);
            s->ccur ^= 1;
        } else { s->pend_dt = run->timestep; s->pend_flags = run->flags; }
        hipEvent_t ev_end = get_event(s, nev++);
        HIPCHK(hipEventRecord(ev_end, s->stream));
        // one round trip for everything the host wants from the chunk: flags, contexts, list counts
        std::vector<unsigned> f((size_t)s->R * GD_NFLAGS);
        std::vector<DevCtx> ctx_new(s->R);
        HIPCHK(hipGetLastError());
        {
            const size_t nf = f.size() * sizeof(unsigned), nc = s->R * sizeof(DevCtx), nl = 2 * (size_t)s->R * sizeof(unsigned long long), nd = s->R * sizeof(float);
            if (!s->h_chunk) HIPCHK(hipHostMalloc((void **)&s->h_chunk, nf + nc + nl + nd + 16, hipHostMallocDefault));
            HIPCHK(hipMemcpyAsync(s->h_chunk + nf + nc + nl + nd, s->pool.p, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));      // the row pool's use
            HIPCHK(hipMemcpyAsync(s->h_chunk, s->flags.p, nf, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipMemcpyAsync(s->h_chunk + nf, s->ctx[s->ccur].p, nc, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipMemcpyAsync(s->h_chunk + nf + nc, s->lcount_d.p, nl, hipMemcpyDeviceToHost, s->stream));
            // the running displacement bound of every replica (tiled path; one word per 128-byte line): it covers the positions the
            // last step WROTE, which no step has checked yet
            HIPCHK(hipMemcpy2DAsync(s->h_chunk + nf + nc + nl, sizeof(float), s->dmax.p, GD_DMAX_STRIDE * sizeof(unsigned), sizeof(float), s->R,
                                    hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            memcpy(f.data(), s->h_chunk, nf); memcpy(ctx_new.data(), s->h_chunk + nf, nc); memcpy(s->lcount.data(), s->h_chunk + nf + nc, nl);
            last_dmax2 = 0;
            for (uint32_t r = 0; r < s->R; r++) { float d2; memcpy(&d2, s->h_chunk + nf + nc + nl + r * sizeof(float), 4); last_dmax2 = std::max(last_dmax2, d2); }
            { unsigned used[4]; memcpy(used, s->h_chunk + nf + nc + nl + nd, 16); if (s->list_tiled && used[0] > 0) { s->pool_used = std::max(used[0], used[1]); s->repairs = used[2]; } }
        }
        bool violated = false; float maxd2 = 0;
        for (uint32_t r = 0; r < s->R; r++) {
            violated |= f[r * GD_NFLAGS + GD_FLAG_VIOLATION] != 0;
            float d2; memcpy(&d2, &f[r * GD_NFLAGS + GD_FLAG_MAXDISP2], 4); maxd2 = std::max(maxd2, d2);
        }
        const bool over = handle_overflow(s, f);
        if (violated || over) {
            // roll the chunk back: restore bead-order positions + context, shorten the interval / widen the list
            s->rollbacks++;
            // (every cause of a rollback changes what the retry runs with -- interval, skin, pool, tile class, list path -- so a chunk
            // converges in a few attempts; a chunk that does not is a defect, reported instead of retried for ever)
            if (++chunk_retries > 24) return fail(GD_ESTATE, "gd_run: a chunk of %lld steps at step %lld was rolled back %d times (%s): giving up",
                                                  (long long)chunk, (long long)s->hctx[0].step, chunk_retries, over ? "list / tile / pool overflow" : "skin violation");
            if (dev_env("GDYN_DEBUG")) fprintf(stderr, "[gdyn] rollback %llu: %s, K %u, skin %.3f, chunk of %lld steps at step %lld\n", (unsigned long long)s->rollbacks,
                                               over ? "overflow" : "skin violation", s->K, s->skin, (long long)chunk, (long long)s->hctx[0].step);
            HIPCHK(hipMemcpy2DAsync(s->pos[s->pcur].p, (size_t)s->Np * sizeof(float4), s->snap.p, (size_t)s->N * sizeof(float4),
                                    (size_t)s->N * sizeof(float4), s->R, hipMemcpyDeviceToDevice, s->stream));
            gd_launch_identity(s->orig[s->ocur].p, s->slot_of.p, s->N, s->Np, s->R, s->stream);
            if (comp) HIPCHK(hipMemcpyAsync(s->lo.p, s->snap_lo.p, RN * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
            s->hctx = snap_ctx; s->ctx_dirty = true; s->w_packed = snap_w_packed;
            GDCHK(upload_ctx(s));
            s->list_valid = false;
            s->bbox_valid = false;      // (the box the abandoned builds recorded may be that of positions stepped on incomplete lists)
            if (violated && !over) {
                if (s->K == 1) {
                    if (s->skin > 8) return fail(GD_ESTATE, "gd_run: Verlet skin cannot cover one step (timestep too large?)");
                    s->skin *= 1.5;
                } else { s->K_bad = s->K; s->K_bad_ttl = 64; s->K = std::max(1u, s->K - std::max(1u, s->K / 4)); s->a2_ema = 0; }   // (a gentler cut, K/8 for 32 chunks, violates again sooner: measured 1% slower)
            }
            s->timing.step_launches -= std::min<uint64_t>(s->timing.step_launches, (uint64_t)chunk);
            tune_skin(s, 0.0, 0, false, true);
            continue;
        }
        // accepted: timing, context mirror, cadence adaptation
        if (s->list_tiled) { s->tiled_backoff = 8; s->tiled_wait = 0; }
        else if (!s->tiled_ok && s->tiled_off == 1 && ++s->tiled_wait >= s->tiled_backoff) {
            // the tiled path was left because one tile did not fit (a dense transient, e.g. the start of a relaxation):
            // try it again at the next build, with the largest tile class; a new overflow costs one rolled-back chunk
            // and doubles the waiting time
            s->tiled_ok = true; s->tiled_off = 0; s->tiled_wait = 0; s->tiled_backoff = std::min(2 * s->tiled_backoff, 1024u);
            s->tile_cap = 8192u;
        }
        float ms = 0;
        for (auto &sp : spans) {
            HIPCHK(hipEventElapsedTime(&ms, s->events[sp.first - 1], s->events[sp.first]));
            (sp.second ? build_ms : step_ms) += ms;
        }
        HIPCHK(hipEventElapsedTime(&ms, ev_begin, ev_end));
        s->timing.total_ms += ms; s->timing.step_kernel_ms += step_ms; s->timing.rebuild_ms += build_ms;
        s->hctx = ctx_new;
        unsigned long long L = 0;
        for (uint32_t r = 0; r < s->R; r++) L += s->lcount[r];
        s->timing.list_entries_visited += L * (uint64_t)chunk;   // L of the last build, per step
        if (s->adapt && with_list && full_interval && !on_search_list) {
            const double cut_now = pair_cutoff(s) * (s->pair.scale_by_bead_scale ? bead_scale_bound(s, nullptr, 0) : 1.0);
            const double lim = 0.5 * (s->rv - cut_now), d = std::sqrt((double)maxd2);
            if (lim > 0 && d > 0) {
                // displacement grows ~ sqrt(steps): aim at 90% of the skin at the end of an interval. The maximum over ~2e6 beads
                // fluctuates by only ~3.5% between intervals (sigma / sqrt(2 ln n)) and the measurement is the largest of a chunk's
                // (up to 12) intervals, averaged over chunks, i.e. already biased upwards: measured on the benchmark state no rollback
                // in 40 000 steps at 0.90, the first ones at 0.92; a violation costs one rolled-back chunk and is remembered (K_bad)
                // (the squared displacement per step of an interval, d^2 / K, is averaged over the chunks -- weight 0.4 for the newest --
                // so that the interval does not jitter with the single measurement: 12 ... 15 on the benchmark state otherwise)
                const double a2 = d * d / (double)s->K;
                s->a2_ema = s->a2_ema > 0 ? 0.6 * s->a2_ema + 0.4 * a2 : a2;
                static const double target = dev_env("GDYN_K_TARGET") ? atof(dev_env("GDYN_K_TARGET")) : 0.90;
                double knew = target * lim * target * lim / s->a2_ema;
                knew = std::min(knew, 2.0 * s->K + 1);
                s->K = (uint32_t)std::max(1.0, std::min(200.0, std::floor(knew)));
            } else if (d == 0) s->K = std::min(200u, s->K * 2);
            if (s->K_bad_ttl > 0) { s->K_bad_ttl--; if (s->K >= s->K_bad) s->K = std::max(1u, s->K_bad - 1); }
        }
        if (with_list) tune_skin(s, ms, chunk, full_interval, false);
        if (with_list && full_interval) class_skin(s, run);
        done += chunk; chunk_retries = 0;
        if (s->repair_wide > 0) s->repair_wide--;
    }
    // The last chunk was accepted: every bead is within the margin the list in use was built for, at the cutoff of the last step.
    // That is still the cutoff an observation sees when the scales did not move behind that step (callback deferred, or no scale
    // updates in this run) -- the resident list then serves gd_compute_energy as it is.
    // (Not with the droplet term: its kernel moves beads behind k_step's check.)
    // The checks of a step cover the positions it READ; the positions the last step wrote are covered by the running bound of the
    // tiled path (dmax: triangle bound of the written positions, read back with the chunk): the list serves an observation only if that
    // bound is inside the margin too.  The generic path keeps no such bound: its observations build a list.
    if (run->steps > 0 && with_list && s->list_valid && !s->sw_n && (!(run->flags & GD_RUN_UPDATE_SCALES) || (run->flags & GD_RUN_DEFER_CALLBACK))) {
        const double cut_obs = pair_cutoff(s) * (s->pair.scale_by_bead_scale ? bead_scale_bound(s, nullptr, 0) : 1.0);
        const double lim = 0.5 * ((double)s->rv - cut_obs);
        if (s->list_tiled && lim > 0 && (double)last_dmax2 <= lim * lim) s->verified_serial = s->state_serial;
    }
    return GD_OK;
}

// ------------------------------------------------------------- observation

extern "C" int gd_compute_energy(gd_system *s, uint32_t mask, double *energy)
{
    if (!s || !energy) return fail(GD_EINVAL, "gd_compute_energy: NULL argument");
    GDCHK(prepare(s));
    GDCHK(ensure_fresh_list(s));
    StepParams p;
    fill_common(s, p);
    p.term_mask = mask;
    gd_launch_step(p, GD_MODE_ENERGY, s->stream);
    const bool droplet = s->sw_n && (mask & GD_TERM_PAIR);
    std::vector<double> esw(s->R, 0.0);
    if (droplet) {
        HIPCHK(hipMemsetAsync(s->sw_esum.p, 0, s->R * sizeof(double), s->stream));
        launch_softwell(s, p, 2);
        HIPCHK(hipMemcpyAsync(esw.data(), s->sw_esum.p, s->R * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    }
    std::vector<double> part((size_t)s->R * s->nblk);
    HIPCHK(hipMemcpyAsync(part.data(), s->epart.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (uint32_t r = 0; r < s->R; r++) {
        double e = esw[r];
        for (uint32_t b = 0; b < s->nblk; b++) e += part[(size_t)r * s->nblk + b];
        energy[r] = e;
    }
    return GD_OK;
}

extern "C" int gd_compute_forces(gd_system *s, uint32_t mask, double *forces)
{
    if (!s || !forces) return fail(GD_EINVAL, "gd_compute_forces: NULL argument");
    GDCHK(prepare(s));
    GDCHK(apply_pending(s));      // a force evaluation replaces the wall-reaction partials: the pending callback consumes its own first
    GDCHK(ensure_fresh_list(s));
    StepParams p;
    fill_common(s, p);
    p.term_mask = mask;
    HIPCHK(hipMemsetAsync(p.react_out, 0, s->react_part[0].n * sizeof(float4), s->stream));
    gd_launch_step(p, GD_MODE_FORCE, s->stream);
    if (s->sw_n && (mask & GD_TERM_PAIR)) launch_softwell(s, p, 1);
    if (s->has_wall) { p.react_in = p.react_out; gd_launch_finalize(p, 1, s->stream); s->ccur ^= 1; }
    const size_t RN = (size_t)s->R * s->N;
    std::vector<float4> h(RN);
    HIPCHK(hipMemcpyAsync(h.data(), s->fout.p, RN * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (size_t i = 0; i < RN; i++) { forces[3 * i] = h[i].x; forces[3 * i + 1] = h[i].y; forces[3 * i + 2] = h[i].z; }
    if (s->has_wall) GDCHK(download_ctx(s));
    return GD_OK;
}

// Pairs within dcut of replicas r0 .. r0 + nrep - 1, one launch: replica r0 + y's pairs at out + y * (out.n / nrep), their number in
// cnt[2 y] (and left on the device in `count`).  The buffer grows until every replica fits.
static int search_device(gd_system *s, uint32_t r0, uint32_t nrep, double dcut, DevBuf<uint2> &out, DevBuf<unsigned long long> &count,
                         std::vector<unsigned long long> &cnt)
{
    const bool with_list = pair_cutoff(s) > 0;
    if (!out.p || out.n % nrep) HIPCHK(out.resize((size_t)nrep * std::max<size_t>((size_t)s->N * 8, 4096), false));
    HIPCHK(count.resize(2 * (size_t)nrep));
    cnt.assign(2 * (size_t)nrep, 0ull);
    for (int attempt = 0; attempt < 6; attempt++) {
        if (!s->list_valid || !((float)dcut <= s->rv) || s->list_W == 0) {
            // the list stays in use as the force list of the next run: built with that run's look-ahead (a growing bead
            // scale over the rest of an interval), like the builds inside gd_run
            gd_run_desc ahead{};
            ahead.timestep = s->last_dt; ahead.flags = s->last_flags;
            take_pending_skin(s);
            const float rv_force = with_list ? list_radius(s, s->last_dt > 0 ? &ahead : nullptr, s->K) : 0.f;
            const float rv_search = (float)(dcut * (1.0 + 1e-6));
            GDCHK(build_now(s, std::max(rv_force, rv_search), true, true, rv_search));
            s->list_valid = true; s->search_list = rv_search > rv_force;
        }
        const double lim = 0.5 * ((double)s->rv - dcut);
        PairsP q;
        memset(&q, 0, sizeof q);
        q.pos = s->pos[s->pcur].p; q.x0 = s->list_tiled ? s->rec_x0.p : s->xb.p; q.rec_mo = s->rec_mo.p; q.meta = s->meta.p;
        q.orig = s->orig[s->ocur].p; q.nbr = s->nbr.p; q.nbr16 = s->nbr16.p; q.tiles = s->tiles.p; q.wtab = s->wtab.p;
        q.N = s->N; q.Np = s->Np; q.nblk = s->nblk; q.r = r0; q.nrep = nrep; q.W = s->list_W;
        q.tiled = s->list_tiled ? 1 : 0; q.s16 = (s->list_tiled && s->list_tile_cap < 4096u) ? 1 : 0;
        q.periodic = s->box_kind == GD_BOX_PERIODIC;
        for (int k = 0; k < 3; k++) { q.box[k] = (float)s->box[k]; q.inv_box[k] = s->box[k] > 0 ? (float)(1.0 / s->box[k]) : 0.f; }
        q.dcut2 = (float)(dcut * dcut); q.lim2 = (float)(lim * lim);
        q.out = out.p; q.cap = out.n / nrep; q.count = count.p;
        q.dmax = s->dmax.p;
        HIPCHK(hipMemsetAsync(count.p, 0, 2 * (size_t)nrep * sizeof(unsigned long long), s->stream));
        gd_launch_pairs(q, s->stream);
        HIPCHK(hipMemcpyAsync(cnt.data(), count.p, 2 * (size_t)nrep * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(hipGetLastError());
        unsigned long long moved = 0, most = 0;
        for (uint32_t y = 0; y < nrep; y++) { most = std::max(most, cnt[2 * y]); moved |= cnt[2 * y + 1]; }
        if (moved) { s->list_valid = false; continue; }                  // a bead moved beyond the margin: fresh list
        if (most > q.cap) { HIPCHK(out.resize((size_t)nrep * (size_t)(most + most / 8 + 64), false)); continue; }
        return GD_OK;
    }
    return fail(GD_ESTATE, "pair search: did not converge");
}

// md::neighbor_searcher{box, dcut}.search(): served on the device from the resident Verlet list when that list is
// complete for dcut (dcut + 2 x largest displacement since the build <= list radius), after ONE list build otherwise -- a
// build at radius max(force-list radius, dcut), so the force list stays valid either way and the next gd_run does not rebuild.
// The result is cached for the repeated call of the count-then-fetch idiom.
extern "C" int gd_search_pairs(gd_system *s, uint32_t r, double dcut, uint32_t *pairs, uint64_t cap, uint64_t *n_pairs)
{
    if (!s || !n_pairs || (cap && !pairs)) return fail(GD_EINVAL, "gd_search_pairs: NULL argument");
    if (r >= s->R || !(dcut > 0)) return fail(GD_EINVAL, "gd_search_pairs: bad replica or cutoff");
    GDCHK(prepare(s));
    if (!(s->sp_valid && s->sp_r == r && s->sp_dcut == dcut && s->sp_serial == s->state_serial)) {
        std::vector<unsigned long long> cnt;
        GDCHK(search_device(s, r, 1, dcut, s->sp_out, s->sp_count, cnt));
        s->sp_host.resize((size_t)cnt[0]);
        if (cnt[0]) HIPCHK(hipMemcpy(s->sp_host.data(), s->sp_out.p, (size_t)cnt[0] * sizeof(uint2), hipMemcpyDeviceToHost));
        s->sp_valid = true; s->sp_r = r; s->sp_dcut = dcut; s->sp_serial = s->state_serial;
    }
    const uint64_t n = s->sp_host.size();
    for (uint64_t k = 0; k < n && k < cap; k++) { pairs[2 * k] = s->sp_host[k].x; pairs[2 * k + 1] = s->sp_host[k].y; }
    *n_pairs = n;
    return GD_OK;
}

// ------------------------------------------------------------- contact maps
// contact_map (simulation_interphase/contact_map.cc:26-91) on the device, for all replicas of the handle at once: update() = one
// pair search over every replica + one insert launch into the per-replica count tables; nothing but R pair counts crosses PCIe
// until a map is dumped.  The tables share one capacity (a power of two) and are grown AHEAD of an update so that no table is more
// than half full after it, whatever the update adds (the search has already counted the pairs when the tables are sized).
static unsigned contact_jbits(const gd_system *s)      // bits of a bead id
{
    unsigned b = 1;
    while (((uint64_t)(s->N - 1) >> b) != 0) b++;
    return b;
}

static ContactTab contact_tab(gd_system *s)
{
    ContactTab t;
    t.words = s->ct_words.p; t.distinct = s->ct_distinct.p; t.cap = s->ct_cap; t.jbits = contact_jbits(s);
    return t;
}

extern "C" int gd_contacts_update(gd_system *s, double distance)
{
    if (!s) return fail(GD_EINVAL, "gd_contacts_update: NULL argument");
    if (!(distance > 0)) return fail(GD_EINVAL, "gd_contacts_update: the contact distance must be positive");
    if (contact_jbits(s) > 20) return fail(GD_EINVAL, "gd_contacts_update: contact maps take at most 1 048 576 beads (the count shares a 64-bit word with the pair)");
    GDCHK(prepare(s));
    std::vector<unsigned long long> cnt;
    GDCHK(search_device(s, 0, s->R, distance, s->ct_pairs, s->ct_count, cnt));
    if (s->ct_distinct_h.size() != s->R) s->ct_distinct_h.assign(s->R, 0u);
    unsigned long long need = 0, most = 0;
    for (uint32_t r = 0; r < s->R; r++) { need = std::max(need, s->ct_distinct_h[r] + cnt[2 * r]); most = std::max(most, cnt[2 * r]); }
    size_t cap = std::max<size_t>(s->ct_cap, 1024);
    while (cap < 2 * need) cap *= 2;
    if (cap != s->ct_cap) {
        DevBuf<unsigned long long> words;
        HIPCHK(words.resize((size_t)s->R * cap, false));
        HIPCHK(hipMemsetAsync(words.p, 0xff, words.n * sizeof(unsigned long long), s->stream));
        if (s->ct_cap) {
            ContactTab to = contact_tab(s), from = to;
            to.words = words.p; to.cap = cap;
            HIPCHK(hipMemsetAsync(s->ct_distinct.p, 0, s->R * sizeof(unsigned), s->stream));      // (recounted by the rehash)
            gd_launch_contacts_rehash(from, to, s->R, s->stream);
            HIPCHK(hipStreamSynchronize(s->stream));      // the old tables are freed below
        } else HIPCHK(s->ct_distinct.resize(s->R));
        std::swap(s->ct_words.p, words.p); std::swap(s->ct_words.n, words.n);
        s->ct_cap = cap;
    }
    gd_launch_contacts_insert(contact_tab(s), s->ct_pairs.p, s->ct_pairs.n / s->R, s->ct_count.p, most, s->R, s->stream);
    HIPCHK(hipMemcpyAsync(s->ct_distinct_h.data(), s->ct_distinct.p, s->R * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    return GD_OK;
}

extern "C" int gd_contacts_fetch(gd_system *s, uint32_t r, uint32_t *rows, uint64_t cap, uint64_t *n_rows)
{
    if (!s || !n_rows || (cap && !rows)) return fail(GD_EINVAL, "gd_contacts_fetch: NULL argument");
    if (r >= s->R) return fail(GD_EINVAL, "gd_contacts_fetch: bad replica");
    const uint64_t n = s->ct_cap ? s->ct_distinct_h[r] : 0;
    *n_rows = n;
    if (!n || !cap) return GD_OK;       // (the count of the count-then-fetch idiom costs nothing: the host mirrors the occupancy)
    HIPCHK(hipSetDevice(s->device));
    for (int k = 0; k < 2; k++)
        if (s->ct_ck[k].n < n) { HIPCHK(s->ct_ck[k].resize((size_t)(n + n / 4), false)); HIPCHK(s->ct_cv[k].resize((size_t)(n + n / 4), false)); }
    HIPCHK(s->ct_n.resize(1));
    HIPCHK(hipMemsetAsync(s->ct_n.p, 0, sizeof(unsigned), s->stream));
    gd_launch_contacts_compact(contact_tab(s), r, s->ct_ck[0].p, s->ct_cv[0].p, s->ct_n.p, s->stream);
    const unsigned jb = contact_jbits(s), bits = 2 * jb;      // key = i << jb | j
    size_t tmp_bytes = 0;
    HIPCHK(gd_sort_contacts(nullptr, &tmp_bytes, s->ct_ck[0].p, s->ct_ck[1].p, s->ct_cv[0].p, s->ct_cv[1].p, (size_t)n, bits, s->stream));
    if (s->ct_tmp.n < tmp_bytes) HIPCHK(s->ct_tmp.resize(tmp_bytes + tmp_bytes / 4, false));
    tmp_bytes = s->ct_tmp.n;
    HIPCHK(gd_sort_contacts(s->ct_tmp.p, &tmp_bytes, s->ct_ck[0].p, s->ct_ck[1].p, s->ct_cv[0].p, s->ct_cv[1].p, (size_t)n, bits, s->stream));
    std::vector<unsigned long long> hk((size_t)n); std::vector<unsigned> hv((size_t)n);
    unsigned found = 0;
    HIPCHK(hipMemcpyAsync(hk.data(), s->ct_ck[1].p, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(hv.data(), s->ct_cv[1].p, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(&found, s->ct_n.p, sizeof found, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    if (found != n) return fail(GD_ESTATE, "gd_contacts_fetch: table occupancy %u differs from the %llu entries counted", found, (unsigned long long)n);
    for (uint64_t k = 0; k < n && k < cap; k++) {
        rows[3 * k] = (uint32_t)(hk[k] >> jb); rows[3 * k + 1] = (uint32_t)(hk[k] & ((1ull << jb) - 1ull)); rows[3 * k + 2] = hv[k];
    }
    return GD_OK;
}

extern "C" int gd_contacts_clear(gd_system *s, uint32_t r)
{
    if (!s) return fail(GD_EINVAL, "gd_contacts_clear: NULL argument");
    if (r != GD_ALL_REPLICAS && r >= s->R) return fail(GD_EINVAL, "gd_contacts_clear: bad replica");
    if (!s->ct_cap) return GD_OK;
    HIPCHK(hipSetDevice(s->device));
    const uint32_t r0 = r == GD_ALL_REPLICAS ? 0 : r, nr = r == GD_ALL_REPLICAS ? s->R : 1;
    HIPCHK(hipMemsetAsync(s->ct_words.p + (size_t)r0 * s->ct_cap, 0xff, (size_t)nr * s->ct_cap * sizeof(unsigned long long), s->stream));
    HIPCHK(hipMemsetAsync(s->ct_distinct.p + r0, 0, nr * sizeof(unsigned), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (uint32_t k = r0; k < r0 + nr; k++) s->ct_distinct_h[k] = 0;
    return GD_OK;
}

// ------------------------------------------------------------ micro-benchmark (developer builds only)
#ifdef GD_DEV
extern "C" int gd_debug_bench(gd_system *s, int what, int n, double *mean_ms)
{
    if (!s || !mean_ms || n < 1) return fail(GD_EINVAL, "gd_debug_bench: bad argument");
    GDCHK(prepare(s));
    GDCHK(ensure_fresh_list(s));
    hipEvent_t e0 = get_event(s, 0), e1 = get_event(s, 1);
    const float rv = list_radius(s, nullptr, 0);
    StepParams p;
    fill_common(s, p);
    p.dt_d = 1e-5; p.dt = 1e-5f; p.kT = 1.0f; p.seed = 1; p.noise_mode = GD_NOISE_PHILOX; p.run_flags = 0;
    p.ctx_out = s->ctx[s->ccur ^ 1].p;     // scratch: the current context is not replaced
    GDCHK(clear_flags(s));
    if (what >= 10) HIPCHK(hipMemsetAsync(s->fout.p, 0, s->fout.n * sizeof(float4), s->stream));
    HIPCHK(hipEventRecord(e0, s->stream));
    for (int i = 0; i < n; i++) {
        if (what == 0 || what >= 30) { GDCHK(enqueue_build(s, rv, pair_cutoff(s) > 0)); }
        else gd_launch_step(p, GD_MODE_STEP, s->stream);
    }
    HIPCHK(hipEventRecord(e1, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *mean_ms = ms / n;
    if (what >= 10) {
        // section stamps of a timing-only kernel build (-DGD_ABL=30): mean shader-clock cycles per wave spent in
        // section what-10 (zeros with the product kernels, which never write the force buffer in step mode)
        // (one 8-word record per wave: 7 section times of the last launch + a presence flag)
        std::vector<unsigned long long> rec(std::min<size_t>(s->fout.n * 2, (size_t)1 << 22));
        HIPCHK(hipMemcpy(rec.data(), s->fout.p, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum = 0, waves = 0;
        const int idx = what >= 30 ? what - 30 : what - 10;
        for (size_t w = 0; w + 16 <= rec.size(); w += 16)
            if (rec[w + 15] == 1ull && idx >= 0 && idx < 12) { sum += (double)rec[w + idx]; waves += 1; }
        *mean_ms = waves > 0 ? sum / waves : 0.0;
    }
    GDCHK(clear_flags(s));
    return GD_OK;
}
#endif
