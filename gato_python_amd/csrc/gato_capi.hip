// C ABI of libgato_hip.so (include/gato_hip.h): solver object, stage-level entry points on device
// pointers, the device-resident whole solve and the host-pointer drop-in for main_call
// (gpu_library.cu:85-234).
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "gato_common.h"

namespace gato {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- type-erased op table -------------------------------------------------------------------
template <typename T, int S, int C>
static Ops make_ops(int dtype)
{
    Ops o;
    o.S = S; o.C = C; o.dtype = dtype;
    o.convert = [](const Dims &d, const int *gr, const int *gc, const void *gv, const int *cr, const int *cc,
                   const void *cv, double rho, void *Gd, void *Cd, void *Gi, hipStream_t st) {
        return launch_convert<T, S, C>(d, gr, gc, (const T *)gv, cr, cc, (const T *)cv, (T)rho, (T *)Gd, (T *)Cd, (T *)Gi, st);
    };
    o.add_rho = [](const Dims &d, const void *Gin, double rho, void *Gd, hipStream_t st) {
        return launch_add_rho<T, S, C>(d, (const T *)Gin, (T)rho, (T *)Gd, st);
    };
    o.form_schur = [](const Dims &d, const void *Gd, const void *Cd, const void *g, const void *c, void *Sb,
                      void *Pb, void *gam, void *Gi, bool have_inv, hipStream_t st) {
        return launch_form_schur<T, S, C>(d, (const T *)Gd, (const T *)Cd, (const T *)g, (const T *)c, (T *)Sb,
                                          (T *)Pb, (T *)gam, (T *)Gi, have_inv, st);
    };
    o.assemble = [](const Dims &d, const AsmArgs &a, hipStream_t st) { return launch_assemble<T, S, C>(d, a, st); };
    o.form_ss = [](const Dims &d, const void *Sb, void *Pb, hipStream_t st) {
        return launch_form_ss<T, S, C>(d, (const T *)Sb, (T *)Pb, st);
    };
    o.point_jacobi = [](const Dims &d, const void *Sb, void *Pb, hipStream_t st) {
        return launch_point_jacobi<T, S, C>(d, (const T *)Sb, (T *)Pb, st);
    };
    o.compute_dz = [](const Dims &d, const void *Gi, const void *Cd, const void *g, const void *lam, void *dz,
                      hipStream_t st) {
        return launch_compute_dz<T, S, C>(d, (const T *)Gi, (const T *)Cd, (const T *)g, (const T *)lam, (T *)dz, st);
    };
    o.pcg_plan = [](PcgPlan *p) { return pcg_resident_plan<T, S>(p); };
    o.pcg_resident = [](const PcgLaunch &a, hipStream_t st) { return launch_pcg_resident<T, S>(a, st); };   // incl. the DPP-row layout
    o.pcg_dma_max_knots = []() { return pcg_dma_max_knots<T, S>(); };
    o.pcg_dma = [](const PcgLaunch &a, hipStream_t st) { return launch_pcg_dma<T, S>(a, st); };
    o.pcg_cg1_max_threads = []() { return pcg_cg1_max_threads<T, S>(); };
    o.pcg_cg1 = [](const PcgLaunch &a, hipStream_t st) { return launch_pcg_cg1<T, S>(a, st); };
    o.stream_grid = [](int K, int mg) { return stream_grid<T, S>(K, mg); };
    o.stream_step = [](int ph, const StreamStep &a, int grid, hipStream_t st) { return launch_stream_step<T, S>(ph, a, grid, st); };
    o.stream_pack = [](const void *sl, int n, const void *y, int K, void *send, hipStream_t st) {
        return launch_stream_pack<T, S>(sl, n, y, K, send, st);
    };
    o.stream_finish = [](const void *part, int n, int stride, double tol, int last_it, int *done, int *iters,
                         double *fe, double *hist, hipStream_t st) {
        return launch_stream_finish<T, S>(part, n, stride, tol, last_it, done, iters, fe, hist, st);
    };
    o.pcg_streaming = [](const Dims &d, const void *Sb, const void *Pb, const void *gam, void *lam, double tol,
                         int max_iters, int *iters, const PcgStreamWork &w, hipStream_t st) {
        return launch_pcg_streaming<T, S>(d, (const T *)Sb, (const T *)Pb, (const T *)gam, (T *)lam, (T)tol,
                                          max_iters, iters, w, st);
    };
    return o;
}

static const std::vector<Ops> &all_ops()
{
    static const std::vector<Ops> v = [] {
        std::vector<Ops> t;
#define X(S_, C_)                                     \
    t.push_back(make_ops<float, S_, C_>(GATO_F32));   \
    t.push_back(make_ops<double, S_, C_>(GATO_F64));
        GATO_SHAPES(X)
#undef X
        return t;
    }();
    return v;
}

const Ops *find_ops(int S, int C, int dtype)
{
    for (const Ops &o : all_ops())
        if (o.S == S && o.C == C && o.dtype == dtype) return &o;
    return nullptr;
}

}  // namespace gato

using namespace gato;

#define GATO_ETA_HIST_MAX 4096

// ---- solver object -----------------------------------------------------------------------------
struct gato_solver {
    Dims d;
    int dtype, device;
    size_t esz;
    const Ops *ops;
    int num_cus;
    // options
    int pcg_mode, pcg_threads, pcg_groups;
    int wave_pub;                       // option: per-wave published partials in launches of up to 32 workgroups (default 1)
    // arena
    char *arena;
    size_t arena_bytes;
    void *G_dense, *C_dense, *Ginv, *Sbd, *Pbd, *gamma, *lambda, *dz;
    int *iters, *status;
    double *final_eta;
    unsigned long long *slots;
    PcgStreamWork sw;
    PcgPlan plan;
    // device copies of host CSR inputs for gato_linsys_solve_* (sized on first use)
    char *in_arena;
    size_t in_bytes;
    char *pin;            // pinned host staging (inputs, then iters | lambda | dz)
    size_t pin_bytes;
    int last_groups, last_threads, last_mode, last_variant, last_semi;
    int time_pcg, stamp_pcg, ablate, xcd_sel, no_single_lds, true_warm_start, no_pair, plan_pair, pcg_variant, xcd_pack;
    hipEvent_t ev_pcg0, ev_pcg1;
    hipEvent_t ev_cal0, ev_cal1;         // XCD calibration of the one-XCD launches
    long long xcd_cal_key;              // geometry the choice below was measured for (0 = none yet)
    int xcd_cal_best, last_xcd_sel;
    int tuning;                         // inside gato_solver_tune: pcg_one plans, runs the trials on scratch and returns
    int *tune_iters, *tune_status;      // scratch words of the trial launches (never the caller's, never the sticky status)
    double *tune_eta;
    // knot-sharded PCG state (gato_shard_pcg_*)
    struct {
        int rank, nranks, k0, k1, grid, max_iters;
        double exit_tol;
        const char *S_full, *P_full, *gamma_full;
    } sh;
    char *ghosts;   // [r|p][ping-pong][left|right][S]
    int plan_semi, pcg_semi;   // semi-resident launch planned / option (-1 auto, 0 off)
    int plan_dpp, dpp_rows;    // DPP-row layout planned / option (-1 auto, 0 never, 1 wherever a plain launch fits)
    unsigned pcg_epoch;        // next free hand-off epoch (resident kernels)
    int pcg_launch_id;
    size_t slots_bytes;
    int asm_mode;       // option: 0 = auto, 1 = stage kernels one by one (convert / invert / schur / stair), 2 = fused launch (workgroup per knot)
    int last_asm_fused, stamp_asm, last_image;
    double *eta_hist;   // eta after init and after every iteration (option record_eta), GATO_ETA_HIST_MAX + 1 entries
    int record_eta;
    // hand-off time-outs: the status word holds the id of the most recent launch that timed out (never cleared by a
    // kernel); ids only grow, so "status differs from the last acknowledged value" = a time-out since the last check
    int status_ack;
    hipStream_t last_stream;          // stream of the most recent PCG launch (gato_pcg_status synchronises it)
    int timeout_ms;                   // option: bound of every in-kernel spin (default 2000)
    int precon_mode;                  // option: GATO_PRECON_* (whole-solve entries)
    int time_stages;                  // option: hipEvents around assembly / PCG / dz of the whole-solve entries
    hipEvent_t ev_stage[4];
    int cluster_flat;                 // option: 1 (default) = flat cluster exchange where it applies, 0 = always two levels
    int max_workgroups;               // option: CUs a persistent launch may count on (0 = all; ranks sharing one GPU in tests)
    int last_fallback;                // the most recent gato_solver_recover re-ran the PCG through the streaming kernels
    struct {                          // arguments of the most recent whole solve, for gato_solver_recover
        int valid;
        const void *S, *P, *gamma, *Cd, *g;
        void *lam, *dz;
        double exit_tol;
        int max_iters;
    } lc;
    // multi-GPU cluster (gato_cluster_*): this rank's mirror, the peers' mirrors as mapped here
    struct {
        int on, rank, nranks, k0, k1;
        unsigned long long *local;
        unsigned long long *peer[GATO_MAX_RANKS];
        bool opened[GATO_MAX_RANKS];
        size_t bytes, flat_off, lam_off;
        unsigned xepoch;
        int last_flat;
        int mem_kind;                 // 0 uncached, 1 fine-grained, 2 plain hipMalloc
        size_t alloc_bytes;           // size of the allocation behind local (>= bytes: recycled mirrors, mirror_take)
    } cl;
    struct { const void *Ginv, *Cd, *g; void *dz; } fz;   // set by the whole-solve entries: dz may ride in the PCG launch
    void *imgS, *imgP;                // column-major images of S and Pinv over all rows (one system; nullptr: none), see PcgLaunch::imgS
    int img_ld;
    int img_fresh;                    // the fused assembly launch of the whole solve in progress has just written them
    int no_image;                     // option: the one-workgroup kernels load from S_bd / P_bd as every other kernel
    int coop_launch;                  // option: multi-workgroup persistent launches through hipLaunchCooperativeKernel
    hipEvent_t host_ev[2];            // the host-pointer drop-in's timing events, kept across calls
    int dz_fused;                     // the most recent PCG launch also did the dz back-substitution (1: in the solving workgroup, 2: in helper blocks)
    int *dz_flag;                     // device word for the helper blocks of the one-workgroup fp64 launch
    int no_fuse_dz;                   // option
    unsigned long long **cl_tab;      // device copy of cl.peer (the kernel reads the peers' mirror addresses from it)
};

// ---- co-residency gate (A12: check_sms + cudaLaunchCooperativeKernel in the reference, gato_utils.cuh:829-854,
// gato_pcg.cuh:502-526).  The workgroups of a multi-workgroup persistent launch hand data to each other inside the
// launch, so all of them must be resident at once.  One launch alone always is (W <= CUs, one workgroup per CU); two
// launches on two streams of one process could each get half their workgroups and spin until the time-out.  Every
// such launch therefore records an event, and a launch that would not fit beside the launches still in flight on OTHER
// streams of the same device first makes its stream wait for them.  (Kernels of foreign processes cannot be seen here:
// that case ends in the bounded time-out and gato_solver_recover.)
namespace {
struct InFlight { hipEvent_t ev; int cus; hipStream_t st; int device; };
std::mutex g_gate_mu;
// held from the admission check over the launch to the record of its event: two host threads must not both find the chip free and
// both launch (round 5: four threads with a solver and a stream each ran into hand-off time-outs - check, then act, was not atomic)
std::mutex g_launch_mu;
std::vector<InFlight> g_inflight;
std::vector<InFlight> g_free_events;      // recycled events, kept with the device they were created on

int gate_before(int device, int num_cus, int need, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_gate_mu);
    size_t w = 0;
    for (size_t i = 0; i < g_inflight.size(); ++i) {
        if (hipEventQuery(g_inflight[i].ev) == hipSuccess) g_free_events.push_back(g_inflight[i]);
        else g_inflight[w++] = g_inflight[i];
    }
    g_inflight.resize(w);
    int busy = 0;
    for (const InFlight &f : g_inflight)
        if (f.device == device && f.st != st) busy += f.cus;
    if (busy + need > num_cus) {
        for (const InFlight &f : g_inflight)
            if (f.device == device && f.st != st) GATO_HIP_CHECK(hipStreamWaitEvent(st, f.ev, 0));
    }
    return GATO_OK;
}

int gate_after(int device, int need, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_gate_mu);
    hipEvent_t ev = nullptr;
    for (size_t i = 0; i < g_free_events.size(); ++i)
        if (g_free_events[i].device == device) {          // an event belongs to the device it was created on
            ev = g_free_events[i].ev;
            g_free_events[i] = g_free_events.back();
            g_free_events.pop_back();
            break;
        }
    if (!ev) GATO_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GATO_HIP_CHECK(hipEventRecord(ev, st));
    g_inflight.push_back(InFlight{ev, need, st, device});
    return GATO_OK;
}
}  // namespace

static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

extern "C" const char *gato_last_error(void) { return g_err; }
extern "C" int gato_version(void) { return 100; }

extern "C" int gato_num_shapes(void)
{
    int n = 0;
#define X(S_, C_) ++n;
    GATO_SHAPES(X)
#undef X
    return n;
}

extern "C" int gato_shape(int i, int *S, int *C)
{
    int n = 0;
#define X(S_, C_)              \
    if (n++ == i) {            \
        *S = S_; *C = C_;      \
        return GATO_OK;        \
    }
    GATO_SHAPES(X)
#undef X
    return GATO_EINVAL;
}

extern "C" int gato_device_info(int device, int *num_cus, int *lds_bytes, char *name, int name_len)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device visible");
        return GATO_ENODEV;
    }
    hipDeviceProp_t p;
    GATO_HIP_CHECK(hipGetDeviceProperties(&p, device));
    if (num_cus) *num_cus = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    return GATO_OK;
}

extern "C" int gato_infer_shape(const int *C_row, int len_C_row, int len_g, int len_c, int *S, int *C, int *K)
{
    // S = number of leading rows of C holding exactly one entry (row-block 0 = identity,
    // gato_schur.cuh:725); then K = len_c / S and C from N = (S+C)K - C.
    if (!C_row || len_C_row != len_c + 1 || len_c <= 0 || len_g <= 0) {
        set_error("infer_shape: len(C_row)=%d must be len(c)+1=%d", len_C_row, len_c + 1);
        return GATO_EINVAL;
    }
    int s = 0;
    while (s < len_c && C_row[s + 1] - C_row[s] == 1) ++s;
    if (s == 0 || len_c % s != 0) {
        // all rows single-entry (K == 1) or no identity block: fall back to the compiled shapes
        for (int i = 0; i < gato_num_shapes(); ++i) {
            int ss, cc;
            gato_shape(i, &ss, &cc);
            if (len_c % ss == 0) {
                int k = len_c / ss;
                if ((long long)(ss + cc) * k - cc == len_g) { *S = ss; *C = cc; *K = k; return GATO_OK; }
            }
        }
        set_error("infer_shape: cannot infer S from C_row (leading identity rows = %d, len(c) = %d)", s, len_c);
        return GATO_EINVAL;
    }
    int k = len_c / s;
    if (k == 1) {
        if (len_g != s) { set_error("infer_shape: K=1 needs len(g) == S"); return GATO_EINVAL; }
        *S = s; *K = 1; *C = 0;
        for (int i = 0; i < gato_num_shapes(); ++i) { int ss, cc; gato_shape(i, &ss, &cc); if (ss == s) *C = cc; }
        return GATO_OK;
    }
    long long num = (long long)len_g - (long long)s * k;   // = C*(K-1)
    if (num < 0 || num % (k - 1) != 0) {
        set_error("infer_shape: len(g)=%d inconsistent with S=%d K=%d", len_g, s, k);
        return GATO_EINVAL;
    }
    *S = s; *K = k; *C = (int)(num / (k - 1));
    return GATO_OK;
}

extern "C" int gato_solver_create_batched(int S, int C, int K, int B, int dtype, int device, gato_solver **out);
extern "C" int gato_cluster_destroy(gato_solver *s);
extern "C" int gato_cluster_rewind(gato_solver *s);
extern "C" int gato_compute_dz(gato_solver *s, const void *d_Ginv_dense, const void *d_C_dense, const void *d_g,
                               const void *d_lambda, void *d_dz, void *stream);
extern "C" int gato_pcg(gato_solver *s, const void *d_S, const void *d_Pinv, const void *d_gamma, void *d_lambda,
                        double exit_tol, int max_iters, int *d_iters, void *stream);
extern "C" int gato_solver_tune(gato_solver *s, void *stream);
extern "C" int gato_solver_destroy(gato_solver *s);

extern "C" int gato_solver_create(int S, int C, int K, int dtype, int device, gato_solver **out)
{
    return gato_solver_create_batched(S, C, K, 1, dtype, device, out);
}

extern "C" int gato_solver_create_batched(int S, int C, int K, int B, int dtype, int device, gato_solver **out)
{
    if (B < 1 || B > 65535) {           // the batched launches take one grid row (blockIdx.y) per system
        set_error("solver_create: batch must be in 1 .. 65535 (got %d); split larger batches over several calls", B);
        return GATO_EINVAL;
    }
    if (!out || K < 1 || (dtype != GATO_F32 && dtype != GATO_F64)) {
        set_error("solver_create: bad arguments (K=%d dtype=%d)", K, dtype);
        return GATO_EINVAL;
    }
    const Ops *ops = find_ops(S, C, dtype);
    if (!ops) {
        set_error("solver_create: (STATE_SIZE=%d, CONTROL_SIZE=%d) is not a compiled shape", S, C);
        return GATO_ESHAPE;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
        set_error("solver_create: no HIP device %d (visible devices: %d)", device, count);
        return GATO_ENODEV;
    }
    GATO_HIP_CHECK(hipSetDevice(device));
    gato_solver *s = new gato_solver();
    memset(s, 0, sizeof(*s));
    s->d = Dims{S, C, K};
    s->d.B = B;
    s->dtype = dtype;
    s->device = device;
    s->esz = dtype == GATO_F32 ? 4 : 8;
    s->ops = ops;
    hipDeviceProp_t p;
    GATO_HIP_CHECK(hipGetDeviceProperties(&p, device));
    s->num_cus = p.multiProcessorCount;
    ops->pcg_plan(&s->plan);
    s->pcg_mode = GATO_PCG_AUTO;
    s->xcd_pack = -1;
    s->xcd_sel = -1;
    s->wave_pub = 1;
    s->dpp_rows = -1;
    s->pcg_semi = -1;
    s->timeout_ms = 2000;
    s->cluster_flat = 1;

    const Dims &d = s->d;
    const size_t e = s->esz;
    const int max_groups = 4096;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes ? bytes : 8); return o; };
    // status block first, granules right behind it: one memset re-initialises both before a launch
    const size_t o_status = take(64);
    const int slot_g = pcg_slot_granules(S, (int)e) > pcg_slot_granules_cg1(S, (int)e) ? pcg_slot_granules(S, (int)e)
                                                                                        : pcg_slot_granules_cg1(S, (int)e);
    const size_t o_slots = take((size_t)2 * 256 * slot_g * 8);
    s->slots_bytes = (size_t)2 * 256 * slot_g * 8;
    const size_t nb = (size_t)B;
    const size_t o_G = take(d.g_dense() * e * nb), o_C = take(d.c_dense() * e * nb), o_Gi = take(d.g_dense() * e * nb);
    const size_t o_S = take(d.bd() * e * nb), o_P = take(d.bd() * e * nb), o_gam = take(d.sk() * e * nb);
    const size_t o_lam = take(d.sk() * e * nb), o_dz = take(d.N() * e * nb);
    const size_t o_its = take(sizeof(int) * nb);
    const size_t o_vec = take(6 * d.sk() * e);
    const size_t o_part = take((size_t)4 * max_groups * e), o_scal = take(64 * 8), o_done = take(64);
    const size_t o_gh = take((size_t)8 * S * e);
    const size_t o_hist = take(sizeof(double) * (GATO_ETA_HIST_MAX + 1));
    const size_t o_xtab = take(sizeof(void *) * GATO_MAX_RANKS);
    // images for the one-workgroup two-rows-per-lane kernels (one system whose rows fit one of them)
    const long long rows = (long long)K * S;
    const bool img = B == 1 && ((s->plan.pair_threads > 0 && rows <= 2ll * s->plan.pair_threads) || (s->plan.mixed_rows > 0 && rows <= s->plan.mixed_rows));
    int img_ld = 0;
    if (img) {
        img_ld = (int)((rows + 63) / 64 * 64) + 128;                    // every lane of the launch reads inside its column
        if (img_ld < s->plan.mixed_rows) img_ld = s->plan.mixed_rows;
    }
    const size_t o_iS = take(img ? (size_t)3 * S * img_ld * e : 0), o_iP = take(img ? (size_t)3 * S * img_ld * e : 0);
    s->arena_bytes = off;
    GATO_HIP_CHECK(hipMalloc((void **)&s->arena, off));
    GATO_HIP_CHECK(hipMemset(s->arena, 0, off));
    char *a = s->arena;
    s->slots = (unsigned long long *)(a + o_slots);
    s->status = (int *)(a + o_status);
    s->iters = B > 1 ? (int *)(a + o_its) : (int *)(a + o_status + 8);
    s->final_eta = (double *)(a + o_status + 16);
    s->tune_status = (int *)(a + o_status + 32); s->tune_iters = (int *)(a + o_status + 40); s->tune_eta = (double *)(a + o_status + 48);
    s->dz_flag = (int *)(a + o_status + 56);
    s->G_dense = a + o_G; s->C_dense = a + o_C; s->Ginv = a + o_Gi;
    s->Sbd = a + o_S; s->Pbd = a + o_P; s->gamma = a + o_gam; s->lambda = a + o_lam; s->dz = a + o_dz;
    s->sw.vecs = a + o_vec;
    s->sw.partials = a + o_part; s->sw.scalars = a + o_scal; s->sw.done = (int *)(a + o_done);
    s->sw.max_groups = max_groups;
    s->ghosts = a + o_gh;
    s->eta_hist = (double *)(a + o_hist);
    s->cl_tab = (unsigned long long **)(a + o_xtab);
    if (img) { s->imgS = a + o_iS; s->imgP = a + o_iP; s->img_ld = img_ld; }
    // placement of the one-XCD launches of the default geometry: measured here, where the caller waits anyway
    // (allocation, memset), never inside an enqueue-only entry.  GATO_NO_TUNE=1 skips it (XCD 0).
    const char *nt = getenv("GATO_NO_TUNE");
    if (!(nt && atoi(nt))) {
        const int rc = gato_solver_tune(s, nullptr);
        if (rc) { gato_solver_destroy(s); return rc; }
    }
    *out = s;
    return GATO_OK;
}

extern "C" int gato_solver_destroy(gato_solver *s)
{
    if (!s) return GATO_OK;
    (void)hipSetDevice(s->device);
    gato_cluster_destroy(s);
    if (s->ev_pcg0) (void)hipEventDestroy(s->ev_pcg0);
    if (s->ev_pcg1) (void)hipEventDestroy(s->ev_pcg1);
    if (s->ev_cal0) (void)hipEventDestroy(s->ev_cal0);
    if (s->ev_cal1) (void)hipEventDestroy(s->ev_cal1);
    for (int i = 0; i < 4; ++i)
        if (s->ev_stage[i]) (void)hipEventDestroy(s->ev_stage[i]);
    if (s->arena) (void)hipFree(s->arena);
    if (s->in_arena) (void)hipFree(s->in_arena);
    for (int i = 0; i < 2; ++i)
        if (s->host_ev[i]) (void)hipEventDestroy(s->host_ev[i]);
    if (s->pin) (void)hipHostFree(s->pin);
    delete s;
    return GATO_OK;
}

extern "C" void *gato_solver_buffer(gato_solver *s, int which)
{
    switch (which) {
        case 0: return s->G_dense;
        case 1: return s->C_dense;
        case 2: return s->Ginv;
        case 3: return s->Sbd;
        case 4: return s->Pbd;
        case 5: return s->gamma;
        case 6: return s->lambda;
        case 7: return s->dz;
        case 8: return s->iters;
        case 9: return (unsigned long long *)s->sw.scalars + 8;   // diagnostic stamps (option stamp_pcg)
        case 10: return s->eta_hist;                              // double[max_iters + 1] (option record_eta)
        default: return nullptr;
    }
}

extern "C" int gato_solver_set_option(gato_solver *s, const char *name, int value)
{
    if (!strcmp(name, "pcg_mode")) s->pcg_mode = value;
    else if (!strcmp(name, "pcg_threads")) s->pcg_threads = value;
    else if (!strcmp(name, "pcg_groups")) s->pcg_groups = value;
    else if (!strcmp(name, "asm_mode")) s->asm_mode = value;
    else if (!strcmp(name, "pcg_semi")) s->pcg_semi = value;
    else if (!strcmp(name, "pcg_epoch")) s->pcg_epoch = (unsigned)value;      // test hook: place the counter near its wrap
    else if (!strcmp(name, "cluster_epoch")) s->cl.xepoch = (unsigned)value;  // test hook: the cluster's counter near its end (every rank alike)
    else if (!strcmp(name, "stamp_asm")) s->stamp_asm = value;
    else if (!strcmp(name, "stamp_pcg")) s->stamp_pcg = value;
    else if (!strcmp(name, "ablate")) s->ablate = value;
    else if (!strcmp(name, "no_single_lds")) s->no_single_lds = value;
    else if (!strcmp(name, "no_pair")) s->no_pair = value;
    else if (!strcmp(name, "wave_pub")) s->wave_pub = value;
    else if (!strcmp(name, "dpp_rows")) s->dpp_rows = value;
    else if (!strcmp(name, "pcg_variant")) s->pcg_variant = value;
    else if (!strcmp(name, "record_eta")) s->record_eta = value;
    else if (!strcmp(name, "xcd_pack")) s->xcd_pack = value;
    else if (!strcmp(name, "xcd_sel")) s->xcd_sel = value < 0 ? -1 : (value & 7);
    else if (!strcmp(name, "true_warm_start")) s->true_warm_start = value;
    else if (!strcmp(name, "timeout_ms")) s->timeout_ms = value > 0 ? value : 2000;
    else if (!strcmp(name, "max_workgroups")) s->max_workgroups = value;
    else if (!strcmp(name, "no_fuse_dz")) s->no_fuse_dz = value;
    else if (!strcmp(name, "no_image")) s->no_image = value;
    else if (!strcmp(name, "coop_launch")) s->coop_launch = value;
    else if (!strcmp(name, "cluster_flat")) s->cluster_flat = value;
    else if (!strcmp(name, "knot_lo") || !strcmp(name, "knot_hi")) {          // stage-level entries: knots [knot_lo, knot_hi)
        if (value < 0 || value > s->d.K) { set_error("%s = %d is outside [0, %d]", name, value, s->d.K); return GATO_EINVAL; }
        (name[5] == 'l' ? s->d.k_lo : s->d.k_hi) = value;
    }
    else if (!strcmp(name, "precon_mode")) {
        if (value < GATO_PRECON_STAIR || value > GATO_PRECON_POINT_JACOBI) { set_error("precon_mode must be 0, 1 or 2"); return GATO_EINVAL; }
        s->precon_mode = value;
    }
    else if (!strcmp(name, "time_stages")) {
        s->time_stages = value;
        if (value && !s->ev_stage[0])
            for (int i = 0; i < 4; ++i) GATO_HIP_CHECK(hipEventCreate(&s->ev_stage[i]));
    }
    else if (!strcmp(name, "batch_nnz_G")) s->d.nnzG = value;
    else if (!strcmp(name, "batch_nnz_C")) s->d.nnzC = value;
    else if (!strcmp(name, "time_pcg")) {
        s->time_pcg = value;
        if (value && !s->ev_pcg0) {
            GATO_HIP_CHECK(hipEventCreate(&s->ev_pcg0));
            GATO_HIP_CHECK(hipEventCreate(&s->ev_pcg1));
        }
    }
    else { set_error("unknown option %s", name); return GATO_EINVAL; }
    return GATO_OK;
}

extern "C" int gato_pcg_last_ms(gato_solver *s, float *ms)
{
    if (!s->time_pcg || !s->ev_pcg0) { set_error("pcg_last_ms: option time_pcg is off"); return GATO_EINVAL; }
    GATO_HIP_CHECK(hipEventSynchronize(s->ev_pcg1));
    GATO_HIP_CHECK(hipEventElapsedTime(ms, s->ev_pcg0, s->ev_pcg1));
    return GATO_OK;
}

// Stage times of the most recent whole solve as data (the reference prints "Forming Schur took" and the solve time,
// gato_schur.cuh:907-913,972-982; gpu_library.cu:186-198): ms[0] = CSR scatter + Schur + preconditioner, ms[1] = PCG,
// ms[2] = dz.  Needs option time_stages = 1; synchronises on the last event.
extern "C" int gato_last_stage_ms(gato_solver *s, float *ms)
{
    if (!s->time_stages || !s->ev_stage[0]) { set_error("last_stage_ms: option time_stages is off"); return GATO_EINVAL; }
    GATO_HIP_CHECK(hipEventSynchronize(s->ev_stage[3]));
    for (int i = 0; i < 3; ++i) GATO_HIP_CHECK(hipEventElapsedTime(ms + i, s->ev_stage[i], s->ev_stage[i + 1]));
    return GATO_OK;
}

extern "C" int gato_solver_get_option(gato_solver *s, const char *name, int *value)
{
    if (!strcmp(name, "pcg_mode")) *value = s->pcg_mode;
    else if (!strcmp(name, "pcg_threads")) *value = s->pcg_threads;
    else if (!strcmp(name, "pcg_groups")) *value = s->pcg_groups;
    else if (!strcmp(name, "last_groups")) *value = s->last_groups;
    else if (!strcmp(name, "last_threads")) *value = s->last_threads;
    else if (!strcmp(name, "last_mode")) *value = s->last_mode;
    else if (!strcmp(name, "last_pair")) *value = s->plan_pair;
    else if (!strcmp(name, "last_dpp")) *value = s->plan_dpp;
    else if (!strcmp(name, "last_xcd_sel")) *value = s->last_xcd_sel;
    else if (!strcmp(name, "last_variant")) *value = s->last_variant;
    else if (!strcmp(name, "asm_mode")) *value = s->asm_mode;
    else if (!strcmp(name, "last_asm_fused")) *value = s->last_asm_fused;
    else if (!strcmp(name, "last_image")) *value = s->last_image;
    else if (!strcmp(name, "last_semi")) *value = s->last_semi;
    else if (!strcmp(name, "last_fallback")) *value = s->last_fallback;
    else if (!strcmp(name, "last_dz_fused")) *value = s->dz_fused;
    else if (!strcmp(name, "last_cluster_flat")) *value = s->cl.on ? s->cl.last_flat : 0;
    else if (!strcmp(name, "timeout_ms")) *value = s->timeout_ms;
    else if (!strcmp(name, "precon_mode")) *value = s->precon_mode;
    else if (!strcmp(name, "cluster_mem_kind")) *value = s->cl.on ? s->cl.mem_kind : -1;
    else if (!strcmp(name, "num_cus")) *value = s->num_cus;
    else if (!strcmp(name, "batch")) *value = s->d.B;
    else if (!strcmp(name, "max_semi_knots"))
        *value = s->plan.semi_threads > 0 ? (s->plan.semi_threads / s->d.S + s->plan.semi_rows * s->plan.semi_threads / s->d.S) * (s->num_cus < 256 ? s->num_cus : 256) : 0;
    else if (!strcmp(name, "max_resident_knots")) *value = s->plan.max_knots_per_wg * (s->num_cus < 256 ? s->num_cus : 256);
    else { set_error("unknown option %s", name); return GATO_EINVAL; }
    return GATO_OK;
}

// ---- stage-level entry points -------------------------------------------------------------------
extern "C" int gato_convert(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val,
                            const int *d_C_row, const int *d_C_col, const void *d_C_val, double rho,
                            void *d_G_dense, void *d_C_dense, void *stream)
{
    if (s->d.B > 1 && (s->d.nnzG <= 0 || s->d.nnzC <= 0)) {
        set_error("convert: a batched solver needs the per-system nnz (gato_linsys_device_batched, or options "
                  "batch_nnz_G / batch_nnz_C)");
        return GATO_EINVAL;
    }
    return s->ops->convert(s->d, d_G_row, d_G_col, d_G_val, d_C_row, d_C_col, d_C_val, rho, d_G_dense, d_C_dense,
                           nullptr, (hipStream_t)stream);
}

extern "C" int gato_form_schur(gato_solver *s, const void *d_G_dense, const void *d_C_dense, const void *d_g,
                               const void *d_c, void *d_S, void *d_Pinv, void *d_gamma, void *d_Ginv_dense,
                               void *stream)
{
    return s->ops->form_schur(s->d, d_G_dense, d_C_dense, d_g, d_c, d_S, d_Pinv, d_gamma, d_Ginv_dense, false,
                              (hipStream_t)stream);
}

extern "C" int gato_form_ss(gato_solver *s, const void *d_S, void *d_Pinv, void *stream)
{
    return s->ops->form_ss(s->d, d_S, d_Pinv, (hipStream_t)stream);
}

// Geometry of the resident launch.  One workgroup per CU at most (all workgroups must be
// co-resident: they hand partial dots and halo blocks to each other inside the launch).
static int plan_resident_k(gato_solver *s, int K, int *groups, int *threads, int *kpw);
static int plan_resident(gato_solver *s, int *groups, int *threads, int *kpw)
{
    return plan_resident_k(s, s->d.K, groups, threads, kpw);
}

// Geometry of a plain launch in the DPP-row layout (L lanes per knot); 0 if K does not fit max_wg such workgroups.
static int plan_dpp_rows(gato_solver *s, int K, int L, int max_wg, int *groups, int *threads, int *kpw)
{
    const int maxT = s->plan.max_threads;
    int t = s->pcg_threads, g = s->pcg_groups;
    if (t > 0) {
        t = (t + 63) / 64 * 64;
        if (t > maxT) t = maxT;
        if (t < 64) t = 64;
    }
    if (g > 0 && t == 0) {
        const int k_per = (K + g - 1) / g;
        t = (k_per * L + 63) / 64 * 64;
        if (t > maxT) return 0;
    }
    if (t == 0) {
        if (K * L <= maxT) t = (K * L + 63) / 64 * 64;
        else {
            t = maxT < 512 ? maxT : 512;
            while (t < maxT && (K + (t / L) - 1) / (t / L) > max_wg) t += 64;
            if ((K + (t / L) - 1) / (t / L) > 32 && (K + (maxT / L) - 1) / (maxT / L) <= 32) {      // one XCD if larger workgroups get there
                while (t < maxT && (K + (t / L) - 1) / (t / L) > 32) t += 64;
            }
        }
    }
    const int k_per_max = t / L;
    if (k_per_max < 1) return 0;
    int W = (K + k_per_max - 1) / k_per_max;
    if (g > 0 && g >= W) W = g;
    if (W > max_wg) return 0;
    const int k_per = (K + W - 1) / W;
    W = (K + k_per - 1) / k_per;
    *groups = W; *threads = t; *kpw = k_per;
    return 1;
}

// K = knots the launch works on (the system's, or one rank's shard of it)
static int plan_resident_k(gato_solver *s, int K, int *groups, int *threads, int *kpw)
{
    const int S = s->d.S;
    int max_wg = s->num_cus < 256 ? s->num_cus : 256;
    if (s->max_workgroups > 0 && s->max_workgroups < max_wg) max_wg = s->max_workgroups;   // CUs this solver may count on
    int t = s->pcg_threads;
    int g = s->pcg_groups;
    const int maxT = s->plan.max_threads;
    s->plan_semi = 0;
    s->plan_dpp = 0;
    // DPP-row layout of the plain / cluster launches (option dpp_rows: -1 auto, 0 never, 1 wherever such a launch fits): auto
    // leaves the one-workgroup special kernels (two rows per lane) and one-workgroup-per-system batches alone
    if (s->plan.dpp_lanes > 0 && s->dpp_rows != 0 && s->stamp_pcg != 1) {
        const int L = s->plan.dpp_lanes;
        const bool one_wg_kernel = t == 0 && g <= 1 &&
            ((!s->no_pair && s->plan.pair_threads > 0 && K * (S / 2) <= s->plan.pair_threads) ||
             (!s->no_pair && !s->no_single_lds && s->plan.mixed_rows > 0 && K * S <= s->plan.mixed_rows && K * L > maxT && !s->cl.on && K == s->d.K) ||
             (K * S > maxT && K * S <= s->plan.single_max_threads && !s->no_single_lds));
        const bool batch_split = s->d.B > 1 && K * L > maxT && K * S <= maxT;          // a batch needs one workgroup per system
        // measured (tools/dpp_ab.py, same box, with the lean hand-off): fp64 14/7/512 2.76 -> 2.56 us per iteration, 14/7/1024
        // 2.94 -> 2.75, 32/16/1024 6.35 -> 5.10, one workgroup 14/7/20 1.71 -> 1.35 (14/7/4096 and 12/6/300 equal); fp32 keeps
        // the LDS windows with packed FMAs: 14/7/512 2.17 against 2.21, 32/16/256 2.53 against 2.78 (the DPP-row kernel spills
        // there), 32/16/1024 3.73 against 3.70, 12/6/300 1.86 against 2.13 (idle lanes cost workgroups)
        // (cluster launches still run the older hand-off, where the LDS-window kernel at S = 32 is the slower one: one-GPU
        //  rehearsal of 32/16/1024 f32 over 2 ranks 4.62 us per iteration with DPP rows, 4.9 without)
        const bool pays = s->esz == 8 || (S > 16 && s->cl.on);
        if ((s->dpp_rows > 0 || (pays && !one_wg_kernel && !batch_split)) && plan_dpp_rows(s, K, L, max_wg, groups, threads, kpw)) {
            s->plan_pair = 0;
            s->plan_dpp = 1;
            return 1;
        }
    }
    if (t > 0) {
        t = (t + 63) / 64 * 64;
        if (t > maxT) t = maxT;
        if (t < 64) t = 64;
    }
    if (g > 0 && t == 0) {
        int k_per = (K + g - 1) / g;
        t = (k_per * S + 63) / 64 * 64;
        if (t < 64) t = 64;
        if (t > maxT) return 0;
    }
    s->plan_pair = 0;
    if (t == 0 && g <= 1 && !s->no_pair && s->plan.pair_threads > 0 && K * (S / 2) <= s->plan.pair_threads) {
        // fp32: one workgroup, two rows per lane (packed FMAs, half the waves)
        *groups = 1; *threads = (K * (S / 2) + 63) / 64 * 64; *kpw = K;
        s->plan_pair = 1;
        return 1;
    }
    // (from the size at which the one-row-per-lane launch no longer fits ONE workgroup: K S > maxT, or - where the DPP-row layout
    //  would be taken, 16 lanes per knot - K 16 > maxT: 14/7/33..36 fp64 ran as two workgroups at 2.22 us per iteration)
    const int one_row_lanes = (s->plan.dpp_lanes > 0 && s->dpp_rows != 0 && s->esz == 8) ? s->plan.dpp_lanes : S;
    if (t == 0 && g <= 1 && !s->no_pair && !s->no_single_lds && s->plan.mixed_rows > 0 && K * S <= s->plan.mixed_rows && K * one_row_lanes > maxT &&
        s->stamp_pcg != 1 && !s->cl.on && K == s->d.K) {
        // fp64 beyond the register-resident single workgroup: one workgroup, two rows per lane in part of the waves
        *groups = 1; *threads = s->plan.mixed_threads; *kpw = K;
        s->plan_pair = 2;
        return 1;
    }
    if (t == 0) {
        // auto: one workgroup while the problem fits one CU's registers (no inter-CU traffic at all);
        // otherwise 512-thread workgroups (measured best on MI355X: 2 waves per SIMD hide the LDS latency
        // of the operand window, and W stays small enough that one wave sweeps all partial granules),
        // growing only if that would need more workgroups than CUs.
        if (K * S <= maxT) t = (K * S + 63) / 64 * 64;
        else if (K * S <= s->plan.single_max_threads && g <= 1 && !s->no_single_lds) {
            *groups = 1; *threads = (K * S + 63) / 64 * 64; *kpw = K;      // one CU, Pinv rows partly in LDS
            return 1;
        }
        else {
            t = maxT < 512 ? maxT : 512;
            while (t < maxT && (long long)((K + (t / S) - 1) / (t / S)) > max_wg) t += 64;
            // up to 32 workgroups fit one XCD (cheaper hand-offs): take larger workgroups if that gets there
            if ((K + (t / S) - 1) / (t / S) > 32 && (K + (maxT / S) - 1) / (maxT / S) <= 32) {
                while (t < maxT && (K + (t / S) - 1) / (t / S) > 32) t += 64;
            }
        }
        if (t < 64) t = 64;
    }
    int k_per_max = t / S;
    if (k_per_max < 1) return 0;
    int W = (K + k_per_max - 1) / k_per_max;
    if (g > 0 && g >= W) W = g;
    if (W > max_wg) {
        // beyond the register file: one workgroup per CU, the knots a workgroup has no lanes for become extra rows whose
        // matrix entries are re-read from memory every product (option pcg_semi: -1 auto, 0 never)
        // option pcg_semi: -1 auto, 0 never (streaming kernels), 1 semi-resident, 2 no resident rows
        if (s->pcg_semi == 0 || s->pcg_threads > 0 || s->pcg_groups > 0) return 0;
        const int kp = (K + max_wg - 1) / max_wg;
        const int Wx = (K + kp - 1) / kp;
        if (Wx < 2) return 0;
        const int xt = s->plan.semi_threads, nt = s->plan.nores_threads;
        const bool semi_ok = xt > 0 && (long long)(kp - xt / S) * S <= (long long)s->plan.semi_rows * xt;
        const bool nores_ok = nt > 0 && (long long)kp * S <= (long long)s->plan.nores_rows * nt;
        // the LDS-DMA ring (option pcg_semi = 3; auto: once the bytes of S and Pinv that ONE launch streams per product are well past
        // the 256 MB Infinity Cache, i.e. the re-read block rows come from HBM; below that the semi-resident launch is served by the
        // caches and wins).  Measured cross-overs against the semi-resident launch (tools/ring_crossover.py, profiles/r05_ring_crossover.log):
        // 14/7 f32 K ~ 90 000 (420 MB), 32/16 f32 K ~ 28 000 (690 MB: its semi-resident launch already reads at 6 TB/s), 14/7 f64
        // between K = 49 152 (462 MB: semi 79.6 / ring 82.9 us per iteration) and K = 65 536 (617 MB: 107.7 / 104.6; driver sweep of
        // round 4: 107.9 / 97.3) - the fp64 ring serves up to 256 knots per workgroup, so auto takes it from 550 MB up to K = 65 536.
        // (K = the knots of THIS launch: a rank's shard in a cluster - what matters is what one GPU streams per product.  A cluster
        //  judges by the LARGEST shard, ceil(K_system / ranks), on every rank: shards differ by a knot and neighbouring ranks must
        //  not land on different sides of the threshold.)
        const bool dma_ok = !s->true_warm_start && s->ops->pcg_dma_max_knots() > 0 && kp <= s->ops->pcg_dma_max_knots();
        const double K_rule = s->cl.on && s->cl.nranks > 0 ? (double)((s->d.K + s->cl.nranks - 1) / s->cl.nranks) : (double)K;
        const double ring_from = S > 16 ? 700e6 : (s->esz == 8 ? 550e6 : 450e6);
        const bool beyond_cache = 2.0 * 3.0 * S * S * K_rule * (double)s->esz > ring_from;
        int which = 0;
        if (s->pcg_semi == 1) which = semi_ok ? 1 : 0;
        else if (s->pcg_semi == 2) which = nores_ok ? 2 : 0;
        else if (s->pcg_semi == 3) which = dma_ok ? 3 : 0;
        else which = (dma_ok && beyond_cache) ? 3 : semi_ok ? 1 : (nores_ok ? 2 : 0);
        if (!which) return 0;
        *groups = Wx; *threads = which == 1 ? xt : which == 2 ? nt : 512; *kpw = kp;
        s->plan_semi = which;
        return 1;
    }
    int k_per = (K + W - 1) / W;                     // balanced
    W = (K + k_per - 1) / k_per;
    *groups = W; *threads = t; *kpw = k_per;
    return 1;
}

// Geometry of the single-reduction variant: a workgroup's lanes cover its own knots plus one ghost-lane knot per side.
static int plan_cg1_k(gato_solver *s, int K, int *groups, int *threads, int *kpw);
static int plan_cg1(gato_solver *s, int *groups, int *threads, int *kpw) { return plan_cg1_k(s, s->d.K, groups, threads, kpw); }
// K = knots the launch works on (the system's, or one rank's shard of it)
static int plan_cg1_k(gato_solver *s, int K, int *groups, int *threads, int *kpw)
{
    const int S = s->d.S;
    int max_wg = s->num_cus < 256 ? s->num_cus : 256;
    if (s->max_workgroups > 0 && s->max_workgroups < max_wg) max_wg = s->max_workgroups;   // CUs this solver may count on
    const int maxT = s->ops->pcg_cg1_max_threads();
    int t = s->pcg_threads > 0 ? (s->pcg_threads + 63) / 64 * 64 : 0;
    if (t > maxT) t = maxT;
    if (t == 0) {
        if ((K + 2) * S <= maxT) t = ((K + 2) * S + 63) / 64 * 64;
        else {
            t = maxT < 512 ? maxT : 512;
            while (t < maxT && (K + (t / S - 2) - 1) / (t / S - 2) > max_wg) t += 64;
        }
    }
    if (t < 4 * S) t = (4 * S + 63) / 64 * 64;
    if (t > maxT) return 0;
    const int per = t / S - 2;
    if (per < 2) return 0;
    int W = (K + per - 1) / per;
    if (s->pcg_groups > W) W = s->pcg_groups;
    if (W > max_wg) return 0;
    int k_per = (K + W - 1) / W;
    W = (K + k_per - 1) / k_per;
    // every workgroup needs two knots of its own (its two edge blocks on either side go to the neighbours).  When the even split
    // leaves the last workgroup ONE knot, the launcher takes the balanced split instead (sizes k_per and k_per - 1: launch_pcg_cg1)
    if (W > 1 && (k_per < 2 || (K - (W - 1) * k_per < 2 && k_per < 3))) return 0;
    *groups = W; *threads = t; *kpw = k_per;
    return 1;
}

// One-XCD launches (xcd_pack): the hand-off granules live in one place in memory and the eight XCDs are not equally far
// from it - measured 3.00 (best XCD) to 3.27 us (worst) per iteration at 14/7/512 f32, 3.10 to 3.37 at 14/7/1024, the
// order depending on where this solver's hand-off area happened to land, stable for the life of the solver
// (tools/xcd_sel_check.py).  The hosting XCD of a geometry is therefore MEASURED: two rounds of eight short trial launches
// (16 iterations each, the second round timed with HIP events; ~1 ms in all, host-blocking), the fastest XCD is kept.
// This happens in gato_solver_tune() only - called by gato_solver_create for the geometry the solver's defaults plan,
// and by the caller again after changing geometry options - on the solver's OWN buffers (work vectors as lambda, a
// scratch iters / status / eta word): the enqueue-only entries (gato_pcg, gato_linsys_device, ...) never calibrate, never
// wait on the host and never touch caller buffers for it; a geometry without a measurement runs on XCD 0.
static int calibrate_xcd(gato_solver *s, const PcgLaunch &a0, bool cg1, hipStream_t st, int *best, bool *measured)
{
    *best = 0;
    *measured = false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return GATO_OK; }
    if (cap != hipStreamCaptureStatusNone || a0.max_iters < 4) return GATO_OK;
    if (!s->ev_cal0) {
        GATO_HIP_CHECK(hipEventCreate(&s->ev_cal0));
        GATO_HIP_CHECK(hipEventCreate(&s->ev_cal1));
    }
    PcgLaunch t = a0;
    t.max_iters = a0.max_iters < 16 ? a0.max_iters : 16;
    t.exit_tol = 0.0;
    t.eta_hist = nullptr;
    t.dz = nullptr;
    t.lambda0 = nullptr;
    t.stamps = nullptr; t.diag = 0; t.ablate = 0;
    t.timeout_ticks = 2000000ull;                       // 20 ms: a trial never waits the solver's 2 s
    t.ev_start = s->ev_cal0; t.ev_stop = s->ev_cal1;
    const unsigned need = 2u * (unsigned)t.max_iters + 8u;
    if (s->pcg_epoch > 0xFFFFFFFFu - 17u * need - 64u) {                    // counter about to wrap: start over on zeroed granules
        GATO_HIP_CHECK(hipMemsetAsync(s->slots, 0, s->slots_bytes, st));
        s->pcg_epoch = 0;
    }
    float best_ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        for (int sel = 0; sel < 8; ++sel) {
            t.xcd_sel = sel;
            t.epoch0 = s->pcg_epoch;
            s->pcg_epoch += need;
            if (++s->pcg_launch_id <= 0) s->pcg_launch_id = 1;
            t.launch_id = s->pcg_launch_id;
            int rc;
            {
                std::lock_guard<std::mutex> launch_lock(g_launch_mu);
                if ((rc = gate_before(s->device, s->num_cus, s->num_cus, st))) return rc;
                rc = cg1 ? s->ops->pcg_cg1(t, st) : s->ops->pcg_resident(t, st);
                if (rc == GATO_OK) rc = gate_after(s->device, s->num_cus, st);
            }
            if (rc) return rc;
            GATO_HIP_CHECK(hipEventSynchronize(s->ev_cal1));
            float ms = 0.f;
            GATO_HIP_CHECK(hipEventElapsedTime(&ms, s->ev_cal0, s->ev_cal1));
            // a 16-iteration trial is ~50 us: one that took 10 ms sat in a hand-off (the CUs are shared with another process,
            // its spin bound is the trial time-out) - give up, the launches run on XCD 0, instead of paying 16 time-outs
            if (ms > 10.f) { *best = 0; return GATO_OK; }
            if (pass == 1 && (sel == 0 || ms < best_ms)) { best_ms = ms; *best = sel; }
        }
    }
    *measured = true;
    return GATO_OK;
}

static bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
    return cap != hipStreamCaptureStatusNone;
}

static int pcg_one(gato_solver *s, const void *d_S, const void *d_Pinv, const void *d_gamma, void *d_lambda,
                   double exit_tol, int max_iters, int *d_iters, int batch, hipStream_t st)
{
    int groups = 0, threads = 0, kpw = 0;
    int mode = s->pcg_mode;
    // pcg_variant: 1 = single-reduction recurrence (opt-in, gato_pcg_cg1.hip)
    const bool cg1 = s->pcg_variant == 1 && mode != GATO_PCG_STREAMING && !s->true_warm_start &&
                     plan_cg1(s, &groups, &threads, &kpw) != 0 && (batch == 1 || groups == 1);
    const bool fits = cg1 || plan_resident(s, &groups, &threads, &kpw) != 0;
    if (cg1) s->plan_pair = 0;
    if (mode == GATO_PCG_AUTO) mode = fits ? GATO_PCG_RESIDENT : GATO_PCG_STREAMING;
    if (mode == GATO_PCG_RESIDENT) {
        if (!fits) {
            set_error("pcg: K=%d does not fit the resident kernel on %d CUs (threads=%d groups=%d)", s->d.K,
                      s->num_cus, s->pcg_threads, s->pcg_groups);
            return GATO_EINVAL;
        }
        // A launch captured into a graph is REPLAYED with the arguments of the capture.  That is fine for a one-workgroup solve
        // (nothing in it depends on the launch's number), but not for what draws fresh values per launch: the hand-off epochs of
        // the multi-workgroup launches (on replay the granules already hold them: polls would pass on stale payloads) and the
        // dz flag of the helper blocks (it would already equal the launch id: dz from an unfinished lambda).  So while the
        // stream is being captured the helper blocks do not do dz (the dz launch of its own follows), and a launch that needs
        // epochs is refused - the streaming kernels (pcg_mode = 2) replay correctly.
        const bool capturing = stream_is_capturing(st);
        if (capturing && !s->tuning && (groups > 1 || s->cl.on)) {
            set_error("pcg: a persistent launch of %d workgroups cannot be captured into a graph (its hand-off epochs are launch "
                      "arguments: a replay would read stale granules); capture the streaming kernels (option pcg_mode = 2) "
                      "or a system that fits one workgroup", groups);
            return GATO_EINVAL;
        }
        PcgLaunch a;
        memset(&a, 0, sizeof(a));
        a.S_bd = d_S; a.P_bd = d_Pinv; a.gamma = d_gamma; a.lambda = d_lambda;
        a.lambda0 = s->true_warm_start ? d_lambda : nullptr;      // in place: every lane reads its lambda0 first
        a.K = s->d.K; a.max_iters = max_iters; a.exit_tol = exit_tol;
        a.batch = batch;
        a.pair = s->plan_pair;
        // (the single-reduction kernel and the LDS-DMA ring keep the plain launch: option coop_launch serves the resident / semi-resident launches)
        a.coop = s->coop_launch && groups > 1 && batch == 1 && !cg1 && s->plan_semi != 3;
        // (every lane of the launch loads rows 2 tid, 2 tid + 1 resp. its own row: all of them must lie inside a column of the image)
        if (s->img_fresh && !s->no_image && batch == 1 && d_S == s->Sbd && d_Pinv == s->Pbd &&
            ((s->plan_pair == 1 && 2 * threads <= s->img_ld) || (s->plan_pair == 2 && s->plan.mixed_rows <= s->img_ld))) {
            a.imgS = s->imgS; a.imgP = s->imgP; a.img_ld = s->img_ld;
        }
        s->last_image = a.imgS != nullptr;
        a.semi = cg1 ? 0 : s->plan_semi;
        a.dpp_rows = cg1 ? 0 : s->plan_dpp;
        // option xcd_pack: -1 = auto (default): up to 32 workgroups are placed on ONE XCD (measured 15-20 % faster hand-offs:
        // 14/7/512 f32 3.96 -> 3.11 us/iteration); spreading over 2..7 XCDs measured no better than the plain grid, so
        // auto leaves larger launches alone.  0 = off, 1..7 = force that many XCDs (tools/xcd_pack_check.py).
        a.xcd_pack = 0;
        if (s->xcd_pack != 0 && batch == 1 && groups > 1) {
            const int need = (groups + 31) / 32;
            if (s->xcd_pack < 0) a.xcd_pack = need == 1 ? 1 : 0;
            else a.xcd_pack = (s->xcd_pack >= need && s->xcd_pack < 8) ? s->xcd_pack : 0;
        }
        if (a.semi) a.xcd_pack = 0;
        a.xcd_sel = s->xcd_sel;
        a.wave_pub = s->wave_pub;
        if (a.xcd_pack > 0 && groups > s->num_cus / 8) a.xcd_pack = 0;     // an XCD with fewer CUs than workgroups (CU mask): plain grid
        a.knots_per_wg = kpw; a.groups = groups; a.threads = threads;
        a.slots = s->slots; a.iters = d_iters ? d_iters : s->iters; a.status = s->status;
        // hand-off epochs: each launch gets a fresh range (two reductions per iteration plus the initial one)
        const unsigned need = max_iters > 0x3FFFFFF0 ? 0x80000000u : 2u * (unsigned)max_iters + 8u;
        if (s->pcg_epoch > 0xFFFFFFFFu - need - 8u) {            // counter about to wrap: start over on zeroed granules
            GATO_HIP_CHECK(hipMemsetAsync(s->slots, 0, s->slots_bytes, st));
            s->pcg_epoch = 0;
        }
        a.epoch0 = s->pcg_epoch;
        s->pcg_epoch += need;
        if (++s->pcg_launch_id <= 0) s->pcg_launch_id = 1;
        a.launch_id = s->pcg_launch_id;
        a.final_eta = s->final_eta;
        a.eta_hist = (s->record_eta && max_iters <= GATO_ETA_HIST_MAX) ? s->eta_hist : nullptr;
        a.timeout_ticks = (unsigned long long)s->timeout_ms * 100000ull;   // s_memrealtime runs at 100 MHz
        // one workgroup (per system) holds every lambda_k: the dz back-substitution rides in the same launch
        s->dz_fused = 0;
        // (batches only: every system's workgroup does its own dz and a launch of 25 600 one-wave workgroups goes away; for
        //  ONE system the single workgroup is as latency bound as that launch was - measured 11 us in the epilogue against
        //  5.3 us + a launch gap - unless asked for with no_fuse_dz = -1)
        // (fp32 two-rows-per-lane kernel: its epilogue exists for batches)
        if (s->fz.dz && (s->no_fuse_dz < 0 || (!s->no_fuse_dz && batch > 1)) && groups == 1 && !cg1 &&
            (s->plan_pair != 1 || batch > 1) && !a.semi && !s->stamp_pcg) {
            a.dz_Ginv = s->fz.Ginv; a.dz_Cd = s->fz.Cd; a.dz_g = s->fz.g; a.dz = s->fz.dz; a.C = s->d.C;
            s->dz_fused = 1;
        }
        // ONE system through a two-rows-per-lane one-workgroup kernel (pcg_single_f64m_kernel = BASELINE configs[1]; pcg_single_f32x2_kernel): its helper blocks - there
        // to warm the L2 - stay and do dz as soon as lambda is published: the dz launch and the gap in front of it (6.5 us of a
        // 215 us step) become ~1 us at the end of the PCG launch.  no_fuse_dz = 1 keeps the launch of its own.
        else if (s->fz.dz && !s->no_fuse_dz && batch == 1 && groups == 1 && !cg1 && (s->plan_pair == 2 || s->plan_pair == 1) && !s->stamp_pcg && !s->tuning && !capturing) {
            a.dz_Ginv = s->fz.Ginv; a.dz_Cd = s->fz.Cd; a.dz_g = s->fz.g; a.dz = s->fz.dz; a.C = s->d.C;
            a.dz_helpers = 1; a.dz_flag = s->dz_flag;
            s->dz_fused = 2;
        }
        a.ablate = s->ablate;
        a.stamps = s->stamp_pcg == 1 ? (unsigned long long *)s->sw.scalars + 8 : nullptr;
        a.diag = s->stamp_pcg;
        a.ev_start = s->time_pcg ? s->ev_pcg0 : nullptr;
        a.ev_stop = s->time_pcg ? s->ev_pcg1 : nullptr;
        s->last_groups = groups; s->last_threads = threads; s->last_mode = GATO_PCG_RESIDENT;
        s->last_variant = cg1 ? 1 : 0;
        s->last_semi = a.semi;
        s->last_stream = st;
        // co-residency: a multi-workgroup launch waits for launches on other streams it would not fit beside
        // (a one-XCD launch counts as the whole chip: two of them may be dealt to the same XCD)
        const bool gated = batch == 1 && groups > 1;
        const int need_cus = a.xcd_pack > 0 ? s->num_cus : groups;
        int rc;
        // one-XCD launches: which of the eight XCDs hosts them (option xcd_sel: -1 = measured once per geometry, 0..7 fixed)
        s->last_xcd_sel = -1;
        if (a.xcd_pack > 0) {
            if (s->xcd_sel >= 0) a.xcd_sel = s->xcd_sel;
            else {
                const long long key = ((long long)groups << 32) | ((long long)threads << 8) | (cg1 ? 2 : 0) | (s->esz == 8 ? 1 : 0) | 4 | (a.dpp_rows ? 8 : 0);
                if (s->tuning) {
                    // gato_solver_tune: the trial launches (scratch outputs, own events); no launch of the caller's follows
                    bool measured = false;
                    a.iters = s->tune_iters; a.status = s->tune_status; a.final_eta = s->tune_eta;
                    if ((rc = calibrate_xcd(s, a, cg1, st, &s->xcd_cal_best, &measured))) return rc;
                    s->xcd_cal_key = measured ? key : 0;
                    s->last_xcd_sel = measured ? s->xcd_cal_best : -1;
                    return GATO_OK;
                }
                a.xcd_sel = s->xcd_cal_key == key ? s->xcd_cal_best : 0;       // not measured for this geometry: XCD 0
            }
            s->last_xcd_sel = a.xcd_sel;
        }
        if (s->tuning) return GATO_OK;                                       // nothing to measure for this geometry
        const bool gate = gated && !capturing;      // (a captured multi-workgroup launch was refused above; the gate records events)
        std::unique_lock<std::mutex> launch_lock(g_launch_mu, std::defer_lock);
        if (gate) launch_lock.lock();
        if (gate && (rc = gate_before(s->device, s->num_cus, need_cus, st))) return rc;
        rc = cg1 ? s->ops->pcg_cg1(a, st) : a.semi == 3 ? s->ops->pcg_dma(a, st) : s->ops->pcg_resident(a, st);
        if (rc == GATO_OK && gate) rc = gate_after(s->device, need_cus, st);
        return rc;
    }
    if (s->tuning) return GATO_OK;
    s->last_stream = st;
    s->last_mode = GATO_PCG_STREAMING; s->last_groups = 0; s->last_threads = 0; s->last_semi = 0;
    s->dz_fused = 0;
    s->sw.warm_start = s->true_warm_start;
    s->sw.eta_hist = (s->record_eta && max_iters <= GATO_ETA_HIST_MAX) ? s->eta_hist : nullptr;
    if (s->time_pcg) GATO_HIP_CHECK(hipEventRecord(s->ev_pcg0, st));
    int rc = s->ops->pcg_streaming(s->d, d_S, d_Pinv, d_gamma, d_lambda, exit_tol, max_iters,
                                   d_iters ? d_iters : s->iters, s->sw, st);
    if (s->time_pcg) GATO_HIP_CHECK(hipEventRecord(s->ev_pcg1, st));
    return rc;
}

extern "C" int gato_pcg(gato_solver *s, const void *d_S, const void *d_Pinv, const void *d_gamma, void *d_lambda,
                        double exit_tol, int max_iters, int *d_iters, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int B = s->d.B;
    if (B == 1) return pcg_one(s, d_S, d_Pinv, d_gamma, d_lambda, exit_tol, max_iters, d_iters, 1, st);
    // batch: one workgroup per system in a single launch when a system fits one CU; otherwise system by system
    int groups = 0, threads = 0, kpw = 0;
    const bool fits = plan_resident(s, &groups, &threads, &kpw) != 0;
    int *its = d_iters ? d_iters : s->iters;
    if (fits && groups == 1 && s->pcg_mode != GATO_PCG_STREAMING)
        return pcg_one(s, d_S, d_Pinv, d_gamma, d_lambda, exit_tol, max_iters, its, B, st);
    const size_t e = s->esz;
    s->fz.dz = nullptr;                      // system by system: dz stays a launch of its own
    for (int b = 0; b < B; ++b) {
        int rc = pcg_one(s, (const char *)d_S + b * s->d.bd() * e, (const char *)d_Pinv + b * s->d.bd() * e,
                         (const char *)d_gamma + b * s->d.sk() * e, (char *)d_lambda + b * s->d.sk() * e, exit_tol,
                         max_iters, its + b, 1, st);
        if (rc) return rc;
    }
    return GATO_OK;
}

// Measures, for the geometry the solver's CURRENT options plan, which XCD should host a one-XCD launch (see above
// calibrate_xcd).  Blocking (~1 ms: 16 short launches, each waited for); runs on `stream`, reads the solver's own S / Pinv /
// gamma work buffers (whatever they hold: the launches run a fixed iteration count and their results are discarded) and
// writes only solver-owned scratch.  A no-op for batches, cluster ranks and geometries that are not one-XCD launches.
// gato_solver_create calls it once; call it again after changing pcg_threads / pcg_groups / pcg_variant / max_workgroups
// / xcd_pack if those launches should keep the measured placement (unmeasured geometries run on XCD 0: a placement
// hint only, results never depend on it).
extern "C" int gato_solver_tune(gato_solver *s, void *stream)
{
    if (!s) { set_error("solver_tune: null solver"); return GATO_EINVAL; }
    if (s->d.B != 1 || s->cl.on || s->xcd_sel >= 0 || s->pcg_mode == GATO_PCG_STREAMING) return GATO_OK;
    GATO_HIP_CHECK(hipSetDevice(s->device));
    const int saved_tws = s->true_warm_start, saved_stamp = s->stamp_pcg;
    hipStream_t saved_stream = s->last_stream;
    s->true_warm_start = 0; s->stamp_pcg = 0;
    s->tuning = 1;
    const int rc = pcg_one(s, s->Sbd, s->Pbd, s->gamma, s->sw.vecs, 0.0, 16, s->tune_iters, 1, (hipStream_t)stream);
    s->tuning = 0;
    s->true_warm_start = saved_tws; s->stamp_pcg = saved_stamp;
    s->last_stream = saved_stream;
    if (rc) return rc;
    GATO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return GATO_OK;
}

// Reports a hand-off time-out of ANY PCG launch since the previous call (the status word keeps the id of the most
// recent launch that timed out; no kernel ever clears it), after synchronising the stream of the latest launch.
// The same condition is visible in-band: the launch wrote iters = -1.
extern "C" int gato_pcg_status(gato_solver *s, int *status)
{
    int v = 0;
    GATO_HIP_CHECK(hipSetDevice(s->device));
    GATO_HIP_CHECK(hipStreamSynchronize(s->last_stream));
    GATO_HIP_CHECK(hipMemcpy(&v, s->status, sizeof(int), hipMemcpyDeviceToHost));
    const bool timed_out = v != s->status_ack;
    s->status_ack = v;
    if (status) *status = timed_out ? 1 : 0;
    if (timed_out) { set_error("pcg: in-kernel hand-off timed out (launch %d)", v); return GATO_ETIMEOUT; }
    return GATO_OK;
}

// A12 fallback: if a persistent launch of the most recent whole solve (gato_linsys_device / _blocks) gave up on a
// hand-off - its workgroups were not co-resident, e.g. another process held the CUs - the PCG is re-run through the
// streaming kernels (no inter-workgroup hand-off inside a launch, any residency) and dz is recomputed: a slower
// correct answer instead of an error.  Synchronises `stream`.  *recovered = 1 when that happened.
extern "C" int gato_solver_recover(gato_solver *s, int *recovered, void *stream)
{
    if (recovered) *recovered = 0;
    s->last_fallback = 0;
    int st_ = 0;
    const int rc = gato_pcg_status(s, &st_);
    if (rc == GATO_OK) return GATO_OK;
    if (rc != GATO_ETIMEOUT || !s->lc.valid) return rc;
    const int saved = s->pcg_mode;
    s->d.k_lo = s->d.k_hi = 0;
    s->pcg_mode = GATO_PCG_STREAMING;
    int rc2 = gato_pcg(s, s->lc.S, s->lc.P, s->lc.gamma, s->lc.lam, s->lc.exit_tol, s->lc.max_iters, s->iters, stream);
    s->pcg_mode = saved;
    if (rc2) return rc2;
    if (s->lc.dz && (rc2 = gato_compute_dz(s, s->Ginv, s->lc.Cd, s->lc.g, s->lc.lam, s->lc.dz, stream))) return rc2;
    GATO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    s->last_fallback = 1;
    if (recovered) *recovered = 1;
    return GATO_OK;
}

extern "C" int gato_compute_dz(gato_solver *s, const void *d_Ginv_dense, const void *d_C_dense, const void *d_g,
                               const void *d_lambda, void *d_dz, void *stream)
{
    return s->ops->compute_dz(s->d, d_Ginv_dense, d_C_dense, d_g, d_lambda, d_dz, (hipStream_t)stream);
}

// A1 + A2 + A3 for the whole-solve entries: ONE fused launch (assemble_kernel) where launch latency is what
// counts, the stage kernels one by one where throughput does (the fused workgroup recomputes its left neighbour's
// Schur block; option asm_mode: 0 = auto, 1 = stage kernels, 2 = fused).
static int assemble(gato_solver *s, int mode, const int *G_row, const int *G_col, const void *G_val, const int *C_row,
                    const int *C_col, const void *C_val, const void *C_dense, const void *d_g, const void *d_c, double rho,
                    hipStream_t st)
{
    int rc;
    s->d.k_lo = s->d.k_hi = 0;              // whole solves work on every knot: the knot-range option is for the stage entries
    // the fused launch always forms the stair blocks: the other preconditioner modes take the stage kernels
    const long long knots = (long long)s->d.K * s->d.B;
    // option asm_mode: 0 auto (2 while one round of workgroups covers the solve, else 1 - measured crossover),
    // 1 stage kernels, 2 one launch with a workgroup per knot (three-fold recomputation)
    const bool stair = s->precon_mode == GATO_PRECON_STAIR;
    const bool fused = stair && (s->asm_mode == 2 || (s->asm_mode == 0 && knots <= 2ll * s->num_cus));
    s->last_asm_fused = fused;
    s->img_fresh = 0;
    if (!fused) {
        if (mode == 0) {                 // CSR: the gather launch also inverts Q_k, R_k while they sit in LDS
            if (s->d.B > 1 && (s->d.nnzG <= 0 || s->d.nnzC <= 0)) return gato_convert(s, G_row, G_col, G_val, C_row, C_col, C_val, rho, s->G_dense, s->C_dense, st);
            rc = s->ops->convert(s->d, G_row, G_col, G_val, C_row, C_col, C_val, rho, s->G_dense, s->C_dense, s->Ginv, st);
        } else rc = s->ops->add_rho(s->d, G_val, rho, s->G_dense, st);
        if (rc) return rc;
        s->d.stair_follows = s->precon_mode == GATO_PRECON_STAIR;
        rc = s->ops->form_schur(s->d, s->G_dense, C_dense, d_g, d_c, s->Sbd, s->Pbd, s->gamma, s->Ginv, mode == 0, st);
        s->d.stair_follows = 0;
        if (rc) return rc;
        // preconditioner (gato_defines.h:9-10): the Schur stage leaves the block-Jacobi one (main blocks, zeros beside them)
        if (s->precon_mode == GATO_PRECON_BLOCK_JACOBI) return GATO_OK;                       // SS_PRECON = 0 (gato_schur.cuh:965-970)
        if (s->precon_mode == GATO_PRECON_POINT_JACOBI) return s->ops->point_jacobi(s->d, s->Sbd, s->Pbd, st);   // both 0 (:424-428)
        return gato_form_ss(s, s->Sbd, s->Pbd, st);
    }
    if (mode == 0 && s->d.B > 1 && (s->d.nnzG <= 0 || s->d.nnzC <= 0)) {
        set_error("a batched solver needs the per-system nnz (gato_linsys_device_batched, or options batch_nnz_G / batch_nnz_C)");
        return GATO_EINVAL;
    }
    AsmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = mode;
    a.G_row = G_row; a.G_col = G_col; a.G_val = G_val; a.C_row = C_row; a.C_col = C_col; a.C_val = C_val;
    a.rho = rho; a.g = d_g; a.c = d_c;
    a.Gd = s->G_dense; a.Cd = const_cast<void *>(C_dense); a.Ginv = s->Ginv; a.Sbd = s->Sbd; a.Pbd = s->Pbd; a.gamma = s->gamma;
    a.stamps = s->stamp_asm ? (unsigned long long *)s->sw.scalars + 8 : nullptr;
    if (s->imgS && !s->no_image && s->d.B == 1) {        // the workgroup-per-knot launch also writes the PCG images
        a.imgS = s->imgS; a.imgP = s->imgP; a.img_ld = s->img_ld;
        s->img_fresh = 1;
    }
    return s->ops->assemble(s->d, a, st);
}

extern "C" int gato_linsys_device(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val,
                                  const int *d_C_row, const int *d_C_col, const void *d_C_val, const void *d_g,
                                  const void *d_c, double exit_tol, int max_iters, double rho, void *d_lambda,
                                  void *d_dz, void *stream)
{
    int rc;
    void *lam = d_lambda ? d_lambda : s->lambda;
    void *dz = d_dz ? d_dz : s->dz;
    const bool ts = s->time_stages != 0;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[0], (hipStream_t)stream));
    if ((rc = assemble(s, 0, d_G_row, d_G_col, d_G_val, d_C_row, d_C_col, d_C_val, s->C_dense, d_g, d_c, rho, (hipStream_t)stream))) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[1], (hipStream_t)stream));
    s->lc = {1, s->Sbd, s->Pbd, s->gamma, s->C_dense, d_g, lam, dz, exit_tol, max_iters};
    s->fz = {s->Ginv, s->C_dense, d_g, dz};
    rc = gato_pcg(s, s->Sbd, s->Pbd, s->gamma, lam, exit_tol, max_iters, s->iters, stream);
    s->fz = {nullptr, nullptr, nullptr, nullptr};
    s->img_fresh = 0;
    if (rc) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[2], (hipStream_t)stream));
    if (!s->dz_fused && (rc = gato_compute_dz(s, s->Ginv, s->C_dense, d_g, lam, dz, stream))) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[3], (hipStream_t)stream));
    return GATO_OK;
}

extern "C" int gato_linsys_device_blocks(gato_solver *s, const void *d_G_blocks, const void *d_C_blocks, const void *d_g,
                                         const void *d_c, double exit_tol, int max_iters, double rho, void *d_lambda,
                                         void *d_dz, void *stream)
{
    int rc;
    void *lam = d_lambda ? d_lambda : s->lambda;
    void *dz = d_dz ? d_dz : s->dz;
    const bool ts = s->time_stages != 0;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[0], (hipStream_t)stream));
    if ((rc = assemble(s, 2, nullptr, nullptr, d_G_blocks, nullptr, nullptr, nullptr, d_C_blocks, d_g, d_c, rho, (hipStream_t)stream))) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[1], (hipStream_t)stream));
    s->lc = {1, s->Sbd, s->Pbd, s->gamma, d_C_blocks, d_g, lam, dz, exit_tol, max_iters};
    s->fz = {s->Ginv, d_C_blocks, d_g, dz};
    rc = gato_pcg(s, s->Sbd, s->Pbd, s->gamma, lam, exit_tol, max_iters, s->iters, stream);
    s->fz = {nullptr, nullptr, nullptr, nullptr};
    s->img_fresh = 0;
    if (rc) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[2], (hipStream_t)stream));
    if (!s->dz_fused && (rc = gato_compute_dz(s, s->Ginv, d_C_blocks, d_g, lam, dz, stream))) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[3], (hipStream_t)stream));
    return GATO_OK;
}

extern "C" int gato_shard_pcg_done(gato_solver *s, int *done, void *stream)
{
    GATO_HIP_CHECK(hipMemcpyAsync(done, s->sw.done, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GATO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return GATO_OK;
}

extern "C" int gato_linsys_device_batched(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val,
                                          int nnz_G, const int *d_C_row, const int *d_C_col, const void *d_C_val,
                                          int nnz_C, const void *d_g, const void *d_c, double exit_tol, int max_iters,
                                          double rho, void *d_lambda, void *d_dz, int *d_iters, void *stream)
{
    s->d.nnzG = nnz_G; s->d.nnzC = nnz_C;
    int rc = gato_linsys_device(s, d_G_row, d_G_col, d_G_val, d_C_row, d_C_col, d_C_val, d_g, d_c, exit_tol, max_iters,
                                rho, d_lambda, d_dz, stream);
    if (rc) return rc;
    if (d_iters && d_iters != s->iters)
        GATO_HIP_CHECK(hipMemcpyAsync(d_iters, s->iters, sizeof(int) * s->d.B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GATO_OK;
}

static std::mutex g_cache_mu;
static gato_solver *g_cached_solver = nullptr;

extern "C" int gato_release_cache(void)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g_cached_solver) gato_solver_destroy(g_cached_solver);
    g_cached_solver = nullptr;
    return GATO_OK;
}

// ---- host-pointer drop-in for main_call (gpu_library.cu:85-234) -----------------------------------
template <typename T>
static int linsys_solve_host(int dtype, const int *G_row, int len_G_row, const int *G_col, const T *G_val, int nnz_G,
                             const int *C_row, int len_C_row, const int *C_col, const T *C_val, int nnz_C,
                             const T *g, int len_g, const T *c, int len_c, const T *lambda_in, int S, int C, int K,
                             int testiters, T exit_tol, int max_iters, int warm_start, T rho, T *lambda_out,
                             T *dz_out, int *iters_out, float *ms_out)
{
    (void)lambda_in; (void)warm_start;   // D5: the reference resets lambda to 0 (gato_pcg.cuh:303)
    const long long N = (long long)(S + C) * K - C;
    if (len_G_row != N + 1 || len_C_row != (long long)S * K + 1 || len_g != N || len_c != S * K ||
        nnz_G < 0 || nnz_C < 0 || testiters < 1) {
        set_error("linsys_solve: lengths do not match S=%d C=%d K=%d: len(G_row)=%d (want %lld), len(C_row)=%d "
                  "(want %d), len(g)=%d (want %lld), len(c)=%d (want %d)",
                  S, C, K, len_G_row, N + 1, len_C_row, S * K + 1, len_g, N, len_c, S * K);
        return GATO_EINVAL;
    }
    // The scatter kernel trusts the CSR arrays (as the reference does, gato_schur.cuh:674-743); an out-of-range index
    // would be an out-of-bounds device write, so the host copy is validated here (O(nnz), the arrays are in cache).
    {
        auto bad = [&](const char *name, const int *row, int nrows, const int *col, int nnz, long long ncols) -> bool {
            if (row[0] != 0) { set_error("linsys_solve: %s_row[0] must be 0", name); return true; }
            for (int i = 0; i < nrows; ++i)
                if (row[i + 1] < row[i] || row[i + 1] > nnz) {
                    set_error("linsys_solve: %s_row is not a monotone indptr at row %d", name, i);
                    return true;
                }
            for (int i = 0; i < nnz; ++i)
                if (col[i] < 0 || col[i] >= ncols) {
                    set_error("linsys_solve: %s_col[%d] = %d is outside [0, %lld)", name, i, col[i], ncols);
                    return true;
                }
            return false;
        };
        if (bad("G", G_row, len_G_row - 1, G_col, nnz_G, N) || bad("C", C_row, len_C_row - 1, C_col, nnz_C, N)) return GATO_EINVAL;
    }
    if (G_row[len_G_row - 1] != nnz_G || C_row[len_C_row - 1] != nnz_C) {
        set_error("linsys_solve: indptr[-1] does not match nnz (G %d vs %d, C %d vs %d)", G_row[len_G_row - 1], nnz_G,
                  C_row[len_C_row - 1], nnz_C);
        return GATO_EINVAL;
    }
    // The reference allocates and frees 22 device buffers per call (gpu_library.cu:36-45,140-147; gato_pcg.cuh:486-492).
    // Here the solver of the most recent (S, C, K, dtype) and its input staging area are kept for the next call.
    std::lock_guard<std::mutex> lock(g_cache_mu);
    gato_solver *&cached = g_cached_solver;
    gato_solver *s = cached;
    int rc;
    if (!s || s->d.S != S || s->d.C != C || s->d.K != K || s->dtype != dtype || s->d.B != 1) {
        if (s) gato_solver_destroy(s);
        cached = s = nullptr;
        if ((rc = gato_solver_create(S, C, K, dtype, 0, &s))) return rc;
        cached = s;
    } else {
        (void)hipSetDevice(s->device);
    }
    const char *env = getenv("GATO_PCG_MODE");
    s->pcg_mode = env ? atoi(env) : GATO_PCG_AUTO;
    const char *envp = getenv("GATO_PRECON");        // 0 stair (the reference's default build), 1 block-Jacobi, 2 point-Jacobi
    s->precon_mode = envp ? atoi(envp) : GATO_PRECON_STAIR;
    if (s->precon_mode < GATO_PRECON_STAIR || s->precon_mode > GATO_PRECON_POINT_JACOBI) s->precon_mode = GATO_PRECON_STAIR;

    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += align_up(b ? b : 8); return o; };
    const size_t oGr = take(sizeof(int) * len_G_row), oGc = take(sizeof(int) * nnz_G), oGv = take(sizeof(T) * nnz_G);
    const size_t oCr = take(sizeof(int) * len_C_row), oCc = take(sizeof(int) * nnz_C), oCv = take(sizeof(T) * nnz_C);
    const size_t og = take(sizeof(T) * len_g), oc = take(sizeof(T) * len_c);
    hipError_t e = hipSuccess;
    if (s->in_bytes < off) {
        if (s->in_arena) (void)hipFree(s->in_arena);
        s->in_arena = nullptr; s->in_bytes = 0;
        e = hipMalloc((void **)&s->in_arena, off);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); return GATO_EHIP; }
        s->in_bytes = off;
    }
    char *a = s->in_arena;
    hipStream_t st = nullptr;
    if (!s->host_ev[0]) {
        (void)hipEventCreate(&s->host_ev[0]);
        (void)hipEventCreate(&s->host_ev[1]);
    }
    const hipEvent_t ev0 = s->host_ev[0], ev1 = s->host_ev[1];
    auto fail = [&](int code) { return code; };
    // one H2D transfer: the eight input arrays are packed into a pinned staging buffer laid out like the device
    // arena (the reference issues eight blocking cudaMemcpy from pageable memory, gpu_library.cu:150-157)
    // lambda and dz are neighbours in the solver's arena: ONE D2H copy brings both (and the padding between them)
    const size_t dz_off = (size_t)((const char *)s->dz - (const char *)s->lambda), out_span = dz_off + sizeof(T) * (size_t)N;
    if (s->pin_bytes < off + 64 + out_span) {
        if (s->pin) (void)hipHostFree(s->pin);
        s->pin = nullptr; s->pin_bytes = 0;
        const size_t want = off + 64 + out_span + 256;
        if ((e = hipHostMalloc((void **)&s->pin, want, hipHostMallocDefault)) != hipSuccess) {
            set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return fail(GATO_EHIP);
        }
        s->pin_bytes = want;
    }
    memcpy(s->pin + oGr, G_row, sizeof(int) * len_G_row); memcpy(s->pin + oGc, G_col, sizeof(int) * nnz_G);
    memcpy(s->pin + oGv, G_val, sizeof(T) * nnz_G);       memcpy(s->pin + oCr, C_row, sizeof(int) * len_C_row);
    memcpy(s->pin + oCc, C_col, sizeof(int) * nnz_C);     memcpy(s->pin + oCv, C_val, sizeof(T) * nnz_C);
    memcpy(s->pin + og, g, sizeof(T) * len_g);            memcpy(s->pin + oc, c, sizeof(T) * len_c);
    if ((e = hipMemcpyAsync(a, s->pin, off, hipMemcpyHostToDevice, st)) != hipSuccess) {
        set_error("H2D copy failed: %s", hipGetErrorString(e));
        return fail(GATO_EHIP);
    }
    char *pout = s->pin + off;                               // pinned landing area: iters | lambda | dz
    int iters = 0;
    for (int i = 0; i < testiters; ++i) {                       // gpu_library.cu:169-192
        (void)hipEventRecord(ev0, st);
        rc = gato_linsys_device(s, (const int *)(a + oGr), (const int *)(a + oGc), a + oGv, (const int *)(a + oCr),
                                (const int *)(a + oCc), a + oCv, a + og, a + oc, (double)exit_tol, max_iters,
                                (double)rho, nullptr, nullptr, st);
        if (rc) return fail(rc);
        if ((e = hipMemcpyAsync(pout + 64, s->lambda, out_span, hipMemcpyDeviceToHost, st)) != hipSuccess ||
            (e = hipMemcpyAsync(pout, s->iters, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) {
            set_error("D2H copy failed: %s", hipGetErrorString(e));
            return fail(GATO_EHIP);
        }
        (void)hipEventRecord(ev1, st);
        if ((e = hipEventSynchronize(ev1)) != hipSuccess) {
            set_error("solve failed: %s", hipGetErrorString(e));
            return fail(GATO_EHIP);
        }
        iters = *(const int *)pout;
        if (iters < 0) {
            // in-band time-out mark of a persistent launch (its workgroups were not co-resident): slower correct answer
            // through the streaming kernels instead of an error, then fetch the results again
            int recovered = 0;
            if ((rc = gato_solver_recover(s, &recovered, st))) return fail(rc);
            if ((e = hipMemcpy(pout + 64, s->lambda, out_span, hipMemcpyDeviceToHost)) != hipSuccess ||
                (e = hipMemcpy(pout, s->iters, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) {
                set_error("D2H copy failed: %s", hipGetErrorString(e));
                return fail(GATO_EHIP);
            }
            (void)hipEventRecord(ev1, st);
            (void)hipEventSynchronize(ev1);
            iters = *(const int *)pout;
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev0, ev1);
        if (ms_out) ms_out[i] = ms;
        if (i == 0 && iters_out) *iters_out = iters;            // the reference prints the first run's count (:189-191)
    }
    memcpy(lambda_out, pout + 64, sizeof(T) * (size_t)S * K);
    memcpy(dz_out, pout + 64 + dz_off, sizeof(T) * (size_t)N);
    return fail(GATO_OK);
}

extern "C" int gato_linsys_solve_f32(const int *G_row, int len_G_row, const int *G_col, const float *G_val, int nnz_G,
                                     const int *C_row, int len_C_row, const int *C_col, const float *C_val, int nnz_C,
                                     const float *g, int len_g, const float *c, int len_c, const float *lambda_in,
                                     int S, int C, int K, int testiters, float exit_tol, int max_iters, int warm_start,
                                     float rho, float *lambda_out, float *dz_out, int *iters_out, float *ms_out)
{
    return linsys_solve_host<float>(GATO_F32, G_row, len_G_row, G_col, G_val, nnz_G, C_row, len_C_row, C_col, C_val,
                                    nnz_C, g, len_g, c, len_c, lambda_in, S, C, K, testiters, exit_tol, max_iters,
                                    warm_start, rho, lambda_out, dz_out, iters_out, ms_out);
}

extern "C" int gato_linsys_solve_f64(const int *G_row, int len_G_row, const int *G_col, const double *G_val, int nnz_G,
                                     const int *C_row, int len_C_row, const int *C_col, const double *C_val, int nnz_C,
                                     const double *g, int len_g, const double *c, int len_c, const double *lambda_in,
                                     int S, int C, int K, int testiters, double exit_tol, int max_iters, int warm_start,
                                     double rho, double *lambda_out, double *dz_out, int *iters_out, float *ms_out)
{
    return linsys_solve_host<double>(GATO_F64, G_row, len_G_row, G_col, G_val, nnz_G, C_row, len_C_row, C_col, C_val,
                                     nnz_C, g, len_g, c, len_c, lambda_in, S, C, K, testiters, exit_tol, max_iters,
                                     warm_start, rho, lambda_out, dz_out, iters_out, ms_out);
}

// ---- knot-sharded PCG (multi-GPU) ------------------------------------------------------------------
static char *ghost_ptr(gato_solver *s, int vec /*0 r, 1 p*/, int pp, int side)
{
    return s->ghosts + ((size_t)((vec * 2 + pp) * 2 + side) * s->d.S) * s->esz;
}

static void shard_vectors(gato_solver *s, char *r[2], char *p[2], char **ups, char **rt)
{
    const size_t sk = s->d.sk() * s->esz;
    char *v = (char *)s->sw.vecs;
    r[0] = v; r[1] = v + sk; p[0] = v + 2 * sk; p[1] = v + 3 * sk; *ups = v + 4 * sk; *rt = v + 5 * sk;
}

static void shard_common(gato_solver *s, StreamStep &a)
{
    memset(&a, 0, sizeof(a));
    a.K = s->sh.k1 - s->sh.k0;
    a.max_iters = s->sh.max_iters; a.exit_tol = s->sh.exit_tol; a.done = s->sw.done; a.iters = s->iters;
    a.first_global = s->sh.k0 == 0; a.last_global = s->sh.k1 == s->d.K;
}

extern "C" int gato_shard_pcg_init(gato_solver *s, int rank, int nranks, int k0, int k1, const void *d_S,
                                   const void *d_Pinv, const void *d_gamma, double exit_tol, int max_iters,
                                   void *d_send, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int S = s->d.S;
    const size_t e = s->esz;
    if (rank < 0 || rank >= nranks || k0 < 0 || k1 <= k0 || k1 > s->d.K || (rank == 0) != (k0 == 0) ||
        (rank == nranks - 1) != (k1 == s->d.K)) {
        set_error("shard_pcg_init: bad shard rank=%d/%d knots [%d,%d) of %d", rank, nranks, k0, k1, s->d.K);
        return GATO_EINVAL;
    }
    s->sh.rank = rank; s->sh.nranks = nranks; s->sh.k0 = k0; s->sh.k1 = k1; s->sh.max_iters = max_iters;
    s->sh.exit_tol = exit_tol;
    s->sh.S_full = (const char *)d_S; s->sh.P_full = (const char *)d_Pinv; s->sh.gamma_full = (const char *)d_gamma;
    s->sh.grid = s->ops->stream_grid(k1 - k0, s->sw.max_groups);
    GATO_HIP_CHECK(hipMemsetAsync(s->lambda, 0, s->d.sk() * e, st));
    char *r[2], *p[2], *ups, *rt;
    shard_vectors(s, r, p, &ups, &rt);
    StreamStep a;
    shard_common(s, a);
    a.M = s->sh.P_full + (size_t)k0 * 3 * S * S * e;
    a.a_old = s->sh.gamma_full + (size_t)k0 * S * e;
    a.gh_a_left = s->sh.gamma_full + (size_t)(k0 > 0 ? k0 - 1 : 0) * S * e;
    a.gh_a_right = s->sh.gamma_full + (size_t)(k1 < s->d.K ? k1 : 0) * S * e;
    a.gh_new_left = ghost_ptr(s, 0, 0, 0); a.gh_new_right = ghost_ptr(s, 0, 0, 1);
    a.a_new = r[0]; a.y = rt; a.lam = (char *)s->lambda + (size_t)k0 * S * e;
    a.part_out = s->sw.partials; a.it = 0;
    int rc;
    if ((rc = s->ops->stream_step(0, a, s->sh.grid, st))) return rc;
    return s->ops->stream_pack(s->sw.partials, s->sh.grid, rt, k1 - k0, d_send, st);
}

extern "C" int gato_shard_pcg_phase_a(gato_solver *s, int it, const void *d_recvB_cur, const void *d_recvB_prev,
                                      void *d_send, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int S = s->d.S, REC = 2 * S + 1, rank = s->sh.rank;
    const size_t e = s->esz;
    char *r[2], *p[2], *ups, *rt;
    shard_vectors(s, r, p, &ups, &rt);
    const int pi = it & 1;
    StreamStep a;
    shard_common(s, a);
    a.M = s->sh.S_full + (size_t)s->sh.k0 * 3 * S * S * e;
    a.a_old = p[pi ^ 1]; a.b = rt; a.a_new = p[pi]; a.y = ups; a.it = it;
    a.part_num = d_recvB_cur; a.num_n = s->sh.nranks; a.num_stride = REC;
    a.part_den = d_recvB_prev; a.den_n = s->sh.nranks; a.den_stride = REC;
    const char *rb = (const char *)d_recvB_cur;                 // r~ blocks of the neighbours
    a.gh_b_left = rb + ((size_t)(rank > 0 ? rank - 1 : 0) * REC + 1 + S) * e;
    a.gh_b_right = rb + ((size_t)(rank + 1 < s->sh.nranks ? rank + 1 : 0) * REC + 1) * e;
    a.gh_a_left = ghost_ptr(s, 1, pi ^ 1, 0); a.gh_a_right = ghost_ptr(s, 1, pi ^ 1, 1);
    a.gh_new_left = ghost_ptr(s, 1, pi, 0); a.gh_new_right = ghost_ptr(s, 1, pi, 1);
    char *PA = (char *)s->sw.partials + (size_t)3 * s->sw.max_groups * e;
    a.part_out = PA;
    int rc;
    if ((rc = s->ops->stream_step(1, a, s->sh.grid, st))) return rc;
    return s->ops->stream_pack(PA, s->sh.grid, ups, s->sh.k1 - s->sh.k0, d_send, st);
}

extern "C" int gato_shard_pcg_phase_b(gato_solver *s, int it, const void *d_recvB_cur, const void *d_recvA,
                                      void *d_send, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int S = s->d.S, REC = 2 * S + 1, rank = s->sh.rank;
    const size_t e = s->esz;
    char *r[2], *p[2], *ups, *rt;
    shard_vectors(s, r, p, &ups, &rt);
    const int ri = it & 1, pi = it & 1;
    StreamStep a;
    shard_common(s, a);
    a.M = s->sh.P_full + (size_t)s->sh.k0 * 3 * S * S * e;
    a.a_old = r[ri]; a.b = ups; a.a_new = r[ri ^ 1]; a.y = rt; a.it = it;
    a.lam = (char *)s->lambda + (size_t)s->sh.k0 * S * e; a.p_cur = p[pi];
    a.part_num = d_recvB_cur; a.num_n = s->sh.nranks; a.num_stride = REC;   // eta(it)
    a.part_den = d_recvA; a.den_n = s->sh.nranks; a.den_stride = REC;       // v(it)
    const char *ra = (const char *)d_recvA;                     // upsilon blocks of the neighbours
    a.gh_b_left = ra + ((size_t)(rank > 0 ? rank - 1 : 0) * REC + 1 + S) * e;
    a.gh_b_right = ra + ((size_t)(rank + 1 < s->sh.nranks ? rank + 1 : 0) * REC + 1) * e;
    a.gh_a_left = ghost_ptr(s, 0, ri, 0); a.gh_a_right = ghost_ptr(s, 0, ri, 1);
    a.gh_new_left = ghost_ptr(s, 0, ri ^ 1, 0); a.gh_new_right = ghost_ptr(s, 0, ri ^ 1, 1);
    a.part_out = s->sw.partials;
    int rc;
    if ((rc = s->ops->stream_step(2, a, s->sh.grid, st))) return rc;
    return s->ops->stream_pack(s->sw.partials, s->sh.grid, rt, s->sh.k1 - s->sh.k0, d_send, st);
}

extern "C" int gato_shard_pcg_finish(gato_solver *s, const void *d_recvB_last, void *d_lambda_full_out, int *d_iters,
                                     void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = s->ops->stream_finish(d_recvB_last, s->sh.nranks, 2 * s->d.S + 1, s->sh.exit_tol, s->sh.max_iters - 1,
                                   s->sw.done, s->iters, s->final_eta, nullptr, st);
    if (rc) return rc;
    if (d_lambda_full_out && d_lambda_full_out != s->lambda)
        GATO_HIP_CHECK(hipMemcpyAsync(d_lambda_full_out, s->lambda, s->d.sk() * s->esz, hipMemcpyDeviceToDevice, st));
    if (d_iters && d_iters != s->iters)
        GATO_HIP_CHECK(hipMemcpyAsync(d_iters, s->iters, sizeof(int), hipMemcpyDeviceToDevice, st));
    return GATO_OK;
}

// ---- multi-GPU cluster: the persistent PCG launch with a device-initiated cross-GPU hand-off level --------------------
// NEW work: the reference is single-device (gato_utils.cuh:831) and has no communication layer.
// One process per GPU.  Every rank owns a MIRROR - a few KB of fine-grained device memory, IPC-shared - into which the
// peers store {epoch, payload} granules with system-scope stores over xGMI; a rank only ever polls its own mirror.  See
// pcg_resident_kernel<..., MR = true> for the protocol.  RCCL (gato_shard_pcg_*) stays as the portable fallback.
// Mirrors are RECYCLED inside the process, never handed back to the allocator while it lives: pages that were mapped uncached
// and come back as ordinary (cached) device memory after hipFree can read stale - round 5, tools/cluster_fuzz.py: a solver arena
// allocated over a freed uncached mirror read whole 128-B lines of zeros where the mirror's polled lines had been (P / gamma rows
// of a later solve; only with the uncached kind, not with fine-grained or plain mirrors).  A few hundred KB per mirror.
namespace {
struct MirrorBuf { void *p; size_t bytes; int device, kind; };
std::mutex g_mirror_mu;
std::vector<MirrorBuf> g_mirror_pool;

void *mirror_take(int device, int kind, size_t bytes, size_t *got)
{
    std::lock_guard<std::mutex> lock(g_mirror_mu);
    for (size_t i = 0; i < g_mirror_pool.size(); ++i) {
        const MirrorBuf b = g_mirror_pool[i];
        if (b.device == device && b.kind == kind && b.bytes >= bytes && b.bytes <= 4 * bytes) {
            g_mirror_pool[i] = g_mirror_pool.back();
            g_mirror_pool.pop_back();
            *got = b.bytes;
            return b.p;
        }
    }
    return nullptr;
}

void mirror_give(void *p, int device, int kind, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_mirror_mu);
    g_mirror_pool.push_back(MirrorBuf{p, bytes, device, kind});
}
}  // namespace

static int cluster_alloc(gato_solver *s)
{
    const char *env = getenv("GATO_XMEM");           // uncached | finegrained | plain (default: first that works)
    const int first = env ? (!strcmp(env, "plain") ? 2 : !strcmp(env, "finegrained") ? 1 : 0) : 0;
    void *p = nullptr;
    s->cl.alloc_bytes = s->cl.bytes;
    for (int kind = first; kind < 3; ++kind) {
        if ((p = mirror_take(s->device, kind, s->cl.bytes, &s->cl.alloc_bytes))) { s->cl.mem_kind = kind; break; }
        hipError_t e = kind == 0 ? hipExtMallocWithFlags(&p, s->cl.bytes, hipDeviceMallocUncached)
                     : kind == 1 ? hipExtMallocWithFlags(&p, s->cl.bytes, hipDeviceMallocFinegrained)
                                 : hipMalloc(&p, s->cl.bytes);
        if (e == hipSuccess && p) { s->cl.mem_kind = kind; break; }
        (void)hipGetLastError();
        p = nullptr;
    }
    if (!p) { set_error("cluster: cannot allocate the %zu-byte mirror", s->cl.bytes); return GATO_EHIP; }
    s->cl.local = (unsigned long long *)p;
    GATO_HIP_CHECK(hipMemset(p, 0, s->cl.bytes));
    GATO_HIP_CHECK(hipDeviceSynchronize());
    return GATO_OK;
}

extern "C" int gato_cluster_knot_range(int K, int rank, int nranks, int *k0, int *k1)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || K < nranks) {
        set_error("cluster: cannot shard %d knots over %d ranks (rank %d)", K, nranks, rank);
        return GATO_EINVAL;
    }
    const int base = K / nranks, extra = K % nranks;           // balanced contiguous ranges, as dist.knot_ranges
    *k0 = rank * base + (rank < extra ? rank : extra);
    *k1 = *k0 + base + (rank < extra ? 1 : 0);
    return GATO_OK;
}

extern "C" int gato_cluster_create(gato_solver *s, int rank, int nranks, void *ipc_handle_out)
{
    if (s->d.B != 1 || nranks > GATO_MAX_RANKS) {
        set_error("cluster: one system per solver, at most %d ranks", GATO_MAX_RANKS);
        return GATO_EINVAL;
    }
    int k0, k1, rc;
    if ((rc = gato_cluster_knot_range(s->d.K, rank, nranks, &k0, &k1))) return rc;
    GATO_HIP_CHECK(hipSetDevice(s->device));
    gato_cluster_destroy(s);
    memset(&s->cl, 0, sizeof(s->cl));
    s->cl.rank = rank; s->cl.nranks = nranks; s->cl.k0 = k0; s->cl.k1 = k1;
    // two-level area (2 parities), then the flat area: a slot for each of up to 256 workgroups of the whole cluster
    s->cl.flat_off = align_up((size_t)2 * pcg_xslot_granules(s->d.S, (int)s->esz), 16);
    // ... then the lambda ghost block a rank receives from its right neighbour at the end of a launch (cluster_lambda_ghost)
    s->cl.lam_off = s->cl.flat_off + (size_t)2 * 256 * pcg_flat_slot_granules(s->d.S, (int)s->esz);
    const size_t need = (s->cl.lam_off + (size_t)pcg_lamghost_granules(s->d.S, (int)s->esz)) * 8;
    s->cl.bytes = need < 65536 ? 65536 : align_up(need, 65536);
    if ((rc = cluster_alloc(s))) return rc;
    s->cl.peer[rank] = s->cl.local;
    if (ipc_handle_out) {
        hipIpcMemHandle_t h;
        GATO_HIP_CHECK(hipIpcGetMemHandle(&h, s->cl.local));
        static_assert(sizeof(h) == 64, "ipc handle size");
        memcpy(ipc_handle_out, &h, sizeof(h));
    }
    return GATO_OK;
}

extern "C" void *gato_cluster_local_mirror(gato_solver *s) { return s->cl.local; }

// handles: nranks x 64 bytes in rank order (other processes' mirrors are opened through them), and / or ptrs: mirrors
// that are plain device pointers in THIS process (ranks living in one process).  After this call and BEFORE the first
// gato_cluster_pcg every rank must pass a host-level barrier (torch.distributed.barrier): the mirrors are zeroed here.
extern "C" int gato_cluster_connect(gato_solver *s, const void *ipc_handles, void *const *ptrs)
{
    if (!s->cl.local) { set_error("cluster_connect: gato_cluster_create first"); return GATO_EINVAL; }
    GATO_HIP_CHECK(hipSetDevice(s->device));
    for (int r = 0; r < s->cl.nranks; ++r) {
        if (r == s->cl.rank) continue;
        if (ptrs && ptrs[r]) { s->cl.peer[r] = (unsigned long long *)ptrs[r]; continue; }
        if (!ipc_handles) { set_error("cluster_connect: no mirror given for rank %d", r); return GATO_EINVAL; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)ipc_handles + (size_t)r * sizeof(h), sizeof(h));
        void *p = nullptr;
        GATO_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        s->cl.peer[r] = (unsigned long long *)p;
        s->cl.opened[r] = true;
    }
    GATO_HIP_CHECK(hipMemcpy(s->cl_tab, s->cl.peer, sizeof(void *) * GATO_MAX_RANKS, hipMemcpyHostToDevice));
    s->cl.on = 1;
    return gato_cluster_rewind(s);                 // fresh epoch spaces on both levels (every rank does the same, then the caller's barrier)
}

// The hand-off epochs of a cluster only grow (32 bits; a launch takes 2 max_iters + 8 of them on every rank alike), and a mirror
// cannot be re-zeroed in stream order as the one-GPU slots are: a peer that is already in its next launch may have stored into it.
// So the epoch space is renewed by the CALLER, on every rank at the same solve: when gato_cluster_launches_left says that the next
// launch does not fit (the counters run in lock-step, every rank sees it at the same call), each rank waits for its own
// launches, all ranks pass a host barrier (nobody stores into a mirror any more), each rank calls gato_cluster_rewind (zeroes its
// mirror and its level-1 slots, counters back to 0), all pass a second barrier, and the solves go on.  dist.ClusterPCG does this.
extern "C" int gato_cluster_launches_left(gato_solver *s, int max_iters, long long *left)
{
    if (!s->cl.on || !left) { set_error("cluster_launches_left: gato_cluster_connect first"); return GATO_EINVAL; }
    if (max_iters < 0) { set_error("cluster_launches_left: max_iters must be >= 0 (got %d)", max_iters); return GATO_EINVAL; }
    const unsigned long long need = max_iters > 0x3FFFFFF0 ? 0x80000000ull : 2ull * (unsigned)max_iters + 8ull;
    const unsigned long long top = 0xFFFFFFFFull - need - 8ull;
    const unsigned long long used = s->cl.xepoch;          // (the level-1 counter of a rank renews itself in stream order: gato_cluster_pcg)
    *left = used > top ? 0 : (long long)((top - used) / need) + 1;
    return GATO_OK;
}

extern "C" int gato_cluster_rewind(gato_solver *s)
{
    if (!s->cl.local) { set_error("cluster_rewind: gato_cluster_create first"); return GATO_EINVAL; }
    GATO_HIP_CHECK(hipSetDevice(s->device));
    GATO_HIP_CHECK(hipDeviceSynchronize());
    GATO_HIP_CHECK(hipMemset(s->cl.local, 0, s->cl.bytes));
    GATO_HIP_CHECK(hipMemset(s->slots, 0, s->slots_bytes));
    GATO_HIP_CHECK(hipDeviceSynchronize());
    s->pcg_epoch = 0;
    s->cl.xepoch = 0;
    return GATO_OK;
}

extern "C" int gato_cluster_destroy(gato_solver *s)
{
    if (!s) return GATO_OK;
    for (int r = 0; r < GATO_MAX_RANKS; ++r)
        if (s->cl.opened[r] && s->cl.peer[r]) (void)hipIpcCloseMemHandle(s->cl.peer[r]);
    if (s->cl.local) mirror_give(s->cl.local, s->device, s->cl.mem_kind, s->cl.alloc_bytes);      // kept for the next cluster
    memset(&s->cl, 0, sizeof(s->cl));
    return GATO_OK;
}

// Geometry a cluster launch of this rank would use (0 workgroups: the rank's knots do not fit a persistent launch).
static int cluster_plan(gato_solver *s, int *groups, int *threads, int *kpw)
{
    // geometry over this rank's knots; the one-workgroup special kernels have no cross-GPU level
    const int np = s->no_pair, nl = s->no_single_lds;
    s->no_pair = 1; s->no_single_lds = 1;
    const int fits = plan_resident_k(s, s->cl.k1 - s->cl.k0, groups, threads, kpw);
    s->no_pair = np; s->no_single_lds = nl;
    return fits;
}

extern "C" int gato_cluster_fits(gato_solver *s, int *groups, int *threads)
{
    if (!s->cl.local) { set_error("cluster_fits: gato_cluster_create first"); return GATO_EINVAL; }
    int g = 0, t = 0, k = 0;
    if (!cluster_plan(s, &g, &t, &k)) g = t = 0;
    if (groups) *groups = g;
    if (threads) *threads = t;
    return GATO_OK;
}

// Single-reduction recurrence in a cluster (option pcg_variant = 1): EVERY rank must be able to run it - its knots fit one launch
// of pcg_cg1_kernel<..., MR> and every workgroup of the cluster owns at least two knots (the exchange carries the first / last two
// blocks of w) - or every rank takes the default recurrence: each rank derives every rank's geometry from the same rule (same
// device type and options on all ranks, as for the flat exchange).  1 = variant 1 runs; geometry of THIS rank, and the flat
// exchange's numbering (total <= 256 workgroups) if it applies.
static int cluster_plan_cg1(gato_solver *s, int *groups, int *threads, int *kpw, int *flat_total, int *flat_base)
{
    if (s->pcg_variant != 1 || s->true_warm_start || s->pcg_mode == GATO_PCG_STREAMING) return 0;
    int total = 0, base = 0;
    for (int r = 0; r < s->cl.nranks; ++r) {
        int k0 = 0, k1 = 0, g = 0, t = 0, kp = 0;
        gato_cluster_knot_range(s->d.K, r, s->cl.nranks, &k0, &k1);
        const int Kr = k1 - k0;
        if (!plan_cg1_k(s, Kr, &g, &t, &kp)) return 0;
        if (s->cl.nranks > 1 && (kp < 2 || Kr - (g - 1) * kp < 2)) return 0;
        if (r < s->cl.rank) base += g;
        if (r == s->cl.rank) { *groups = g; *threads = t; *kpw = kp; }
        total += g;
    }
    *flat_total = total; *flat_base = base;
    return 1;
}

// One rank's part of a PCG solve sharded over the cluster: d_S / d_Pinv / d_gamma / d_lambda are FULL-system arrays
// (block row 0 first) of which this rank reads / writes the rows of its range only (variant 1: Pinv and gamma also on the
// neighbouring knots, see cluster_plan_cg1 / gato_cluster_linsys).  Every rank must call it with the same exit_tol and
// max_iters; the launches synchronise with each other on the device (bounded spins), never on the host.  d_iters: as gato_pcg
// (-1 = a hand-off timed out).  d_lambda holds this rank's slice on return - and, on every rank but the last, the right
// neighbour's first block at row k_end (cluster_lambda_ghost: what the dz of this rank's last knot needs).
extern "C" int gato_cluster_pcg(gato_solver *s, const void *d_S, const void *d_Pinv, const void *d_gamma, void *d_lambda,
                                double exit_tol, int max_iters, int *d_iters, void *stream)
{
    if (!s->cl.on) { set_error("cluster_pcg: gato_cluster_connect first"); return GATO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    // a captured launch would be REPLAYED with the epochs of the capture: stale granules would pass the polls (see pcg_one)
    if (stream_is_capturing(st)) {
        set_error("cluster_pcg: a cluster launch cannot be captured into a graph (its hand-off epochs are launch arguments)");
        return GATO_EINVAL;
    }
    if (max_iters < 0) { set_error("cluster_pcg: max_iters must be >= 0 (got %d)", max_iters); return GATO_EINVAL; }
    int groups = 0, threads = 0, kpw = 0, cg1_total = 0, cg1_base = 0;
    const bool cg1 = cluster_plan_cg1(s, &groups, &threads, &kpw, &cg1_total, &cg1_base) != 0;
    const int fits = cg1 ? 1 : cluster_plan(s, &groups, &threads, &kpw);
    if (!fits) {
        set_error("cluster_pcg: %d knots per rank do not fit a persistent launch on %d CUs", s->cl.k1 - s->cl.k0, s->num_cus);
        return GATO_EINVAL;
    }
    const unsigned need = max_iters > 0x3FFFFFF0 ? 0x80000000u : 2u * (unsigned)max_iters + 8u;
    if (s->cl.xepoch > 0xFFFFFFFFu - need - 8u) {
        set_error("cluster_pcg: epoch space used up - renew it on every rank (gato_cluster_launches_left / gato_cluster_rewind between two barriers)");
        return GATO_EINVAL;
    }
    if (s->pcg_epoch > 0xFFFFFFFFu - need - 8u) {
        GATO_HIP_CHECK(hipMemsetAsync(s->slots, 0, s->slots_bytes, st));
        s->pcg_epoch = 0;
    }
    PcgLaunch a;
    memset(&a, 0, sizeof(a));
    a.S_bd = d_S; a.P_bd = d_Pinv; a.gamma = d_gamma; a.lambda = d_lambda;
    a.lambda0 = s->true_warm_start ? d_lambda : nullptr;
    a.K = s->d.K; a.max_iters = max_iters; a.exit_tol = exit_tol;
    a.batch = 1; a.semi = cg1 ? 0 : s->plan_semi; a.dpp_rows = cg1 ? 0 : s->plan_dpp;
    a.wave_pub = s->wave_pub;
    a.knots_per_wg = kpw; a.groups = groups; a.threads = threads;
    a.slots = s->slots; a.iters = d_iters ? d_iters : s->iters; a.status = s->status;
    a.epoch0 = s->pcg_epoch; s->pcg_epoch += need;
    a.xepoch0 = s->cl.xepoch; s->cl.xepoch += need;
    a.lam_off = s->cl.lam_off;
    a.lam_tag = a.xepoch0 + need;                   // > every epoch of this launch, < every epoch of the next: unique, never 0
    if (++s->pcg_launch_id <= 0) s->pcg_launch_id = 1;
    a.launch_id = s->pcg_launch_id;
    a.final_eta = s->final_eta;
    a.eta_hist = (s->record_eta && max_iters <= GATO_ETA_HIST_MAX) ? s->eta_hist : nullptr;
    a.timeout_ticks = (unsigned long long)s->timeout_ms * 100000ull;
    a.k_begin = s->cl.k0; a.k_end = s->cl.k1; a.rank = s->cl.rank; a.nranks = s->cl.nranks;
    a.xslots = s->cl.local;
    a.xpeer = s->cl_tab;
    // flat exchange when the whole cluster has at most 256 workgroups and every rank runs the plain resident variant:
    // every rank derives every rank's geometry from the same rule (same device type, same options on all ranks)
    a.flat = 0;
    if (cg1) {
        if (s->cluster_flat != 0 && s->cl.nranks > 1 && cg1_total <= 256) {
            a.flat = 1; a.flat_groups = cg1_total; a.flat_base = cg1_base; a.flat_off = s->cl.flat_off;
        }
    } else if (s->cluster_flat != 0 && s->cl.nranks > 1 && !a.semi) {
        int total = 0, base = 0, ok = 1;
        const int k0s = s->cl.k0, k1s = s->cl.k1;
        for (int r = 0; r < s->cl.nranks && ok; ++r) {
            int g = 0, t = 0, kp = 0;
            gato_cluster_knot_range(s->d.K, r, s->cl.nranks, &s->cl.k0, &s->cl.k1);
            if (!cluster_plan(s, &g, &t, &kp) || s->plan_semi) ok = 0;
            if (r < s->cl.rank) base += g;
            total += g;
        }
        s->cl.k0 = k0s; s->cl.k1 = k1s;
        { int g = 0, t = 0, kp = 0; cluster_plan(s, &g, &t, &kp); }       // restore this rank's plan state (plan_semi)
        if (ok && total <= 256) { a.flat = 1; a.flat_groups = total; a.flat_base = base; a.flat_off = s->cl.flat_off; }
    }
    s->cl.last_flat = a.flat;
    // (One-XCD placement of a rank's <= 32 workgroups, as one-GPU launches get, was measured for cluster launches in round 5 and
    //  not kept: a cluster of one rank at 14/7/512 f32 3.47 -> 3.35 us per iteration, fp64 5.2 -> 5.8; with 8 / 4 ranks sharing a chip
    //  the flat exchange (5.74 / 4.88) beats two levels with packed level 1 (6.67 / 5.43).  What separates these launches from the
    //  2.2 us of the plain launch at the same knot count is the lean hand-off with workgroup-scope stores, which the MR kernels'
    //  level 1 does not have - DESIGN_LOG.md R5.7.)
    a.ev_start = s->time_pcg ? s->ev_pcg0 : nullptr;
    a.ev_stop = s->time_pcg ? s->ev_pcg1 : nullptr;
    s->last_groups = groups; s->last_threads = threads; s->last_mode = GATO_PCG_RESIDENT; s->last_variant = cg1 ? 1 : 0;
    s->last_semi = a.semi; s->last_stream = st;
    // the launches of a cluster wait for EACH OTHER: they are never queued behind one another (ranks sharing a device
    // exist in tests only), but they count for the other launches of this process
    int rc;
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    if ((rc = cg1 ? s->ops->pcg_cg1(a, st) : a.semi == 3 ? s->ops->pcg_dma(a, st) : s->ops->pcg_resident(a, st))) return rc;
    return gate_after(s->device, groups, st);
}

// One rank's part of a WHOLE solve sharded over the cluster (gato_linsys, gpu_library.cu:25-83, on this rank's knot range): the
// stage kernels on the knots its PCG shard reads (S / Pinv rows k0..k1-1 complete: S[k].right comes from the Schur step of knot
// k+1 and the stair blocks need theta^-1 of both neighbours, gamma on k0-1..k1 - hence CSR scatter + inversions on [k0-2-h, k1+1+h),
// Schur steps on [k0-1-h, k1+1+h), stair on [k0-h, k1+h); h = 1 for the single-reduction recurrence, whose edge workgroups also
// multiply with the neighbouring knots' Pinv rows), the rank's cluster launch, and dz on [k0, k1) - lambda_{k1} arrives inside
// the launch (cluster_lambda_ghost), so NOTHING crosses the host or a collective between assembly, PCG and dz: one call, a handful
// of enqueues.  CSR inputs, d_g, d_c: the full system (replicated); d_lambda / d_dz: full-length arrays of which the rank writes
// its rows (lambda: + row k1).  Work buffers: the solver's own.
extern "C" int gato_cluster_linsys(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val, const int *d_C_row,
                                   const int *d_C_col, const void *d_C_val, const void *d_g, const void *d_c, double exit_tol,
                                   int max_iters, double rho, void *d_lambda, void *d_dz, int *d_iters, void *stream)
{
    if (!s->cl.on) { set_error("cluster_linsys: gato_cluster_connect first"); return GATO_EINVAL; }
    if (s->precon_mode != GATO_PRECON_STAIR) { set_error("cluster_linsys: the stair preconditioner only"); return GATO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int K = s->d.K, k0 = s->cl.k0, k1 = s->cl.k1;
    const int h = (s->pcg_variant == 1 && !s->true_warm_start) ? 1 : 0;           // wide enough for either recurrence the launch may take
    auto clip = [&](int k) { return k < 0 ? 0 : (k > K ? K : k); };
    auto range = [&](int lo, int hi) { s->d.k_lo = clip(lo); s->d.k_hi = clip(hi); if (s->d.k_hi == 0) s->d.k_lo = 0; };
    int rc;
    const bool ts = s->time_stages != 0;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[0], st));
    // (the fused one-launch assembly of small one-GPU solves was tried here for small shards - 512 knots, what K = 4096 over 8 GPUs
    //  gives - and measured no better: 31 against 28 us outside the loop, tools/cluster_step_time.py; the stage kernels stay)
    range(k0 - 2 - h, k1 + 1 + h);
    rc = s->ops->convert(s->d, d_G_row, d_G_col, d_G_val, d_C_row, d_C_col, d_C_val, rho, s->G_dense, s->C_dense, nullptr, st);
    // (form_schur inverts the Q_k, R_k of its knot range first and then runs the Schur steps on the same range: the step of the
    //  range's first knot reads an inverse outside the range and only writes rows k0-2-h of S / Pinv, which nobody reads)
    if (!rc) rc = s->ops->form_schur(s->d, s->G_dense, s->C_dense, d_g, d_c, s->Sbd, s->Pbd, s->gamma, s->Ginv, false, st);
    if (!rc) { range(k0 - h, k1 + h); rc = s->ops->form_ss(s->d, s->Sbd, s->Pbd, st); }
    s->d.k_lo = s->d.k_hi = 0;
    if (rc) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[1], st));
    if ((rc = gato_cluster_pcg(s, s->Sbd, s->Pbd, s->gamma, d_lambda, exit_tol, max_iters, d_iters, stream))) return rc;
    if (ts) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[2], st));
    s->d.k_lo = k0; s->d.k_hi = k1;
    rc = s->ops->compute_dz(s->d, s->Ginv, s->C_dense, d_g, d_lambda, d_dz, st);
    s->d.k_lo = s->d.k_hi = 0;
    if (ts && !rc) GATO_HIP_CHECK(hipEventRecord(s->ev_stage[3], st));
    return rc;
}
