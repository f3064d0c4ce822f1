// capi.hip — the C ABI of libingvio_hip.so (include/ingvio_hip.h): context, staging, launches.
// Host code only; kernels live in kernels_{cov,msckf,ekf}.hip.  gfx950 only, no CPU fallback:
// every entry point fails with INGVIO_E_HIP if the HIP runtime has no device.
#include "../../include/ingvio_hip.h"
#include "dev_common.h"
#include "launch_ekf.h"
#include "launch_msckf.h"
#include "launch_factored.h"
#include "launch_tri.h"
#include "launch_tracks.h"
#include "launch_lm.h"
#include "launch_qr.h"
#include "launch_chol.h"
#include "launch_gnss.h"
#include "launch_lmbatch.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#define KMAX 64            // IMU steps per fused propagation call
#ifndef UPLOAD_KERNEL_MAX      // host-to-device uploads up to this size travel as a kernel reading the pinned slab (Uploader::copy)
#define UPLOAD_KERNEL_MAX (256u << 10)
#endif
#define IMU_SLAB_NB 4      // ingvio_propagate(_fused) for up to this many filters: inputs travel as one copy (ingvio_ctx::d_imu)
#define CHI2_CAP 1024

namespace {

enum ProfId { PF_RESTORE, PF_PROPAGATE, PF_AUGMENT, PF_GATE, PF_FOLD, PF_MERGE, PF_EKF_CORE, PF_DOWNDATE, PF_MARG,
              PF_GATE2, PF_GRAM, PF_INFO, PF_APPLY, PF_ROWGATE, PF_LM_BUILD, PF_LM_GEMM, PF_LM_CHOL, PF_POSTCOLS, PF_COUNT };
const char* kProfNames[PF_COUNT] = { "restore", "k_propagate", "k_augment", "k_msckf_gate", "k_msckf_fold",
                                     "k_msckf_merge", "k_ekf_core", "k_downdate", "k_marginalize",
                                     "gate", "gram", "solve", "apply", "k_rows_gate",      // stage slots: bench.py names the kernel that ran
                                     "k_lm_build", "k_lm_gemm", "k_lm_chol", "k_post_cols" };
struct ProfRec { int id; hipEvent_t a, b; };

}  // namespace

struct ingvio_ctx {
    ingvio_ctx_desc d;
    int ldp, mld, nc_cap, G, rstride, hstride, cstride, ystride;
    int cls;                       // CMAX class of the MSCKF kernels
    hipStream_t st;
    bool own_stream;
    std::string err;
    // covariances
    double *Pbase, *Psnap;
    size_t pp = 0;                 // doubles between consecutive filters' covariances (ldp * ldp + pad)
    int *d_cur, *d_n, *d_n_snap;
    std::vector<int> h_n, h_cur, h_n_snap;
    bool has_snap;
    // partial restore: valid while half 0 still equals the snapshot outside the propagation strips, i.e. right after a
    // fused frame step that began with a restore; any other use of the covariance (view()) invalidates it
    int prof_only = -1;          // >= 0: only this kernel id is bracketed by events (ingvio_profile_select)
    bool strip_ok = false;
    std::vector<int> h_nclones;      // per filter: n_clones of the staged frame (picks the window class of the MSCKF kernels)
    // ingvio_gnss_sat_eval's device buffers, grown on demand and kept (psr_pos / dopp_vel call it once per Gauss-Newton iteration,
    // the aligner once per buffered epoch: per-call hipMalloc / hipFree were hundreds of implicit device syncs)
    struct { double *e = nullptr, *o = nullptr, *r = nullptr, *f = nullptr; int cap = 0; } se;
    bool phase_restore = false;      // the pending split step was started with restore_prior
    unsigned long long mut_seq = 0, strip_seq = 0;
    // propagation / structure staging
    double *d_Phi, *d_G, *d_dt, *d_R, *d_blk;
    int *d_gnss, *d_idx;
    int* d_zero_idx = nullptr;      // [B] zeros: "marginalise nothing at 0" = an out-of-place write-back without compaction (frame with landmarks)
    // frame staging
    char* d_result_slab = nullptr;                     // d_dx | d_gamma | d_used | d_m | d_status are views into it; h_result: pinned mirror
    char* h_result = nullptr;
    size_t result_bytes = 0, ro_gam = 0, ro_used = 0, ro_m = 0, ro_st = 0, ro_tok = 0, ro_tpf = 0;      // ro_tok / ro_tpf: triangulation flags and points (ingvio_msckf_update_tri)
    char* d_frame_slab[2] = { nullptr, nullptr };      // the frame arrays below are views into these (frame_slab_carve); [1]: the async set
    int *d_clone_idx, *d_nclones, *d_nfeat, *d_anchor, *d_dof;
    double *d_clone_R, *d_clone_p, *d_pf, *d_uv, *d_chi2;
    unsigned long long* d_mask;
    // msckf / ekf workspaces
    double *d_gamma, *d_Rpart, *d_H, *d_res, *d_noise, *d_noise1, *d_Y, *d_Yc, *d_dx, *d_rec, *d_hnew;
    int method;                    // 0 dense TSQR path, 1 factored (information-form) path
    int *d_accept, *d_used, *d_chunk_used, *d_colmap, *d_m, *d_nc, *d_status, *d_pcbase;
    double *d_big_sg, *d_big_wk;        // large-window workspaces (c_max > 16)
    double* d_Asum = nullptr;           // [B][rstride]: the chunk partials of a filter summed (k_chunk_sum), windows up to 16 clones with G > 1
    int* d_used_sum = nullptr;          // [B]
    // what d_chi2 and d_noise hold (single-filter latency: an update re-sent the same gate table and the same noise variance with every
    // call, two host-to-device copies of ~7 us each in front of the kernels); invalidated by every other writer of those buffers
    struct { std::vector<double> chi2; bool chi2_ok = false; double var = 0.0; int b0 = -1, nb = 0; bool noise_ok = false; } upc;
    int apply_flipped = 0;              // set by run_msckf_factored: the fused marginalisation's flip (cur, n) was done by the write-back kernel itself (k_info_apply)
    bool fork_recorded = false;         // large windows: ev_fork already recorded on the main stream by the caller of run_msckf_factored
    unsigned long long* d_tri_mask = nullptr;      // [B][f_max], allocated on first use: triangulation masks of ingvio_msckf_update_tri
    char* d_imu = nullptr;              // [IMU_SLAB_NB filters] Phi | G | dt | gnss_idx of ingvio_propagate(_fused) in one piece (few filters per call)
    double* d_Tflat = nullptr;          // [min(B, APPLY_FLAT_NB)][ldp * 100]: T of the few-filter apply (k_apply_T_flat), windows up to 16 clones

    int* d_tri_ok;                      // [B][f_max] triangulation flags
    // ingvio_qr_compress, general path: device buffers and the captured launch sequence (~300 kernels) of the last shape
    struct QrCache { int m = 0, n = 0, ldh = 0, chol = 0; double *dA = nullptr, *db = nullptr, *ws = nullptr, *dT = nullptr; hipGraphExec_t exec = nullptr; } qr;
    int qr_method = 0;                  // ingvio_set_qr_method: 0 auto, 1 Householder, 2 Cholesky-QR
    double* d_noiseB = nullptr;         // ingvio_ekf_update_batch: [B][mld] scalar / diagonal noise per filter
    // ingvio_gnss_stage / _run / _fetch: the staged candidate rows of the batch (pristine: every run gates and compacts them
    // into the generic update's working buffers d_H / d_res / d_noiseB)
    struct GnssStage {
        double *H = nullptr, *res = nullptr, *noise = nullptr, *gamma = nullptr, *chi2 = nullptr;
        int *m = nullptr, *nc = nullptr, *colmap = nullptr, *keep = nullptr;
        double *feph = nullptr, *fobs = nullptr, *frcv = nullptr, *front = nullptr;      // ingvio_gnss_front_stage inputs / per-satellite results
        int ncw = 0, m_cap = 0, n_vars_hi = 0, chi2_len = 0, gate_rows = 0, strong = 0;
        double thr1 = 0.0;
        std::vector<int> hi;            // per filter: highest state index named by the staged var_order (+1)
        bool staged = false;
        // in-frame stage (ingvio_gnss_opts::in_frame): applied by ingvio_frame_run.  W [B][ldp * 16]: the var_order columns of the MSCKF
        // posterior (k_post_cols); Yf [B][ldp * 16]: the update's Cholesky-form gain, folded into k_info_apply; its own working rows
        // count / column count / dx (the frame's m, nc and dx slots belong to the MSCKF update)
        bool in_frame = false, fused_last = false;
        bool results = false;           // an update has run on the current stage: ingvio_gnss_fetch has something to return.  A non-restoring
                                        // ingvio_frame_run CONSUMES an in-frame stage (staged = false, as it does a landmark stage: the rows
                                        // belong to one frame); its results stay fetchable until the next stage
        int nc_max = 0;
        double *W = nullptr, *Yf = nullptr, *dxf = nullptr;
        int *mf = nullptr, *ncf = nullptr;
    } gn;
    // dense-H update workspace (kernels_lmbatch.hip + kernels_chol.hip): the batched landmark update and generic updates whose S
    // does not fit in LDS.  Rows live in Hd [m_cap][n_ld] per filter, the sweep in X / Y [ldx][m_cap].
    struct DenseWs {
        double *Hd = nullptr, *X = nullptr, *Y = nullptr, *Tb = nullptr, *U = nullptr, *noise = nullptr, *noiseB = nullptr;      // U: factor tiles of the register-resident solve (m_cap <= 256); noise: one filter's R (ingvio_ekf_update); noiseB [B][m_cap]: scalar / diagonal R per filter (batch)
        int *m = nullptr, *cidx = nullptr, *rowmap = nullptr;      // rowmap [B][m_cap]: the landmark front's accepted rows (k_lm_front)
        int m_cap = 0, n_ld = 0, n32 = 0, ldx = 0;
        size_t hstride = 0, xstride = 0, tstride = 0, ustride = 0;
    } dw;
    struct LmStage {
        double *pose = nullptr, *pf = nullptr, *uv = nullptr, *gamma = nullptr, *dx = nullptr;      // dx: its own [B][ldp] (the frame's MSCKF dx stays in d_dx)
        int *idx = nullptr, *n_lm = nullptr, *lm_idx = nullptr, *anchor_idx = nullptr, *tracked = nullptr, *accept = nullptr;
        LmOpts op;
        int l_hi = 0, in_frame = 0;
        bool alloc = false, staged = false;
        std::vector<int> hi;            // per filter: one past the highest state index the staged rows name (re-checked against the live n at run time)
    } lm;
    char* d_multi = nullptr;            // ingvio_chi2_gamma_multi: packed blocks (grown on demand)
    size_t multi_cap = 0;
    // staged frame state
    int st_k, st_stereo, st_enable_gnss, st_fmax_used;
    double st_sigma[4], st_scb, st_srw;
    MsckfOpts st_op;
    std::vector<int> st_gnss;      // [B][5] clock-state indices of the staged steps (a stage with the same ones keeps the strip restore valid)
    std::vector<int> st_marg;      // per filter marg idx
    std::vector<int> st_cidx_hi;   // per filter: highest staged clone idx (checked against the live state by frame_run)
    bool staged;
    double* d_xchg = nullptr;      // [B][rstride + 8]: [A | b | n_accepted] of the split step's exchange (ingvio_info_reduce / _commit)
    bool phase_pending = false;    // ingvio_frame_run_phase(.., 1) ran and (.., 2) has not yet: the filters are half-stepped (n + 6, no update)
    // second set of device input buffers + copy stream (ingvio_frame_stage_async): the inputs of frame i+1 travel over
    // PCIe while frame i computes; the sets swap roles at every asynchronous stage
    struct InputSet {
        double *Phi = nullptr, *G = nullptr, *dt = nullptr, *R = nullptr, *clone_R = nullptr, *clone_p = nullptr, *pf = nullptr, *uv = nullptr,
               *chi2 = nullptr, *noise = nullptr;
        int *gnss = nullptr, *idx = nullptr, *clone_idx = nullptr, *nclones = nullptr, *nfeat = nullptr, *anchor = nullptr, *dof = nullptr;
        unsigned long long* mask = nullptr;
    } alt;
    bool alt_ready = false;
    // large windows: the measurement-independent front of the Kalman solve runs here, under the gate and the Gram kernel (run_msckf_factored)
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_ref = nullptr;
    hipEvent_t ev_fetch = nullptr;      // ingvio_frame_fetch_begin / _end
    int fetch_b0 = 0, fetch_nb = 0;
    hipStream_t st_copy = nullptr;
    hipEvent_t ev_copy = nullptr, ev_free[2] = { nullptr, nullptr };      // inputs landed / set no longer read by the compute stream
    bool copy_pending = false, free_valid[2] = { false, false };
    int set_id = 0;                                                        // which physical set the d_* pointers name
    // pinned host staging ring: inputs are packed straight into page-locked memory and copied asynchronously; a slab is
    // reused only after the copies issued from it have completed (its event), so no call has to synchronise the stream
    struct PinSlab { char* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; };
    PinSlab pin[4];
    int pin_next = 0;
    // Split frame step (round 6): ingvio_frame_run deals the batch to `parts` slices, each with its own stream and
    // its own chain restore -> propagate -> gate -> Gram -> solve -> apply.  The gates are chained by events (slice p's gate starts
    // when slice p - 1's has ended; slice 0's when the LAST slice's of the previous step has), so the slices run half a step apart:
    // the HBM-bound kernels of one slice (apply, propagate, restore) sit under the FP64-bound ones of the other (gate, Gram).
    // The slices' streams are NOT joined at the end of the call - consecutive steps pipeline; every other entry point that touches
    // the context joins them first (join_parts, called through enter()).
    struct PartStream { hipStream_t st = nullptr; hipEvent_t ev_gate = nullptr, ev_apply = nullptr, ev_done = nullptr; };
    PartStream part[4];
    int parts_alloc = 0;
    int parts_req = -1;                 // ingvio_set_frame_parts: -1 automatic, 1 off, 2..4 forced
    bool split_pending = false;         // slices of the last split step may still be running on their own streams
    hipEvent_t tok_last = nullptr;      // end of the throughput segment issued last (nullptr: none yet / chain broken)
    hipEvent_t ev_split_fork = nullptr;
    hipStream_t run_st = nullptr;       // the stream run_msckf_factored and ProfScope issue on (c->st except inside a split step)
    hipEvent_t tok_wait = nullptr, tok_rec = nullptr;      // run_msckf_factored: wait before / record after its throughput segment (phase 1: gate + Gram, phase 2: apply)
    // device-resident track store (ingvio_tracks_create / ingvio_frame_stage_tracks, kernels_tracks.hip)
    struct Tracks { double *uv = nullptr, *pf = nullptr; unsigned long long* mask = nullptr; int t_max = 0; char* stage[2] = { nullptr, nullptr }; size_t stage_cap = 0; } trk;
    // profiling
    bool prof;
    std::vector<ProfRec> recs;
    double prof_ms[PF_COUNT];
    int prof_calls[PF_COUNT];
};

namespace {

#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                    \
            return INGVIO_E_HIP;                                                                \
        }                                                                                       \
    } while (0)

template <class T>
int dalloc(ingvio_ctx* c, T** p, size_t count)
{
    HIPCHK(c, hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
    HIPCHK(c, hipMemsetAsync(*p, 0, sizeof(T) * (count ? count : 1), c->st));
    return 0;
}

CovView view(ingvio_ctx* c)
{
    CovView v;
    v.Pbase = c->Pbase; v.cur = c->d_cur; v.n = c->d_n; v.ldp = c->ldp; v.B = c->d.batch; v.pstride = c->pp;
    ++c->mut_seq;
    return v;
}

FrameView fview(ingvio_ctx* c)
{
    FrameView f;
    f.clone_idx = c->d_clone_idx; f.clone_R = c->d_clone_R; f.clone_p = c->d_clone_p;
    f.n_clones = c->d_nclones; f.n_feat = c->d_nfeat; f.pf = c->d_pf; f.anchor = c->d_anchor;
    f.obs_mask = c->d_mask; f.uv = c->d_uv; f.dof = c->d_dof; f.cmax = c->d.c_max; f.fmax = c->d.f_max;
    return f;
}

struct ProfScope {
    ingvio_ctx* c; int id; hipEvent_t a, b; bool on;
    ProfScope(ingvio_ctx* c_, int id_) : c(c_), id(id_), on(c_->prof && (c_->prof_only < 0 || c_->prof_only == id_))
    {
        if (on) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, c->run_st); }
    }
    ~ProfScope()
    {
        if (on) { hipEventRecord(b, c->run_st); c->recs.push_back({ id, a, b }); }
    }
};

int check_range(ingvio_ctx* c, int b0, int nb)
{
    if (!c || b0 < 0 || nb < 1 || b0 + nb > c->d.batch) return INGVIO_E_ARG;
    return 0;
}

// Entry points that change (or snapshot) the covariance refuse to run between the two halves of a split frame step
// (ingvio_frame_run_phase 1 -> 2): the filters are half-stepped there (cloned, not yet updated / marginalised).
static int phase_busy(ingvio_ctx* c)
{
    if (c && c->phase_pending) {
        c->err = "a split frame step is pending: finish it with ingvio_frame_run_phase(ctx, 0, 2) first";
        return 1;
    }
    return 0;
}

int up(ingvio_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return 0;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->st));
    return 0;
}

int down_sync(ingvio_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

int last_launch(ingvio_ctx* c)
{
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Split frame step: the compute stream waits for the slices' streams (a device-side wait, the host does not block).  Called by every
// entry point other than ingvio_frame_run itself before it touches the context.
int join_parts(ingvio_ctx* c)
{
    if (!c->split_pending) return 0;
    for (int p = 0; p < c->parts_alloc; ++p) HIPCHK(c, hipStreamWaitEvent(c->st, c->part[p].ev_done, 0));
    c->split_pending = false;
    if (c->alt_ready) {                            // the input set the slices read may be refilled once they are done with it
        HIPCHK(c, hipEventRecord(c->ev_free[c->set_id], c->st));
        c->free_valid[c->set_id] = true;
    }
    return 0;
}
#define ENTER(c) do { if ((c) && (c)->split_pending && join_parts(c)) return INGVIO_E_HIP; } while (0)

int parts_prepare(ingvio_ctx* c, int P)
{
    while (c->parts_alloc < P) {
        auto& q = c->part[c->parts_alloc];
        HIPCHK(c, hipStreamCreateWithFlags(&q.st, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&q.ev_gate, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&q.ev_apply, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&q.ev_done, hipEventDisableTiming));
        ++c->parts_alloc;
    }
    if (!c->ev_split_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_split_fork, hipEventDisableTiming));
    return 0;
}

// INGVIO_RESTORE=pass: ingvio_frame_run(restore_prior) keeps the separate restore pass in front of the propagation (comparison runs)
static bool no_snap_propagate()
{
    static const bool v = [] { const char* e = getenv("INGVIO_RESTORE"); return e && !strcmp(e, "pass"); }();
    return v;
}

// INGVIO_FLIP=launch: the fused marginalisation's flip of the halves stays a launch of its own (k_post_marg; comparison runs)
static bool no_apply_flip()
{
    static const bool v = [] { const char* e = getenv("INGVIO_FLIP"); return e && !strcmp(e, "launch"); }();
    return v;
}

// ---- pinned staging ------------------------------------------------------------------------------------------------
struct Uploader {
    ingvio_ctx* c;
    hipStream_t stream = nullptr;      // nullptr: the context's compute stream
    ingvio_ctx::PinSlab* slab = nullptr;
    size_t off = 0;
    int rc = 0;
    int begin(size_t bytes)
    {
        slab = &c->pin[c->pin_next];
        c->pin_next = (c->pin_next + 1) & 3;
        if (slab->busy) { HIPCHK(c, hipEventSynchronize(slab->ev)); slab->busy = false; }
        if (slab->cap < bytes) {
            if (slab->p) hipHostFree(slab->p);
            slab->p = nullptr; slab->cap = 0;
            const size_t cap = (bytes + (1u << 20)) & ~((size_t)(1u << 20) - 1);
            HIPCHK(c, hipHostMalloc((void**)&slab->p, cap, hipHostMallocDefault));
            memset(slab->p, 0, cap);
            slab->cap = cap;
        }
        if (!slab->ev) HIPCHK(c, hipEventCreateWithFlags(&slab->ev, hipEventDisableTiming));
        off = 0;
        return 0;
    }
    template <class T>
    T* take(size_t count)
    {
        off = (off + 63) & ~(size_t)63;
        T* p = reinterpret_cast<T*>(slab->p + off);
        off += sizeof(T) * count;
        return p;
    }
    template <class T>
    void copy(T* dst, const T* src, size_t count)
    {
        const size_t bytes = sizeof(T) * count;
        if (!count) return;
        // small uploads on the compute stream: a kernel that reads the pinned slab (k_upload_words) - no copy-engine hand-over
        if (!stream && bytes <= UPLOAD_KERNEL_MAX && (bytes & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0) {
            launch_upload_words(dst, src, bytes, c->st);
            if (hipGetLastError() != hipSuccess) rc = INGVIO_E_HIP;
            return;
        }
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream ? stream : c->st) != hipSuccess) rc = INGVIO_E_HIP;
    }
    int end()
    {
        if (rc) { c->err = "upload from the pinned staging slab failed (hipMemcpyAsync / k_upload_words launch)"; return rc; }
        HIPCHK(c, hipEventRecord(slab->ev, stream ? stream : c->st));
        slab->busy = true;
        return 0;
    }
};
static size_t pad64(size_t b) { return (b + 63) & ~(size_t)63; }

// Small host inputs of the covariance operations that return nothing from the device (propagate, clone, marginalise, append):
// copied into the pinned ring first, so the call neither reads caller memory after it returns nor has to wait for the stream -
// a stream synchronisation per call was 15-30 us, four of them made up most of a single filter's 0.09 ms frame (round 4).  A
// failed launch is reported by the call that made it (hipGetLastError), a device fault by the next synchronising call.
struct UpItem { void* dst; const void* src; size_t bytes; };
static int stage_small(ingvio_ctx* c, const UpItem* it, int n)
{
    size_t tot = 0;
    for (int i = 0; i < n; ++i) tot += pad64(it[i].bytes);
    Uploader up{ c };
    int rc = up.begin(tot + 64);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        if (!it[i].bytes) continue;
        char* p = up.take<char>(it[i].bytes);
        memcpy(p, it[i].src, it[i].bytes);
        up.copy(reinterpret_cast<char*>(it[i].dst), p, it[i].bytes);
    }
    return up.end();
}

// fn(i) for i in [0, n): a few host threads when the batch is large enough to pay for them.  The workers are a process-wide POOL
// (round 6): spawning seven std::threads per call was ~0.2 ms of a 0.3 ms stage call once the hand-over itself had shrunk to 8 KB per
// update.  One user at a time; a second caller (another context staging from another host thread) runs its loop with freshly spawned
// threads as before instead of waiting.
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    // runs job(t) for t = 1 .. T - 1 on the workers and job(0) on the caller; false: the pool is busy (caller falls back)
    bool run(int T, const std::function<void(int)>& job)
    {
        if (getpid() != owner_) return false;                             // a fork()ed child has the object but not its threads
        std::unique_lock<std::mutex> user(user_m_, std::try_to_lock);
        if (!user.owns_lock()) return false;
        ensure(T - 1);
        if ((int)th_.size() < T - 1) return false;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job; want_ = T - 1; pending_ = T - 1; ++gen_;
        }
        cv_start_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
        return true;
    }
private:
    HostPool() : owner_(getpid()) {}
    ~HostPool()
    {
        if (getpid() != owner_) { for (auto& t : th_) t.detach(); return; }
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_start_.notify_all();
        for (auto& t : th_) t.join();
    }
    void ensure(int workers)
    {
        try {
            while ((int)th_.size() < workers) { const int id = (int)th_.size() + 1; th_.emplace_back([this, id] { loop(id); }); }
        } catch (...) {                                                   // thread limit: the caller falls back
        }
    }
    void loop(int id)
    {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_start_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (id <= want_) job = job_;
            }
            if (job) {
                (*job)(id);
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) cv_done_.notify_one();
            }
        }
    }
    const pid_t owner_;
    std::mutex user_m_, m_;
    std::condition_variable cv_start_, cv_done_;
    std::vector<std::thread> th_;
    const std::function<void(int)>* job_ = nullptr;
    int want_ = 0, pending_ = 0;
    unsigned gen_ = 0;
    bool stop_ = false;
};

template <class F>
void parallel_for(int n, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)(hw ? hw : 1);
    if (T > 8) T = 8;
    if (T > n / 16) T = n / 16;
    if (T <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    const std::function<void(int)> stripe = [&](int t) { for (int i = t; i < n; i += T) fn(i); };
    if (HostPool::get().run(T, stripe)) return;
    std::vector<std::thread> th;
    int started = 1;                                                     // stripe 0 is this thread's
    try {
        for (int t = 1; t < T; ++t) { th.emplace_back([=]() { for (int i = t; i < n; i += T) fn(i); }); ++started; }
    } catch (...) {                                                      // thread limit reached: the stripes not started run here
    }
    for (int i = 0; i < n; i += T) fn(i);
    for (int t = started; t < T; ++t) for (int i = t; i < n; i += T) fn(i);
    for (auto& x : th) x.join();
}

// The ten device arrays of a staged frame set live in ONE allocation, laid out exactly as pack_frames lays them out in the pinned
// slab (same order, every array 64-byte aligned): a stage of the WHOLE batch is then one host-to-device copy instead of ten (each
// costs ~3.5 us of API time and a ~4.7 us copy kernel on the stream - 80 us of a single filter's 0.3 ms update, round 4).
struct FrameSlabPtrs { int *clone_idx, *nclones, *nfeat, *anchor, *dof; double *clone_R, *clone_p, *pf, *uv; unsigned long long* mask; };
size_t frame_slab_carve(const ingvio_ctx* c, char* base, FrameSlabPtrs* out)
{
    const size_t B = c->d.batch, cm = c->d.c_max, fm = c->d.f_max;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 63) & ~(size_t)63; char* p = base ? base + off : nullptr; off += bytes; return p; };
    FrameSlabPtrs q;
    q.clone_idx = (int*)take(4 * B * cm); q.nclones = (int*)take(4 * B); q.nfeat = (int*)take(4 * B);
    q.anchor = (int*)take(4 * B * fm); q.dof = (int*)take(4 * B * fm);
    q.clone_R = (double*)take(8 * B * cm * 9); q.clone_p = (double*)take(8 * B * cm * 3);
    // uv last: a whole-batch upload then ends with the measurements of the last filter's LAST USED feature (pack_frames) - a single
    // filter that lost 6 tracks uploads 14 KB instead of the 360 KB its f_max x c_max measurement array takes
    q.pf = (double*)take(8 * B * fm * 3); q.mask = (unsigned long long*)take(8 * B * fm);
    q.uv = (double*)take(8 * B * fm * cm * 4);
    if (out) *out = q;
    return (off + 63) & ~(size_t)63;
}
int frame_slab_alloc(ingvio_ctx* c, char** slab, FrameSlabPtrs* out)
{
    const size_t bytes = frame_slab_carve(c, nullptr, nullptr);
    if (dalloc(c, slab, bytes)) return INGVIO_E_HIP;
    frame_slab_carve(c, *slab, out);
    return 0;
}

// the compute stream must not read staged inputs before the copy stream has delivered them
int wait_inputs(ingvio_ctx* c)
{
    if (c->copy_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->st, c->ev_copy, 0));
        c->copy_pending = false;
    }
    return 0;
}

int prepare_async_set(ingvio_ctx* c)
{
    if (!c->alt_ready) {
        const int B = c->d.batch, cm = c->d.c_max, fm = c->d.f_max;
        auto& a = c->alt;
        int rc = 0;
        rc |= dalloc(c, &a.Phi, (size_t)B * KMAX * 225); rc |= dalloc(c, &a.G, (size_t)B * KMAX * 180); rc |= dalloc(c, &a.dt, (size_t)B * KMAX);
        rc |= dalloc(c, &a.R, (size_t)B * 9); rc |= dalloc(c, &a.gnss, (size_t)B * 5); rc |= dalloc(c, &a.idx, B);
        {
            FrameSlabPtrs q;
            if (frame_slab_alloc(c, &c->d_frame_slab[1], &q)) rc |= 1;
            else {
                a.clone_idx = q.clone_idx; a.nclones = q.nclones; a.nfeat = q.nfeat; a.anchor = q.anchor; a.dof = q.dof;
                a.clone_R = q.clone_R; a.clone_p = q.clone_p; a.pf = q.pf; a.uv = q.uv; a.mask = q.mask;
            }
        }
        rc |= dalloc(c, &a.chi2, CHI2_CAP); rc |= dalloc(c, &a.noise, B);
        if (rc) return INGVIO_E_HIP;
        HIPCHK(c, hipStreamSynchronize(c->st));                        // the zero fills of dalloc
        HIPCHK(c, hipStreamCreateWithFlags(&c->st_copy, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
        for (auto& e : c->ev_free) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->alt_ready = true;
    }
    return 0;
}

void swap_input_sets(ingvio_ctx* c)
{
    auto& a = c->alt;
    std::swap(c->d_Phi, a.Phi); std::swap(c->d_G, a.G); std::swap(c->d_dt, a.dt); std::swap(c->d_R, a.R);
    std::swap(c->d_gnss, a.gnss); std::swap(c->d_idx, a.idx); std::swap(c->d_clone_idx, a.clone_idx);
    std::swap(c->d_nclones, a.nclones); std::swap(c->d_nfeat, a.nfeat); std::swap(c->d_anchor, a.anchor); std::swap(c->d_dof, a.dof);
    std::swap(c->d_clone_R, a.clone_R); std::swap(c->d_clone_p, a.clone_p); std::swap(c->d_pf, a.pf); std::swap(c->d_uv, a.uv);
    std::swap(c->d_chi2, a.chi2); std::swap(c->d_mask, a.mask); std::swap(c->d_noise, a.noise);
    c->upc.chi2_ok = false; c->upc.noise_ok = false;
    c->set_id ^= 1;
}

size_t frames_bytes(const ingvio_ctx* c, int nb)
{
    const size_t cm = c->d.c_max, fm = c->d.f_max, n = nb;
    return pad64(4 * n * cm) + 2 * pad64(4 * n) + 2 * pad64(4 * n * fm) + pad64(8 * n * cm * 9) + pad64(8 * n * cm * 3) + pad64(8 * n * fm * 3) +
           pad64(8 * n * fm * cm * 4) + pad64(8 * n * fm) + 1024;
}

// packs frames [b0, b0+nb) into the slab of `up` and enqueues the copies into the SoA device staging; max F in *fmax_used
// Everything that can be wrong with the caller's frames, checked BEFORE any context state is touched.  `grow` = rows/columns
// the state gains between now and the update (6: the clone ingvio_frame_run appends; 0: ingvio_msckf_update).
int validate_frames(ingvio_ctx* c, int b0, int nb, const ingvio_msckf_frame* fr, int grow, int* fmax_used)
{
    const int cm = c->d.c_max, fm = c->d.f_max;
    int fmx = 0;
    for (int i = 0; i < nb; ++i) {
        const ingvio_msckf_frame& f = fr[i];
        if (f.n_clones < 0 || f.n_clones > cm || f.n_feat < 0 || f.n_feat > fm) return INGVIO_E_CAPACITY;
        if (f.n_clones > 0 && (!f.clone_idx || !f.clone_R || !f.clone_p)) return INGVIO_E_ARG;
        if (f.n_feat > 0 && (!f.obs_mask || !f.uv)) return INGVIO_E_ARG;
        if (f.n_feat > fmx) fmx = f.n_feat;
        // a clone must lie inside the filter's LIVE state at update time (not merely inside the buffer): the gate and gram
        // kernels would otherwise read stale covariance (checkSubOrder, StateManager.cpp:340-357)
        const int n_live = grow >= 0 ? c->h_n[b0 + i] + grow : c->d.n_max;
        const int n_lim = n_live < c->d.n_max ? n_live : c->d.n_max;
        for (int s = 0; s < f.n_clones; ++s) {
            if (f.clone_idx[s] < 0) return INGVIO_E_ARG;
            if (f.clone_idx[s] + 6 > n_lim) return f.clone_idx[s] + 6 > c->d.n_max ? INGVIO_E_ARG : INGVIO_E_NOT_IN_STATE;
        }
        if (f.anchor)
            for (int j = 0; j < f.n_feat; ++j) if (f.anchor[j] < 0 || f.anchor[j] >= f.n_clones) return INGVIO_E_ARG;
    }
    if (fmax_used) *fmax_used = fmx;
    return 0;
}

// packs already VALIDATED frames: cannot fail
int pack_frames(ingvio_ctx* c, Uploader& up, int b0, int nb, const ingvio_msckf_frame* fr, int* fmax_used)
{
    const int cm = c->d.c_max, fm = c->d.f_max;
    int fmx = 0;
    for (int i = 0; i < nb; ++i) if (fr[i].n_feat > fmx) fmx = fr[i].n_feat;
    const bool whole = b0 == 0 && nb == c->d.batch;          // the pinned image then IS the device slab's image
    up.off = (up.off + 63) & ~(size_t)63;
    const size_t slab0 = up.off;
    int* cidx = up.take<int>((size_t)nb * cm); int* ncl = up.take<int>(nb); int* nft = up.take<int>(nb);
    int* anc = up.take<int>((size_t)nb * fm); int* dof = up.take<int>((size_t)nb * fm);
    double* cR = up.take<double>((size_t)nb * cm * 9); double* cp = up.take<double>((size_t)nb * cm * 3);
    double* pf = up.take<double>((size_t)nb * fm * 3);
    unsigned long long* mk = up.take<unsigned long long>((size_t)nb * fm);
    double* uv = up.take<double>((size_t)nb * fm * cm * 4);      // same order as frame_slab_carve: uv last
    parallel_for(nb, [=](int i) {
        const ingvio_msckf_frame& f = fr[i];
        const int C = f.n_clones, F = f.n_feat;
        ncl[i] = C; nft[i] = F;
        memcpy(cidx + (size_t)i * cm, f.clone_idx, sizeof(int) * (size_t)C);
        memcpy(cR + (size_t)i * cm * 9, f.clone_R, 72 * (size_t)C);
        memcpy(cp + (size_t)i * cm * 3, f.clone_p, 24 * (size_t)C);
        if (f.pf) memcpy(pf + (size_t)i * fm * 3, f.pf, 24 * (size_t)F); else memset(pf + (size_t)i * fm * 3, 0, 24 * (size_t)F);
        if (f.anchor) memcpy(anc + (size_t)i * fm, f.anchor, sizeof(int) * (size_t)F); else memset(anc + (size_t)i * fm, 0, sizeof(int) * (size_t)F);
        if (f.dof) memcpy(dof + (size_t)i * fm, f.dof, sizeof(int) * (size_t)F); else memset(dof + (size_t)i * fm, 0, sizeof(int) * (size_t)F);
        const unsigned long long cmask = C >= 64 ? ~0ULL : ((1ULL << C) - 1ULL);
        unsigned long long* mki = mk + (size_t)i * fm;
        for (int j = 0; j < F; ++j) mki[j] = f.obs_mask[j] & cmask;
        for (int j = F; j < fm; ++j) mki[j] = 0ULL;
        double* uvi = uv + (size_t)i * fm * cm * 4;
        if (C == cm) memcpy(uvi, f.uv, 32 * (size_t)F * C);                           // same layout: one block
        else for (int j = 0; j < F; ++j) memcpy(uvi + (size_t)j * cm * 4, f.uv + (size_t)j * C * 4, 32 * (size_t)C);
    });
    for (int i = 0; i < nb; ++i) c->h_nclones[b0 + i] = fr[i].n_clones;
    if (whole) {
        // the image up to the measurements of the last filter's last feature (what lies behind is not read by any kernel)
        const size_t used = (size_t)(reinterpret_cast<char*>(uv) - (up.slab->p + slab0)) + 8 * ((size_t)(nb - 1) * fm + (size_t)fr[nb - 1].n_feat) * cm * 4;
        up.copy(reinterpret_cast<char*>(c->d_clone_idx), up.slab->p + slab0, std::max(used, (size_t)64));
        *fmax_used = fmx;
        return 0;
    }
    up.copy(c->d_clone_idx + (size_t)b0 * cm, cidx, (size_t)nb * cm);
    up.copy(c->d_nclones + b0, ncl, nb);
    up.copy(c->d_nfeat + b0, nft, nb);
    up.copy(c->d_anchor + (size_t)b0 * fm, anc, (size_t)nb * fm);
    up.copy(c->d_dof + (size_t)b0 * fm, dof, (size_t)nb * fm);
    up.copy(c->d_clone_R + (size_t)b0 * cm * 9, cR, (size_t)nb * cm * 9);
    up.copy(c->d_clone_p + (size_t)b0 * cm * 3, cp, (size_t)nb * cm * 3);
    up.copy(c->d_pf + (size_t)b0 * fm * 3, pf, (size_t)nb * fm * 3);
    up.copy(c->d_uv + (size_t)b0 * fm * cm * 4, uv, (size_t)nb * fm * cm * 4);
    up.copy(c->d_mask + (size_t)b0 * fm, mk, (size_t)nb * fm);
    *fmax_used = fmx;
    return 0;
}

// uploads frames [b0, b0+nb) into the SoA staging; returns max F in *fmax_used
int stage_frames(ingvio_ctx* c, int b0, int nb, const ingvio_msckf_frame* fr, int* fmax_used, int grow = 0)
{
    int rc = validate_frames(c, b0, nb, fr, grow, nullptr);
    if (rc) return rc;
    if (wait_inputs(c)) return INGVIO_E_HIP;
    Uploader up{ c };
    rc = up.begin(frames_bytes(c, nb));
    if (rc) return rc;
    pack_frames(c, up, b0, nb, fr, fmax_used);
    return up.end();
}

int make_opts(ingvio_ctx* c, const ingvio_msckf_opts* o, MsckfOpts* op)
{
    if (!o || !o->chi2_table || o->chi2_len < 2 || o->chi2_len > CHI2_CAP) return INGVIO_E_ARG;
    memcpy(op->R_lr, o->R_cl2cr, 72);
    memcpy(op->t_lr, o->t_cl2cr, 24);
    op->var = o->noise * o->noise;
    op->max_accept = o->max_accept;
    op->selected_variant = o->selected_variant;
    op->chi2 = c->d_chi2;
    op->chi2_len = o->chi2_len;
    if (wait_inputs(c)) return INGVIO_E_HIP;
    auto& u = c->upc;
    if (u.chi2_ok && (int)u.chi2.size() >= o->chi2_len && !memcmp(u.chi2.data(), o->chi2_table, 8 * (size_t)o->chi2_len)) return 0;      // already there
    const UpItem item = { c->d_chi2, o->chi2_table, 8 * (size_t)o->chi2_len };      // through the pinned ring: no stream synchronisation
    if (stage_small(c, &item, 1)) return INGVIO_E_HIP;
    u.chi2.assign(o->chi2_table, o->chi2_table + o->chi2_len); u.chi2_ok = true;
    return 0;
}

// K3..K11 for filters [b0, b0+nb) using the staged frames; asynchronous.
// phase 0: the whole update; 1: gate + gram only (the chunk partials [A | b] stay in d_Rpart); 2: solve + apply from the partials
#define GNSS_NCW 32        // widest var_order of a staged GNSS update (the reference's is 9 + 1 + 4 + 1 = 15 columns)

// In-frame GNSS update, between the MSCKF solve (stage 2) and its write-back (stage 3): the var_order columns of the posterior
// (k_post_cols) -> per-row gates + compaction -> S, Cholesky, gain (k_ekf_core reading those columns instead of P) -> the gain
// rides on the write-back as a rank-16 downdate (L.gY).  Working rows / counts / dx live in the stage's own buffers.
static int gnss_in_frame_launch(ingvio_ctx* c, int b0, int nb, FactoredLaunch& L)
{
    auto& g = c->gn;
    const size_t mld = c->mld, hs = mld * GNSS_NCW, ws = (size_t)c->ldp * 16;
    L.gcolmap = g.colmap + (size_t)b0 * GNSS_NCW; L.gnc = g.nc + b0; L.gcstride = GNSS_NCW;
    L.gW = g.W + (size_t)b0 * ws; L.gWstride = ws;
    { ProfScope p(c, PF_POSTCOLS); L.stage = 4; launch_factored(L, c->run_st); }
    EkfLaunch E;
    memset(&E, 0, sizeof E);
    E.cv = L.cv; E.b0 = b0; E.nb = nb; E.H = c->d_H + (size_t)b0 * hs; E.res = c->d_res + (size_t)b0 * mld;
    E.colmap = c->d_colmap + (size_t)b0 * GNSS_NCW; E.m = g.mf + b0; E.nc = g.ncf + b0;
    E.noise = c->d_noiseB + (size_t)b0 * mld; E.r_kind = INGVIO_R_DIAG; E.mld = c->mld; E.hstride = (int)hs; E.cstride = GNSS_NCW;
    E.nstride = c->mld; E.Y = g.Yf + (size_t)b0 * ws; E.ystride = (int)ws; E.dx = g.dxf;
    E.status = c->d_status; E.m_cap = g.m_cap; E.nc_cap = GNSS_NCW;
    E.W = L.gW; E.wstride = ws; E.ypad = 16; E.marg_idx = L.marg_idx; E.marg_size = L.marg_size;
    if (g.strong) { E.chi2 = g.chi2; E.chi2_len = g.chi2_len; E.gate_max_rows = 14; }      // GnssUpdate.cpp:286
    RowsGateIn in{ g.H + (size_t)b0 * hs, g.res + (size_t)b0 * mld, g.noise + (size_t)b0 * mld, g.m + b0, g.colmap + (size_t)b0 * GNSS_NCW,
                   g.nc + b0, (int)hs, GNSS_NCW };
    {
        ProfScope p(c, PF_ROWGATE);
        if (launch_rows_gate(E, in, g.thr1, g.gamma + (size_t)b0 * mld, g.keep + (size_t)b0 * mld, c->run_st)) return INGVIO_E_CAPACITY;
    }
    { ProfScope p(c, PF_EKF_CORE); launch_ekf_core(E, c->run_st); }
    L.gY = E.Y; L.gYstride = ws; L.gm = g.mf + b0;
    g.fused_last = true;
    g.results = true;
    return 0;
}

int run_msckf_factored(ingvio_ctx* c, int b0, int nb, const MsckfOpts& op, int stereo, int fmax_used,
                       const int* marg_idx = nullptr, int marg_size = 0, int phase = 0, bool gnss_fuse = false)
{
    FactoredLaunch L;
    memset(&L, 0, sizeof L);
    L.stereo = stereo; L.cv = view(c); L.fv = fview(c); L.op = op; L.b0 = b0; L.nb = nb;
    L.fmax_used = fmax_used > 0 ? fmax_used : 1;
    L.ncol_cap = 6 * c->d.c_max;
    for (int i = 0; i < nb; ++i) if (c->h_nclones[b0 + i] > L.c_used) L.c_used = c->h_nclones[b0 + i];
    L.gamma = c->d_gamma; L.accept = c->d_accept; L.used = c->d_used; L.rec = c->d_rec;
    L.Apart = c->d_Rpart + (size_t)b0 * c->G * c->rstride; L.chunk_used = c->d_chunk_used + (size_t)b0 * c->G;
    L.G = c->G; L.rstride = c->rstride; L.noise = c->d_noise + b0;
    L.Tflat = c->d_Tflat; L.tfstride = (size_t)c->ldp * 100; L.flat_nb = std::min(c->d.batch, APPLY_FLAT_NB);
    L.Asum = c->d_Asum ? c->d_Asum + (size_t)b0 * c->rstride : nullptr; L.used_sum = c->d_used_sum ? c->d_used_sum + b0 : nullptr;
#ifdef INGVIO_ALT_KERNELS      // INGVIO_FEW=off: few filters take the same kernels as a full batch (chunk partials added inside the solve, k_info_apply)
    static const bool few_off = [] { const char* e = getenv("INGVIO_FEW"); return e && !strcmp(e, "off"); }();
    if (few_off) { L.Tflat = nullptr; L.Asum = nullptr; }
#endif
    L.T = c->d_Y + (size_t)b0 * c->ystride; L.Pc = c->d_Yc + (size_t)b0 * c->ystride; L.ystride = c->ystride;
    L.dx = c->d_dx; L.m_out = c->d_m + b0; L.nc_out = c->d_nc + b0; L.status = c->d_status;
    L.big_sg = c->d_big_sg ? c->d_big_sg + (size_t)b0 * bigwin_sg_doubles(c->G) : nullptr;
    L.big_wk = c->d_big_wk ? c->d_big_wk + (size_t)b0 * bigwin_wk_doubles() : nullptr;
    // Large windows (kernels_bigwin.hip): the part of the solve that needs the prior only - gauge reference, [Pdd; I], its Cholesky
    // sweep, the Pc copy: 8 of the chain's dependent launches, 0.12 ms alone - runs on a SECOND, higher-priority stream next to the
    // gate and the Gram kernel (round 5).  Measured per 32 filters at N = 807: in line 1.167 ms per step; forked before the gate
    // 1.134 (the sweep's launches crawl - 47 instead of 13 us each - and the gate loses 25 us, but the chain ends with the Gram
    // kernel); forked after the gate, under the Gram kernel only, 1.192.  Both hosts keep the register file full (the gate: 8 waves of
    // 231 VGPRs per CU, k_feat_gram_big: one workgroup of 8 waves at 256), so a sweep workgroup only gets a slot when a gate
    // workgroup retires - which the gate's 9600 short workgroups do all the time and the Gram kernel's 256 long ones never do.
    const bool big = c->d.c_max > 16;
    L.mstride = c->ystride; L.n_cap = c->d.n_max;
    L.marg_idx = marg_idx; L.marg_size = marg_size; L.pc_base = c->d_pcbase + b0;
    // the write-back flips the halves itself where its kernel can (k_info_apply); the caller asks c->apply_flipped before launch_post_marg
    c->apply_flipped = 0;
    L.flip_cnt = c->d_pcbase + c->d.batch + b0; L.did_flip = no_apply_flip() ? nullptr : &c->apply_flipped;
    // Order of ISSUE (round 5, single-filter latency): the fork point is recorded first (by the caller already when it has something to
    // run between the fork and the gate - the triangulation of ingvio_msckf_update_tri), then the main stream's gate and Gram launches,
    // and only then the eight launches of the side stream - issued first they kept the host busy for 24 us during which the gate could
    // not start (one filter, 27 clones: gate at +53 us after the frame's upload instead of +30).
    bool forked = false;
    if (big && phase == 0 && c->st2) {
        if (!c->fork_recorded) HIPCHK(c, hipEventRecord(c->ev_fork, c->run_st));      // after everything that wrote P on the main stream
        c->fork_recorded = false;
        forked = true;
    }
    if (phase != 2) {
        if (c->tok_wait) HIPCHK(c, hipStreamWaitEvent(c->run_st, c->tok_wait, 0));      // split frame step: one throughput segment at a time
        {
            // the gate's events ride on its dispatch packet (FactoredLaunch::prof_a / prof_b) instead of bracketing it
            const bool pon = c->prof && (c->prof_only < 0 || c->prof_only == PF_GATE2);
            int used = 0;
            if (pon) { hipEventCreate(&L.prof_a); hipEventCreate(&L.prof_b); L.prof_used = &used; }
            L.stage = 0;
            const int grc = launch_factored(L, c->run_st);
            if (pon) {
                if (used) c->recs.push_back({ PF_GATE2, L.prof_a, L.prof_b });
                else { hipEventDestroy(L.prof_a); hipEventDestroy(L.prof_b); }
                L.prof_a = nullptr; L.prof_b = nullptr; L.prof_used = nullptr;
            }
            if (grc) return INGVIO_E_UNSUPPORTED;
        }
        { ProfScope p(c, PF_GRAM); L.stage = 1; launch_factored(L, c->run_st); }
        if (c->tok_rec) HIPCHK(c, hipEventRecord(c->tok_rec, c->run_st));
    }
    if (forked) {
        // the side stream in two pieces: [A; b^T] of the main stream (stage 8) needs the gauge reference the set-up kernel picks, not the
        // sweep behind it - it runs while the sweep's last panels are still on their way (round 6)
        HIPCHK(c, hipStreamWaitEvent(c->st2, c->ev_fork, 0));
        L.stage = 6; launch_factored(L, c->st2);
        HIPCHK(c, hipEventRecord(c->ev_ref, c->st2));
        L.stage = 7; launch_factored(L, c->st2);
        HIPCHK(c, hipEventRecord(c->ev_join, c->st2));
    }
    if (phase == 1) return last_launch(c);
    if (forked) {
        ProfScope p(c, PF_INFO);
        HIPCHK(c, hipStreamWaitEvent(c->run_st, c->ev_ref, 0));
        L.stage = 8; launch_factored(L, c->run_st);
        HIPCHK(c, hipStreamWaitEvent(c->run_st, c->ev_join, 0));
        L.stage = 9; launch_factored(L, c->run_st);
    } else {
        if (big) { ProfScope p(c, PF_INFO); L.stage = 5; launch_factored(L, c->run_st); }      // split step / no side stream: in line
        { ProfScope p(c, PF_INFO); L.stage = 2; launch_factored(L, c->run_st); }
    }
    if (gnss_fuse) { if (int rc = gnss_in_frame_launch(c, b0, nb, L)) return rc; }
    if (phase == 2 && c->tok_wait) HIPCHK(c, hipStreamWaitEvent(c->run_st, c->tok_wait, 0));
    { ProfScope p(c, PF_APPLY); L.stage = 3; launch_factored(L, c->run_st); }
    if (phase == 2 && c->tok_rec) HIPCHK(c, hipEventRecord(c->tok_rec, c->run_st));
    return last_launch(c);
}

int run_msckf(ingvio_ctx* c, int b0, int nb, const MsckfOpts& op, int stereo, int fmax_used)
{
    if (c->method == 1) return run_msckf_factored(c, b0, nb, op, stereo, fmax_used);
    MsckfLaunch L;
    memset(&L, 0, sizeof L);
    L.stereo = stereo; L.cv = view(c); L.fv = fview(c); L.op = op; L.b0 = b0; L.nb = nb;
    L.fmax_used = fmax_used > 0 ? fmax_used : 1;
    L.gamma = c->d_gamma; L.accept = c->d_accept; L.used = c->d_used;
    L.Rpart = c->d_Rpart + (size_t)b0 * c->G * c->rstride; L.chunk_used = c->d_chunk_used + (size_t)b0 * c->G;
    L.G = c->G; L.rstride = c->rstride;
    L.Hout = c->d_H + (size_t)b0 * c->hstride; L.res_out = c->d_res + (size_t)b0 * c->mld;
    L.colmap = c->d_colmap + (size_t)b0 * c->cstride; L.m_out = c->d_m + b0; L.nc_out = c->d_nc + b0;
    L.mld = c->mld; L.hstride = c->hstride; L.cstride = c->cstride;
    { ProfScope p(c, PF_GATE); L.stage = 0; if (launch_msckf(L, c->st)) return INGVIO_E_UNSUPPORTED; }
    { ProfScope p(c, PF_FOLD); L.stage = 1; launch_msckf(L, c->st); }
    { ProfScope p(c, PF_MERGE); L.stage = 2; launch_msckf(L, c->st); }
    EkfLaunch E;
    memset(&E, 0, sizeof E);
    E.cv = view(c); E.b0 = b0; E.nb = nb; E.H = L.Hout; E.res = L.res_out; E.colmap = L.colmap; E.m = L.m_out; E.nc = L.nc_out;
    E.noise = c->d_noise + b0; E.r_kind = 0; E.mld = c->mld; E.hstride = c->hstride; E.cstride = c->cstride; E.nstride = 1;
    E.Y = c->d_Y + (size_t)b0 * c->ystride; E.ystride = c->ystride; E.dx = c->d_dx; E.status = c->d_status;
    E.m_cap = 6 * c->d.c_max; E.nc_cap = 6 * c->d.c_max;
    { ProfScope p(c, PF_EKF_CORE); launch_ekf_core(E, c->st); }
    { ProfScope p(c, PF_DOWNDATE); launch_downdate(E, c->d.n_max, c->st); }
    return last_launch(c);
}

int fill_noise_scalar(ingvio_ctx* c, int b0, int nb, double var)
{
    auto& u = c->upc;
    if (u.noise_ok && u.b0 == b0 && u.nb == nb && u.var == var) return 0;                           // already there
    std::vector<double> v(nb, var);
    const UpItem item = { c->d_noise + b0, v.data(), 8 * (size_t)nb };
    if (stage_small(c, &item, 1)) return INGVIO_E_HIP;
    u.noise_ok = true; u.b0 = b0; u.nb = nb; u.var = var;
    return 0;
}

}  // namespace

extern "C" {

int ingvio_ctx_create(const ingvio_ctx_desc* desc, ingvio_ctx** out)
{
    if (!desc || !out || desc->batch < 1 || desc->n_max < 21 || desc->c_max < 1 || desc->f_max < 1) return INGVIO_E_ARG;
    if (desc->c_max > bigwin_cmax()) return INGVIO_E_CAPACITY;      // 17..36 clones: factored path only (kernels_bigwin.hip)
    if (desc->m_max > 128) return INGVIO_E_CAPACITY;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= desc->device) return INGVIO_E_HIP;
    ingvio_ctx* c = new ingvio_ctx();
    c->d = *desc;
    if (c->d.m_max < 1) c->d.m_max = 32;
    c->err.clear();
    if (hipSetDevice(desc->device) != hipSuccess) { delete c; return INGVIO_E_HIP; }
    c->own_stream = desc->stream == nullptr;
    if (c->own_stream) {
        if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { delete c; return INGVIO_E_HIP; }
    } else c->st = (hipStream_t)desc->stream;
    c->run_st = c->st;
    if (const char* e = getenv("INGVIO_FRAME_PARTS")) c->parts_req = atoi(e);
    if (desc->c_max > 16) {
        // a higher priority than the compute stream's default: its short dependent launches take the slots the gate's workgroups free
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->st2, hipStreamNonBlocking, hi) != hipSuccess) c->st2 = nullptr;
        if (c->st2 && (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&c->ev_ref, hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) { hipStreamDestroy(c->st2); c->st2 = nullptr; }
    }
    const int B = desc->batch;
    c->ldp = (desc->n_max + 15) & ~15;
    // the factored kernels run at the padded width of their window class (6 / 11 / 16 clones): size the row/column
    // workspaces (M, the Pc copy) for the class, not for c_max itself
    const int cls0 = desc->c_max > 16 ? bigwin_cmax() : msckf_cmax_class(desc->c_max);
    const int cpad = cls0 > desc->c_max ? cls0 : desc->c_max;
    const int mcap = c->d.m_max > 6 * cpad ? c->d.m_max : 6 * cpad;
    c->mld = (mcap + 15) & ~15;
    // column capacity of a generic update (var_order may name any part of the state: 15 + 6 C + 3 L columns for the
    // landmark stack, LandmarkUpdate.cpp:32-149) is bounded by the state dimension, NOT by the row capacity m_max
    c->nc_cap = c->ldp > mcap ? c->ldp : mcap;
    c->G = (desc->c_max > 16 ? 256 : 512) / B;                              // ~2 resident gram workgroups per CU (large windows: 1,
    { const int gmax = desc->c_max > 16 ? 32 : 16; if (c->G > gmax) c->G = gmax; }   // and every chunk costs a 0.35 MB partial + sparse sums)
    if (c->G < 1) c->G = 1;
#ifdef INGVIO_ALT_KERNELS      // tuning knobs and forced code paths exist in variant builds only (tools/build_variant.sh alt -DINGVIO_ALT_KERNELS)
    if (const char* e = getenv("INGVIO_GRAM_CHUNKS")) { const int g = atoi(e); if (g >= 1 && g <= 64) c->G = g; }
#endif
    c->cls = msckf_cmax_class(desc->c_max);
    const int ncm = 6 * desc->c_max;
    c->rstride = ncm * (ncm + 1);
    c->hstride = c->mld * c->nc_cap;
    c->cstride = c->nc_cap;
    c->ystride = std::max(c->ldp * (c->mld + 4), c->mld * (c->mld + 1));      // Pc copy (ldp x MP) or M | t (MP x MP + MP)
    c->has_snap = false; c->staged = false; c->prof = false;
    c->method = 1;                                                          // information form; ingvio_set_msckf_method(0) selects the literal dense path
#ifdef INGVIO_ALT_KERNELS
    if (const char* e = getenv("INGVIO_MSCKF_METHOD")) c->method = (!strcmp(e, "dense") || !strcmp(e, "0")) ? 0 : 1;
#endif
    memset(c->prof_ms, 0, sizeof c->prof_ms); memset(c->prof_calls, 0, sizeof c->prof_calls);
    c->h_n.assign(B, 0); c->h_cur.assign(B, 0); c->h_n_snap.assign(B, 0); c->st_marg.assign(B, -1); c->st_cidx_hi.assign(B, -1); c->st_gnss.assign((size_t)B * 5, -2);
    c->h_nclones.assign(B, desc->c_max);
    // Consecutive filters' covariances must not sit a power of two apart: with ldp = 256 the stride would be 512 KB, and the SAME
    // element of every filter (the window block P_cc every gate wave of a filter reads, 64 filters per XCD) would fall into the same
    // L2 sets.  67 cache lines of pad walk the filters through the sets.  (Precaution: one default bench run showed the gate at
    // 1.30 ms instead of 0.28 in its config-3 pass; tools/gpu_alloc_sensitivity.py could not reproduce that with either layout -
    // six contexts per process, perturbed allocations, 0.289 - 0.294 ms for both -, so the pad is not the proven cure.)
#ifdef INGVIO_ALT_KERNELS
    static const bool no_pad = [] { const char* e = getenv("INGVIO_P_PAD"); return e && e[0] == '0'; }();
#else
    constexpr bool no_pad = false;
#endif
    c->pp = (size_t)c->ldp * c->ldp + (no_pad ? 0 : 67 * 16);
    const size_t pp = c->pp;
    const int cm = desc->c_max, fm = desc->f_max;
    int rc = 0;
    rc |= dalloc(c, &c->Pbase, 2 * (size_t)B * pp);
    rc |= dalloc(c, &c->Psnap, (size_t)B * pp);
    rc |= dalloc(c, &c->d_cur, B); rc |= dalloc(c, &c->d_n, B); rc |= dalloc(c, &c->d_n_snap, B);
    rc |= dalloc(c, &c->d_Phi, (size_t)B * KMAX * 225); rc |= dalloc(c, &c->d_G, (size_t)B * KMAX * 180);
    rc |= dalloc(c, &c->d_dt, (size_t)B * KMAX); rc |= dalloc(c, &c->d_R, (size_t)B * 9);
    rc |= dalloc(c, &c->d_blk, (size_t)B * 36); rc |= dalloc(c, &c->d_gnss, (size_t)B * 5); rc |= dalloc(c, &c->d_idx, B);
    rc |= dalloc(c, &c->d_zero_idx, B);
    if (!rc && hipMemset(c->d_zero_idx, 0, sizeof(int) * (size_t)B) != hipSuccess) rc = 1;
    {
        FrameSlabPtrs q;
        if (frame_slab_alloc(c, &c->d_frame_slab[0], &q)) rc |= 1;
        else {
            c->d_clone_idx = q.clone_idx; c->d_nclones = q.nclones; c->d_nfeat = q.nfeat; c->d_anchor = q.anchor; c->d_dof = q.dof;
            c->d_clone_R = q.clone_R; c->d_clone_p = q.clone_p; c->d_pf = q.pf; c->d_uv = q.uv; c->d_mask = q.mask;
        }
    }
    rc |= dalloc(c, &c->d_chi2, CHI2_CAP);
    {
        // the results an update call hands back - dx, gamma, used flags, row count, status - sit in ONE device slab (and a pinned
        // mirror): a whole-batch fetch is one device-to-host copy instead of five (~18 us each for a single filter)
        const size_t o_dx = 0, o_gam = pad64(8 * (size_t)B * c->ldp), o_used = o_gam + pad64(8 * (size_t)B * fm), o_m = o_used + pad64(4 * (size_t)B * fm),
                     o_st = o_m + pad64(4 * (size_t)B), o_tok = o_st + pad64(4 * (size_t)B), o_tpf = o_tok + pad64(4 * (size_t)B * fm),
                     tot = o_tpf + pad64(8 * (size_t)B * fm * 3);
        if (dalloc(c, &c->d_result_slab, tot) || hipHostMalloc((void**)&c->h_result, tot, hipHostMallocDefault) != hipSuccess) rc |= 1;
        else {
            c->result_bytes = tot; c->ro_gam = o_gam; c->ro_used = o_used; c->ro_m = o_m; c->ro_st = o_st; c->ro_tok = o_tok; c->ro_tpf = o_tpf;
            c->d_dx = (double*)(c->d_result_slab + o_dx); c->d_gamma = (double*)(c->d_result_slab + o_gam);
            c->d_used = (int*)(c->d_result_slab + o_used); c->d_m = (int*)(c->d_result_slab + o_m); c->d_status = (int*)(c->d_result_slab + o_st);
        }
    }
    rc |= dalloc(c, &c->d_accept, (size_t)B * fm);
    rc |= dalloc(c, &c->d_Rpart, (size_t)B * c->G * c->rstride); rc |= dalloc(c, &c->d_chunk_used, (size_t)B * c->G);
    rc |= dalloc(c, &c->d_H, (size_t)B * c->hstride); rc |= dalloc(c, &c->d_res, (size_t)B * c->mld);
    rc |= dalloc(c, &c->d_colmap, (size_t)B * c->cstride); rc |= dalloc(c, &c->d_nc, B); rc |= dalloc(c, &c->d_pcbase, 2 * (size_t)B);      // [B] pc_base | [B] arrival counters of the fused flip (k_info_apply)
    rc |= dalloc(c, &c->d_tri_ok, (size_t)B * fm);
    rc |= dalloc(c, &c->d_imu, (size_t)IMU_SLAB_NB * (8 * (size_t)KMAX * (225 + 180 + 1) + 64) + 256);
    if (desc->c_max <= 16) rc |= dalloc(c, &c->d_Tflat, (size_t)std::min(B, APPLY_FLAT_NB) * c->ldp * 100);
    if (desc->c_max <= 16 && c->G > 1) { rc |= dalloc(c, &c->d_Asum, (size_t)B * c->rstride); rc |= dalloc(c, &c->d_used_sum, B); }
    c->d_big_sg = nullptr; c->d_big_wk = nullptr;
    if (desc->c_max > 16) {
        rc |= dalloc(c, &c->d_big_sg, (size_t)B * bigwin_sg_doubles(c->G)); rc |= dalloc(c, &c->d_big_wk, (size_t)B * bigwin_wk_doubles());
    }
    rc |= dalloc(c, &c->d_noise, B); rc |= dalloc(c, &c->d_noise1, (size_t)c->mld * c->mld);
    rc |= dalloc(c, &c->d_hnew, (size_t)c->mld * 6);
    rc |= dalloc(c, &c->d_Y, (size_t)B * c->ystride); rc |= dalloc(c, &c->d_Yc, (size_t)B * c->ystride);
    rc |= dalloc(c, &c->d_rec, (size_t)B * fm * factored_rec_size(desc->c_max));
    if (rc || hipStreamSynchronize(c->st) != hipSuccess) { *out = c; return INGVIO_E_HIP; }
    *out = c;
    return INGVIO_OK;
}

int ingvio_ctx_destroy(ingvio_ctx* c)
{
    if (!c) return INGVIO_E_ARG;
    for (int p = 0; p < c->parts_alloc; ++p) hipStreamSynchronize(c->part[p].st);
    hipStreamSynchronize(c->st);
    void* ptrs[] = { c->Pbase, c->Psnap, c->d_cur, c->d_n, c->d_n_snap, c->d_Phi, c->d_G, c->d_dt, c->d_R, c->d_blk, c->d_gnss,
                     c->d_idx, c->d_frame_slab[0], c->d_frame_slab[1],
                     c->d_chi2, c->d_result_slab, c->d_accept, c->d_Rpart, c->d_chunk_used,
                     c->d_H, c->d_res, c->d_colmap, c->d_nc, c->d_noise, c->d_noise1, c->d_Y, c->d_Yc, c->d_rec, c->d_pcbase, c->d_big_sg, c->d_big_wk, c->d_tri_ok, c->d_hnew, c->d_multi, c->d_noiseB,
                     c->gn.H, c->gn.res, c->gn.noise, c->gn.gamma, c->gn.chi2, c->gn.m, c->gn.nc, c->gn.colmap, c->gn.keep,
                     c->gn.feph, c->gn.fobs, c->gn.frcv, c->gn.front,
                     c->dw.Hd, c->dw.X, c->dw.Y, c->dw.Tb, c->dw.noise, c->dw.noiseB, c->dw.m, c->dw.cidx, c->lm.pose, c->lm.pf, c->lm.uv, c->lm.gamma, c->lm.idx, c->lm.n_lm,
                     c->lm.lm_idx, c->lm.anchor_idx, c->lm.tracked, c->lm.accept, c->lm.dx, c->d_xchg, c->dw.U, c->dw.rowmap, c->d_zero_idx,
                     c->d_Asum, c->d_used_sum, c->d_Tflat, c->d_imu, c->d_tri_mask };
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& sl : c->pin) { if (sl.p) hipHostFree(sl.p); if (sl.ev) hipEventDestroy(sl.ev); }
    if (c->h_result) hipHostFree(c->h_result);
    if (c->qr.exec) hipGraphExecDestroy(c->qr.exec);
    for (double* p : { c->qr.dA, c->qr.db, c->qr.ws, c->qr.dT, c->se.e, c->se.o, c->se.r, c->se.f, c->gn.W, c->gn.Yf, c->gn.dxf }) if (p) hipFree(p);
    for (int* p : { c->gn.mf, c->gn.ncf }) if (p) hipFree(p);
    {
        auto& a = c->alt;
        void* ap[] = { a.Phi, a.G, a.dt, a.R, a.gnss, a.idx, a.chi2, a.noise };      // its frame arrays: d_frame_slab (freed above)
        for (void* p : ap) if (p) hipFree(p);
        if (c->st_copy) hipStreamDestroy(c->st_copy);
        if (c->ev_copy) hipEventDestroy(c->ev_copy);
        for (auto e : c->ev_free) if (e) hipEventDestroy(e);
    }
    for (void* p : { (void*)c->trk.uv, (void*)c->trk.pf, (void*)c->trk.mask, (void*)c->trk.stage[0], (void*)c->trk.stage[1] }) if (p) hipFree(p);
    for (auto& r : c->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (int p = 0; p < c->parts_alloc; ++p) { hipStreamDestroy(c->part[p].st); hipEventDestroy(c->part[p].ev_gate); hipEventDestroy(c->part[p].ev_apply); hipEventDestroy(c->part[p].ev_done); }
    if (c->ev_split_fork) hipEventDestroy(c->ev_split_fork);
    if (c->ev_fetch) hipEventDestroy(c->ev_fetch);
    if (c->st2) hipStreamDestroy(c->st2);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_ref) hipEventDestroy(c->ev_ref);
    if (c->own_stream) hipStreamDestroy(c->st);
    delete c;
    return INGVIO_OK;
}

int ingvio_sync(ingvio_ctx* c)
{
    ENTER(c);
    if (!c) return INGVIO_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    return INGVIO_OK;
}
void* ingvio_ctx_stream(ingvio_ctx* c)
{
    if (c && c->split_pending) join_parts(c);      // the caller orders its own work behind this stream
    return c ? (void*)c->st : nullptr;
}
const char* ingvio_last_error(ingvio_ctx* c) { return c ? c->err.c_str() : "null context"; }
int ingvio_ldp(ingvio_ctx* c) { return c ? c->ldp : 0; }
int ingvio_f_max(ingvio_ctx* c) { return c ? c->d.f_max : 0; }
int ingvio_c_max(ingvio_ctx* c) { return c ? c->d.c_max : 0; }

int ingvio_cov_set(ingvio_ctx* c, int b, const double* P, int ld, int n)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b, 1) || !P || n < 0 || n > c->d.n_max || ld < n) return INGVIO_E_ARG;
    double* dst = c->Pbase + ((size_t)c->h_cur[b] * c->d.batch + b) * c->pp;
    if (n) HIPCHK(c, hipMemcpy2DAsync(dst, 8 * (size_t)c->ldp, P, 8 * (size_t)ld, 8 * (size_t)n, n, hipMemcpyHostToDevice, c->st));
    c->h_n[b] = n;
    HIPCHK(c, hipMemcpyAsync(c->d_n + b, &c->h_n[b], sizeof(int), hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return INGVIO_OK;
}

int ingvio_cov_get(ingvio_ctx* c, int b, double* P, int ld)
{
    ENTER(c);
    if (check_range(c, b, 1) || !P) return INGVIO_E_ARG;
    const int n = c->h_n[b];
    if (ld < n) return INGVIO_E_ARG;
    const double* src = c->Pbase + ((size_t)c->h_cur[b] * c->d.batch + b) * c->pp;
    if (n) HIPCHK(c, hipMemcpy2DAsync(P, 8 * (size_t)ld, src, 8 * (size_t)c->ldp, 8 * (size_t)n, n, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    return INGVIO_OK;
}

int ingvio_get_n(ingvio_ctx* c, int b, int* n)
{
    ENTER(c);
    if (check_range(c, b, 1) || !n) return INGVIO_E_ARG;
    *n = c->h_n[b];
    return INGVIO_OK;
}

int ingvio_cov_get_marginal(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, double* out)
{
    ENTER(c);
    if (check_range(c, b, 1) || !vidx || !vsize || !out || k < 1) return INGVIO_E_ARG;
    const int n = c->h_n[b];
    std::vector<double> P((size_t)n * n);
    int rc = ingvio_cov_get(c, b, P.data(), n);
    if (rc) return rc;
    int ns = 0;
    for (int i = 0; i < k; ++i) { if (vidx[i] < 0 || vidx[i] + vsize[i] > n) return INGVIO_E_NOT_IN_STATE; ns += vsize[i]; }
    int r0 = 0;
    for (int a = 0; a < k; ++a) {
        int c0 = 0;
        for (int bb = 0; bb < k; ++bb) {
            for (int j = 0; j < vsize[bb]; ++j) for (int i = 0; i < vsize[a]; ++i)
                out[(size_t)(c0 + j) * ns + r0 + i] = P[(size_t)(vidx[bb] + j) * n + vidx[a] + i];
            c0 += vsize[bb];
        }
        r0 += vsize[a];
    }
    return INGVIO_OK;
}

int ingvio_cov_snapshot(ingvio_ctx* c)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (!c) return INGVIO_E_ARG;
    launch_snapshot(view(c), c->d.n_max, c->Psnap, c->d_n_snap, c->st);
    c->h_n_snap = c->h_n;
    c->has_snap = true;
    return last_launch(c);
}

int ingvio_cov_restore(ingvio_ctx* c)
{
    ENTER(c);
    if (!c || !c->has_snap) return INGVIO_E_ARG;
    c->phase_pending = false;                      // the way out of an abandoned split step: every filter returns to the snapshot
    launch_restore(view(c), 0, c->d.batch, c->d.n_max, c->Psnap, c->d_n_snap, c->st);
    c->h_n = c->h_n_snap;
    std::fill(c->h_cur.begin(), c->h_cur.end(), 0);
    return last_launch(c);
}

int ingvio_propagate_fused(ingvio_ctx* c, int b0, int nb, int k, const double* Phi, const double* G, const double* dt,
                           const double sigma[4], int enable_gnss, const int* gnss_idx, double scb, double srw)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || k < 1 || k > KMAX || !Phi || !G || !dt || !sigma) return INGVIO_E_ARG;
    for (int i = 0; i < nb; ++i) if (c->h_n[b0 + i] < 15) return INGVIO_E_ARG;
    if (wait_inputs(c)) return INGVIO_E_HIP;
    const bool with_gnss = enable_gnss && gnss_idx;
    if (c->d_imu && nb <= IMU_SLAB_NB) {
        // few filters: Phi | G | dt | gnss_idx packed into ONE pinned image and ONE copy into a device slab of the same layout (the four
        // arrays are four allocations, i.e. four copies of ~7 us each on the stream in front of the kernel - a third of what a single
        // filter's propagation costs its host)
        const size_t nPhi = (size_t)nb * k * 225, nG = (size_t)nb * k * 180, ndt = (size_t)nb * k;
        const size_t oG = 8 * nPhi, odt = oG + 8 * nG, ogn = pad64(odt + 8 * ndt), tot = ogn + (with_gnss ? sizeof(int) * (size_t)nb * 5 : 0);
        Uploader up{ c };
        if (up.begin(tot + 64)) return INGVIO_E_HIP;
        char* img = up.take<char>(tot);
        memcpy(img, Phi, 8 * nPhi); memcpy(img + oG, G, 8 * nG); memcpy(img + odt, dt, 8 * ndt);
        if (with_gnss) memcpy(img + ogn, gnss_idx, sizeof(int) * (size_t)nb * 5);
        up.copy(c->d_imu, img, tot);
        if (up.end()) return INGVIO_E_HIP;
        ProfScope p(c, PF_PROPAGATE);
        launch_propagate(view(c), b0, nb, c->d.n_max, (const double*)c->d_imu, (const double*)(c->d_imu + oG), (const double*)(c->d_imu + odt), k,
                         with_gnss ? (const int*)(c->d_imu + ogn) : nullptr, sigma, enable_gnss, scb, srw, c->st);
        return last_launch(c);
    }
    const UpItem items[4] = { { c->d_Phi, Phi, 8 * (size_t)nb * k * 225 }, { c->d_G, G, 8 * (size_t)nb * k * 180 }, { c->d_dt, dt, 8 * (size_t)nb * k },
                              { c->d_gnss, gnss_idx, with_gnss ? sizeof(int) * (size_t)nb * 5 : 0 } };
    if (stage_small(c, items, 4)) return INGVIO_E_HIP;
    {
        ProfScope p(c, PF_PROPAGATE);
        launch_propagate(view(c), b0, nb, c->d.n_max, c->d_Phi, c->d_G, c->d_dt, k, with_gnss ? c->d_gnss : nullptr,
                         sigma, enable_gnss, scb, srw, c->st);
    }
    return last_launch(c);                      // no stream synchronisation: the inputs were staged through the pinned ring (stage_small)
}

int ingvio_propagate(ingvio_ctx* c, int b0, int nb, const double* Phi, const double* G, const double* dt,
                     const double sigma[4], int enable_gnss, const int* gnss_idx, double scb, double srw)
{
    ENTER(c);
    return ingvio_propagate_fused(c, b0, nb, 1, Phi, G, dt, sigma, enable_gnss, gnss_idx, scb, srw);
}

int ingvio_augment_clone(ingvio_ctx* c, int b0, int nb, const double* R, int* new_idx)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !R) return INGVIO_E_ARG;
    for (int i = 0; i < nb; ++i) {
        if (c->h_n[b0 + i] < 21) return INGVIO_E_ARG;
        if (c->h_n[b0 + i] + 6 > c->d.n_max) return INGVIO_E_CAPACITY;
    }
    if (wait_inputs(c)) return INGVIO_E_HIP;
    if (nb == 1) { ProfScope p(c, PF_AUGMENT); launch_augment_one(view(c), b0, R, c->st); }      // R as a kernel argument
    else {
        const UpItem item = { c->d_R, R, 8 * (size_t)nb * 9 };
        if (stage_small(c, &item, 1)) return INGVIO_E_HIP;
        ProfScope p(c, PF_AUGMENT); launch_augment(view(c), b0, nb, c->d_R, c->st);
    }
    for (int i = 0; i < nb; ++i) { if (new_idx) new_idx[i] = c->h_n[b0 + i]; c->h_n[b0 + i] += 6; }
    return last_launch(c);
}

int ingvio_marginalize(ingvio_ctx* c, int b0, int nb, const int* idx, int size)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !idx || size < 1) return INGVIO_E_ARG;
    for (int i = 0; i < nb; ++i)
        if (idx[i] >= 0 && idx[i] + size > c->h_n[b0 + i]) return INGVIO_E_NOT_IN_STATE;
    if (wait_inputs(c)) return INGVIO_E_HIP;
    if (nb == 1) { ProfScope p(c, PF_MARG); launch_marginalize(view(c), b0, 1, c->d.n_max, nullptr, size, c->st, idx[0]); }      // the index as a kernel argument
    else {
        const UpItem item = { c->d_idx, idx, sizeof(int) * (size_t)nb };
        if (stage_small(c, &item, 1)) return INGVIO_E_HIP;
        ProfScope p(c, PF_MARG); launch_marginalize(view(c), b0, nb, c->d.n_max, c->d_idx, size, c->st);
    }
    for (int i = 0; i < nb; ++i) if (idx[i] >= 0) { c->h_n[b0 + i] -= size; c->h_cur[b0 + i] ^= 1; }
    return last_launch(c);
}

int ingvio_append_independent(ingvio_ctx* c, int b0, int nb, int size, const double* blk, int* new_idx)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !blk || size < 1 || size > 6) return INGVIO_E_ARG;
    for (int i = 0; i < nb; ++i) if (c->h_n[b0 + i] + size > c->d.n_max) return INGVIO_E_CAPACITY;
    if (wait_inputs(c)) return INGVIO_E_HIP;
    const UpItem item = { c->d_blk, blk, 8 * (size_t)nb * size * size };
    if (stage_small(c, &item, 1)) return INGVIO_E_HIP;
    launch_append(view(c), b0, nb, size, c->d_blk, c->st);
    for (int i = 0; i < nb; ++i) { if (new_idx) new_idx[i] = c->h_n[b0 + i]; c->h_n[b0 + i] += size; }
    return last_launch(c);
}

static int dense_ws_alloc(ingvio_ctx* c, int m_need);
static int run_dense_update(ingvio_ctx* c, int b0, int nb, double var, int r_kind, const double* d_noise, int nstride, double* d_dx, bool products_done = false,
                            const int* rowmap = nullptr, const int* marg_idx = nullptr, int marg_size = 0, bool* marg_fused = nullptr);
#define DENSE_M_MAX 1024      // rows of one generic update through the dense-H route (S factorised out of HBM, kernels_chol.hip)

// ingvio_ekf_update for row counts whose S does not fit in LDS (or beyond the context's m_max): the host scatters the columns
// into the dense row layout of kernels_lmbatch.hip and the update runs as GEMM + Cholesky sweep + downdate.
static int ekf_update_dense_route(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, const double* H, int ldh, int m,
                                  const double* res, const double* R, int r_kind, double* dx_out)
{
    if (check_range(c, b, 1) || !vidx || !vsize || !H || !res || !R || k < 1 || m < 1 || ldh < m || r_kind < 0 || r_kind > 2) return INGVIO_E_ARG;
    if (m > DENSE_M_MAX) return INGVIO_E_CAPACITY;
    for (int i = 0; i < k; ++i) if (vidx[i] < 0 || vsize[i] < 0 || vidx[i] + vsize[i] > c->h_n[b]) return INGVIO_E_NOT_IN_STATE;
    if (int rc = dense_ws_alloc(c, m)) return rc;
    auto& w = c->dw;
    std::vector<double> Hd((size_t)w.m_cap * w.n_ld, 0.0), rr((size_t)w.m_cap, 0.0);
    int col = 0;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < vsize[i]; ++j, ++col)
            for (int r = 0; r < m; ++r) Hd[(size_t)r * w.n_ld + vidx[i] + j] += H[(size_t)r + (size_t)col * ldh];
    memcpy(rr.data(), res, 8 * (size_t)m);
    int zero = 0, rc = up(c, w.Hd + (size_t)b * w.hstride, Hd.data(), 8 * Hd.size());
    if (hipMemcpy2DAsync(w.X + (size_t)b * w.xstride + w.m_cap + w.n32, 8 * (size_t)w.ldx, rr.data(), 8, 8, (size_t)w.m_cap, hipMemcpyHostToDevice, c->st) != hipSuccess) rc = INGVIO_E_HIP;
    rc |= up(c, w.m + b, &m, sizeof(int));
    rc |= up(c, w.noise, R, 8 * (size_t)(r_kind == 0 ? 1 : (r_kind == 1 ? m : (size_t)m * m)));
    rc |= up(c, c->d_status + b, &zero, sizeof(int));
    if (rc) return INGVIO_E_HIP;
    HIPCHK(c, hipStreamSynchronize(c->st));                                       // the host vectors go out of scope
    rc = run_dense_update(c, b, 1, 0.0, r_kind, w.noise, 0, c->d_dx);
    if (rc) return rc;
    int status = 0;
    if (dx_out && down_sync(c, dx_out, c->d_dx + (size_t)b * c->ldp, 8 * (size_t)c->h_n[b])) return INGVIO_E_HIP;
    if (down_sync(c, &status, c->d_status + b, sizeof(int))) return INGVIO_E_HIP;
    return (status & 4) ? INGVIO_E_NOT_PD : ((status & 2) ? INGVIO_NEG_DIAG : INGVIO_OK);
}

// k_ekf_core keeps S (+ one border row/column) and the column map in LDS (launch_ekf_core): 160 KB per workgroup on gfx950
static bool ekf_core_fits(int m, int nc)
{
    return sizeof(double) * (size_t)(m + 1) * (m + 1) + sizeof(int) * (size_t)nc + 16 <= 160 * 1024;
}

static int stage_generic(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, const double* H, int ldh, int m,
                         const double* res, const double* R, int r_kind, int* nc_out)
{
    if (check_range(c, b, 1) || !vidx || !vsize || !H || !res || !R || k < 1 || m < 1 || ldh < m) return INGVIO_E_ARG;
    if (r_kind < 0 || r_kind > 2) return INGVIO_E_ARG;
    if ((m > c->d.m_max && m > 6 * c->d.c_max) || m > c->mld) return INGVIO_E_CAPACITY;
    int nc = 0;
    std::vector<int> cm;
    for (int i = 0; i < k; ++i) {
        if (vidx[i] < 0 || vidx[i] + vsize[i] > c->h_n[b]) return INGVIO_E_NOT_IN_STATE;      // checkSubOrder
        for (int j = 0; j < vsize[i]; ++j) cm.push_back(vidx[i] + j);
        nc += vsize[i];
    }
    if (nc > c->nc_cap) return INGVIO_E_CAPACITY;
    std::vector<double> Hp((size_t)c->mld * nc, 0.0);
    for (int cc = 0; cc < nc; ++cc) memcpy(&Hp[(size_t)cc * c->mld], H + (size_t)cc * ldh, 8 * (size_t)m);
    int rc = up(c, c->d_H + (size_t)b * c->hstride, Hp.data(), 8 * Hp.size());
    rc |= up(c, c->d_res + (size_t)b * c->mld, res, 8 * (size_t)m);
    rc |= up(c, c->d_colmap + (size_t)b * c->cstride, cm.data(), sizeof(int) * (size_t)nc);
    rc |= up(c, c->d_m + b, &m, sizeof(int));
    rc |= up(c, c->d_nc + b, &nc, sizeof(int));
    rc |= up(c, c->d_noise1, R, 8 * (size_t)(r_kind == 0 ? 1 : (r_kind == 1 ? m : m * m)));
    if (rc) return INGVIO_E_HIP;
    HIPCHK(c, hipStreamSynchronize(c->st));
    *nc_out = nc;
    return 0;
}

int ingvio_ekf_update(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, const double* H, int ldh, int m,
                      const double* res, const double* R, int r_kind, double* dx_out)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    int nc = 0;
    if (c && vsize && k >= 1) {                                     // S beyond LDS / rows beyond m_max: the dense route
        int ncq = 0;
        for (int i = 0; i < k; ++i) ncq += vsize[i];
        if (m > c->mld || (m > c->d.m_max && m > 6 * c->d.c_max) || ncq > c->nc_cap || !ekf_core_fits(m, ncq))
            return ekf_update_dense_route(c, b, vidx, vsize, k, H, ldh, m, res, R, r_kind, dx_out);
    }
    int rc = stage_generic(c, b, vidx, vsize, k, H, ldh, m, res, R, r_kind, &nc);
    if (rc) return rc;
    if (!ekf_core_fits(m, nc)) return INGVIO_E_CAPACITY;          // before anything touches the covariance
    int zero = 0;
    if (up(c, c->d_status + b, &zero, sizeof(int))) return INGVIO_E_HIP;
    EkfLaunch E;
    memset(&E, 0, sizeof E);
    E.cv = view(c); E.b0 = b; E.nb = 1; E.H = c->d_H + (size_t)b * c->hstride; E.res = c->d_res + (size_t)b * c->mld;
    E.colmap = c->d_colmap + (size_t)b * c->cstride; E.m = c->d_m + b; E.nc = c->d_nc + b;
    E.noise = c->d_noise1; E.r_kind = r_kind; E.mld = c->mld; E.hstride = c->hstride; E.cstride = c->cstride;
    E.nstride = c->mld * c->mld; E.Y = c->d_Y + (size_t)b * c->ystride; E.ystride = c->ystride; E.dx = c->d_dx;
    E.status = c->d_status; E.m_cap = m; E.nc_cap = nc;
    { ProfScope p(c, PF_EKF_CORE); launch_ekf_core(E, c->st); }
    { ProfScope p(c, PF_DOWNDATE); launch_downdate(E, c->h_n[b], c->st); }
    int status = 0;
    if (dx_out && down_sync(c, dx_out, c->d_dx + (size_t)b * c->ldp, 8 * (size_t)c->h_n[b])) return INGVIO_E_HIP;
    if (down_sync(c, &status, c->d_status + b, sizeof(int))) return INGVIO_E_HIP;
    rc = last_launch(c);
    if (rc) return rc;
    return (status & 4) ? INGVIO_E_NOT_PD : ((status & 2) ? INGVIO_NEG_DIAG : INGVIO_OK);
}

// ekfUpdate for filters [b0, b0+nb) in one launch and one synchronisation (the GNSS update of a whole batch, config 3 x 4):
// block i = filter b0+i with its own var_order / H / res / noise (R scalar or diagonal).  dx_out [nb][ldp], status_out [nb]
// (INGVIO_OK / INGVIO_NEG_DIAG per filter, may be NULL).
int ingvio_ekf_update_batch(ingvio_ctx* c, int b0, int nb, const ingvio_update_block* blk, int r_kind, double* dx_out, int* status_out)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !blk) return INGVIO_E_ARG;
    if (r_kind != INGVIO_R_SCALAR && r_kind != INGVIO_R_DIAG) return INGVIO_E_UNSUPPORTED;
    int m_cap = 0, nc_cap = 0, n_cap = 0;
    bool dense = false;
    std::vector<int> ncs(nb);
    for (int i = 0; i < nb; ++i) {
        const ingvio_update_block& q = blk[i];
        const int b = b0 + i;
        if (!q.vidx || !q.vsize || !q.H || !q.res || !q.R || q.k < 1 || q.m < 1 || q.ldh < q.m) return INGVIO_E_ARG;
        if (q.m > DENSE_M_MAX) return INGVIO_E_CAPACITY;
        int nc = 0;
        for (int j = 0; j < q.k; ++j) {
            if (q.vidx[j] < 0 || q.vsize[j] < 0 || q.vidx[j] + q.vsize[j] > c->h_n[b]) return INGVIO_E_NOT_IN_STATE;      // checkSubOrder
            nc += q.vsize[j];
        }
        if (q.m > c->mld || nc > c->nc_cap || !ekf_core_fits(q.m, nc)) dense = true;      // S beyond LDS: the whole call takes the dense-H route
        ncs[i] = nc;
        if (q.m > m_cap) m_cap = q.m;
        if (nc > nc_cap) nc_cap = nc;
        if (c->h_n[b] > n_cap) n_cap = c->h_n[b];
    }
    if (dense) {
        // GEMM + Cholesky sweep with carried rows + downdate on dense rows addressed by state column (kernels_lmbatch.hip /
        // kernels_chol.hip); the host scatters the columns, filter by filter
        if (int rc = dense_ws_alloc(c, m_cap)) return rc;
        auto& w = c->dw;
        std::vector<double> Hd((size_t)w.m_cap * w.n_ld), rr((size_t)w.m_cap), nz((size_t)w.m_cap);
        std::vector<int> zeros(nb, 0), ms(nb);
        int rc = 0;
        for (int i = 0; i < nb && !rc; ++i) {
            const ingvio_update_block& q = blk[i];
            std::fill(Hd.begin(), Hd.end(), 0.0); std::fill(rr.begin(), rr.end(), 0.0); std::fill(nz.begin(), nz.end(), 0.0);
            int col = 0;
            for (int j = 0; j < q.k; ++j)
                for (int t = 0; t < q.vsize[j]; ++t, ++col)
                    for (int r = 0; r < q.m; ++r) Hd[(size_t)r * w.n_ld + q.vidx[j] + t] += q.H[(size_t)r + (size_t)col * q.ldh];
            memcpy(rr.data(), q.res, 8 * (size_t)q.m);
            memcpy(nz.data(), q.R, 8 * (size_t)(r_kind == INGVIO_R_SCALAR ? 1 : q.m));
            ms[i] = q.m;
            rc |= up(c, w.Hd + (size_t)(b0 + i) * w.hstride, Hd.data(), 8 * Hd.size());
            if (hipMemcpy2DAsync(w.X + (size_t)(b0 + i) * w.xstride + w.m_cap + w.n32, 8 * (size_t)w.ldx, rr.data(), 8, 8, (size_t)w.m_cap,
                                 hipMemcpyHostToDevice, c->st) != hipSuccess) rc = INGVIO_E_HIP;
            rc |= up(c, w.noiseB + (size_t)(b0 + i) * w.m_cap, nz.data(), 8 * nz.size());
            if (!rc && hipStreamSynchronize(c->st) != hipSuccess) rc = INGVIO_E_HIP;      // the staging vectors are reused
        }
        rc |= up(c, w.m + b0, ms.data(), sizeof(int) * (size_t)nb);
        rc |= up(c, c->d_status + b0, zeros.data(), sizeof(int) * (size_t)nb);
        if (rc) return INGVIO_E_HIP;
        HIPCHK(c, hipStreamSynchronize(c->st));
        rc = run_dense_update(c, b0, nb, 0.0, r_kind, w.noiseB + (size_t)b0 * w.m_cap, w.m_cap, c->d_dx);
        if (rc) return rc;
        std::vector<int> status(nb, 0);
        if (dx_out) HIPCHK(c, hipMemcpyAsync(dx_out, c->d_dx + (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp, hipMemcpyDeviceToHost, c->st));
        if (down_sync(c, status.data(), c->d_status + b0, sizeof(int) * (size_t)nb)) return INGVIO_E_HIP;
        int soft = INGVIO_OK;
        for (int i = 0; i < nb; ++i) {
            const int st = (status[i] & 4) ? INGVIO_E_NOT_PD : ((status[i] & 2) ? INGVIO_NEG_DIAG : INGVIO_OK);
            if (status_out) status_out[i] = st;
            if (st != INGVIO_OK) soft = st;
        }
        return soft;
    }
    if (!c->d_noiseB) {
        if (dalloc(c, &c->d_noiseB, (size_t)c->d.batch * c->mld)) return INGVIO_E_HIP;
    }
    // compact strides for this call: only the widest var_order of the batch travels over PCIe (the d_H / d_colmap slots are
    // sized for nc_cap = ldp columns per filter, so the compact layout always fits)
    const size_t mld = c->mld, hs = (size_t)c->mld * nc_cap, cs = nc_cap;
    Uploader upl{ c };
    int rc = upl.begin(pad64(8 * (size_t)nb * hs) + pad64(8 * (size_t)nb * mld) * 2 + pad64(4 * (size_t)nb * cs) + pad64(4 * (size_t)nb) * 3 + 1024);
    if (rc) return rc;
    double* hH = upl.take<double>((size_t)nb * hs); double* hr = upl.take<double>((size_t)nb * mld); double* hn = upl.take<double>((size_t)nb * mld);
    int* hc = upl.take<int>((size_t)nb * cs); int* hm = upl.take<int>(nb); int* hnc = upl.take<int>(nb); int* hz = upl.take<int>(nb);
    parallel_for(nb, [=, &ncs](int i) {
        const ingvio_update_block& q = blk[i];
        const int nc = ncs[i];
        double* Hd = hH + (size_t)i * hs;
        for (int cc = 0; cc < nc; ++cc) {
            memcpy(Hd + (size_t)cc * mld, q.H + (size_t)cc * q.ldh, 8 * (size_t)q.m);
            memset(Hd + (size_t)cc * mld + q.m, 0, 8 * (mld - q.m));                              // k_ekf_core reads whole 16-row groups
        }
        memcpy(hr + (size_t)i * mld, q.res, 8 * (size_t)q.m);
        memcpy(hn + (size_t)i * mld, q.R, 8 * (size_t)(r_kind == INGVIO_R_SCALAR ? 1 : q.m));
        int* cm = hc + (size_t)i * cs;
        int w = 0;
        for (int j = 0; j < q.k; ++j) for (int t = 0; t < q.vsize[j]; ++t) cm[w++] = q.vidx[j] + t;
        hm[i] = q.m; hnc[i] = nc; hz[i] = 0;
    });
    upl.copy(c->d_H + (size_t)b0 * hs, hH, (size_t)nb * hs);
    upl.copy(c->d_res + (size_t)b0 * mld, hr, (size_t)nb * mld);
    upl.copy(c->d_noiseB + (size_t)b0 * mld, hn, (size_t)nb * mld);
    upl.copy(c->d_colmap + (size_t)b0 * cs, hc, (size_t)nb * cs);
    upl.copy(c->d_m + b0, hm, nb); upl.copy(c->d_nc + b0, hnc, nb); upl.copy(c->d_status + b0, hz, nb);
    rc = upl.end();
    if (rc) return rc;
    EkfLaunch E;
    memset(&E, 0, sizeof E);
    E.cv = view(c); E.b0 = b0; E.nb = nb; E.H = c->d_H + (size_t)b0 * hs; E.res = c->d_res + (size_t)b0 * mld;
    E.colmap = c->d_colmap + (size_t)b0 * cs; E.m = c->d_m + b0; E.nc = c->d_nc + b0;
    E.noise = c->d_noiseB + (size_t)b0 * mld; E.r_kind = r_kind; E.mld = c->mld; E.hstride = (int)hs; E.cstride = (int)cs;
    E.nstride = c->mld; E.Y = c->d_Y + (size_t)b0 * c->ystride; E.ystride = c->ystride; E.dx = c->d_dx;
    E.status = c->d_status; E.m_cap = m_cap; E.nc_cap = nc_cap;
    { ProfScope p(c, PF_EKF_CORE); launch_ekf_core(E, c->st); }
    { ProfScope p(c, PF_DOWNDATE); launch_downdate(E, n_cap, c->st); }
    std::vector<int> status(nb, 0);
    if (dx_out) HIPCHK(c, hipMemcpyAsync(dx_out, c->d_dx + (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp, hipMemcpyDeviceToHost, c->st));
    if (down_sync(c, status.data(), c->d_status + b0, sizeof(int) * (size_t)nb)) return INGVIO_E_HIP;
    rc = last_launch(c);
    if (rc) return rc;
    int soft = INGVIO_OK;
    for (int i = 0; i < nb; ++i) {
        const int st = (status[i] & 4) ? INGVIO_E_NOT_PD : ((status[i] & 2) ? INGVIO_NEG_DIAG : INGVIO_OK);
        if (status_out) status_out[i] = st;
        if (st != INGVIO_OK) soft = st;
    }
    return soft;
}

// ---- GnssUpdate::updateTrackedSys for a batch (GnssUpdate.cpp:148-290): per-row gates, compaction, block gate, ekfUpdate ----

static int gnss_alloc(ingvio_ctx* c)
{
    auto& g = c->gn;
    if (g.H) return 0;
    const size_t B = c->d.batch, mld = c->mld, hs = mld * GNSS_NCW;
    int rc = 0;
    rc |= dalloc(c, &g.H, B * hs); rc |= dalloc(c, &g.res, B * mld); rc |= dalloc(c, &g.noise, B * mld); rc |= dalloc(c, &g.gamma, B * mld);
    rc |= dalloc(c, &g.chi2, CHI2_CAP); rc |= dalloc(c, &g.m, B); rc |= dalloc(c, &g.nc, B); rc |= dalloc(c, &g.colmap, B * GNSS_NCW);
    rc |= dalloc(c, &g.keep, B * mld);
    if (!c->d_noiseB) rc |= dalloc(c, &c->d_noiseB, B * mld);
    if (rc) return INGVIO_E_HIP;
    g.hi.assign(B, 0);
    return 0;
}

int ingvio_gnss_stage(ingvio_ctx* c, int b0, int nb, const ingvio_update_block* blk, const ingvio_gnss_opts* o)
{
    ENTER(c);
    if (check_range(c, b0, nb) || !blk || !o) return INGVIO_E_ARG;
    if ((o->gate_rows || o->strong_reject) && (!o->chi2_table || o->chi2_len < 2 || o->chi2_len > CHI2_CAP)) return INGVIO_E_ARG;
    auto& g = c->gn;
    int m_cap = 0;
    std::vector<int> ncs(nb), his(nb);
    for (int i = 0; i < nb; ++i) {
        const ingvio_update_block& q = blk[i];
        if (q.m < 0 || q.k < 0) return INGVIO_E_ARG;
        if (q.m == 0) { ncs[i] = 0; his[i] = 0; continue; }                     // no measurement for this filter
        if (!q.vidx || !q.vsize || !q.H || !q.res || !q.R || q.k < 1 || q.ldh < q.m) return INGVIO_E_ARG;
        if (q.m > c->mld || q.m > 256) return INGVIO_E_CAPACITY;
        int nc = 0, hi = 0;
        for (int j = 0; j < q.k; ++j) {
            if (q.vidx[j] < 0 || q.vsize[j] < 1 || q.vidx[j] + q.vsize[j] > c->d.n_max) return INGVIO_E_NOT_IN_STATE;
            if (q.vidx[j] + q.vsize[j] > hi) hi = q.vidx[j] + q.vsize[j];
            nc += q.vsize[j];
        }
        if (nc > GNSS_NCW || !ekf_core_fits(q.m, nc)) return INGVIO_E_CAPACITY;
        ncs[i] = nc; his[i] = hi;
        if (q.m > m_cap) m_cap = q.m;
    }
    const size_t mld = c->mld, hs = mld * GNSS_NCW;
    if (gnss_alloc(c)) return INGVIO_E_HIP;
    Uploader upl{ c };
    int rc = upl.begin(pad64(8 * (size_t)nb * hs) + 2 * pad64(8 * (size_t)nb * mld) + pad64(4 * (size_t)nb * GNSS_NCW) + 2 * pad64(4 * (size_t)nb) +
                       pad64(8 * CHI2_CAP) + 1024);
    if (rc) return rc;
    double* hH = upl.take<double>((size_t)nb * hs); double* hr = upl.take<double>((size_t)nb * mld); double* hn = upl.take<double>((size_t)nb * mld);
    int* hc = upl.take<int>((size_t)nb * GNSS_NCW); int* hm = upl.take<int>(nb); int* hnc = upl.take<int>(nb);
    double* hchi = upl.take<double>(CHI2_CAP);
    parallel_for(nb, [=, &ncs](int i) {
        const ingvio_update_block& q = blk[i];
        const int nc = ncs[i];
        double* Hd = hH + (size_t)i * hs;
        for (int cc = 0; cc < nc; ++cc) {
            memcpy(Hd + (size_t)cc * mld, q.H + (size_t)cc * q.ldh, 8 * (size_t)q.m);
            memset(Hd + (size_t)cc * mld + q.m, 0, 8 * (mld - q.m));
        }
        if (q.m) { memcpy(hr + (size_t)i * mld, q.res, 8 * (size_t)q.m); memcpy(hn + (size_t)i * mld, q.R, 8 * (size_t)q.m); }
        int* cm = hc + (size_t)i * GNSS_NCW;
        int w = 0;
        if (q.m) for (int j = 0; j < q.k; ++j) for (int t = 0; t < q.vsize[j]; ++t) cm[w++] = q.vidx[j] + t;
        hm[i] = q.m; hnc[i] = nc;
    });
    for (int i = 0; i < nb; ++i) g.hi[b0 + i] = his[i];
    upl.copy(g.H + (size_t)b0 * hs, hH, (size_t)nb * hs);
    upl.copy(g.res + (size_t)b0 * mld, hr, (size_t)nb * mld);
    upl.copy(g.noise + (size_t)b0 * mld, hn, (size_t)nb * mld);
    upl.copy(g.colmap + (size_t)b0 * GNSS_NCW, hc, (size_t)nb * GNSS_NCW);
    upl.copy(g.m + b0, hm, nb); upl.copy(g.nc + b0, hnc, nb);
    g.chi2_len = 0;
    if (o->chi2_table) {
        memcpy(hchi, o->chi2_table, 8 * (size_t)o->chi2_len);
        upl.copy(g.chi2, hchi, (size_t)o->chi2_len);
        g.chi2_len = o->chi2_len;
    }
    rc = upl.end();
    if (rc) return rc;
    g.gate_rows = o->gate_rows ? 1 : 0; g.strong = o->strong_reject ? 1 : 0;
    g.thr1 = o->gate_rows ? o->chi2_table[1] : __builtin_inf();
    if (!g.staged || m_cap > g.m_cap) g.m_cap = m_cap;
    int ncm = 0;
    for (int i = 0; i < nb; ++i) if (ncs[i] > ncm) ncm = ncs[i];
    if (!g.staged || ncm > g.nc_max) g.nc_max = ncm;
    g.in_frame = o->in_frame != 0;
    g.fused_last = false;
    g.results = false;
    if (g.in_frame && !g.W) {
        const size_t B = c->d.batch, ws = (size_t)c->ldp * 16;
        if (dalloc(c, &g.W, B * ws) | dalloc(c, &g.Yf, B * ws) | dalloc(c, &g.dxf, B * (size_t)c->ldp) | dalloc(c, &g.mf, B) | dalloc(c, &g.ncf, B)) return INGVIO_E_HIP;
        HIPCHK(c, hipStreamSynchronize(c->st));
    }
    g.staged = true;
    return INGVIO_OK;
}

// The staged GNSS update as its own pass over the covariance (ingvio_gnss_run, and ingvio_frame_run for an in-frame stage that
// cannot ride on the MSCKF write-back)
static int gnss_run_separate(ingvio_ctx* c, int b0, int nb, bool own_slots = false);

// raw GNSS epochs -> candidate rows, on the device (kernels_gnss.hip)
int ingvio_gnss_front_stage(ingvio_ctx* c, int b0, int nb, const ingvio_gnss_epoch* ep, const ingvio_gnss_opts* o)
{
    ENTER(c);
    static_assert(GNSS_FRONT_NCW == GNSS_NCW, "column stride of the staged rows");
    if (check_range(c, b0, nb) || !ep || !o) return INGVIO_E_ARG;
    if ((o->gate_rows || o->strong_reject) && (!o->chi2_table || o->chi2_len < 2 || o->chi2_len > CHI2_CAP)) return INGVIO_E_ARG;
    int smax = 1;
    std::vector<int> his(nb, 0);
    for (int i = 0; i < nb; ++i) {
        const ingvio_gnss_epoch& e = ep[i];
        if (e.n_sat < 0 || e.n_sat > INGVIO_GNSS_MAX_SAT || (e.n_sat && (!e.eph || !e.obs))) return INGVIO_E_ARG;
        if (2 * e.n_sat > c->mld || 2 * e.n_sat > 256) return INGVIO_E_CAPACITY;
        if (e.n_sat > smax) smax = e.n_sat;
        int hi = 0;
        const int idx[7] = { e.idx_se23 + 9, e.idx_yof + 1, e.idx_fs + 1, e.idx_cb[0] + 1, e.idx_cb[1] + 1, e.idx_cb[2] + 1, e.idx_cb[3] + 1 };
        if (e.n_sat) { if (e.idx_se23 < 0 || e.idx_yof < 0 || e.idx_fs < 0) return INGVIO_E_NOT_IN_STATE; for (int v : idx) { if (v > c->d.n_max) return INGVIO_E_NOT_IN_STATE; if (v > hi) hi = v; } }
        his[i] = hi;
        if (!ekf_core_fits(2 * e.n_sat ? 2 * e.n_sat : 1, 15)) return INGVIO_E_CAPACITY;
    }
    int rc = gnss_alloc(c);
    if (rc) return rc;
    auto& g = c->gn;
    const size_t B = c->d.batch;
    if (!g.front) {
        rc |= dalloc(c, &g.feph, B * INGVIO_GNSS_MAX_SAT * GE_N); rc |= dalloc(c, &g.fobs, B * INGVIO_GNSS_MAX_SAT * GO_N);
        rc |= dalloc(c, &g.frcv, B * GR_N); rc |= dalloc(c, &g.front, B * 64 * GF_N);
        if (rc) return INGVIO_E_HIP;
    }
    const size_t S = INGVIO_GNSS_MAX_SAT;
    Uploader upl{ c };
    rc = upl.begin(pad64(8 * (size_t)nb * S * GE_N) + pad64(8 * (size_t)nb * S * GO_N) + pad64(8 * (size_t)nb * GR_N) + pad64(8 * CHI2_CAP) + 1024);
    if (rc) return rc;
    double* he = upl.take<double>((size_t)nb * S * GE_N); double* ho = upl.take<double>((size_t)nb * S * GO_N);
    double* hr = upl.take<double>((size_t)nb * GR_N); double* hchi = upl.take<double>(CHI2_CAP);
    parallel_for(nb, [=](int i) {
        const ingvio_gnss_epoch& e = ep[i];
        memcpy(he + (size_t)i * S * GE_N, e.eph, 8 * (size_t)e.n_sat * GE_N);
        memcpy(ho + (size_t)i * S * GO_N, e.obs, 8 * (size_t)e.n_sat * GO_N);
        double* r = hr + (size_t)i * GR_N;
        memset(r, 0, 8 * GR_N);
        r[GR_NSAT] = e.n_sat; r[GR_DOY] = e.doy; r[GR_HAVE_ION] = e.ion ? 1.0 : 0.0;
        if (e.ion) memcpy(r + GR_ION, e.ion, 64);
        memcpy(r + GR_PW, e.p_w, 24); memcpy(r + GR_VW, e.v_w, 24); memcpy(r + GR_CB, e.cb, 32);
        r[GR_FS] = e.fs; r[GR_YAW] = e.yaw_offset;
        memcpy(r + GR_RENU, e.R_enu2ecef, 72); memcpy(r + GR_ANCHOR, e.anchor_ecef, 24);
        r[GR_IDX_SE23] = e.idx_se23; r[GR_IDX_YOF] = e.idx_yof; r[GR_IDX_FS] = e.idx_fs;
        for (int s = 0; s < 4; ++s) r[GR_IDX_CB + s] = e.idx_cb[s];
        r[GR_PSR_AMP] = e.psr_noise_amp; r[GR_DOPP_AMP] = e.dopp_noise_amp;
    });
    upl.copy(g.feph + (size_t)b0 * S * GE_N, he, (size_t)nb * S * GE_N);
    upl.copy(g.fobs + (size_t)b0 * S * GO_N, ho, (size_t)nb * S * GO_N);
    upl.copy(g.frcv + (size_t)b0 * GR_N, hr, (size_t)nb * GR_N);
    g.chi2_len = 0;
    if (o->chi2_table) {
        memcpy(hchi, o->chi2_table, 8 * (size_t)o->chi2_len);
        upl.copy(g.chi2, hchi, (size_t)o->chi2_len);
        g.chi2_len = o->chi2_len;
    }
    rc = upl.end();
    if (rc) return rc;
    const size_t mld = c->mld, hs = mld * GNSS_NCW;
    GnssFrontLaunch L;
    L.eph = g.feph + (size_t)b0 * S * GE_N; L.obs = g.fobs + (size_t)b0 * S * GO_N; L.rcv = g.frcv + (size_t)b0 * GR_N; L.smax = (int)S;
    L.front = g.front + (size_t)b0 * 64 * GF_N;
    L.H = g.H + (size_t)b0 * hs; L.res = g.res + (size_t)b0 * mld; L.noise = g.noise + (size_t)b0 * mld;
    L.m = g.m + b0; L.nc = g.nc + b0; L.colmap = g.colmap + (size_t)b0 * GNSS_NCW; L.mld = c->mld; L.hstride = (int)hs;
    launch_gnss_front(L, nb, c->st);
    for (int i = 0; i < nb; ++i) g.hi[b0 + i] = his[i];
    g.gate_rows = o->gate_rows ? 1 : 0; g.strong = o->strong_reject ? 1 : 0;
    g.thr1 = o->gate_rows ? o->chi2_table[1] : __builtin_inf();
    const int m_cap = 2 * smax;
    if (!g.staged || m_cap > g.m_cap) g.m_cap = m_cap;
    // the front builds the reference's var_order [SE23, YOF, <= 4 clocks, FS] = at most 15 columns; a front-staged update is applied
    // by ingvio_gnss_run (the flags of an earlier in-frame ingvio_gnss_stage must not leak into this stage)
    if (o->in_frame) { c->err = "ingvio_gnss_front_stage: in_frame is not supported (stage the rows with ingvio_gnss_stage)"; return INGVIO_E_UNSUPPORTED; }
    g.in_frame = false; g.nc_max = 15; g.fused_last = false; g.results = false;
    g.staged = true;
    return last_launch(c);
}

// gnss_comm::sat_states + psr_res + dopp_res for an arbitrary list of epochs, detached from the filters of the context and from
// the staged update rows: what GvioAligner::batchAlign (GvioAligner.cpp:88-383) and gnss_comm::psr_pos (gnss_spp.cpp:148-254)
// iterate on.  One launch for all epochs, temporary device buffers, nothing of the context's state is touched.
int ingvio_gnss_sat_eval(ingvio_ctx* c, int n_epochs, const ingvio_gnss_epoch* ep, double* out)
{
    ENTER(c);
    if (!c || n_epochs < 1 || !ep || !out) return INGVIO_E_ARG;
    const size_t S = INGVIO_GNSS_MAX_SAT;
    for (int i = 0; i < n_epochs; ++i) {
        const ingvio_gnss_epoch& e = ep[i];
        if (e.n_sat < 0 || e.n_sat > INGVIO_GNSS_MAX_SAT) return INGVIO_E_CAPACITY;
        if (e.n_sat && (!e.eph || !e.obs)) return INGVIO_E_ARG;
    }
    std::vector<double> he((size_t)n_epochs * S * GE_N, 0.0), ho((size_t)n_epochs * S * GO_N, 0.0), hr((size_t)n_epochs * GR_N, 0.0);
    for (int i = 0; i < n_epochs; ++i) {
        const ingvio_gnss_epoch& e = ep[i];
        if (e.n_sat) {
            memcpy(he.data() + (size_t)i * S * GE_N, e.eph, 8 * (size_t)e.n_sat * GE_N);
            memcpy(ho.data() + (size_t)i * S * GO_N, e.obs, 8 * (size_t)e.n_sat * GO_N);
        }
        double* r = hr.data() + (size_t)i * GR_N;
        r[GR_NSAT] = e.n_sat; r[GR_DOY] = e.doy; r[GR_HAVE_ION] = e.ion ? 1.0 : 0.0;
        if (e.ion) memcpy(r + GR_ION, e.ion, 64);
        memcpy(r + GR_PW, e.p_w, 24); memcpy(r + GR_VW, e.v_w, 24); memcpy(r + GR_CB, e.cb, 32);
        r[GR_FS] = e.fs; r[GR_YAW] = e.yaw_offset;
        memcpy(r + GR_RENU, e.R_enu2ecef, 72); memcpy(r + GR_ANCHOR, e.anchor_ecef, 24);
        for (int s4 = 0; s4 < 4; ++s4) r[GR_IDX_CB + s4] = -1.0;
        r[GR_PSR_AMP] = e.psr_noise_amp; r[GR_DOPP_AMP] = e.dopp_noise_amp;
    }
    const size_t S64 = 64;
    if (n_epochs > c->se.cap) {
        HIPCHK(c, hipStreamSynchronize(c->st));
        for (double** q : { &c->se.e, &c->se.o, &c->se.r, &c->se.f }) { if (*q) hipFree(*q); *q = nullptr; }
        c->se.cap = 0;
        const int cap = std::max(n_epochs, 32);
        if (hipMalloc((void**)&c->se.e, 8 * (size_t)cap * S * GE_N) != hipSuccess || hipMalloc((void**)&c->se.o, 8 * (size_t)cap * S * GO_N) != hipSuccess ||
            hipMalloc((void**)&c->se.r, 8 * (size_t)cap * GR_N) != hipSuccess || hipMalloc((void**)&c->se.f, 8 * (size_t)cap * S64 * GF_N) != hipSuccess) {
            for (double** q : { &c->se.e, &c->se.o, &c->se.r, &c->se.f }) { if (*q) hipFree(*q); *q = nullptr; }
            c->err = "hipMalloc failed (ingvio_gnss_sat_eval workspace)";
            return INGVIO_E_HIP;
        }
        c->se.cap = cap;
    }
    double *de = c->se.e, *dob = c->se.o, *dr = c->se.r, *df = c->se.f;
    int rc = up(c, de, he.data(), 8 * he.size()) | up(c, dob, ho.data(), 8 * ho.size()) | up(c, dr, hr.data(), 8 * hr.size());
    if (!rc) {
        GnssFrontLaunch L;
        memset(&L, 0, sizeof L);
        L.eph = de; L.obs = dob; L.rcv = dr; L.smax = (int)S; L.front = df;      // L.H == nullptr: no candidate rows
        launch_gnss_front(L, n_epochs, c->st);
        rc = down_sync(c, out, df, 8 * (size_t)n_epochs * 64 * GF_N);           // synchronises: the host vectors above may go
    }
    else hipStreamSynchronize(c->st);
    return rc ? INGVIO_E_HIP : last_launch(c);
}

int ingvio_gnss_front_fetch(ingvio_ctx* c, int b0, int nb, double* out)
{
    ENTER(c);
    if (check_range(c, b0, nb) || !out || !c->gn.front) return INGVIO_E_ARG;
    if (down_sync(c, out, c->gn.front + (size_t)b0 * 64 * GF_N, 8 * (size_t)nb * 64 * GF_N)) return INGVIO_E_HIP;
    return last_launch(c);
}

int ingvio_gnss_run(ingvio_ctx* c, int b0, int nb)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !c->gn.staged) return INGVIO_E_ARG;
    if (c->gn.in_frame) { c->err = "the staged GNSS update is in-frame: ingvio_frame_run applies it"; return INGVIO_E_ARG; }
    return gnss_run_separate(c, b0, nb);
}

// own_slots: called by ingvio_frame_run for an in-frame stage - row count, column count and dx go to the stage's own buffers (the
// frame's slots keep the MSCKF update's results for ingvio_frame_fetch) and the frame's status bits stay
static int gnss_run_separate(ingvio_ctx* c, int b0, int nb, bool own_slots)
{
    auto& g = c->gn;
    g.fused_last = own_slots;
    g.results = true;
    int n_cap = 0;
    for (int i = 0; i < nb; ++i) {
        if (g.hi[b0 + i] > c->h_n[b0 + i]) return INGVIO_E_NOT_IN_STATE;          // checkSubOrder against the LIVE state
        if (c->h_n[b0 + i] > n_cap) n_cap = c->h_n[b0 + i];
    }
    if (g.m_cap == 0) return INGVIO_OK;
    const size_t mld = c->mld, hs = mld * GNSS_NCW;
    if (!own_slots) HIPCHK(c, hipMemsetAsync(c->d_status + b0, 0, sizeof(int) * (size_t)nb, c->st));
    // after a fused frame step every filter's live covariance sits in the SECOND ping-pong half and this update (k_downdate) is in
    // place there: the first half still holds the prior up to the propagation strips, so a following ingvio_frame_run(restore_prior)
    // may keep restoring the strips only (0.03 instead of 0.10 ms per 512 filters)
    bool keep_strips = c->strip_ok && c->mut_seq == c->strip_seq;
    for (int i = 0; i < nb && keep_strips; ++i) if (c->h_cur[b0 + i] != 1) keep_strips = false;      // ... which only holds for filters that sit in the second half
    EkfLaunch E;
    memset(&E, 0, sizeof E);
    E.cv = view(c);
    // k_downdate below is strictly in place in the live half (launch_downdate never writes the other one): the strips stay valid
    if (keep_strips) c->strip_seq = c->mut_seq;
    E.b0 = b0; E.nb = nb; E.H = c->d_H + (size_t)b0 * hs; E.res = c->d_res + (size_t)b0 * mld;
    E.colmap = c->d_colmap + (size_t)b0 * GNSS_NCW; E.m = (own_slots ? g.mf : c->d_m) + b0; E.nc = (own_slots ? g.ncf : c->d_nc) + b0;
    E.noise = c->d_noiseB + (size_t)b0 * mld; E.r_kind = INGVIO_R_DIAG; E.mld = c->mld; E.hstride = (int)hs; E.cstride = GNSS_NCW;
    E.nstride = c->mld; E.Y = c->d_Y + (size_t)b0 * c->ystride; E.ystride = c->ystride; E.dx = own_slots ? g.dxf : c->d_dx;
    E.status = c->d_status; E.m_cap = g.m_cap; E.nc_cap = GNSS_NCW;
    if (g.strong) { E.chi2 = g.chi2; E.chi2_len = g.chi2_len; E.gate_max_rows = 14; }      // GnssUpdate.cpp:286
    RowsGateIn in{ g.H + (size_t)b0 * hs, g.res + (size_t)b0 * mld, g.noise + (size_t)b0 * mld, g.m + b0, g.colmap + (size_t)b0 * GNSS_NCW,
                   g.nc + b0, (int)hs, GNSS_NCW };
    {
        ProfScope p(c, PF_ROWGATE);
        if (launch_rows_gate(E, in, g.thr1, g.gamma + (size_t)b0 * mld, g.keep + (size_t)b0 * mld, c->st)) return INGVIO_E_CAPACITY;
    }
    { ProfScope p(c, PF_EKF_CORE); launch_ekf_core(E, c->st); }
    { ProfScope p(c, PF_DOWNDATE); launch_downdate(E, n_cap, c->st); }
    return last_launch(c);
}

int ingvio_gnss_fetch(ingvio_ctx* c, int b0, int nb, double* dx_out, int* rows_out, int* keep_out, double* gamma_out, int* status_out)
{
    ENTER(c);
    if (check_range(c, b0, nb)) return INGVIO_E_ARG;
    if (!c->gn.results) { c->err = "ingvio_gnss_fetch: no GNSS update has run on the staged rows (an in-frame stage is applied by ingvio_frame_run)"; return INGVIO_E_ARG; }
    const size_t mld = c->mld;
    std::vector<int> status(nb, 0);
    const bool fz = c->gn.fused_last;             // the in-frame update kept its results apart from the frame's
    if (dx_out) HIPCHK(c, hipMemcpyAsync(dx_out, (fz ? c->gn.dxf : c->d_dx) + (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp, hipMemcpyDeviceToHost, c->st));
    if (rows_out) HIPCHK(c, hipMemcpyAsync(rows_out, (fz ? c->gn.mf : c->d_m) + b0, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, c->st));
    if (keep_out) HIPCHK(c, hipMemcpyAsync(keep_out, c->gn.keep + (size_t)b0 * mld, sizeof(int) * (size_t)nb * mld, hipMemcpyDeviceToHost, c->st));
    if (gamma_out) HIPCHK(c, hipMemcpyAsync(gamma_out, c->gn.gamma + (size_t)b0 * mld, 8 * (size_t)nb * mld, hipMemcpyDeviceToHost, c->st));
    if (down_sync(c, status.data(), c->d_status + b0, sizeof(int) * (size_t)nb)) return INGVIO_E_HIP;
    int rc = last_launch(c);
    if (rc) return rc;
    int soft = INGVIO_OK;
    for (int i = 0; i < nb; ++i) {
        const int st = (status[i] & 8) ? INGVIO_REJECTED : ((status[i] & 4) ? INGVIO_E_NOT_PD : ((status[i] & 2) ? INGVIO_NEG_DIAG : INGVIO_OK));
        if (status_out) status_out[i] = st;
        if (st == INGVIO_NEG_DIAG) soft = st;
    }
    return soft;
}

int ingvio_gnss_update_batch(ingvio_ctx* c, int b0, int nb, const ingvio_update_block* blk, const ingvio_gnss_opts* o, double* dx_out,
                             int* rows_out, int* keep_out, int* status_out)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    int rc = ingvio_gnss_stage(c, b0, nb, blk, o);
    if (rc) return rc;
    rc = ingvio_gnss_run(c, b0, nb);
    if (rc) return rc;
    return ingvio_gnss_fetch(c, b0, nb, dx_out, rows_out, keep_out, nullptr, status_out);
}

int ingvio_mld(ingvio_ctx* c) { return c ? c->mld : 0; }

int ingvio_chi2_gamma(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, const double* H, int ldh, int m,
                      const double* res, const double* R, int r_kind, double* gamma)
{
    ENTER(c);
    if (!gamma) return INGVIO_E_ARG;
    int nc = 0;
    int rc = stage_generic(c, b, vidx, vsize, k, H, ldh, m, res, R, r_kind, &nc);
    if (rc) return rc;
    if ((size_t)8 * ((size_t)nc * m + (size_t)(m + 1) * (m + 1)) > 150 * 1024) return INGVIO_E_CAPACITY;
    launch_gamma(view(c), b, c->d_H + (size_t)b * c->hstride, c->d_res + (size_t)b * c->mld, c->d_colmap + (size_t)b * c->cstride,
                 m, nc, c->d_noise1, r_kind, c->mld, c->d_gamma + (size_t)b * c->d.f_max, c->st);
    if (down_sync(c, gamma, c->d_gamma + (size_t)b * c->d.f_max, 8)) return INGVIO_E_HIP;
    return last_launch(c);
}

// Many whitenResidual gates against the same prior in ONE launch and one synchronisation (all SLAM landmarks of a frame,
// LandmarkUpdate.cpp:98-99; the per-row GNSS gates, GnssUpdate.cpp:190,259): gamma_out[g] for block g, R = noise_var * I.
int ingvio_chi2_gamma_multi(ingvio_ctx* c, int b, int nblk, const ingvio_gate_block* blk, double noise_var, double* gamma_out)
{
    ENTER(c);
    if (check_range(c, b, 1) || nblk < 0 || (nblk && (!blk || !gamma_out))) return INGVIO_E_ARG;
    if (nblk == 0) return INGVIO_OK;
    size_t nd = 1, ni = 0, lds = 0;                         // doubles (slot 0 = noise), ints
    for (int g = 0; g < nblk; ++g) {
        const ingvio_gate_block& q = blk[g];
        if (!q.vidx || !q.vsize || !q.H || !q.res || q.k < 1 || q.m < 1 || q.ldh < q.m) return INGVIO_E_ARG;
        int nc = 0;
        for (int i = 0; i < q.k; ++i) {
            if (q.vidx[i] < 0 || q.vidx[i] + q.vsize[i] > c->h_n[b]) return INGVIO_E_NOT_IN_STATE;
            nc += q.vsize[i];
        }
        nd += (size_t)q.m * nc + q.m; ni += nc;
        const size_t l = 8 * ((size_t)nc * q.m + (size_t)(q.m + 1) * (q.m + 1));
        if (l > lds) lds = l;
    }
    if (lds > 150 * 1024) return INGVIO_E_CAPACITY;
    const size_t bytes_d = pad64(8 * nd), bytes_i = pad64(4 * ni), bytes_desc = pad64(4 * 5 * (size_t)nblk), bytes_g = pad64(8 * (size_t)nblk);
    const size_t total = bytes_d + bytes_i + bytes_desc + bytes_g;
    if (c->multi_cap < total) {
        HIPCHK(c, hipStreamSynchronize(c->st));
        if (c->d_multi) hipFree(c->d_multi);
        c->d_multi = nullptr; c->multi_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_multi, total * 2));
        c->multi_cap = total * 2;
    }
    Uploader upl{ c };
    int rc = upl.begin(total + 1024);
    if (rc) return rc;
    double* hd = upl.take<double>(bytes_d / 8); int* hi = upl.take<int>(bytes_i / 4); int* hdesc = upl.take<int>(bytes_desc / 4);
    hd[0] = noise_var;
    size_t od = 1, oi = 0;
    for (int g = 0; g < nblk; ++g) {
        const ingvio_gate_block& q = blk[g];
        int nc = 0;
        int* cm = hi + oi;
        for (int i = 0; i < q.k; ++i) for (int j = 0; j < q.vsize[i]; ++j) cm[nc++] = q.vidx[i] + j;
        hdesc[5 * g] = (int)od;
        for (int cc = 0; cc < nc; ++cc) memcpy(hd + od + (size_t)cc * q.m, q.H + (size_t)cc * q.ldh, 8 * (size_t)q.m);
        od += (size_t)q.m * nc;
        hdesc[5 * g + 1] = (int)od;
        memcpy(hd + od, q.res, 8 * (size_t)q.m);
        od += q.m;
        hdesc[5 * g + 2] = (int)oi; hdesc[5 * g + 3] = q.m; hdesc[5 * g + 4] = nc;
        oi += nc;
    }
    double* dd = reinterpret_cast<double*>(c->d_multi);
    int* di = reinterpret_cast<int*>(c->d_multi + bytes_d);
    int* ddesc = reinterpret_cast<int*>(c->d_multi + bytes_d + bytes_i);
    double* dg = reinterpret_cast<double*>(c->d_multi + bytes_d + bytes_i + bytes_desc);
    upl.copy(dd, hd, nd); upl.copy(di, hi, ni); upl.copy(ddesc, hdesc, 5 * (size_t)nblk);
    rc = upl.end();
    if (rc) return rc;
    launch_gamma_multi(view(c), b, nblk, dd, di, ddesc, dd, dg, lds, c->st);
    if (down_sync(c, gamma_out, dg, 8 * (size_t)nblk)) return INGVIO_E_HIP;
    return last_launch(c);
}

// tri != nullptr: the points of the staged features are triangulated on the device first (ingvio_msckf_update_tri)
static int msckf_update_impl(ingvio_ctx* c, int b0, int nb, const ingvio_msckf_frame* frames, const ingvio_msckf_opts* opts, const ingvio_tri_opts* tri,
                             const unsigned long long* const* tri_masks, double* dx_out, int* accepted, double* gamma, int* rows_out, double* pf_out,
                             int* tri_ok)
{
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !frames || !opts) return INGVIO_E_ARG;
    if (tri && (tri->outer_loop_max_iter < 0 || tri->inner_loop_max_iter < 0)) return INGVIO_E_ARG;
    MsckfOpts op;
    c->fork_recorded = false;
    HIPCHK(c, hipMemsetAsync(c->d_status + b0, 0, sizeof(int) * (size_t)nb, c->st));      // first: it runs while the host packs the frame
    int rc = make_opts(c, opts, &op);
    if (rc) return rc;
    int fmx = 0;
    rc = stage_frames(c, b0, nb, frames, &fmx);
    if (rc) return rc;
    rc = fill_noise_scalar(c, b0, nb, op.var);
    if (rc) return rc;
    const int fm = c->d.f_max;
    //  fork_recorded must not outlive this call: an early error return between here and run_msckf_factored - the only
    // place that consumes the flag - would make the NEXT large-window step skip recording ev_fork and start its side stream
    // against a stale event
    struct ForkGuard { ingvio_ctx* c; ~ForkGuard() { c->fork_recorded = false; } } fork_guard{ c };
    if (tri && c->d.c_max > 16 && c->st2 && c->method == 1) {      // the prior-only half of the large-window solve may start now, beside the triangulation
        HIPCHK(c, hipEventRecord(c->ev_fork, c->st));
        c->fork_recorded = true;
    }
    if (tri) {
        TriLaunch T;
        memset(&T, 0, sizeof T);
        T.fv = fview(c); T.b0 = b0;
        memcpy(T.R_lr, tri->R_cl2cr, 72); memcpy(T.t_lr, tri->t_cl2cr, 24);
        T.trans_thres = tri->trans_thres; T.huber_epsilon = tri->huber_epsilon; T.conv_precision = tri->conv_precision;
        T.init_damping = tri->init_damping; T.max_depth = tri->max_depth; T.min_depth = tri->min_depth;
        T.outer_loop_max_iter = tri->outer_loop_max_iter; T.inner_loop_max_iter = tri->inner_loop_max_iter;
        T.pf = c->d_pf; T.ok = c->d_tri_ok; T.mask_failed = 1; T.mask_rw = c->d_mask; T.check_anchor = 1;
        T.ok2 = (int*)(c->d_result_slab + c->ro_tok); T.pf2 = (double*)(c->d_result_slab + c->ro_tpf);
        if (tri_masks) {                                       // triangulation masks that differ from the update's: their own device array
            if (!c->d_tri_mask && dalloc(c, &c->d_tri_mask, (size_t)c->d.batch * fm)) return INGVIO_E_HIP;
            std::vector<unsigned long long> tm((size_t)nb * fm, 0ULL);
            for (int i = 0; i < nb; ++i) {
                const unsigned long long* src = tri_masks[i] ? tri_masks[i] : frames[i].obs_mask;
                const int C = frames[i].n_clones;
                const unsigned long long cmask = C >= 64 ? ~0ULL : ((1ULL << C) - 1ULL);
                for (int j = 0; j < frames[i].n_feat; ++j) tm[(size_t)i * fm + j] = src[j] & cmask;
            }
            const UpItem item = { c->d_tri_mask + (size_t)b0 * fm, tm.data(), 8 * tm.size() };
            if (stage_small(c, &item, 1)) return INGVIO_E_HIP;
            T.tri_mask = c->d_tri_mask;
        }
        if (launch_triangulate(T, nb, fmx > 0 ? fmx : 1, tri->stereo, c->st)) return INGVIO_E_UNSUPPORTED;
    }
    rc = run_msckf(c, b0, nb, op, opts->stereo, fmx);
    if (rc) return rc;
    std::vector<int> rows(nb), status(nb);
    if (b0 == 0 && nb == c->d.batch && (tri ? c->result_bytes : c->ro_tok) <= (1u << 20)) {      // the whole batch, small: one copy of the result slab (gated on the bytes this call copies)
        HIPCHK(c, hipMemcpyAsync(c->h_result, c->d_result_slab, tri ? c->result_bytes : c->ro_tok, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        if (dx_out) memcpy(dx_out, c->h_result, 8 * (size_t)nb * c->ldp);
        if (gamma) memcpy(gamma, c->h_result + c->ro_gam, 8 * (size_t)nb * fm);
        if (accepted) memcpy(accepted, c->h_result + c->ro_used, sizeof(int) * (size_t)nb * fm);
        memcpy(rows.data(), c->h_result + c->ro_m, sizeof(int) * (size_t)nb);
        memcpy(status.data(), c->h_result + c->ro_st, sizeof(int) * (size_t)nb);
        if (tri && tri_ok) memcpy(tri_ok, c->h_result + c->ro_tok, sizeof(int) * (size_t)nb * fm);
        if (tri && pf_out) memcpy(pf_out, c->h_result + c->ro_tpf, 8 * (size_t)nb * fm * 3);
    } else {
        if (tri && tri_ok) HIPCHK(c, hipMemcpyAsync(tri_ok, c->d_result_slab + c->ro_tok + sizeof(int) * (size_t)b0 * fm, sizeof(int) * (size_t)nb * fm, hipMemcpyDeviceToHost, c->st));
        if (tri && pf_out) HIPCHK(c, hipMemcpyAsync(pf_out, c->d_result_slab + c->ro_tpf + 8 * (size_t)b0 * fm * 3, 8 * (size_t)nb * fm * 3, hipMemcpyDeviceToHost, c->st));
        if (dx_out) HIPCHK(c, hipMemcpyAsync(dx_out, c->d_dx + (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp, hipMemcpyDeviceToHost, c->st));
        if (accepted) HIPCHK(c, hipMemcpyAsync(accepted, c->d_used + (size_t)b0 * fm, sizeof(int) * (size_t)nb * fm, hipMemcpyDeviceToHost, c->st));
        if (gamma) HIPCHK(c, hipMemcpyAsync(gamma, c->d_gamma + (size_t)b0 * fm, 8 * (size_t)nb * fm, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(rows.data(), c->d_m + b0, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(status.data(), c->d_status + b0, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
    }
    rc = last_launch(c);
    if (rc) return rc;
    int soft = INGVIO_OK;
    for (int i = 0; i < nb; ++i) {
        if (rows_out) rows_out[i] = rows[i];
        if (rows[i] == 0 && soft == INGVIO_OK) soft = INGVIO_NO_ROWS;
        if (status[i] & 2) soft = INGVIO_NEG_DIAG;
    }
    return nb == 1 ? soft : (soft == INGVIO_NEG_DIAG ? soft : INGVIO_OK);
}

int ingvio_msckf_update(ingvio_ctx* c, int b0, int nb, const ingvio_msckf_frame* frames, const ingvio_msckf_opts* opts,
                        double* dx_out, int* accepted, double* gamma, int* rows_out)
{
    ENTER(c);
    return msckf_update_impl(c, b0, nb, frames, opts, nullptr, nullptr, dx_out, accepted, gamma, rows_out, nullptr, nullptr);
}

int ingvio_msckf_update_tri(ingvio_ctx* c, int b0, int nb, const ingvio_msckf_frame* frames, const ingvio_msckf_opts* opts,
                            const ingvio_tri_opts* tri, const unsigned long long* const* tri_masks, double* dx_out, int* accepted,
                            double* gamma, int* rows_out, double* pf_out, int* tri_ok)
{
    ENTER(c);
    if (!tri) return INGVIO_E_ARG;
    return msckf_update_impl(c, b0, nb, frames, opts, tri, tri_masks, dx_out, accepted, gamma, rows_out, pf_out, tri_ok);
}

// ---- SLAM-landmark covariance operations -------------------------------------------------------
static int stage_hnew(ingvio_ctx* c, const double* H_new, int ldn, int m, int s)
{
    if (!H_new || s < 1 || s > 6 || ldn < m || m > c->mld) return INGVIO_E_ARG;
    std::vector<double> Hn((size_t)c->mld * s, 0.0);
    for (int j = 0; j < s; ++j) memcpy(&Hn[(size_t)j * c->mld], H_new + (size_t)j * ldn, 8 * (size_t)m);
    if (up(c, c->d_hnew, Hn.data(), 8 * Hn.size())) return INGVIO_E_HIP;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

int ingvio_add_variable_delayed_invertible(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, const double* H_old, int ldh,
                                           const double* H_new, int ldn, int s, double noise, int* new_idx)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b, 1)) return INGVIO_E_ARG;
    if (c->h_n[b] + s > c->d.n_max) return INGVIO_E_CAPACITY;
    int nc = 0;
    const double var = noise * noise;
    std::vector<double> zero((size_t)(s > 0 ? s : 1), 0.0);
    int rc = stage_generic(c, b, vidx, vsize, k, H_old, ldh, s, zero.data(), &var, 0, &nc);
    if (rc) return rc;
    rc = stage_hnew(c, H_new, ldn, s, s);
    if (rc) return rc;
    if (launch_delayed_add(view(c), b, c->d_H + (size_t)b * c->hstride, c->d_hnew, c->d_colmap + (size_t)b * c->cstride, s, nc, c->mld,
                           var, c->d_Y + (size_t)b * c->ystride, c->st)) return INGVIO_E_CAPACITY;
    if (new_idx) *new_idx = c->h_n[b];
    c->h_n[b] += s;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return last_launch(c);
}

int ingvio_add_variable_delayed(ingvio_ctx* c, int b, const int* vidx, const int* vsize, int k, const double* H_old, int ldh,
                                const double* H_new, int ldn, int m, int s, const double* res, double noise, double chi2_mult,
                                int do_chi2, double chi2_check, double* dx_out, int* added, int* new_idx, double* chi2_out)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b, 1) || !added) return INGVIO_E_ARG;
    *added = 0;
    if (m <= s) return INGVIO_OK;                                                  // StateManager.cpp:571-575
    if (c->h_n[b] + s > c->d.n_max) return INGVIO_E_CAPACITY;
    int nc = 0;
    const double var = noise * noise;
    int rc = stage_generic(c, b, vidx, vsize, k, H_old, ldh, m, res, &var, 0, &nc);
    if (rc) return rc;
    rc = stage_hnew(c, H_new, ldn, m, s);
    if (rc) return rc;
    if (!ekf_core_fits(m - s, nc)) return INGVIO_E_CAPACITY;     // the trailing EKF update must fit before the state is touched
    double* dH = c->d_H + (size_t)b * c->hstride;
    double* dres = c->d_res + (size_t)b * c->mld;
    const int* dcm = c->d_colmap + (size_t)b * c->cstride;
    if (launch_delayed_qr(dH, dres, c->d_hnew, m, s, nc, c->mld, c->st)) return INGVIO_E_CAPACITY;      // :577-589
    const int mu = m - s;
    if ((size_t)8 * ((size_t)nc * mu + (size_t)(mu + 1) * (mu + 1)) > 150 * 1024) return INGVIO_E_CAPACITY;
    launch_gamma(view(c), b, dH + s, dres + s, dcm, mu, nc, c->d_noise1, 0, c->mld, c->d_gamma + (size_t)b * c->d.f_max, c->st);   // :601-608
    double chi2 = 0.0;
    if (down_sync(c, &chi2, c->d_gamma + (size_t)b * c->d.f_max, 8)) return INGVIO_E_HIP;
    if (chi2_out) *chi2_out = chi2;
    if (chi2 > chi2_mult * chi2_check && do_chi2) return last_launch(c);           // :614-618
    if (launch_delayed_add(view(c), b, dH, c->d_hnew, dcm, s, nc, c->mld, var, c->d_Y + (size_t)b * c->ystride, c->st))
        return INGVIO_E_CAPACITY;
    if (new_idx) *new_idx = c->h_n[b];
    c->h_n[b] += s;
    *added = 1;
    // :623-624 ekfUpdate with the lower rows on the extended state
    int zero = 0;
    if (up(c, c->d_status + b, &zero, sizeof(int)) || up(c, c->d_m + b, &mu, sizeof(int))) return INGVIO_E_HIP;
    EkfLaunch E;
    memset(&E, 0, sizeof E);
    E.cv = view(c); E.b0 = b; E.nb = 1; E.H = dH + s; E.res = dres + s; E.colmap = dcm; E.m = c->d_m + b; E.nc = c->d_nc + b;
    E.noise = c->d_noise1; E.r_kind = 0; E.mld = c->mld; E.hstride = c->hstride; E.cstride = c->cstride;
    E.nstride = c->mld * c->mld; E.Y = c->d_Y + (size_t)b * c->ystride; E.ystride = c->ystride; E.dx = c->d_dx;
    E.status = c->d_status; E.m_cap = mu; E.nc_cap = nc;
    launch_ekf_core(E, c->st);
    launch_downdate(E, c->h_n[b], c->st);
    int status = 0;
    if (dx_out && down_sync(c, dx_out, c->d_dx + (size_t)b * c->ldp, 8 * (size_t)c->h_n[b])) return INGVIO_E_HIP;
    if (down_sync(c, &status, c->d_status + b, sizeof(int))) return INGVIO_E_HIP;
    rc = last_launch(c);
    if (rc) return rc;
    return (status & 4) ? INGVIO_E_NOT_PD : ((status & 2) ? INGVIO_NEG_DIAG : INGVIO_OK);
}

int ingvio_replace_var_linear(ingvio_ctx* c, int b, int tidx, int tsize, const int* vidx, const int* vsize, int k, const double* H, int ldh)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b, 1) || tsize < 1 || tsize > 6) return INGVIO_E_ARG;
    if (tidx < 0 || tidx + tsize > c->h_n[b]) return INGVIO_E_NOT_IN_STATE;       // "Target var not in state" (:653-657)
    int nc = 0;
    const double one = 1.0;
    std::vector<double> zero((size_t)tsize, 0.0);
    int rc = stage_generic(c, b, vidx, vsize, k, H, ldh, tsize, zero.data(), &one, 0, &nc);
    if (rc) return rc;
    if (launch_replace_var(view(c), b, c->d_H + (size_t)b * c->hstride, c->d_colmap + (size_t)b * c->cstride, tidx, tsize, nc, c->mld,
                           c->d_Y + (size_t)b * c->ystride, c->st)) return INGVIO_E_CAPACITY;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return last_launch(c);
}

// Triangulator::triangulate{Mono,Stereo}Obs for every feature of the given (or staged) frames
int ingvio_triangulate(ingvio_ctx* c, int b0, int nb, const ingvio_msckf_frame* frames, const ingvio_tri_opts* o,
                       double* pf_out, int* ok_out)
{
    ENTER(c);
    if (check_range(c, b0, nb) || !o) return INGVIO_E_ARG;
    if (o->outer_loop_max_iter < 0 || o->inner_loop_max_iter < 0) return INGVIO_E_ARG;
    int fmx = c->d.f_max;
    if (frames) {
        const int rc = stage_frames(c, b0, nb, frames, &fmx, -1);      // poses and observations only: no state indices needed
        if (rc) return rc;
        c->strip_ok = false;
    } else if (!c->staged) return INGVIO_E_ARG;
    else if (wait_inputs(c)) return INGVIO_E_HIP;
    TriLaunch L;
    memset(&L, 0, sizeof L);
    L.fv = fview(c); L.b0 = b0;
    memcpy(L.R_lr, o->R_cl2cr, 72); memcpy(L.t_lr, o->t_cl2cr, 24);
    L.trans_thres = o->trans_thres; L.huber_epsilon = o->huber_epsilon; L.conv_precision = o->conv_precision;
    L.init_damping = o->init_damping; L.max_depth = o->max_depth; L.min_depth = o->min_depth;
    L.outer_loop_max_iter = o->outer_loop_max_iter; L.inner_loop_max_iter = o->inner_loop_max_iter;
    L.pf = c->d_pf; L.ok = c->d_tri_ok; L.mask_failed = o->mask_failed ? 1 : 0; L.mask_rw = c->d_mask;
    if (launch_triangulate(L, nb, fmx > 0 ? fmx : 1, o->stereo, c->st)) return INGVIO_E_UNSUPPORTED;
    const int fm = c->d.f_max;
    if (pf_out) HIPCHK(c, hipMemcpyAsync(pf_out, c->d_pf + (size_t)b0 * fm * 3, 8 * (size_t)nb * fm * 3, hipMemcpyDeviceToHost, c->st));
    if (ok_out) HIPCHK(c, hipMemcpyAsync(ok_out, c->d_tri_ok + (size_t)b0 * fm, sizeof(int) * (size_t)nb * fm, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return last_launch(c);
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense-H update: rows already in dw.Hd (by state column), residual in the carried row of dw.X, row counts in dw.m.
//   S = H P H^T (+ noise), Y = P H^T L^-T, z = L^-1 res by one Cholesky sweep with carried rows; dx = Y z; P -= Y Y^T.
// ---------------------------------------------------------------------------------------------------------------------
static int dense_ws_alloc(ingvio_ctx* c, int m_need)
{
    auto& w = c->dw;
    const int m_cap = (m_need + 31) / 32 * 32;
    if (w.Hd && w.m_cap >= m_cap) return 0;
    HIPCHK(c, hipStreamSynchronize(c->st));
    for (double** p : { &w.Hd, &w.X, &w.Y, &w.Tb, &w.U, &w.noise, &w.noiseB }) { if (*p) hipFree(*p); *p = nullptr; }
    if (w.m) { hipFree(w.m); w.m = nullptr; }
    if (w.cidx) { hipFree(w.cidx); w.cidx = nullptr; }
    if (w.rowmap) { hipFree(w.rowmap); w.rowmap = nullptr; }
    const int B = c->d.batch;
    w.m_cap = m_cap; w.n32 = (c->d.n_max + 31) / 32 * 32; w.n_ld = w.n32; w.ldx = m_cap + w.n32 + 32;
    w.hstride = std::max((size_t)m_cap * w.n_ld, (size_t)LM_MAX * 100);      // also holds the landmark path's compact blocks
    w.xstride = (size_t)w.ldx * m_cap;
    // launch_chol_sweep works with at least TWO T slots (kernels_chol.hip): a 32-row workspace used to be sized for one, and the last
    // filter's second slot lay past the allocation (found in round 4 when a change of the allocation order put unmapped memory there)
    w.tstride = (size_t)std::max(m_cap / 32, 2) * 1024 + (size_t)m_cap;
    w.ustride = m_cap <= 256 ? lm_chol_ws_doubles(m_cap) : 0;
    int rc = dalloc(c, &w.Hd, (size_t)B * w.hstride) | dalloc(c, &w.X, (size_t)B * w.xstride) | dalloc(c, &w.Y, (size_t)B * w.xstride)
           | dalloc(c, &w.Tb, (size_t)B * w.tstride) | (w.ustride ? dalloc(c, &w.U, (size_t)B * w.ustride) : 0) | dalloc(c, &w.m, (size_t)B) | dalloc(c, &w.noise, (size_t)m_cap * m_cap) | dalloc(c, &w.cidx, (size_t)B * LM_MAX * 4) | dalloc(c, &w.noiseB, (size_t)B * m_cap)
           | dalloc(c, &w.rowmap, (size_t)B * m_cap);
    return rc ? INGVIO_E_HIP : 0;
}

// noise: r_kind < 0 -> scalar variance `var` on the whole diagonal (the GEMM's epilogue); else d_noise (stride nstride) through k_add_noise
static int run_dense_update(ingvio_ctx* c, int b0, int nb, double var, int r_kind, const double* d_noise, int nstride, double* d_dx, bool products_done,
                            const int* rowmap, const int* marg_idx, int marg_size, bool* marg_fused)
{
    auto& w = c->dw;
    const int mc = w.m_cap, n_cap = c->d.n_max, B = c->d.batch;
    const size_t pp = c->pp;
    double *Hd = w.Hd + (size_t)b0 * w.hstride, *X = w.X + (size_t)b0 * w.xstride, *Y = w.Y + (size_t)b0 * w.xstride;
    const int* act = w.m + b0;
    if (!products_done) {
        ProfScope p(c, PF_LM_GEMM);
        GemmArgs g = {};
        // P H^T -> carried rows mc .. of X
        g.A = c->Pbase + (size_t)b0 * pp; g.sa = pp; g.lda = c->ldp; g.modeA = 0; g.a_sel = c->d_cur + b0; g.a_sel_stride = (size_t)B * pp;
        g.B = Hd; g.sb = w.hstride; g.ldb = w.n_ld; g.modeB = 1;
        g.C = X + mc; g.sc = w.xstride; g.rs = 1; g.cs = w.ldx;
        g.M = n_cap; g.N = mc; g.K = n_cap; g.m_lim = n_cap; g.n_lim = mc; g.ksplit = 1; g.active = act; g.batch = nb;
        launch_gemm(g, c->st);
        // S = H (P H^T), lower blocks
        g = GemmArgs{};
        g.A = Hd; g.sa = w.hstride; g.lda = w.n_ld; g.modeA = 1;
        g.B = X + mc; g.sb = w.xstride; g.ldb = w.ldx; g.modeB = 1;
        g.C = X; g.sc = w.xstride; g.rs = 1; g.cs = w.ldx;
        g.M = mc; g.N = mc; g.K = n_cap; g.m_lim = mc; g.n_lim = mc; g.ksplit = 1; g.lower = 1; g.diag_add = r_kind < 0 ? var : 0.0;
        g.active = act; g.batch = nb;
        launch_gemm(g, c->st);
        if (r_kind >= 0) launch_add_noise(X, w.xstride, w.ldx, d_noise, nstride, r_kind, act, mc, nb, c->st);
    }
    // S of up to 256 rows: one workgroup per filter, S in registers (kernels_lmchol.hip); larger: the sweep out of L2 (kernels_chol.hip)
#ifdef INGVIO_ALT_KERNELS
    static const bool sweep_only = getenv("INGVIO_LM_SOLVE") && !strcmp(getenv("INGVIO_LM_SOLVE"), "sweep");
#else
    constexpr bool sweep_only = false;
#endif
    if (w.U && !sweep_only) {
        ProfScope p(c, PF_LM_CHOL);
        LmCholArgs a = {};
        a.cv = view(c); a.b0 = b0; a.nb = nb; a.X = X; a.Y = Y; a.xs = w.xstride; a.ldx = w.ldx; a.mc = mc; a.res_row = mc + w.n32;
        a.U = w.U + (size_t)b0 * w.ustride; a.us = w.ustride; a.m = act; a.status = c->d_status + b0; a.fail_bit = 4; a.dx = d_dx;
        a.rowmap = rowmap; a.rm_stride = mc;
        launch_lm_chol(a, c->st);
    } else {
        ProfScope p(c, PF_LM_CHOL);
        CholArgs a = {};
        a.W = X; a.Y = Y; a.xs = w.xstride; a.ld = w.ldx; a.Tb = w.Tb + (size_t)b0 * w.tstride; a.ts = w.tstride; a.t_slots = mc / 32;
        a.rows = w.ldx; a.ncols = mc; a.status = c->d_status + b0; a.fail_bit = 4; a.active = act; a.batch = nb;
        launch_chol_sweep(a, c->st);
        launch_lm_finish(view(c), b0, nb, Y, w.xstride, w.ldx, mc, mc + w.n32, act, d_dx, c->d_status, c->st);
    }
    {
        ProfScope p(c, PF_DOWNDATE);
        EkfLaunch E;
        memset(&E, 0, sizeof E);
        E.cv = view(c); E.b0 = b0; E.nb = nb; E.Y = Y + mc; E.m = w.m + b0; E.status = c->d_status; E.m_cap = mc;
        const bool fused = launch_downdate(E, n_cap, c->st, nullptr, w.ldx, w.xstride, marg_idx, marg_size);
        if (marg_fused) *marg_fused = fused;
    }
    c->mut_seq++;
    return last_launch(c);
}

// marg_idx / marg_size (optional): the marginalisation that follows the update in the frame, fused into the write-back where the
// downdate kernel supports it (*marg_fused; the caller then only flips the halves: launch_post_marg)
static int landmark_update_launch(ingvio_ctx* c, int b0, int nb, const int* marg_idx = nullptr, int marg_size = 0, bool* marg_fused = nullptr)
{
    if (marg_fused) *marg_fused = false;
    auto& w = c->dw;
    auto& s = c->lm;
    const int* rowmap = nullptr;
    {
        ProfScope p(c, PF_LM_BUILD);
        LmBuild L;
        memset(&L, 0, sizeof L);
        L.cv = view(c); L.op = s.op; L.b0 = b0; L.nb = nb;
        L.lv.pose = s.pose; L.lv.idx = s.idx; L.lv.n_lm = s.n_lm; L.lv.lm_idx = s.lm_idx; L.lv.anchor_idx = s.anchor_idx;
        L.lv.pf = s.pf; L.lv.uv = s.uv; L.lv.tracked = s.tracked; L.lv.lmax = LM_MAX;
        L.Hd = w.Hd + (size_t)b0 * w.hstride; L.hstride = w.hstride; L.n_ld = w.n_ld; L.m_cap = w.m_cap;
        L.X = w.X + (size_t)b0 * w.xstride; L.xstride = w.xstride; L.ldx = w.ldx; L.res_row = w.m_cap + w.n32;
        L.gamma = s.gamma + (size_t)b0 * LM_MAX; L.accept = s.accept + (size_t)b0 * LM_MAX; L.m_out = w.m + b0; L.dx = s.dx;
        L.cidx = w.cidx + (size_t)b0 * LM_MAX * 4; L.n_rows = w.n32;
        // states of up to 256 rows with the register-resident solve: rows, products and gate in one kernel (k_lm_front), the accepted
        // rows handed on as a row map; otherwise k_lm_build + k_lm_products write the compacted system
#ifdef INGVIO_ALT_KERNELS
        static const bool split_front = getenv("INGVIO_LM_FRONT") && !strcmp(getenv("INGVIO_LM_FRONT"), "split");
        static const bool sweep_only = getenv("INGVIO_LM_SOLVE") && !strcmp(getenv("INGVIO_LM_SOLVE"), "sweep");
#else
        constexpr bool split_front = false, sweep_only = false;
#endif
        if (w.U && !split_front && !sweep_only && launch_lm_front(L, s.l_hi, w.rowmap + (size_t)b0 * w.m_cap, c->st))
            rowmap = w.rowmap + (size_t)b0 * w.m_cap;
        else launch_lm_build(L, c->st);
    }
    return run_dense_update(c, b0, nb, s.op.var, -1, nullptr, 0, s.dx, true, rowmap, marg_idx, marg_size, marg_fused);
}

int ingvio_landmark_stage(ingvio_ctx* c, int b0, int nb, const ingvio_landmark_frame* fr, const ingvio_landmark_opts* o)
{
    ENTER(c);
    if (check_range(c, b0, nb) || !fr || !o || !(o->noise > 0.0)) return INGVIO_E_ARG;
    auto& s = c->lm;
    int l_hi = 0;
    std::vector<int> hi_new(nb, 0);
    for (int i = 0; i < nb; ++i) {
        const auto& f = fr[i];
        if (f.n_lm < 0 || f.n_lm > LM_MAX) return INGVIO_E_CAPACITY;
        if (f.n_lm && (!f.lm_idx || !f.anchor_idx || !f.pf || !f.uv || !f.tracked)) return INGVIO_E_ARG;
        // every variable must lie inside the filter's LIVE state at update time (in_frame: the frame step appends the 6-column clone
        // first), not merely inside the buffer - rows beyond it would read stale covariance
        const int n_lim = std::min(c->d.n_max, c->h_n[b0 + i] + (o->in_frame ? 6 : 0));
        if (f.idx_epose < 0 || f.idx_epose + 9 > n_lim || f.idx_ext < 0 || f.idx_ext + 6 > n_lim) return INGVIO_E_NOT_IN_STATE;
        for (int l = 0; l < f.n_lm; ++l)
            if (f.lm_idx[l] < 0 || f.lm_idx[l] + 3 > n_lim || f.anchor_idx[l] < 0 || f.anchor_idx[l] + 6 > n_lim) return INGVIO_E_NOT_IN_STATE;
        int hi = f.n_lm ? std::max(f.idx_epose + 9, f.idx_ext + 6) : 0;
        for (int l = 0; l < f.n_lm; ++l) hi = std::max(hi, std::max(f.lm_idx[l] + 3, f.anchor_idx[l] + 6));
        hi_new[i] = hi;
        l_hi = std::max(l_hi, f.n_lm);
    }
    const int B = c->d.batch;
    if (!s.alloc) {
        int rc = dalloc(c, &s.pose, (size_t)B * 24) | dalloc(c, &s.pf, (size_t)B * LM_MAX * 3) | dalloc(c, &s.uv, (size_t)B * LM_MAX * 4)
               | dalloc(c, &s.gamma, (size_t)B * LM_MAX) | dalloc(c, &s.idx, (size_t)B * 2) | dalloc(c, &s.n_lm, (size_t)B)
               | dalloc(c, &s.lm_idx, (size_t)B * LM_MAX) | dalloc(c, &s.anchor_idx, (size_t)B * LM_MAX) | dalloc(c, &s.tracked, (size_t)B * LM_MAX)
               | dalloc(c, &s.accept, (size_t)B * LM_MAX) | dalloc(c, &s.dx, (size_t)B * c->ldp);
        if (rc) return INGVIO_E_HIP;
        s.alloc = true;
    }
    s.l_hi = std::max(s.l_hi, l_hi);
    if (int rc = dense_ws_alloc(c, std::max(32, 4 * s.l_hi))) return rc;
    Uploader upl{ c };
    if (int rc = upl.begin(pad64(8 * (size_t)nb * (24 + 7 * LM_MAX)) + pad64(4 * (size_t)nb * (3 + 3 * LM_MAX)) + 1024)) return rc;
    double* hp = upl.take<double>((size_t)nb * 24);
    double* hpf = upl.take<double>((size_t)nb * LM_MAX * 3);
    double* huv = upl.take<double>((size_t)nb * LM_MAX * 4);
    int* hidx = upl.take<int>((size_t)nb * 2);
    int* hn = upl.take<int>((size_t)nb);
    int* hl = upl.take<int>((size_t)nb * LM_MAX);
    int* ha = upl.take<int>((size_t)nb * LM_MAX);
    int* ht = upl.take<int>((size_t)nb * LM_MAX);
    for (int i = 0; i < nb; ++i) {
        const auto& f = fr[i];
        memcpy(hp + 24 * i, f.R_i2w, 72); memcpy(hp + 24 * i + 9, f.p_i2w, 24); memcpy(hp + 24 * i + 12, f.R_cl2i, 72); memcpy(hp + 24 * i + 21, f.p_c2i, 24);
        hidx[2 * i] = f.idx_epose; hidx[2 * i + 1] = f.idx_ext; hn[i] = f.n_lm;
        for (int l = 0; l < LM_MAX; ++l) {
            const bool on = l < f.n_lm;
            hl[i * LM_MAX + l] = on ? f.lm_idx[l] : -1; ha[i * LM_MAX + l] = on ? f.anchor_idx[l] : -1; ht[i * LM_MAX + l] = on ? f.tracked[l] : 0;
            for (int k = 0; k < 3; ++k) hpf[((size_t)i * LM_MAX + l) * 3 + k] = on ? f.pf[3 * l + k] : 0.0;
            for (int k = 0; k < 4; ++k) huv[((size_t)i * LM_MAX + l) * 4 + k] = on ? ((o->stereo || k < 2) ? f.uv[4 * l + k] : 0.0) : 0.0;
        }
    }
    upl.copy(s.pose + (size_t)b0 * 24, hp, (size_t)nb * 24);
    upl.copy(s.pf + (size_t)b0 * LM_MAX * 3, hpf, (size_t)nb * LM_MAX * 3);
    upl.copy(s.uv + (size_t)b0 * LM_MAX * 4, huv, (size_t)nb * LM_MAX * 4);
    upl.copy(s.idx + (size_t)b0 * 2, hidx, (size_t)nb * 2);
    upl.copy(s.n_lm + b0, hn, (size_t)nb);
    upl.copy(s.lm_idx + (size_t)b0 * LM_MAX, hl, (size_t)nb * LM_MAX);
    upl.copy(s.anchor_idx + (size_t)b0 * LM_MAX, ha, (size_t)nb * LM_MAX);
    upl.copy(s.tracked + (size_t)b0 * LM_MAX, ht, (size_t)nb * LM_MAX);
    if (int rc = upl.end()) return rc;
    memcpy(s.op.R_lr, o->R_cl2cr, 72); memcpy(s.op.t_lr, o->t_cl2cr, 24);
    s.op.var = o->noise * o->noise; s.op.chi2_thr = o->chi2_thr; s.op.stereo = o->stereo ? 1 : 0;
    s.in_frame = o->in_frame ? 1 : 0;
    if ((int)s.hi.size() != B) s.hi.assign(B, 0);
    for (int i = 0; i < nb; ++i) s.hi[b0 + i] = hi_new[i];
    s.staged = true;
    return INGVIO_OK;
}

// The staged landmark rows name state columns: a marginalisation between stage and run may have shrunk or shifted the state,
// so every run re-validates them against the live dimension (as ingvio_gnss_run does with its staged var_order).
static int landmark_in_state(ingvio_ctx* c, int b0, int nb, const std::vector<int>& n_live, int extra)
{
    if ((int)c->lm.hi.size() != c->d.batch) return INGVIO_OK;
    for (int b = b0; b < b0 + nb; ++b)
        if (c->lm.hi[b] > n_live[b] + extra) { c->err = "a staged landmark row names a column beyond the live state"; return INGVIO_E_NOT_IN_STATE; }
    return INGVIO_OK;
}

int ingvio_landmark_run(ingvio_ctx* c, int b0, int nb)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !c->lm.staged) return INGVIO_E_ARG;
    if (int rc = landmark_in_state(c, b0, nb, c->h_n, 0)) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_status + b0, 0, sizeof(int) * (size_t)nb, c->st));
    return landmark_update_launch(c, b0, nb);
}

int ingvio_landmark_fetch(ingvio_ctx* c, int b0, int nb, double* dx, int* rows, int* accept, double* gamma, int* status)
{
    ENTER(c);
    if (check_range(c, b0, nb) || !c->lm.alloc) return INGVIO_E_ARG;
    if (dx) HIPCHK(c, hipMemcpyAsync(dx, c->lm.dx + (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp, hipMemcpyDeviceToHost, c->st));
    if (rows) HIPCHK(c, hipMemcpyAsync(rows, c->dw.m + b0, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, c->st));
    if (accept) HIPCHK(c, hipMemcpyAsync(accept, c->lm.accept + (size_t)b0 * LM_MAX, sizeof(int) * (size_t)nb * LM_MAX, hipMemcpyDeviceToHost, c->st));
    if (gamma) HIPCHK(c, hipMemcpyAsync(gamma, c->lm.gamma + (size_t)b0 * LM_MAX, 8 * (size_t)nb * LM_MAX, hipMemcpyDeviceToHost, c->st));
    if (status) HIPCHK(c, hipMemcpyAsync(status, c->d_status + b0, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (status) for (int i = 0; i < nb; ++i) status[i] = (status[i] & 4) ? INGVIO_E_NOT_PD : ((status[i] & 2) ? INGVIO_NEG_DIAG : INGVIO_OK);   // bit 4: S not positive definite
    return last_launch(c);
}

int ingvio_set_qr_method(ingvio_ctx* c, int method)
{
    ENTER(c);
    if (!c || method < 0 || method > 2) return INGVIO_E_ARG;
    c->qr_method = method;
    return INGVIO_OK;
}

int ingvio_qr_compress(ingvio_ctx* c, const double* H, int ldh, int m, int n, const double* res, double* Ht, int ldt, double* rt)
{
    ENTER(c);
    if (!c || !H || !res || !Ht || !rt || m < 1 || n < 1 || ldh < m || ldt < n) return INGVIO_E_ARG;
    // Every shape takes the blocked Householder QR of kernels_qr.hip (device buffers and the launch graph cached per shape): no
    // per-call allocation and no use of any filter's staging buffers - a drop-in for six SPQR call sites must not touch batch state.
    // Matrices taller than 6144 rows are factorised in row chunks (launch_qr_dense).
    {
    if (n > 4096) return INGVIO_E_CAPACITY;
    auto& q = c->qr;
    // Every tall stack (m >= 6 n: the stacked MSCKF rows 6150 x 66 and 35100 x 180, the 6000 x 800 stress shape) takes Cholesky-QR,
    // measured 10 - 40 x faster than the blocked Householder QR on all three (profiles/r03_qr_shapes.json: 0.088 / 0.228 / 0.436 ms
    // against 0.85 / 9.0 / 7.2 ms); Householder keeps the shapes that are not tall, where forming the Gram matrix buys nothing.
    // ekfUpdate reads nothing but R^T R and R^T z (exact to eps |H|^2 either way); entry by entry R carries cond(H)^2 eps.
    const int chol = c->qr_method == 2 || (c->qr_method == 0 && (long long)m >= 6LL * n);
    if (q.m != m || q.n != n || q.ldh != ldh || q.chol != chol) {                      // new shape: buffers and the launch graph are rebuilt
        HIPCHK(c, hipStreamSynchronize(c->st));
        if (q.exec) { hipGraphExecDestroy(q.exec); q.exec = nullptr; }
        for (double** p : { &q.dA, &q.db, &q.ws, &q.dT }) { if (*p) hipFree(*p); *p = nullptr; }
        q.m = q.n = q.ldh = 0;
        HIPCHK(c, hipMalloc((void**)&q.dA, 8 * (size_t)ldh * n));
        HIPCHK(c, hipMalloc((void**)&q.db, 8 * (size_t)m));
        const size_t wsd = chol ? qr_chol_workspace_doubles(m, n) : qr_dense_workspace_doubles(m, n);
        HIPCHK(c, hipMalloc((void**)&q.ws, 8 * wsd));
        HIPCHK(c, hipMemsetAsync(q.ws, 0, 8 * wsd, c->st));
        HIPCHK(c, hipMalloc((void**)&q.dT, 8 * ((size_t)n * n + n)));
        // the factorisation is a fixed sequence of 3 launches per 8-column panel: captured once, replayed as one graph
        hipGraph_t g = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
        const int lrc = chol ? launch_qr_chol(q.dA, ldh, q.db, m, n, q.ws, q.dT, n, q.dT + (size_t)n * n, c->st)
                             : launch_qr_dense(q.dA, ldh, q.db, m, n, q.ws, q.dT, n, q.dT + (size_t)n * n, c->st);
        const hipError_t ce = hipStreamEndCapture(c->st, &g);
        if (lrc) { if (g) hipGraphDestroy(g); return INGVIO_E_CAPACITY; }
        if (ce != hipSuccess || !g) { c->err = "hipStreamEndCapture failed"; return INGVIO_E_HIP; }
        const hipError_t ie = hipGraphInstantiate(&q.exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (ie != hipSuccess) { q.exec = nullptr; c->err = "hipGraphInstantiate failed"; return INGVIO_E_HIP; }
        q.m = m; q.n = n; q.ldh = ldh; q.chol = chol;
    }
    int rc2 = up(c, q.dA, H, 8 * (size_t)ldh * n) | up(c, q.db, res, 8 * (size_t)m);
    if (!rc2) {
        ProfScope p(c, PF_FOLD);
        if (hipGraphLaunch(q.exec, c->st) != hipSuccess) rc2 = INGVIO_E_HIP;
    }
    if (!rc2) {
        double* dT = q.dT;
        if (ldt == n) rc2 = hipMemcpyAsync(Ht, dT, 8 * (size_t)n * n, hipMemcpyDeviceToHost, c->st) != hipSuccess;
        else rc2 = hipMemcpy2DAsync(Ht, 8 * (size_t)ldt, dT, 8 * (size_t)n, 8 * (size_t)n, n, hipMemcpyDeviceToHost, c->st) != hipSuccess;
        rc2 |= hipMemcpyAsync(rt, dT + (size_t)n * n, 8 * (size_t)n, hipMemcpyDeviceToHost, c->st) != hipSuccess;
        if (rc2) rc2 = INGVIO_E_HIP;
    }
    hipStreamSynchronize(c->st);
    if (rc2) return rc2 < 0 ? rc2 : INGVIO_E_HIP;
    return last_launch(c);
    }
}

static int frame_stage_impl(ingvio_ctx* c, int b0, int nb, const ingvio_frame_step* steps, const ingvio_msckf_frame* frames,
                            const ingvio_msckf_opts* opts, const double sigma[4], int enable_gnss, double scb, double srw, bool async)
{
    if (phase_busy(c)) return INGVIO_E_ARG;
    // ---- validation: nothing of the context is modified until every input has been accepted ---------------------------
    if (check_range(c, b0, nb) || !steps || !frames || !opts || !sigma) return INGVIO_E_ARG;
    if (async && (b0 != 0 || nb != c->d.batch)) return INGVIO_E_ARG;      // a whole input set is replaced
    const int k = steps[0].k;
    if (k < 1 || k > KMAX) return INGVIO_E_ARG;
    if (!opts->chi2_table || opts->chi2_len < 2 || opts->chi2_len > CHI2_CAP) return INGVIO_E_ARG;
    for (int i = 0; i < nb; ++i)
        if (steps[i].k != k || !steps[i].Phi || !steps[i].G || !steps[i].dt) return INGVIO_E_ARG;
    int fmx = 0;
    // clone indices are checked against the buffer only (grow < 0): the live size at run time depends on whether the run
    // restores the snapshot; ingvio_frame_run checks them against the live state before its first launch
    int rc = validate_frames(c, b0, nb, frames, -1, &fmx);
    if (rc) return rc;
    // ---- resources that may fail, still without side effects on the staged state --------------------------------------
    Uploader upl{ c };
    const size_t n = nb;
    if (async) {
        rc = prepare_async_set(c);
        if (rc) return rc;
    }
    rc = upl.begin(pad64(8 * n * k * 225) + pad64(8 * n * k * 180) + pad64(8 * n * k) + pad64(8 * n * 9) + pad64(4 * n * 5) + pad64(4 * n) +
                   pad64(8 * CHI2_CAP) + pad64(8 * n) + frames_bytes(c, nb) + 1024);
    if (rc) return rc;
    // ---- from here on only HIP runtime errors can occur; they invalidate the staged frame -----------------------------
    // new clock-state indices: the next restore is a full one (the strips the last step wrote are named by the OLD indices); a stage
    // with the same indices - every frame of a running filter between two GNSS initialisations - keeps the partial restore valid
    {
        bool same = c->staged && enable_gnss == c->st_enable_gnss;
        for (int i = 0; i < nb; ++i)
            for (int g = 0; g < 5; ++g) {
                int& old = c->st_gnss[(size_t)(b0 + i) * 5 + g];
                if (old != steps[i].gnss_idx[g]) { same = false; old = steps[i].gnss_idx[g]; }
            }
        if (!same) c->strip_ok = false;
    }
    auto fail = [&](int code) {
        hipStreamSynchronize(c->st);
        if (c->st_copy) hipStreamSynchronize(c->st_copy);
        c->staged = false; c->copy_pending = false;
        return code;
    };
    if (async) {
        // the set about to be overwritten was last read two frames ago; its "free" event is on the compute stream
        if (c->free_valid[c->set_id ^ 1] && hipStreamWaitEvent(c->st_copy, c->ev_free[c->set_id ^ 1], 0) != hipSuccess) return fail(INGVIO_E_HIP);
        swap_input_sets(c);
        upl.stream = c->st_copy;
    } else {
        if (wait_inputs(c)) return fail(INGVIO_E_HIP);   // an earlier asynchronous stage into this set must land before it is overwritten in-stream
    }
    for (int i = 0; i < nb; ++i) {
        c->st_marg[b0 + i] = steps[i].marg_idx;
        int hi = -1;
        for (int q = 0; q < frames[i].n_clones; ++q) if (frames[i].clone_idx[q] > hi) hi = frames[i].clone_idx[q];
        c->st_cidx_hi[b0 + i] = hi;
    }
    // everything goes through ONE pinned slab: packed by a few host threads, copied asynchronously, no stream sync
    double* Phi = upl.take<double>(n * k * 225); double* G = upl.take<double>(n * k * 180); double* dt = upl.take<double>(n * k);
    double* R = upl.take<double>(n * 9); int* gi = upl.take<int>(n * 5); int* mi = upl.take<int>(n);
    double* chi2 = upl.take<double>(CHI2_CAP); double* nz = upl.take<double>(n);
    const double var = opts->noise * opts->noise;
    parallel_for(nb, [=](int i) {
        memcpy(Phi + (size_t)i * k * 225, steps[i].Phi, 8 * (size_t)k * 225);
        memcpy(G + (size_t)i * k * 180, steps[i].G, 8 * (size_t)k * 180);
        memcpy(dt + (size_t)i * k, steps[i].dt, 8 * (size_t)k);
        memcpy(R + (size_t)i * 9, steps[i].R_i2w, 72);
        for (int g = 0; g < 5; ++g) gi[(size_t)i * 5 + g] = steps[i].gnss_idx[g];
        mi[i] = steps[i].marg_idx;
        nz[i] = var;
    });
    memcpy(chi2, opts->chi2_table, 8 * (size_t)opts->chi2_len);
    upl.copy(c->d_Phi + (size_t)b0 * k * 225, Phi, n * k * 225);
    upl.copy(c->d_G + (size_t)b0 * k * 180, G, n * k * 180);
    upl.copy(c->d_dt + (size_t)b0 * k, dt, n * k);
    upl.copy(c->d_R + (size_t)b0 * 9, R, n * 9);
    upl.copy(c->d_gnss + (size_t)b0 * 5, gi, n * 5);
    upl.copy(c->d_idx + b0, mi, n);
    upl.copy(c->d_chi2, chi2, (size_t)opts->chi2_len);
    upl.copy(c->d_noise + b0, nz, n);
    c->upc.chi2_ok = false; c->upc.noise_ok = false;
    MsckfOpts& op = c->st_op;
    memcpy(op.R_lr, opts->R_cl2cr, 72);
    memcpy(op.t_lr, opts->t_cl2cr, 24);
    op.var = var; op.max_accept = opts->max_accept; op.selected_variant = opts->selected_variant;
    op.chi2 = c->d_chi2; op.chi2_len = opts->chi2_len;
    pack_frames(c, upl, b0, nb, frames, &fmx);
    rc = upl.end();
    if (rc) return fail(rc);
    if (async) {
        if (hipEventRecord(c->ev_copy, c->st_copy) != hipSuccess) return fail(INGVIO_E_HIP);
        c->copy_pending = true;
    }
    c->st_k = k; c->st_stereo = opts->stereo; c->st_enable_gnss = enable_gnss; c->st_scb = scb; c->st_srw = srw;
    memcpy(c->st_sigma, sigma, 32);
    if (!c->staged || fmx > c->st_fmax_used) c->st_fmax_used = fmx;
    c->staged = true;
    return INGVIO_OK;
}

// ---- device-resident track store ---------------------------------------------------------------------------------------------
int ingvio_tracks_create(ingvio_ctx* c, int t_max)
{
    ENTER(c);
    if (!c || t_max < 1 || t_max > 65536) return INGVIO_E_ARG;
    if (c->d.c_max > 64) return INGVIO_E_UNSUPPORTED;
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (c->st_copy) HIPCHK(c, hipStreamSynchronize(c->st_copy));
    auto& t = c->trk;
    const size_t B = c->d.batch, T = t_max, C = c->d.c_max, F = c->d.f_max;
    if (t.t_max != t_max) {
        for (void* p : { (void*)t.uv, (void*)t.pf, (void*)t.mask, (void*)t.stage[0], (void*)t.stage[1] }) if (p) hipFree(p);
        t = ingvio_ctx::Tracks();
        if (dalloc(c, &t.uv, B * T * C * 4) || dalloc(c, &t.pf, B * T * 3) || dalloc(c, &t.mask, B * T)) return INGVIO_E_HIP;
        // worst case of one staged delta per filter: header + ints (drop C, free T, obs T, pf T, clone idx C, features F, gnss 5) +
        // masks F + doubles (obs 4 T, points 3 T, clone poses 12 C, IMU 7 KMAX, state 24)
        t.stage_cap = B * (4 * (TRK_HDR + 3 * T + 2 * C + F + 8) + 8 * F + 8 * (7 * T + 12 * C + 7 * KMAX + 24) + 256) + 4096;
        if (dalloc(c, &t.stage[0], t.stage_cap) || dalloc(c, &t.stage[1], t.stage_cap)) return INGVIO_E_HIP;
        t.t_max = t_max;
    } else {
        HIPCHK(c, hipMemsetAsync(t.mask, 0, 8 * B * T, c->st));
        HIPCHK(c, hipMemsetAsync(t.pf, 0, 8 * B * T * 3, c->st));
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    return INGVIO_OK;
}

int ingvio_frame_stage_tracks(ingvio_ctx* c, int b0, int nb, const ingvio_frame_step_raw* steps, const ingvio_track_frame* frames,
                              const ingvio_msckf_opts* opts, const double sigma[4], int enable_gnss, double scb, double srw, int async)
{
    ENTER(c);
    if (phase_busy(c)) return INGVIO_E_ARG;
    if (check_range(c, b0, nb) || !steps || !frames || !opts || !sigma) return INGVIO_E_ARG;
    if (!c->trk.t_max) { c->err = "ingvio_frame_stage_tracks without ingvio_tracks_create"; return INGVIO_E_ARG; }
    if (async && (b0 != 0 || nb != c->d.batch)) return INGVIO_E_ARG;
    const int k = steps[0].k, T = c->trk.t_max, cm = c->d.c_max, fm = c->d.f_max;
    if (k < 1 || k > KMAX) return INGVIO_E_ARG;
    if (!opts->chi2_table || opts->chi2_len < 2 || opts->chi2_len > CHI2_CAP) return INGVIO_E_ARG;
    // ---- layout (counts only), then validation + packing per filter on a few host threads; nothing of the context is touched before
    //      every filter's delta has been found consistent (the pinned slab is scratch) ----
    std::vector<int> hdr((size_t)nb * TRK_HDR, 0);
    size_t ni = 0, nd = 0, nm = 0;
    int fmx = 0;
    for (int i = 0; i < nb; ++i) {
        const ingvio_track_frame& f = frames[i];
        const ingvio_frame_step_raw& s = steps[i];
        if (s.k != k || !s.imu) return INGVIO_E_ARG;
        if (f.n_drop < 0 || f.n_drop > cm || f.n_free < 0 || f.n_free > T || f.n_obs < 0 || f.n_obs > T || f.n_pf < 0 || f.n_pf > T) return INGVIO_E_CAPACITY;
        if (f.n_clones < 0 || f.n_clones > cm || f.n_feat < 0 || f.n_feat > fm) return INGVIO_E_CAPACITY;
        if ((f.n_drop && !f.drop_slots) || (f.n_free && !f.free_tracks) || (f.n_obs && (!f.obs_track || !f.obs_uv)) || (f.n_pf && (!f.pf_track || !f.pf)) ||
            (f.n_clones && (!f.clone_idx || !f.clone_R || !f.clone_p)) || (f.n_feat && (!f.feat_track || !f.feat_anchor || !f.feat_dof))) return INGVIO_E_ARG;
        if (f.append_slot >= cm || (f.n_obs > 0 && f.append_slot < 0)) return INGVIO_E_ARG;
        if (f.n_feat > fmx) fmx = f.n_feat;
        int* h = &hdr[(size_t)i * TRK_HDR];
        h[TRK_N_DROP] = f.n_drop; h[TRK_APPEND] = f.append_slot; h[TRK_N_OBS] = f.n_obs; h[TRK_N_FREE] = f.n_free; h[TRK_N_PF] = f.n_pf;
        h[TRK_N_CLONES] = f.n_clones; h[TRK_N_FEAT] = f.n_feat; h[TRK_K] = k; h[TRK_HAS_SEL] = f.feat_sel ? 1 : 0; h[TRK_MARG] = s.marg_idx;
        h[TRK_OFF_I] = (int)ni; h[TRK_OFF_D] = (int)nd; h[TRK_OFF_M] = (int)nm;
        int oi = 0, od = 0;
        h[TRK_I_DROP] = oi; oi += f.n_drop;
        h[TRK_I_FREE] = oi; oi += f.n_free;
        h[TRK_I_OBS] = oi; oi += f.n_obs;
        h[TRK_I_PF] = oi; oi += f.n_pf;
        h[TRK_I_CIDX] = oi; oi += f.n_clones;
        h[TRK_I_FEAT] = oi; oi += f.n_feat;
        h[TRK_I_GNSS] = oi; oi += 5;
        od = (od + 3) & ~3; h[TRK_D_OBS] = od; od += 4 * f.n_obs;                 // 32-byte aligned: read as double4
        h[TRK_D_PF] = od; od += 3 * f.n_pf;
        h[TRK_D_CR] = od; od += 9 * f.n_clones;
        h[TRK_D_CP] = od; od += 3 * f.n_clones;
        h[TRK_D_IMU] = od; od += 7 * k;
        h[TRK_D_STATE] = od; od += 24;
        ni += (size_t)oi; nd += ((size_t)od + 3) & ~(size_t)3;
        if (f.feat_sel) nm += (size_t)f.n_feat;
    }
    const size_t off_i = pad64(4 * (size_t)nb * TRK_HDR), off_m = off_i + pad64(4 * ni), off_d = off_m + pad64(8 * nm), total = off_d + pad64(8 * nd);
    if (total > c->trk.stage_cap) return INGVIO_E_CAPACITY;
    Uploader upl{ c };
    int rc = 0;
    if (async) { rc = prepare_async_set(c); if (rc) return rc; }
    rc = upl.begin(total + pad64(8 * CHI2_CAP) + pad64(8 * (size_t)nb) + 1024);
    if (rc) return rc;
    char* slab = upl.take<char>(total);
    int* hp = reinterpret_cast<int*>(slab);
    int* ipool = reinterpret_cast<int*>(slab + off_i);
    unsigned long long* mpool = reinterpret_cast<unsigned long long*>(slab + off_m);
    double* dpool = reinterpret_cast<double*>(slab + off_d);
    memcpy(hp, hdr.data(), 4 * hdr.size());
    const int* hd = hdr.data();
    const int n_max = c->d.n_max;
    std::vector<int> bad((size_t)nb, 0);
    int* badp = bad.data();
    parallel_for(nb, [=](int i) {
        const ingvio_track_frame& f = frames[i];
        const ingvio_frame_step_raw& s = steps[i];
        const int* h = hd + (size_t)i * TRK_HDR;
        int* ip = ipool + h[TRK_OFF_I];
        double* dp = dpool + h[TRK_OFF_D];
        int err = 0;
        for (int q = 0; q < f.n_drop; ++q) if (f.drop_slots[q] < 0 || f.drop_slots[q] >= cm || (q && f.drop_slots[q] <= f.drop_slots[q - 1])) err = 1;
        for (int q = 0; q < f.n_free; ++q) if (f.free_tracks[q] < 0 || f.free_tracks[q] >= T) err = 1;
        for (int q = 0; q < f.n_obs; ++q) if (f.obs_track[q] < 0 || f.obs_track[q] >= T) err = 1;
        for (int q = 0; q < f.n_pf; ++q) if (f.pf_track[q] < 0 || f.pf_track[q] >= T) err = 1;
        for (int q = 0; q < f.n_clones; ++q) if (f.clone_idx[q] < 0 || f.clone_idx[q] + 6 > n_max) err = 1;
        int* fw = ip + h[TRK_I_FEAT];
        for (int j = 0; j < f.n_feat; ++j) {
            if (f.feat_track[j] < 0 || f.feat_track[j] >= T || f.feat_anchor[j] < 0 || f.feat_anchor[j] >= f.n_clones || f.feat_dof[j] < 0 || f.feat_dof[j] > 255) err = 1;
            fw[j] = (int)((unsigned)f.feat_track[j] | ((unsigned)f.feat_anchor[j] << 16) | ((unsigned)f.feat_dof[j] << 24));
        }
        badp[i] = err;
        if (f.n_drop) memcpy(ip + h[TRK_I_DROP], f.drop_slots, 4 * (size_t)f.n_drop);
        if (f.n_free) memcpy(ip + h[TRK_I_FREE], f.free_tracks, 4 * (size_t)f.n_free);
        if (f.n_obs) { memcpy(ip + h[TRK_I_OBS], f.obs_track, 4 * (size_t)f.n_obs); memcpy(dp + h[TRK_D_OBS], f.obs_uv, 32 * (size_t)f.n_obs); }
        if (f.n_pf) { memcpy(ip + h[TRK_I_PF], f.pf_track, 4 * (size_t)f.n_pf); memcpy(dp + h[TRK_D_PF], f.pf, 24 * (size_t)f.n_pf); }
        if (f.n_clones) {
            memcpy(ip + h[TRK_I_CIDX], f.clone_idx, 4 * (size_t)f.n_clones);
            memcpy(dp + h[TRK_D_CR], f.clone_R, 72 * (size_t)f.n_clones); memcpy(dp + h[TRK_D_CP], f.clone_p, 24 * (size_t)f.n_clones);
        }
        for (int g = 0; g < 5; ++g) ip[h[TRK_I_GNSS] + g] = s.gnss_idx[g];
        if (f.feat_sel) memcpy(mpool + h[TRK_OFF_M], f.feat_sel, 8 * (size_t)f.n_feat);
        memcpy(dp + h[TRK_D_IMU], s.imu, 56 * (size_t)k);
        double* st0 = dp + h[TRK_D_STATE];
        memcpy(st0, s.R, 72); memcpy(st0 + 9, s.p, 24); memcpy(st0 + 12, s.v, 24); memcpy(st0 + 15, s.bg, 24); memcpy(st0 + 18, s.ba, 24); memcpy(st0 + 21, s.gravity, 24);
    });
    for (int i = 0; i < nb; ++i) if (bad[i]) { c->err = "ingvio_frame_stage_tracks: a slot, track, clone index, anchor or dof is out of range"; return INGVIO_E_ARG; }
    // ---- from here on only HIP runtime errors can occur ----
    {
        bool same = c->staged && enable_gnss == c->st_enable_gnss;      // as frame_stage_impl: the partial restore survives unchanged clock-state indices
        for (int i = 0; i < nb; ++i)
            for (int g = 0; g < 5; ++g) {
                int& old = c->st_gnss[(size_t)(b0 + i) * 5 + g];
                if (old != steps[i].gnss_idx[g]) { same = false; old = steps[i].gnss_idx[g]; }
            }
        if (!same) c->strip_ok = false;
    }
    auto fail = [&](int code) {
        hipStreamSynchronize(c->st);
        if (c->st_copy) hipStreamSynchronize(c->st_copy);
        c->staged = false; c->copy_pending = false;
        return code;
    };
    hipStream_t S = c->st;
    char* dstage = c->trk.stage[0];
    if (async) {
        if (c->free_valid[c->set_id ^ 1] && hipStreamWaitEvent(c->st_copy, c->ev_free[c->set_id ^ 1], 0) != hipSuccess) return fail(INGVIO_E_HIP);
        swap_input_sets(c);
        upl.stream = c->st_copy; S = c->st_copy; dstage = c->trk.stage[1];
    } else if (wait_inputs(c)) return fail(INGVIO_E_HIP);
    double* chi2 = upl.take<double>(CHI2_CAP); double* nz = upl.take<double>(nb);
    const double var = opts->noise * opts->noise;
    memcpy(chi2, opts->chi2_table, 8 * (size_t)opts->chi2_len);
    for (int i = 0; i < nb; ++i) nz[i] = var;
    upl.copy(dstage, slab, total);
    upl.copy(c->d_chi2, chi2, (size_t)opts->chi2_len);
    upl.copy(c->d_noise + b0, nz, (size_t)nb);
    c->upc.chi2_ok = false; c->upc.noise_ok = false;
    TrackStage ts{ reinterpret_cast<const int*>(dstage), reinterpret_cast<const int*>(dstage + off_i),
                   reinterpret_cast<const double*>(dstage + off_d), reinterpret_cast<const unsigned long long*>(dstage + off_m) };
    TrackStore store{ c->trk.uv, c->trk.mask, c->trk.pf, T, cm };
    FrameOut fo{ c->d_clone_idx, c->d_clone_R, c->d_clone_p, c->d_nclones, c->d_nfeat, c->d_pf, c->d_anchor, c->d_mask, c->d_uv, c->d_dof, cm, fm };
    launch_imu_steps(ts, b0, nb, k, c->d_Phi, c->d_G, c->d_dt, c->d_R, S);
    launch_tracks_apply(ts, store, b0, nb, S);
    launch_tracks_gather(ts, store, fo, b0, nb, c->d_idx, c->d_gnss, S);
    if (hipGetLastError() != hipSuccess) return fail(INGVIO_E_HIP);
    rc = upl.end();
    if (rc) return fail(rc);
    if (async) {
        if (hipEventRecord(c->ev_copy, c->st_copy) != hipSuccess) return fail(INGVIO_E_HIP);
        c->copy_pending = true;
    }
    for (int i = 0; i < nb; ++i) {
        c->st_marg[b0 + i] = steps[i].marg_idx;
        int hi = -1;
        for (int q = 0; q < frames[i].n_clones; ++q) if (frames[i].clone_idx[q] > hi) hi = frames[i].clone_idx[q];
        c->st_cidx_hi[b0 + i] = hi;
        c->h_nclones[b0 + i] = frames[i].n_clones;
    }
    MsckfOpts& op = c->st_op;
    memcpy(op.R_lr, opts->R_cl2cr, 72);
    memcpy(op.t_lr, opts->t_cl2cr, 24);
    op.var = var; op.max_accept = opts->max_accept; op.selected_variant = opts->selected_variant;
    op.chi2 = c->d_chi2; op.chi2_len = opts->chi2_len;
    c->st_k = k; c->st_stereo = opts->stereo; c->st_enable_gnss = enable_gnss; c->st_scb = scb; c->st_srw = srw;
    memcpy(c->st_sigma, sigma, 32);
    if (!c->staged || fmx > c->st_fmax_used) c->st_fmax_used = fmx;
    c->staged = true;
    return INGVIO_OK;
}

int ingvio_frame_stage(ingvio_ctx* c, int b0, int nb, const ingvio_frame_step* steps, const ingvio_msckf_frame* frames,
                       const ingvio_msckf_opts* opts, const double sigma[4], int enable_gnss, double scb, double srw)
{
    ENTER(c);
    return frame_stage_impl(c, b0, nb, steps, frames, opts, sigma, enable_gnss, scb, srw, false);
}

int ingvio_frame_stage_async(ingvio_ctx* c, int b0, int nb, const ingvio_frame_step* steps, const ingvio_msckf_frame* frames,
                             const ingvio_msckf_opts* opts, const double sigma[4], int enable_gnss, double scb, double srw)
{
    ENTER(c);
    return frame_stage_impl(c, b0, nb, steps, frames, opts, sigma, enable_gnss, scb, srw, true);
}

// The fused frame step of windows up to 16 clones with the batch dealt to P slices.  Slice p = filters [p B / P, (p + 1) B / P) rounded
// to multiples of 8 (the kernels' XCD-aware block order works on groups of 8 filters), on stream part[p].st:
//     restore -> propagate + clone -> [wait: gate of slice p - 1] gate [record] -> Gram -> solve -> (GNSS in-frame) -> apply + marginalise
// Same kernels, same arguments per filter as the unsplit step: results are bit-identical (tests/test_gpu_parity.py runs both).
static int frame_run_split(ingvio_ctx* c, int restore_prior, int P, bool gnss_fuse)
{
    const int B = c->d.batch;
    if (parts_prepare(c, P)) return INGVIO_E_HIP;
    if (c->split_pending && c->copy_pending) ENTER(c);                     // fresh inputs on the copy stream: order them through c->st
    if (wait_inputs(c)) return INGVIO_E_HIP;
    if (!c->split_pending) {                                                // work on c->st since the last split step: the slices start behind it
        HIPCHK(c, hipEventRecord(c->ev_split_fork, c->st));
        for (int p = 0; p < P; ++p) HIPCHK(c, hipStreamWaitEvent(c->part[p].st, c->ev_split_fork, 0));
        c->tok_last = nullptr;
    }
    const bool strips = restore_prior && c->strip_ok && c->mut_seq == c->strip_seq;
    if (restore_prior) { c->h_n = c->h_n_snap; std::fill(c->h_cur.begin(), c->h_cur.end(), 0); }
    for (int b = 0; b < B; ++b) c->h_n[b] += 6;
    const int k = c->st_k;
    int rc = 0;
    auto range = [&](int p, int& b0, int& nb) {
        b0 = (int)((long long)B * p / P) & ~7;
        const int b1 = p + 1 == P ? B : ((int)((long long)B * (p + 1) / P) & ~7);
        nb = b1 - b0;
    };
    // Issue order = order of the throughput segments on the device: gate + Gram of every slice, then the apply of every slice; a
    // segment waits for the one issued before it (whatever stream that was on) - the slices' latency-bound kernels (restore, propagate,
    // solve, the flips) run freely in between, under the other slices' segments
    for (int p = 0; p < P && !rc; ++p) {
        int b0, nb; range(p, b0, nb);
        if (nb <= 0) continue;
        auto& q = c->part[p];
        c->run_st = q.st;
        const bool from_snap = strips && propagate_can_restore(c->d.n_max) && !no_snap_propagate();
        if (restore_prior && !from_snap) {
            ProfScope pr(c, PF_RESTORE);
            if (strips) launch_restore_strips(view(c), b0, nb, c->d.n_max, c->Psnap, c->d_n_snap, c->st_enable_gnss ? c->d_gnss : nullptr, q.st);
            else launch_restore(view(c), b0, nb, c->d.n_max, c->Psnap, c->d_n_snap, q.st);
        }
        {
            ProfScope pr(c, PF_PROPAGATE);
            launch_propagate(view(c), b0, nb, c->d.n_max, c->d_Phi + (size_t)b0 * k * 225, c->d_G + (size_t)b0 * k * 180, c->d_dt + (size_t)b0 * k, k,
                             c->st_enable_gnss ? c->d_gnss + (size_t)b0 * 5 : nullptr, c->st_sigma, c->st_enable_gnss, c->st_scb, c->st_srw, q.st,
                             c->d_R + (size_t)b0 * 9, c->d_status, from_snap ? c->Psnap : nullptr, c->d_n_snap);
        }
        c->tok_wait = c->tok_last; c->tok_rec = q.ev_gate;
        rc = run_msckf_factored(c, b0, nb, c->st_op, c->st_stereo, c->st_fmax_used, c->d_idx + b0, 6, 1, gnss_fuse);
        c->tok_last = rc ? nullptr : q.ev_gate;
    }
    for (int p = 0; p < P && !rc; ++p) {
        int b0, nb; range(p, b0, nb);
        if (nb <= 0) continue;
        auto& q = c->part[p];
        c->run_st = q.st;
        c->tok_wait = c->tok_last; c->tok_rec = q.ev_apply;
        rc = run_msckf_factored(c, b0, nb, c->st_op, c->st_stereo, c->st_fmax_used, c->d_idx + b0, 6, 2, gnss_fuse);
        c->tok_last = rc ? nullptr : q.ev_apply;
        if (!rc && !c->apply_flipped) {
            ProfScope pr(c, PF_MARG);
            launch_post_marg(view(c), b0, nb, c->d_idx + b0, 6, q.st);
        }
    }
    for (int p = 0; p < P; ++p) hipEventRecord(c->part[p].ev_done, c->part[p].st);
    c->tok_wait = nullptr; c->tok_rec = nullptr;
    c->run_st = c->st;
    c->split_pending = true;
    if (rc) { c->tok_last = nullptr; ENTER(c); return rc; }
    bool all_marg = true;
    for (int b = 0; b < B; ++b) { if (c->st_marg[b] >= 0) { c->h_n[b] -= 6; c->h_cur[b] ^= 1; } else all_marg = false; }
    c->strip_ok = all_marg && restore_prior;
    c->strip_seq = c->mut_seq;
    if (c->gn.staged && c->gn.in_frame && !restore_prior) c->gn.staged = false;
    return INGVIO_OK;
}

static int frame_run_impl(ingvio_ctx* c, int restore_prior, int phase)
{
    if (!c || !c->staged) return INGVIO_E_ARG;
    if (phase && c->method != 1) return INGVIO_E_UNSUPPORTED;             // the split needs the information form's [A | b]
    // protocol of the split step: 1 -> (ingvio_debug_msckf_info / ingvio_info_set) -> 2, exactly once each; the accepted-feature
    // cap is a global order over the features and is not defined across shards
    if (phase != 2 && phase_busy(c)) return INGVIO_E_ARG;
    if (phase == 2 && !c->phase_pending) { c->err = "ingvio_frame_run_phase(.., 2) without a preceding phase 1"; return INGVIO_E_ARG; }
    if (phase != 0 && c->st_op.max_accept > 0) { c->err = "max_accept is not supported by the split (feature-sharded) frame step"; return INGVIO_E_ARG; }
    if (phase == 2) {                                                       // back half: solve + apply + marginalise from the partials
        c->phase_pending = false;
        const int B2 = c->d.batch;
        const bool with_lm2 = c->lm.staged && c->lm.in_frame;
        int rc2 = run_msckf_factored(c, 0, B2, c->st_op, c->st_stereo, c->st_fmax_used, with_lm2 ? nullptr : c->d_idx, with_lm2 ? 0 : 6, 2);
        if (rc2) return rc2;
        if (with_lm2) { rc2 = landmark_update_launch(c, 0, B2); if (rc2) return rc2; }
        {
            ProfScope p(c, PF_MARG);
            if (!with_lm2) { if (!c->apply_flipped) launch_post_marg(view(c), 0, B2, c->d_idx, 6, c->st); }
            else launch_marginalize(view(c), 0, B2, c->d.n_max, c->d_idx, 6, c->st);
        }
        for (int b = 0; b < B2; ++b) if (c->st_marg[b] >= 0) { c->h_n[b] -= 6; c->h_cur[b] ^= 1; }
        // as in the unsplit step: a landmark stage belongs to ONE frame unless the caller replays the same prior
        if (with_lm2 && !c->phase_restore) c->lm.staged = false;
        c->strip_ok = false;
        c->mut_seq++;
        // an in-frame GNSS stage: the split step never folds it into the write-back - its own pass, as the unsplit step does for the
        // shapes it cannot fold (it used to be dropped silently here)
        if (c->gn.staged && c->gn.in_frame && c->gn.m_cap > 0) {
            rc2 = gnss_run_separate(c, 0, B2, true);
            if (rc2) return rc2;
            if (!c->phase_restore) c->gn.staged = false;
        }
        if (c->alt_ready) { HIPCHK(c, hipEventRecord(c->ev_free[c->set_id], c->st)); c->free_valid[c->set_id] = true; }
        return INGVIO_OK;
    }
    const int B = c->d.batch;
    // ---- every check that can refuse the step comes before the first launch and before any host-side state changes ----
    if (restore_prior && !c->has_snap) return INGVIO_E_ARG;
    if (c->method == 0 && c->d.c_max > 16) return INGVIO_E_UNSUPPORTED;       // the dense MSCKF method stops at 16 clones
    {
        const std::vector<int>& n0 = restore_prior ? c->h_n_snap : c->h_n;
        for (int b = 0; b < B; ++b) {
            if (n0[b] < 21) return INGVIO_E_ARG;
            if (n0[b] + 6 > c->d.n_max) return INGVIO_E_CAPACITY;
            if (c->st_cidx_hi[b] + 6 > n0[b] + 6) return INGVIO_E_NOT_IN_STATE;      // a staged clone lies beyond the live state
            if (c->st_marg[b] >= 0 && c->st_marg[b] + 6 > n0[b] + 6) return INGVIO_E_NOT_IN_STATE;
        }
        if (c->lm.staged && c->lm.in_frame) { if (int rc = landmark_in_state(c, 0, B, n0, 6)) return rc; }
    }
    // ---- split step (round 6): the batch as `parts` slices on their own streams, gates chained, nothing joined at the end ----
    {
        const bool with_lm0 = c->lm.staged && c->lm.in_frame;
        const bool gnss0 = c->gn.staged && c->gn.in_frame && c->gn.m_cap > 0;
        bool gfuse0 = gnss0 && c->d.c_max <= 16 && c->gn.m_cap <= 16 && c->gn.nc_max <= 16;
        for (int b = 0; b < B && gfuse0; ++b) if (c->st_marg[b] < 0 || c->gn.hi[b] > c->st_marg[b]) gfuse0 = false;
        // automatic = off: measured on MI355X at 512 filters x (150 features, 11 clones, N = 249) the split LOSES - 0.535 ms unsplit, 0.58
        // (gates chained) / 0.65 (throughput segments chained) with two slices, 0.73-0.97 with three and four
        int P = c->parts_req > 0 ? c->parts_req : 1;
        if (P > 4) P = 4;
        if (phase == 0 && P > 1 && c->method == 1 && !with_lm0 && c->d.c_max <= 16 && (!gnss0 || gfuse0) && B >= 16 * P)
            return frame_run_split(c, restore_prior, P, gfuse0);
    }
    ENTER(c);
    if (wait_inputs(c)) return INGVIO_E_HIP;
    // restore_prior right after a fused frame step that itself began with a restore: half 0 still equals the snapshot outside the
    // propagation's strips - the propagation then READS the snapshot and writes half 0, no restore pass at all (round 6; before:
    // k_restore_strips, 23 us and 94 MB per 512 filters)
    bool from_snap = false;
    if (restore_prior) {
        const bool strips = c->strip_ok && c->mut_seq == c->strip_seq;
        from_snap = strips && propagate_can_restore(c->d.n_max) && !no_snap_propagate();
        if (!from_snap) {
            ProfScope p(c, PF_RESTORE);
            if (strips) launch_restore_strips(view(c), 0, B, c->d.n_max, c->Psnap, c->d_n_snap, c->st_enable_gnss ? c->d_gnss : nullptr, c->st);
            else launch_restore(view(c), 0, B, c->d.n_max, c->Psnap, c->d_n_snap, c->st);
        }
        c->h_n = c->h_n_snap;
        std::fill(c->h_cur.begin(), c->h_cur.end(), 0);
    }
    {
        // one launch: status reset + K1 (k composed IMU steps) + K2 (clone) when a workgroup owns a whole filter
        ProfScope p(c, PF_PROPAGATE);
        launch_propagate(view(c), 0, B, c->d.n_max, c->d_Phi, c->d_G, c->d_dt, c->st_k, c->st_enable_gnss ? c->d_gnss : nullptr,
                         c->st_sigma, c->st_enable_gnss, c->st_scb, c->st_srw, c->st, c->d_R, c->d_status,
                         from_snap ? c->Psnap : nullptr, c->d_n_snap);
    }
    for (int b = 0; b < B; ++b) c->h_n[b] += 6;
    // factored path: the marginalisation of the oldest clone rides on the update's write-back (k_info_apply
    // stores the updated covariance compacted into the other ping-pong half); dense path: separate kernel
    // staged in-frame landmark update (IngvioFilter.cpp:296-322: MSCKF updates, landmark update, then the marginalisation):
    // the marginalisation cannot ride on the MSCKF write-back then
    const bool with_lm = c->lm.staged && c->lm.in_frame;
    const bool fuse = c->method == 1 && !with_lm;
    if (phase == 1) {                                                       // front half only: [A | b] partials stay on the device
        c->strip_ok = false;
        c->mut_seq++;
        const int rc1 = run_msckf_factored(c, 0, B, c->st_op, c->st_stereo, c->st_fmax_used, nullptr, 0, 1);
        // h_n is already + 6 (the clone exists on the device): only phase 2 may follow - and only after a front half that was
        // launched; after a refused one the step has to be abandoned with ingvio_cov_restore or replayed with
        // ingvio_frame_run(restore_prior)
        if (rc1 == 0) { c->phase_pending = true; c->phase_restore = restore_prior != 0; }
        return rc1;
    }
    // with a landmark update to follow, the MSCKF update is still written OUT OF PLACE (a "marginalisation" of zero columns into the
    // other half, then the halves flip): the solve then reads P[:, clone columns] straight from the untouched prior instead of
    // copying them first (0.035 ms per 512 filters), and the landmark downdate writes back into the first half with the real
    // marginalisation fused
    const bool lm_oop = with_lm && c->method == 1;
    // in-frame GNSS stage (ingvio_gnss_opts::in_frame): on the MSCKF write-back when everything fits one sweep, else as its own pass below
    const bool gnss_here = c->gn.staged && c->gn.in_frame && c->gn.m_cap > 0;
    bool gnss_fuse = gnss_here && fuse && c->d.c_max <= 16 && c->gn.m_cap <= 16 && c->gn.nc_max <= 16;
    // the update's columns must keep their indices through the frame's marginalisation: var_order entirely below the clone that goes
    for (int b = 0; b < B && gnss_fuse; ++b) if (c->st_marg[b] < 0 || c->gn.hi[b] > c->st_marg[b]) gnss_fuse = false;
    int rc = fuse ? run_msckf_factored(c, 0, B, c->st_op, c->st_stereo, c->st_fmax_used, c->d_idx, 6, 0, gnss_fuse)
                  : (c->method == 1 ? (lm_oop ? run_msckf_factored(c, 0, B, c->st_op, c->st_stereo, c->st_fmax_used, c->d_zero_idx, 0)
                                              : run_msckf_factored(c, 0, B, c->st_op, c->st_stereo, c->st_fmax_used))
                                    : run_msckf(c, 0, B, c->st_op, c->st_stereo, c->st_fmax_used));
    if (rc) return rc;
    if (lm_oop) {
        if (!c->apply_flipped) launch_post_marg(view(c), 0, B, c->d_zero_idx, 0, c->st);
        for (int b = 0; b < B; ++b) c->h_cur[b] ^= 1;
    }
    bool lm_fused = false;
    if (with_lm) { rc = landmark_update_launch(c, 0, B, c->d_idx, 6, &lm_fused); if (rc) return rc; }
    if (!(fuse && c->apply_flipped)) {             // (the write-back of the fused step has flipped the halves itself: no launch, no profile slot)
        ProfScope p(c, PF_MARG);
        if (fuse || lm_fused) launch_post_marg(view(c), 0, B, c->d_idx, 6, c->st);
        else launch_marginalize(view(c), 0, B, c->d.n_max, c->d_idx, 6, c->st);
    }
    // a landmark stage belongs to ONE frame: unless the caller replays the same prior (restore_prior, the bench and the parity tests)
    // the staged rows are consumed here, so that a later frame cannot re-apply them to a state that has moved on
    if (with_lm && !restore_prior) c->lm.staged = false;
    bool all_marg = true;
    for (int b = 0; b < B; ++b) { if (c->st_marg[b] >= 0) { c->h_n[b] -= 6; c->h_cur[b] ^= 1; } else all_marg = false; }
    c->strip_ok = fuse && all_marg && restore_prior;
    c->strip_seq = c->mut_seq;
    if (gnss_here && !gnss_fuse) {                 // not foldable (large window, landmarks in the frame, many rows): its own pass, now
        rc = gnss_run_separate(c, 0, B, true);
        if (rc) return rc;
    }
    // like a landmark stage, an in-frame GNSS stage belongs to ONE frame: consumed here unless the caller replays the same prior, so
    // that a later frame cannot apply the old rows again; ingvio_gnss_fetch still returns this frame's results
    if (c->gn.staged && c->gn.in_frame && !restore_prior) c->gn.staged = false;
    if (c->alt_ready) {                            // this input set may be refilled once the kernels above are done with it
        HIPCHK(c, hipEventRecord(c->ev_free[c->set_id], c->st));
        c->free_valid[c->set_id] = true;
    }
    return INGVIO_OK;
}

int ingvio_frame_run(ingvio_ctx* c, int restore_prior) { return frame_run_impl(c, restore_prior, 0); }
int ingvio_set_frame_parts(ingvio_ctx* c, int parts)
{
    if (!c || parts < -1 || parts > 4) return INGVIO_E_ARG;
    ENTER(c);
    c->parts_req = parts == 0 ? 1 : parts;
    return INGVIO_OK;
}
int ingvio_frame_run_phase(ingvio_ctx* c, int restore_prior, int phase)
{
    if (phase < 0 || phase > 2) return INGVIO_E_ARG;
    return frame_run_impl(c, restore_prior, phase);
}

// the stacked information [A | b] = [sum_j H_j^T H_j | sum_j H_j^T r_j] of filter b replaces the chunk partials (feature-sharded
// single filter: every rank sums the ranks' partials and continues with ingvio_frame_run_phase(.., 2))
int ingvio_info_set(ingvio_ctx* c, int b, const double* A, int ncol, int n_accepted)
{
    ENTER(c);
    if (check_range(c, b, 1) || !A || ncol < 6 || (size_t)ncol * (ncol + 1) > (size_t)c->rstride || n_accepted < 0) return INGVIO_E_ARG;
    std::vector<int> used(c->G, 0);
    used[0] = n_accepted;
    int rc = up(c, c->d_Rpart + (size_t)b * c->G * c->rstride, A, 8 * (size_t)ncol * (ncol + 1));
    rc |= up(c, c->d_chunk_used + (size_t)b * c->G, used.data(), sizeof(int) * (size_t)c->G);
    if (rc) return INGVIO_E_HIP;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return INGVIO_OK;
}

// ---- device-resident exchange of the feature-sharded filter (the ONE exchange step per frame) ---------------------
// ingvio_info_reduce sums filter b's chunk partials into one contiguous device buffer [A | b | n_accepted] (ncol (ncol + 1) + 1
// doubles; the accepted count rides along as a double so that ONE all-reduce carries everything) and hands out the DEVICE
// pointer: the caller all-reduces it in place (RCCL on the pointer itself, or torch.distributed on a zero-copy view,
// ingvio_amd/parallel.py) - nothing crosses PCIe.  ingvio_info_commit makes the reduced buffer the filter's information
// (chunk 0 = the sum, its count = the summed n_accepted) for ingvio_frame_run_phase(.., 2).
namespace {
__global__ __launch_bounds__(256) void k_info_reduce(const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
                                                     int cnt, double* __restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e > cnt) return;
    double s = 0.0;
    if (e < cnt) { for (int g = 0; g < G; ++g) if (chunk_used[g]) s += Apart[(size_t)g * rstride + e]; }
    else { for (int g = 0; g < G; ++g) s += (double)chunk_used[g]; }
    out[e] = s;
}
__global__ __launch_bounds__(256) void k_info_commit(const double* __restrict__ in, int cnt, double* __restrict__ Apart0, int* __restrict__ chunk_used, int G)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < cnt) Apart0[e] = in[e];
    if (e < G) chunk_used[e] = e == 0 ? (int)llrint(in[cnt]) : 0;
}
}  // namespace

int ingvio_info_reduce(ingvio_ctx* c, int b, double** dev_ptr, int* count, int* ncol_out)
{
    ENTER(c);
    if (check_range(c, b, 1) || !dev_ptr || !count || c->method != 1) return INGVIO_E_ARG;
    int C = 0;
    if (down_sync(c, &C, c->d_nclones + b, sizeof(int))) return INGVIO_E_HIP;
    const int ncol = 6 * C, cnt = ncol * (ncol + 1);
    if (cnt <= 0 || cnt > c->rstride) return INGVIO_E_ARG;
    if (!c->d_xchg && dalloc(c, &c->d_xchg, (size_t)c->d.batch * (c->rstride + 8))) return INGVIO_E_HIP;
    double* out = c->d_xchg + (size_t)b * (c->rstride + 8);
    hipLaunchKernelGGL(k_info_reduce, dim3((cnt + 256) / 256), dim3(256), 0, c->st, c->d_Rpart + (size_t)b * c->G * c->rstride,
                       c->d_chunk_used + (size_t)b * c->G, c->G, c->rstride, cnt, out);
    HIPCHK(c, hipStreamSynchronize(c->st));              // the caller's collective runs on ITS stream: the buffer must be complete
    *dev_ptr = out; *count = cnt + 1;
    if (ncol_out) *ncol_out = ncol;
    return last_launch(c);
}

int ingvio_info_commit(ingvio_ctx* c, int b)
{
    ENTER(c);
    if (check_range(c, b, 1) || !c->d_xchg || c->method != 1) return INGVIO_E_ARG;
    int C = 0;
    if (down_sync(c, &C, c->d_nclones + b, sizeof(int))) return INGVIO_E_HIP;
    const int ncol = 6 * C, cnt = ncol * (ncol + 1);
    if (cnt <= 0 || cnt > c->rstride) return INGVIO_E_ARG;
    hipLaunchKernelGGL(k_info_commit, dim3((std::max(cnt, c->G) + 255) / 256), dim3(256), 0, c->st, c->d_xchg + (size_t)b * (c->rstride + 8), cnt,
                       c->d_Rpart + (size_t)b * c->G * c->rstride, c->d_chunk_used + (size_t)b * c->G, c->G);
    return last_launch(c);
}

// The frame's results travel through the pinned mirror of the result slab (h_result): a device-to-host copy into pageable caller memory
// is staged by the runtime in small synchronous pieces (1.3 MB of dx + flags for 512 filters: 0.2 ms during which the device idles).
// _begin only ENQUEUES the copies behind the frame's kernels and records an event; _end waits for it and hands the data out - between
// the two the host may stage and launch the next frame (the copies sit in front of its kernels in the stream, so the slab is read
// before it is rewritten).
int ingvio_frame_fetch_begin(ingvio_ctx* c, int b0, int nb)
{
    ENTER(c);
    if (check_range(c, b0, nb)) return INGVIO_E_ARG;
    const int fm = c->d.f_max;
    if (!c->ev_fetch) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fetch, hipEventDisableTiming));
    HIPCHK(c, hipMemcpyAsync(c->h_result + 8 * (size_t)b0 * c->ldp, c->d_dx + (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(c->h_result + c->ro_used + sizeof(int) * (size_t)b0 * fm, c->d_used + (size_t)b0 * fm, sizeof(int) * (size_t)nb * fm, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(c->h_result + c->ro_m + sizeof(int) * (size_t)b0, c->d_m + b0, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipEventRecord(c->ev_fetch, c->st));
    c->fetch_b0 = b0; c->fetch_nb = nb;
    return last_launch(c);
}

int ingvio_frame_fetch_end(ingvio_ctx* c, double* dx_out, int* accepted, int* rows_out)
{
    if (!c || c->fetch_nb <= 0) return INGVIO_E_ARG;                       // (no ENTER: the results are already on their way)
    const int fm = c->d.f_max, b0 = c->fetch_b0, nb = c->fetch_nb;
    HIPCHK(c, hipEventSynchronize(c->ev_fetch));
    c->fetch_nb = 0;
    if (dx_out) memcpy(dx_out, c->h_result + 8 * (size_t)b0 * c->ldp, 8 * (size_t)nb * c->ldp);
    if (accepted) memcpy(accepted, c->h_result + c->ro_used + sizeof(int) * (size_t)b0 * fm, sizeof(int) * (size_t)nb * fm);
    if (rows_out) memcpy(rows_out, c->h_result + c->ro_m + sizeof(int) * (size_t)b0, sizeof(int) * (size_t)nb);
    return INGVIO_OK;
}

int ingvio_frame_fetch(ingvio_ctx* c, int b0, int nb, double* dx_out, int* accepted, int* rows_out)
{
    int rc = ingvio_frame_fetch_begin(c, b0, nb);
    if (rc) return rc;
    rc = ingvio_frame_fetch_end(c, dx_out, accepted, rows_out);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->st));                              // as before: the call leaves the stream drained
    return last_launch(c);
}

int ingvio_set_msckf_method(ingvio_ctx* c, int method)
{
    ENTER(c);
    if (!c || method < 0 || method > 1) return INGVIO_E_ARG;
    c->method = method;
    return INGVIO_OK;
}

// parity hook: [A | b] = [sum_j H_j^T H_j | sum_j H_j^T r_j] of the last MSCKF update of filter b, ncol x (ncol + 1) row-major
int ingvio_debug_msckf_info(ingvio_ctx* c, int b, double* A_out, int* ncol_out)
{
    ENTER(c);
    if (check_range(c, b, 1) || !A_out || !ncol_out) return INGVIO_E_ARG;
    int C = 0;
    if (down_sync(c, &C, c->d_nclones + b, sizeof(int))) return INGVIO_E_HIP;
    const int ncol = 6 * C;
    *ncol_out = ncol;
    const size_t w = (size_t)ncol + 1;
    for (size_t e = 0; e < (size_t)ncol * w; ++e) A_out[e] = 0.0;
    if (c->method == 1) {
        std::vector<int> used(c->G);
        if (down_sync(c, used.data(), c->d_chunk_used + (size_t)b * c->G, sizeof(int) * (size_t)c->G)) return INGVIO_E_HIP;
        std::vector<double> part((size_t)ncol * w);
        for (int g = 0; g < c->G; ++g) {
            if (!used[g]) continue;
            if (down_sync(c, part.data(), c->d_Rpart + ((size_t)b * c->G + g) * c->rstride, 8 * part.size())) return INGVIO_E_HIP;
            for (size_t e = 0; e < part.size(); ++e) A_out[e] += part[e];
        }
        return INGVIO_OK;
    }
    // dense method: the merged factor R (upper triangular, column-major ld = mld) and Q^T r in d_H / d_res
    int m = 0;
    if (down_sync(c, &m, c->d_m + b, sizeof(int))) return INGVIO_E_HIP;
    if (m == 0) return INGVIO_OK;
    std::vector<double> R((size_t)c->mld * ncol), z(c->mld);
    if (down_sync(c, R.data(), c->d_H + (size_t)b * c->hstride, 8 * R.size()) || down_sync(c, z.data(), c->d_res + (size_t)b * c->mld, 8 * (size_t)ncol))
        return INGVIO_E_HIP;
    for (int i = 0; i < ncol; ++i) {
        for (int j = 0; j < ncol; ++j) {
            double a = 0.0;
            for (int k = 0; k <= (i < j ? i : j); ++k) a += R[k + (size_t)i * c->mld] * R[k + (size_t)j * c->mld];
            A_out[i * w + j] = a;
        }
        double bb = 0.0;
        for (int k = 0; k <= i; ++k) bb += R[k + (size_t)i * c->mld] * z[k];
        A_out[i * w + ncol] = bb;
    }
    return INGVIO_OK;
}

// parity hook: [M | t] handed from the information solve to the apply kernel (MP x MP row-major, then MP entries), filter b
int ingvio_debug_info_solution(ingvio_ctx* c, int b, double* out, int count)
{
    ENTER(c);
    if (check_range(c, b, 1) || !out || count < 1 || count > c->ystride) return INGVIO_E_ARG;
    if (down_sync(c, out, c->d_Y + (size_t)b * c->ystride, 8 * (size_t)count)) return INGVIO_E_HIP;
    return INGVIO_OK;
}

int ingvio_debug_read(ingvio_ctx* c, long long* out, int n)
{
    ENTER(c);
    if (!c || !out || n < 1 || n > 64) return INGVIO_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->st));
    long long a[64], bq[64], cq[64], sq[64];
    if (dbg_read_factored(a, 64) || dbg_read_cov(bq, 64) || dbg_read_bigwin(cq, 64) || dbg_read_solve(sq, 64)) return INGVIO_E_HIP;
    long long hq[64];
    if (dbg_read_chol(hq, 64)) return INGVIO_E_HIP;
    long long lq[64], mq[64];
    if (dbg_read_lmbatch(lq, 64) || dbg_read_lmchol(mq, 64)) return INGVIO_E_HIP;
    if (const char* e = getenv("INGVIO_DBG_TU")) {                     // debugging: all slots of one translation unit
        const long long* src = e[0] == 'b' ? cq : (e[0] == 'c' ? hq : (e[0] == 's' ? sq : (e[0] == 'l' ? lq : (e[0] == 'm' ? mq : a))));
        for (int i = 0; i < n; ++i) out[i] = src[i];
        return INGVIO_OK;
    }
    for (int i = 0; i < n; ++i)      // 16..23 cov, 24..31 the symmetric solve (its slots 0..7), 48..55 large-window TU, 56..63 kernels_chol (0..7)
        out[i] = i >= 56 ? hq[i - 56] : (i >= 48 ? cq[i] : ((i >= 24 && i < 32) ? sq[i - 24] : ((i < 16 || i >= 32) ? a[i] : bq[i])));
    return INGVIO_OK;
}

int ingvio_profile_select(ingvio_ctx* c, const char* kernel_name)
{
    ENTER(c);
    if (!c) return INGVIO_E_ARG;
    c->prof_only = -1;
    if (!kernel_name || !*kernel_name) return INGVIO_OK;
    for (int i = 0; i < PF_COUNT; ++i) if (!strcmp(kernel_name, kProfNames[i])) { c->prof_only = i; return INGVIO_OK; }
    return INGVIO_E_ARG;
}

int ingvio_profile_enable(ingvio_ctx* c, int enable)
{
    ENTER(c);
    if (!c) return INGVIO_E_ARG;
    c->prof = enable != 0;
    return INGVIO_OK;
}

static void prof_collect(ingvio_ctx* c)
{
    hipStreamSynchronize(c->st);
    for (auto& r : c->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.id] += ms; c->prof_calls[r.id] += 1; }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    c->recs.clear();
}

int ingvio_profile_reset(ingvio_ctx* c)
{
    ENTER(c);
    if (!c) return INGVIO_E_ARG;
    prof_collect(c);
    memset(c->prof_ms, 0, sizeof c->prof_ms); memset(c->prof_calls, 0, sizeof c->prof_calls);
    return INGVIO_OK;
}

int ingvio_profile_get(ingvio_ctx* c, const char** names, double* ms, int* calls, int cap)
{
    ENTER(c);
    if (!c || !names || !ms || !calls) return INGVIO_E_ARG;
    prof_collect(c);
    int k = 0;
    for (int i = 0; i < PF_COUNT && k < cap; ++i) { names[k] = kProfNames[i]; ms[k] = c->prof_ms[i]; calls[k] = c->prof_calls[i]; ++k; }
    return k;
}

}  // extern "C"
