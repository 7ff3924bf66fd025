// ek_api.hip -- the C ABI of include/enspara_hip.h (host side).
#include "ek_common.h"
#include "ek_pam_sparse.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

void ek_msm_scratch_free(void *w);      // ek_msm.hip

static thread_local char g_err[512] = "";

// shared with ek_msm.hip
int ek_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define ek_fail ek_set_error

#define EK_HIP(call)                                                           \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess)                                                  \
            return ek_fail(EK_EHIP, "%s failed: %s (%d) at %s:%d", #call,      \
                           hipGetErrorString(e_), (int)e_, __FILE__,           \
                           __LINE__);                                          \
    } while (0)

#define EK_CHECK_LAUNCH()                                                      \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess)                                                  \
            return ek_fail(EK_EHIP, "kernel launch failed: %s (%d) at %s:%d",  \
                           hipGetErrorString(e_), (int)e_, __FILE__,           \
                           __LINE__);                                          \
    } while (0)

// A few host threads that copy slices of a chunk into pinned memory
// (ek_load_frames); they live as long as the context that first needed them.
struct EkCopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    const char *src = nullptr;
    char *dst = nullptr;
    size_t bytes = 0, per = 0;
    int next = 0, n_parts = 0, left = 0;
    bool quit = false;
    void worker()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || next < n_parts; });
            if (quit)
                return;
            const int part = next++;
            const size_t lo = std::min(bytes, (size_t)part * per);
            const size_t hi = std::min(bytes, lo + per);
            const char *s_ = src;
            char *d_ = dst;
            lk.unlock();
            if (hi > lo)
                memcpy(d_ + lo, s_ + lo, hi - lo);
            lk.lock();
            if (--left == 0)
                done_cv.notify_all();
        }
    }
    void start(int n)
    {
        for (int i = (int)th.size(); i < n; ++i)
            th.emplace_back([this] { worker(); });
    }
    // copy `n` bytes in slices of 2 MiB-aligned size, all threads; returns when done
    void copy(char *d, const char *s, size_t n)
    {
        const int parts = (int)th.size();
        std::unique_lock<std::mutex> lk(mu);
        src = s;
        dst = d;
        bytes = n;
        per = (n / parts + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
        next = 0;
        n_parts = left = parts;
        cv.notify_all();
        done_cv.wait(lk, [&] { return left == 0; });
        n_parts = 0;
    }
    ~EkCopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        for (auto &t : th)
            t.join();
    }
};

struct ek_ctx {
    int device = 0;
    int64_t n = 0;
    int32_t A = 0;
    int64_t goff = 0;
    int64_t n_tiles = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool loaded = false;

    float *tiles = nullptr;      // [n_tiles][3A][EK_TILE]
    float *qtiles = nullptr;     // quad copy [n_tiles][ceil(A/4)][3][EK_TILE][4]: what the
                                 // 16-candidate pass streams; made when one first runs
    bool qt_valid = false;       //   (ek_ensure_qtiles), again after frames are loaded
    float *aos = nullptr;        // [n][3A] the same centred frames, frame-major
    double *G = nullptr;         // [n]
    float *dist = nullptr;       // [n]
    int32_t *assign = nullptr;   // [n]
    float *scratch = nullptr;    // [n]   distances-only output
    unsigned char *rec = nullptr;     // own candidate record
    unsigned char *rec_tmp = nullptr; // record of an explicit center
    EkBlockMax *blockmax = nullptr;
    int blockmax_cap = 0;
    EkHist *hist = nullptr;
    int32_t hist_cap = 0;
    EkCtl *ctl = nullptr;

    // host uploads (ek_load_frames): two pinned host buffers filled by a few
    // threads, two device staging buffers, an event per pair
    float *stage[2] = {nullptr, nullptr};    // AoS staging on the device
    float *pin[2] = {nullptr, nullptr};      // pinned host memory
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    bool up_busy[2] = {false, false};        // the event of the pair was recorded
    EkCopyPool *pool = nullptr;
    int64_t stage_frames = 0;
    float *cen_aos = nullptr;    // centred center-major centers
    double *cen_G = nullptr;
    int32_t cen_cap = 0;
    float *cen_tiles = nullptr;  // the same centers, frame-minor tiles
    int32_t cen_tiles_cap = 0;   // in centers (multiple of EK_TILE)
    int assign_variant = 0;      // 0 auto, 1 vector FMA, 2 MFMA

    // PAM working set (allocated by ek_pam_begin)
    float *ndist = nullptr;
    int32_t *nassign = nullptr;
    uint32_t *amb = nullptr;
    unsigned long long *amb_best = nullptr;
    unsigned int *amb_count = nullptr;
    int32_t *blockcnt = nullptr;
    int64_t *scan = nullptr;
    int64_t *sel = nullptr;          // [0] member count, [1] selected frame
    double *sq_part = nullptr;       // leaf sums + chunk sums (ek_pam.hip, numpy's order)
    EkPwShape *pw_shapes = nullptr;  // [2]: a full chunk, the last chunk
    int pw_n_full = 0, pw_leaves = 0, pw_chunks = 0;
    bool pw_tail_ok = false;         // full chunks are perfect 64-leaf trees
    double *sq_out = nullptr;
    float *med_aos = nullptr;        // [K+1][3A]; row K = saved row
    double *med_G = nullptr;
    int64_t *med_idx = nullptr;      // [K+1] device copy of medoid frames
    float *ambt = nullptr;           // [3A][ambt_cap] compacted ambiguous frames
    double *ambG = nullptr;
    int64_t ambt_cap = 0;
    int32_t med_K = 0, med_cap = 0;
    int32_t pam_cid = -1;            // proposal pending commit
    int32_t cnt_cid = -1;            // cluster of the last member count
    int64_t cnt_m = 0;
    int64_t pam_frame = -1;
    // proposal prefetch: member lists of a window of clusters and the distance
    // vectors of up to EK_PAM_WIN proposed frames
    int32_t *bat_blockcnt = nullptr; // [EK_PAM_WIN][nb]
    int64_t *bat_scan = nullptr;     // [EK_PAM_WIN][nb]
    int64_t *bat_sel = nullptr;      // [0..8) counts, [8..16) selected frames
    int32_t bat_cid0 = -1, bat_count = 0;
    float *pam_vecs = nullptr;       // [EK_PAM_WIN][n_pad]
    unsigned char *pam_recs = nullptr;
    EkPlan *pam_plan = nullptr;
    unsigned int *moved = nullptr;
    int64_t pf_frames[EK_PAM_WIN];
    int32_t pf_count = 0;
    bool pf_external = false;        // slots hold caller-supplied centers
    EkPamOut *pam_out_dev = nullptr; // result record of a proposal
    EkPamOut *pam_out_host = nullptr;    // pinned copy the host polls for
    EkPamWin *pam_win_dev = nullptr;     // a window of proposals decided on the device
    EkPamWin *pam_win_host = nullptr;    // pinned
    int32_t pam_restore = -1;        // row of the trial table a rejected proposal left
    int32_t *med_list = nullptr;     // [med_cap] medoids within reach (ek_pam_prune_kernel)
    float *dtab = nullptr;           // window tables, three blocks of EK_PAM_WIN * (med_cap + 1):
                                     // T medoid-to-proposal, O medoid-to-old-medoid, dmin
    int32_t tab_lo = -1, tab_n = 0;  // the window (first cluster, slots) the tables were made for
    unsigned int *act_n_host = nullptr;  // pinned
    int64_t pf_sparse = 0, pf_full = 0;  // prefetch passes of either kind
    int32_t pf_backoff = 0;          // windows to go before the restricted form is tried again
    int prune = 1;                   // use it (option key 6)
    bool state_exact = true;         // dist[f] IS the distance to medoid assign[f]
    int64_t *tmp_idx = nullptr;      // scratch for index lists
    int64_t tmp_idx_cap = 0;
    int64_t pf_hits = 0, pf_misses = 0;
    // windows worked through by one workgroup (ek_pam_sparse.hip)
    int pam_sparse = 1;              // use them where they apply (option key 12)
    int64_t sp_max_pairs = EK_SP_MAX_PAIRS;  // (option key 13)
    int sp_exact = 0;                // (option key 14)
    bool sp_ready = false;           // act_list holds the list of the window just prefetched
    uint32_t *act_list = nullptr;    // [n] the frames a window's proposals can touch
    int64_t vecs_rows = -1;          // >= 0: pam_vecs is +inf except at act_list[0 .. vecs_rows)
    int32_t vecs_cols = 0;           //   of its first vecs_cols vectors
    int64_t sp_nact = 0;
    unsigned char *sp_buf = nullptr; // the slots' buckets and their lengths
    int64_t sp_windows = 0, sp_bailed = 0;
    int32_t sp_backoff = 0, sp_backoff_next = 8;    // windows to go the three-launch way after one ended early

    // multi-candidate rounds (ek_spec.hip)
    int cands = -1;              // candidates per pass: -1 auto, 1 = one-center passes
    unsigned char *recsT = nullptr;   // EK_MAX_CANDS records
    EkPlan *plan = nullptr;
    float *vecs = nullptr;       // [EK_MAX_CANDS-1][n_pad] stored distance vectors
    EkMaxHdr *hdr = nullptr;
    EkBlockMax *pm = nullptr;    // [EK_MAX_CANDS-1][nb] per-prefix maxima (ek_chain.hip)
    unsigned char *top = nullptr;    // scratch of the candidate pick (ek_spec.hip)
    float *planD = nullptr;          // [64][64] distances between the records on offer
    int fused = 1;               // single-shard rounds in three launches (ek_round.hip)
    int tri = 0;                 // triangle-inequality tile skip (one-center steps)
    float *ti_D = nullptr;       // [ti_cap] distances of the existing centers to the new one
    int32_t ti_cap = 0;
    uint8_t *ti_skip = nullptr;  // [n_tiles]
    unsigned long long *ti_stats = nullptr;  // [2] tiles looked at, skipped
    int64_t ti_tiles = 0, ti_skipped = 0;    // of the last run
    float *ti_tab = nullptr;     // sharded steps: the accepted centers, [ti_tab_cap][3A]
    double *ti_tabG = nullptr;
    int32_t ti_tab_cap = 0;
    int32_t ti_tab_n = 0;        // rows 0 .. ti_tab_n - 1 are the centers of labels 0 ..
    EkPend *pend = nullptr;      // accepted chain not yet applied
    EkChainOrd *ord = nullptr;
    EkChainRow *rows = nullptr;      // [EK_MAX_CANDS] candidate frames' rows
    uint32_t *vmask = nullptr;       // [n_pad / 64] which vectors a wave stored
    unsigned int *tick = nullptr;    // [256] arrival counters: [0] pass, [1] chain,
                                     // [2] + [64..96) next, [3] legacy chain maxima,
                                     // [5] [6] multi-shard helpers, [128] + [129..161) PAM
    float *ctile = nullptr;      // the round's candidates, [atom][pair][xyz][2]
    double *ctrace = nullptr;    // their traces
    int chain = 1;               // 1: chained cheap steps, 0: one launch pair per center
    int64_t n_pad = 0;
    int32_t last_passes = 0;
    int adapt = 1;               // choose the candidates per pass from measured rates
    int64_t st_rounds[4] = {0, 0, 0, 0};    // passes run as 1 / 4 / 8 / 16 candidates (last run)
    int64_t st_centers[4] = {0, 0, 0, 0};   // centers they accepted
    hipEvent_t evb0 = nullptr, evb1 = nullptr;   // per-batch timing

    // rounds across shards (ek_mshard.hip)
    EkMsState *ms = nullptr;         // device-side state
    unsigned char *ms_mbox = nullptr;    // own mailbox area [2][world][msg]
    uint32_t *ms_flags = nullptr;        // own flags [2][world][16]
    EkMsXchg ms_x;                   // transport (host copy, passed by value)
    int ms_peers = 0;                // peers connected (mailbox transport on at == world)
    std::vector<void *> ms_ipc;      // mappings opened with hipIpcOpenMemHandle
    int ms_T = 0;                    // candidates per pass of the run in progress

    void *msm_scratch = nullptr;     // ek_msm.hip: buffers of ek_msm_counts_ctx

    int fpl = 0;                 // 0 = auto
    int nt = -1;                 // non-temporal frame loads: -1 = auto
    // sampled per-launch timing of the distance kernel (bench only)
    std::vector<hipEvent_t> samp_ev;
    std::vector<int> samp_form;  // candidates per pass of each sampled launch
    int samp_dom = 0;            // the form most samples of the last timing had
    int samp_every = 0;
    int samp_used = 0;
    int64_t samp_count = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    int32_t last_launches = 0;
};

void ek_ctx_msm_view(ek_ctx *c, int *device, int64_t *n, const int32_t **assign,
                     hipStream_t *stream, void ***scratch_slot)
{
    *device = c->device;
    *n = c->loaded ? c->n : -1;
    *assign = c->assign;
    *stream = c->stream;
    *scratch_slot = &c->msm_scratch;
}

// Wait for the stream by polling.  The loops that read a few bytes back per
// step (PAM proposals, the k-centers progress checks) use this:
// hipStreamSynchronize may put the thread to sleep and a wake-up costs anything
// from 0.1 ms to tens of ms on a busy host -- more than the step itself.
static hipError_t ek_wait(ek_ctx *c)
{
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e != hipErrorNotReady)
            return e;
    }
}

static int ek_pick_fpl(const ek_ctx *c)
{
    if (c->fpl == 1 || c->fpl == 2 || c->fpl == 4)
        return c->fpl;
    // enough waves to cover 256 CUs x 4 SIMDs several times over, widest loads
    // that still leave that many
    const int64_t waves4 = (c->n + 255) / 256;
    const int64_t waves2 = (c->n + 127) / 128;
    if (waves4 >= 4096)
        return 4;
    if (waves2 >= 4096)
        return 2;
    return 1;
}

// Non-temporal loads for the frame stream unless the whole shard could stay
// resident in the 256 MiB Infinity Cache between two passes.
static int ek_pick_nt(const ek_ctx *c)
{
    if (c->nt >= 0)
        return c->nt;
    const size_t bytes = (size_t)c->n_tiles * 3 * (size_t)c->A * EK_TILE * 4;
    return bytes > ((size_t)192 << 20) ? 1 : 0;
}

// the widest form of a round this context may use (candidates per pass):
// EK_MAX_CANDS unless option key 4 pins it; 1 = one-center passes only
static int ek_pick_cands(const ek_ctx *c)
{
    const int t = c->cands == -1 ? EK_MAX_CANDS : c->cands;
    return (t == 16 || t == 8 || t == 4) ? t : 1;
}

// The quad copy of the frames (ek_pass16.hip) is made when a 16-candidate pass
// first needs it and again after frames were loaded: a third copy of the
// coordinates (12 A bytes per frame, 3.6 GB at 10^6 x 300 of the 288 GB), one
// read and one write of the shard.
static int ek_ensure_qtiles(ek_ctx *c)
{
    if (c->qt_valid)
        return EK_OK;
    if (!c->qtiles)
        EK_HIP(hipMalloc((void **)&c->qtiles,
                         ek_quad_tiles_bytes(std::max<int64_t>(c->n_tiles, 1), c->A)));
    ek_launch_quad_tiles(c->tiles, c->n_tiles, c->A, c->qtiles, c->stream);
    EK_CHECK_LAUNCH();
    c->qt_valid = true;
    return EK_OK;
}

// slot of a form in the run statistics: passes run as 1 / 4 / 8 / 16 candidates
static int ek_form_slot(int T) { return T <= 1 ? 0 : (T == 4 ? 1 : (T == 8 ? 2 : 3)); }

static int ek_spec_alloc(ek_ctx *c)
{
    if (!c->top)
        EK_HIP(hipMalloc((void **)&c->top, ek_top_scratch_bytes(c->A)));
    if (!c->planD)
        EK_HIP(hipMalloc((void **)&c->planD, 64 * 64 * sizeof(float)));
    if (!c->pm) {
        const size_t nb = ((size_t)std::max<int64_t>(c->n, 1) + EK_BLOCK - 1) /
                          EK_BLOCK;
        EK_HIP(hipMalloc((void **)&c->pm,
                         (size_t)(EK_MAX_CANDS - 1) * nb * sizeof(EkBlockMax)));
    }
    if (!c->vecs) {
        EK_HIP(hipMalloc((void **)&c->vecs, (size_t)(EK_MAX_CANDS - 1) *
                                                std::max<int64_t>(c->n_pad, 1) *
                                                sizeof(float)));
    }
    return EK_OK;
}

extern "C" int ek_abi_version(void) { return EK_ABI_VERSION; }
extern "C" const char *ek_last_error(void) { return g_err; }

extern "C" int ek_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return ek_fail(EK_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" size_t ek_record_bytes(int32_t n_atoms) { return ek_rec_bytes(n_atoms); }

static int ek_free_all(ek_ctx *c)
{
    if (!c)
        return EK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->tiles);
    (void)hipFree(c->qtiles);
    (void)hipFree(c->aos);
    (void)hipFree(c->G);
    (void)hipFree(c->dist);
    (void)hipFree(c->assign);
    (void)hipFree(c->scratch);
    (void)hipFree(c->rec);
    (void)hipFree(c->rec_tmp);
    (void)hipFree(c->blockmax);
    (void)hipFree(c->hist);
    (void)hipFree(c->ctl);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(c->stage[b]);
        (void)hipHostFree(c->pin[b]);
        if (c->up_ev[b])
            (void)hipEventDestroy(c->up_ev[b]);
    }
    delete c->pool;
    (void)hipFree(c->cen_aos);
    (void)hipFree(c->cen_G);
    (void)hipFree(c->cen_tiles);
    (void)hipFree(c->bat_blockcnt);
    (void)hipFree(c->bat_scan);
    (void)hipFree(c->bat_sel);
    (void)hipFree(c->pam_vecs);
    (void)hipFree(c->pam_recs);
    (void)hipFree(c->pam_plan);
    (void)hipFree(c->moved);
    (void)hipFree(c->tmp_idx);
    (void)hipFree(c->pam_out_dev);
    (void)hipFree(c->med_list);
    (void)hipFree(c->dtab);
    if (c->act_n_host)
        (void)hipHostFree(c->act_n_host);
    if (c->pam_out_host)
        (void)hipHostFree(c->pam_out_host);
    (void)hipFree(c->pam_win_dev);
    (void)hipFree(c->sp_buf);
    (void)hipFree(c->act_list);
    if (c->pam_win_host)
        (void)hipHostFree(c->pam_win_host);
    ek_msm_scratch_free(c->msm_scratch);
    for (void *m : c->ms_ipc)
        (void)hipIpcCloseMemHandle(m);
    (void)hipFree(c->ms);
    (void)hipFree(c->ms_mbox);
    (void)hipFree(c->ms_flags);
    (void)hipFree(c->recsT);
    (void)hipFree(c->ctile);
    (void)hipFree(c->ctrace);
    (void)hipFree(c->ti_D);
    (void)hipFree(c->ti_tab);
    (void)hipFree(c->ti_tabG);
    (void)hipFree(c->ti_skip);
    (void)hipFree(c->ti_stats);
    (void)hipFree(c->pend);
    (void)hipFree(c->ord);
    (void)hipFree(c->tick);
    (void)hipFree(c->rows);
    (void)hipFree(c->vmask);
    (void)hipFree(c->plan);
    (void)hipFree(c->vecs);
    (void)hipFree(c->hdr);
    (void)hipFree(c->pm);
    (void)hipFree(c->top);
    (void)hipFree(c->planD);
    (void)hipFree(c->ndist);
    (void)hipFree(c->nassign);
    (void)hipFree(c->amb);
    (void)hipFree(c->amb_best);
    (void)hipFree(c->amb_count);
    (void)hipFree(c->blockcnt);
    (void)hipFree(c->scan);
    (void)hipFree(c->sel);
    (void)hipFree(c->sq_part);
    (void)hipFree(c->pw_shapes);
    (void)hipFree(c->sq_out);
    (void)hipFree(c->med_aos);
    (void)hipFree(c->med_G);
    (void)hipFree(c->med_idx);
    (void)hipFree(c->ambt);
    (void)hipFree(c->ambG);
    for (hipEvent_t e : c->samp_ev)
        (void)hipEventDestroy(e);
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    if (c->evb0)
        (void)hipEventDestroy(c->evb0);
    if (c->evb1)
        (void)hipEventDestroy(c->evb1);
    if (c->own_stream && c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return EK_OK;
}

extern "C" int ek_ctx_create(int device, int64_t n_frames, int32_t n_atoms,
                             int64_t global_offset, void *stream, ek_ctx **out)
{
    if (!out)
        return ek_fail(EK_EARG, "ek_ctx_create: out is NULL");
    *out = nullptr;
    if (n_frames < 0 || n_frames > 0x7fffff00LL)
        return ek_fail(EK_EARG, "ek_ctx_create: n_frames=%lld out of range",
                       (long long)n_frames);
    if (n_atoms < 1 || n_atoms > EK_MAX_ATOMS)
        return ek_fail(EK_EARG, "ek_ctx_create: n_atoms=%d out of range [1,%d]",
                       n_atoms, EK_MAX_ATOMS);
    EK_HIP(hipSetDevice(device));
    ek_ctx *c = new (std::nothrow) ek_ctx();
    if (!c)
        return ek_fail(EK_ENOMEM, "ek_ctx_create: out of host memory");
    c->device = device;
    c->n = n_frames;
    c->A = n_atoms;
    c->goff = global_offset;
    c->n_tiles = (n_frames + EK_TILE - 1) / EK_TILE;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return ek_fail(EK_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    const size_t tile_floats = (size_t)3 * n_atoms * EK_TILE;
    const size_t nn = (size_t)std::max<int64_t>(n_frames, 1);
    const size_t nt = (size_t)std::max<int64_t>(c->n_tiles, 1);
    const size_t recb = ek_rec_bytes(n_atoms);
    c->blockmax_cap = (int)((n_frames + EK_BLOCK - 1) / EK_BLOCK) + 1;
    c->hist_cap = 1024;
    hipError_t e = hipSuccess;
#define EK_ALLOC(ptr, bytes)                                                   \
    if (e == hipSuccess)                                                       \
        e = hipMalloc((void **)&(ptr), (bytes));
    EK_ALLOC(c->tiles, nt * tile_floats * sizeof(float));
    EK_ALLOC(c->aos, nn * (size_t)3 * n_atoms * sizeof(float));
    EK_ALLOC(c->G, nn * sizeof(double));
    EK_ALLOC(c->dist, nn * sizeof(float));
    EK_ALLOC(c->assign, nn * sizeof(int32_t));
    EK_ALLOC(c->scratch, nn * sizeof(float));
    EK_ALLOC(c->rec, recb);
    EK_ALLOC(c->rec_tmp, recb);
    EK_ALLOC(c->blockmax, (size_t)c->blockmax_cap * sizeof(EkBlockMax));
    EK_ALLOC(c->hist, (size_t)c->hist_cap * sizeof(EkHist));
    EK_ALLOC(c->ctl, sizeof(EkCtl));
    EK_ALLOC(c->recsT, recb * EK_MAX_CANDS);
    EK_ALLOC(c->plan, sizeof(EkPlan));
    EK_ALLOC(c->hdr, sizeof(EkMaxHdr));
    EK_ALLOC(c->pend, sizeof(EkPend));
    EK_ALLOC(c->ord, sizeof(EkChainOrd));
    EK_ALLOC(c->tick, 256 * sizeof(unsigned int));
    EK_ALLOC(c->rows, EK_MAX_CANDS * sizeof(EkChainRow));
    EK_ALLOC(c->vmask, (nt * EK_TILE / EK_WAVE) * sizeof(uint32_t));
    EK_ALLOC(c->ctile, ek_ctile_bytes(n_atoms));
    EK_ALLOC(c->ctrace, EK_MAX_CANDS * sizeof(double));
#undef EK_ALLOC
    c->n_pad = (int64_t)nt * EK_TILE;
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev0);
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev1);
    if (e == hipSuccess && c->n_tiles > 0)  // zero the (padded) last tile
        e = hipMemsetAsync(c->tiles + (size_t)(c->n_tiles - 1) * tile_floats, 0,
                           tile_floats * sizeof(float), c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->hist, 0, (size_t)c->hist_cap * sizeof(EkHist),
                           c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->ctl, 0, sizeof(EkCtl), c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->rec, 0, recb, c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->pend, 0, sizeof(EkPend), c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->ord, 0, sizeof(EkChainOrd), c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->tick, 0, 256 * sizeof(unsigned int), c->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->rows, 0, EK_MAX_CANDS * sizeof(EkChainRow),
                           c->stream);
    if (e != hipSuccess) {
        ek_free_all(c);
        return ek_fail(e == hipErrorOutOfMemory ? EK_ENOMEM : EK_EHIP,
                       "ek_ctx_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return EK_OK;
}

extern "C" int ek_ctx_destroy(ek_ctx *ctx) { return ek_free_all(ctx); }

extern "C" int ek_ctx_sync(ek_ctx *c)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    EK_HIP(hipStreamSynchronize(c->stream));
    return EK_OK;
}

extern "C" void *ek_ctx_stream(ek_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" void *ek_own_record(ek_ctx *c) { return c ? (void *)c->rec : nullptr; }

extern "C" int ek_set_frames_per_lane(ek_ctx *c, int fpl)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (fpl != 0 && fpl != 1 && fpl != 2 && fpl != 4)
        return ek_fail(EK_EARG, "frames per lane must be 0, 1, 2 or 4");
    c->fpl = fpl;
    return EK_OK;
}

extern "C" int ek_set_option(ek_ctx *c, int32_t key, int32_t value)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    switch (key) {
    case 1:
        c->nt = value < 0 ? -1 : (value ? 1 : 0);
        return EK_OK;
    case 4:
        if (value != -1 && value != 1 && value != 4 && value != 8 && value != 16)
            return ek_fail(EK_EARG, "ek_set_option: candidates per pass must "
                                    "be -1 (auto), 1, 4, 8 or 16");
        c->cands = value;
        return EK_OK;
    case 5:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: chained rounds 0 or 1");
        c->chain = value;
        return EK_OK;
    case 6:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: PAM medoid pruning 0 or 1");
        c->prune = value;
        return EK_OK;
    case 7:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: state-is-exact flag 0 or 1");
        c->state_exact = value != 0;
        return EK_OK;
    case 11:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: triangle inequality 0 or 1");
        c->tri = value;
        return EK_OK;
    case 10:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: fused rounds 0 or 1");
        c->fused = value;
        return EK_OK;
    case 9:     // (round 1's LDS form of the pass kernel is retired)
        if (value != 1)
            return ek_fail(EK_EARG, "ek_set_option: pass kernel form must be 1");
        return EK_OK;
    case 8:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: adaptive candidates 0 or 1");
        c->adapt = value;
        return EK_OK;
    case 12:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: one-workgroup PAM windows 0 or 1");
        c->pam_sparse = value;
        return EK_OK;
    case 14:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: both cost sums for every proposal 0 or 1");
        c->sp_exact = value;
        return EK_OK;
    case 13:
        if (value < 0)
            return ek_fail(EK_EARG, "ek_set_option: pairs one workgroup searches >= 0");
        c->sp_max_pairs = value;
        return EK_OK;
    case 2:
        if (value < 0 || value > 2)
            return ek_fail(EK_EARG, "ek_set_option: assign variant 0..2");
        c->assign_variant = value;
        return EK_OK;
    default:
        return ek_fail(EK_EARG, "ek_set_option: unknown key %d", key);
    }
}

extern "C" int ek_last_run_timing(ek_ctx *c, float *ms, int32_t *launches)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (ms)
        *ms = c->last_ms;
    if (launches)
        *launches = c->last_launches;
    return EK_OK;
}

// ---- loading ---------------------------------------------------------------
extern "C" int ek_load_frames(ek_ctx *c, const float *xyz, int64_t first,
                              int64_t count, int src_is_device)
{
    if (!c || (!xyz && count > 0))
        return ek_fail(EK_EARG, "ek_load_frames: NULL argument");
    if (first < 0 || count < 0 || first + count > c->n)
        return ek_fail(EK_EARG, "ek_load_frames: [%lld,+%lld) outside [0,%lld)",
                       (long long)first, (long long)count, (long long)c->n);
    if (first % EK_TILE)
        return ek_fail(EK_EARG, "ek_load_frames: first=%lld is not a multiple "
                                "of %d", (long long)first, EK_TILE);
    EK_HIP(hipSetDevice(c->device));
    const size_t frame_floats = (size_t)3 * c->A;
    if (src_is_device) {
        ek_launch_prepare_tiles(xyz, count, c->A, c->tiles, c->G, first, c->n,
                                c->aos, c->stream);
        EK_CHECK_LAUNCH();
    } else {
        // Pageable host memory reaches the device through the driver's own bounce
        // buffer at ~16 GB/s (3.6 GB: 0.23 s next to a 0.35 s fit).  Here: chunks
        // of <= 128 MiB (whole tiles; EK_UPLOAD_CHUNK_MB: every chunk costs ~0.75 ms
        // beside its bytes at ~53 GB/s, measured) copied by a few host threads
        // (EK_UPLOAD_THREADS, default 8) into one of two
        // PINNED buffers, a DMA from there into one of two device staging buffers
        // and the layout kernel behind it on the context's stream -- while the
        // threads fill the other buffer.  The caller's array has been read
        // completely when this returns; the stream may still be working.
        size_t chunk_mb = 128;
        if (const char *e = getenv("EK_UPLOAD_CHUNK_MB"))
            chunk_mb = (size_t)std::max(1, std::min(atoi(e), 1024));
        int64_t chunk = (int64_t)((chunk_mb << 20) / (frame_floats * sizeof(float)));
        chunk = std::max<int64_t>(EK_TILE, chunk / EK_TILE * EK_TILE);
        chunk = std::min<int64_t>(chunk, (count + EK_TILE - 1) / EK_TILE * EK_TILE);
        if (chunk != c->stage_frames) {
            EK_HIP(hipStreamSynchronize(c->stream));
            for (int b = 0; b < 2; ++b) {
                (void)hipFree(c->stage[b]);
                (void)hipHostFree(c->pin[b]);
                c->stage[b] = c->pin[b] = nullptr;
                c->up_busy[b] = false;
            }
            c->stage_frames = 0;
            const size_t bytes = (size_t)chunk * frame_floats * sizeof(float);
            for (int b = 0; b < 2; ++b) {
                EK_HIP(hipMalloc((void **)&c->stage[b], bytes));
                EK_HIP(hipHostMalloc((void **)&c->pin[b], bytes, hipHostMallocDefault));
                if (!c->up_ev[b])
                    EK_HIP(hipEventCreateWithFlags(&c->up_ev[b], hipEventDisableTiming));
            }
            c->stage_frames = chunk;
        }
        int n_thr = 8;
        if (const char *e = getenv("EK_UPLOAD_THREADS"))
            n_thr = atoi(e);
        n_thr = std::max(1, std::min(n_thr, 64));
        int k = 0;
        for (int64_t done = 0; done < count; done += chunk, ++k) {
            const int b = k & 1;
            const int64_t cnt = std::min(chunk, count - done);
            const size_t bytes = (size_t)cnt * frame_floats * sizeof(float);
            if (c->up_busy[b])      // the DMA and the kernel that read this pair
                EK_HIP(hipEventSynchronize(c->up_ev[b]));
            const char *src = (const char *)(xyz + (size_t)done * frame_floats);
            char *dst = (char *)c->pin[b];
            if (bytes < ((size_t)4 << 20) || n_thr == 1) {
                memcpy(dst, src, bytes);
            } else {
                if (!c->pool)
                    c->pool = new (std::nothrow) EkCopyPool();
                if (!c->pool)
                    return ek_fail(EK_EARG, "ek_load_frames: out of host memory");
                c->pool->start(n_thr);
                c->pool->copy(dst, src, bytes);
            }
            EK_HIP(hipMemcpyAsync(c->stage[b], c->pin[b], bytes, hipMemcpyHostToDevice,
                                  c->stream));
            ek_launch_prepare_tiles(c->stage[b], cnt, c->A, c->tiles, c->G,
                                    first + done, c->n, c->aos, c->stream);
            EK_CHECK_LAUNCH();
            EK_HIP(hipEventRecord(c->up_ev[b], c->stream));
            c->up_busy[b] = true;
        }
    }
    c->loaded = true;
    c->qt_valid = false;
    return EK_OK;
}

// ---- centers given as coordinates -----------------------------------------------
static int ek_upload_centers(ek_ctx *c, const float *xyz, int32_t K)
{
    const size_t frame_floats = (size_t)3 * c->A;
    if (K > c->cen_cap) {
        EK_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->cen_aos);
        (void)hipFree(c->cen_G);
        c->cen_aos = nullptr;
        c->cen_G = nullptr;
        c->cen_cap = 0;
        // 2 buffers: raw (second half) and centred (first half)
        EK_HIP(hipMalloc((void **)&c->cen_aos,
                         (size_t)2 * K * frame_floats * sizeof(float)));
        EK_HIP(hipMalloc((void **)&c->cen_G, (size_t)K * sizeof(double)));
        c->cen_cap = K;
    }
    float *raw = c->cen_aos + (size_t)c->cen_cap * frame_floats;
    EK_HIP(hipMemcpyAsync(raw, xyz, (size_t)K * frame_floats * sizeof(float),
                          hipMemcpyHostToDevice, c->stream));
    ek_launch_prepare_centers(raw, K, c->A, c->cen_aos, c->cen_G, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

// ---- metric parity ---------------------------------------------------------------
extern "C" int ek_rmsd_to_center(ek_ctx *c, int64_t frame_index,
                                 const float *center_xyz, float *out_host)
{
    if (!c || !out_host)
        return ek_fail(EK_EARG, "ek_rmsd_to_center: NULL argument");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_rmsd_to_center: no frames loaded");
    EK_HIP(hipSetDevice(c->device));
    if (frame_index >= 0) {
        if (frame_index >= c->n)
            return ek_fail(EK_EARG, "ek_rmsd_to_center: frame %lld >= %lld",
                           (long long)frame_index, (long long)c->n);
        ek_launch_record_from_frame(c->tiles, c->G, c->A, frame_index, nullptr,
                                    c->goff, c->rec_tmp, c->stream);
    } else {
        if (!center_xyz)
            return ek_fail(EK_EARG, "ek_rmsd_to_center: no center given");
        int rc = ek_upload_centers(c, center_xyz, 1);
        if (rc)
            return rc;
        ek_launch_record_from_center(c->cen_aos, c->cen_G, c->A, c->rec_tmp,
                                     c->stream);
    }
    EK_CHECK_LAUNCH();
    ek_launch_step(ek_pick_fpl(c), 1, ek_pick_nt(c), c->tiles, c->G, c->dist, c->assign,
                   c->scratch, c->rec_tmp, 1, c->n, c->A, 0, 0.0, c->blockmax,
                   c->hist, c->ctl, c->stream);
    EK_CHECK_LAUNCH();
    EK_HIP(hipMemcpyAsync(out_host, c->scratch, (size_t)c->n * sizeof(float),
                          hipMemcpyDeviceToHost, c->stream));
    EK_HIP(hipStreamSynchronize(c->stream));
    return EK_OK;
}

// ---- state -----------------------------------------------------------------------
extern "C" int ek_history_reset(ek_ctx *c)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    EK_HIP(hipMemsetAsync(c->hist, 0, (size_t)c->hist_cap * sizeof(EkHist),
                          c->stream));
    EK_HIP(hipMemsetAsync(c->ctl, 0, sizeof(EkCtl), c->stream));
    return EK_OK;
}

// the state is about to be replaced: what a PAM prefetch derived from it (the
// frames a window's proposals can touch, +inf for the rest) no longer holds
static void ek_pam_forget(ek_ctx *c)
{
    c->pf_count = 0;
    c->tab_n = 0;
    c->sp_ready = false;
}

extern "C" int ek_state_reset(ek_ctx *c)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_state_reset: no frames loaded");
    EK_HIP(hipSetDevice(c->device));
    ek_launch_fill_state(c->dist, c->assign, c->n, __builtin_inff(), -1,
                         c->stream);
    EK_CHECK_LAUNCH();
    c->state_exact = true;      // k-centers labels name frames, at their distance
    ek_pam_forget(c);
    int rc = ek_history_reset(c);
    if (rc)
        return rc;
    return ek_local_candidate(c, nullptr);
}

extern "C" int ek_state_download(ek_ctx *c, float *dist_host,
                                 int32_t *assign_host)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    if (dist_host && c->n)
        EK_HIP(hipMemcpyAsync(dist_host, c->dist, (size_t)c->n * sizeof(float),
                              hipMemcpyDeviceToHost, c->stream));
    if (assign_host && c->n)
        EK_HIP(hipMemcpyAsync(assign_host, c->assign,
                              (size_t)c->n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, c->stream));
    EK_HIP(hipStreamSynchronize(c->stream));
    return EK_OK;
}

extern "C" int ek_state_upload(ek_ctx *c, const float *dist_host,
                               const int32_t *assign_host)
{
    if (!c || !dist_host || !assign_host)
        return ek_fail(EK_EARG, "ek_state_upload: NULL argument");
    EK_HIP(hipSetDevice(c->device));
    if (c->n) {
        EK_HIP(hipMemcpyAsync(c->dist, dist_host, (size_t)c->n * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
        EK_HIP(hipMemcpyAsync(c->assign, assign_host,
                              (size_t)c->n * sizeof(int32_t),
                              hipMemcpyHostToDevice, c->stream));
    }
    EK_HIP(hipStreamSynchronize(c->stream));
    c->state_exact = false;     // the caller's numbers: taken as they are
    ek_pam_forget(c);
    return ek_local_candidate(c, nullptr);
}

// ---- records / k-centers ---------------------------------------------------------
extern "C" int ek_local_candidate(ek_ctx *c, void *rec_dev)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_local_candidate: no frames loaded");
    EK_HIP(hipSetDevice(c->device));
    unsigned char *rec = rec_dev ? (unsigned char *)rec_dev : c->rec;
    ek_launch_pick(nullptr, 0, c->dist, c->tiles, c->G, c->n, c->A, c->goff,
                   rec, c->ctl, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

static int ek_ensure_hist(ek_ctx *c, int32_t label)
{
    if (label < c->hist_cap)
        return EK_OK;
    int32_t cap = c->hist_cap;
    while (cap <= label)
        cap *= 2;
    EkHist *h = nullptr;
    EK_HIP(hipMalloc((void **)&h, (size_t)cap * sizeof(EkHist)));
    EK_HIP(hipMemsetAsync(h, 0, (size_t)cap * sizeof(EkHist), c->stream));
    EK_HIP(hipMemcpyAsync(h, c->hist, (size_t)c->hist_cap * sizeof(EkHist),
                          hipMemcpyDeviceToDevice, c->stream));
    EK_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->hist);
    c->hist = h;
    c->hist_cap = cap;
    return EK_OK;
}

extern "C" int ek_kcenters_step(ek_ctx *c, const void *recs_dev, int32_t n_recs,
                                int32_t label, double dist_cutoff,
                                void *own_rec_out)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_kcenters_step: no frames loaded");
    if (label < 0)
        return ek_fail(EK_EARG, "ek_kcenters_step: negative label");
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_ensure_hist(c, label);
    if (rc)
        return rc;
    const unsigned char *recs =
        recs_dev ? (const unsigned char *)recs_dev : c->rec;
    if (!recs_dev)
        n_recs = 1;
    if (n_recs < 1)
        return ek_fail(EK_EARG, "ek_kcenters_step: n_recs < 1");
    unsigned char *own = own_rec_out ? (unsigned char *)own_rec_out : c->rec;
    const int fpl = ek_pick_fpl(c);
    // Triangle inequality in the sharded iteration (option key 11; reference
    // kcenters.py:351-364): every shard keeps the accepted centers in a table
    // (they are other shards' frames as often as its own) and skips the tiles
    // none of whose frames can move.  From label 0 of a fresh state only.
    bool tri = false;
    if (c->tri && recs_dev && c->state_exact && c->A >= 3) {
        if (label == 0)
            c->ti_tab_n = 0;
        if (c->ti_tab_n == label) {
            if (label + 1 > c->ti_tab_cap) {
                const int32_t cap = std::max(2 * c->ti_tab_cap, std::max(label + 1, 256));
                float *t2 = nullptr, *d2 = nullptr;
                double *g2 = nullptr;
                EK_HIP(ek_wait(c));
                EK_HIP(hipMalloc((void **)&t2, (size_t)cap * 3 * c->A * sizeof(float)));
                EK_HIP(hipMalloc((void **)&g2, (size_t)cap * sizeof(double)));
                EK_HIP(hipMalloc((void **)&d2, (size_t)cap * sizeof(float)));
                if (c->ti_tab_n > 0) {
                    EK_HIP(hipMemcpy(t2, c->ti_tab, (size_t)c->ti_tab_n * 3 * c->A *
                                                        sizeof(float),
                                     hipMemcpyDeviceToDevice));
                    EK_HIP(hipMemcpy(g2, c->ti_tabG, (size_t)c->ti_tab_n * sizeof(double),
                                     hipMemcpyDeviceToDevice));
                }
                (void)hipFree(c->ti_tab);
                (void)hipFree(c->ti_tabG);
                (void)hipFree(c->ti_D);
                c->ti_tab = t2;
                c->ti_tabG = g2;
                c->ti_D = d2;
                c->ti_tab_cap = cap;
                c->ti_cap = cap;
            }
            if (!c->ti_skip) {
                EK_HIP(hipMalloc((void **)&c->ti_skip,
                                 (size_t)std::max<int64_t>(c->n_tiles, 1)));
                EK_HIP(hipMalloc((void **)&c->ti_stats, 2 * sizeof(unsigned long long)));
                EK_HIP(hipMemsetAsync(c->ti_stats, 0, 2 * sizeof(unsigned long long),
                                      c->stream));
            }
            if (label == 0)
                EK_HIP(hipMemsetAsync(c->ti_stats, 0, 2 * sizeof(unsigned long long),
                                      c->stream));
            ek_launch_ti_tab(c->ti_tab, c->ti_tabG, c->A, label, recs, n_recs, c->ti_D,
                             c->dist, c->assign, c->n, c->ctl, c->ti_skip, c->ti_stats,
                             c->stream);
            EK_CHECK_LAUNCH();
            c->ti_tab_n = label + 1;
            tri = label >= 1 && c->n > 0;
        }
    }
    if (c->n > 0) {
        const bool sample =
            c->samp_every > 0 && (c->samp_count++ % c->samp_every) == 0 &&
            2 * (size_t)c->samp_used + 1 < c->samp_ev.size();
        if (sample)
        c->samp_form[c->samp_used] = 1;
    if (sample)
            EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used], c->stream));
        ek_launch_step(fpl, 0, ek_pick_nt(c), c->tiles, c->G, c->dist, c->assign, c->scratch,
                       recs, n_recs, c->n, c->A, label, dist_cutoff,
                       c->blockmax, c->hist, c->ctl, c->stream,
                       tri ? c->ti_skip : nullptr);
        EK_CHECK_LAUNCH();
        if (sample) {
            EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used + 1], c->stream));
            c->samp_used++;
        }
    }
    ek_launch_pick(c->n > 0 ? c->blockmax : nullptr,
                   ek_step_blocks(fpl, c->n), c->dist, c->tiles, c->G, c->n,
                   c->A, c->goff, own, c->ctl, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_history_download(ek_ctx *c, int32_t first, int32_t count,
                                   int64_t *center_index_out,
                                   float *center_dist_out, int32_t *n_done)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (first < 0 || count < 0)
        return ek_fail(EK_EARG, "ek_history_download: bad range");
    EK_HIP(hipSetDevice(c->device));
    EkCtl ctl;
    EK_HIP(hipMemcpyAsync(&ctl, c->ctl, sizeof(ctl), hipMemcpyDeviceToHost,
                          c->stream));
    std::vector<EkHist> h;
    const int32_t avail = std::max(0, std::min(count, c->hist_cap - first));
    if (avail > 0) {
        h.resize(avail);
        EK_HIP(hipMemcpyAsync(h.data(), c->hist + first,
                              (size_t)avail * sizeof(EkHist),
                              hipMemcpyDeviceToHost, c->stream));
    }
    EK_HIP(ek_wait(c));
    for (int32_t i = 0; i < count; ++i) {
        const bool ok = i < avail && h[i].set;
        if (center_index_out)
            center_index_out[i] = ok ? h[i].gidx : -1;
        if (center_dist_out)
            center_dist_out[i] = ok ? h[i].dist : 0.f;
    }
    if (n_done)
        *n_done = ctl.n_done;
    return EK_OK;
}

// ---- k-centers rounds, in the form that pays ---------------------------------------
// A round with T candidates costs more than a one-center step (more FMAs per
// byte, the small kernels that decide the chain) and accepts between 1 and T
// centers.  At the very start of a fit the guesses rarely hit -- every new
// center reshapes the distances of most frames -- and one-center steps are the
// faster way forward; soon after, nearly every guess is accepted.  Both forms
// produce the same centers, labels and distances, so the choice is free: every
// batch is timed on the device (centers per millisecond) and the other form is
// tried for a short batch at intervals that double while it keeps losing.
static int ek_run_rounds(ek_ctx *c, int Tmax, int32_t first_label,
                         int32_t max_new, double dist_cutoff, int32_t *n_added,
                         int64_t *center_index_out, float *center_dist_out,
                         float *final_maxdist)
{
    int rc = ek_spec_alloc(c);
    if (rc)
        return rc;
    if (!c->evb0) {
        EK_HIP(hipEventCreate(&c->evb0));
        EK_HIP(hipEventCreate(&c->evb1));
    }
    EkCtl ctlw;
    memset(&ctlw, 0, sizeof(ctlw));
    ctlw.n_done = first_label;
    ctlw.limit = first_label + max_new;
    EK_HIP(hipMemcpyAsync(c->ctl, &ctlw, sizeof(ctlw), hipMemcpyHostToDevice,
                          c->stream));
    EK_HIP(ek_wait(c));
    for (int m = 0; m < 4; ++m)
        c->st_rounds[m] = c->st_centers[m] = 0;
    const int nb = (int)((c->n + EK_BLOCK - 1) / EK_BLOCK);
    // Triangle inequality (option key 11; reference `use_triangle_inequality`,
    // kcenters.py:287-296): one center at a time, tiles that cannot change are
    // not read.  Needs every frame's distance to be the one to the center its
    // label names (a fresh run) and distances that behave like a metric.
    const bool tri = c->tri && c->state_exact && first_label == 0 && c->A >= 3;
    if (tri) {
        if (first_label + max_new > c->ti_cap) {
            EK_HIP(ek_wait(c));
            (void)hipFree(c->ti_D);
            c->ti_D = nullptr;
            c->ti_cap = 0;
            EK_HIP(hipMalloc((void **)&c->ti_D,
                             (size_t)(first_label + max_new + 1) * sizeof(float)));
            c->ti_cap = first_label + max_new + 1;
        }
        if (!c->ti_skip) {
            EK_HIP(hipMalloc((void **)&c->ti_skip,
                             (size_t)std::max<int64_t>(c->n_tiles, 1)));
            EK_HIP(hipMalloc((void **)&c->ti_stats, 2 * sizeof(unsigned long long)));
        }
        EK_HIP(hipMemsetAsync(c->ti_stats, 0, 2 * sizeof(unsigned long long),
                              c->stream));
    }
    c->ti_tiles = c->ti_skipped = 0;
    if (Tmax == 16 && !tri) {
        const int eq = ek_ensure_qtiles(c);
        if (eq != EK_OK)
            return eq;
    }
    // an explicit request (option key 4 = 4, 8 or 16) pins the wide form
    const bool adaptive = c->cands == -1 && c->adapt && !tri;
    const int fpl = ek_pick_fpl(c);
    const int nt = ek_pick_nt(c);
    // the forms the run moves between, by candidates per pass
    int ladder[3], n_ladder = 0;
    if (adaptive || tri)
        ladder[n_ladder++] = 1;
    if (!tri) {
        if (adaptive && Tmax == 16)
            ladder[n_ladder++] = 8;
        ladder[n_ladder++] = Tmax;
    }
    int home = 0;               // ladder index of the form being run
    int form = ladder[0];       // ... its candidates per pass (the probe's while probing)
    int held = -1;              // form the candidate record(s) were picked for
    bool probing = false;
    int probe_up = 1;           // direction of the next probe (they alternate)
    double rate_home = 0.0;     // centers per ms of the home form
    int32_t gap = 8;            // centers until another form is tried again
    int32_t since = 0;          // centers since one last was
    // three launches per round (ek_round.hip) when the pieces it is built from
    // are the ones selected
    const bool fused = c->fused && c->chain == 1;
    EkRound R;
    R.dist = c->dist;
    R.assign = c->assign;
    R.vecs = c->vecs;
    R.n = c->n;
    R.n_pad = c->n_pad;
    R.goff = c->goff;
    R.A = c->A;
    R.T = Tmax;
    R.tiles = c->tiles;
    R.qtiles = c->qtiles;
    R.aos = c->aos;
    R.G = c->G;
    R.recs = c->recsT;
    R.plan = c->plan;
    R.pend = c->pend;
    R.ord = c->ord;
    R.blockmax = c->blockmax;
    R.pm = c->pm;
    R.top = c->top;
    R.ctile = c->ctile;
    R.ctrace = c->ctrace;
    R.hist = c->hist;
    R.ctl = c->ctl;
    R.tick = c->tick;
    R.rows = c->rows;
    R.vmask = c->vmask;
    R.cutoff = dist_cutoff;
    bool pending = false;       // a fused round may have left a chain to apply
    EK_HIP(hipEventRecord(c->ev0, c->stream));
    EkCtl cr;
    memset(&cr, 0, sizeof(cr));
    cr.n_done = first_label;
    const int32_t goal = first_label + max_new;
    double per_round = 0.6 * form;  // centers per round of the current wide form
    int32_t rounds_before = 0;
    while (cr.n_done < goal) {
        const int32_t left = goal - cr.n_done;
        const bool one = form == 1;
        R.T = form;
        // ---- the record(s) this form starts from -------------------------------------
        if (held != form) {
            if (pending) {      // leaving a fused form: the state as it stands
                ek_launch_round_flush(R, c->stream);
                pending = false;
            } else if (held <= 1) {
                // the step kernel's partials are per FPL frames
                ek_launch_blockmax(c->dist, c->n, c->blockmax, c->stream);
            }
            if (one) {
                ek_launch_pick(c->blockmax, nb, c->dist, c->tiles, c->G, c->n,
                               c->A, c->goff, c->recsT, c->ctl, c->stream);
            } else if (fused) {
                ek_launch_round_chain(R, 1, c->stream);
                ek_launch_round_next(R, 1, c->stream);
            } else {
                ek_launch_pickT(c->blockmax, nb, c->tiles, c->G, c->assign, c->A,
                                form, c->goff, c->recsT, c->ctl, c->top,
                                c->stream);
            }
            EK_CHECK_LAUNCH();
            if (held != form && !one)
                per_round = 0.6 * form;
            held = form;
        }
        // ---- one batch: steps (one-center form) or rounds ----------------------------
        // long enough to amortise the host's look at the control word, short
        // enough to come back when another form is due
        const int32_t due = adaptive ? std::max(gap - since, 1) : left;
        int32_t batch;
        if (one)    // (never past the goal: a step has no limit check of its own)
            batch = std::min(left, probing ? 4 : std::min(due, 256));
        else if (probing)
            batch = 3;
        else
            batch = std::max(2, std::min(256, (int32_t)(std::min(left, due) /
                                                        per_round) + 1));
        EK_HIP(hipEventRecord(c->evb0, c->stream));
        for (int32_t r = 0; r < batch; ++r) {
            if (one) {
                const int label = cr.n_done + r;
                const bool skip = tri && label >= 1;
                if (skip)
                    ek_launch_ti(c->aos, c->G, c->A, c->hist, label, c->goff,
                                 c->recsT, c->ti_D, c->dist, c->assign, c->n,
                                 c->ctl, c->ti_skip, c->ti_stats, c->stream);
                ek_launch_step(fpl, 0, nt, c->tiles, c->G, c->dist, c->assign,
                               c->scratch, c->recsT, 1, c->n, c->A, label,
                               dist_cutoff, c->blockmax, c->hist, c->ctl, c->stream,
                               skip ? c->ti_skip : nullptr);
                ek_launch_pick(c->blockmax, ek_step_blocks(fpl, c->n), c->dist,
                               c->tiles, c->G, c->n, c->A, c->goff, c->recsT,
                               c->ctl, c->stream);
                EK_CHECK_LAUNCH();
                continue;
            }
            // sampled timing of the dominant kernel (bench.py)
            const bool sample =
                c->samp_every > 0 && (c->samp_count++ % c->samp_every) == 0 &&
                2 * (size_t)c->samp_used + 1 < c->samp_ev.size();
            if (sample)
                c->samp_form[c->samp_used] = form;
            if (fused) {
                if (sample)
                    EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used],
                                          c->stream));
                ek_launch_round_pass(R, c->stream);
                if (sample) {
                    EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used + 1],
                                          c->stream));
                    c->samp_used++;
                }
                ek_launch_round_chain(R, 0, c->stream);
                ek_launch_round_next(R, 0, c->stream);
                EK_CHECK_LAUNCH();
                pending = true;
                continue;
            }
            ek_launch_plan(c->recsT, form, c->A, form, dist_cutoff, c->planD,
                           c->plan, c->hist, c->ctl, c->stream);
            if (sample)
                EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used], c->stream));
            ek_launch_pass(form, c->tiles, c->qtiles, c->G, c->dist, c->assign,
                           c->vecs, c->n, c->n_pad, c->A, c->recsT, c->plan,
                           c->blockmax, c->ctile, c->ctrace, c->stream);
            if (sample) {
                EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used + 1],
                                      c->stream));
                c->samp_used++;
            }
            if (c->chain) {
                ek_launch_chain_max(c->dist, c->vecs, c->n, c->n_pad, c->plan,
                                    c->pm, 1, c->goff, c->stream);
                ek_launch_chain_decide_local(c->blockmax, c->pm, nb,
                                             ek_chain_max_blocks(c->n), c->goff,
                                             dist_cutoff, c->plan, c->hist,
                                             c->ctl, c->stream);
                ek_launch_chain_apply(c->vecs, c->n, c->n_pad, c->dist, c->assign,
                                      c->plan, c->blockmax, c->stream);
            } else {
                for (int j = 1; j < form; ++j) {
                    ek_launch_localmax_check(c->blockmax, nb, c->goff, dist_cutoff,
                                             c->plan, c->hist, c->ctl, c->stream);
                    ek_launch_apply(c->vecs, c->G, c->n, c->n_pad, c->A, c->dist,
                                    c->assign, c->plan, c->blockmax, c->stream);
                }
            }
            ek_launch_pickT(c->blockmax, nb, c->tiles, c->G, c->assign, c->A, form,
                            c->goff, c->recsT, c->ctl, c->top, c->stream);
            EK_CHECK_LAUNCH();
        }
        EK_HIP(hipEventRecord(c->evb1, c->stream));
        const int32_t before = cr.n_done;
        EK_HIP(hipMemcpyAsync(&cr, c->ctl, sizeof(cr), hipMemcpyDeviceToHost,
                              c->stream));
        EK_HIP(ek_wait(c));
        const int32_t got = cr.n_done - before;
        // passes that really ran (a step that finds the stop rule met returns at once)
        const int32_t ran = one ? got : cr.n_rounds - rounds_before;
        rounds_before = cr.n_rounds;
        c->st_rounds[ek_form_slot(form)] += ran;
        c->st_centers[ek_form_slot(form)] += got;
        if (cr.stopped || cr.n_done >= goal)
            break;
        if (!one)
            per_round = std::max(1.0, (double)got / std::max(ran, 1));
        if (!adaptive)
            continue;
        // ---- which form next ---------------------------------------------------------
        float bms = 0.f;
        EK_HIP(hipEventElapsedTime(&bms, c->evb0, c->evb1));
        const double rate = got / std::max((double)bms, 1e-6);
        if (probing) {
            probing = false;
            since = 0;
            if (rate > rate_home) {     // the probed form wins: it is home now
                for (int q = 0; q < n_ladder; ++q)
                    if (ladder[q] == form)
                        home = q;
                rate_home = rate;
                gap = 8;
            } else {                    // back, and wait twice as long
                form = ladder[home];
                gap = std::min(gap * 2, 1024);
            }
            continue;
        }
        rate_home = rate;
        since += got;
        if (since >= gap) {
            // Which neighbour on the ladder is worth a look?  One-center steps
            // can only win while rounds accept fewer than ~1.4 centers (the cost
            // ratio of the two forms); a narrower round only while the wider one
            // accepts fewer than the narrower could at its lower cost (8 / 16:
            // cost ratio <= 1.35); a wider round only if the chain of the
            // current one is usually accepted whole -- it breaks where the
            // farthest point is not a stored candidate, and more candidates
            // behind that point change nothing.
            const int T = ladder[home];
            const bool up_ok = home + 1 < n_ladder &&
                               (T == 1 || per_round >= 0.8 * T);
            bool down_ok = home > 0;
            if (down_ok && ladder[home - 1] == 1)
                down_ok = per_round < 1.8;
            else if (down_ok)
                down_ok = per_round < 1.35 * ladder[home - 1];
            int target = -1;
            if (up_ok && down_ok)
                target = probe_up ? home + 1 : home - 1;
            else if (up_ok)
                target = home + 1;
            else if (down_ok)
                target = home - 1;
            if (target >= 0) {
                probe_up = target > home ? 0 : 1;
                form = ladder[target];
                probing = true;
            } else {
                since = 0;
            }
        }
    }
    if (pending) {              // the last round's accepted chain
        ek_launch_round_flush(R, c->stream);
        EK_CHECK_LAUNCH();
    }
    if (tri) {
        unsigned long long st[2] = {0, 0};
        EK_HIP(hipMemcpyAsync(st, c->ti_stats, sizeof(st), hipMemcpyDeviceToHost,
                              c->stream));
        EK_HIP(ek_wait(c));
        c->ti_tiles = (int64_t)st[0];
        c->ti_skipped = (int64_t)st[1];
    }
    EK_HIP(hipEventRecord(c->ev1, c->stream));
    EK_HIP(ek_wait(c));
    EK_HIP(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    // leave the records describing the state: [0] = the shard's farthest point
    if (held == 1) {
        ek_launch_blockmax(c->dist, c->n, c->blockmax, c->stream);
        EK_CHECK_LAUNCH();
    }
    if (held >= 0)              // (nothing ran: the slot still describes the state)
        EK_HIP(hipMemcpyAsync(c->rec, c->recsT, ek_rec_bytes(c->A),
                              hipMemcpyDeviceToDevice, c->stream));
    EK_HIP(ek_wait(c));
    const int32_t added_t = std::min(max_new, std::max(0, cr.n_done - first_label));
    const int64_t passes = c->st_rounds[0] + c->st_rounds[1] + c->st_rounds[2];
    c->last_launches = (int32_t)passes;
    c->last_passes = (int32_t)passes;
    if (n_added)
        *n_added = added_t;
    if (final_maxdist)
        *final_maxdist = cr.last_max;
    if (added_t > 0 && (center_index_out || center_dist_out)) {
        rc = ek_history_download(c, first_label, added_t, center_index_out,
                                 center_dist_out, nullptr);
        if (rc)
            return rc;
    }
    return EK_OK;
}

extern "C" int ek_kcenters_run(ek_ctx *c, int32_t first_label, int32_t max_new,
                               double dist_cutoff, int32_t *n_added,
                               int64_t *center_index_out,
                               float *center_dist_out, float *final_maxdist)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_kcenters_run: no frames loaded");
    if (first_label < 0 || max_new < 0)
        return ek_fail(EK_EARG, "ek_kcenters_run: negative argument");
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_ensure_hist(c, first_label + max_new);
    if (rc)
        return rc;
    // the accepted-label counter restarts at first_label for this run
    EkCtl ctl0;
    memset(&ctl0, 0, sizeof(ctl0));
    ctl0.n_done = first_label;
    EK_HIP(hipMemcpyAsync(c->ctl, &ctl0, offsetof(EkCtl, last_max),
                          hipMemcpyHostToDevice, c->stream));
    EK_HIP(ek_wait(c));

    const int T = ek_pick_cands(c);
    if (T > 1 && c->n > 0)
        return ek_run_rounds(c, T, first_label, max_new, dist_cutoff, n_added,
                             center_index_out, center_dist_out, final_maxdist);

    // With no distance cut-off the trip count is known: enqueue everything.
    // With a cut-off, enqueue in batches and look at the stop flag in between
    // (steps enqueued past the stopping point are no-ops on the device).
    const bool open_loop = !(dist_cutoff > 0.0);
    const int32_t batch = open_loop ? max_new : 16;
    int32_t issued = 0;
    EkCtl ctl;
    memset(&ctl, 0, sizeof(ctl));
    ctl.n_done = first_label;
    EK_HIP(hipEventRecord(c->ev0, c->stream));
    while (issued < max_new) {
        const int32_t todo = std::min(batch, max_new - issued);
        for (int32_t i = 0; i < todo; ++i) {
            rc = ek_kcenters_step(c, nullptr, 1, first_label + issued + i,
                                  dist_cutoff, nullptr);
            if (rc)
                return rc;
        }
        issued += todo;
        if (!open_loop) {
            EK_HIP(hipMemcpyAsync(&ctl, c->ctl, sizeof(ctl),
                                  hipMemcpyDeviceToHost, c->stream));
            EK_HIP(ek_wait(c));
            if (ctl.stopped)
                break;
        }
    }
    EK_HIP(hipEventRecord(c->ev1, c->stream));
    EK_HIP(hipMemcpyAsync(&ctl, c->ctl, sizeof(ctl), hipMemcpyDeviceToHost,
                          c->stream));
    EK_HIP(ek_wait(c));
    EK_HIP(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    const int32_t added = std::max(0, ctl.n_done - first_label);
    c->last_launches = added;
    if (n_added)
        *n_added = added;
    if (final_maxdist)
        *final_maxdist = ctl.last_max;
    if (added > 0 && (center_index_out || center_dist_out)) {
        rc = ek_history_download(c, first_label, added, center_index_out,
                                 center_dist_out, nullptr);
        if (rc)
            return rc;
    }
    return EK_OK;
}

// ---- sampled timing of the distance kernel ---------------------------------------
extern "C" int ek_timing_begin(ek_ctx *c, int32_t sample_every,
                               int32_t max_samples)
{
    if (!c || sample_every < 1 || max_samples < 1)
        return ek_fail(EK_EARG, "ek_timing_begin: bad argument");
    EK_HIP(hipSetDevice(c->device));
    while (c->samp_ev.size() < 2 * (size_t)max_samples) {
        hipEvent_t e;
        EK_HIP(hipEventCreate(&e));
        c->samp_ev.push_back(e);
    }
    c->samp_form.assign((size_t)max_samples, 0);
    c->samp_every = sample_every;
    c->samp_used = 0;
    c->samp_count = 0;
    return EK_OK;
}

extern "C" int ek_timing_end(ek_ctx *c, float *avg_ms, int32_t *n_samples)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    EK_HIP(ek_wait(c));
    // the average is over the launches of the form that was sampled most (a run
    // moves between 1, 8 and 16 candidates per pass; ek_timing_form says which)
    int cnt[EK_MAX_CANDS + 1] = {0};
    for (int i = 0; i < c->samp_used; ++i)
        cnt[std::min(std::max(c->samp_form[i], 0), EK_MAX_CANDS)]++;
    int dom = 0;
    for (int f = 1; f <= EK_MAX_CANDS; ++f)
        if (cnt[f] > cnt[dom])
            dom = f;
    c->samp_dom = dom;
    // launches enqueued past the stopping point return immediately (device
    // no-ops): only samples within 4x of the longest count as real launches
    std::vector<float> t((size_t)c->samp_used);
    float mx = 0.f;
    for (int i = 0; i < c->samp_used; ++i) {
        EK_HIP(hipEventElapsedTime(&t[i], c->samp_ev[2 * i],
                                   c->samp_ev[2 * i + 1]));
        if (c->samp_form[i] == dom)
            mx = std::max(mx, t[i]);
    }
    double sum = 0.0;
    int used = 0;
    for (int i = 0; i < c->samp_used; ++i)
        if (c->samp_form[i] == dom && t[i] > 0.25f * mx) {
            sum += t[i];
            ++used;
        }
    if (avg_ms)
        *avg_ms = used ? (float)(sum / used) : 0.f;
    if (n_samples)
        *n_samples = used;
    c->samp_every = 0;
    return EK_OK;
}

extern "C" int ek_timing_form(ek_ctx *c, int32_t *candidates)
{
    if (!c || !candidates)
        return ek_fail(EK_EARG, "ek_timing_form: NULL argument");
    *candidates = c->samp_dom;
    return EK_OK;
}

// ---- nearest-center assignment -------------------------------------------------------
extern "C" int ek_assign_nearest(ek_ctx *c, const float *centers_xyz,
                                 int32_t n_centers)
{
    if (!c || (!centers_xyz && n_centers > 0))
        return ek_fail(EK_EARG, "ek_assign_nearest: NULL argument");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_assign_nearest: no frames loaded");
    if (n_centers < 0)
        return ek_fail(EK_EARG, "ek_assign_nearest: negative n_centers");
    if ((size_t)3 * c->A * 8 * sizeof(float) > 150 * 1024)
        return ek_fail(EK_EARG, "ek_assign_nearest: %d atoms exceed the LDS "
                                "center tile (limit 1600)", c->A);
    EK_HIP(hipSetDevice(c->device));
    c->state_exact = false;     // labels name the caller's centers, not frames
    ek_pam_forget(c);
    if (n_centers > 0) {
        int rc = ek_upload_centers(c, centers_xyz, n_centers);
        if (rc)
            return rc;
    }
    // frames x centers is a dense contraction: matrix cores unless the problem
    // is too small to fill 32 x 32 tiles (results are bit-identical)
    const bool mfma = c->assign_variant == 2 ||
                      (c->assign_variant == 0 && n_centers >= 24 && c->n >= 64);
    if (mfma && n_centers > 0) {
        const int32_t need = (n_centers + EK_TILE - 1) / EK_TILE * EK_TILE;
        if (need > c->cen_tiles_cap) {
            EK_HIP(ek_wait(c));
            (void)hipFree(c->cen_tiles);
            c->cen_tiles = nullptr;
            c->cen_tiles_cap = 0;
            EK_HIP(hipMalloc((void **)&c->cen_tiles,
                             (size_t)need * 3 * c->A * sizeof(float)));
            c->cen_tiles_cap = need;
        }
        EK_HIP(hipMemsetAsync(c->cen_tiles, 0,
                              (size_t)need * 3 * c->A * sizeof(float),
                              c->stream));
        const float *raw = c->cen_aos + (size_t)c->cen_cap * 3 * c->A;
        ek_launch_prepare_tiles(raw, n_centers, c->A, c->cen_tiles, c->cen_G, 0,
                                n_centers, nullptr, c->stream);
        ek_launch_assign_mfma(c->tiles, c->G, c->n, c->A, c->cen_tiles, c->cen_G,
                              n_centers, c->dist, c->assign, c->stream);
    } else {
        ek_launch_assign(c->tiles, c->G, c->n, c->A, c->cen_aos, c->cen_G,
                         n_centers, c->dist, c->assign, c->stream);
    }
    EK_CHECK_LAUNCH();
    return ek_local_candidate(c, nullptr);
}

// ---- PAM (k-medoids) proposals ----------------------------------------------------------
// working set of a sweep over K medoids
static int ek_pam_alloc(ek_ctx *c, int32_t K)
{
    const size_t nn = (size_t)std::max<int64_t>(c->n, 1);
    const size_t nb = (nn + EK_BLOCK - 1) / EK_BLOCK;
    if (!c->ndist) {
        EK_HIP(hipMalloc((void **)&c->ndist, nn * sizeof(float)));
        EK_HIP(hipMalloc((void **)&c->nassign, nn * sizeof(int32_t)));
        EK_HIP(hipMalloc((void **)&c->amb, nn * sizeof(uint32_t)));
        EK_HIP(hipMalloc((void **)&c->amb_best, nn * sizeof(unsigned long long)));
        // [0] ambiguous members, [1] their reach (float bits), [2] listed medoids
        EK_HIP(hipMalloc((void **)&c->amb_count, 4 * sizeof(unsigned int)));
        EK_HIP(hipMalloc((void **)&c->blockcnt, nb * sizeof(int32_t)));
        EK_HIP(hipMalloc((void **)&c->scan, nb * sizeof(int64_t)));
        EK_HIP(hipMalloc((void **)&c->sel, 2 * sizeof(int64_t)));
        {
            EkPwShape hs[2];
            const int64_t n_full = c->n / EK_PW_CHUNK;
            const int last_len = (int)(c->n - n_full * EK_PW_CHUNK);
            ek_pw_build_shape(n_full > 0 ? EK_PW_CHUNK : 0, &hs[0]);
            ek_pw_build_shape(last_len, &hs[1]);
            // the one-launch cost sums (ek_pw_window_kernel) add a full chunk's
            // 64 leaves of 128 as a perfect in-order binary tree: true for
            // numpy's pairwise split of 8192 elements, checked here
            {
                bool ok = n_full == 0 ||
                          (hs[0].n_leaves == EK_PW_FULL_LEAVES && hs[0].n_levels == 6);
                for (int i = 0; ok && n_full > 0 && i < EK_PW_FULL_LEAVES; ++i)
                    ok = hs[0].leaf_off[i] == 128 * i && hs[0].leaf_len[i] == 128;
                for (int k = 0; ok && n_full > 0 && k < hs[0].n_nodes; ++k) {
                    // level-ordered nodes: level 1 joins leaves (2j, 2j+1), ..
                    const int lev_start[7] = {0, 32, 48, 56, 60, 62, 63};
                    int lev = 0;
                    while (k >= lev_start[lev + 1])
                        ++lev;
                    const int j = k - lev_start[lev];
                    const int base = lev == 0 ? 0 : hs[0].n_leaves + lev_start[lev - 1];
                    ok = hs[0].node_l[k] == base + 2 * j &&
                         hs[0].node_r[k] == base + 2 * j + 1;
                }
                c->pw_tail_ok = ok;
            }
            c->pw_n_full = (int)n_full;
            c->pw_leaves = (int)n_full * EK_PW_FULL_LEAVES + hs[1].n_leaves;
            c->pw_chunks = (int)n_full + (last_len > 0 ? 1 : 0);
            EK_HIP(hipMalloc((void **)&c->pw_shapes, sizeof(hs)));
            EK_HIP(hipMemcpy(c->pw_shapes, hs, sizeof(hs), hipMemcpyHostToDevice));
            EK_HIP(hipMalloc((void **)&c->sq_part,
                             (2 * (size_t)std::max(c->pw_leaves, 1) +
                              2 * (size_t)std::max(c->pw_chunks, 1)) *
                                 sizeof(double)));
        }
        EK_HIP(hipMalloc((void **)&c->sq_out, 2 * sizeof(double)));
        EK_HIP(hipMalloc((void **)&c->bat_blockcnt,
                         (size_t)EK_PAM_WIN * nb * sizeof(int32_t)));
        EK_HIP(hipMalloc((void **)&c->bat_scan,
                         (size_t)EK_PAM_WIN * nb * sizeof(int64_t)));
        // [0,8) counts, [8,16) selected frames, [16,24) requested member ranks
        EK_HIP(hipMalloc((void **)&c->bat_sel,
                         3 * EK_PAM_WIN * sizeof(int64_t)));
        EK_HIP(hipMalloc((void **)&c->moved, sizeof(unsigned int)));
        EK_HIP(hipMemsetAsync(c->moved, 0, sizeof(unsigned int), c->stream));
        EK_HIP(hipMalloc((void **)&c->pam_out_dev, sizeof(EkPamOut)));
        EK_HIP(hipHostMalloc((void **)&c->pam_out_host, sizeof(EkPamOut),
                             hipHostMallocDefault));
        EK_HIP(hipMalloc((void **)&c->pam_win_dev, sizeof(EkPamWin)));
        EK_HIP(hipHostMalloc((void **)&c->pam_win_host, sizeof(EkPamWin),
                             hipHostMallocDefault));
    }
    c->pam_restore = -1;
    c->pf_backoff = 0;
    c->sp_ready = false;
    c->sp_backoff = 0;
    c->sp_backoff_next = 8;
    c->bat_cid0 = -1;
    c->bat_count = 0;
    c->pf_count = 0;
    c->pf_external = false;
    if (K > c->med_cap) {
        EK_HIP(ek_wait(c));
        (void)hipFree(c->med_aos);
        (void)hipFree(c->med_G);
        (void)hipFree(c->med_idx);
        (void)hipFree(c->med_list);
        (void)hipFree(c->dtab);
        c->med_list = nullptr;
        c->dtab = nullptr;
        c->med_aos = nullptr;
        c->med_G = nullptr;
        c->med_idx = nullptr;
        c->med_cap = 0;
        EK_HIP(hipMalloc((void **)&c->med_aos,
                         (size_t)(K + 1) * 3 * c->A * sizeof(float)));
        EK_HIP(hipMalloc((void **)&c->med_G, (size_t)(K + 1) * sizeof(double)));
        EK_HIP(hipMalloc((void **)&c->med_idx, (size_t)(K + 1) * sizeof(int64_t)));
        EK_HIP(hipMalloc((void **)&c->med_list, (size_t)(K + 1) * sizeof(int32_t)));
        EK_HIP(hipMalloc((void **)&c->dtab,
                         (size_t)3 * EK_PAM_WIN * (K + 1) * sizeof(float)));
        c->med_cap = K;
    }
    c->med_K = K;
    c->pam_cid = -1;
    c->cnt_cid = -1;
    c->tab_n = 0;
    return EK_OK;
}

static int ek_tmp_idx(ek_ctx *c, int64_t count)
{
    if (count <= c->tmp_idx_cap)
        return EK_OK;
    EK_HIP(ek_wait(c));
    (void)hipFree(c->tmp_idx);
    c->tmp_idx = nullptr;
    c->tmp_idx_cap = 0;
    const int64_t cap = std::max<int64_t>(1024, count);
    EK_HIP(hipMalloc((void **)&c->tmp_idx, (size_t)cap * sizeof(int64_t)));
    c->tmp_idx_cap = cap;
    return EK_OK;
}

extern "C" int ek_pam_begin(ek_ctx *c, const int64_t *medoid_frames, int32_t K)
{
    if (!c || !medoid_frames || K < 1)
        return ek_fail(EK_EARG, "ek_pam_begin: bad argument");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_pam_begin: no frames loaded");
    for (int32_t i = 0; i < K; ++i)
        if (medoid_frames[i] < 0 || medoid_frames[i] >= c->n)
            return ek_fail(EK_EARG, "ek_pam_begin: medoid %d = frame %lld out "
                                    "of range", i, (long long)medoid_frames[i]);
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_pam_alloc(c, K);
    if (rc)
        return rc;
    EK_HIP(hipMemcpyAsync(c->med_idx, medoid_frames, (size_t)K * sizeof(int64_t),
                          hipMemcpyHostToDevice, c->stream));
    EK_HIP(ek_wait(c));
    ek_launch_gather_frames(c->tiles, c->G, c->A, c->med_idx, K, 0, c->med_aos,
                            c->med_G, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

// ---- PAM with medoids / proposals that live on other shards ---------------------------
extern "C" int ek_centered_frames(ek_ctx *c, const int64_t *local_frames,
                                  const int32_t *rows, int32_t count,
                                  float *aos_dev, double *G_dev)
{
    if (!c || count < 0 || (count > 0 && (!local_frames || !rows)) || !aos_dev ||
        !G_dev)
        return ek_fail(EK_EARG, "ek_centered_frames: bad argument");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_centered_frames: no frames loaded");
    if (count == 0)
        return EK_OK;
    for (int32_t i = 0; i < count; ++i)
        if (local_frames[i] < 0 || local_frames[i] >= c->n || rows[i] < 0)
            return ek_fail(EK_EARG, "ek_centered_frames: item %d (frame %lld, "
                                    "row %d) out of range", i,
                           (long long)local_frames[i], rows[i]);
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_tmp_idx(c, 2 * (int64_t)count);
    if (rc)
        return rc;
    std::vector<int64_t> h(2 * (size_t)count);
    for (int32_t i = 0; i < count; ++i) {
        h[i] = local_frames[i];
        h[(size_t)count + i] = rows[i];
    }
    EK_HIP(hipMemcpyAsync(c->tmp_idx, h.data(), h.size() * sizeof(int64_t),
                          hipMemcpyHostToDevice, c->stream));
    EK_HIP(ek_wait(c));
    ek_launch_gather_rows(c->tiles, c->G, c->A, c->tmp_idx, c->tmp_idx + count,
                          count, aos_dev, G_dev, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_pam_begin_table(ek_ctx *c, const float *aos_dev,
                                  const double *G_dev, int32_t K)
{
    if (!c || !aos_dev || !G_dev || K < 1)
        return ek_fail(EK_EARG, "ek_pam_begin_table: bad argument");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_pam_begin_table: no frames loaded");
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_pam_alloc(c, K);
    if (rc)
        return rc;
    EK_HIP(hipMemcpyAsync(c->med_aos, aos_dev, (size_t)K * 3 * c->A * sizeof(float),
                          hipMemcpyDeviceToDevice, c->stream));
    EK_HIP(hipMemcpyAsync(c->med_G, G_dev, (size_t)K * sizeof(double),
                          hipMemcpyDeviceToDevice, c->stream));
    return EK_OK;
}

extern "C" int ek_pam_count_members(ek_ctx *c, int32_t cid, int64_t *count)
{
    if (!c || !count)
        return ek_fail(EK_EARG, "ek_pam_count_members: NULL argument");
    if (!c->ndist)
        return ek_fail(EK_ESTATE, "ek_pam_count_members: call ek_pam_begin first");
    EK_HIP(hipSetDevice(c->device));
    ek_launch_count_members(c->assign, c->n, cid, c->blockcnt, c->scan, c->sel,
                            c->stream);
    EK_CHECK_LAUNCH();
    EK_HIP(hipMemcpyAsync(count, c->sel, sizeof(int64_t), hipMemcpyDeviceToHost,
                          c->stream));
    EK_HIP(ek_wait(c));
    c->cnt_cid = cid;
    c->cnt_m = *count;
    return EK_OK;
}

extern "C" int ek_pam_select_member(ek_ctx *c, int32_t cid, int64_t j,
                                    int64_t *frame_index)
{
    if (!c || !frame_index)
        return ek_fail(EK_EARG, "ek_pam_select_member: NULL argument");
    if (!c->ndist)
        return ek_fail(EK_ESTATE, "ek_pam_select_member: call ek_pam_begin first");
    EK_HIP(hipSetDevice(c->device));
    // relies on the scan left by the preceding ek_pam_count_members(cid)
    ek_launch_select_member(c->assign, c->n, cid, c->scan, j, c->sel + 1,
                            c->stream);
    EK_CHECK_LAUNCH();
    EK_HIP(hipMemcpyAsync(frame_index, c->sel + 1, sizeof(int64_t),
                          hipMemcpyDeviceToHost, c->stream));
    EK_HIP(ek_wait(c));
    if (*frame_index < 0)
        return ek_fail(EK_EARG, "ek_pam_select_member: cluster %d has no "
                                "member %lld", cid, (long long)j);
    return EK_OK;
}

// Everything of a proposal after the distance vector `newd` is known and the
// trial medoid table holds the proposal in row cid (ek_pam_trial_kernel, which
// also clears the counters): classification, the ambiguous subset against all
// medoids, both cost sums and the moved-cluster mask, packed into *out (device).
// No read-back.  max_amb bounds the ambiguous set (a subset of cluster cid's
// members) and sizes the follow-up launches.
// decide != nullptr (a slot of the window run): three launches -- the
// classification first takes over the trial state of the slot before if
// *prev_accept says it was accepted, ambiguous members stay marked in the trial
// state until the cost-sum launch resolves them, and that launch's last
// workgroup decides the proposal (EkPamDecide) -- and nothing else to do
static int ek_pam_tail(ek_ctx *c, int32_t cid, const float *newd,
                       int64_t max_amb, int32_t win_lo, int32_t win_count,
                       EkPamOut *out, const EkPamDecide *decide = nullptr,
                       const int32_t *prev_accept = nullptr)
{
    const int K = c->med_K;
    const int fuse = decide != nullptr;
    c->sp_ready = false;        // c->amb is the ambiguous members' list from here on
    // only when dist[f] is known to be the distance to medoid assign[f] (a state
    // this library produced; not one uploaded by the caller) may the search skip
    // medoids out of the members' reach
    // (and not for 1- or 2-atom "structures": collinear points make the largest
    // root of the QCP quartic a double root, the computed distances are then too
    // erratic to be treated as a metric)
    const bool prune = c->prune && c->state_exact && c->A >= 3;
    // inside a window whose distance tables are in place the classification's
    // last workgroup lists the medoids within reach itself
    const bool tabs = fuse && prune && c->tab_lo == cid - decide->slot &&
                      decide->slot < c->tab_n;
    if (fuse) {
        const size_t tb = (size_t)EK_PAM_WIN * (c->med_cap + 1);
        EkPamClsWin w;
        w.prev_accept = prev_accept;
        w.frames_aos = c->aos;
        w.G = c->G;
        w.A = c->A;
        w.ambt = c->ambt;
        w.ambG = c->ambG;
        w.cap = c->ambt_cap;
        w.O = tabs ? c->dtab + tb + (size_t)decide->slot * K : nullptr;
        w.T = c->dtab;
        w.accepted = decide->win->accept;
        w.K = K;
        w.cid0 = cid - decide->slot;
        w.slot = decide->slot;
        w.list = c->med_list;
        w.tick = c->tick + 192;
        ek_launch_pam_classify_window(c->dist, c->assign, newd, c->n, cid, c->ndist,
                                      c->nassign, c->amb, c->amb_best,
                                      c->amb_count, w, c->stream);
    } else {
        ek_launch_pam_classify(c->dist, c->assign, newd, c->n, cid, c->ndist,
                               c->nassign, c->amb, c->amb_best, c->amb_count,
                               c->amb_count + 1, c->stream, 0);
    }
    if (prune && !tabs)
        ek_launch_pam_prune(c->med_aos, c->med_G, c->A, K, cid, c->amb_count + 1,
                            c->med_list, c->amb_count + 2, c->stream);
    ek_launch_subset_assign(c->tiles, c->G, c->A, c->amb, c->amb_count, max_amb,
                            c->ambt, c->ambG, c->ambt_cap, c->med_aos, c->med_G,
                            K, prune ? c->med_list : nullptr, c->amb_count + 2,
                            newd, cid, c->amb_best, c->stream, fuse != 0);
    if (!fuse)
        ek_launch_pam_scatter(c->amb, c->amb_best, c->amb_count, max_amb, c->ndist,
                              c->nassign, c->stream);
    ek_launch_sumsq_pack(c->dist, c->ndist, c->assign, c->nassign, c->n, win_lo,
                         win_count, c->pw_shapes, c->pw_n_full, c->pw_leaves,
                         c->pw_chunks, c->sq_part, c->amb_count, c->moved, out,
                         c->stream, fuse ? c->amb_best : nullptr,
                         fuse ? c->tick + 128 : nullptr, decide);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

// room for the compacted ambiguous members (before anything of the proposal is
// enqueued: growing it synchronises)
static int ek_pam_amb_room(ek_ctx *c, int64_t max_amb)
{
    if (max_amb <= c->ambt_cap)
        return EK_OK;
    EK_HIP(ek_wait(c));
    (void)hipFree(c->ambt);
    (void)hipFree(c->ambG);
    c->ambt = nullptr;
    c->ambG = nullptr;
    c->ambt_cap = 0;
    const int64_t cap = std::max<int64_t>(
        4096, (max_amb * 5 / 4 + EK_BLOCK - 1) / EK_BLOCK * EK_BLOCK);
    EK_HIP(hipMalloc((void **)&c->ambt, (size_t)cap * 3 * c->A * sizeof(float)));
    EK_HIP(hipMalloc((void **)&c->ambG, (size_t)cap * sizeof(double)));
    c->ambt_cap = cap;
    return EK_OK;
}

// the prefetched distance vector of a local frame, or nullptr
static const float *ek_pam_prefetched(ek_ctx *c, int64_t frame_index)
{
    if (frame_index < 0 || c->pf_external)
        return nullptr;
    for (int32_t j = 0; j < c->pf_count; ++j)
        if (c->pf_frames[j] == frame_index)
            return c->pam_vecs + (size_t)j * c->n_pad;
    return nullptr;
}

// shared body of the local-frame proposal entry points.  The proposed frame's
// index is either `frame_index` (>= 0) or already on the device in c->sel[1].
static int ek_pam_propose_impl(ek_ctx *c, int32_t cid, int64_t frame_index,
                               int64_t max_amb, int64_t *frame_out,
                               double *old_cost, double *new_cost,
                               int64_t *n_ambiguous, int32_t win_lo = 0,
                               int32_t win_count = 0,
                               uint32_t *moved_mask = nullptr)
{
    const int K = c->med_K;
    int rc = ek_pam_amb_room(c, max_amb);
    if (rc)
        return rc;
    const float *newd = ek_pam_prefetched(c, frame_index);
    if (newd)
        ++c->pf_hits;
    else
        ++c->pf_misses;
    const int64_t *idx_dev = c->sel + 1;    // read only when frame_index < 0
    if (!newd) {
        // distances of every frame to the proposed medoid (kmedoids.py:637)
        ek_launch_record_from_frame(c->tiles, c->G, c->A,
                                    frame_index >= 0 ? frame_index : 0,
                                    frame_index >= 0 ? nullptr : idx_dev, c->goff,
                                    c->rec_tmp, c->stream);
        ek_launch_step(ek_pick_fpl(c), 1, ek_pick_nt(c), c->tiles, c->G, c->dist,
                       c->assign, c->scratch, c->rec_tmp, 1, c->n, c->A, 0, 0.0,
                       c->blockmax, c->hist, c->ctl, c->stream);
        EK_CHECK_LAUNCH();
        newd = c->scratch;
    }
    // trial medoid table (undoing a rejected proposal's row first), counters
    c->tab_n = 0;
    ek_launch_pam_trial(c->tiles, c->G, c->A, c->med_aos, c->med_G, K, cid,
                        c->pam_restore, frame_index, idx_dev, nullptr, nullptr,
                        c->amb_count, c->moved, c->stream);
    c->pam_restore = -1;
    rc = ek_pam_tail(c, cid, newd, max_amb, win_lo, moved_mask ? win_count : 0,
                     c->pam_out_dev);
    if (rc)
        return rc;
    int64_t fidx = frame_index;
    EK_HIP(hipMemcpyAsync(c->pam_out_host, c->pam_out_dev, sizeof(EkPamOut),
                          hipMemcpyDeviceToHost, c->stream));
    if (frame_index < 0)
        EK_HIP(hipMemcpyAsync(&fidx, idx_dev, sizeof(int64_t),
                              hipMemcpyDeviceToHost, c->stream));
    EK_HIP(ek_wait(c));
    const EkPamOut r = *c->pam_out_host;
    c->pam_cid = cid;           // pending even if the check below fails
    c->pam_frame = fidx;
    if ((int64_t)r.n_amb > max_amb)
        return ek_fail(EK_EARG, "PAM proposal: cluster %d has %u members that "
                                "stay put, more than the %lld members declared",
                       cid, r.n_amb, (long long)max_amb);
    if (old_cost)
        *old_cost = r.sum_old / (double)c->n;
    if (new_cost)
        *new_cost = r.sum_new / (double)c->n;
    if (n_ambiguous)
        *n_ambiguous = r.n_amb;
    if (moved_mask)
        *moved_mask = r.moved;
    if (frame_out)
        *frame_out = fidx;
    return EK_OK;
}

static int ek_pam_precheck(ek_ctx *c, int32_t cid, const char *who)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->ndist || c->med_K < 1)
        return ek_fail(EK_ESTATE, "%s: call ek_pam_begin first", who);
    if (c->pam_cid >= 0)
        return ek_fail(EK_ESTATE, "%s: previous proposal not committed", who);
    if (cid < 0 || cid >= c->med_K)
        return ek_fail(EK_EARG, "%s: cid=%d out of range", who, cid);
    if ((size_t)3 * c->A * 8 * sizeof(float) > 150 * 1024)
        return ek_fail(EK_EARG, "%s: %d atoms exceed the LDS center tile "
                                "(limit 1600)", who, c->A);
    return EK_OK;
}

extern "C" int ek_pam_propose(ek_ctx *c, int32_t cid, int64_t frame_index,
                              double *old_cost, double *new_cost,
                              int64_t *n_ambiguous)
{
    int rc = ek_pam_precheck(c, cid, "ek_pam_propose");
    if (rc)
        return rc;
    if (frame_index < 0 || frame_index >= c->n)
        return ek_fail(EK_EARG, "ek_pam_propose: frame %lld out of range",
                       (long long)frame_index);
    EK_HIP(hipSetDevice(c->device));
    int64_t m = 0;
    if (c->cnt_cid == cid) {
        m = c->cnt_m;
    } else {
        rc = ek_pam_count_members(c, cid, &m);
        if (rc)
            return rc;
    }
    c->cnt_cid = -1;
    return ek_pam_propose_impl(c, cid, frame_index, m, nullptr, old_cost,
                               new_cost, n_ambiguous);
}

extern "C" int ek_pam_propose_member(ek_ctx *c, int32_t cid, int64_t j,
                                     int64_t *frame_index, double *old_cost,
                                     double *new_cost, int64_t *n_ambiguous)
{
    int rc = ek_pam_precheck(c, cid, "ek_pam_propose_member");
    if (rc)
        return rc;
    if (c->cnt_cid != cid)
        return ek_fail(EK_ESTATE, "ek_pam_propose_member: call "
                                  "ek_pam_count_members(%d) first", cid);
    if (j < 0 || j >= c->cnt_m)
        return ek_fail(EK_EARG, "ek_pam_propose_member: member %lld of %lld",
                       (long long)j, (long long)c->cnt_m);
    EK_HIP(hipSetDevice(c->device));
    ek_launch_select_member(c->assign, c->n, cid, c->scan, j, c->sel + 1,
                            c->stream);
    EK_CHECK_LAUNCH();
    const int64_t m = c->cnt_m;
    c->cnt_cid = -1;
    return ek_pam_propose_impl(c, cid, -1, m, frame_index, old_cost, new_cost,
                               n_ambiguous);
}

extern "C" int ek_pam_commit(ek_ctx *c, int accept)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (c->pam_cid < 0)
        return ek_fail(EK_ESTATE, "ek_pam_commit: no proposal pending");
    EK_HIP(hipSetDevice(c->device));
    if (accept) {
        std::swap(c->dist, c->ndist);
        std::swap(c->assign, c->nassign);
    } else {
        // the trial row is put back by the next proposal's first kernel
        c->pam_restore = c->pam_cid;
    }
    c->pam_cid = -1;
    c->cnt_cid = -1;
    return EK_OK;
}

extern "C" int32_t ek_pam_window_max(void)
{
    return EK_PAM_WIN;
}

// ---- PAM proposal prefetch ------------------------------------------------------------
// A sweep visits clusters 0..K-1 in order and an accepted proposal rarely
// touches the clusters visited next, so the host draws the next few proposals
// ahead of time, gets their distance vectors from ONE pass over the frames
// (ek_pass_kernel<T,false>), and checks each guess when its turn comes.
extern "C" int ek_pam_count_members_batch(ek_ctx *c, int32_t cid0, int32_t count,
                                          int64_t *counts)
{
    if (!c || !counts)
        return ek_fail(EK_EARG, "ek_pam_count_members_batch: NULL argument");
    if (!c->ndist || c->med_K < 1)
        return ek_fail(EK_ESTATE, "ek_pam_count_members_batch: call ek_pam_begin "
                                  "first");
    if (count < 1 || count > EK_PAM_WIN || cid0 < 0 || cid0 + count > c->med_K)
        return ek_fail(EK_EARG, "ek_pam_count_members_batch: clusters [%d,+%d) "
                                "outside [0,%d) or more than %d", cid0, count,
                       c->med_K, EK_PAM_WIN);
    EK_HIP(hipSetDevice(c->device));
    const size_t nb = ((size_t)std::max<int64_t>(c->n, 1) + EK_BLOCK - 1) / EK_BLOCK;
    (void)nb;
    ek_launch_count_members_multi(c->assign, c->n, cid0, count, c->bat_blockcnt,
                                  c->bat_scan, c->bat_sel, c->stream);
    EK_CHECK_LAUNCH();
    EK_HIP(hipMemcpyAsync(counts, c->bat_sel, (size_t)count * sizeof(int64_t),
                          hipMemcpyDeviceToHost, c->stream));
    EK_HIP(ek_wait(c));
    c->bat_cid0 = cid0;
    c->bat_count = count;
    return EK_OK;
}

extern "C" int ek_pam_select_members_batch(ek_ctx *c, int32_t cid0, int32_t count,
                                           const int64_t *js, int64_t *frames)
{
    if (!c || !js || !frames)
        return ek_fail(EK_EARG, "ek_pam_select_members_batch: NULL argument");
    if (c->bat_cid0 != cid0 || count < 1 || count > c->bat_count)
        return ek_fail(EK_ESTATE, "ek_pam_select_members_batch: call "
                                  "ek_pam_count_members_batch(%d, >=%d) first",
                       cid0, count);
    EK_HIP(hipSetDevice(c->device));
    const size_t nb = ((size_t)std::max<int64_t>(c->n, 1) + EK_BLOCK - 1) / EK_BLOCK;
    (void)nb;
    // (a negative rank: that member lives on another shard)
    EK_HIP(hipMemcpyAsync(c->bat_sel + 2 * EK_PAM_WIN, js,
                          (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice,
                          c->stream));
    ek_launch_select_member_multi(c->assign, c->n, cid0, count, c->bat_scan,
                                  c->bat_sel + 2 * EK_PAM_WIN,
                                  c->bat_sel + EK_PAM_WIN, c->stream);
    EK_CHECK_LAUNCH();
    EK_HIP(hipMemcpyAsync(frames, c->bat_sel + EK_PAM_WIN,
                          (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost,
                          c->stream));
    EK_HIP(ek_wait(c));
    // the scans describe the state at count time only
    c->bat_cid0 = -1;
    c->bat_count = 0;
    for (int32_t j = 0; j < count; ++j)
        if (frames[j] < 0 && js[j] >= 0)
            return ek_fail(EK_EARG, "ek_pam_select_members_batch: cluster %d has "
                                    "no member %lld", cid0 + j, (long long)js[j]);
    return EK_OK;
}

static int ek_pam_vecs_alloc(ek_ctx *c)
{
    if (!c->pam_vecs) {
        EK_HIP(hipMalloc((void **)&c->pam_vecs,
                         (size_t)EK_PAM_WIN * std::max<int64_t>(c->n_pad, 1) *
                             sizeof(float)));
        EK_HIP(hipMalloc((void **)&c->pam_recs,
                         (size_t)EK_PAM_WIN * ek_rec_bytes(c->A)));
        EK_HIP(hipMalloc((void **)&c->pam_plan, sizeof(EkPlan)));
    }
    return EK_OK;
}

// The distance vectors of the `count` records in c->pam_recs.  When the state is
// exact (every frame's distance is the distance to the medoid its label names)
// and the window of clusters being worked through is known, only the frames a
// proposal can touch get exact distances (ek_pam.hip, "proposal prefetch
// restricted ..."): the others get +inf.
// local: the proposals are frames of this shard, proposal j for cluster
// win_lo + j when win_count == count (the layout ek_pam_window_run expects)
// prepared: ek_launch_pam_setup made the records (plan, candidate tile and the
// cleared active-frame counter come with them)
static int ek_pam_prefetch_vectors(ek_ctx *c, int count, int32_t win_lo,
                                   int32_t win_count, bool local,
                                   bool prepared = false)
{
    c->tab_n = 0;
    c->sp_ready = false;
    const int K = c->med_K;
    if (c->pf_backoff > 0)
        --c->pf_backoff;
    else if (c->prune && c->state_exact && c->A >= 3 && win_count > 0 &&
             c->n >= 16384) {
        if (!c->act_n_host)
            EK_HIP(hipHostMalloc((void **)&c->act_n_host, sizeof(unsigned int),
                                 hipHostMallocDefault));
        // O (old medoids of the window's clusters) only where the window's slots
        // and the proposals coincide: ek_pam_window_run's pruning reads it
        const bool slots = local && win_count == count;
        const size_t tb = (size_t)EK_PAM_WIN * (c->med_cap + 1);
        const int groups = (count + EK_PAM_GROUP - 1) / EK_PAM_GROUP;
        ek_launch_pam_tables(c->med_aos, c->med_G, c->A, K, c->pam_restore,
                             c->pam_recs, count, win_lo, slots ? count : 0, c->dtab,
                             c->dtab + tb, c->dtab + 2 * tb, c->stream);
        c->tab_lo = win_lo;
        c->tab_n = slots ? count : 0;
        if (!c->act_list)
            EK_HIP(hipMalloc((void **)&c->act_list,
                             (size_t)std::max<int64_t>(c->n, 1) * sizeof(uint32_t)));
        // the vectors go back to +inf: the entries the window before wrote, if
        // that is all there is (before the list is overwritten)
        const bool sparse_reset = c->vecs_rows >= 0 && c->vecs_cols >= count;
        if (sparse_reset)
            ek_launch_pam_vecs_reset(c->act_list, c->vecs_rows, c->vecs_cols, c->n_pad,
                                     c->pam_vecs, c->stream);
        c->vecs_rows = -1;
        ek_launch_pam_active(c->dist, c->assign, c->n, c->dtab + 2 * tb, groups, K,
                             win_lo, win_count, c->act_list, c->amb_count + 3,
                             c->stream, prepared);
        EK_CHECK_LAUNCH();
        EK_HIP(hipMemcpyAsync(c->act_n_host, c->amb_count + 3, sizeof(unsigned int),
                              hipMemcpyDeviceToHost, c->stream));
        EK_HIP(ek_wait(c));
        const int64_t n_act = *c->act_n_host;
        if (n_act * 4 <= c->n) {
            // a short list: straight from the frame-major copy, 64 frames x the
            // proposals per workgroup, results scattered into the full vectors
            // (a quarter of the frames costs about what the passes over all of
            // them do)
            if (!sparse_reset)
                EK_HIP(hipMemsetD32Async((hipDeviceptr_t)c->pam_vecs, 0x7f800000,
                                         (size_t)count * c->n_pad, c->stream));
            ek_launch_pam_list_dist(c->aos, c->G, c->A, c->act_list, n_act, c->pam_recs,
                                    count, c->pam_vecs, c->n_pad, c->stream);
            EK_CHECK_LAUNCH();
            c->vecs_rows = n_act;
            c->vecs_cols = sparse_reset ? c->vecs_cols : count;
            ++c->pf_sparse;
            // the list stays in c->amb until something else uses it: a window run
            // right away may work from it (ek_pam_sparse.hip)
            c->sp_ready = slots && n_act <= EK_SP_CAP;
            c->sp_nact = n_act;
            return EK_OK;
        }
        // too many frames within reach (large clusters): the test cost a table,
        // a scan and a read-back for nothing -- leave it out for a while
        c->pf_backoff = 15;
    }
    // a pass over all frames per group of EK_PAM_GROUP proposals
    c->vecs_rows = -1;          // (whole vectors are written)
    const size_t rstride = ek_rec_bytes(c->A);
    for (int g0 = 0; g0 < count; g0 += EK_PAM_GROUP)
        ek_launch_pass_dist(std::min(count - g0, EK_PAM_GROUP), c->tiles, c->G,
                            c->pam_vecs + (size_t)g0 * c->n_pad, c->n, c->n_pad,
                            c->A, c->pam_recs + (size_t)g0 * rstride, c->pam_plan,
                            c->ctile, c->ctrace, c->stream, prepared && g0 == 0);
    EK_CHECK_LAUNCH();
    ++c->pf_full;
    return EK_OK;
}

static int ek_pam_prefetch_frames(ek_ctx *c, const int64_t *frames, int32_t count,
                                  int32_t win_lo, int32_t win_count);

extern "C" int ek_pam_prefetch_window(ek_ctx *c, const int64_t *frames,
                                      int32_t count, int32_t win_lo,
                                      int32_t win_count)
{
    if (c && (win_lo < 0 || win_count < 0 || win_lo + win_count > c->med_K))
        return ek_fail(EK_EARG, "ek_pam_prefetch_window: clusters [%d,+%d) outside "
                                "[0,%d)", win_lo, win_count, c->med_K);
    return ek_pam_prefetch_frames(c, frames, count, win_lo, win_count);
}

extern "C" int ek_pam_prefetch(ek_ctx *c, const int64_t *frames, int32_t count)
{
    return ek_pam_prefetch_frames(c, frames, count, 0, 0);
}

static int ek_pam_prefetch_frames(ek_ctx *c, const int64_t *frames, int32_t count,
                                  int32_t win_lo, int32_t win_count)
{
    if (!c || (!frames && count > 0))
        return ek_fail(EK_EARG, "ek_pam_prefetch: NULL argument");
    if (!c->ndist || c->med_K < 1)
        return ek_fail(EK_ESTATE, "ek_pam_prefetch: call ek_pam_begin first");
    if (count < 0 || count > EK_PAM_WIN)
        return ek_fail(EK_EARG, "ek_pam_prefetch: count=%d outside [0,%d]", count,
                       EK_PAM_WIN);
    for (int32_t j = 0; j < count; ++j)
        if (frames[j] < 0 || frames[j] >= c->n)
            return ek_fail(EK_EARG, "ek_pam_prefetch: frame %lld out of range",
                           (long long)frames[j]);
    EK_HIP(hipSetDevice(c->device));
    c->pf_count = 0;
    if (count == 0)
        return EK_OK;
    {
        int rc = ek_pam_vecs_alloc(c);
        if (rc)
            return rc;
    }
    const bool prepared = c->ctile != nullptr;
    if (prepared)
        ek_launch_pam_setup(c->aos, c->G, c->A, frames, count, c->goff, c->pam_recs,
                            c->ctile, c->ctrace, c->pam_plan, c->amb_count + 3,
                            c->stream);
    else
        ek_launch_records_from_frames(c->aos, c->G, c->A, frames, count, c->goff,
                                      c->pam_recs, c->stream);
    {
        int rc = ek_pam_prefetch_vectors(c, count, win_lo, win_count, true,
                                         prepared);
        if (rc)
            return rc;
    }
    for (int32_t j = 0; j < count; ++j)
        c->pf_frames[j] = frames[j];
    c->pf_count = count;
    c->pf_external = false;
    return EK_OK;
}

// The same window worked through by one workgroup (ek_pam_sparse.hip): possible
// when the prefetch just made was restricted to a list of frames (still in
// c->amb) and the window's tables are in place.  Enqueues; the caller reads the
// window record back.
static int ek_pam_window_sparse(ek_ctx *c, int32_t cid0, int32_t count,
                                const int64_t *frames, const int64_t *n_members,
                                int32_t win_count)
{
    const size_t cap = EK_SP_CAP;
    const size_t o_bucket = 0;
    const size_t o_bcnt = o_bucket + (size_t)EK_PAM_WIN * cap * sizeof(uint2);
    if (!c->sp_buf)
        EK_HIP(hipMalloc((void **)&c->sp_buf, o_bcnt + 256));
    const int K = c->med_K;
    const size_t tb = (size_t)EK_PAM_WIN * (c->med_cap + 1);
    // the state's cost tree, the slots' frames
    ek_launch_pw_tree(c->dist, c->assign, c->n, c->pw_shapes, c->pw_n_full,
                      c->pw_leaves, c->pw_chunks, c->sq_part, c->moved, c->stream);
    ek_launch_sp_bucket(c->act_list, c->sp_nact, c->dist, c->assign, c->pam_vecs, c->n_pad,
                        cid0, count, (uint2 *)(c->sp_buf + o_bucket),
                        (unsigned int *)(c->sp_buf + o_bcnt), (int64_t)cap, c->stream);
    EkSpArgs a = {};
    a.dist = c->dist;
    a.assign = c->assign;
    a.n = c->n;
    a.n_total = (double)c->n;
    a.A = c->A;
    a.K = K;
    a.cid0 = cid0;
    a.count = count;
    a.win_count = win_count;
    a.bucket = (const uint2 *)(c->sp_buf + o_bucket);
    a.bcnt = (const unsigned int *)(c->sp_buf + o_bcnt);
    a.bcap = (int64_t)cap;
    for (int32_t i = 0; i < EK_PAM_WIN; ++i) {
        a.frames[i] = i < count ? frames[i] : 0;
        a.max_amb[i] = i < count ? n_members[i] : 0;
    }
    a.O = c->dtab + tb;
    a.T = c->dtab;
    a.med_aos = c->med_aos;
    a.med_G = c->med_G;
    a.med_idx = c->med_idx;
    a.restore = c->pam_restore;
    a.frames_aos = c->aos;
    a.G = c->G;
    a.leaf = c->sq_part;
    a.chunk = c->sq_part + 2 * (size_t)c->pw_leaves;
    a.shapes = c->pw_shapes;
    a.n_full = c->pw_n_full;
    a.n_leaves = c->pw_leaves;
    a.n_chunks = c->pw_chunks;
    a.max_pairs = c->sp_max_pairs;
    a.exact_always = c->sp_exact;
    a.win = c->pam_win_dev;
#ifdef EK_SP_PROF
    static unsigned long long *prof_dev = nullptr;
    static unsigned long long prof_tot[16];
    static int prof_n = 0;
    if (!prof_dev) {
        EK_HIP(hipMalloc((void **)&prof_dev, 16 * sizeof(unsigned long long)));
        EK_HIP(hipMemset(prof_dev, 0, 16 * sizeof(unsigned long long)));
    }
    a.prof = prof_dev;
    if (++prof_n % 100 == 0) {
        EK_HIP(ek_wait(c));
        EK_HIP(hipMemcpy(prof_tot, prof_dev, sizeof(prof_tot), hipMemcpyDeviceToHost));
        fprintf(stderr, "sp prof after %d windows (ms): classify %.2f search %.2f apply %.2f "
                        "leaves %.2f chunks %.2f total %.2f verdict %.2f keep/undo %.2f [issue %.2f loop %.2f]\n",
                prof_n, prof_tot[0] * 1e-5, prof_tot[1] * 1e-5, prof_tot[2] * 1e-5,
                prof_tot[3] * 1e-5, prof_tot[4] * 1e-5, prof_tot[5] * 1e-5,
                prof_tot[6] * 1e-5, prof_tot[7] * 1e-5, prof_tot[8] * 1e-5, prof_tot[9] * 1e-5);
    }
#endif
    ek_launch_sp_window(a, c->stream);
    EK_CHECK_LAUNCH();
    c->pam_restore = -1;
    c->sp_ready = false;
    c->pf_hits += count;
    ++c->sp_windows;
    return EK_OK;
}

// A window of proposals without a host round trip each (reference
// kmedoids.py:575-699 for clusters cid0 .. cid0 + count - 1, in order).  frames[i]
// is the frame proposed for cluster cid0 + i -- the caller drew it from the
// member list as it stood when the window was opened -- and n_members[i] that
// list's length; all of them must have been prefetched (ek_pam_prefetch_window).
// Every proposal's kernels are enqueued at once; the device decides each
// (mean of squares, float64, strict <), commits or undoes it, and stops the
// window at the first cluster whose membership an accepted proposal changed:
// *n_done slots were decided, the caller handles slot *n_done one at a time
// (its member list has to be counted again) and opens a new window after it.
// next_count > 0: the member counts of clusters next_cid0 .. +next_count (what the
// window after this one starts with) are taken on the state this window leaves,
// right behind its kernels, and come back with its record -- one wait less per
// window; they stand if the window runs to its end (the caller checks)
static int ek_pam_window_run_impl(ek_ctx *c, int32_t cid0, int32_t count,
                                  const int64_t *frames, const int64_t *n_members,
                                  int32_t win_lo, int32_t win_count,
                                  int32_t *n_done, int32_t *accept,
                                  double *old_cost, double *new_cost,
                                  int64_t *n_ambiguous, int32_t next_cid0,
                                  int32_t next_count, int64_t *next_counts)
{
    int rc = ek_pam_precheck(c, cid0, "ek_pam_window_run");
    if (rc)
        return rc;
    if (count < 1 || count > EK_PAM_WIN || cid0 + count > c->med_K || !frames ||
        !n_members || !n_done || !accept)
        return ek_fail(EK_EARG, "ek_pam_window_run: bad window [%d,+%d)", cid0,
                       count);
    // bit i of a proposal's moved-cluster mask is cluster win_lo + i, and the
    // device reads it as slot i of this run
    if (win_lo != cid0 || win_count < count || win_count > 32)
        return ek_fail(EK_EARG, "ek_pam_window_run: the stale-mask window "
                                "[%d,+%d) must start at cid0 = %d and cover the "
                                "%d slots", win_lo, win_count, cid0, count);
    if (!c->pw_tail_ok)
        return ek_fail(EK_ESTATE, "ek_pam_window_run: the pairwise-sum shape of a "
                                  "full chunk is not the expected perfect tree");
    EK_HIP(hipSetDevice(c->device));
    int64_t max_m = 0;
    const float *newd[EK_PAM_WIN];
    for (int32_t i = 0; i < count; ++i) {
        if (frames[i] < 0 || frames[i] >= c->n || n_members[i] < 0 ||
            n_members[i] > c->n)
            return ek_fail(EK_EARG, "ek_pam_window_run: slot %d: frame %lld, %lld "
                                    "members", i, (long long)frames[i],
                           (long long)n_members[i]);
        newd[i] = ek_pam_prefetched(c, frames[i]);
        if (!newd[i])
            return ek_fail(EK_ESTATE, "ek_pam_window_run: frame %lld was not "
                                      "prefetched", (long long)frames[i]);
        max_m = std::max(max_m, n_members[i]);
    }
    // (a window that had to hand a proposal back -- more ambiguous members x
    // medoids within reach than one workgroup should search -- cost a window's
    // set-up for one slot: the next few go the three-launch way, twice as many
    // each time it happens again)
    if (c->sp_backoff > 0)
        --c->sp_backoff;
    bool sparse = c->pam_sparse && c->sp_ready && c->tab_lo == cid0 &&
                  count <= c->tab_n && c->prune && c->state_exact && c->A >= 3 &&
                  c->aos != nullptr && c->pw_chunks <= EK_SP_MAX_CHUNKS &&
                  c->pw_leaves <= EK_SP_MAX_CHUNKS * EK_PW_FULL_LEAVES &&
                  (c->sp_backoff == 0 || c->sp_max_pairs == 0);
    for (int32_t i = 0; sparse && i < count; ++i)
        sparse = newd[i] == c->pam_vecs + (size_t)i * c->n_pad;
    if (sparse) {
        rc = ek_pam_window_sparse(c, cid0, count, frames, n_members, win_count);
        if (rc)
            return rc;
    } else {
        rc = ek_pam_amb_room(c, max_m);  // may synchronise: before anything is enqueued
        if (rc)
            return rc;
    }
    const int K = c->med_K;
    for (int32_t i = 0; !sparse && i < count; ++i) {
        const int32_t cid = cid0 + i;
        // the first proposal's trial table (and the window record: `count`
        // slots, nothing decided); the others' are set up by the last workgroup
        // of the proposal before
        if (i == 0)
            ek_launch_pam_trial(c->tiles, c->G, c->A, c->med_aos, c->med_G, K, cid,
                                c->pam_restore, frames[i], nullptr, nullptr,
                                nullptr, c->amb_count, c->moved, c->stream,
                                c->pam_win_dev, count);
        c->pam_restore = -1;
        const bool more = i + 1 < count;
        EkPamDecide dc;
        dc.win = c->pam_win_dev;
        dc.slot = i;
        dc.n_total = (double)c->n;
        dc.aos = c->med_aos;
        dc.Gm = c->med_G;
        dc.A = c->A;
        dc.K = K;
        dc.cid = cid;
        dc.med_idx = c->med_idx;
        dc.frame = frames[i];
        dc.max_amb = n_members[i];
        dc.next_cid = more ? cid + 1 : -1;
        dc.next_frame = more ? frames[i + 1] : 0;
        dc.frames_aos = c->aos;
        dc.G = c->G;
        dc.amb_count = c->amb_count;
        dc.moved = c->moved;
        rc = ek_pam_tail(c, cid, newd[i], n_members[i], win_lo, win_count,
                         &c->pam_win_dev->out[i], &dc,
                         i > 0 ? &c->pam_win_dev->accept[i - 1] : &c->pam_win_dev->pad);
        if (rc)
            return rc;
        ++c->pf_hits;
    }
    // the last slot's trial state, if accepted (the others were taken over by
    // the classification of the slot after them)
    if (!sparse)
        ek_launch_pam_apply(&c->pam_win_dev->accept[count - 1], c->dist, c->ndist,
                            c->assign, c->nassign, c->n, c->stream);
    EK_CHECK_LAUNCH();
    EK_HIP(hipMemcpyAsync(c->pam_win_host, c->pam_win_dev, sizeof(EkPamWin),
                          hipMemcpyDeviceToHost, c->stream));
    if (next_count > 0) {
        ek_launch_count_members_multi(c->assign, c->n, next_cid0, next_count,
                                      c->bat_blockcnt, c->bat_scan, c->bat_sel, c->stream);
        EK_CHECK_LAUNCH();
        EK_HIP(hipMemcpyAsync(next_counts, c->bat_sel, (size_t)next_count * sizeof(int64_t),
                              hipMemcpyDeviceToHost, c->stream));
    }
    EK_HIP(ek_wait(c));
    c->bat_cid0 = -1;
    c->bat_count = 0;
    const EkPamWin &w = *c->pam_win_host;
    c->tab_n = 0;            // the medoid table has moved on
    c->pam_cid = -1;
    c->cnt_cid = -1;
    c->pf_hits -= count - w.stop;       // the slots past the stop were not served
    if (sparse) {
        if (w.pad) {
            ++c->sp_bailed;
            c->sp_backoff = c->sp_backoff_next;
            c->sp_backoff_next = std::min(256, 2 * c->sp_backoff_next);
        } else {
            c->sp_backoff_next = 8;
        }
    }
    if (w.err)
        return ek_fail(EK_EARG, "PAM proposal: cluster %d has more members that "
                                "stay put than the %lld members declared",
                       cid0 + w.err - 1, (long long)n_members[w.err - 1]);
    *n_done = w.stop;
    for (int32_t i = 0; i < count; ++i) {
        accept[i] = i < w.stop ? w.accept[i] : 0;
        if (old_cost)
            old_cost[i] = w.out[i].sum_old / (double)c->n;
        if (new_cost)
            new_cost[i] = w.out[i].sum_new / (double)c->n;
        if (n_ambiguous)
            n_ambiguous[i] = w.out[i].n_amb;
    }
    if (next_count > 0 && w.stop == count) {
        // the state the counts were taken on is the one the next window opens with
        c->bat_cid0 = next_cid0;
        c->bat_count = next_count;
    }
    return EK_OK;
}

extern "C" int ek_pam_window_run(ek_ctx *c, int32_t cid0, int32_t count,
                                 const int64_t *frames, const int64_t *n_members,
                                 int32_t win_lo, int32_t win_count,
                                 int32_t *n_done, int32_t *accept,
                                 double *old_cost, double *new_cost,
                                 int64_t *n_ambiguous)
{
    return ek_pam_window_run_impl(c, cid0, count, frames, n_members, win_lo, win_count,
                                  n_done, accept, old_cost, new_cost, n_ambiguous, 0, 0,
                                  nullptr);
}

// ---- a whole sweep's window loop on the host side of the library ------------------------
// numpy's legacy RandomState.choice(m) / randint(0, m): 32-bit outputs of the
// Mersenne Twister masked to the bits of m - 1, values above it rejected; m == 1
// consumes nothing.  `raw` are such outputs, *pos the next unused one.
// -> 0: drawn, 1: the outputs ran out (nothing consumed)
static int ek_draw_at(const uint32_t *raw, int64_t n_raw, int64_t *pos, int64_t m,
                      int64_t *out)
{
    const uint64_t rng = (uint64_t)(m - 1);
    if (rng == 0) {
        *out = 0;
        return 0;
    }
    uint64_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    int64_t p = *pos;
    for (;;) {
        if (p >= n_raw)
            return 1;
        const uint64_t v = raw[p++] & mask;
        if (v <= rng) {
            *out = (int64_t)v;
            *pos = p;
            return 0;
        }
    }
}

// The draws above, exposed for tests that hold them against numpy itself (no
// device involved): out[i] = RandomState.choice(m[i]) taken from `raw` at *pos on.
// -> how many were made (fewer than count: the outputs ran out, or m[i] < 1)
extern "C" int64_t ek_np_choice_draws(const uint32_t *raw, int64_t n_raw, int64_t *pos,
                                      const int64_t *m, int64_t count, int64_t *out)
{
    if (!raw || !pos || !m || !out || *pos < 0)
        return -1;
    int64_t i = 0;
    for (; i < count; ++i) {
        if (m[i] < 1 || (uint64_t)m[i] > 0x100000000ull)
            break;
        if (ek_draw_at(raw, n_raw, pos, m[i], &out[i]))
            break;
    }
    return i;
}

// The loop of kmedoids.py:575-699 over clusters *cid .. K - 1 in windows of up to
// `width` proposals decided on the device (ek_pam_window_run), the cluster a
// window stops at one proposal at a time -- what enspara_amd/cluster/kmedoids.py's
// _pam_sweep_device_on does call by call, without the interpreter between the
// calls (four read-backs per window, and as many waits for the host to come
// back).  The draws are numpy's: `raw` holds the next outputs of the caller's
// RandomState (RandomState.randint(0, 2**32, dtype=uint32)), *pos how many of
// them the draws made so far have consumed.
// *status: 0 the sweep is through (*cid == K); 1 more random outputs are needed
// (call again with a longer `raw`: *cid, *pos and the outputs so far stand);
// 2 cluster *cid has no member to draw (RandomState.choice raises there).
extern "C" int ek_pam_sweep(ek_ctx *c, int32_t K, int32_t width, const uint32_t *raw,
                            int64_t n_raw, int64_t *pos, const int64_t *proposals,
                            int32_t *cid_io, int64_t *medoids, int32_t *accept,
                            double *old_cost, double *new_cost, int64_t *n_amb,
                            int32_t *status)
{
    if (!c || !pos || !cid_io || !medoids || !accept || !old_cost || !new_cost ||
        !n_amb || !status || (!raw && n_raw > 0))
        return ek_fail(EK_EARG, "ek_pam_sweep: NULL argument");
    if (!c->ndist || c->med_K != K)
        return ek_fail(EK_ESTATE, "ek_pam_sweep: call ek_pam_begin with these %d "
                                  "medoids first", K);
    if (width < 2 || width > EK_PAM_WIN)
        return ek_fail(EK_EARG, "ek_pam_sweep: windows of 2..%d proposals", EK_PAM_WIN);
    if (*cid_io < 0 || *cid_io > K || *pos < 0)
        return ek_fail(EK_EARG, "ek_pam_sweep: cluster %d, position %lld", *cid_io,
                       (long long)*pos);
    if (n_raw > 0 && (uint64_t)c->n > 0xffffffffull)
        return ek_fail(EK_EARG, "ek_pam_sweep: member lists of 2**32 frames and more");
    int32_t cid = *cid_io;
    *status = 0;
    int64_t counts[EK_PAM_WIN], js[EK_PAM_WIN], frames[EK_PAM_WIN], na[EK_PAM_WIN];
    int32_t acc[EK_PAM_WIN];
    double oc[EK_PAM_WIN], nc[EK_PAM_WIN];
    int64_t ahead[EK_PAM_WIN];
    bool have_counts = false;       // `ahead` holds the counts of the window at `cid`
    while (cid < K) {
        const int32_t hi = std::min(K, cid + width), cnt = hi - cid;
        int rc = EK_OK;
        if (have_counts && c->bat_cid0 == cid && c->bat_count == cnt) {
            for (int32_t s = 0; s < cnt; ++s)
                counts[s] = ahead[s];
        } else {
            rc = ek_pam_count_members_batch(c, cid, cnt, counts);
        }
        have_counts = false;
        if (rc)
            return rc;
        int32_t n_slots = 0;
        if (!proposals) {
            // the draws the real stream will produce if these counts still hold
            // when each cluster's turn comes
            int64_t p = *pos;
            for (; n_slots < cnt && counts[n_slots] > 0; ++n_slots)
                if (ek_draw_at(raw, n_raw, &p, counts[n_slots], &js[n_slots])) {
                    *cid_io = cid;
                    *status = 1;
                    return EK_OK;
                }
            if (n_slots > 0) {
                rc = ek_pam_select_members_batch(c, cid, n_slots, js, frames);
                if (rc)
                    return rc;
            }
        } else {
            n_slots = cnt;
            for (int32_t s = 0; s < cnt; ++s)
                frames[s] = proposals[cid + s];
        }
        rc = ek_pam_prefetch_window(c, frames, n_slots, cid, cnt);
        if (rc)
            return rc;
        int32_t n_done = 0;
        if (n_slots > 0) {
            // (the counts the next window starts with ride along when this one
            // covers its clusters: they stand if it runs to its end)
            const int32_t ncnt = (n_slots == cnt) ? std::min(K, hi + width) - hi : 0;
            rc = ek_pam_window_run_impl(c, cid, n_slots, frames, counts, cid, cnt, &n_done,
                                        acc, oc, nc, na, hi, ncnt, ahead);
            if (rc)
                return rc;
            have_counts = ncnt > 0 && n_done == n_slots;
        }
        for (int32_t s = 0; s < n_done; ++s) {
            if (!proposals) {
                // the real draws, in order: the member lists are the ones the
                // guesses were drawn from, so they are the same draws
                int64_t j = -1;
                if (ek_draw_at(raw, n_raw, pos, counts[s], &j) || j != js[s])
                    return ek_fail(EK_ESTATE, "ek_pam_sweep: draw %lld for cluster %d, "
                                              "guessed %lld", (long long)j, cid + s,
                                   (long long)js[s]);
            }
            accept[cid + s] = acc[s];
            old_cost[cid + s] = oc[s];
            new_cost[cid + s] = nc[s];
            n_amb[cid + s] = na[s];
            if (acc[s])
                medoids[cid + s] = frames[s];
        }
        cid += n_done;
        if (cid < hi) {
            // the window stopped here: this cluster's members changed under an
            // accepted proposal (or it is empty) -- counted and drawn now
            int64_t prop = -1;
            double o = 0.0, nw = 0.0;
            int64_t amb = 0;
            if (!proposals) {
                int64_t m = 0;
                rc = ek_pam_count_members(c, cid, &m);
                if (rc)
                    return rc;
                if (m <= 0) {
                    *cid_io = cid;
                    *status = 2;
                    return EK_OK;
                }
                int64_t j = 0;
                if (ek_draw_at(raw, n_raw, pos, m, &j)) {
                    *cid_io = cid;
                    *status = 1;
                    return EK_OK;
                }
                rc = ek_pam_propose_member(c, cid, j, &prop, &o, &nw, &amb);
            } else {
                prop = proposals[cid];
                rc = ek_pam_propose(c, cid, prop, &o, &nw, &amb);
            }
            if (rc)
                return rc;
            const int a = nw < o;                               // kmedoids.py:683
            rc = ek_pam_commit(c, a);
            if (rc)
                return rc;
            accept[cid] = a;
            old_cost[cid] = o;
            new_cost[cid] = nw;
            n_amb[cid] = amb;
            if (a)
                medoids[cid] = prop;
            ++cid;
        }
    }
    *cid_io = cid;
    return EK_OK;
}

extern "C" int ek_pam_propose_ex(ek_ctx *c, int32_t cid, int64_t frame_index,
                                 int64_t n_members, int32_t win_lo,
                                 int32_t win_count, double *old_cost,
                                 double *new_cost, int64_t *n_ambiguous,
                                 uint32_t *moved_mask)
{
    int rc = ek_pam_precheck(c, cid, "ek_pam_propose_ex");
    if (rc)
        return rc;
    if (frame_index < 0 || frame_index >= c->n)
        return ek_fail(EK_EARG, "ek_pam_propose_ex: frame %lld out of range",
                       (long long)frame_index);
    if (n_members < 0 || n_members > c->n)
        return ek_fail(EK_EARG, "ek_pam_propose_ex: n_members=%lld",
                       (long long)n_members);
    if (win_count < 0 || win_count > 32 || (win_count > 0 && !moved_mask))
        return ek_fail(EK_EARG, "ek_pam_propose_ex: bad window");
    EK_HIP(hipSetDevice(c->device));
    c->cnt_cid = -1;
    return ek_pam_propose_impl(c, cid, frame_index, n_members, nullptr, old_cost,
                               new_cost, n_ambiguous, win_lo, win_count,
                               win_count > 0 ? moved_mask : nullptr);
}

static int ek_pam_prefetch_centers_impl(ek_ctx *c, const float *aos_dev,
                                        const double *G_dev, int32_t count,
                                        int32_t win_lo, int32_t win_count);

extern "C" int ek_pam_prefetch_centers(ek_ctx *c, const float *aos_dev,
                                       const double *G_dev, int32_t count)
{
    return ek_pam_prefetch_centers_impl(c, aos_dev, G_dev, count, 0, 0);
}

extern "C" int ek_pam_prefetch_centers_window(ek_ctx *c, const float *aos_dev,
                                              const double *G_dev, int32_t count,
                                              int32_t win_lo, int32_t win_count)
{
    if (c && (win_lo < 0 || win_count < 0 || win_lo + win_count > c->med_K))
        return ek_fail(EK_EARG, "ek_pam_prefetch_centers_window: clusters [%d,+%d) "
                                "outside [0,%d)", win_lo, win_count, c->med_K);
    return ek_pam_prefetch_centers_impl(c, aos_dev, G_dev, count, win_lo, win_count);
}

static int ek_pam_prefetch_centers_impl(ek_ctx *c, const float *aos_dev,
                                        const double *G_dev, int32_t count,
                                        int32_t win_lo, int32_t win_count)
{
    if (!c || (count > 0 && (!aos_dev || !G_dev)))
        return ek_fail(EK_EARG, "ek_pam_prefetch_centers: NULL argument");
    if (!c->ndist || c->med_K < 1)
        return ek_fail(EK_ESTATE, "ek_pam_prefetch_centers: call ek_pam_begin[_table] "
                                  "first");
    if (count < 0 || count > EK_PAM_GROUP)
        return ek_fail(EK_EARG, "ek_pam_prefetch_centers: count=%d outside [0,%d]",
                       count, EK_PAM_GROUP);
    EK_HIP(hipSetDevice(c->device));
    c->pf_count = 0;
    c->pf_external = true;
    if (count == 0)
        return EK_OK;
    int rc = ek_pam_vecs_alloc(c);
    if (rc)
        return rc;
    const size_t rstride = ek_rec_bytes(c->A);
    for (int32_t j = 0; j < count; ++j)
        ek_launch_record_from_center(aos_dev + (size_t)j * 3 * c->A, G_dev + j, c->A,
                                     c->pam_recs + j * rstride, c->stream);
    rc = ek_pam_prefetch_vectors(c, count, win_lo, win_count, false);
    if (rc)
        return rc;
    for (int32_t j = 0; j < count; ++j)
        c->pf_frames[j] = -1;
    c->pf_count = count;
    return EK_OK;
}

extern "C" int ek_pam_propose_center(ek_ctx *c, int32_t cid, int32_t slot,
                                     const float *center_aos_dev,
                                     const double *center_G_dev,
                                     int64_t n_members_local, int32_t win_lo,
                                     int32_t win_count, void *out_dev)
{
    int rc = ek_pam_precheck(c, cid, "ek_pam_propose_center");
    if (rc)
        return rc;
    if (!center_aos_dev || !center_G_dev || !out_dev)
        return ek_fail(EK_EARG, "ek_pam_propose_center: NULL argument");
    if (n_members_local < 0 || n_members_local > c->n)
        return ek_fail(EK_EARG, "ek_pam_propose_center: n_members_local=%lld",
                       (long long)n_members_local);
    if (win_count < 0 || win_count > 32)
        return ek_fail(EK_EARG, "ek_pam_propose_center: bad window");
    if (slot >= 0 && (!c->pf_external || slot >= c->pf_count))
        return ek_fail(EK_ESTATE, "ek_pam_propose_center: slot %d was not "
                                  "prefetched", slot);
    EK_HIP(hipSetDevice(c->device));
    c->cnt_cid = -1;
    rc = ek_pam_amb_room(c, n_members_local);
    if (rc)
        return rc;
    const int K = c->med_K;
    const float *newd;
    if (slot >= 0) {
        newd = c->pam_vecs + (size_t)slot * c->n_pad;
        ++c->pf_hits;
    } else {
        ek_launch_record_from_center(center_aos_dev, center_G_dev, c->A, c->rec_tmp,
                                     c->stream);
        ek_launch_step(ek_pick_fpl(c), 1, ek_pick_nt(c), c->tiles, c->G, c->dist,
                       c->assign, c->scratch, c->rec_tmp, 1, c->n, c->A, 0, 0.0,
                       c->blockmax, c->hist, c->ctl, c->stream);
        EK_CHECK_LAUNCH();
        newd = c->scratch;
        ++c->pf_misses;
    }
    ek_launch_pam_trial(c->tiles, c->G, c->A, c->med_aos, c->med_G, K, cid,
                        c->pam_restore, -1, nullptr, center_aos_dev, center_G_dev,
                        c->amb_count, c->moved, c->stream);
    c->pam_restore = -1;
    rc = ek_pam_tail(c, cid, newd, n_members_local, win_lo, win_count,
                     (EkPamOut *)out_dev);
    if (rc)
        return rc;
    c->pam_cid = cid;
    c->pam_frame = -1;
    return EK_OK;
}

extern "C" int ek_pam_prefetch_passes(ek_ctx *c, int64_t *restricted,
                                      int64_t *full)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (restricted)
        *restricted = c->pf_sparse;
    if (full)
        *full = c->pf_full;
    return EK_OK;
}

extern "C" int ek_pam_sparse_stats(ek_ctx *c, int64_t *windows, int64_t *ended_early)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (windows)
        *windows = c->sp_windows;
    if (ended_early)
        *ended_early = c->sp_bailed;
    return EK_OK;
}

extern "C" int ek_pam_prefetch_stats(ek_ctx *c, int64_t *hits, int64_t *misses)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (hits)
        *hits = c->pf_hits;
    if (misses)
        *misses = c->pf_misses;
    return EK_OK;
}

// ---- multi-candidate rounds across shards --------------------------------------------
extern "C" int ek_spec_candidates(ek_ctx *c)
{
    return c ? ek_pick_cands(c) : 0;
}

extern "C" int ek_spec_begin(ek_ctx *c, int32_t first_label, int32_t limit,
                             void *recs_out)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_spec_begin: no frames loaded");
    if (first_label < 0 || limit < first_label)
        return ek_fail(EK_EARG, "ek_spec_begin: bad label range");
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_ensure_hist(c, limit);
    if (rc)
        return rc;
    rc = ek_spec_alloc(c);
    if (rc)
        return rc;
    const int T = std::max(ek_pick_cands(c), 1);
    EkCtl w;
    memset(&w, 0, sizeof(w));
    w.n_done = first_label;
    w.limit = limit;
    EK_HIP(hipMemcpyAsync(c->ctl, &w, sizeof(w), hipMemcpyHostToDevice,
                          c->stream));
    EK_HIP(ek_wait(c));
    const int nb = (int)((c->n + EK_BLOCK - 1) / EK_BLOCK);
    ek_launch_blockmax(c->dist, c->n, c->blockmax, c->stream);
    ek_launch_pickT(c->blockmax, nb, c->tiles, c->G, c->assign, c->A, T, c->goff,
                    recs_out ? (unsigned char *)recs_out : c->recsT, c->ctl, c->top,
                    c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_round(ek_ctx *c, const void *recs_all, int32_t n_recs,
                             double dist_cutoff)
{
    if (!c || !recs_all || n_recs < 1 || n_recs > 64)
        return ek_fail(EK_EARG, "ek_spec_round: bad argument (1..64 records)");
    EK_HIP(hipSetDevice(c->device));
    const int T = ek_pick_cands(c);
    if (T < 4)
        return ek_fail(EK_ESTATE, "ek_spec_round: multi-candidate rounds are "
                                  "off (use ek_kcenters_step)");
    if (!c->vecs)
        return ek_fail(EK_ESTATE, "ek_spec_round: call ek_spec_begin first");
    ek_launch_plan((const unsigned char *)recs_all, n_recs, c->A, T, dist_cutoff, c->planD,
                   c->plan, c->hist, c->ctl, c->stream);
    const bool sample = c->samp_every > 0 &&
                        (c->samp_count++ % c->samp_every) == 0 &&
                        2 * (size_t)c->samp_used + 1 < c->samp_ev.size();
    if (sample)
        c->samp_form[c->samp_used] = T;
    if (sample)
        EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used], c->stream));
    if (T == 16) {
        const int eq = ek_ensure_qtiles(c);
        if (eq != EK_OK)
            return eq;
    }
    ek_launch_pass(T, c->tiles, c->qtiles, c->G, c->dist, c->assign, c->vecs,
                   c->n, c->n_pad, c->A, (const unsigned char *)recs_all,
                   c->plan, c->blockmax, c->ctile, c->ctrace, c->stream);
    if (sample) {
        EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used + 1], c->stream));
        c->samp_used++;
    }
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_localmax(ek_ctx *c, void *hdr_out)
{
    if (!c || !hdr_out)
        return ek_fail(EK_EARG, "ek_spec_localmax: NULL argument");
    EK_HIP(hipSetDevice(c->device));
    const int nb = (int)((c->n + EK_BLOCK - 1) / EK_BLOCK);
    ek_launch_localmax(c->blockmax, nb, c->goff, (EkMaxHdr *)hdr_out, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_apply(ek_ctx *c, const void *hdrs_all, int32_t n_hdrs,
                             double dist_cutoff)
{
    if (!c || !hdrs_all || n_hdrs < 1)
        return ek_fail(EK_EARG, "ek_spec_apply: bad argument");
    EK_HIP(hipSetDevice(c->device));
    ek_launch_check((const EkMaxHdr *)hdrs_all, n_hdrs, dist_cutoff, c->plan,
                    c->hist, c->ctl, c->stream);
    ek_launch_apply(c->vecs, c->G, c->n, c->n_pad, c->A, c->dist, c->assign,
                    c->plan, c->blockmax, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

// chained form of the cheap steps (ek_chain.hip): rows -> [all-gather] -> order +
// per-prefix maxima -> [all-gather] -> decide + apply
extern "C" int ek_spec_chain_rows(ek_ctx *c, void *rows_out)
{
    if (!c || !rows_out)
        return ek_fail(EK_EARG, "ek_spec_chain_rows: NULL argument");
    if (!c->vecs || !c->pm)
        return ek_fail(EK_ESTATE, "ek_spec_chain_rows: call ek_spec_begin first");
    EK_HIP(hipSetDevice(c->device));
    ek_launch_chain_rows(c->plan, c->dist, c->vecs, c->n, c->n_pad, c->goff,
                         (EkChainRow *)rows_out, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_chain_max(ek_ctx *c, const void *rows_all, int32_t n_shards,
                                 void *hdrs_out)
{
    if (!c || !rows_all || !hdrs_out || n_shards < 1)
        return ek_fail(EK_EARG, "ek_spec_chain_max: bad argument");
    if (!c->vecs || !c->pm)
        return ek_fail(EK_ESTATE, "ek_spec_chain_max: call ek_spec_begin first");
    EK_HIP(hipSetDevice(c->device));
    // (order, per-prefix maxima and this shard's headers in one launch)
    ek_launch_chain_max2(c->dist, c->vecs, c->n, c->n_pad, c->plan,
                         (const EkChainRow *)rows_all, n_shards, c->blockmax, c->pm,
                         c->goff, (EkMaxHdr *)hdrs_out, c->tick + 3, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_chain_apply(ek_ctx *c, const void *hdrs_all,
                                   int32_t n_shards, double dist_cutoff)
{
    if (!c || !hdrs_all || n_shards < 1)
        return ek_fail(EK_EARG, "ek_spec_chain_apply: bad argument");
    if (!c->vecs || !c->pm)
        return ek_fail(EK_ESTATE, "ek_spec_chain_apply: call ek_spec_begin first");
    EK_HIP(hipSetDevice(c->device));
    ek_launch_chain_decide((const EkMaxHdr *)hdrs_all, n_shards, dist_cutoff,
                           c->plan, c->hist, c->ctl, c->stream);
    ek_launch_chain_apply(c->vecs, c->n, c->n_pad, c->dist, c->assign, c->plan,
                          c->blockmax, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_chain_bytes(int32_t *rows_bytes, int32_t *hdrs_bytes)
{
    if (rows_bytes)
        *rows_bytes = (int32_t)(EK_MAX_CANDS * sizeof(EkChainRow));
    if (hdrs_bytes)
        *hdrs_bytes = (int32_t)(EK_MAX_CANDS * sizeof(EkMaxHdr));
    return EK_OK;
}

extern "C" int ek_spec_round_end(ek_ctx *c, void *recs_out)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    const int T = std::max(ek_pick_cands(c), 1);
    const int nb = (int)((c->n + EK_BLOCK - 1) / EK_BLOCK);
    ek_launch_pickT(c->blockmax, nb, c->tiles, c->G, c->assign, c->A, T, c->goff,
                    recs_out ? (unsigned char *)recs_out : c->recsT, c->ctl, c->top,
                    c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_spec_rounds(ek_ctx *c, int32_t *rounds)
{
    if (!c || !rounds)
        return ek_fail(EK_EARG, "ek_spec_rounds: NULL argument");
    EK_HIP(hipSetDevice(c->device));
    EkCtl r;
    EK_HIP(hipMemcpyAsync(&r, c->ctl, sizeof(r), hipMemcpyDeviceToHost,
                          c->stream));
    EK_HIP(ek_wait(c));
    *rounds = r.n_rounds;
    return EK_OK;
}

extern "C" int ek_ti_stats(ek_ctx *c, int64_t *tiles, int64_t *skipped)
{
    if (!c || !tiles || !skipped)
        return ek_fail(EK_EARG, "ek_ti_stats: NULL argument");
    if (c->ti_tab_n > 0 && c->ti_stats) {       // sharded steps: counted on the device
        unsigned long long st[2] = {0, 0};
        EK_HIP(hipSetDevice(c->device));
        EK_HIP(hipMemcpyAsync(st, c->ti_stats, sizeof(st), hipMemcpyDeviceToHost,
                              c->stream));
        EK_HIP(ek_wait(c));
        c->ti_tiles = (int64_t)st[0];
        c->ti_skipped = (int64_t)st[1];
    }
    *tiles = c->ti_tiles;
    *skipped = c->ti_skipped;
    return EK_OK;
}

extern "C" int ek_run_stats(ek_ctx *c, int64_t *passes, int64_t *centers)
{
    if (!c || !passes || !centers)
        return ek_fail(EK_EARG, "ek_run_stats: NULL argument");
    for (int m = 0; m < 4; ++m) {
        passes[m] = c->st_rounds[m];
        centers[m] = c->st_centers[m];
    }
    return EK_OK;
}

extern "C" int ek_spec_progress(ek_ctx *c, int32_t *n_done, int32_t *stopped)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    EkCtl r;
    EK_HIP(hipMemcpyAsync(&r, c->ctl, sizeof(r), hipMemcpyDeviceToHost,
                          c->stream));
    EK_HIP(ek_wait(c));
    if (n_done)
        *n_done = r.n_done;
    if (stopped)
        *stopped = r.stopped;
    return EK_OK;
}

// ---- rounds across shards: one exchange per round (ek_mshard.hip) ------------------------
static int ek_ms_offer(int world) { return std::max(1, 64 / std::max(world, 1)); }

static void ek_round_of(ek_ctx *c, int T, double cutoff, EkRound &R)
{
    R.dist = c->dist;
    R.assign = c->assign;
    R.vecs = c->vecs;
    R.n = c->n;
    R.n_pad = c->n_pad;
    R.goff = c->goff;
    R.A = c->A;
    R.T = T;
    R.tiles = c->tiles;
    R.qtiles = c->qtiles;
    R.aos = c->aos;
    R.G = c->G;
    R.recs = c->recsT;
    R.plan = c->plan;
    R.pend = c->pend;
    R.ord = c->ord;
    R.blockmax = c->blockmax;
    R.pm = c->pm;
    R.top = c->top;
    R.ctile = c->ctile;
    R.ctrace = c->ctrace;
    R.hist = c->hist;
    R.ctl = c->ctl;
    R.tick = c->tick;
    R.rows = c->rows;
    R.vmask = c->vmask;
    R.cutoff = cutoff;
}

extern "C" int ek_ms_setup(ek_ctx *c, int32_t world, int32_t rank,
                           size_t *message_bytes)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (world < 1 || world > EK_MS_MAX_WORLD || rank < 0 || rank >= world)
        return ek_fail(EK_EARG, "ek_ms_setup: world=%d rank=%d (1..%d shards)", world,
                       rank, EK_MS_MAX_WORLD);
    EK_HIP(hipSetDevice(c->device));
    EK_HIP(ek_wait(c));
    for (void *m : c->ms_ipc)
        (void)hipIpcCloseMemHandle(m);
    c->ms_ipc.clear();
    (void)hipFree(c->ms_mbox);
    (void)hipFree(c->ms_flags);
    c->ms_mbox = nullptr;
    c->ms_flags = nullptr;
    if (!c->ms)     // (+ a scratch control block for ek_ms_end's pick)
        EK_HIP(hipMalloc((void **)&c->ms, 64 + sizeof(EkCtl)));
    EkMsXchg x;
    x.world = world;
    x.rank = rank;
    x.offer = ek_ms_offer(world);
    x.msg_bytes = ek_ms_msg_bytes(c->A, x.offer);
    const size_t mb = 2 * (size_t)world * x.msg_bytes;
    const size_t fb = 2 * (size_t)world * 16 * sizeof(uint32_t);
    // Uncached (fine-grained) device memory: a peer's stores -- another GPU's over
    // xGMI, or another XCD's of this one -- must be what a polling load sees.  In
    // ordinary (coarse-grained) memory an XCD's L2 keeps the line a poll fetched
    // too early, whatever scope the load names: measured, two shards on one GPU
    // that started an exchange at the same moment waited for each other's flag
    // until the time-out.
    EK_HIP(hipExtMallocWithFlags((void **)&c->ms_mbox, mb, hipDeviceMallocUncached));
    EK_HIP(hipExtMallocWithFlags((void **)&c->ms_flags, fb, hipDeviceMallocUncached));
    EK_HIP(hipMemsetAsync(c->ms_mbox, 0, mb, c->stream));
    EK_HIP(hipMemsetAsync(c->ms_flags, 0, fb, c->stream));
    // the sequence numbers restart with the mailboxes (and with them the
    // helpers' go-ahead word, which carries one)
    EK_HIP(hipMemsetAsync(c->ms, 0, 64 + sizeof(EkCtl), c->stream));
    EK_HIP(hipMemsetAsync(c->tick + 5, 0, 2 * sizeof(unsigned int), c->stream));
    EK_HIP(ek_wait(c));
    c->ms_x = x;
    c->ms_peers = 0;
    if (message_bytes)
        *message_bytes = x.msg_bytes;
    return EK_OK;
}

extern "C" int ek_ms_mailbox(ek_ctx *c, void **mbox, void **flags, void *ipc_mbox,
                             void *ipc_flags)
{
    if (!c || !c->ms_mbox)
        return ek_fail(EK_ESTATE, "ek_ms_mailbox: call ek_ms_setup first");
    EK_HIP(hipSetDevice(c->device));
    if (mbox)
        *mbox = c->ms_mbox;
    if (flags)
        *flags = c->ms_flags;
    static_assert(sizeof(EkMsState) <= 64, "scratch control block behind it");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handles travel as 64 bytes");
    if (ipc_mbox)
        EK_HIP(hipIpcGetMemHandle((hipIpcMemHandle_t *)ipc_mbox, c->ms_mbox));
    if (ipc_flags)
        EK_HIP(hipIpcGetMemHandle((hipIpcMemHandle_t *)ipc_flags, c->ms_flags));
    return EK_OK;
}

extern "C" int ek_ms_connect(ek_ctx *c, int32_t peer, void *mbox, void *flags,
                             const void *ipc_mbox, const void *ipc_flags)
{
    if (!c || !c->ms_mbox)
        return ek_fail(EK_ESTATE, "ek_ms_connect: call ek_ms_setup first");
    if (peer < 0 || peer >= c->ms_x.world)
        return ek_fail(EK_EARG, "ek_ms_connect: peer %d of %d", peer, c->ms_x.world);
    EK_HIP(hipSetDevice(c->device));
    if (peer == c->ms_x.rank) {
        mbox = c->ms_mbox;
        flags = c->ms_flags;
    } else if (ipc_mbox && ipc_flags) {
        hipIpcMemHandle_t hm, hf;
        memcpy(&hm, ipc_mbox, sizeof(hm));
        memcpy(&hf, ipc_flags, sizeof(hf));
        EK_HIP(hipIpcOpenMemHandle(&mbox, hm, hipIpcMemLazyEnablePeerAccess));
        c->ms_ipc.push_back(mbox);
        EK_HIP(hipIpcOpenMemHandle(&flags, hf, hipIpcMemLazyEnablePeerAccess));
        c->ms_ipc.push_back(flags);
    }
    if (!mbox || !flags)
        return ek_fail(EK_EARG, "ek_ms_connect: no address for peer %d", peer);
    if (!c->ms_x.dst[peer])
        c->ms_peers++;
    c->ms_x.dst[peer] = (unsigned char *)mbox;
    c->ms_x.dflag[peer] = (uint32_t *)flags;
    return EK_OK;
}

static int ek_ms_check(ek_ctx *c, const char *who)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "%s: no frames loaded", who);
    if (!c->ms_mbox)
        return ek_fail(EK_ESTATE, "%s: call ek_ms_setup first", who);
    return EK_OK;
}

static int ek_ms_begin_T(ek_ctx *c, int32_t first_label, int32_t limit, int T);

extern "C" int ek_ms_begin(ek_ctx *c, int32_t first_label, int32_t limit)
{
    int rc = ek_ms_check(c, "ek_ms_begin");
    if (rc)
        return rc;
    if (first_label < 0 || limit < first_label)
        return ek_fail(EK_EARG, "ek_ms_begin: bad label range");
    const int T = ek_pick_cands(c);
    if (T < 4)
        return ek_fail(EK_ESTATE, "ek_ms_begin: multi-candidate rounds are off "
                                  "(option key 4 = 1: use ek_kcenters_step)");
    return ek_ms_begin_T(c, first_label, limit, T);
}

// rounds of T candidates from the state as it stands (also where a run changes
// its form: ek_ms_run)
static int ek_ms_begin_T(ek_ctx *c, int32_t first_label, int32_t limit, int T)
{
    int rc;
    EK_HIP(hipSetDevice(c->device));
    rc = ek_ensure_hist(c, limit);
    if (rc)
        return rc;
    rc = ek_spec_alloc(c);
    if (rc)
        return rc;
    c->ms_T = T;
    if (T == 16) {
        const int eq = ek_ensure_qtiles(c);
        if (eq != EK_OK)
            return eq;
    }
    EkCtl w;
    memset(&w, 0, sizeof(w));
    w.n_done = first_label;
    w.limit = limit;
    EK_HIP(hipMemcpyAsync(c->ctl, &w, sizeof(w), hipMemcpyHostToDevice, c->stream));
    EK_HIP(hipMemsetAsync(c->plan, 0, sizeof(EkPlan), c->stream));
    EK_HIP(hipMemsetAsync(c->pend, 0, sizeof(EkPend), c->stream));
    EK_HIP(hipMemsetAsync(c->ord, 0, sizeof(EkChainOrd), c->stream));
    // the first exchange offers the records of the state as it stands
    const int32_t start[2] = {2, 0};
    EK_HIP(hipMemcpyAsync(c->ms, start, sizeof(start), hipMemcpyHostToDevice,
                          c->stream));
    EK_HIP(ek_wait(c));
    ek_launch_blockmax(c->dist, c->n, c->blockmax, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

// the launches of a round before its exchange: pass, chain (message out)
static int ek_ms_enqueue_local(ek_ctx *c, const EkRound &R, const EkMsXchg &x)
{
    const bool sample = c->samp_every > 0 &&
                        (c->samp_count++ % c->samp_every) == 0 &&
                        2 * (size_t)c->samp_used + 1 < c->samp_ev.size();
    if (sample) {
        c->samp_form[c->samp_used] = R.T;
        EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used], c->stream));
    }
    ek_launch_round_pass(R, c->stream, false);
    if (sample) {
        EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used + 1], c->stream));
        c->samp_used++;
    }
    ek_launch_ms_chain(R, c->ms, x, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_ms_local(ek_ctx *c, double dist_cutoff, void *message_out)
{
    int rc = ek_ms_check(c, "ek_ms_local");
    if (rc)
        return rc;
    if (!message_out || !c->ms_T)
        return ek_fail(EK_EARG, "ek_ms_local: no message buffer / ek_ms_begin first");
    EK_HIP(hipSetDevice(c->device));
    EkRound R;
    ek_round_of(c, c->ms_T, dist_cutoff, R);
    EkMsXchg x = c->ms_x;
    x.sys = 0;
    x.dst[0] = (unsigned char *)message_out;
    return ek_ms_enqueue_local(c, R, x);
}

extern "C" int ek_ms_global(ek_ctx *c, double dist_cutoff, const void *messages_all)
{
    int rc = ek_ms_check(c, "ek_ms_global");
    if (rc)
        return rc;
    if (!messages_all || !c->ms_T)
        return ek_fail(EK_EARG, "ek_ms_global: no messages / ek_ms_begin first");
    EK_HIP(hipSetDevice(c->device));
    EkRound R;
    ek_round_of(c, c->ms_T, dist_cutoff, R);
    EkMsXchg x = c->ms_x;
    x.sys = 0;
    x.src = (const unsigned char *)messages_all;
    ek_launch_ms_plan(R, c->ms, x, c->planD, c->stream);
    EK_CHECK_LAUNCH();
    return EK_OK;
}

extern "C" int ek_ms_state(ek_ctx *c, int32_t *mode, int32_t *exchanges, int32_t *err)
{
    if (!c || !c->ms)
        return ek_fail(EK_ESTATE, "ek_ms_state: call ek_ms_setup first");
    EK_HIP(hipSetDevice(c->device));
    EkMsState st;
    EK_HIP(hipMemcpyAsync(&st, c->ms, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    EK_HIP(ek_wait(c));
    if (mode)
        *mode = st.mode;
    if (exchanges)
        *exchanges = (int32_t)st.seq;
    if (err)
        *err = st.err;
    return EK_OK;
}

// after the last round: the accepted chain still pending, and the record of the
// state's farthest point where the other entry points expect it
extern "C" int ek_ms_end(ek_ctx *c)
{
    int rc = ek_ms_check(c, "ek_ms_end");
    if (rc)
        return rc;
    EK_HIP(hipSetDevice(c->device));
    if (c->ms_T) {
        EkRound R;
        ek_round_of(c, c->ms_T, 0.0, R);
        ek_launch_round_flush(R, c->stream);
        if (c->n <= 0)
            EK_HIP(hipMemsetAsync(&c->pend->n, 0, sizeof(int32_t), c->stream));
    }
    else
        ek_launch_blockmax(c->dist, c->n, c->blockmax, c->stream);
    // (the pick leaves this shard's maximum in its control block: not the run's)
    const int nb = (int)((c->n + EK_BLOCK - 1) / EK_BLOCK);
    EkCtl *scratch = (EkCtl *)((unsigned char *)c->ms + 64);
    EK_HIP(hipMemsetAsync(scratch, 0, sizeof(EkCtl), c->stream));
    ek_launch_pick(c->blockmax, nb, c->dist, c->tiles, c->G, c->n, c->A, c->goff,
                   c->rec, scratch, c->stream);
    EK_CHECK_LAUNCH();
    EkMsState st;
    EK_HIP(hipMemcpyAsync(&st, c->ms, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    EK_HIP(ek_wait(c));
    c->ms_T = 0;
    if (st.err) {
        // (the next run starts clean)
        EK_HIP(hipMemsetAsync(&c->ms->err, 0, 2 * sizeof(int32_t), c->stream));
        EK_HIP(ek_wait(c));
        if (st.err >= 0x100)
            return ek_fail(EK_ESTATE, "multi-shard round: the message of shard %d did "
                                      "not arrive (exchange %u)", st.err - 0x100,
                           st.err_seq);
        return ek_fail(EK_ESTATE, "multi-shard round: the helper workgroups were not "
                                  "told the shard's records (exchange %u)", st.err_seq);
    }
    return EK_OK;
}

// the whole loop with the exchange on the device (peer mailboxes): no host, no
// collective in a round
extern "C" int ek_ms_run(ek_ctx *c, int32_t first_label, int32_t max_new,
                         double dist_cutoff, int32_t *n_added,
                         int64_t *center_index_out, float *center_dist_out,
                         float *final_maxdist)
{
    int rc = ek_ms_check(c, "ek_ms_run");
    if (rc)
        return rc;
    if (c->ms_peers != c->ms_x.world)
        return ek_fail(EK_ESTATE, "ek_ms_run: %d of %d peers connected (ek_ms_connect)",
                       c->ms_peers, c->ms_x.world);
    if (first_label < 0 || max_new < 0)
        return ek_fail(EK_EARG, "ek_ms_run: negative argument");
    // Rounds of 8 or of 16 candidates: early in a fit every new center reshapes
    // most frames' distances and a round accepts one to three of its guesses --
    // eight of them then cost less than sixteen.  ek_run_rounds moves between
    // the forms by measured centers per ms; here every shard has to take the
    // SAME decision at the same round, so it is taken from what they all see
    // alike: the centers the rounds of a batch accepted.  Rounds of 8 while they
    // accept fewer than 6.5; a batch of 16 that accepts fewer than 8.5 per round
    // goes back to 8 and the next try waits twice as many batches.  A change of
    // form costs one exchange without a pass (the state's farthest frames are
    // offered again).  Results do not depend on the form.
    const int Tmax = ek_pick_cands(c);
    if (Tmax < 4)
        return ek_fail(EK_ESTATE, "ek_ms_run: multi-candidate rounds are off "
                                  "(option key 4 = 1: use ek_kcenters_step)");
    const bool ladder = Tmax == 16 && c->cands == -1 && c->adapt;
    int T = ladder ? 8 : Tmax;
    rc = ek_ms_begin_T(c, first_label, first_label + max_new, T);
    if (rc)
        return rc;
    EkRound R;
    ek_round_of(c, T, dist_cutoff, R);
    EkMsXchg x = c->ms_x;
    x.sys = 1;
    x.src = c->ms_mbox;
    x.sflag = c->ms_flags;
    EK_HIP(hipEventRecord(c->ev0, c->stream));
    const int32_t goal = first_label + max_new;
    EkCtl cr;
    memset(&cr, 0, sizeof(cr));
    cr.n_done = first_label;
    EkMsState st;
    memset(&st, 0, sizeof(st));
    double per_round = 0.6 * T;
    int32_t rounds_before = 0, passes = 0;
    int wait16 = 0, next_wait = 1;
    for (int k = 0; k < 4; ++k)
        c->st_rounds[k] = c->st_centers[k] = 0;
    while (max_new > 0) {
        const int32_t left = goal - cr.n_done;
        int32_t batch = std::max(2, std::min(256, (int32_t)(left / per_round) + 2));
        if (ladder)
            batch = std::min(batch, T == 8 ? 24 : 64);
        for (int32_t r = 0; r < batch; ++r) {
            rc = ek_ms_enqueue_local(c, R, x);
            if (rc)
                return rc;
            ek_launch_ms_plan(R, c->ms, x, c->planD, c->stream);
            EK_CHECK_LAUNCH();
        }
        const int32_t before = cr.n_done;
        EK_HIP(hipMemcpyAsync(&cr, c->ctl, sizeof(cr), hipMemcpyDeviceToHost,
                              c->stream));
        EK_HIP(hipMemcpyAsync(&st, c->ms, sizeof(st), hipMemcpyDeviceToHost,
                              c->stream));
        EK_HIP(ek_wait(c));
        const int32_t ran = cr.n_rounds - rounds_before;
        rounds_before = cr.n_rounds;
        passes += ran;
        c->st_rounds[ek_form_slot(T)] += ran;
        c->st_centers[ek_form_slot(T)] += cr.n_done - before;
        if (st.err || st.mode == 0 || cr.stopped || cr.n_done >= goal)
            break;
        per_round = std::max(1.0, (double)(cr.n_done - before) / std::max(ran, 1));
        if (ladder && ran > 0) {
            int want = T;
            if (T == 8) {
                if (wait16 > 0)
                    --wait16;
                else if (per_round >= 6.5)
                    want = 16;
            } else if (per_round < 8.5) {
                want = 8;
                wait16 = next_wait;
                next_wait = std::min(2 * next_wait, 64);
            } else {
                next_wait = 1;
            }
            if (want != T) {
                ek_launch_round_flush(R, c->stream);    // the chain still pending
                EK_CHECK_LAUNCH();
                rc = ek_ms_begin_T(c, cr.n_done, goal, want);
                if (rc)
                    return rc;
                T = want;
                ek_round_of(c, T, dist_cutoff, R);
                rounds_before = 0;
                per_round = std::max(per_round, 0.6 * T);
            }
        }
    }
    EK_HIP(hipEventRecord(c->ev1, c->stream));
    rc = ek_ms_end(c);
    if (rc)
        return rc;
    EK_HIP(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    EK_HIP(hipMemcpyAsync(&cr, c->ctl, sizeof(cr), hipMemcpyDeviceToHost, c->stream));
    EK_HIP(ek_wait(c));
    c->last_launches = passes;
    c->last_passes = passes;
    const int32_t added = std::min(max_new, std::max(0, cr.n_done - first_label));
    if (n_added)
        *n_added = added;
    if (final_maxdist)
        *final_maxdist = cr.last_max;
    if (added > 0 && (center_index_out || center_dist_out)) {
        rc = ek_history_download(c, first_label, added, center_index_out,
                                 center_dist_out, nullptr);
        if (rc)
            return rc;
    }
    return EK_OK;
}

// ---- the copy rate of this GPU (bench.py: the ceiling beside the nominal peak) -----------
typedef float ek_probe_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(EK_BLOCK)
ek_copy_probe_kernel(const ek_probe_f4 *__restrict__ src,
                     ek_probe_f4 *__restrict__ dst, size_t n)
{
    // eight 16-byte elements per thread, a workgroup's loads contiguous
    const size_t base = (size_t)blockIdx.x * (EK_BLOCK * 8) + threadIdx.x;
    ek_probe_f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (base + (size_t)u * EK_BLOCK < n)
            v[u] = __builtin_nontemporal_load(&src[base + (size_t)u * EK_BLOCK]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (base + (size_t)u * EK_BLOCK < n)
            __builtin_nontemporal_store(v[u], &dst[base + (size_t)u * EK_BLOCK]);
}

// the same stream read only (what the distance kernels do): 16 bytes per lane,
// non-temporal, summed so that nothing can be dropped
__global__ void __launch_bounds__(EK_BLOCK)
ek_read_probe_kernel(const ek_probe_f4 *__restrict__ src, float *__restrict__ out,
                     size_t n)
{
    const size_t base = (size_t)blockIdx.x * (EK_BLOCK * 8) + threadIdx.x;
    ek_probe_f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        v[u] = base + (size_t)u * EK_BLOCK < n
                   ? __builtin_nontemporal_load(&src[base + (size_t)u * EK_BLOCK])
                   : (ek_probe_f4){0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    if (s == 12345.678f)        // (never: the source is zeros)
        out[blockIdx.x] = s;
}

extern "C" int ek_hbm_copy_rate(int device, size_t bytes, double *gbytes_per_s)
{
    if (!gbytes_per_s || bytes < 16)
        return ek_fail(EK_EARG, "ek_hbm_copy_rate: bad argument");
    EK_HIP(hipSetDevice(device));
    const size_t n = bytes / 16;
    ek_probe_f4 *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&src, n * 16);
    if (e == hipSuccess)
        e = hipMalloc((void **)&dst, n * 16);
    if (e == hipSuccess)
        e = hipMemset(src, 0, n * 16);
    if (e == hipSuccess)
        e = hipEventCreate(&e0);
    if (e == hipSuccess)
        e = hipEventCreate(&e1);
    float best = 0.f, best_read = 0.f;
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)((n + EK_BLOCK * 8 - 1) / (EK_BLOCK * 8));
        for (int rep = 0; rep < 5 && e == hipSuccess; ++rep) {
            (void)hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(ek_copy_probe_kernel, dim3(blocks), dim3(EK_BLOCK), 0,
                               nullptr, src, dst, n);
            (void)hipEventRecord(e1, nullptr);
            e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess)
                e = hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && (best == 0.f || ms < best))
                best = ms;
        }
        for (int rep = 0; rep < 5 && e == hipSuccess; ++rep) {
            (void)hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(ek_read_probe_kernel, dim3(blocks), dim3(EK_BLOCK), 0,
                               nullptr, src, (float *)dst, n);
            (void)hipEventRecord(e1, nullptr);
            e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess)
                e = hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && (best_read == 0.f || ms < best_read))
                best_read = ms;
        }
    }
    (void)hipFree(src);
    (void)hipFree(dst);
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    if (e != hipSuccess)
        return ek_fail(EK_EHIP, "ek_hbm_copy_rate: %s", hipGetErrorString(e));
    gbytes_per_s[0] = best > 0.f ? 2.0 * (double)(n * 16) / (best * 1e-3) / 1e9 : 0.0;
    gbytes_per_s[1] = best_read > 0.f ? (double)(n * 16) / (best_read * 1e-3) / 1e9 : 0.0;
    return EK_OK;
}
