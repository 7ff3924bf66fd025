// ek_ctx.h -- the context behind the C ABI and what the files that implement it
// share (ek_api.hip: context, loading, state, k-centers; ek_api_pam.hip: the PAM
// entry points; ek_api_ms.hip: rounds across shards).  Host side only.
#pragma once
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

#define EK_N_FORMS 5        // forms of a round in the run statistics: 1 / 4 / 8 / 16 / 32

// (thread-local message behind ek_last_error; defined in ek_api.hip)
int ek_set_error(int code, const char *fmt, ...);
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
    bool no_qtiles = false;      // there was no memory for it: rounds of 8 at most
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
    int assign_variant = 0;      // 0 auto, 1 vector FMA, 2 MFMA 32x32x2, 3 MFMA 16x16x4
    float *cen_blocks = nullptr; // the same centers in blocks of 16, candidate-tile layout
    size_t cen_blocks_cap = 0;   // in bytes

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
    float *pam_dprop = nullptr;      // [EK_PAM_WIN] the proposals' distances to their medoids
    int64_t *sel_host = nullptr;     // pinned [EK_PAM_WIN]: the selected frames on their way back
    int64_t *cnt_host = nullptr;     // pinned [EK_PAM_WIN]: the next window's member counts
    int64_t *js_host = nullptr;      // pinned [EK_PAM_WIN]: the drawn member ranks on their way in
    int pam_zero_copy = 1;           // kernels write the window's small results into mapped
                                     //   host memory themselves (option key 20)
    bool pf_members = false;         // the window's proposals are members of its clusters
                                     //   (drawn by ek_pam_sweep): tables as bounds
    int pam_bounds = 1;              // use that (option key 16)
    int pam_bounds_off = 0;          // windows to go with exact tables (the bounds were too loose)
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
    int pam_spec = 1;                // a window's slots evaluated at once, ahead of their turn (option key 19)
    unsigned long long *sp_marks = nullptr;  // [n] which slots' buckets a frame is in and which
                                     //   would change it (zero between windows)
    hipEvent_t win_ev = nullptr;     // a window's record is on the host
    int64_t sp_ahead = 0;            // slots whose evaluation ahead was taken over
    bool sp_bcnt_clean = false;      // the buckets' lengths are zero (the finish kernel's doing)
    int32_t sp_backoff = 0, sp_backoff_next = 8;    // windows to go the three-launch way after one ended early

    // multi-candidate rounds (ek_spec.hip)
    int cands = -1;              // candidates per pass: -1 auto, 1 = one-center passes
    unsigned char *recsT = nullptr;   // EK_MAX_CANDS records
    EkPlan *plan = nullptr;
    float *vecs = nullptr;       // [EK_MAX_CANDS-1][n_pad] stored distance vectors
    EkMaxHdr *hdr = nullptr;
    EkBlockMax *pm = nullptr;    // [EK_MAX_CANDS-1][nb] per-prefix maxima (ek_chain.hip)
    EkBlockMax *fm = nullptr;    // [4 nb] maxima per 64 frames of a round's last prefix
    int fine_pick = 1;           // the candidate pick reads them (option key 15)
    int pass_sweep = 1;          // rounds of 16: per-prefix maxima taken by the pass (EK_OPT_PASS_SWEEP)
    int ms_two_phase = 1;        // mailbox rounds: headers first, offers of the state the chain left (EK_OPT_MS_TWO_PHASE)
    int pick_cap = 0;            // far frames per label on the pick's list: 0 by the yield
                                 //   (4 <-> 16), else fixed 1 .. 16 (option key 17)
    unsigned char *top = nullptr;    // scratch of the candidate pick (ek_spec.hip)
    float *planD = nullptr;          // [64][64] distances between the records on offer
    int fused = 1;               // single-shard rounds in three launches (ek_round.hip)
    int tri = 0;                 // triangle-inequality tile skip (one-center steps)
    float *ti_D = nullptr;       // [ti_cap] distances of the existing centers to the new one
    int32_t ti_cap = 0;
    uint8_t *ti_skip = nullptr;  // [n_tiles]
    unsigned long long *ti_stats = nullptr;  // [2] tiles looked at, skipped
    int64_t ti_tiles = 0, ti_skipped = 0;    // of the last run
    float *ti_rtab = nullptr;    // rounds: [ti_rtab_cap][EK_MAX_CANDS] center-to-candidate distances
    int32_t ti_rtab_cap = 0;
    uint32_t *ti_tmask = nullptr;    // rounds: [n_tiles] candidates that can change the tile
    float *ti_tab = nullptr;     // sharded steps: the accepted centers, [ti_tab_cap][3A]
    double *ti_tabG = nullptr;
    int32_t ti_tab_cap = 0;
    int32_t ti_tab_n = 0;        // rows 0 .. ti_tab_n - 1 are the centers of labels 0 ..
    EkPend *pend = nullptr;      // accepted chain not yet applied
    EkChainOrd *ord = nullptr;
    EkChainRow *rows = nullptr;      // [EK_MAX_CANDS] candidate frames' rows
    uint32_t *vmask = nullptr;       // [n_pad / 64] which vectors a wave stored
    // EK_POISON=1: every working buffer is followed by a guard page of 0xA5 bytes
    // (ek_debug_guards: which ones something wrote past their end)
    struct Guard { const char *name; unsigned char *at; };
    std::vector<Guard> guards;
    unsigned int *tick = nullptr;    // [256] arrival counters: [0] pass, [1] chain,
                                     // [2] + [64..96) next, [3] legacy chain maxima,
                                     // [5] [6] multi-shard helpers, [128] + [129..161) PAM
    float *ctile = nullptr;      // the round's candidates, [atom][pair][xyz][2]
    double *ctrace = nullptr;    // their traces
    int chain = 1;               // 1: chained cheap steps, 0: one launch pair per center
    int64_t n_pad = 0;
    int32_t last_passes = 0;
    int adapt = 1;               // choose the candidates per pass from measured rates
    int64_t st_rounds[EK_N_FORMS] = {0, 0, 0, 0, 0};    // rounds run as 1 / 4 / 8 / 16 / 32 candidates (last run)
    int64_t st_centers[EK_N_FORMS] = {0, 0, 0, 0, 0};   // centers they accepted
    hipEvent_t evb0 = nullptr, evb1 = nullptr;   // per-batch timing

    // rounds across shards (ek_mshard.hip)
    EkMsState *ms = nullptr;         // device-side state
    unsigned char *ms_mbox = nullptr;    // own mailbox area [2][world][msg]
    uint32_t *ms_flags = nullptr;        // own flags [2][world][16]
    EkMsXchg ms_x;                   // transport (host copy, passed by value)
    int ms_peers = 0;                // peers connected (mailbox transport on at == world)
    std::vector<void *> ms_ipc;      // mappings opened with hipIpcOpenMemHandle
    int ms_T = 0;                    // candidates per pass of the run in progress
    int ms_small = 0;                // the group's shards are small: ek_ms_run's ladder moves
                                     //   to rounds of 16 earlier (option key 18)
    // what the last ek_ms_run spent where (ek_ms_diag): the first round of every
    // batch is bracketed by events -- pass | chain (with the exchange) | plan
    hipEvent_t ms_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double ms_t[3] = {0.0, 0.0, 0.0};    // milliseconds, summed over the sampled rounds
    int64_t ms_t_n = 0;              // rounds sampled
    EkMsState ms_last;               // the device-side counters when the run ended

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


// helpers defined in ek_api.hip
// Wait for the stream by polling.  The loops that read a few bytes back per
// step (PAM proposals, the k-centers progress checks) use this:
// hipStreamSynchronize may put the thread to sleep and a wake-up costs anything
// from 0.1 ms to tens of ms on a busy host -- more than the step itself.
hipError_t ek_wait(ek_ctx *c);
int ek_pick_fpl(const ek_ctx *c);
int ek_pick_nt(const ek_ctx *c);
int ek_pick_cands(const ek_ctx *c, bool wide = false, bool group = false);
int ek_ensure_qtiles(ek_ctx *c);
// nearest centers on v_mfma_f32_16x16x4_f32 (ek_assign.hip): the centers in blocks of
// 16 laid out like a pass's candidate tile, the frames from the quad copy
size_t ek_cblocks16_bytes(int32_t K, int A);
void ek_launch_cblocks16(const float *cen_aos, int32_t K, int A, float *cblocks,
                         hipStream_t s);
void ek_launch_assign16(const float *qtiles, const double *G, int64_t n, int A,
                        const float *cblocks, const double *Gc, int32_t K, float *dist,
                        int32_t *assign, hipStream_t s);
void ek_launch_pam_setup_dev(const float *aos, const double *G, int A,
                             const int64_t *frames_dev, int count, int64_t global_offset,
                             unsigned char *recs, float *ctile, double *ctrace,
                             EkPlan *plan, unsigned int *counter, hipStream_t s,
                             const float *dist, float *dprop);
// (as in ek_common.h, with mapped host memory that receives the same values -- no copy
// kernel behind the launch)
void ek_launch_count_members_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, int32_t *blockcnt, int64_t *scan,
                                   int64_t *total, hipStream_t s, int64_t *total_host);
void ek_launch_select_member_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, const int64_t *scan,
                                   const int64_t *js_dev, int64_t *out,
                                   hipStream_t s, int64_t *out_host);
extern int ek_pam_pairs_form;    // ek_pam.hip: the pairs kernels through the matrix cores (key 21)
int ek_form_slot(int T);
int ek_spec_alloc(ek_ctx *c);
bool ek_poison();
// uncached device memory, kept for reuse instead of freed (ek_api.hip)
hipError_t ek_uncached_alloc(int device, void **ptr, size_t bytes);
void ek_uncached_free(int device, void *ptr);
#define EK_GUARD_BYTES 4096
int ek_ensure_hist(ek_ctx *c, int32_t label);
void ek_pam_forget(ek_ctx *c);
int ek_upload_centers(ek_ctx *c, const float *xyz, int32_t K);
int ek_free_all(ek_ctx *c);

