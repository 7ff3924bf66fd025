// ek_api_ms.hip -- the C ABI of include/enspara_hip.h: multi-candidate rounds across
// shards (ek_spec_*: three exchanges per round; ek_ms_*: one).
#include "ek_ctx.h"

// ---- multi-candidate rounds across shards --------------------------------------------
extern "C" int ek_spec_candidates(ek_ctx *c)
{
    return c ? ek_pick_cands(c, false, true) : 0;
}

extern "C" int ek_round_candidates(ek_ctx *c)
{
    return c ? ek_pick_cands(c, true) : 0;
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
    const int T = std::max(ek_pick_cands(c, false, true), 1);
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
    const int T = ek_pick_cands(c, false, true);
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
    const int T = std::max(ek_pick_cands(c, false, true), 1);
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
    for (int m = 0; m < EK_N_FORMS; ++m) {
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
// records a shard offers per exchange: EK_MS_SLOTS over all shards, at most the 64 its
// pick lists (round 6: twice round 5's -- the plan kernel lets the 64 farthest compete)
// Rounds of 8 between two looks at the ladder.  Rounds 3-5: 24 exchanges, of which the
// early ones were a third exchanges without a pass -- 17 to 19 passes.  With the headers
// first (round 6) an exchange is a pass, and 24 of them kept the million-frame fit in
// rounds of 8 for 74 passes instead of 56 (0.338 s against 0.321); 12: 47 passes, 0.318 s
// (16: 0.340; the 125 000-frame shard does not care: 12.85 / 13.0 / 12.9 us per center).
// -DEK_MS_BATCH8=.. builds the others.
#ifndef EK_MS_BATCH8
#define EK_MS_BATCH8 12
#endif
static int ek_ms_offer(int world)
{
    return std::max(1, std::min(64, EK_MS_SLOTS / std::max(world, 1)));
}

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
    R.fm = c->fine_pick ? c->fm : nullptr;
    R.sweep = (c->pass_sweep == 2 ||
               (c->pass_sweep == 1 && c->n <= (int64_t)2048 * EK_TILE)) ? 1 : 0;
    R.top = c->top;
    R.ctile = c->ctile;
    R.ctrace = c->ctrace;
    R.hist = c->hist;
    R.ctl = c->ctl;
    R.tick = c->tick;
    R.rows = c->rows;
    R.vmask = c->vmask;
    R.cutoff = cutoff;
    R.pick_cap = c->pick_cap > 0 ? c->pick_cap : 4;
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
    ek_uncached_free(c->device, c->ms_mbox);
    ek_uncached_free(c->device, c->ms_flags);
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
    const size_t guard = ek_poison() ? EK_GUARD_BYTES : 0;      // (EK_POISON: ek_debug_guards)
    EK_HIP(ek_uncached_alloc(c->device, (void **)&c->ms_mbox, mb + guard));
    EK_HIP(ek_uncached_alloc(c->device, (void **)&c->ms_flags, fb + guard));
    EK_HIP(hipMemsetAsync(c->ms_mbox, 0, mb, c->stream));
    EK_HIP(hipMemsetAsync(c->ms_flags, 0, fb, c->stream));
    if (guard) {
        EK_HIP(hipMemsetAsync((unsigned char *)c->ms_mbox + mb, 0xA5, guard, c->stream));
        EK_HIP(hipMemsetAsync((unsigned char *)c->ms_flags + fb, 0xA5, guard, c->stream));
        for (size_t k = 0; k < c->guards.size();)       // (a second ek_ms_setup)
            if (!strcmp(c->guards[k].name, "mailbox") || !strcmp(c->guards[k].name, "flags"))
                c->guards.erase(c->guards.begin() + k);
            else
                ++k;
        c->guards.push_back({"mailbox", (unsigned char *)c->ms_mbox + mb});
        c->guards.push_back({"flags", (unsigned char *)c->ms_flags + fb});
    }
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

// Everything a run of the rounds to `n_centers` centers allocates, now: the history
// of the accepted centers, the rounds' working set, the quad copy of the frames for
// rounds of 16.  hipMalloc / hipFree wait for the whole device: inside ek_ms_run --
// peers already polling for this shard's message on the same GPU (contexts of one
// process: tools/c4_one_gpu.py, the tests) or a collective in flight -- that is a
// deadlock until the mailbox time-out (round 6: eight shards of one process, 3000
// centers: the history grows beyond its first 1024 entries at the start of the run).
extern "C" int ek_reserve_centers(ek_ctx *c, int32_t n_centers)
{
    if (!c || n_centers < 0)
        return ek_fail(EK_EARG, "ek_reserve_centers: bad argument");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_reserve_centers: no frames loaded");
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_ensure_hist(c, n_centers);
    if (rc)
        return rc;
    rc = ek_spec_alloc(c);
    if (rc)
        return rc;
    if (ek_pick_cands(c, true, true) >= 16) {
        const int eq = ek_ensure_qtiles(c);
        if (eq != EK_OK && eq != EK_ENOMEM)     // (no room: the run itself reports it)
            return eq;
    }
    for (int k = 0; k < 4; ++k)
        if (!c->ms_ev[k])
            EK_HIP(hipEventCreate(&c->ms_ev[k]));
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
    // (the caller's loop has no ladder: rounds of 32 only where option key 4 asks)
    const int T = ek_pick_cands(c, c->cands == 32, true);
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
    if (T >= 16) {
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
static int ek_ms_enqueue_local(ek_ctx *c, const EkRound &R, const EkMsXchg &x,
                               hipEvent_t *ev = nullptr)
{
    const bool sample = c->samp_every > 0 &&
                        (c->samp_count++ % c->samp_every) == 0 &&
                        2 * (size_t)c->samp_used + 1 < c->samp_ev.size();
    if (sample) {
        c->samp_form[c->samp_used] = R.T;
        EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used], c->stream));
    }
    if (ev)
        EK_HIP(hipEventRecord(ev[0], c->stream));
    ek_launch_round_pass(R, c->stream, false);
    if (sample) {
        EK_HIP(hipEventRecord(c->samp_ev[2 * c->samp_used + 1], c->stream));
        c->samp_used++;
    }
    if (ev)
        EK_HIP(hipEventRecord(ev[1], c->stream));
    ek_launch_ms_chain(R, c->ms, x, c->stream);
    if (ev)
        EK_HIP(hipEventRecord(ev[2], c->stream));
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
    // goes back to 8 and the next try waits twice as many batches.  On SMALL shards
    // (option key 18 = 1, set by sharded.kcenters_sharded for every rank alike when
    // the largest shard has under 300 000 frames) 4.5 and 5.5: with the exchange's
    // tails a round of 16 costs only 1.3 x a round of 8 on a 125 000-frame shard and
    // the ladder lost 7 % there to staying narrow too long -- while at 10^6 frames
    // per shard the lower thresholds cost 8 % (a third of the early rounds of 16
    // break and are offered again, 0.2 ms each).  A change of
    // form costs one exchange without a pass (the state's farthest frames are
    // offered again).  Results do not depend on the form.
    // Round 5: rounds of 32 -- two passes of 16 behind ONE plan, chain and
    // exchange -- once rounds of 16 are usually accepted whole (>= 14 per round);
    // back to 16 when a batch of them accepts fewer than 22 per round (what 16
    // would accept at most, with half the passes), the next try twice as far off.
    const int Tmax = ek_pick_cands(c, true, true);
    if (Tmax < 4)
        return ek_fail(EK_ESTATE, "ek_ms_run: multi-candidate rounds are off "
                                  "(option key 4 = 1: use ek_kcenters_step)");
    const bool ladder = Tmax >= 16 && c->cands == -1 && c->adapt;
    const double up16 = c->ms_small ? 4.5 : 6.5, down16 = c->ms_small ? 5.5 : 8.5;
    int T = ladder ? 8 : Tmax;
    rc = ek_ms_begin_T(c, first_label, first_label + max_new, T);
    if (rc)
        return rc;
    EkRound R;
    ek_round_of(c, T, dist_cutoff, R);
    EkMsXchg x = c->ms_x;
    x.sys = 1;
    x.two_phase = c->ms_two_phase ? 1 : 0;
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
    int wait16 = 0, next_wait = 1, wait32 = 0, next_wait32 = 1;
    // (far frames per label on the pick's list, 4 <-> 16 by the yield: as in
    // ek_run_rounds, from numbers every shard sees alike)
    int cap = c->pick_cap > 0 ? c->pick_cap : 4;
    const bool cap_adaptive = c->pick_cap <= 0;
    bool cap_probing = false;
    int cap_wait = 0, cap_next_wait = 2, cap_form = 0;
    double cap_yield_home = 0.0;
    R.pick_cap = cap;
    for (int k = 0; k < EK_N_FORMS; ++k)
        c->st_rounds[k] = c->st_centers[k] = 0;
    // (the run's own counters: ek_ms_diag)
    EK_HIP(hipMemsetAsync(&c->ms->n_reoffer, 0,
                          sizeof(EkMsState) - offsetof(EkMsState, n_reoffer), c->stream));
    for (int k = 0; k < 4; ++k)
        if (!c->ms_ev[k])
            EK_HIP(hipEventCreate(&c->ms_ev[k]));
    c->ms_t[0] = c->ms_t[1] = c->ms_t[2] = 0.0;
    c->ms_t_n = 0;
    while (max_new > 0) {
        const int32_t left = goal - cr.n_done;
        int32_t batch = std::max(2, std::min(256, (int32_t)(left / per_round) + 2));
        if (ladder)
            batch = std::min(batch, T == 8 ? EK_MS_BATCH8 : (T == 16 ? 64 : 48));

        for (int32_t r = 0; r < batch; ++r) {
            rc = ek_ms_enqueue_local(c, R, x, r == 0 ? c->ms_ev : nullptr);
            if (rc)
                return rc;
            ek_launch_ms_plan(R, c->ms, x, c->planD, c->stream);
            if (r == 0)
                EK_HIP(hipEventRecord(c->ms_ev[3], c->stream));
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
        if (ran > 0) {          // the batch's first round really ran: where its time went
            for (int k = 0; k < 3; ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, c->ms_ev[k], c->ms_ev[k + 1]) == hipSuccess)
                    c->ms_t[k] += ms;
            }
            ++c->ms_t_n;
        }
        c->ms_last = st;
        passes += (T == 32 ? 2 : 1) * ran;      // (a round of 32 streams the frames twice)
        c->st_rounds[ek_form_slot(T)] += ran;
        c->st_centers[ek_form_slot(T)] += cr.n_done - before;
        if (st.err || st.mode == 0 || cr.stopped || cr.n_done >= goal)
            break;
        per_round = std::max(1.0, (double)(cr.n_done - before) / std::max(ran, 1));
        if (cap_adaptive && ran > 0) {
            const double yield = (double)(cr.n_done - before) / ((double)ran * T);
            if (cap_probing) {
                cap_probing = false;
                if (T == cap_form && yield > cap_yield_home + 0.1) {
                    cap_next_wait = 2;
                } else {
                    cap = cap == 4 ? 16 : 4;
                    cap_wait = cap_next_wait;
                    cap_next_wait = std::min(2 * cap_next_wait, 64);
                }
            } else if (cap_wait > 0) {
                --cap_wait;
            } else if (yield < 0.65 && goal - cr.n_done > 4 * T) {
                cap_yield_home = yield;
                cap_form = T;
                cap = cap == 4 ? 16 : 4;
                cap_probing = true;
            }
            R.pick_cap = cap;
        }
        if (ladder && ran > 0) {
            int want = T;
            if (T == 8) {
                if (wait16 > 0)
                    --wait16;
                else if (per_round >= up16)
                    want = 16;
            } else if (T == 16) {
                if (per_round < down16) {
                    want = 8;
                    wait16 = next_wait;
                    next_wait = std::min(2 * next_wait, 64);
                } else {
                    next_wait = 1;
                    if (wait32 > 0)
                        --wait32;
                    else if (Tmax == 32 && per_round >= 14.0)
                        want = 32;
                }
            } else if (per_round < 22.0) {
                want = 16;
                wait32 = next_wait32;
                next_wait32 = std::min(2 * next_wait32, 64);
            } else {
                next_wait32 = 1;
            }
            if (want != T) {
                ek_launch_round_flush(R, c->stream);    // the chain still pending
                EK_CHECK_LAUNCH();
                rc = ek_ms_begin_T(c, cr.n_done, goal, want);
                if (rc)
                    return rc;
                T = want;
                ek_round_of(c, T, dist_cutoff, R);
                R.pick_cap = cap;
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


// What the last ek_ms_run spent where, for a run that has to explain itself
// (bench.py reports it per rank).  counts: [0] exchanges, [1] of them without a
// pass (a chain that broke: its state offered again), [2] 10 ns ticks the shard
// waited for its peers' messages (per exchange the longest wait, summed), [3] ...
// for its OWN flag (a store that goes nowhere: the floor of [2]), [4] rounds
// sampled for `ms`; ms: mean milliseconds of a sampled round's pass, chain kernel
// (the exchange's wait is inside it) and plan kernel(s).
extern "C" int ek_ms_diag(ek_ctx *c, int64_t *counts, double *ms)
{
    if (!c || !counts || !ms)
        return ek_fail(EK_EARG, "ek_ms_diag: NULL argument");
    counts[0] = (int64_t)c->ms_last.seq;
    counts[1] = (int64_t)c->ms_last.n_reoffer;
    counts[2] = (int64_t)c->ms_last.wait_ticks_max;
    counts[3] = (int64_t)c->ms_last.wait_ticks_own;
    counts[4] = c->ms_t_n;
    for (int k = 0; k < 3; ++k)
        ms[k] = c->ms_t_n > 0 ? c->ms_t[k] / (double)c->ms_t_n : 0.0;
    return EK_OK;
}
