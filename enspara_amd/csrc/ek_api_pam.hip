// ek_api_pam.hip -- the C ABI of include/enspara_hip.h: PAM (k-medoids) entry points.
#include "ek_ctx.h"

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
        // (coherent and mapped: the window's kernel writes the record there itself)
        EK_HIP(hipHostMalloc((void **)&c->pam_win_host, sizeof(EkPamWin),
                             hipHostMallocCoherent | hipHostMallocMapped));
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
        EK_HIP(hipMalloc((void **)&c->pam_dprop, EK_PAM_WIN * sizeof(float)));
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
                                   bool prepared = false, bool *waited = nullptr)
{
    if (waited)
        *waited = false;
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
        // (proposals drawn among their clusters' members, ek_pam_sweep: the
        // proposal-to-medoid table only as the lower bounds the old medoids'
        // table gives, half the pairs -- ek_pam_pairs_kernel<0>.  Where the bounds
        // are too loose -- few atoms, clusters as wide as they are apart: more than
        // a quarter of the frames "within reach" -- the window's tables are made
        // again exactly, and the next windows' right away)
        bool bounds = slots && prepared && c->pf_members && c->pam_bounds &&
                      c->pam_bounds_off == 0;
        if (c->pam_bounds_off > 0)
            --c->pam_bounds_off;
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
        for (int attempt = 0;; ++attempt) {
            ek_launch_pam_tables(c->med_aos, c->med_G, c->A, K, c->pam_restore,
                                 c->pam_recs, count, win_lo, slots ? count : 0, c->dtab,
                                 c->dtab + tb, c->dtab + 2 * tb, c->stream,
                                 bounds ? c->pam_dprop : nullptr);
            c->tab_lo = win_lo;
            c->tab_n = slots ? count : 0;
            if (attempt > 0)    // (the counter the set-up kernel cleared has been used)
                EK_HIP(hipMemsetAsync(c->amb_count + 3, 0, sizeof(unsigned int), c->stream));
            ek_launch_pam_active(c->dist, c->assign, c->n, c->dtab + 2 * tb, groups, K,
                                 win_lo, win_count, c->act_list, c->amb_count + 3,
                                 c->stream, prepared || attempt > 0);
            EK_CHECK_LAUNCH();
            EK_HIP(hipMemcpyAsync(c->act_n_host, c->amb_count + 3, sizeof(unsigned int),
                                  hipMemcpyDeviceToHost, c->stream));
            EK_HIP(ek_wait(c));
            if (waited)
                *waited = true;
            if (!bounds || (int64_t)*c->act_n_host * 4 <= c->n)
                break;
            bounds = false;
            c->pam_bounds_off = 32;
        }
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
                            c->stream, c->dist, c->pam_dprop);
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

// Round 5: the window's proposals without a wait of their own.  The selected frames
// stay where ek_select_member_multi_kernel leaves them -- the set-up kernel reads
// them there -- and travel to the host behind it; they have arrived when the
// prefetch's own wait (the length of the list of frames within reach) is over, or
// one is made here.  Then they are checked like ek_pam_select_members_batch and
// ek_pam_prefetch do.  One wait per window less (three were two thirds of the
// ~60 us a window spends outside its kernels).
static int ek_pam_select_prefetch(ek_ctx *c, int32_t cid0, int32_t count,
                                  const int64_t *js, int32_t win_count, int64_t *frames)
{
    if (c->bat_cid0 != cid0 || count < 1 || count > c->bat_count)
        return ek_fail(EK_ESTATE, "ek_pam_sweep: member counts of clusters [%d,+%d) "
                                  "are not the ones at hand", cid0, count);
    EK_HIP(hipSetDevice(c->device));
    int rc = ek_pam_vecs_alloc(c);
    if (rc)
        return rc;
    if (!c->sel_host)
        EK_HIP(hipHostMalloc((void **)&c->sel_host, EK_PAM_WIN * sizeof(int64_t),
                             hipHostMallocCoherent | hipHostMallocMapped));
    // (the drawn ranks are read by the selection from mapped host memory, the selected
    // frames written into it: no copy either way)
    int64_t *sel_dev_view = nullptr;
    const int64_t *js_dev_view = nullptr;
    if (c->pam_zero_copy) {
        if (!c->js_host)
            EK_HIP(hipHostMalloc((void **)&c->js_host, EK_PAM_WIN * sizeof(int64_t),
                                 hipHostMallocCoherent | hipHostMallocMapped));
        void *dp = nullptr, *dj = nullptr;
        if (hipHostGetDevicePointer(&dp, c->sel_host, 0) == hipSuccess &&
            hipHostGetDevicePointer(&dj, c->js_host, 0) == hipSuccess) {
            sel_dev_view = (int64_t *)dp;
            js_dev_view = (const int64_t *)dj;
            for (int32_t j = 0; j < count; ++j)
                c->js_host[j] = js[j];
        } else {
            (void)hipGetLastError();
        }
    }
    if (!js_dev_view)
        EK_HIP(hipMemcpyAsync(c->bat_sel + 2 * EK_PAM_WIN, js,
                              (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice,
                              c->stream));
    ek_launch_select_member_multi(c->assign, c->n, cid0, count, c->bat_scan,
                                  js_dev_view ? js_dev_view : c->bat_sel + 2 * EK_PAM_WIN,
                                  c->bat_sel + EK_PAM_WIN, c->stream, sel_dev_view);
    EK_CHECK_LAUNCH();
    if (!sel_dev_view)
        EK_HIP(hipMemcpyAsync(c->sel_host, c->bat_sel + EK_PAM_WIN,
                              (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost,
                              c->stream));
    c->bat_cid0 = -1;           // the scans describe the state at count time only
    c->bat_count = 0;
    c->pf_count = 0;
    ek_launch_pam_setup_dev(c->aos, c->G, c->A, c->bat_sel + EK_PAM_WIN, count, c->goff,
                            c->pam_recs, c->ctile, c->ctrace, c->pam_plan, c->amb_count + 3,
                            c->stream, c->dist, c->pam_dprop);
    bool waited = false;
    rc = ek_pam_prefetch_vectors(c, count, cid0, win_count, true, true, &waited);
    if (rc)
        return rc;
    if (!waited)
        EK_HIP(ek_wait(c));
    for (int32_t j = 0; j < count; ++j) {
        frames[j] = c->sel_host[j];
        if (frames[j] < 0 || frames[j] >= c->n)
            return ek_fail(EK_EARG, "ek_pam_sweep: cluster %d has no member %lld",
                           cid0 + j, (long long)js[j]);
        c->pf_frames[j] = frames[j];
    }
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
                                int32_t win_count, EkSpArgs *args)
{
    const size_t cap = EK_SP_CAP;
    const size_t o_bucket = 0;
    const size_t o_bcnt = o_bucket + (size_t)EK_PAM_WIN * cap * sizeof(uint2);
    const size_t o_spec = o_bcnt + 256;
    if (!c->sp_buf) {
        EK_HIP(hipMalloc((void **)&c->sp_buf, o_spec + ek_sp_spec_bytes()));
        c->sp_bcnt_clean = false;
    }
    if (c->pam_spec && !c->sp_marks) {
        EK_HIP(hipMalloc((void **)&c->sp_marks, (size_t)c->n * sizeof(unsigned long long)));
        EK_HIP(hipMemsetAsync(c->sp_marks, 0, (size_t)c->n * sizeof(unsigned long long),
                              c->stream));
    }
    const int K = c->med_K;
    const size_t tb = (size_t)EK_PAM_WIN * (c->med_cap + 1);
    // the state's cost tree, the slots' frames
    ek_launch_pw_tree(c->dist, c->assign, c->n, c->pw_shapes, c->pw_n_full,
                      c->pw_leaves, c->pw_chunks, c->sq_part, c->moved, c->stream);
    ek_launch_sp_bucket(c->act_list, c->sp_nact, c->dist, c->assign, c->pam_vecs, c->n_pad,
                        cid0, count, (uint2 *)(c->sp_buf + o_bucket),
                        (unsigned int *)(c->sp_buf + o_bcnt), (int64_t)cap, c->stream,
                        c->sp_bcnt_clean);
    c->sp_bcnt_clean = false;
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
    a.win_host = nullptr;
    if (c->pam_zero_copy) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->pam_win_host, 0) == hipSuccess)
            a.win_host = (EkPamWin *)dp;
        else
            (void)hipGetLastError();
    }
    // every slot evaluated at once on the state the window opens with (not while a
    // rejected proposal's row of the medoid table is still to be put back)
    a.use_spec = (c->pam_spec && c->pam_restore < 0 && count > 1) ? 1 : 0;
    a.spec = (EkSpSpecRec *)(c->sp_buf + o_spec);
    a.spec_lists = (uint32_t *)(c->sp_buf + o_spec + EK_PAM_WIN * sizeof(EkSpSpecRec));
    a.marks = c->sp_marks;
    a.finish_later = a.use_spec;
    // (the vectors' reset rides in the finish kernel where every column written is one
    // of the window's slots)
    const bool reset_there = a.finish_later && c->vecs_rows == c->sp_nact &&
                             c->vecs_cols == count;
    a.act_list = c->act_list;
    a.n_act = c->sp_nact;
    a.n_pad = c->n_pad;
    a.vecs = reset_there ? c->pam_vecs : nullptr;
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
        fprintf(stderr, "sp prof after %d windows (us per window): before the slots %.1f  "
                        "evaluation / take-over %.1f  apply %.1f  leaves %.1f  chunks %.1f  "
                        "total %.1f  verdict %.1f  end of slot %.1f  loop exit %.1f  after the slots %.1f\n",
                prof_n, prof_tot[8] * 1e-2 / prof_n, prof_tot[1] * 1e-2 / prof_n,
                prof_tot[2] * 1e-2 / prof_n, prof_tot[3] * 1e-2 / prof_n,
                prof_tot[4] * 1e-2 / prof_n, prof_tot[5] * 1e-2 / prof_n,
                prof_tot[6] * 1e-2 / prof_n, prof_tot[7] * 1e-2 / prof_n,
                prof_tot[0] * 1e-2 / prof_n, prof_tot[9] * 1e-2 / prof_n);
    }
#endif
    if (a.use_spec)
        ek_launch_sp_spec(a, c->stream);
    ek_launch_sp_window(a, c->stream);
    EK_CHECK_LAUNCH();
    *args = a;
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
    EkSpArgs sp_args = {};
    if (sparse) {
        rc = ek_pam_window_sparse(c, cid0, count, frames, n_members, win_count, &sp_args);
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
    // (the window's kernel and the scan write their results into mapped host memory
    // themselves where they can: a copy kernel each, in the chain the host waits for)
    if (!(sparse && sp_args.win_host))
        EK_HIP(hipMemcpyAsync(c->pam_win_host, c->pam_win_dev, sizeof(EkPamWin),
                              hipMemcpyDeviceToHost, c->stream));
    int64_t *cnt_dev_view = nullptr;
    if (next_count > 0) {
        if (c->pam_zero_copy) {
            if (!c->cnt_host)
                EK_HIP(hipHostMalloc((void **)&c->cnt_host, EK_PAM_WIN * sizeof(int64_t),
                                     hipHostMallocCoherent | hipHostMallocMapped));
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, c->cnt_host, 0) == hipSuccess)
                cnt_dev_view = (int64_t *)dp;
            else
                (void)hipGetLastError();
        }
        ek_launch_count_members_multi(c->assign, c->n, next_cid0, next_count,
                                      c->bat_blockcnt, c->bat_scan, c->bat_sel, c->stream,
                                      cnt_dev_view);
        EK_CHECK_LAUNCH();
        if (!cnt_dev_view)
            EK_HIP(hipMemcpyAsync(next_counts, c->bat_sel,
                                  (size_t)next_count * sizeof(int64_t),
                                  hipMemcpyDeviceToHost, c->stream));
    }
    if (sparse && sp_args.finish_later) {
        // the accepted proposals' rows of the medoid table and the slots' marks: behind
        // what the host waits for
        if (!c->win_ev)
            EK_HIP(hipEventCreateWithFlags(&c->win_ev, hipEventDisableTiming));
        EK_HIP(hipEventRecord(c->win_ev, c->stream));
        ek_launch_sp_finish(sp_args, c->stream);
        EK_CHECK_LAUNCH();
        c->sp_bcnt_clean = sp_args.count == EK_PAM_WIN;  // (all of the lengths)
        for (;;) {
            const hipError_t e = hipEventQuery(c->win_ev);
            if (e == hipSuccess)
                break;
            if (e != hipErrorNotReady)
                return ek_fail(EK_EHIP, "ek_pam_window_run: %s", hipGetErrorString(e));
        }
    } else {
        EK_HIP(ek_wait(c));
    }
    c->bat_cid0 = -1;
    c->bat_count = 0;
    if (cnt_dev_view)
        for (int32_t i = 0; i < next_count; ++i)
            next_counts[i] = c->cnt_host[i];
    const EkPamWin &w = *c->pam_win_host;
    if (sparse && sp_args.finish_later && sp_args.vecs && w.stop == count) {
        c->vecs_rows = 0;       // (ek_sp_finish_kernel has put the vectors back to +inf)
        c->pf_count = 0;        //  ... nothing of them is a prefetched vector any more
    }
    c->tab_n = 0;            // the medoid table has moved on
    c->pam_cid = -1;
    c->cnt_cid = -1;
    c->pf_hits -= count - w.stop;       // the slots past the stop were not served
    if (sparse) {
        c->sp_ahead += w.pad >> 8;      // slots taken over as evaluated ahead
        if (w.pad & 0xff) {
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
        } else {
            n_slots = cnt;
            for (int32_t s = 0; s < cnt; ++s)
                frames[s] = proposals[cid + s];
        }
        // (drawn proposals are members of their clusters: see ek_pam_prefetch_vectors)
        c->pf_members = !proposals;
        if (!proposals && n_slots > 0 && c->ctile)
            rc = ek_pam_select_prefetch(c, cid, n_slots, js, cnt, frames);
        else {
            if (!proposals && n_slots > 0)
                rc = ek_pam_select_members_batch(c, cid, n_slots, js, frames);
            if (!rc)
                rc = ek_pam_prefetch_window(c, frames, n_slots, cid, cnt);
        }
        c->pf_members = false;
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

extern "C" int ek_pam_ahead_stats(ek_ctx *c, int64_t *slots_taken_over)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (slots_taken_over)
        *slots_taken_over = c->sp_ahead;
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

