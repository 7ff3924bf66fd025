// ek_api.hip -- the C ABI of include/enspara_hip.h (host side).
#include "ek_ctx.h"
#include <mutex>
#include "ek_qcp.h"

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
// (what ek_msm.hip may see of a context)
void ek_ctx_msm_view(ek_ctx *c, int *device, int64_t *n, const int32_t **assign,
                     hipStream_t *stream, void ***scratch_slot)
{
    *device = c->device;
    *n = c->loaded ? c->n : -1;
    *assign = c->assign;
    *stream = c->stream;
    *scratch_slot = &c->msm_scratch;
}

hipError_t ek_wait(ek_ctx *c)
{
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e != hipErrorNotReady)
            return e;
    }
}

int ek_pick_fpl(const ek_ctx *c)
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
int ek_pick_nt(const ek_ctx *c)
{
    if (c->nt >= 0)
        return c->nt;
    const size_t bytes = (size_t)c->n_tiles * 3 * (size_t)c->A * EK_TILE * 4;
    return bytes > ((size_t)192 << 20) ? 1 : 0;
}

// the widest form of a round this context may use (candidates per pass):
// EK_MAX_CANDS unless option key 4 pins it; 1 = one-center passes only
// (wide: the fused single-shard rounds and the one-exchange rounds across shards
// know rounds of 32; the one-launch-per-step forms stop at EK_LEGACY_CANDS)
// (group: a shard of a group never narrows its rounds on its own -- its peers
// would plan sixteen candidates where it plans eight; without room for the quad
// copy the multi-shard entry points fail with EK_ENOMEM instead, and
// sharded.kcenters_sharded agrees on the form before anything runs: every rank
// asks ek_quad_copy_ready, one that says no pins key 4 to 8 on all)
int ek_pick_cands(const ek_ctx *c, bool wide, bool group)
{
    // (automatic: up to 16.  Rounds of 32 are built, tested and measured -- a round
    // of 32 accepts 18.8 of its guesses at 10^6 x 300 where one of 16 accepts 15.2,
    // 20 / 32 against 13 / 16 on 125 000-frame shards: 0.090 against 0.062 ms per
    // center, DESIGN.md 4a -- and lose everywhere measured; key 4 = 32 asks for them)
    int t = c->cands == -1 ? EK_LEGACY_CANDS : c->cands;
    if (t == 32 && !wide)
        t = EK_LEGACY_CANDS;
    if (t >= 16 && c->no_qtiles && !group)  // (ek_ensure_qtiles found no room)
        t = 8;
    return (t == 32 || t == 16 || t == 8 || t == 4) ? t : 1;
}

// The quad copy of the frames (ek_pass16.hip) is made when a 16-candidate pass
// first needs it and again after frames were loaded: a third copy of the
// coordinates (12 A bytes per frame, 3.6 GB at 10^6 x 300 of the 288 GB), one
// read and one write of the shard.
int ek_ensure_qtiles(ek_ctx *c)
{
    if (c->qt_valid)
        return EK_OK;
    if (!c->qtiles) {
        hipError_t e = hipMalloc(
            (void **)&c->qtiles, ek_quad_tiles_bytes(std::max<int64_t>(c->n_tiles, 1), c->A));
        if (e == hipErrorOutOfMemory && c->stage_frames > 0) {
            // the upload's staging buffers (two pinned, two on the device, up to
            // 256 MiB each) are only needed while frames are loaded: give them
            // back and try once more (the next load allocates them again)
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->stream);
            for (int b = 0; b < 2; ++b) {
                (void)hipFree(c->stage[b]);
                (void)hipHostFree(c->pin[b]);
                c->stage[b] = c->pin[b] = nullptr;
                c->up_busy[b] = false;
            }
            c->stage_frames = 0;
            e = hipMalloc((void **)&c->qtiles,
                          ek_quad_tiles_bytes(std::max<int64_t>(c->n_tiles, 1), c->A));
        }
        if (e == hipErrorOutOfMemory) {
            // no room for a third copy of the coordinates: a single shard then
            // runs rounds of 8 (ek_pick_cands), which stream the frame-minor tiles
            (void)hipGetLastError();
            c->qtiles = nullptr;
            c->no_qtiles = true;
            return ek_fail(EK_ENOMEM, "no memory for the quad copy of the frames "
                                      "(16-candidate rounds)");
        }
        EK_HIP(e);
    }
    ek_launch_quad_tiles(c->tiles, c->n_tiles, c->A, c->qtiles, c->stream);
    EK_CHECK_LAUNCH();
    c->qt_valid = true;
    return EK_OK;
}

extern "C" int ek_quad_copy_ready(ek_ctx *c)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    if (!c->loaded)
        return ek_fail(EK_ESTATE, "ek_quad_copy_ready: no frames loaded");
    EK_HIP(hipSetDevice(c->device));
    c->no_qtiles = false;           // (memory may have been given back since)
    const int eq = ek_ensure_qtiles(c);
    if (eq == EK_ENOMEM)
        return 0;
    return eq == EK_OK ? 1 : eq;
}

// slot of a form in the run statistics: rounds run as 1 / 4 / 8 / 16 / 32 candidates
int ek_form_slot(int T)
{
    return T <= 1 ? 0 : (T == 4 ? 1 : (T == 8 ? 2 : (T == 16 ? 3 : 4)));
}

// Uncached device memory (the mailboxes of ek_mshard.hip) is never given back to the
// runtime: a block that a context is done with waits here for the next context that asks
// for one of its size.  Round 6, tools/fuzz_ms.py: hipFree of memory that came from
// hipExtMallocWithFlags(hipDeviceMallocUncached), followed by ordinary allocations, left a
// process in which LATER allocations overlapped live ones -- a context's upload landed in
// another context's frames (256 of 768 frames of shard 0 changed when shard 1 loaded its
// own), runs accepted wrong centers or faulted, the more often the larger the blocks
// (three or more shards, 300 - 500 atoms: 7 of 60 runs; with the blocks kept: 0).  A process
// that sets its mailboxes up once (one per GPU: the product's case) never frees one.
namespace {
struct EkUcBlock { int device; void *ptr; size_t bytes; };
std::mutex g_uc_mutex;
std::vector<EkUcBlock> g_uc_free, g_uc_live;
}
hipError_t ek_uncached_alloc(int device, void **ptr, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_uc_mutex);
    size_t best = (size_t)-1;
    for (size_t k = 0; k < g_uc_free.size(); ++k)
        if (g_uc_free[k].device == device && g_uc_free[k].bytes >= bytes &&
            (best == (size_t)-1 || g_uc_free[k].bytes < g_uc_free[best].bytes))
            best = k;
    if (best != (size_t)-1) {
        *ptr = g_uc_free[best].ptr;
        g_uc_live.push_back(g_uc_free[best]);
        g_uc_free.erase(g_uc_free.begin() + best);
        return hipSuccess;
    }
    // (whole 64 KB: fewer sizes, more reuse)
    const size_t rounded = (bytes + 65535) & ~(size_t)65535;
    const hipError_t e = hipExtMallocWithFlags(ptr, rounded, hipDeviceMallocUncached);
    if (e == hipSuccess)
        g_uc_live.push_back({device, *ptr, rounded});
    return e;
}
void ek_uncached_free(int device, void *ptr)
{
    if (!ptr)
        return;
    std::lock_guard<std::mutex> lock(g_uc_mutex);
    for (size_t k = 0; k < g_uc_live.size(); ++k)
        if (g_uc_live[k].ptr == ptr && g_uc_live[k].device == device) {
            g_uc_free.push_back(g_uc_live[k]);
            g_uc_live.erase(g_uc_live.begin() + k);
            return;
        }
}

// EK_POISON=1 in the environment (tools/fuzz_*.py): working buffers start as 0x5a bytes
// (a huge distance, a frame number nobody has) instead of whatever the allocator hands out -- a fresh process
// gets zeroed pages, which hide a read of something nobody wrote yet; the hundredth
// context of a process gets the ninety-ninth's remains.
bool ek_poison()
{
    static const bool on = getenv("EK_POISON") != nullptr;
    return on;
}
// hipMalloc, or (EK_POISON) hipMalloc with the contents poisoned and a guard page behind
hipError_t ek_malloc_named(ek_ctx *c, void **ptr, size_t bytes, const char *name)
{
    if (!ek_poison())
        return hipMalloc(ptr, bytes);
    hipError_t e = hipMalloc(ptr, bytes + EK_GUARD_BYTES);
    // (not 0xff: an index of -1 and a NaN distance are what the kernels treat as "nothing
    // there"; 0x5a5a5a5a is a frame number no shard has and 1.5e16 a distance that wins)
    static const int fill = getenv("EK_POISON_BYTE") ? (int)strtol(getenv("EK_POISON_BYTE"), nullptr, 0)
                                                     : 0x5a;
    if (e == hipSuccess)
        e = hipMemset(*ptr, fill, bytes);
    if (e == hipSuccess)
        e = hipMemset((unsigned char *)*ptr + bytes, 0xA5, EK_GUARD_BYTES);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        c->guards.push_back({name, (unsigned char *)*ptr + bytes});
    return e;
}
extern "C" int ek_debug_guards(ek_ctx *c)
{
    if (!c)
        return ek_fail(EK_EARG, "NULL context");
    EK_HIP(hipSetDevice(c->device));
    EK_HIP(hipDeviceSynchronize());
    int bad = 0;
    std::vector<unsigned char> h(EK_GUARD_BYTES);
    for (const ek_ctx::Guard &g : c->guards) {
        // (a buffer that was grown since -- the history -- is gone with its guard)
        if (hipMemcpy(h.data(), g.at, EK_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        for (size_t k = 0; k < h.size(); ++k)
            if (h[k] != 0xA5) {
                fprintf(stderr, "enspara_hip: buffer %s overrun: byte + %zu behind its end "
                                "is 0x%02x\n", g.name, k, h[k]);
                ++bad;
                break;
            }
    }
    return bad;
}

int ek_spec_alloc(ek_ctx *c)
{
    if (!c->top)
        EK_HIP(ek_malloc_named(c, (void **)&c->top, ek_top_scratch_bytes(c->A), "top"));
    if (!c->planD)
        EK_HIP(ek_malloc_named(c, (void **)&c->planD, 64 * 64 * sizeof(float), "planD"));
    if (!c->pm) {
        const size_t nb = ((size_t)std::max<int64_t>(c->n, 1) + EK_BLOCK - 1) /
                          EK_BLOCK;
        EK_HIP(ek_malloc_named(c, (void **)&c->pm,
                               (size_t)(EK_MAX_CANDS - 1) * nb * sizeof(EkBlockMax), "pm"));
        EK_HIP(ek_malloc_named(c, (void **)&c->fm, 4 * nb * sizeof(EkBlockMax), "fm"));
    }
    if (!c->vecs) {
        const size_t bytes = (size_t)(EK_MAX_CANDS - 1) * std::max<int64_t>(c->n_pad, 1) *
                             sizeof(float);
        EK_HIP(ek_malloc_named(c, (void **)&c->vecs, bytes, "vecs"));
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

int ek_free_all(ek_ctx *c)
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
    (void)hipFree(c->cen_blocks);
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
    ek_uncached_free(c->device, c->ms_mbox);
    ek_uncached_free(c->device, c->ms_flags);
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
    (void)hipFree(c->ti_rtab);
    (void)hipFree(c->ti_tmask);
    (void)hipFree(c->pam_dprop);
    (void)hipHostFree(c->sel_host);
    (void)hipHostFree(c->cnt_host);
    (void)hipHostFree(c->js_host);
    (void)hipFree(c->sp_marks);
    if (c->win_ev)
        (void)hipEventDestroy(c->win_ev);
    (void)hipFree(c->fm);
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
        e = ek_malloc_named(c, (void **)&(ptr), (bytes), #ptr);
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
#ifdef EK_P16_STATS
    EK_ALLOC(c->tick, (256 + 64 * 16 + 2048) * sizeof(unsigned int));
    (void)hipMemset(c->tick, 0, (256 + 64 * 16 + 2048) * sizeof(unsigned int));
#else
    EK_ALLOC(c->tick, 256 * sizeof(unsigned int));
#endif
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
    if (e == hipSuccess)    // (n_rec = 0: no record chosen yet, ek_round_ctile16_kernel)
        e = hipMemsetAsync(c->plan, 0, sizeof(EkPlan), c->stream);
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
    case EK_OPT_NONTEMPORAL:
        c->nt = value < 0 ? -1 : (value ? 1 : 0);
        return EK_OK;
    case EK_OPT_CANDIDATES:
        if (value != -1 && value != 1 && value != 4 && value != 8 && value != 16 &&
            value != 32)
            return ek_fail(EK_EARG, "ek_set_option: candidates per round must "
                                    "be -1 (auto), 1, 4, 8, 16 or 32");
        c->cands = value;
        return EK_OK;
    case EK_OPT_CHAINED:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: chained rounds 0 or 1");
        c->chain = value;
        return EK_OK;
    case EK_OPT_PAM_PRUNE:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: PAM medoid pruning 0 or 1");
        c->prune = value;
        return EK_OK;
    case EK_OPT_STATE_EXACT:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: state-is-exact flag 0 or 1");
        c->state_exact = value != 0;
        return EK_OK;
    case EK_OPT_TRIANGLE:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: triangle inequality 0 or 1");
        c->tri = value;
        return EK_OK;
    case EK_OPT_FUSED_ROUNDS:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: fused rounds 0 or 1");
        c->fused = value;
        return EK_OK;
    case 9:     // (round 1's LDS form of the pass kernel is retired)
        if (value != 1)
            return ek_fail(EK_EARG, "ek_set_option: pass kernel form must be 1");
        return EK_OK;
    case EK_OPT_ADAPTIVE:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: adaptive candidates 0 or 1");
        c->adapt = value;
        return EK_OK;
    case EK_OPT_PAM_ONE_WORKGROUP:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: one-workgroup PAM windows 0 or 1");
        c->pam_sparse = value;
        return EK_OK;
    case EK_OPT_PAM_BOTH_SUMS:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: both cost sums for every proposal 0 or 1");
        c->sp_exact = value;
        return EK_OK;
    case EK_OPT_PAM_MAX_PAIRS:
        if (value < 0)
            return ek_fail(EK_EARG, "ek_set_option: pairs one workgroup searches >= 0");
        c->sp_max_pairs = value;
        return EK_OK;
    case EK_OPT_PAM_PAIRS_MFMA:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: PAM pairs kernels on the matrix cores 0 or 1");
        ek_pam_pairs_form = value;      // (process-wide: a measurement switch)
        return EK_OK;
    case EK_OPT_PAM_ZERO_COPY:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: PAM results into mapped host memory 0 or 1");
        c->pam_zero_copy = value;
        return EK_OK;
    case EK_OPT_PAM_AHEAD:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: PAM slots evaluated ahead 0 or 1");
        c->pam_spec = value;
        return EK_OK;
    case EK_OPT_SMALL_SHARDS:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: small shards 0 or 1");
        c->ms_small = value;
        return EK_OK;
    case EK_OPT_PICK_CAP:
        if (value < 0 || value > 16)
            return ek_fail(EK_EARG, "ek_set_option: far frames per label on the pick's list "
                                    "0 (by the yield) or 1 .. 16");
        c->pick_cap = value;
        return EK_OK;
    case EK_OPT_PAM_BOUNDS:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: PAM tables as bounds 0 or 1");
        c->pam_bounds = value;
        return EK_OK;
    case EK_OPT_FINE_PICK:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: maxima per 64 frames for the pick 0 or 1");
        c->fine_pick = value;
        return EK_OK;
    case EK_OPT_MS_TWO_PHASE:
        if (value != 0 && value != 1)
            return ek_fail(EK_EARG, "ek_set_option: headers first in the mailbox rounds 0 or 1");
        c->ms_two_phase = value;
        return EK_OK;
    case EK_OPT_PASS_SWEEP:
        if (value < 0 || value > 2)
            return ek_fail(EK_EARG, "ek_set_option: per-prefix maxima in the pass 0, 1 "
                                    "(by the shard's size) or 2 (always)");
        c->pass_sweep = value;
        return EK_OK;
    case EK_OPT_ASSIGN_KERNEL:
        if (value < 0 || value > 3)
            return ek_fail(EK_EARG, "ek_set_option: assign variant 0..3");
        c->assign_variant = value;
        return EK_OK;
    default:
        return ek_fail(EK_EARG, "ek_set_option: unknown key %d", key);
    }
}

extern "C" int ek_get_option(ek_ctx *c, int32_t key, int32_t *value)
{
    if (!c || !value)
        return ek_fail(EK_EARG, "ek_get_option: NULL argument");
    switch (key) {
    case EK_OPT_NONTEMPORAL: *value = c->nt; return EK_OK;
    case EK_OPT_ASSIGN_KERNEL: *value = c->assign_variant; return EK_OK;
    case EK_OPT_CANDIDATES: *value = c->cands; return EK_OK;
    case EK_OPT_CHAINED: *value = c->chain; return EK_OK;
    case EK_OPT_PAM_PRUNE: *value = c->prune; return EK_OK;
    case EK_OPT_STATE_EXACT: *value = c->state_exact ? 1 : 0; return EK_OK;
    case EK_OPT_ADAPTIVE: *value = c->adapt; return EK_OK;
    case EK_OPT_PASS_FORM: *value = 1; return EK_OK;
    case EK_OPT_FUSED_ROUNDS: *value = c->fused; return EK_OK;
    case EK_OPT_TRIANGLE: *value = c->tri; return EK_OK;
    case EK_OPT_PAM_ONE_WORKGROUP: *value = c->pam_sparse; return EK_OK;
    case EK_OPT_PAM_MAX_PAIRS: *value = c->sp_max_pairs; return EK_OK;
    case EK_OPT_PAM_BOTH_SUMS: *value = c->sp_exact; return EK_OK;
    case EK_OPT_FINE_PICK: *value = c->fine_pick; return EK_OK;
    case EK_OPT_PASS_SWEEP: *value = c->pass_sweep; return EK_OK;
    case EK_OPT_MS_TWO_PHASE: *value = c->ms_two_phase; return EK_OK;
    case EK_OPT_PAM_BOUNDS: *value = c->pam_bounds; return EK_OK;
    case EK_OPT_PICK_CAP: *value = c->pick_cap; return EK_OK;
    case EK_OPT_SMALL_SHARDS: *value = c->ms_small; return EK_OK;
    case EK_OPT_PAM_AHEAD: *value = c->pam_spec; return EK_OK;
    case EK_OPT_PAM_ZERO_COPY: *value = c->pam_zero_copy; return EK_OK;
    case EK_OPT_PAM_PAIRS_MFMA: *value = ek_pam_pairs_form; return EK_OK;
    default:
        return ek_fail(EK_EARG, "ek_get_option: unknown key %d", key);
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
        // of <= 256 MiB (whole tiles; EK_UPLOAD_CHUNK_MB: every chunk costs ~0.75 ms
        // beside its bytes at ~53 GB/s, measured: 64 MiB chunks 34 GB/s, 128 MiB 42,
        // 256 MiB 45) copied by a few host threads
        // (EK_UPLOAD_THREADS, default 8) into one of two
        // PINNED buffers, a DMA from there into one of two device staging buffers
        // and the layout kernel behind it on the context's stream -- while the
        // threads fill the other buffer.  The caller's array has been read
        // completely when this returns; the stream may still be working.
        size_t chunk_mb = 256;
        if (const char *e = getenv("EK_UPLOAD_CHUNK_MB"))
            chunk_mb = (size_t)std::max(1, std::min(atoi(e), 1024));
        int64_t chunk = (int64_t)((chunk_mb << 20) / (frame_floats * sizeof(float)));
        chunk = std::max<int64_t>(EK_TILE, chunk / EK_TILE * EK_TILE);
        chunk = std::min<int64_t>(chunk, (count + EK_TILE - 1) / EK_TILE * EK_TILE);
        if (chunk > c->stage_frames) {      // (the buffers only grow)
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
        if (const char *e = getenv("EK_UPLOAD_THREADS")) {
            char *end = nullptr;
            const long v = strtol(e, &end, 10);
            if (end == e || *end != '\0' || v < 1 || v > 64)
                return ek_fail(EK_EARG, "EK_UPLOAD_THREADS='%s': an integer 1 .. 64", e);
            n_thr = (int)v;
        }
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
                // (no C++ exception may cross the C ABI: a thread that cannot be
                // started -- std::system_error -- or a failed allocation leaves the
                // copy to this thread alone)
                bool pooled = false;
                try {
                    if (!c->pool)
                        c->pool = new EkCopyPool();
                    c->pool->start(n_thr);
                    pooled = !c->pool->th.empty();
                } catch (...) {
                    pooled = c->pool && !c->pool->th.empty();
                }
                if (pooled)
                    c->pool->copy(dst, src, bytes);
                else
                    memcpy(dst, src, bytes);
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
    c->no_qtiles = false;       // (another load, another try)
    return EK_OK;
}

// ---- centers given as coordinates -----------------------------------------------
int ek_upload_centers(ek_ctx *c, const float *xyz, int32_t K)
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
void ek_pam_forget(ek_ctx *c)
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

int ek_ensure_hist(ek_ctx *c, int32_t label)
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
    for (int m = 0; m < EK_N_FORMS; ++m)
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
        if (first_label + max_new > c->ti_rtab_cap) {
            EK_HIP(ek_wait(c));
            (void)hipFree(c->ti_rtab);
            c->ti_rtab = nullptr;
            c->ti_rtab_cap = 0;
            EK_HIP(hipMalloc((void **)&c->ti_rtab, (size_t)(first_label + max_new + 1) *
                                                       EK_MAX_CANDS * sizeof(float)));
            c->ti_rtab_cap = first_label + max_new + 1;
        }
        if (!c->ti_tmask)
            EK_HIP(hipMalloc((void **)&c->ti_tmask,
                             (size_t)std::max<int64_t>(c->n_tiles, 1) * sizeof(uint32_t)));
    }
    c->ti_tiles = c->ti_skipped = 0;
    // three launches per round (ek_round.hip) when the pieces it is built from
    // are the ones selected; rounds of 32 exist in that form only
    const bool fused = c->fused && c->chain == 1;
    if (Tmax == 32 && !fused)
        Tmax = EK_LEGACY_CANDS;
    if (Tmax >= 16) {
        const int eq = ek_ensure_qtiles(c);
        if (eq == EK_ENOMEM)
            Tmax = 8;
        else if (eq != EK_OK)
            return eq;
    }
    // an explicit request (option key 4 = 4, 8 or 16) pins the wide form
    const bool adaptive = c->cands == -1 && c->adapt;
    const int fpl = ek_pick_fpl(c);
    const int nt = ek_pick_nt(c);
    // the forms the run moves between, by candidates per pass
    int ladder[4], n_ladder = 0;
    // (triangle inequality, round 5: the rounds of 4 .. 32 candidates leave out
    // the tiles none of their candidates can change -- ek_round_ti_* --, the
    // one-center steps the tiles the new center cannot.
    // Round 4 ran one center per pass with the option on, 3.7 x slower than
    // without it on data where nothing can be left out.)
    if (adaptive || (tri && Tmax <= 1))
        ladder[n_ladder++] = 1;
    if (Tmax > 1) {
        // (the ladder is 1 / 8 / 16: rounds of 32 run only where the option pins them,
        // and a pinned form has no ladder -- `adaptive` needs candidates = -1)
        if (adaptive && Tmax >= 16)
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
    R.fm = c->fine_pick ? c->fm : nullptr;
    // (the per-prefix maxima in the pass: where they pay -- shards of up to ~half a
    // million frames: 10 % of a fit at 125 000; at 10^6 the pass's extra instructions
    // cost what the chain kernel's sweep did, profiles/r06/sweep_ab_*.log)
    R.sweep = (fused && (c->pass_sweep == 2 ||
                         (c->pass_sweep == 1 && c->n <= (int64_t)2048 * EK_TILE))) ? 1 : 0;
    R.top = c->top;
    R.ctile = c->ctile;
    R.ctrace = c->ctrace;
    R.hist = c->hist;
    R.ctl = c->ctl;
    R.tick = c->tick;
    R.rows = c->rows;
    R.vmask = c->vmask;
    R.cutoff = dist_cutoff;
    R.ti_tab = tri ? c->ti_rtab : nullptr;
    R.ti_stats = tri ? c->ti_stats : nullptr;
    bool pending = false;       // a fused round may have left a chain to apply
    int ti_pause = 0, ti_pause_next = 2;            // batches without masks / the next pause
    bool masks_fresh = false;   // ti_tmask describes the plan the next pass will run
    // How many far frames per label the candidate pick keeps (ek_top_dev.h): 4 suits
    // frames in clouds around templates, 16 a continuous landscape (round 5: the
    // walk 1.3e10 -> 2.5e10 pairs/s).  By the yield the rounds show: a batch that
    // accepts under 65 % of its guesses lets the next batch of the same form try the
    // other value; it stays if it accepts a tenth more, a look that loses waits twice
    // as long (80 % and a twentieth cost the ladder on template data 9 % at 125 000
    // frames: the early rounds of 8 sit just under 80 % there).
    int cap = c->pick_cap > 0 ? c->pick_cap : 4;
    const bool cap_adaptive = c->pick_cap <= 0;
    bool cap_probing = false;
    int cap_wait = 0, cap_next_wait = 2, cap_form = 0;
    double cap_yield_home = 0.0;
    unsigned long long ti_seen[2] = {0, 0};         // the counters at the last look
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
        // (the masks cost two small launches per round; where they leave nothing
        // out -- frames in no order: the bench's data -- they pause, for twice as
        // many batches each time a look finds nothing again)
        const bool masks = tri && fused && form >= 4 && ti_pause == 0;
        R.tmask = masks ? c->ti_tmask : nullptr;
        R.pick_cap = cap;
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
                ek_launch_round_ti(R, goal, c->stream);
                masks_fresh = R.tmask != nullptr;
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
        if (R.tmask && !masks_fresh) {  // (after a pause: the masks of the plan at hand)
            ek_launch_round_ti(R, goal, c->stream);
            masks_fresh = true;
        }
        if (!R.tmask)
            masks_fresh = false;        // (this batch's rounds do not make any)
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
                ek_launch_round_ti(R, goal, c->stream);
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
        unsigned long long ti_now[2] = {ti_seen[0], ti_seen[1]};
        if (R.tmask)
            EK_HIP(hipMemcpyAsync(ti_now, c->ti_stats, sizeof(ti_now),
                                  hipMemcpyDeviceToHost, c->stream));
        EK_HIP(ek_wait(c));
        if (R.tmask) {
            const unsigned long long looked = ti_now[0] - ti_seen[0],
                                     left = ti_now[1] - ti_seen[1];
            ti_seen[0] = ti_now[0];
            ti_seen[1] = ti_now[1];
            if (looked > 0 && left * 50 < looked) {     // under 2 % left out
                ti_pause = ti_pause_next;
                ti_pause_next = std::min(2 * ti_pause_next, 64);
            } else {
                ti_pause_next = 2;
            }
        } else if (ti_pause > 0 && !one) {
            --ti_pause;
        }
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
        if (cap_adaptive && !one && fused && ran > 0 && !probing) {
            const double yield = (double)got / ((double)ran * form);
            if (cap_probing) {
                cap_probing = false;
                if (form == cap_form && yield > cap_yield_home + 0.1) {
                    cap_next_wait = 2;          // the other value is home now
                } else {
                    cap = cap == 4 ? 16 : 4;    // back
                    cap_wait = cap_next_wait;
                    cap_next_wait = std::min(2 * cap_next_wait, 64);
                }
            } else if (cap_wait > 0) {
                --cap_wait;
            } else if (yield < 0.65 && left > 4 * form) {
                cap_yield_home = yield;
                cap_form = form;
                cap = cap == 4 ? 16 : 4;
                cap_probing = true;
            }
        }
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
    // (passes over the frames: a round of 32 is two)
    const int64_t passes = c->st_rounds[0] + c->st_rounds[1] + c->st_rounds[2] +
                           c->st_rounds[3] + 2 * c->st_rounds[4];
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

    const int T = ek_pick_cands(c, true);
    // (the triangle inequality's bookkeeping lives in ek_run_rounds, also for one
    // center per pass)
    if ((T > 1 || c->tri) && c->n > 0)
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
    // is too small to fill 32 x 32 tiles (results are bit-identical).  With
    // blocks of 64 centers to fill and room for the frames' quad copy: the
    // 16x16x4 form (ek_assign16_kernel), else 32x32x2 on the frame-minor tiles.
    const bool mfma = c->assign_variant == 2 || c->assign_variant == 3 ||
                      (c->assign_variant == 0 && n_centers >= 24 && c->n >= 64);
    bool k4 = c->assign_variant == 3 ||
              (c->assign_variant == 0 && n_centers >= 64 && c->n >= 64 && !c->no_qtiles);
    if (k4 && n_centers > 0) {
        const int eq = ek_ensure_qtiles(c);
        if (eq == EK_ENOMEM && c->assign_variant == 0)
            k4 = false;         // (no room for a third copy of the frames)
        else if (eq != EK_OK)
            return eq;
    }
    if (k4 && n_centers > 0) {
        const size_t need = ek_cblocks16_bytes(n_centers, c->A);
        if (need > c->cen_blocks_cap) {
            EK_HIP(ek_wait(c));
            (void)hipFree(c->cen_blocks);
            c->cen_blocks = nullptr;
            c->cen_blocks_cap = 0;
            EK_HIP(hipMalloc((void **)&c->cen_blocks, need));
            c->cen_blocks_cap = need;
        }
        ek_launch_cblocks16(c->cen_aos, n_centers, c->A, c->cen_blocks, c->stream);
        ek_launch_assign16(c->qtiles, c->G, c->n, c->A, c->cen_blocks, c->cen_G,
                           n_centers, c->dist, c->assign, c->stream);
    } else if (mfma && n_centers > 0) {
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


// ---- the QCP arithmetic as a kernel, on the caller's matrices (tests) -----------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_qcp_probe_kernel(const float *__restrict__ S, const double *__restrict__ Gx,
                    const double *__restrict__ Gy, int n_atoms,
                    const float *__restrict__ cur, int64_t m, float *__restrict__ full,
                    float *__restrict__ below, unsigned char *__restrict__ cert)
{
    const int64_t i = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i >= m)
        return;
    float s[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
        s[j] = S[9 * i + j];
    full[i] = ek_rmsd_from_S(s, Gx[i], Gy[i], n_atoms);
    below[i] = ek_rmsd_from_S_below(s, Gx[i], Gy[i], n_atoms, cur[i]);
    // bit 0: the closed-form certificate; bit 1: the second level (two Newton steps
    // from its bound), what ek_pass16_kernel asks of the pairs the first leaves
    cert[i] = (ek_far_certified_f32(s, (float)(Gx[i] + Gy[i]), n_atoms, cur[i]) ? 1 : 0) |
              (ek_far_certified2_f32(s, ek_far_t_frame((float)(Gx[i] + Gy[i]), n_atoms,
                                                       cur[i])) ? 2 : 0);
}

extern "C" int ek_qcp_probe(int device, const float *S, const double *Gx, const double *Gy,
                            int32_t n_atoms, const float *cur, int64_t m, float *full,
                            float *below, unsigned char *cert)
{
    if (!S || !Gx || !Gy || !cur || !full || !below || !cert || m < 0 || n_atoms < 1)
        return ek_fail(EK_EARG, "ek_qcp_probe: bad argument");
    if (m == 0)
        return EK_OK;
    EK_HIP(hipSetDevice(device));
    const size_t mm = (size_t)m;
    unsigned char *buf = nullptr;
    // Gx | Gy | S | cur | full | below | cert  (the float64 arrays first: every
    // offset is then a multiple of its element size whatever m is)
    const size_t off_Gx = 0, off_Gy = off_Gx + mm * 8, off_S = off_Gy + mm * 8,
                 off_cur = off_S + mm * 36, off_full = off_cur + mm * 4,
                 off_below = off_full + mm * 4, off_cert = off_below + mm * 4,
                 total = off_cert + mm;
    EK_HIP(hipMalloc((void **)&buf, total));
    hipError_t e = hipMemcpy(buf + off_S, S, mm * 36, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(buf + off_Gx, Gx, mm * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(buf + off_Gy, Gy, mm * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(buf + off_cur, cur, mm * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ek_qcp_probe_kernel, dim3((unsigned)((m + EK_BLOCK - 1) / EK_BLOCK)),
                           dim3(EK_BLOCK), 0, 0, (const float *)(buf + off_S),
                           (const double *)(buf + off_Gx), (const double *)(buf + off_Gy),
                           (int)n_atoms, (const float *)(buf + off_cur), m,
                           (float *)(buf + off_full), (float *)(buf + off_below),
                           buf + off_cert);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(full, buf + off_full, mm * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(below, buf + off_below, mm * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(cert, buf + off_cert, mm, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess)
        return ek_fail(EK_EHIP, "ek_qcp_probe: %s", hipGetErrorString(e));
    return EK_OK;
}
