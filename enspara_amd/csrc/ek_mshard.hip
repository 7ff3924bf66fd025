// ek_mshard.hip -- a k-centers round ACROSS SHARDS in three launches and one
// exchange.
//
// The reference's MPI iteration (enspara/cluster/kcenters.py:314-378) moves, per
// center, two pickled allgathers (:332-335), the owner's frame (mpi/ops.py:
// 169-212) and an allreduce for the stop test (mpi/ops.py:128-140).  Here a ROUND
// of up to 16 candidates (ek_spec.hip, ek_pass16.hip) accepts ~15 centers and
// moves one message per shard:
//
//   pass    every shard streams its frames against the round's candidates
//           (the same candidates on every shard: the plan below is computed
//           from the same messages by the same code), applies the chain the
//           previous round accepted on the way in, candidate 0, and keeps the
//           guesses' distances (ek_pass2_kernel / ek_pass16_kernel, FUSE).
//   chain   per-256-frame maxima of the states every prefix of the presumed
//           order would leave; last workgroup: this shard's (max distance,
//           global index) per prefix -- what the decision needs (kcenters.py:337:
//           the largest, lowest rank = lowest global index on ties) -- and,
//           SPECULATING that the whole chain will be accepted, its farthest
//           frames of the state that leaves (ek_pick_top_body: label-diverse),
//           as records.  That is the message; it goes out from here.
//   plan    all messages in.  Every workgroup: pairwise distances of the
//           records on offer (they steer the choice of guesses only).  Last
//           workgroup: the decision (ek_chain_walk on the global maxima: the
//           k-th candidate is accepted iff the farthest point of the state
//           before it is its frame), then -- if the whole chain was accepted --
//           the next round's candidates (greedy, as ek_round_next_kernel),
//           their tile, the plan.  If the chain broke at prefix k < n the
//           offers describe a state that never came to be: the next round runs
//           no pass, the chain kernel offers the records of state k instead
//           (the per-prefix maxima are still there), and the plan after that
//           is made from those.  ~5 % of the rounds at 10^6 frames per shard,
//           30-40 % at 125 000.
//
// Round 6, mailbox transport (EK_OPT_MS_TWO_PHASE, default): the per-prefix maxima
// go out FIRST, with a flag word of their own; every shard gathers the peers'
// (ek_ms_gather_maxima), walks the chain as the plan kernel will (ek_ms_walk) and
// offers the far frames of the state the chain REALLY leaves -- a broken chain
// costs no exchange of its own.  If the peers' maxima are in when a shard looks,
// it decides at once; if not it picks for the whole chain's state while they
// travel and again if the chain turns out to break.  The wait is bounded
// (EK_MS_HDR_TICKS): late maxima leave the offers as speculated, every message
// names the state its offers describe (EkMsMsg::state), and the plan kernel falls
// back to the exchange without a pass when they do not all describe the state the
// chain left.
//
// The presumed order is the order in which the greedy choice took the
// candidates (it IS the simulation ek_chain_simulate runs, on the offered
// distances); being a guess, it needs no exchange of its own: a wrong guess
// costs acceptance, never correctness.
//
// Transport.  `sys = 0`: the chain kernel leaves the message in a local buffer,
// the caller all-gathers (torch.distributed: RCCL, or gloo in the CPU tests of
// the host restatement), the plan kernel reads the gathered buffer.  `sys = 1`:
// per-peer mailboxes -- the chain kernel's last workgroup writes the message
// straight into every peer's memory (system-scope write-through stores over
// xGMI; hipIpc mappings, or plain pointers between contexts of one process),
// waits for its stores, and raises a sequence flag per peer; the plan kernel
// polls its own flags.  No host, no collective launch in a round: SURVEY.md
// section 8(e) "Fabric note".  Two mailbox sets alternate by round parity: a
// shard can only be one exchange ahead of a peer (it needs that peer's next
// message to go on).
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"
#include "ek_chain_dev.h"
#include "ek_top_dev.h"

#define EK_MS_THREADS 1024
#define EK_MS_FPT 4
// how long a shard polls a flag (s_sleep between polls) before it gives up on a
// peer: 10 s of the 100 MHz constant clock (wall_clock64: a time, whatever the
// shader clock and whether the flag is local or across xGMI) -- ranks enter a
// fit seconds apart when one is still loading frames --; a missing peer then
// shows as EkMsState::err (and ends the run: every later launch is a no-op)
// instead of a hung GPU.  Both waits of an exchange use the same bound.
#ifndef EK_MS_WAIT_TICKS
#define EK_MS_WAIT_TICKS (10ull * 100000000ull)
#endif
// ... and how long for the peers' per-prefix maxima before the offers go out as speculated
// (headers first, round 6): 2 ms.  Shards of one fit finish a pass within microseconds of
// one another; several contexts sharing ONE GPU (the tests) are the case that runs into it.
#ifndef EK_MS_HDR_TICKS
#define EK_MS_HDR_TICKS 200000ull
#endif

// measurement builds (-DEK_MS_STAMPS; tools/ms_stamps.py): where the chain and plan kernels'
// single workgroups spend their time -- 10 ns ticks summed per phase, read by ek_ms_stamps
#ifdef EK_MS_STAMPS
__device__ unsigned long long ek_ms_acc[32];
__device__ unsigned int ek_ms_cnt[32];
#define EK_MST_BEGIN unsigned long long mst_prev = wall_clock64()
#define EK_MST(k)                                                              \
    do {                                                                       \
        __syncthreads();                                                       \
        if (threadIdx.x == 0) {                                                \
            const unsigned long long mst_now = wall_clock64();                 \
            ek_ms_acc[k] += mst_now - mst_prev;                                \
            ek_ms_cnt[k] += 1u;                                                \
            mst_prev = mst_now;                                                \
        }                                                                      \
    } while (0)
extern "C" int ek_ms_stamps(double *mean_us, int64_t *counts, int clear)
{
    unsigned long long acc[32];
    unsigned int cnt[32];
    if (hipMemcpyFromSymbol(acc, HIP_SYMBOL(ek_ms_acc), sizeof(acc)) != hipSuccess ||
        hipMemcpyFromSymbol(cnt, HIP_SYMBOL(ek_ms_cnt), sizeof(cnt)) != hipSuccess)
        return -1;
    for (int k = 0; k < 32; ++k) {
        mean_us[k] = cnt[k] ? 0.01 * (double)acc[k] / cnt[k] : 0.0;
        counts[k] = cnt[k];
    }
    if (clear) {
        for (int k = 0; k < 32; ++k) {
            acc[k] = 0;
            cnt[k] = 0;
        }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(ek_ms_acc), acc, sizeof(acc));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(ek_ms_cnt), cnt, sizeof(cnt));
    }
    return 0;
}
#else
#define EK_MST_BEGIN
#define EK_MST(k)
#endif

// ---- system-scope accesses (mailbox transport) ---------------------------------
__device__ __forceinline__ void ek_sys_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ek_sys_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
typedef float ek_f4 __attribute__((ext_vector_type(4)));
// 16 bytes at system scope (write-through / past the caches); the loads of a
// batch are issued together and waited for once
__device__ __forceinline__ void ek_sys_store4(void *p, ek_f4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <bool SYS, int N>
__device__ __forceinline__ void ek_msg_load4(const void *const (&p)[N], ek_f4 (&v)[N])
{
    if (SYS) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1"
                         : "=v"(v[i]) : "v"(p[i]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < N; ++i)     // (the values are there only after the wait)
            asm volatile("" : "+v"(v[i]));
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            v[i] = *(const ek_f4 *)p[i];
    }
}
template <bool SYS>
__device__ __forceinline__ uint32_t ek_msg_load(const uint32_t *p)
{
    return SYS ? ek_sys_load(p) : *p;
}
template <bool SYS>
__device__ __forceinline__ float ek_msg_loadf(const float *p)
{
    return __uint_as_float(ek_msg_load<SYS>((const uint32_t *)p));
}

// shard `rk`'s message of the exchange with sequence number `seq`
__device__ __forceinline__ const unsigned char *ek_ms_src(const EkMsXchg &x, int rk,
                                                          uint32_t seq)
{
    const size_t slot = x.sys ? (size_t)(seq & 1u) * x.world + rk : (size_t)rk;
    return x.src + slot * x.msg_bytes;
}

// ---------------------------------------------------------------------------
// chain: per-prefix maxima, this shard's headers and offers, the message out
// ---------------------------------------------------------------------------
// Helper workgroups: writing up to 64 records of 12 A bytes to up to 8 peers is
// 230 KB of stores -- 45 us from the one workgroup that knows which frames they
// are.  EK_MS_HELPERS extra workgroups at the END of the grid wait for that
// workgroup's list and share the records.  (They are dispatched after the
// workgroups they wait for, which in turn wait for nobody: no residency
// assumption.)
#define EK_MS_HELPERS 16
struct EkMsPub {            // the list, published for the helpers (in r.top + 2048)
    int32_t n_off;
    int32_t pad[3];
    uint32_t idx[EK_TOP_M];
    float val[EK_TOP_M];
};

// The global per-prefix maxima: state k's maximum over the shards -- the largest, the lowest
// global index on ties (kcenters.py:337) -- out of the cn headers of every shard's message.
// The whole workgroup (NT threads): one 16-byte load per (state, shard), all in flight at
// once -- a thread that walked them one after the other spent 0.5 us per header on the
// latency of uncached memory --, then an LDS max on the value's ordered bits and an LDS min
// on the index among the holders of that value.  Also fetched here, in parallel: what the
// walk compares them with (the stored frame of the presumed order's k-th candidate).
struct EkMsMaxima {
    float gv[EK_MAX_CANDS];         // global maximum of state k
    long long gg[EK_MAX_CANDS];     // ... its frame
    int gok[EK_MAX_CANDS];          // ... whether any shard has one
    int cj[EK_MAX_CANDS];           // the presumed order's k-th candidate
    long long cg[EK_MAX_CANDS];     // ... its frame
    unsigned int key[EK_MAX_CANDS];
    unsigned long long idx[EK_MAX_CANDS];
};
__device__ __forceinline__ unsigned int ek_ord_bits(float v)
{
    const unsigned int b = __float_as_uint(v == 0.f ? 0.f : v);     // (-0 ties with +0)
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
template <int NT> struct EkMsHdrRegs {
    static constexpr int IT = (EK_MAX_CANDS * EK_MS_MAX_WORLD + NT - 1) / NT;
    ek_f4 h[IT];
};
template <bool SYS, int NT>
__device__ __forceinline__ void ek_ms_headers_load(const EkMsXchg &x, uint32_t seq, int cn,
                                                   EkMsHdrRegs<NT> &hr)
{
    const int tid = threadIdx.x;
    constexpr int IT = EkMsHdrRegs<NT>::IT;
    const int total = cn * x.world;
    const void *p[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int t = tid + u * NT < total ? tid + u * NT : 0;
        p[u] = ek_ms_src(x, t % x.world, seq) + sizeof(EkMsMsg) + (size_t)(t / x.world) * 16;
    }
    if (tid < total)        // (the threads of a first stride that holds nothing skip it all)
        ek_msg_load4<SYS, IT>(p, hr.h);
}
template <int NT>
__device__ __forceinline__ void ek_ms_maxima_combine(const EkRound &r, const EkMsXchg &x,
                                                     int cn, const EkMsHdrRegs<NT> &hr,
                                                     EkMsMaxima &m)
{
    const int tid = threadIdx.x;
    constexpr int IT = EkMsHdrRegs<NT>::IT;
    const int total = cn * x.world;
    const ek_f4 *h = hr.h;
    if (tid < EK_MAX_CANDS) {
        m.key[tid] = 0u;
        m.idx[tid] = ~0ull;
        if (tid < cn) {
            const int j = r.ord->cand[tid];
            m.cj[tid] = j;
            m.cg[tid] = r.plan->gidx[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IT; ++u)
        if (tid + u * NT < total && __float_as_uint(h[u][1]) != 0u)
            atomicMax(&m.key[(tid + u * NT) / x.world], ek_ord_bits(h[u][0]));
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IT; ++u)
        if (tid + u * NT < total && __float_as_uint(h[u][1]) != 0u &&
            ek_ord_bits(h[u][0]) == m.key[(tid + u * NT) / x.world])
            atomicMin(&m.idx[(tid + u * NT) / x.world],
                      (unsigned long long)__float_as_uint(h[u][2]) |
                          ((unsigned long long)__float_as_uint(h[u][3]) << 32));
    __syncthreads();
    if (tid < EK_MAX_CANDS) {
        const unsigned int k = m.key[tid];
        const unsigned int b = k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu);
        m.gok[tid] = (tid < cn && k != 0u) ? 1 : 0;
        m.gv[tid] = m.gok[tid] ? __uint_as_float(b) : 0.f;
        m.gg[tid] = m.gok[tid] ? (long long)m.idx[tid] : 0;
    }
    __syncthreads();
}
template <bool SYS, int NT>
__device__ __forceinline__ void ek_ms_gather_maxima(const EkRound &r, const EkMsXchg &x,
                                                    uint32_t seq, int cn, EkMsMaxima &m)
{
    EkMsHdrRegs<NT> hr;
    ek_ms_headers_load<SYS, NT>(x, seq, cn, hr);
    ek_ms_maxima_combine<NT>(r, x, cn, hr, m);
}

// Headers first (round 6): the walk of ek_ms_plan_kernel over the global per-prefix maxima,
// read-only -- how many of the chain's cn candidates become centers.  One thread.
__device__ __forceinline__ int ek_ms_walk(const EkRound &r, const EkMsState *ms,
                                          const EkMsMaxima &m, int cn)
{
    const EkCtl c = *r.ctl;
    int n_done = r.plan->label + 1;     // (candidate 0 is a center by then)
    int na = 0;
    if (ms->err)
        return 0;
    for (int k = 0; k < cn; ++k) {
        if (c.stopped || n_done >= c.limit || !m.gok[k])
            break;
        if (!((double)m.gv[k] > r.cutoff))      // kcenters.py:217
            break;
        if (m.gg[k] != m.cg[k])
            break;                      // the farthest point is not stored
        ++n_done;
        ++na;
    }
    return na;
}

__global__ void __launch_bounds__(EK_MS_THREADS)
ek_ms_chain_kernel(EkRound r, EkMsState *ms, EkMsXchg x, int nblk)
{
    __shared__ float sv[EK_MAX_CANDS];
    __shared__ uint32_t si[EK_MAX_CANDS];
    __shared__ uint32_t t_idx[EK_TOP_M];
    __shared__ float t_val[EK_TOP_M];
    __shared__ int t_n;
    extern __shared__ uint32_t skip[];      // pick fallback for very large shards
    const int mode = ms->mode;              // (written by the launch before)
    if (mode == 0)
        return;                             // the run is over
    const int tid = threadIdx.x;
    const uint32_t seq = ms->seq;
    const int n_dst = x.sys ? x.world : 1;
    const size_t slot = x.sys ? (size_t)(seq & 1u) * x.world + x.rank : 0;
    const size_t head_bytes = sizeof(EkMsMsg) + EK_MAX_CANDS * sizeof(EkMaxHdr);
    EkMsPub *pub = (EkMsPub *)(r.top + 2048);
    if ((int)blockIdx.x >= nblk) {
        // ---- a helper: records h, h + H, .. of the list, to every destination ----------
        const int h = (int)blockIdx.x - nblk;
        __shared__ int s_ok;
        EK_MST_BEGIN;
        if (tid == 0) {
            const uint64_t t_start = wall_clock64();
            s_ok = 1;
            while (__hip_atomic_load(r.tick + 5, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) != seq + 1u) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t_start > EK_MS_WAIT_TICKS) {
                    s_ok = 0;
                    break;
                }
            }
        }
        __syncthreads();
        if (h == 0)
            EK_MST(6);  // helper 0: from its start to the go-ahead (the tail's time)
        if (tid < EK_TOP_M) {
            t_idx[tid] = (uint32_t)ek_coh_load((const int32_t *)&pub->idx[tid]);
            t_val[tid] = ek_coh_load(&pub->val[tid]);
        }
        if (tid == 0)
            t_n = s_ok ? ek_coh_load(&pub->n_off) : 0;
        __syncthreads();
        const int n_off = t_n;
        const int A3 = 3 * r.A;
        const int cpr = (int)(ek_rec_bytes(r.A) / 16);      // 16-byte chunks per record
        // (its records x the destinations x the chunks as ONE index space: with few
        // destinations a loop per record left most threads idle, four times over)
        const int n_mine = h < n_off ? (n_off - h + EK_MS_HELPERS - 1) / EK_MS_HELPERS : 0;
        const int per_rec = n_dst * cpr;
        {
            for (int item0 = tid; item0 < n_mine * per_rec; item0 += EK_MS_THREADS) {
                const int j = h + (item0 / per_rec) * EK_MS_HELPERS;
                const int item = item0 % per_rec;
                const uint32_t fi = t_idx[j];
                const float *src = r.aos + (size_t)fi * A3;
                const int p = item / cpr, q = item % cpr;
                ek_f4 v;
                if (q == 0) {
                    const long long g = r.goff + (long long)fi;
                    v[0] = t_val[j];
                    v[1] = __uint_as_float(1u);
                    v[2] = __uint_as_float((uint32_t)((unsigned long long)g & 0xffffffffu));
                    v[3] = __uint_as_float((uint32_t)((unsigned long long)g >> 32));
                } else if (q == 1) {
                    const unsigned long long tr = __double_as_longlong(r.G[fi]);
                    v[0] = __uint_as_float((uint32_t)(tr & 0xffffffffu));
                    v[1] = __uint_as_float((uint32_t)(tr >> 32));
                    v[2] = 0.f;
                    v[3] = 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int w = 4 * (q - 2) + e;
                        v[e] = w < A3 ? src[w] : 0.f;
                    }
                }
                unsigned char *d = x.dst[p] + slot * x.msg_bytes + head_bytes +
                                   (size_t)j * ek_rec_bytes(r.A) + (size_t)q * 16;
                if (x.sys)
                    ek_sys_store4(d, v);
                else
                    *(ek_f4 *)d = v;
            }
        }
        // this helper's stores acknowledged; the one that finishes last tells the peers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (h == 0)
            EK_MST(7);  // helper 0: its records written and acknowledged
#ifdef EK_MS_STAMPS
        const unsigned long long mst_t5 = wall_clock64();
#endif
        __shared__ int s_last;
        __shared__ unsigned long long s_wait;
        if (tid == 0) {
            s_wait = 0;
            s_last = __hip_atomic_fetch_add(r.tick + 6, 1u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT) ==
                     EK_MS_HELPERS - 1;
            if (s_last)
                r.tick[6] = 0;
            if (!s_ok && !ms->err) {        // the list of this shard's own records
                ms->err = 2;
                ms->err_seq = seq;
            }
        }
        __syncthreads();
        if (s_last && x.sys && tid < x.world) {
            ek_sys_store(x.dflag[tid] + (slot * 16), seq + 1u);
            // ... and this launch ends when every peer's message of this exchange
            // is in: ONE workgroup waits, not the thousand of the plan kernel
            // that needs them -- a kernel that fills the chip while it polls
            // starves whatever else shares the GPU (another shard's launches, in
            // the tests: that is a deadlock until the time-out)
            const uint32_t *f = x.sflag + ((size_t)(seq & 1u) * x.world + tid) * 16;
            const uint64_t t_start = wall_clock64();
            while (ek_sys_load(f) != seq + 1u) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t_start > EK_MS_WAIT_TICKS) {
                    if (!ms->err) {         // peer `tid`'s message did not arrive
                        ms->err = 0x100 + tid;
                        ms->err_seq = seq;
                    }
                    break;
                }
            }
            // how long this exchange kept the shard waiting (ek_ms_diag): the
            // longest wait over the peers, and the wait for its own flag (the
            // latency of a store that goes nowhere: the floor)
            const unsigned long long waited = wall_clock64() - t_start;
            atomicMax(&s_wait, waited);
            if (tid == x.rank)
                atomicAdd(&ms->wait_ticks_own, waited);
        }
        __syncthreads();
        if (s_last && x.sys && tid == 0)
            atomicAdd(&ms->wait_ticks_max, s_wait);
#ifdef EK_MS_STAMPS
        if (s_last && tid == 0) {   // the last helper: its flags out, all peers' flags seen
            ek_ms_acc[5] += wall_clock64() - mst_t5;
            ek_ms_cnt[5] += 1u;
        }
#endif
        return;
    }
    const int nb = (int)((r.n + EK_BLOCK - 1) / EK_BLOCK);
    const int cn = mode == 1 ? r.ord->n : 0;
    // (maxima per 64 frames for the offers: on shards up to ~half a million frames,
    // see ek_round_chain_kernel)
    const bool fine_ok = r.fm != nullptr && 4 * nb <= 8 * EK_RED_THREADS;
    // (round 6: a pass of 16 candidates took these maxima itself, EkFuse::sweep_pm)
    const bool swept = r.sweep && r.T == 16;
    if (cn > 0 && !swept) {
        // pm[(k - 1) * nb + w] = first-index arg-max over frames [256 w, 256 w + 256)
        // of min(dist, vec[order[0]], .., vec[order[k-1]]), k = 1 .. cn (state 0
        // is what the pass left in blockmax): as ek_round_chain_kernel, sixteen
        // prefixes at a time
        const int64_t f0 = ((int64_t)blockIdx.x * EK_MS_THREADS + tid) * EK_MS_FPT;
        const bool whole = f0 + EK_MS_FPT <= r.n;
        float run[EK_MS_FPT];
        const uint32_t vm = f0 < r.n ? r.vmask[f0 >> 6] : 0u;
        if (whole) {
            const float4 t = *(const float4 *)(r.dist + f0);
            run[0] = t.x; run[1] = t.y; run[2] = t.z; run[3] = t.w;
        } else {
#pragma unroll
            for (int q = 0; q < EK_MS_FPT; ++q)
                run[q] = (f0 + q < r.n) ? r.dist[f0 + q] : 0.f;
        }
        const int64_t wg = ((int64_t)blockIdx.x * EK_MS_THREADS + tid) / EK_WAVE;
#pragma unroll
        for (int kb = 0; kb < EK_MAX_CANDS; kb += 16) {
            if (kb >= cn + 1)               // uniform
                break;
            float dv[16][EK_MS_FPT];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = kb + kk;
#pragma unroll
                for (int q = 0; q < EK_MS_FPT; ++q)
                    dv[kk][q] = __builtin_inff();
                if (k >= 1 && k <= cn && ((vm >> r.ord->cand[k - 1]) & 1u)) {
                    const float *v = r.vecs + (size_t)(r.ord->cand[k - 1] - 1) * r.n_pad + f0;
                    if (whole) {
                        const float4 t = *(const float4 *)v;
                        dv[kk][0] = t.x; dv[kk][1] = t.y; dv[kk][2] = t.z; dv[kk][3] = t.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < EK_MS_FPT; ++q)
                            if (f0 + q < r.n)
                                dv[kk][q] = v[q];
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = kb + kk;
                if (k >= 1 && k <= cn) {        // uniform
                    float v = -__builtin_inff();
                    uint32_t i = 0xffffffffu;
#pragma unroll
                    for (int q = 0; q < EK_MS_FPT; ++q) {
                        if (f0 + q < r.n) {
                            if (dv[kk][q] < run[q])     // kcenters.py:304
                                run[q] = dv[kk][q];
                            if (ek_better(run[q], (uint32_t)(f0 + q), v, i)) {
                                v = run[q];
                                i = (uint32_t)(f0 + q);
                            }
                        }
                    }
                    // (the rows of 16 lanes first: of the state the whole chain would
                    // leave, the maxima per 64 frames are kept for the offers)
                    ek_row_argmax(v, i);
                    if (k == cn && fine_ok && (tid & 15) == 0 && wg < nb)
                        ek_coh_store_bm(&r.fm[4 * (size_t)wg + ((tid >> 4) & 3)], v, i);
                    ek_rows_to_wave_argmax(v, i);
                    if ((tid & (EK_WAVE - 1)) == 0 && wg < nb)
                        ek_coh_store_bm(&r.pm[(size_t)(k - 1) * nb + wg], v, i);
                }
            }
        }
    }
    // (a single workgroup that swept still has to see ALL its waves' maxima before it
    // reduces them: round 6 skipped the arrival -- waitcnt + barrier -- for one workgroup
    // and a shard of up to 4096 frames in rounds of 8 read maxima of the round before, or
    // of nobody: tools/fuzz_ms.py)
    if ((nblk > 1 || (cn > 0 && !swept)) && !ek_arrive_last(r.tick + 1, (unsigned int)nblk))
        return;
    // ---- the last workgroup ---------------------------------------------------------
    EK_MST_BEGIN;
    // this shard's maximum of states 0 .. cn - 1 (what the decision looks at)
    if (cn > 0)
        ek_chain_reduce<true>(r.blockmax, r.pm, nb, nb, cn, sv, si);
    __syncthreads();
    EK_MST(0);      // per-prefix maxima reduced
    // ---- round 6, mailbox transport: the headers first, the decision here ---------------
    // Every shard sends its per-prefix maxima (the head of its message), waits for the
    // others', and walks the chain as the plan kernel will (ek_chain_walk on the global
    // maxima): it then knows the state the chain REALLY leaves and offers that state's far
    // frames.  Before, the offers speculated that the whole chain would hold, and a chain
    // that broke -- 30 % of the rounds of the headline case split 8 ways -- was offered
    // again in an exchange of its own (chain + plan + tile kernels on every shard).
    __shared__ int s_na, s_late;
    __shared__ EkMsMaxima s_mx;
    const int head_words = (int)(head_bytes / 4);
    const bool two = x.sys && x.two_phase && mode == 1 && cn > 0;
    if (two) {
        for (int item = tid; item < n_dst * head_words; item += EK_MS_THREADS) {
            const int p = item / head_words, w = item % head_words;
            if (w < (int)(sizeof(EkMsMsg) / 4))
                continue;               // (n_recs, cn: with the list, below)
            const int k = (w - (int)(sizeof(EkMsMsg) / 4)) / 4;
            const int u = (w - (int)(sizeof(EkMsMsg) / 4)) % 4;
            const bool ok = k < cn && si[k] != 0xffffffffu;
            const long long g = ok ? r.goff + (long long)si[k] : -1;
            const uint32_t val = u == 0 ? __float_as_uint(ok ? sv[k] : -__builtin_inff())
                               : u == 1 ? (ok ? 1u : 0u)
                               : u == 2 ? (uint32_t)((unsigned long long)g & 0xffffffffu)
                                        : (uint32_t)((unsigned long long)g >> 32);
            ek_sys_store((uint32_t *)(x.dst[p] + slot * x.msg_bytes) + w, val);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (word 1 of a flag slot: "its headers are in"; word 0 stays "its message is in")
        if (tid < x.world)
            ek_sys_store(x.dflag[tid] + (slot * 16) + 1, seq + 1u);
    }
    EK_MST(1);      // headers out
    // its farthest frames of the state the offers are for: the one the whole chain would
    // leave or (the chain broke before: mode 2) the one it did leave.  Headers first: if
    // the peers' are in already (this shard was the last, or the only one) the chain is
    // decided at once; if not, the pick for the whole chain's state runs while they travel,
    // and a chain that then turns out to break (one round in five to three) is picked
    // again for the state it leaves.
    int ps = mode == 1 ? cn : ms->pick_state;
    bool decided = false;
    if (two) {
        __shared__ int s_in;
        if (tid == 0)
            s_in = 1;
        __syncthreads();
        // (the flags, THEN the headers: a header fetched before its flag was seen could
        // be an older exchange's)
        if (tid < x.world && tid != x.rank &&
            ek_sys_load(x.sflag + ((size_t)(seq & 1u) * x.world + tid) * 16 + 1) != seq + 1u)
            s_in = 0;
        __syncthreads();
        if (s_in) {
            ek_ms_gather_maxima<true, EK_MS_THREADS>(r, x, seq, cn, s_mx);
            if (tid == 0)
                s_na = ek_ms_walk(r, ms, s_mx, cn);
            __syncthreads();
            ps = s_na;
            decided = true;
        }
    }
    EK_MST(2);      // the look at the peers' flags, the walk if they are in
    EkTop *top = (EkTop *)r.top;
    for (;;) {
        // (the state the whole chain would leave: its maxima per 64 frames)
        const bool fine = fine_ok && mode == 1 && cn > 0 && ps == cn;
        const EkBlockMax *state = fine ? r.fm
                                       : (ps == 0 ? r.blockmax : r.pm + (size_t)(ps - 1) * nb);
        ek_pick_top_body<true>(state, fine ? 4 * nb : nb, top, skip, r.assign, r.pick_cap);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!two || decided)
            break;
        decided = true;
        // (a BOUNDED wait: headers that are not in after EK_MS_HDR_TICKS -- a peer far behind
        // -- leave the offers as speculated, and the plan kernel deals with a broken chain
        // as rounds 3-5 did; sixteen helper workgroups are waiting for this one)
        if (tid == 0)
            s_late = 0;
        __syncthreads();
        if (tid < x.world && tid != x.rank) {
            const uint32_t *f = x.sflag + ((size_t)(seq & 1u) * x.world + tid) * 16 + 1;
            const uint64_t t_start = wall_clock64();
            while (ek_sys_load(f) != seq + 1u) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t_start > EK_MS_HDR_TICKS) {
                    s_late = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (s_late)
            break;
        ek_ms_gather_maxima<true, EK_MS_THREADS>(r, x, seq, cn, s_mx);
        if (tid == 0)
            s_na = ek_ms_walk(r, ms, s_mx, cn);
        __syncthreads();
        if (s_na == cn)
            break;
        ps = s_na;
    }
    EK_MST(3);      // pick(s), the wait for the headers
    // the list for the helpers, and the head of the message: this workgroup's part
    if (tid < EK_TOP_M) {
        ek_coh_store((int32_t *)&pub->idx[tid], (int32_t)top->idx[tid]);
        ek_coh_store(&pub->val[tid], top->val[tid]);
    }
    const int n_off = top->n < x.offer ? top->n : x.offer;
    if (tid == 0)
        ek_coh_store(&pub->n_off, (int32_t)n_off);
    for (int item = tid; item < n_dst * head_words; item += EK_MS_THREADS) {
        const int p = item / head_words, w = item % head_words;
        uint32_t val = 0;
        if (two && w >= (int)(sizeof(EkMsMsg) / 4))
            continue;                   // (the headers went out first)
        if (w < (int)(sizeof(EkMsMsg) / 4)) {
            val = w == 0 ? (uint32_t)n_off
                : (w == 1 ? (uint32_t)cn : (w == 2 ? (uint32_t)(mode == 1 ? ps : -1) : 0u));
        } else {
            const int k = (w - (int)(sizeof(EkMsMsg) / 4)) / 4;
            const int u = (w - (int)(sizeof(EkMsMsg) / 4)) % 4;
            const bool ok = k < cn && si[k] != 0xffffffffu;
            const long long g = ok ? r.goff + (long long)si[k] : -1;
            val = u == 0 ? __float_as_uint(ok ? sv[k] : -__builtin_inff())
                : u == 1 ? (ok ? 1u : 0u)
                : u == 2 ? (uint32_t)((unsigned long long)g & 0xffffffffu)
                         : (uint32_t)((unsigned long long)g >> 32);
        }
        uint32_t *d = (uint32_t *)(x.dst[p] + slot * x.msg_bytes) + w;
        if (x.sys)
            ek_sys_store(d, val);
        else
            *d = val;
    }
    // all of that in place, then the helpers may go
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    EK_MST(4);      // list + head published
    if (tid == 0) {
        r.tick[1] = 0;
        __hip_atomic_store(r.tick + 5, seq + 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

void ek_launch_ms_chain(const EkRound &r, EkMsState *ms, const EkMsXchg &x,
                        hipStream_t s)
{
    const int64_t per = (int64_t)EK_MS_THREADS * EK_MS_FPT;
    // (nothing to sweep where the pass took the maxima: one workgroup + the helpers)
    const unsigned blocks = (r.sweep && r.T == 16)
                                ? 1u : (unsigned)std::max<int64_t>(1, (r.n + per - 1) / per);
    const int nb = (int)((r.n + EK_BLOCK - 1) / EK_BLOCK);
    const size_t lds = (size_t)((nb + 31) / 32 + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL(ek_ms_chain_kernel, dim3(blocks + EK_MS_HELPERS),
                       dim3(EK_MS_THREADS), lds, s, r, ms, x, (int)blocks);
}

// ---------------------------------------------------------------------------
// plan: all shards' messages -> the decision, the next round's candidates
// ---------------------------------------------------------------------------
// record slot s = shard s / offer, its (s % offer)-th offer
template <bool SYS>
__device__ __forceinline__ const uint32_t *ek_ms_rec(const EkMsXchg &x, int slot,
                                                     uint32_t seq, int A)
{
    const unsigned char *m = ek_ms_src(x, slot / x.offer, seq);
    return (const uint32_t *)(m + sizeof(EkMsMsg) + EK_MAX_CANDS * sizeof(EkMaxHdr) +
                              (size_t)(slot % x.offer) * ek_rec_bytes(A));
}

template <int T, bool SYS>
__global__ void __launch_bounds__(EK_BLOCK)
ek_ms_plan_kernel(EkRound r, EkMsState *ms, EkMsXchg x, float *D)
{
    const int mode = ms->mode;
    if (mode == 0)
        return;
    const int tid = threadIdx.x;
    const uint32_t seq = ms->seq;
    const int A = r.A;
    __shared__ int s_nrec[EK_MS_MAX_WORLD];
    __shared__ int s_state[EK_MS_MAX_WORLD];
    EK_MST_BEGIN;
    // (mailbox transport: the chain kernel did not end before every peer's
    // message of this exchange was in)
    if (tid < x.world)
        s_nrec[tid] = 0;
    __syncthreads();
    // (the values of all record slots are fetched with the counts, not after them: one
    // round trip to uncached memory instead of two; a slot beyond a shard's count holds an
    // older exchange's record and is masked below)
    const int n_slots = x.world * x.offer;      // <= EK_MS_SLOTS
    float hv_raw = -__builtin_inff();
    if (tid < EK_MS_SLOTS && tid < n_slots)
        hv_raw = __uint_as_float(ek_msg_load<SYS>(ek_ms_rec<SYS>(x, tid, seq, A)));
    if (tid < x.world) {
        s_nrec[tid] = (int)ek_msg_load<SYS>((const uint32_t *)ek_ms_src(x, tid, seq));
        s_state[tid] = (int)ek_msg_load<SYS>((const uint32_t *)ek_ms_src(x, tid, seq) + 2);
    }
    __syncthreads();
    // Up to 128 records are on offer (EK_MS_SLOTS / world per shard), of which the 64 with the
    // largest distances -- slot order on ties -- compete: the far frames of a state are not
    // spread evenly over the shards (64 of them over 8 shards: 8 +- 2.6 per shard), and with
    // 64 / world offers per shard the ones beyond a shard's quota were on no list.  smap[i] =
    // the slot of the i-th best; every workgroup works the map out for itself.
    __shared__ float s_hv[EK_MS_SLOTS];
    __shared__ int smap[64];
    __shared__ int s_nval;
    if (tid < 64)
        smap[tid] = 0;
    if (tid == 0)
        s_nval = 0;
    if (tid < EK_MS_SLOTS) {
        float v = -__builtin_inff();
        if (tid < n_slots && tid % x.offer < s_nrec[tid / x.offer])
            v = hv_raw;
        s_hv[tid] = v;
    }
    __syncthreads();
    if (tid < EK_MS_SLOTS && tid < n_slots && tid % x.offer < s_nrec[tid / x.offer]) {
        const float v = s_hv[tid];
        int rank = 0;
        for (int u = 0; u < n_slots; ++u) {
            const float o = s_hv[u];
            rank += (o > v || (o == v && u < tid)) ? 1 : 0;
        }
        if (rank < 64)
            smap[rank] = tid;
        atomicAdd(&s_nval, 1);
    }
    __syncthreads();
    const int n_cmp = s_nval < 64 ? s_nval : 64;    // records that compete
    if (blockIdx.x == 0)
        EK_MST(8);      // workgroup 0: counts, values and ranks of the records on offer
    {
        // one wave per pair of competing records; the values only steer the guesses,
        // so the lanes may stride over the atoms (but every shard computes the
        // same values from the same messages, and so the same plan)
        const int lane = tid & (EK_WAVE - 1);
        const int w = blockIdx.x * (EK_BLOCK / EK_WAVE) + tid / EK_WAVE;
        const int i = w / 64, j = w % 64;
        if (i < j && j < n_cmp) {
            const uint32_t *ri = ek_ms_rec<SYS>(x, smap[i], seq, A);
            const uint32_t *rj = ek_ms_rec<SYS>(x, smap[j], seq, A);
            const unsigned char *bi = (const unsigned char *)(ri + 8);
            const unsigned char *bj = (const unsigned char *)(rj + 8);
            const int cpr_c = (3 * A + 3) / 4;      // 16-byte chunks of coordinates
            float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // a lane takes 4 atoms = three chunks of either record at a time
            // (two strides of the atoms in flight at once: 300 atoms are one round trip to
            // the mailboxes, not two)
            for (int a0 = 4 * lane; a0 < A; a0 += 8 * EK_WAVE) {
                const void *p[12];
                ek_f4 v[12];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int c = min(3 * ((a0 + t * 4 * EK_WAVE) / 4) + u, cpr_c - 1);
                        p[6 * t + u] = bi + 16 * c;
                        p[6 * t + 3 + u] = bj + 16 * c;
                    }
                ek_msg_load4<SYS, 12>(p, v);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (a0 + t * 4 * EK_WAVE + e < A) {
                            const float x0 = v[6 * t + (3 * e) / 4][(3 * e) % 4],
                                        x1 = v[6 * t + (3 * e + 1) / 4][(3 * e + 1) % 4],
                                        x2 = v[6 * t + (3 * e + 2) / 4][(3 * e + 2) % 4];
                            const float y0 = v[6 * t + 3 + (3 * e) / 4][(3 * e) % 4],
                                        y1 = v[6 * t + 3 + (3 * e + 1) / 4][(3 * e + 1) % 4],
                                        y2 = v[6 * t + 3 + (3 * e + 2) / 4][(3 * e + 2) % 4];
                            S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
                            S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
                            S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
                        }
                    }
            }
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1)
                    S[q] += __shfl_xor(S[q], off, 64);
            if (lane == 0) {
                const unsigned long long ti =
                    (unsigned long long)ek_msg_load<SYS>(ri + 4) |
                    ((unsigned long long)ek_msg_load<SYS>(ri + 5) << 32);
                const unsigned long long tj =
                    (unsigned long long)ek_msg_load<SYS>(rj + 4) |
                    ((unsigned long long)ek_msg_load<SYS>(rj + 5) << 32);
                const float d = ek_rmsd_from_S(S, __longlong_as_double((long long)ti),
                                               __longlong_as_double((long long)tj), A);
                ek_coh_store(&D[i * 64 + j], d);
                ek_coh_store(&D[j * 64 + i], d);
            }
        }
    }
    if (blockIdx.x == 0)
        EK_MST(9);      // workgroup 0: its pairs
    if (!ek_arrive_last_tree(r.tick + 2, r.tick + 64))
        return;
    // ---- the last workgroup ---------------------------------------------------------
#ifdef EK_MS_STAMPS
    mst_prev = wall_clock64();
#endif
    __shared__ float sD[64 * 64];
    __shared__ float sval[64];
    __shared__ long long sgidx[64];
    __shared__ int sel[EK_MAX_CANDS];
    __shared__ int n_sel;
    __shared__ EkMsMaxima s_mx;
    __shared__ int s_over, s_repick, s_short;
    {
        constexpr int PER = 64 * 64 / EK_BLOCK;
        float dreg[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u)
            dreg[u] = ek_coh_load(&D[tid + u * EK_BLOCK]);
#pragma unroll
        for (int u = 0; u < PER; ++u)
            sD[tid + u * EK_BLOCK] = dreg[u];
    }
    if (tid < 64) {
        float v = -__builtin_inff();
        long long g = -1;
        if (tid < n_cmp) {
            const uint32_t *rr = ek_ms_rec<SYS>(x, smap[tid], seq, A);
            v = __uint_as_float(ek_msg_load<SYS>(rr));
            g = (long long)((unsigned long long)ek_msg_load<SYS>(rr + 2) |
                            ((unsigned long long)ek_msg_load<SYS>(rr + 3) << 32));
        }
        sval[tid] = v;
        sgidx[tid] = g;
    }
    const int cn = mode == 1 ? r.ord->n : 0;
    if (tid < EK_MAX_CANDS)
        sel[tid] = 0;
    // state k's maximum over the shards: the largest, the lowest global index on ties
    // (kcenters.py:337)
    ek_ms_gather_maxima<SYS, EK_BLOCK>(r, x, seq, cn, s_mx);
    const float *gv = s_mx.gv;
    const long long *gg = s_mx.gg;
    const int *gok = s_mx.gok;
    EK_MST(10);     // last workgroup: distances into LDS, values, global maxima
    if (tid == 0) {
        // the decision: ek_chain_walk / ek_round_chain_kernel on the global maxima
        // (everything it reads, before anything it writes: one trip to memory)
        EkCtl c = *r.ctl;
        const int l0 = r.plan->label;
        const long long g0 = r.plan->gidx[0];
        const float m0 = r.plan->maxdist[0];
        uint32_t used = r.plan->used;
        const int run_err = ms->err;
        if (mode == 1) {
            // candidate 0 of the pass that has just run is a center now
            // (kcenters.py:306-309): the count and the history move when its
            // distances are in, not when it was planned -- the host stops
            // enqueuing rounds when the count reaches its goal
            r.hist[l0].gidx = g0;
            r.hist[l0].dist = m0;
            r.hist[l0].set = 1;
            c.n_done = l0 + 1;
            c.n_rounds = c.n_rounds + 1;
        }
        const int label0 = c.n_done;
        int na = 0;
        for (int k = 0; k < cn; ++k) {
            if (c.stopped || c.n_done >= c.limit || !gok[k])
                break;
            c.last_max = gv[k];
            if (!((double)gv[k] > r.cutoff)) {      // kcenters.py:217
                c.stopped = 1;
                break;
            }
            const int j = s_mx.cj[k];
            if (gg[k] != s_mx.cg[k])                // the farthest point is not stored
                break;
            const int label = c.n_done;
            r.hist[label].gidx = gg[k];
            r.hist[label].dist = gv[k];
            r.hist[label].set = 1;
            c.n_done = label + 1;
            used |= 1u << j;
            r.pend->slot[na] = j - 1;
            ++na;
        }
        // distances.max() after the last update (kcenters.py:226), if the chain
        // ended early: the maximum of the state it left
        if (na < cn && gok[na])
            c.last_max = gv[na];
        s_short = na < cn ? 1 : 0;
        if (mode == 1) {
            r.plan->used = used;
            r.plan->napply = na;
            r.plan->chain_label0 = label0;
            r.pend->n = na;
            r.pend->label0 = label0;
        }
        r.ctl->n_done = c.n_done;
        r.ctl->n_rounds = c.n_rounds;
        r.ctl->stopped = c.stopped;
        r.ctl->last_max = c.last_max;
        // (a message that never came: the run ends here, on every later launch too)
        s_over = (c.stopped || c.n_done >= c.limit || run_err) ? 1 : 0;
        // the offers describe the state this chain left on every shard (headers first:
        // ek_ms_chain_kernel; or the whole chain held), or they are offered again
        bool agree = true;
        for (int rk = 0; rk < x.world; ++rk)
            agree = agree && s_state[rk] == na;
        s_repick = (!s_over && mode == 1 && !agree) ? 1 : 0;
        if (s_repick) {
            ms->pick_state = na;
            ms->n_reoffer = ms->n_reoffer + 1u;
        }
    }
    __syncthreads();
    EK_MST(11);     // the decision
    const bool over = s_over != 0, repick = s_repick != 0;
    // ---- the next round's candidates among the records on offer ----------------------
    // (greedy, as ek_round_next_kernel: the record with the largest remaining
    // distance, then every other one's is lowered by its distance to it; slot order
    // = (shard, local rank) breaks ties like the lowest global index does)
    if (tid < EK_WAVE) {
        const int lane = tid;
        bool open = !repick && sgidx[lane] >= 0;
        float cur = open ? sval[lane] : 0.f;
        int ns = 0;
        while (ns < T) {
            // (largest remaining distance, lowest slot on ties -- ek_better's order --: a
            // maximum over the values' ordered bits, then the first lane that holds it)
            if (__ballot(open) == 0ull)
                break;
            const unsigned int key = open ? ek_ord_bits(cur) : 0u;
            unsigned int mx = key;
            mx = max(mx, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mx, 0xB1, 0xf, 0xf, false));
            mx = max(mx, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mx, 0x4E, 0xf, 0xf, false));
            mx = max(mx, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mx, 0x141, 0xf, 0xf, false));
            mx = max(mx, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mx, 0x140, 0xf, 0xf, false));
            const unsigned int top =
                max(max((unsigned int)__builtin_amdgcn_readlane((int)mx, 0),
                        (unsigned int)__builtin_amdgcn_readlane((int)mx, 16)),
                    max((unsigned int)__builtin_amdgcn_readlane((int)mx, 32),
                        (unsigned int)__builtin_amdgcn_readlane((int)mx, 48)));
            const int best = __ffsll((unsigned long long)__ballot(open && key == top)) - 1;
            if (lane == best)
                open = false;
            if (lane == 0)
                sel[ns] = best;
            ++ns;
            const float d = sD[best * 64 + lane];
            if (open && d < cur)
                cur = d;
        }
        if (lane == 0)
            n_sel = ns;
    }
    __syncthreads();
    EK_MST(12);     // the greedy choice
    const int ns = n_sel;
    const float first_max = ns > 0 ? sval[sel[0]] : -__builtin_inff();
    const bool go = !over && !repick && ns > 0 && (double)first_max > r.cutoff;
    const size_t rstride = ek_rec_bytes(A);
    // the chosen records (kept for the other entry points: [0] = the farthest
    // point of the state) and the candidate tile of the next pass
    if (!repick) {
        const int A3 = 3 * A;
        for (int e = tid; e < T * 8; e += EK_BLOCK) {
            const int c = e / 8, u = e % 8;
            uint32_t *h = (uint32_t *)(r.recs + (size_t)c * rstride);
            if (c < ns)
                h[u] = ek_msg_load<SYS>(ek_ms_rec<SYS>(x, smap[sel[c]], seq, A) + u);
            else
                h[u] = u == 0 ? __float_as_uint(-__builtin_inff())
                              : ((u == 2 || u == 3) ? 0xffffffffu : 0u);
        }
        // (T >= 16: the coordinates are ek_ms_ctile16_kernel's, a launch of its own --
        // written by this one workgroup they were most of the kernel's tail)
        const int cpr_c = (A3 + 3) / 4;         // 16-byte chunks of coordinates
        for (int e0 = tid; T < 16 && e0 < T * cpr_c; e0 += 4 * EK_BLOCK) {
            const void *p[4];
            ek_f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * EK_BLOCK, T * cpr_c - 1);
                const int c = e / cpr_c, q = e % cpr_c;
                p[u] = (const unsigned char *)(ek_ms_rec<SYS>(x, smap[sel[c < ns ? c : 0]],
                                                              seq, A) + 8) + 16 * q;
            }
            ek_msg_load4<SYS, 4>(p, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * EK_BLOCK;
                if (e >= T * cpr_c)
                    continue;
                const int c = e / cpr_c, q = e % cpr_c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * q + k;
                    if (row >= A3)
                        continue;
                    const float val = c < ns ? v[u][k] : 0.f;
                    if (c < ns)
                        ((float *)(r.recs + (size_t)c * rstride + sizeof(EkRecHdr)))[row] =
                            val;
                    if (go)     // (zero for a slot without a candidate)
                        r.ctile[ek_ctile_index(T, row / 3, c, row % 3)] = val;
                }
            }
        }
        if (go && T < 16)                           // the atoms of padding
            for (int k = tid; k < (ek_ctile_atoms(A) - A) * 3 * T; k += EK_BLOCK)
                r.ctile[ek_ctile_index(T, A + k / (3 * T), (k % (3 * T)) / 3, k % 3)] =
                    0.f;
        if (tid < T) {
            EkPlan *plan = r.plan;
            double tr = 0.0;
            plan->offer[tid] = tid < ns ? smap[sel[tid]] : 0;    // (the record's slot)
            if (tid < ns) {
                const uint32_t *rr = ek_ms_rec<SYS>(x, smap[sel[tid]], seq, A);
                tr = __longlong_as_double(
                    (long long)((unsigned long long)ek_msg_load<SYS>(rr + 4) |
                                ((unsigned long long)ek_msg_load<SYS>(rr + 5) << 32)));
                plan->src[tid] = tid;
                plan->gidx[tid] = sgidx[sel[tid]];
                plan->maxdist[tid] = sval[sel[tid]];
                plan->trace[tid] = tr;
            }
            r.ctrace[tid] = tr;
        }
        if (tid >= 1 && tid < EK_MAX_CANDS)     // the presumed order: as chosen
            r.ord->cand[tid - 1] = tid;
    }
    if (tid == 0) {
        EkPlan *plan = r.plan;
        plan->n_rec = repick ? 0 : ns;
        plan->apply = -1;
        plan->chain_n = 0;
        if (!repick && !s_short && ns > 0)  // (the offers describe the state as it is)
            r.ctl->last_max = first_max;
        if (go) {
            // (candidate 0 is counted by the plan kernel that follows its pass)
            plan->go = 1;
            plan->teff = ns;
            plan->label = r.ctl->n_done;
            plan->used = 1;
            plan->miss = 0;
            r.ord->n = ns - 1;
            ms->mode = 1;
        } else {
            plan->go = 0;
            plan->teff = 0;
            plan->miss = 1;
            if (repick) {
                ms->mode = 2;
            } else {
                if (!over && ns > 0)
                    r.ctl->stopped = 1;     // maxdist <= cutoff
                plan->used = 0;
                ms->mode = 0;
            }
        }
        ms->seq = seq + 1u;
    }
    EK_MST(13);     // records, plan
}

// The chosen records' coordinates of a round of 16 or 32, out of the mailboxes: into
// the round's records and, in 16-byte pieces, into the candidate tile(s) of the
// next pass (as ek_round_ctile16_kernel does for a single shard).  It reads the
// exchange the plan kernel has just closed (ms->seq - 1); after a plan kernel that
// had nothing to do it writes the same bytes again.
template <bool SYS>
__global__ void __launch_bounds__(EK_BLOCK)
ek_ms_ctile16_kernel(EkRound r, EkMsState *ms, EkMsXchg x, int halves)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    __shared__ int soff[EK_MAX_CANDS];
    const EkPlan *plan = r.plan;
    const int tid = threadIdx.x;
    const int ns = plan->n_rec, go = plan->go;
    if (ns <= 0)
        return;
    const uint32_t seq = ms->seq - 1u;
    if (tid < EK_MAX_CANDS)
        soff[tid] = tid < ns ? plan->offer[tid] : 0;
    __syncthreads();
    const int A = r.A, A3 = 3 * A;
    const int n_ct = ek_ctile_atoms(A) / 16 * 3;        // (16 atoms, axis) blocks of 1 KB
    const int ct_wgs = (n_ct + 3) / 4;
    if ((int)blockIdx.x < halves * ct_wgs) {
        const int half = blockIdx.x / ct_wgs;
        const int blk = (blockIdx.x % ct_wgs) * 4 + (tid >> 6), lane = tid & 63;
        if (!go || blk >= n_ct || 16 * half >= ns)
            return;
        const int S = blk / 3, k = blk % 3, kk = lane >> 4, c = 16 * half + (lane & 15);
        const float *src = (const float *)(ek_ms_rec<SYS>(x, soff[c], seq, A) + 8) + k;
        v4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = 16 * S + 4 * q + kk;
            v[q] = (a < A && c < ns) ? ek_msg_loadf<SYS>(src + 3 * a) : 0.f;
        }
        *(v4 *)(r.ctile + half * ek_ctile_half_floats(A) +
                ek_ctile_index(16, 16 * S + kk, c, k)) = v;
    } else {
        const int c = blockIdx.x - halves * ct_wgs;     // one record per workgroup
        if (c >= ns)
            return;
        const float *src = (const float *)(ek_ms_rec<SYS>(x, soff[c], seq, A) + 8);
        float *rec = (float *)(r.recs + (size_t)c * ek_rec_bytes(A) + sizeof(EkRecHdr));
        for (int row = tid; row < A3; row += EK_BLOCK)
            rec[row] = ek_msg_loadf<SYS>(src + row);
    }
}

void ek_launch_ms_plan(const EkRound &r, EkMsState *ms, const EkMsXchg &x, float *D,
                       hipStream_t s)
{
    const unsigned blocks = (unsigned)(64 * 64 / (EK_BLOCK / EK_WAVE));
#define EK_MS_PLAN(TT)                                                         \
    do {                                                                       \
        if (x.sys)                                                             \
            hipLaunchKernelGGL((ek_ms_plan_kernel<TT, true>), dim3(blocks),    \
                               dim3(EK_BLOCK), 0, s, r, ms, x, D);             \
        else                                                                   \
            hipLaunchKernelGGL((ek_ms_plan_kernel<TT, false>), dim3(blocks),   \
                               dim3(EK_BLOCK), 0, s, r, ms, x, D);             \
    } while (0)
    if (r.T >= 16) {
        const int halves = r.T / 16;
        if (r.T == 32)
            EK_MS_PLAN(32);
        else
            EK_MS_PLAN(16);
        const unsigned wgs =
            (unsigned)(halves * ((ek_ctile_atoms(r.A) / 16 * 3 + 3) / 4) + r.T);
        if (x.sys)
            hipLaunchKernelGGL((ek_ms_ctile16_kernel<true>), dim3(wgs), dim3(EK_BLOCK), 0, s,
                               r, ms, x, halves);
        else
            hipLaunchKernelGGL((ek_ms_ctile16_kernel<false>), dim3(wgs), dim3(EK_BLOCK), 0, s,
                               r, ms, x, halves);
    } else if (r.T == 8)
        EK_MS_PLAN(8);
    else
        EK_MS_PLAN(4);
#undef EK_MS_PLAN
}
