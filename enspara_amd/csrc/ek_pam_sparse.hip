// ek_pam_sparse.hip -- a window of PAM proposals worked through by ONE workgroup.
//
// Same arithmetic and same outcome as the three launches per proposal of
// ek_pam.hip ("a window of proposals decided on the device"; reference
// enspara/cluster/kmedoids.py:610-690), for windows whose prefetch was
// restricted to the frames a proposal can touch (ek_pam.hip, "proposal prefetch
// restricted ..."): every frame outside that list keeps its label and distance
// whatever the window does, so a proposal is
//   * the classification (kmedoids.py:644-658) of ITS frames only -- the members
//     of its cluster and the listed frames closer to it than to their medoid at
//     the time the window was opened (ek_sp_bucket_kernel; a distance only grows
//     for the members an accepted earlier slot's proposal is farther from, and
//     the list allows for that);
//   * the search of kmedoids.py:666 for the members the proposal is farther
//     from, over the medoids within their reach (the window's tables, as in
//     ek_pam_classify_window_kernel's last workgroup): 64 members x 8 medoids
//     per step through LDS, every pair one lane's IEEE FMA chains in ascending
//     atom order and ek_rmsd_from_S, as everywhere else;
//   * the two cost sums in numpy's order (ek_pam.hip, "cost sums in numpy's
//     order") -- kept as a tree: leaf sums and chunk sums of the current state
//     sit in memory, a proposal writes its changes into the state, re-adds the
//     leaves and chunks it touched, adds the chunks up left to right, decides
//     (mean of squares, float64, strict <: kmedoids.py:478-479, :683) and either
//     keeps all that or puts the old values back.  That is: where the sums decide.
//     Where the sum of new^2 - old^2 over the frames the proposal changes is far
//     beyond what the rounding of the two big sums can amount to, its sign is the
//     outcome and the sums are not taken (see "what the proposal would change").
// No launch, no arrival counter and no other workgroup between two proposals:
// the steps are separated by workgroup barriers.  A dependent launch costs
// ~4.5 us on this GPU and the three-launch form spends ~29 us per proposal at
// 10^6 frames, two thirds of it launch gaps and single-workgroup tails.
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_pam_sparse.h"
#include "ek_lanes.h"
#include <algorithm>

#define SP_NT EK_SP_THREADS
#ifdef EK_SP_PROF
// (kept in registers and written once at the end: a read-modify-write of memory
// per mark cost more than most steps)
#define SP_T(k)                                                                \
    do {                                                                       \
        const unsigned long long now = wall_clock64();                         \
        sp_acc[k] += now - sp_t0;                                              \
        sp_t0 = now;                                                           \
    } while (0)
#else
#define SP_T(k)
#endif
#define SP_WAVES (SP_NT / EK_WAVE)
#define SP_CH 64                            // atoms per LDS slice of the search
#define SP_LD (EK_WAVE + 1)

// ---- per slot: the frames it has to look at --------------------------------------------
// list position p, slot j: frame f = list[p] goes into bucket j if it is a member
// of cluster cid0 + j or closer to proposal j than to its medoid (the test of
// kmedoids.py:644) -- on the state the window opens with, which is enough: until
// slot j's turn a frame's distance only shrinks, except for the members an
// accepted proposal i < j is farther from (see `grown` below)
// One wave per 64 listed frames and group of SP_BK_G slots: the gathers out of the
// distance vectors (a hundred-odd megabytes apart) are what this kernel waits for,
// and the more waves on the more compute units share them the better (1024-thread
// workgroups: 72 us; 256: 25 us; this: 8 us per window at 10^6 frames).
#define SP_BK_G 8
__global__ void __launch_bounds__(EK_WAVE)
ek_sp_bucket_kernel(const uint32_t *__restrict__ list, int64_t n_act,
                    const float *__restrict__ dist,
                    const int32_t *__restrict__ assign,
                    const float *__restrict__ vecs, int64_t n_pad, int32_t cid0,
                    int count, uint2 *__restrict__ bucket,
                    unsigned int *__restrict__ bcnt, int64_t bcap)
{
    const int lane = threadIdx.x;
    const int j0 = blockIdx.y * SP_BK_G;
    const int64_t p = (int64_t)blockIdx.x * EK_WAVE + lane;
    const bool ok = p < n_act;
    const uint32_t f = ok ? list[p] : 0u;
    const float d = ok ? dist[f] : 0.f;
    const int32_t a = ok ? assign[f] : -1;
    float nd[SP_BK_G];                          // all gathers in flight at once
#pragma unroll
    for (int u = 0; u < SP_BK_G; ++u)
        nd[u] = (ok && j0 + u < count) ? vecs[(size_t)(j0 + u) * n_pad + f]
                                       : __builtin_inff();
    // A member of the window's cluster i may be left farther from its medoid than
    // it is now when proposal i is accepted -- but not farther than from proposal
    // i itself, which is a medoid then: for the slots after i that distance is
    // the bound the test is made with.
    const int i = a - cid0;
    float grown = d;
    if (ok && i >= 0 && i < count && i < j0 + SP_BK_G - 1)
        grown = fmaxf(d, vecs[(size_t)i * n_pad + f]);
    uint32_t hit = 0;
#pragma unroll
    for (int u = 0; u < SP_BK_G; ++u) {
        const int j = j0 + u;
        if (ok && j < count && (a == cid0 + j || (j > i ? grown : d) > nd[u]))
            hit |= 1u << u;
    }
    // lane u asks for the room of the wave's entries in bucket j0 + u
    unsigned int mine = 0;
#pragma unroll
    for (int u = 0; u < SP_BK_G; ++u) {
        const unsigned long long m = __ballot((hit >> u) & 1u);
        if (lane == u)
            mine = (unsigned int)__popcll(m);
    }
    const unsigned int base = mine ? atomicAdd(&bcnt[j0 + lane], mine) : 0u;
#pragma unroll
    for (int u = 0; u < SP_BK_G; ++u) {
        const bool h = (hit >> u) & 1u;
        const unsigned long long m = __ballot(h);
        const unsigned int b = __shfl(base, u, EK_WAVE);
        if (h) {
            const unsigned int pos = b + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            // with its distance to the proposal: the window's workgroup would
            // wait a cold gather for it
            if ((int64_t)pos < bcap)
                bucket[(size_t)(j0 + u) * bcap + pos] = make_uint2(f, __float_as_uint(nd[u]));
        }
    }
}

void ek_launch_sp_bucket(const uint32_t *list, int64_t n_act, const float *dist,
                         const int32_t *assign, const float *vecs, int64_t n_pad,
                         int32_t cid0, int count, uint2 *bucket,
                         unsigned int *bcnt, int64_t bcap, hipStream_t s, bool cleared)
{
    if (!cleared)       // (ek_sp_finish_kernel of the window before leaves them at zero)
        (void)hipMemsetAsync(bcnt, 0, EK_PAM_WIN * sizeof(unsigned int), s);
    if (n_act <= 0)
        return;
    hipLaunchKernelGGL(ek_sp_bucket_kernel,
                       dim3((unsigned)((n_act + EK_WAVE - 1) / EK_WAVE),
                            (count + SP_BK_G - 1) / SP_BK_G),
                       dim3(EK_WAVE), 0, s, list, n_act, dist, assign, vecs, n_pad,
                       cid0, count, bucket, bcnt, bcap);
}

// ---- numpy's leaf: eight interleaved running sums, combined pairwise, then the tail ------
// called by eight consecutive lanes (l8 = lane & 7); the sum is lane l8 == 0's
__device__ __forceinline__ double ek_sp_leaf(const float *__restrict__ d, int64_t off,
                                             int len, int l8)
{
    double r = 0.0;
    const int body = (len < 8) ? 0 : len - (len % 8);
    if (len == 128) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = d[off + 8 * i + l8];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double x = v[i];
            r = (i == 0) ? x * x : r + x * x;
        }
    } else {
        for (int i = 0; i < body; i += 8) {
            const double x = d[off + i + l8];
            r = (i == 0) ? x * x : r + x * x;
        }
    }
    if (body > 0) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1)
            r = r + __shfl_xor(r, o, 8);
    }
    if (l8 == 0)
        for (int i = body; i < len; ++i) {
            const double x = d[off + i];
            r = r + x * x;
        }
    return r;
}

// the leaf a frame belongs to
__device__ __forceinline__ int ek_sp_leaf_of(int64_t f, const EkSpArgs &p)
{
    const int64_t full = (int64_t)p.n_full * EK_PW_CHUNK;
    if (f < full)
        return (int)(f / 128);
    const EkPwShape *sh = &p.shapes[1];
    const int rel = (int)(f - full);
    int lo = 0, hi = sh->n_leaves - 1;          // last leaf starting at or before rel
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (sh->leaf_off[mid] <= rel)
            lo = mid;
        else
            hi = mid - 1;
    }
    return p.n_full * EK_PW_FULL_LEAVES + lo;
}

// room in a list for the lanes of a wave that want to append (all lanes of the wave
// call this): one LDS atomic per wave instead of one per lane -- same-address
// atomics are served one after the other.  -> this lane's position (if `want`)
__device__ __forceinline__ unsigned int ek_sp_append(unsigned int *counter, bool want, int lane)
{
    const unsigned long long m = __ballot(want);
    if (m == 0)
        return 0u;
    const int leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (lane == leader)
        base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = __shfl(base, leader, EK_WAVE);
    return base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ unsigned long long ek_sp_key(float d, int32_t c)
{
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)c;
}

// the chunks left to right (numpy adds its buffer's partial sums in order).  The
// chunk sums of the state live in LDS for the whole window (padded with +0.0 to a
// multiple of 16: the sum of squares is >= +0 and stays what it is); every wave
// adds them up for itself, all lanes alike -- broadcast reads, sixteen in flight,
// the chain of additions is the only thing that takes time -- so nothing has to
// be handed from one wave to the others.
__device__ __forceinline__ double ek_sp_total(const double *s_chunk, int n_chunks)
{
    double sum = 0.0;
    for (int c0 = 0; c0 < n_chunks; c0 += 16) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            x[j] = s_chunk[c0 + j];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            sum = sum + x[j];
    }
    return sum;
}

// the last, shorter chunk's own tree over its leaves (whole workgroup) -> *out
__device__ void ek_sp_ragged(const EkSpArgs &p, double *la, double *out)
{
    const EkPwShape *sh = &p.shapes[1];
    const int t = threadIdx.x, nl = sh->n_leaves;
    const size_t g0 = (size_t)p.n_full * EK_PW_FULL_LEAVES;
    for (int i = t; i < nl; i += SP_NT)
        la[i] = p.leaf[2 * (g0 + i)];
    __syncthreads();
    for (int lev = 0; lev < sh->n_levels; ++lev) {
        const int k0 = sh->level_start[lev], k1 = sh->level_start[lev + 1];
        for (int k = k0 + t; k < k1; k += SP_NT)
            la[nl + k] = la[sh->node_l[k]] + la[sh->node_r[k]];
        __syncthreads();
    }
    if (t == 0)
        *out = la[(sh->n_nodes > 0) ? nl + sh->n_nodes - 1 : 0];
    __syncthreads();
}

#define SP_LB 6         // leaves a group of eight lanes re-adds with its loads in flight
#define SP_CB 8         // pairs of chunks a wave does with its loads in flight

// the listed leaves' sums, re-added from the state (numpy's leaf: eight
// interleaved running sums, combined pairwise, then the tail); the value before
// goes to p.leaf[2 g + 1], the leaf's chunk is marked in `cbits`
__device__ __forceinline__ void ek_sp_readd_leaves(const EkSpArgs &p, const int32_t *tleaf,
                                                   unsigned int n_leaf, uint32_t *cbits)
{
    const int t = threadIdx.x;
    const int full_leaves = p.n_full * EK_PW_FULL_LEAVES;
    for (unsigned int q0 = 0; q0 < n_leaf; q0 += SP_LB * (SP_NT / 8)) {
        int g[SP_LB];
        double old[SP_LB];
        float v[SP_LB][16];
#pragma unroll
        for (int b = 0; b < SP_LB; ++b) {
            const unsigned int q = q0 + b * (SP_NT / 8) + t / 8;
            g[b] = (q < n_leaf) ? tleaf[q] : -1;        // (uniform over a group of 8 lanes)
        }
#pragma unroll
        for (int b = 0; b < SP_LB; ++b) {
            old[b] = (g[b] >= 0) ? p.leaf[2 * (size_t)g[b]] : 0.0;
            if (g[b] >= 0 && g[b] < full_leaves) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    v[b][i] = p.dist[(int64_t)g[b] * 128 + 8 * i + (t & 7)];
            }
        }
#pragma unroll
        for (int b = 0; b < SP_LB; ++b) {
            if (g[b] < 0)
                continue;
            double r;
            int chunk;
            if (g[b] < full_leaves) {
                r = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double x = v[b][i];
                    r = (i == 0) ? x * x : r + x * x;
                }
                r = ek_tree_sum8(r);
                chunk = g[b] / EK_PW_FULL_LEAVES;
            } else {
                const EkPwShape *sh = &p.shapes[1];
                const int l = g[b] - full_leaves;
                r = ek_sp_leaf(p.dist, (int64_t)p.n_full * EK_PW_CHUNK + sh->leaf_off[l],
                               sh->leaf_len[l], t & 7);
                chunk = p.n_full;
            }
            if ((t & 7) == 0) {
                p.leaf[2 * (size_t)g[b] + 1] = old[b];
                p.leaf[2 * (size_t)g[b]] = r;
                atomicOr(&cbits[chunk >> 5], 1u << (chunk & 31));
            }
        }
    }
}

// the marked chunks' sums, re-added from their leaf sums (after a barrier that
// waits for the stores above); the value before goes to s_chunk_old.  A full
// chunk: 64 leaves, a perfect in-order tree (checked by ek_pam_alloc) -- half a
// wave per chunk: a lane adds leaves 2l and 2l + 1 itself, ek_tree_sum32 does the
// five levels above.  -> whether the last, shorter chunk was among them
__device__ __forceinline__ bool ek_sp_readd_chunks(const EkSpArgs &p, const uint32_t *cbits,
                                                   double *s_chunk, double *s_chunk_old,
                                                   double *la, int lane, int wv)
{
    for (int c0 = 0; c0 < p.n_full; c0 += 2 * SP_CB * SP_WAVES) {
        double r[SP_CB];
        bool on[SP_CB];
        const int half = lane >> 5, l32 = lane & 31;
#pragma unroll
        for (int k = 0; k < SP_CB; ++k) {
            const int c = c0 + 2 * (k * SP_WAVES + wv) + half;
            on[k] = c < p.n_full && ((cbits[c >> 5] >> (c & 31)) & 1u);
            const size_t g = (size_t)c * EK_PW_FULL_LEAVES + 2 * l32;
            const double a0 = on[k] ? p.leaf[2 * g] : 0.0;
            const double a1 = on[k] ? p.leaf[2 * g + 2] : 0.0;
            r[k] = a0 + a1;
        }
#pragma unroll
        for (int k = 0; k < SP_CB; ++k)
            r[k] = ek_tree_sum32(r[k]);
#pragma unroll
        for (int k = 0; k < SP_CB; ++k) {
            const int c = c0 + 2 * (k * SP_WAVES + wv) + half;
            if (on[k] && l32 == 0) {
                s_chunk_old[c] = s_chunk[c];
                s_chunk[c] = r[k];
            }
        }
    }
    const bool ragged = p.n_chunks > p.n_full &&
                        ((cbits[p.n_full >> 5] >> (p.n_full & 31)) & 1u);
    if (ragged) {
        if (threadIdx.x == 0)
            s_chunk_old[p.n_full] = s_chunk[p.n_full];
        ek_sp_ragged(p, la, &s_chunk[p.n_full]);
    }
    return ragged;
}

// the lists of a proposal are kept in LDS: a dependent trip to memory costs about
// a microsecond, and that -- not arithmetic -- is what a proposal's time is made of
#define SP_KU 10        // table entries a thread asks for ahead of the search
#define SP_CAP_CHG EK_SP_CAP_CHG // frames a proposal may change
#define SP_CAP_AMB 1024 // members it may leave behind
#define SP_CAP_COLS 256 // medoids within their reach

struct EkSpLds {        // byte offsets into the dynamic LDS block
    static constexpr size_t tile = 0;
    static constexpr size_t ytile = tile + (size_t)3 * SP_CH * SP_LD * 4;
    static constexpr size_t chunk = ytile + (size_t)SP_WAVES * 3 * SP_CH * 4;
    static constexpr size_t chunk_old = chunk + (size_t)EK_SP_MAX_CHUNKS * 8;
    static constexpr size_t amb_key = chunk_old + (size_t)EK_SP_MAX_CHUNKS * 8;
    static constexpr size_t lbits = amb_key + (size_t)SP_CAP_AMB * 8;
    static constexpr size_t dirty = lbits + (size_t)EK_SP_MAX_CHUNKS * EK_PW_FULL_LEAVES / 8;
    static constexpr size_t chg_f = dirty + (size_t)EK_SP_MAX_CHUNKS * EK_PW_FULL_LEAVES / 8;
    static constexpr size_t chg_od = chg_f + (size_t)SP_CAP_CHG * 4;
    static constexpr size_t chg_nd = chg_od + (size_t)SP_CAP_CHG * 4;
    static constexpr size_t chg_oa = chg_nd + (size_t)SP_CAP_CHG * 4;
    static constexpr size_t chg_na = chg_oa + (size_t)SP_CAP_CHG * 4;
    static constexpr size_t amb_f = chg_na + (size_t)SP_CAP_CHG * 4;
    static constexpr size_t amb_d = amb_f + (size_t)SP_CAP_AMB * 4;
    static constexpr size_t tleaf = amb_d + (size_t)SP_CAP_AMB * 4;
    static constexpr size_t cols = tleaf + (size_t)SP_CAP_CHG * 4;
    static constexpr size_t end = cols + (size_t)SP_CAP_COLS * 4;
};
static_assert(EkSpLds::chunk % 8 == 0 && EkSpLds::amb_key % 8 == 0, "doubles");
static_assert(EkSpLds::end + 4096 <= 160 * 1024, "one workgroup's LDS");
static_assert((size_t)3 * SP_CH * SP_LD * 4 >= 2 * EK_PW_MAX_LEAVES * 8,
              "the last chunk's tree shares the search's tile");

// the lists and counters of the slot being worked on (all in LDS)
struct EkSpShared {
    float *tile, *ytile;
    unsigned long long *amb_key;
    uint32_t *chg_f;
    float *chg_od, *chg_nd;
    int32_t *chg_oa, *chg_na;
    uint32_t *amb_f;
    float *amb_d;
    int32_t *cols;
    unsigned long long *best;   // [EK_WAVE]
    uint32_t *rowf;             // [EK_WAVE]
    unsigned int *n_chg, *n_amb, *reach, *n_col;
    float *T;                   // [EK_PAM_WIN] accepted earlier slots' proposals to this slot's medoid
    const int *acc;             // [EK_PAM_WIN] slots accepted so far
    const unsigned int *bcnt;   // [EK_PAM_WIN + 1] bucket lengths
};

// ---- a slot's evaluation: classification, medoids within reach, search ----------------------
// Leaves the frames the proposal would change in L.chg_* (*L.n_chg of them) and
// the members it is farther from in L.amb_*; -> 0: complete, 1: more than one
// workgroup should take on, 2: more members stay put than were declared.
// SPEC: on the state the window opens with, no earlier slot taken as accepted
// (ek_sp_spec_kernel).  have_first: f_first is this thread's first bucket entry
// (asked for while the slot before was worked on); *f_next, if given, gets the
// next slot's.
template <bool SPEC>
__device__ __forceinline__ int ek_sp_evaluate(const EkSpArgs &p, const EkSpShared &L, int slot,
                                              bool have_first, uint2 f_first, uint2 *f_next,
                                              unsigned int *n_amb_out)
{
    const int t = threadIdx.x, lane = t & (EK_WAVE - 1),
              wv = __builtin_amdgcn_readfirstlane(t / EK_WAVE);
    const int A = p.A, K = p.K;
    const int32_t cid = p.cid0 + slot;
    // ---- classification (kmedoids.py:644-658) of this slot's frames ------------------
    const unsigned int nb = L.bcnt[slot];
    if (!have_first && (unsigned int)t < nb)
        f_first = p.bucket[(size_t)slot * p.bcap + t];
    // (what the search may need of the tables, see below: asked for now, there
    // when the classification is through)
    // (loads come back in the order they were asked for: the state of this
    // thread's first frame goes before the tables and the next slot's frames)
    float d_first = 0.f;
    int32_t a_first = 0;
    if ((unsigned int)t < nb) {
        d_first = p.dist[f_first.x];
        a_first = p.assign[f_first.x];
    }
    float t_reg = 0.f;      // (into LDS after the classification: no wait for it here)
    if (!SPEC && t < slot && L.acc[t])
        t_reg = p.T[(size_t)t * K + cid];
    float Dtab[SP_KU];
    {
        const float *O = p.O + (size_t)slot * K;
#pragma unroll
        for (int u = 0; u < SP_KU; ++u) {
            const int c = t + u * SP_NT;
            Dtab[u] = (c < K) ? O[c] : 0.f;
        }
    }
    if (f_next && (unsigned int)t < L.bcnt[slot + 1])
        *f_next = p.bucket[(size_t)(slot + 1) * p.bcap + t];
    for (unsigned int e0 = 0; e0 < nb; e0 += SP_NT) {      // (uniform over the wave)
        const unsigned int e = e0 + t;
        const bool in = e < nb;
        uint2 fe = f_first;
        if (in && e0 > 0)
            fe = p.bucket[(size_t)slot * p.bcap + e];
        const uint32_t f = fe.x;
        const float nd = __uint_as_float(fe.y);
        const float d = (e0 == 0) ? d_first : (in ? p.dist[f] : 0.f);
        const int32_t a = (e0 == 0) ? a_first : (in ? p.assign[f] : 0);
        const bool closer = in && d > nd;
        const bool member = in && !closer && a == cid;
        const unsigned int q1 = ek_sp_append(L.n_chg, closer, lane);
        if (closer && q1 < SP_CAP_CHG) {
            L.chg_f[q1] = f;
            L.chg_od[q1] = d;
            L.chg_oa[q1] = a;
            L.chg_nd[q1] = nd;
            L.chg_na[q1] = cid;
        }
        const unsigned int q2 = ek_sp_append(L.n_amb, member, lane);
        if (member) {
            if (q2 < SP_CAP_AMB) {
                L.amb_f[q2] = f;
                L.amb_d[q2] = d;
                L.amb_key[q2] = ek_sp_key(nd, cid);
            }
            // how far a medoid may be from the old one and still matter to
            // this frame; non-negative floats order like their bits
            atomicMax(L.reach, __float_as_uint(d + nd));
        }
    }
    if (!SPEC && t < slot && L.acc[t])
        L.T[t] = t_reg;
    ek_lds_barrier();
    const unsigned int n_amb = *L.n_amb;
    *n_amb_out = n_amb;
    const bool too_many = (int64_t)n_amb > p.max_amb[slot];
    bool bail = n_amb > SP_CAP_AMB;
    if (n_amb > 0 && !too_many && !bail) {
        // ---- the medoids within reach of those members (ek_pam_prune_kernel's
        // test, from the window's tables) -------------------------------------------
        const float lim = __uint_as_float(*L.reach) * 1.001f + 1e-3f;
        const float *O = p.O + (size_t)slot * K;
        for (int c0 = t; c0 < K; c0 += SP_KU * SP_NT) {
            float D[SP_KU];                 // the table reads in flight
#pragma unroll
            for (int u = 0; u < SP_KU; ++u) {
                const int c = c0 + u * SP_NT;
                D[u] = (c0 == t) ? Dtab[u] : ((c < K) ? O[c] : 0.f);
            }
#pragma unroll
            for (int u = 0; u < SP_KU; ++u) {
                const int c = c0 + u * SP_NT;
                if (c >= K)
                    continue;
                const int i = c - p.cid0;
                if (!SPEC && i >= 0 && i < slot && L.acc[i])
                    D[u] = L.T[i];  // its proposal sits there now: T[i][cid]
                if (c != cid && !(D[u] > lim)) {
                    const unsigned int q = atomicAdd(L.n_col, 1u);
                    if (q < SP_CAP_COLS)
                        L.cols[q] = c;
                }
            }
        }
        ek_lds_barrier();
        const unsigned int n_col = *L.n_col;
        bail = n_col > SP_CAP_COLS || (uint64_t)n_amb * n_col > (uint64_t)p.max_pairs;
        // ---- kmedoids.py:666 over them: rows = members, columns = medoids ------------
        for (unsigned int r0 = 0; !bail && n_col > 0 && r0 < n_amb; r0 += EK_WAVE) {
            if (t < EK_WAVE) {
                const bool ok = r0 + t < n_amb;
                L.rowf[t] = ok ? L.amb_f[r0 + t] : 0u;
                L.best[t] = ok ? L.amb_key[r0 + t] : ~0ull;
            }
            __syncthreads();
            const bool rok = r0 + lane < n_amb;
            constexpr int TPR = SP_NT / EK_WAVE;            // threads per row
            constexpr int NLD = 3 * SP_CH / TPR;
            const int lm = t / TPR, le = t % TPR;
            const bool lok = r0 + lm < n_amb;
            const float *lrow = p.frames_aos + (size_t)L.rowf[lm] * 3 * A;
            for (unsigned int k0 = 0; k0 < n_col; k0 += SP_WAVES) {
                const bool live = k0 + wv < n_col;
                const int32_t col = live ? L.cols[k0 + wv] : 0;
                // (a medoid accepted in this window: its row of the table is
                // written when the window is over)
                const int ci = col - p.cid0;
                const bool moved_in = !SPEC && live && ci >= 0 && ci < slot && L.acc[ci];
                const float *y = moved_in
                                     ? p.frames_aos + (size_t)p.frames[ci] * 3 * A
                                     : p.med_aos + (size_t)col * 3 * A;
                float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int a0 = 0; a0 < A; a0 += SP_CH) {
                    const int ch = (A - a0 < SP_CH) ? A - a0 : SP_CH;
                    float v[NLD], vy[3];
#pragma unroll
                    for (int k = 0; k < NLD; ++k) {
                        const int e = le + TPR * k;
                        v[k] = (3 * a0 + e < 3 * A && lok) ? lrow[3 * a0 + e] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int e = lane + EK_WAVE * k;
                        vy[k] = (live && 3 * a0 + e < 3 * A) ? y[3 * a0 + e] : 0.f;
                    }
                    __syncthreads();                // the slice before is done with
#pragma unroll
                    for (int k = 0; k < NLD; ++k)
                        L.tile[(le + TPR * k) * SP_LD + lm] = v[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        L.ytile[wv * 3 * SP_CH + lane + EK_WAVE * k] = vy[k];
                    __syncthreads();
                    if (live) {
                        const float *yt = L.ytile + wv * 3 * SP_CH;
#pragma unroll 8
                        for (int a = 0; a < ch; ++a) {
                            const float x0 = L.tile[(3 * a + 0) * SP_LD + lane],
                                        x1 = L.tile[(3 * a + 1) * SP_LD + lane],
                                        x2 = L.tile[(3 * a + 2) * SP_LD + lane];
                            const float y0 = yt[3 * a], y1 = yt[3 * a + 1],
                                        y2 = yt[3 * a + 2];
                            S[0] = fmaf(x0, y0, S[0]); S[1] = fmaf(x0, y1, S[1]);
                            S[2] = fmaf(x0, y2, S[2]); S[3] = fmaf(x1, y0, S[3]);
                            S[4] = fmaf(x1, y1, S[4]); S[5] = fmaf(x1, y2, S[5]);
                            S[6] = fmaf(x2, y0, S[6]); S[7] = fmaf(x2, y1, S[7]);
                            S[8] = fmaf(x2, y2, S[8]);
                        }
                    }
                }
                if (live && rok) {
                    const double Gy = moved_in ? p.G[p.frames[ci]] : p.med_G[col];
                    const float D = ek_rmsd_from_S(S, p.G[L.rowf[lane]], Gy, A);
                    // equal distances: the lowest medoid index, util.py:199-203's
                    // strict-< scan in ascending order
                    atomicMin(&L.best[lane], ek_sp_key(D, col));
                }
            }
            __syncthreads();
            if (t < EK_WAVE && r0 + t < n_amb)
                L.amb_key[r0 + t] = L.best[t];
            __syncthreads();
        }
        // the members' new labels and distances
        for (unsigned int r0 = 0; !bail && r0 < n_amb; r0 += SP_NT) {
            const unsigned int r = r0 + t;
            const bool in = r < n_amb;
            const unsigned long long key = in ? L.amb_key[r] : 0ull;
            const float d = in ? L.amb_d[r] : 0.f;
            const float ndv = __uint_as_float((unsigned int)(key >> 32));
            const int32_t na = (int32_t)(key & 0xffffffffu);
            const bool moved = in && (na != cid ||
                                      __float_as_uint(ndv) != __float_as_uint(d));
            const unsigned int q = ek_sp_append(L.n_chg, moved, lane);
            if (moved && q < SP_CAP_CHG) {
                L.chg_f[q] = L.amb_f[r];
                L.chg_od[q] = d;
                L.chg_oa[q] = cid;
                L.chg_nd[q] = ndv;
                L.chg_na[q] = na;
            }
        }
    }
    ek_lds_barrier();
    if (too_many)
        return 2;
    return (bail || *L.n_chg > SP_CAP_CHG) ? 1 : 0;
}

// ---- what the proposal would change in the sum of squares -----------------------------------
// Both sums are numpy's pairwise sums of ~n terms: each within (chunks + 30)
// eps of the exact value, eps = 1.1e-16 -- 1.2e-13 relative at the 1024 chunks
// a window may have.  The sum of (new^2 - old^2) over the frames the proposal
// changes is exact to 2048 eps of the sum of its terms' magnitudes at worst
// (each square of a float32 is exact in float64).  If it is four orders of
// magnitude beyond what the rounding of the two big sums can amount to, its
// sign IS the outcome of kmedoids.py:683's comparison and neither sum has to
// be taken; otherwise (a two-member cluster swapping its medoid changes
// nothing in exact arithmetic) both are, in numpy's order.
__device__ __forceinline__ void ek_sp_delta(const EkSpShared &L, unsigned int n_chg,
                                            double *s_part, double *delta_out, double *dab_out)
{
    const int t = threadIdx.x, lane = t & (EK_WAVE - 1),
              wv = __builtin_amdgcn_readfirstlane(t / EK_WAVE);
    double dsum = 0.0, dabs = 0.0;
    for (unsigned int q = t; q < n_chg; q += SP_NT) {
        const double od = L.chg_od[q], nd = L.chg_nd[q];
        const double term = nd * nd - od * od;
        dsum += term;
        dabs += fabs(term);
    }
    dsum = ek_tree_sum64(dsum);             // (any order will do)
    dabs = ek_tree_sum64(dabs);
    if (lane == 0) {
        s_part[wv] = dsum;
        s_part[SP_WAVES + wv] = dabs;
    }
    ek_lds_barrier();
    double delta = 0.0, dab = 0.0;
#pragma unroll
    for (int w = 0; w < SP_WAVES; ++w) {    // the same order in every thread
        delta += s_part[w];
        dab += s_part[SP_WAVES + w];
    }
    *delta_out = delta;
    *dab_out = dab;
}

#define SP_SHARED_FROM(lds)                                                             \
    {(float *)((lds) + EkSpLds::tile), (float *)((lds) + EkSpLds::ytile),                \
     (unsigned long long *)((lds) + EkSpLds::amb_key), (uint32_t *)((lds) + EkSpLds::chg_f), \
     (float *)((lds) + EkSpLds::chg_od), (float *)((lds) + EkSpLds::chg_nd),              \
     (int32_t *)((lds) + EkSpLds::chg_oa), (int32_t *)((lds) + EkSpLds::chg_na),          \
     (uint32_t *)((lds) + EkSpLds::amb_f), (float *)((lds) + EkSpLds::amb_d),             \
     (int32_t *)((lds) + EkSpLds::cols), s_best, s_rowf, &s_n_chg, &s_n_amb, &s_reach,    \
     &s_n_col, s_T, s_acc, s_bcnt}

__device__ __forceinline__ uint32_t *ek_sp_list(const EkSpArgs &p, int slot, int k)
{
    return p.spec_lists + ((size_t)slot * 5 + k) * SP_CAP_CHG;
}

// ---- every slot of the window at once, on the state the window opens with ---------------------
// One workgroup per slot.  A proposal reads the labels and distances of its
// bucket's frames and -- where members stay behind -- the medoids within their
// reach; the frames it would change and the two sums over them go to memory.
// The window's workgroup (below) takes a slot's lists as they are unless an
// earlier slot it ACCEPTED changed a frame of this slot's bucket (`bmask`: which
// slots' buckets a frame is in) or put a medoid within the members' reach, or
// had one there (`tabconf`); then the slot is evaluated in its turn as before.
__global__ void __launch_bounds__(SP_NT)
ek_sp_spec_kernel(EkSpArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    __shared__ unsigned long long s_best[EK_WAVE];
    __shared__ uint32_t s_rowf[EK_WAVE];
    __shared__ unsigned int s_n_chg, s_n_amb, s_reach, s_n_col, s_tabconf, s_cin, s_cout, s_moved;
    __shared__ float s_T[EK_PAM_WIN];
    __shared__ double s_part[2 * SP_WAVES];
    __shared__ unsigned int s_bcnt[EK_PAM_WIN + 1];
    __shared__ int s_acc[EK_PAM_WIN];
    const EkSpShared L = SP_SHARED_FROM(sp_lds);
    const int t = threadIdx.x;
    const int slot = blockIdx.x;
    if (t <= EK_PAM_WIN)
        s_bcnt[t] = (t < p.count) ? min(p.bcnt[t], (unsigned int)p.bcap) : 0u;
    if (t < EK_PAM_WIN)
        s_acc[t] = 0;
    if (t == 0) {
        s_n_chg = 0;
        s_n_amb = 0;
        s_reach = 0;
        s_n_col = 0;
        s_tabconf = 0;
        s_cin = 0;
        s_cout = 0;
        s_moved = 0;
    }
    __syncthreads();
    unsigned int n_amb = 0;
    const int status = ek_sp_evaluate<true>(p, L, slot, false, make_uint2(0u, 0u), nullptr,
                                            &n_amb);
    const unsigned int n_chg = s_n_chg;
    double delta = 0.0, dab = 0.0;
    if (status == 0) {
        ek_sp_delta(L, n_chg, s_part, &delta, &dab);
        uint32_t *gf = ek_sp_list(p, slot, 0), *god = ek_sp_list(p, slot, 1),
                 *gnd = ek_sp_list(p, slot, 2), *goa = ek_sp_list(p, slot, 3),
                 *gna = ek_sp_list(p, slot, 4);
        unsigned int mv = 0;
        for (unsigned int q = t; q < n_chg; q += SP_NT) {
            gf[q] = L.chg_f[q];
            god[q] = __float_as_uint(L.chg_od[q]);
            gnd[q] = __float_as_uint(L.chg_nd[q]);
            goa[q] = (uint32_t)L.chg_oa[q];
            gna[q] = (uint32_t)L.chg_na[q];
            if (L.chg_oa[q] != L.chg_na[q]) {   // (the stale-window mask of the window kernel)
                const int32_t ia = L.chg_oa[q] - p.cid0, ib = L.chg_na[q] - p.cid0;
                if (ia >= 0 && ia < p.win_count)
                    mv |= 1u << ia;
                if (ib >= 0 && ib < p.win_count)
                    mv |= 1u << ib;
            }
        }
        if (mv)
            atomicOr(&s_moved, mv);
        // the search's columns depend on which earlier slots are accepted where
        // the old medoid of one or its proposal is within the members' reach
        if (n_amb > 0 && t < slot) {
            const float lim = __uint_as_float(s_reach) * 1.001f + 1e-3f;
            const float d_old = p.O[(size_t)slot * p.K + p.cid0 + t];
            const float d_new = p.T[(size_t)t * p.K + p.cid0 + slot];
            if (!(d_old > lim) || !(d_new > lim))
                atomicOr(&s_tabconf, 1u << t);
        }
    }
    // ---- which slots read what which slots would change ----------------------------------------
    // One 64-bit word per frame: bit j = in slot j's bucket, bit 32 + j = slot j would
    // change it.  Every mark is an atomic OR that returns the word before it, and the
    // marks of one word happen one after the other: of a slot's change mark and
    // another slot's bucket mark on the same frame, the one made second sees the first.
    {
        const unsigned int nb = s_bcnt[slot];
        uint32_t cin = 0, cout = 0;
        for (unsigned int e = t; e < nb; e += SP_NT) {
            const unsigned long long old =
                atomicOr(&p.marks[p.bucket[(size_t)slot * p.bcap + e].x], 1ull << slot);
            cin |= (uint32_t)(old >> 32);
        }
        if (status == 0)
            for (unsigned int q = t; q < n_chg; q += SP_NT) {
                const unsigned long long old = atomicOr(&p.marks[L.chg_f[q]], 1ull << (32 + slot));
                cout |= (uint32_t)old;
            }
        cin &= (1u << slot) - 1u;                               // the slots before this one
        cout &= (slot >= 31) ? 0u : ~((2u << slot) - 1u);       // the slots after it
        if (cin)
            atomicOr(&s_cin, cin);
        if (cout)
            atomicOr(&s_cout, cout);
    }
    __syncthreads();
    if (t == 0) {
        EkSpSpecRec r;
        r.n_chg = n_chg;
        r.n_amb = n_amb;
        r.status = (uint32_t)status;
        r.tabconf = s_tabconf;
        r.conf_in = s_cin;
        r.conf_out = s_cout;
        r.moved = s_moved;
        r.pad = 0;
        r.delta = delta;
        r.dab = dab;
        p.spec[slot] = r;
    }
}

__global__ void __launch_bounds__(SP_NT)
ek_sp_window_kernel(EkSpArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    double *la = (double *)(sp_lds + EkSpLds::tile);            // (outside the search)
    double *s_chunk = (double *)(sp_lds + EkSpLds::chunk);
    double *s_chunk_old = (double *)(sp_lds + EkSpLds::chunk_old);
    uint32_t *s_lbits = (uint32_t *)(sp_lds + EkSpLds::lbits);
    uint32_t *s_dirty = (uint32_t *)(sp_lds + EkSpLds::dirty);  // leaves whose sum is out of date
    uint32_t *chg_f = (uint32_t *)(sp_lds + EkSpLds::chg_f);
    float *chg_od = (float *)(sp_lds + EkSpLds::chg_od);
    float *chg_nd = (float *)(sp_lds + EkSpLds::chg_nd);
    int32_t *chg_oa = (int32_t *)(sp_lds + EkSpLds::chg_oa);
    int32_t *chg_na = (int32_t *)(sp_lds + EkSpLds::chg_na);
    int32_t *tleaf = (int32_t *)(sp_lds + EkSpLds::tleaf);
    __shared__ unsigned long long s_best[EK_WAVE];
    __shared__ uint32_t s_rowf[EK_WAVE];
    __shared__ uint32_t s_cbits[EK_SP_MAX_CHUNKS / 32];
    __shared__ unsigned int s_n_chg, s_n_amb, s_reach, s_n_col, s_mask, s_n_leaf;
    __shared__ float s_T[EK_PAM_WIN];
    __shared__ double s_part[2 * SP_WAVES];
    __shared__ unsigned int s_bcnt[EK_PAM_WIN + 1];
    __shared__ int s_stop, s_acc[EK_PAM_WIN];
    __shared__ uint32_t s_stale;
    __shared__ EkPamWin s_win;
    __shared__ EkSpSpecRec s_rec[EK_PAM_WIN];
    const EkSpShared L = SP_SHARED_FROM(sp_lds);
    const int t = threadIdx.x, lane = t & (EK_WAVE - 1),
              wv = __builtin_amdgcn_readfirstlane(t / EK_WAVE);
    const int A = p.A, K = p.K;
#ifdef EK_SP_PROF
    unsigned long long sp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long sp_t0 = wall_clock64();
#endif

    // ---- the window record; a row a rejected proposal still sits in --------------------
    for (int i = t; i < (int)(sizeof(EkPamWin) / 4); i += SP_NT)
        ((uint32_t *)&s_win)[i] = (i == 0) ? (uint32_t)p.count : 0u;
    if (p.restore >= 0) {
        for (int r = t; r < 3 * A; r += SP_NT)
            p.med_aos[(size_t)p.restore * 3 * A + r] = p.med_aos[(size_t)K * 3 * A + r];
        if (t == 0)
            p.med_G[p.restore] = p.med_G[K];
    }
    for (int c = t; c < EK_SP_MAX_CHUNKS; c += SP_NT)
        s_chunk[c] = (c < p.n_chunks) ? p.chunk[2 * (size_t)c] : 0.0;
    for (int i = t; i < EK_SP_MAX_CHUNKS * EK_PW_FULL_LEAVES / 32; i += SP_NT) {
        s_lbits[i] = 0;
        s_dirty[i] = 0;
    }
    bool tree_fresh = true;     // leaf and chunk sums are those of the state
    if (t < EK_SP_MAX_CHUNKS / 32)
        s_cbits[t] = 0;
    if (t < EK_PAM_WIN)
        s_acc[t] = 0;
    if (t <= EK_PAM_WIN)
        s_bcnt[t] = (t < p.count) ? min(p.bcnt[t], (unsigned int)p.bcap) : 0u;
    if (p.use_spec)
        for (int i = t; i < (int)(EK_PAM_WIN * sizeof(EkSpSpecRec) / 4); i += SP_NT)
            ((uint32_t *)s_rec)[i] = ((const uint32_t *)p.spec)[i];
    if (t == 0) {
        s_stop = p.count;
        s_stale = 0;
        s_n_chg = 0;
        s_n_amb = 0;
        s_reach = 0;
        s_n_col = 0;
        s_mask = 0;
        s_n_leaf = 0;
    }
    // the first slot's frames; every slot evaluated here fetches the next one's while it works
    uint2 f_pre = make_uint2(0u, 0u);
    int pre_slot = -1;
    if (!p.use_spec) {
        if ((unsigned int)t < min(p.bcnt[0], (unsigned int)p.bcap))
            f_pre = p.bucket[t];
        pre_slot = 0;
    }
    __syncthreads();
    double total = ek_sp_total(s_chunk, p.n_chunks);      // the state's sum of squares
    SP_T(8);

    // (the same in every thread)
    uint32_t acc_mask = 0;      // slots accepted so far
    uint32_t poison = 0;        // slots whose evaluation ahead read what those changed (as
                                //   seen from the changing slot: conf_out; from the reading
                                //   one: conf_in & acc_mask)
    bool in_turn = !p.use_spec; // every slot from here on is evaluated in its turn
    bool stores_pending = false;// the state was written and nobody has waited for the stores
    int n_spec = 0;
    // this thread's entry of a slot's lists (slots evaluated ahead; most change fewer
    // frames than the workgroup has threads): asked for while the slot before is decided
    uint32_t nx_f = 0, nx_od = 0, nx_nd = 0, nx_oa = 0, nx_na = 0;
    int nx_slot = -1;
    int slot = 0;
    for (; slot < p.count; ++slot) {
        if (slot >= s_stop)
            break;
        if (p.use_spec && !in_turn) {
            // ---- a run of slots whose evaluation ahead stands and whose sum of changes
            // decides: their verdicts need nothing but the slots' records (every thread
            // works them out alike), and what the accepted ones change -- disjoint
            // frames: none reads or writes what another accepted one changed -- is
            // written afterwards with the lists' loads in flight together, not slot
            // after slot at a trip to memory each
            const int run0 = slot;
            uint32_t run_acc = 0;
            int stop_r = s_stop;
            uint32_t stale_r = s_stale;
            while (slot < p.count && slot < stop_r) {
                const EkSpSpecRec &r = s_rec[slot];
                if (r.status != 0 || ((poison >> slot) & 1u) ||
                    ((r.tabconf | r.conf_in) & acc_mask))
                    break;
                const bool obv = !p.exact_always && fabs(r.delta) > 1e-9 * total + 1e-11 * r.dab;
                if (!obv)
                    break;
                const bool acc = r.delta < 0.0;
                const double total_new = total + r.delta;
                if (t == 0) {
                    EkPamOut o;
                    o.sum_old = total;
                    o.sum_new = total_new;
                    o.n_frames = p.n;
                    o.n_amb = r.n_amb;
                    o.moved = acc ? r.moved : 0u;
                    s_win.out[slot] = o;
                    s_acc[slot] = acc ? 1 : 0;
                    s_win.accept[slot] = acc ? 1 : 0;
                }
                if (acc) {
                    total = total_new;
                    acc_mask |= 1u << slot;
                    poison |= r.conf_out;
                    run_acc |= 1u << slot;
                    stale_r |= r.moved;
                    const uint32_t later = (slot >= 31) ? 0u : (stale_r >> (slot + 1));
                    if (later) {
                        const int first = slot + 1 + (__ffs((int)later) - 1);
                        if (first < stop_r)
                            stop_r = first;
                    }
                }
                ++n_spec;
                ++slot;
            }
            if (t == 0 && slot > run0) {
                s_stop = stop_r;
                s_stale = stale_r;
                s_win.stale = stale_r;
                s_win.stop = stop_r;
            }
            if (run_acc) {
                constexpr int G = 8;            // slots whose lists are in flight together
                for (int base = run0; base < slot; base += G) {
                    uint32_t f[G], na[G];
                    float nd[G];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int j = base + u;
                        const bool ok = j < slot && ((run_acc >> j) & 1u) &&
                                        (unsigned int)t < s_rec[j].n_chg;
                        f[u] = ok ? ek_sp_list(p, j, 0)[t] : 0xffffffffu;
                        nd[u] = ok ? __uint_as_float(ek_sp_list(p, j, 2)[t]) : 0.f;
                        na[u] = ok ? ek_sp_list(p, j, 4)[t] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        if (f[u] == 0xffffffffu)
                            continue;
                        p.dist[f[u]] = nd[u];
                        p.assign[f[u]] = (int32_t)na[u];
                        const int g = ek_sp_leaf_of(f[u], p);
                        atomicOr(&s_dirty[g >> 5], 1u << (g & 31));
                    }
                }
                for (int j = run0; j < slot; ++j) {     // (lists longer than the workgroup)
                    if (!((run_acc >> j) & 1u))
                        continue;
                    for (unsigned int q = SP_NT + t; q < s_rec[j].n_chg; q += SP_NT) {
                        const uint32_t f = ek_sp_list(p, j, 0)[q];
                        p.dist[f] = __uint_as_float(ek_sp_list(p, j, 2)[q]);
                        p.assign[f] = (int32_t)ek_sp_list(p, j, 4)[q];
                        const int g = ek_sp_leaf_of(f, p);
                        atomicOr(&s_dirty[g >> 5], 1u << (g & 31));
                    }
                }
                tree_fresh = false;
                stores_pending = true;
            }
            if (slot > run0)
                ek_lds_barrier();
            if (slot >= p.count || slot >= s_stop)
                break;
        }
        const bool spec_ok = !in_turn && s_rec[slot].status == 0 &&
                             !((poison >> slot) & 1u) &&
                             !((s_rec[slot].tabconf | s_rec[slot].conf_in) & acc_mask);
        unsigned int n_amb = 0;
        int status = 0;
        bool in_regs = false;
        uint32_t cf = 0, coa = 0, cna = 0;
        float cod = 0.f, cnd = 0.f;
        if (spec_ok) {
            // ---- evaluated ahead, and nothing it read has changed: its lists ---------------
            const unsigned int n = s_rec[slot].n_chg;
            n_amb = s_rec[slot].n_amb;
            const uint32_t *gf = ek_sp_list(p, slot, 0), *god = ek_sp_list(p, slot, 1),
                           *gnd = ek_sp_list(p, slot, 2), *goa = ek_sp_list(p, slot, 3),
                           *gna = ek_sp_list(p, slot, 4);
            ++n_spec;
            if (n <= SP_NT) {
                in_regs = true;
                if (nx_slot != slot && (unsigned int)t < n) {
                    nx_f = gf[t];
                    nx_od = god[t];
                    nx_nd = gnd[t];
                    nx_oa = goa[t];
                    nx_na = gna[t];
                }
                cf = nx_f;
                cod = __uint_as_float(nx_od);
                cnd = __uint_as_float(nx_nd);
                coa = nx_oa;
                cna = nx_na;
            } else {
                for (unsigned int q = t; q < n; q += SP_NT) {
                    chg_f[q] = gf[q];
                    chg_od[q] = __uint_as_float(god[q]);
                    chg_nd[q] = __uint_as_float(gnd[q]);
                    chg_oa[q] = (int32_t)goa[q];
                    chg_na[q] = (int32_t)gna[q];
                }
                if (t == 0)
                    s_n_chg = n;
                __syncthreads();
                stores_pending = false;
            }
        } else {
            if (stores_pending) {       // it reads the state
                __syncthreads();
                stores_pending = false;
            }
            uint2 f_next = make_uint2(0u, 0u);
            status = ek_sp_evaluate<false>(p, L, slot, pre_slot == slot, f_pre, &f_next, &n_amb);
            f_pre = f_next;
            pre_slot = slot + 1;
        }
        const unsigned int n_chg = in_regs ? s_rec[slot].n_chg : s_n_chg;
        if (status != 0) {
            // more than one workgroup should take on (or more members than
            // declared): the window ends before this slot, nothing of it is kept
            if (t == 0) {
                if (status == 2)
                    s_win.err = 1 + slot;
                else
                    s_win.pad = 1 + slot;
                s_win.stop = slot;
            }
            break;
        }
        SP_T(1);
        // ---- what the proposal would change in the sum of squares (ek_sp_delta) ------------
        double delta, dab;
        if (spec_ok) {
            delta = s_rec[slot].delta;
            dab = s_rec[slot].dab;
        } else {
            ek_sp_delta(L, n_chg, s_part, &delta, &dab);
        }
        const bool obvious = !p.exact_always && fabs(delta) > 1e-9 * total + 1e-11 * dab;
        bool accept, wrote = false;
        double total_new;
        unsigned int n_leaf = 0;
        if (obvious) {
            accept = delta < 0.0;
            total_new = total + delta;
            if (accept) {
                // the trial state becomes the state; its leaves' sums are out of date
                unsigned int m = 0;
                const unsigned int q_end = in_regs ? (((unsigned int)t < n_chg) ? t + 1 : 0u)
                                                   : n_chg;
                for (unsigned int q = t; q < q_end; q += SP_NT) {
                    const uint32_t f = in_regs ? cf : chg_f[q];
                    const int32_t oa = in_regs ? (int32_t)coa : chg_oa[q],
                                  na = in_regs ? (int32_t)cna : chg_na[q];
                    p.dist[f] = in_regs ? cnd : chg_nd[q];
                    p.assign[f] = na;
                    if (oa != na) {
                        const int32_t ia = oa - p.cid0, ib = na - p.cid0;
                        if (ia >= 0 && ia < p.win_count)
                            m |= 1u << ia;
                        if (ib >= 0 && ib < p.win_count)
                            m |= 1u << ib;
                    }
                    const int g = ek_sp_leaf_of(f, p);
                    atomicOr(&s_dirty[g >> 5], 1u << (g & 31));
                }
                if (m)
                    atomicOr(&s_mask, m);
                tree_fresh = false;
                // (nobody waits for these stores here: a slot taken over as evaluated
                // ahead neither reads nor writes a frame an accepted one changed --
                // `poison` -- and whoever reads the state waits first)
                stores_pending = true;
            }
            ek_lds_barrier();
            SP_T(2);
        } else {
            if (in_regs) {
                if ((unsigned int)t < n_chg) {
                    chg_f[t] = cf;
                    chg_od[t] = cod;
                    chg_nd[t] = cnd;
                    chg_oa[t] = (int32_t)coa;
                    chg_na[t] = (int32_t)cna;
                }
            }
            if (stores_pending || in_regs) {    // the sums below read the state
                __syncthreads();
                stores_pending = false;
            }
            if (!tree_fresh) {
                // ---- the leaves changed since the tree was last brought up to date -------
                for (;;) {
                    if (t == 0)
                        s_n_leaf = 0;
                    ek_lds_barrier();
                    const int n_words = (p.n_leaves + 31) / 32;
                    for (int w = t; w < n_words; w += SP_NT) {
                        uint32_t bits = s_dirty[w];
                        while (bits) {
                            const int b = __ffs((int)bits) - 1;
                            const unsigned int q = atomicAdd(&s_n_leaf, 1u);
                            if (q >= SP_CAP_CHG)
                                break;          // the next round takes the rest
                            tleaf[q] = w * 32 + b;
                            bits &= bits - 1;
                        }
                        s_dirty[w] = bits;
                    }
                    ek_lds_barrier();
                    const unsigned int listed = s_n_leaf;
                    if (listed == 0)
                        break;
                    ek_sp_readd_leaves(p, tleaf, min(listed, (unsigned int)SP_CAP_CHG),
                                       s_cbits);
                    __syncthreads();
                }
                ek_sp_readd_chunks(p, s_cbits, s_chunk, s_chunk_old, la, lane, wv);
                ek_lds_barrier();
                if (t < EK_SP_MAX_CHUNKS / 32)
                    s_cbits[t] = 0;
                if (t == 0)
                    s_n_leaf = 0;
                total = ek_sp_total(s_chunk, p.n_chunks);
                tree_fresh = true;
                ek_lds_barrier();
            }
            // ---- the trial state, written into the state; old values kept ---------------
            unsigned int m = 0;
            for (unsigned int q = t; q < n_chg; q += SP_NT) {
                const uint32_t f = chg_f[q];
                const int32_t oa = chg_oa[q], na = chg_na[q];
                p.dist[f] = chg_nd[q];
                p.assign[f] = na;
                if (oa != na) {
                    const int32_t ia = oa - p.cid0, ib = na - p.cid0;
                    if (ia >= 0 && ia < p.win_count)
                        m |= 1u << ia;
                    if (ib >= 0 && ib < p.win_count)
                        m |= 1u << ib;
                }
                const int g = ek_sp_leaf_of(f, p);
                const uint32_t bit = 1u << (g & 31);
                if (!(atomicOr(&s_lbits[g >> 5], bit) & bit))
                    tleaf[atomicAdd(&s_n_leaf, 1u)] = g;
            }
            if (m)
                atomicOr(&s_mask, m);
            __syncthreads();
            SP_T(2);
            // ---- the leaves and chunks that changed, re-added ------------------------------
            n_leaf = s_n_leaf;
            ek_sp_readd_leaves(p, tleaf, n_leaf, s_cbits);
            __syncthreads();
            SP_T(3);
            ek_sp_readd_chunks(p, s_cbits, s_chunk, s_chunk_old, la, lane, wv);
            ek_lds_barrier();
            SP_T(4);
            total_new = ek_sp_total(s_chunk, p.n_chunks);
            SP_T(5);
            // kmedoids.py:478-479, :683 (ek_pw_window_kernel's verdict)
            accept = total_new / p.n_total < total / p.n_total;
            wrote = true;
        }
        // ---- verdict, taken by every thread alike --------------------------------------------
        if (t == 0) {
            EkPamOut o;
            o.sum_old = total;
            o.sum_new = total_new;
            o.n_frames = p.n;
            o.n_amb = n_amb;
            o.moved = s_mask;
            s_win.out[slot] = o;
            s_acc[slot] = accept ? 1 : 0;
            s_win.accept[slot] = accept ? 1 : 0;
            if (accept) {
                const uint32_t stale = s_stale | o.moved;
                s_stale = stale;
                s_win.stale = stale;
                const uint32_t later = (slot >= 31) ? 0u : (stale >> (slot + 1));
                if (later) {
                    const int first = slot + 1 + (__ffs((int)later) - 1);
                    if (first < s_stop) {
                        s_stop = first;
                        s_win.stop = first;
                    }
                }
            }
        }
        SP_T(6);
        if (accept) {
            total = total_new;
            acc_mask |= 1u << slot;
            if (spec_ok)
                poison |= s_rec[slot].conf_out;
            else if (p.use_spec)
                in_turn = true;         // what it changed was never held against the buckets
        } else if (!obvious) {
            for (unsigned int q = t; q < n_chg; q += SP_NT) {
                const uint32_t f = chg_f[q];
                p.dist[f] = chg_od[q];
                p.assign[f] = chg_oa[q];
            }
            for (unsigned int q = t; q < n_leaf; q += SP_NT) {
                const int g = tleaf[q];
                p.leaf[2 * (size_t)g] = p.leaf[2 * (size_t)g + 1];
            }
            for (int c = t; c < p.n_chunks; c += SP_NT)
                if ((s_cbits[c >> 5] >> (c & 31)) & 1u)
                    s_chunk[c] = s_chunk_old[c];
        }
        for (unsigned int q = t; q < n_leaf; q += SP_NT)
            s_lbits[tleaf[q] >> 5] = 0;
        if (wrote)
            __syncthreads();    // the state was written: the stores have to be through
        else
            ek_lds_barrier();
        if (!obvious && t < EK_SP_MAX_CHUNKS / 32)
            s_cbits[t] = 0;
        if (t == 0) {
            s_n_chg = 0;
            s_n_amb = 0;
            s_reach = 0;
            s_n_col = 0;
            s_mask = 0;
            s_n_leaf = 0;
        }
        ek_lds_barrier();
        SP_T(7);
    }
    SP_T(0);
    // ---- the accepted proposals are their clusters' medoids (kmedoids.py:684-690) ---------
    // a wave per accepted slot, its loads in flight together (one slot after the
    // other, each a trip to memory, was a quarter of a short window's time)
    __syncthreads();
    if (!p.finish_later) {
        int rank = 0;
        for (int i = 0; i < p.count; ++i) {
            if (!s_acc[i])
                continue;
            if ((rank++ % SP_WAVES) != wv)
                continue;
            const int32_t cid = p.cid0 + i;
            const float *src = p.frames_aos + (size_t)p.frames[i] * 3 * A;
            float *dst = p.med_aos + (size_t)cid * 3 * A;
            for (int r0 = 0; r0 < 3 * A; r0 += 8 * EK_WAVE) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = r0 + k * EK_WAVE + lane;
                    v[k] = (r < 3 * A) ? src[r] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = r0 + k * EK_WAVE + lane;
                    if (r < 3 * A)
                        dst[r] = v[k];
                }
            }
            if (lane == 0) {
                p.med_G[cid] = p.G[p.frames[i]];
                if (p.med_idx)
                    p.med_idx[cid] = p.frames[i];
            }
        }
    }
    if (p.use_spec && t == 0)
        s_win.pad |= n_spec << 8;       // (slots taken over as evaluated ahead: a diagnostic)
    for (int i = t; i < (int)(sizeof(EkPamWin) / 4); i += SP_NT) {
        ((uint32_t *)p.win)[i] = ((const uint32_t *)&s_win)[i];
        if (p.win_host)
            ((uint32_t *)p.win_host)[i] = ((const uint32_t *)&s_win)[i];
    }
    SP_T(9);
#ifdef EK_SP_PROF
    if (t == 0 && p.prof)
        for (int k = 0; k < 10; ++k)
            p.prof[k] += sp_acc[k];
#endif
}

// ---- after a window whose slots were evaluated ahead: one workgroup per slot ----------------------
// takes the slot's marks back and, if its proposal was accepted, puts it into the
// medoid table (kmedoids.py:684-690) -- off the window's own workgroup, which did
// these one slot after the other, a trip to memory each
__global__ void __launch_bounds__(256)
ek_sp_finish_kernel(EkSpArgs p)
{
    const int t = threadIdx.x, slot = blockIdx.x;
    const unsigned int nb = min(p.bcnt[slot], (unsigned int)p.bcap);
    for (unsigned int e = t; e < nb; e += 256)
        p.marks[p.bucket[(size_t)slot * p.bcap + e].x] = 0ull;
    __syncthreads();            // (everybody has read the bucket's length)
    if (t == 0)                 // ... which the next window's buckets start from
        ((unsigned int *)p.bcnt)[slot] = 0u;
    // the slot's distance vector back to +inf at the listed frames -- what the next
    // window's prefetch did in a launch of its own; only after a window that ran to its
    // end (the proposal a window stopped at is made again from its vector)
    if (p.vecs && p.win->stop == p.count)
        for (int64_t i = t; i < p.n_act; i += 256)
            p.vecs[(size_t)slot * p.n_pad + p.act_list[i]] = __builtin_inff();
    if (slot >= p.win->stop || !p.win->accept[slot])
        return;
    const int32_t cid = p.cid0 + slot;
    const float *src = p.frames_aos + (size_t)p.frames[slot] * 3 * p.A;
    float *dst = p.med_aos + (size_t)cid * 3 * p.A;
    for (int r = t; r < 3 * p.A; r += 256)
        dst[r] = src[r];
    if (t == 0) {
        p.med_G[cid] = p.G[p.frames[slot]];
        if (p.med_idx)
            p.med_idx[cid] = p.frames[slot];
    }
}

void ek_launch_sp_finish(const EkSpArgs &p, hipStream_t s)
{
    hipLaunchKernelGGL(ek_sp_finish_kernel, dim3(p.count), dim3(256), 0, s, p);
}

size_t ek_sp_lds_bytes()
{
    return EkSpLds::end;
}

size_t ek_sp_spec_bytes()
{
    return EK_PAM_WIN * sizeof(EkSpSpecRec) + (size_t)EK_PAM_WIN * 5 * SP_CAP_CHG * 4;
}

void ek_launch_sp_spec(const EkSpArgs &p, hipStream_t s)
{
    const size_t lds = ek_sp_lds_bytes();
    (void)hipFuncSetAttribute((const void *)ek_sp_spec_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ek_sp_spec_kernel, dim3(p.count), dim3(SP_NT), lds, s, p);
}

void ek_launch_sp_window(const EkSpArgs &p, hipStream_t s)
{
    const size_t lds = ek_sp_lds_bytes();
    (void)hipFuncSetAttribute((const void *)ek_sp_window_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ek_sp_window_kernel, dim3(1), dim3(SP_NT), lds, s, p);
}
