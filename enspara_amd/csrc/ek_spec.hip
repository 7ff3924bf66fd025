// ek_spec.hip -- k-centers with several candidate centers per pass over the
// frames ("speculative" k-centers).  Exactly the sequential algorithm of
// enspara/cluster/kcenters.py:217-231 (same centers, same order, same labels
// and distances, bit for bit), reorganised around the fact that the
// one-center pass is HBM-bound with ~90 % of the vector ALUs idle:
//
//   round:  the frames are streamed ONCE against T candidate centers: the
//           true farthest point (candidate 0, applied immediately, exactly as
//           ek_step_kernel does) and T-1 further far points, whose distance
//           vectors are kept (4 bytes per frame each).
//   then:   the farthest point of the updated distances is found again
//           (kcenters.py:282).  If it is one of the stored candidates its
//           distances are already there: applying them is a 16-byte-per-frame
//           update instead of a (12 A + 20)-byte pass.  This repeats
//           until the farthest point is not a stored candidate; then the next
//           round starts.
//
// Every accepted center is the first-index arg-max of the current distances and
// its distances come from the same per-pair FMA chain + quartic solve as the
// one-center kernel, so nothing about the result depends on which candidates
// were guessed -- only the number of passes does.  All decisions are taken on
// the device.  In the one-launch-per-step form (this file + ek_chain.hip: what
// the multi-shard protocol uses, an exchange sitting between the steps) they
// are taken by single-workgroup kernels that are the sole writers of the plan
// the following kernels read; a single shard runs the same steps in three
// launches (ek_round.hip), the single-workgroup parts riding in the last
// workgroup of the launch that produces their input.
//
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"
#include "ek_top_dev.h"
#include "ek_chain_dev.h"

// ---------------------------------------------------------------------------
// plan: choose the round's candidates among the records on offer
// (n_recs = ranks x per-rank candidates), ordered by (distance desc, global
// index asc) -- position 0 is the global first-index arg-max (kcenters.py:337:
// lowest rank wins ties; ranks own ascending contiguous blocks).
// ---------------------------------------------------------------------------
// D (optional): pairwise distances of the records on offer, [64][64].  With it
// the candidates after the first are taken greedily like ek_top_records_kernel
// does within one shard -- the record with the largest remaining distance, then
// every other one's is lowered by its distance to it -- instead of by their own
// distances alone: records offered by different shards may be close to one
// another.  One wave, lane r = record r.
__global__ void __launch_bounds__(EK_WAVE)
ek_plan_kernel(const unsigned char *__restrict__ recs, int n_recs, int A, int T,
               double cutoff, const float *__restrict__ D,
               EkPlan *__restrict__ plan, EkHist *__restrict__ hist,
               EkCtl *__restrict__ ctl)
{
    const int lane = threadIdx.x;
    const size_t rstride = ek_rec_bytes(A);
    if (lane == 0) {
        plan->go = 0;
        plan->apply = -1;
        plan->miss = 1;
        plan->used = 0;
        plan->teff = 0;
        plan->chain_n = 0;
        plan->napply = 0;
    }
    if (ctl->stopped || ctl->n_done >= ctl->limit)
        return;
    bool open = false;
    float orig = -__builtin_inff();
    long long g = 0x7fffffffffffffffLL;
    double tr = 0.0;
    if (lane < n_recs) {
        const EkRecHdr *h = (const EkRecHdr *)(recs + (size_t)lane * rstride);
        if (h->valid) {
            open = true;
            orig = h->maxdist;
            g = h->gidx;
            tr = h->trace;
        }
    }
    float cur = orig;
    int teff = 0;
    float first_max = 0.f;
    long long first_g = 0;
    for (int j = 0; j < T; ++j) {
        // arg-max over the open records: largest remaining distance, lowest
        // global index on ties (kcenters.py:337 for position 0)
        float v = open ? cur : -__builtin_inff();
        long long bg = open ? g : 0x7fffffffffffffffLL;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64);
            const long long og = __shfl_xor(bg, off, 64);
            if (ov > v || (ov == v && og < bg)) {
                v = ov;
                bg = og;
            }
        }
        const unsigned long long who = __ballot(open && g == bg);
        if (who == 0)
            break;
        const int w = __ffsll((long long)who) - 1;
        const float w_orig = __shfl(orig, w, 64);
        const double w_tr = __shfl(tr, w, 64);
        if (lane == 0) {
            plan->src[j] = w;
            plan->gidx[j] = bg;
            plan->maxdist[j] = w_orig;
            plan->trace[j] = w_tr;
        }
        if (j == 0) {
            first_max = w_orig;
            first_g = bg;
        }
        if (lane == w)
            open = false;
        if (D) {
            const float d = D[w * 64 + lane];
            if (open && d < cur)
                cur = d;
        }
        ++teff;
    }
    if (teff == 0 || lane != 0)
        return;
    // stop rule of kcenters.py:217
    if (!((double)first_max > cutoff)) {
        ctl->stopped = 1;
        return;
    }
    const int label = ctl->n_done;
    plan->go = 1;
    plan->teff = teff;
    plan->label = label;
    plan->used = 1;
    plan->miss = 0;
    hist[label].gidx = first_g;
    hist[label].dist = first_max;
    hist[label].set = 1;
    ctl->n_done = label + 1;
    ctl->n_rounds = ctl->n_rounds + 1;
}

// pairwise distances of the valid records on offer (one wave per pair; used only
// to steer the choice of guesses, so the summation order is free -- but it is
// the same on every rank, and so is the choice)
__global__ void __launch_bounds__(EK_BLOCK)
ek_rec_pair_kernel(const unsigned char *__restrict__ recs, int n_recs, int A,
                   float *__restrict__ D)
{
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int w = blockIdx.x * (EK_BLOCK / EK_WAVE) + threadIdx.x / EK_WAVE;
    const int i = w / 64, j = w % 64;
    if (i >= j || j >= n_recs)
        return;
    const size_t rstride = ek_rec_bytes(A);
    const EkRecHdr *hi = (const EkRecHdr *)(recs + (size_t)i * rstride);
    const EkRecHdr *hj = (const EkRecHdr *)(recs + (size_t)j * rstride);
    if (!hi->valid || !hj->valid)
        return;
    const float *x = (const float *)(recs + (size_t)i * rstride + sizeof(EkRecHdr));
    const float *y = (const float *)(recs + (size_t)j * rstride + sizeof(EkRecHdr));
    float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = lane; a < A; a += EK_WAVE) {
        const float x0 = x[3 * a], x1 = x[3 * a + 1], x2 = x[3 * a + 2];
        const float y0 = y[3 * a], y1 = y[3 * a + 1], y2 = y[3 * a + 2];
        S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
        S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
        S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            S[q] += __shfl_xor(S[q], off, 64);
    if (lane == 0) {
        const float d = ek_rmsd_from_S(S, hi->trace, hj->trace, A);
        D[i * 64 + j] = d;
        D[j * 64 + i] = d;
    }
}

// D != nullptr and more records than candidates: greedy choice among them
void ek_launch_plan(const unsigned char *recs, int n_recs, int A, int T,
                    double cutoff, float *D, EkPlan *plan, EkHist *hist,
                    EkCtl *ctl, hipStream_t s)
{
    const bool greedy = D != nullptr && n_recs > T;
    if (greedy)
        hipLaunchKernelGGL(ek_rec_pair_kernel,
                           dim3(64 * 64 / (EK_BLOCK / EK_WAVE)), dim3(EK_BLOCK),
                           0, s, recs, n_recs, A, D);
    hipLaunchKernelGGL(ek_plan_kernel, dim3(1), dim3(EK_WAVE), 0, s, recs,
                       n_recs, A, T, cutoff, greedy ? D : nullptr, plan, hist,
                       ctl);
}

// ---------------------------------------------------------------------------
// the pass: every frame against the round's T candidates (T = 4, 8, 16)
// ---------------------------------------------------------------------------
// One lane = one frame.  The candidates are the same for every lane, so they do
// not need vector registers or LDS at all: the round's candidates are laid out
// once in global memory as [atom][pair][xyz][2] (ek_ctile_kernel, or the kernel
// that chose them) and every wave reads them with scalar loads (s_load_dwordx8,
// through the scalar cache) one CHUNK ahead of the FMAs -- a chunk is one atom
// of up to 8 candidates, 24 floats in 24 SGPRs; 16 candidates are two chunks
// per atom, so that two chunk buffers (48 SGPRs) serve every T --;
// v_pk_fma_f32 takes the (candidate 2p, candidate 2p+1) pair straight from an
// SGPR pair.  The frame rows come through buffer loads (SGPR descriptor of the
// tile + scalar row offset + the lane's constant offset: no address registers;
// rows past the tile's end read as 0) into the two halves of register pairs,
// atoms a and a+1 of a coordinate sharing a pair and op_sel picking the half.
// T = 8: 72 accumulators + 36 row registers (rows two trips ahead), 4 waves per
// SIMD; T = 16: 144 accumulators, 2 waves per SIMD -- that form is bound by the
// packed-FMA rate, not by HBM, and is chosen when its extra centers per pass
// pay (ek_run_rounds), i.e. always on shards too small to be bandwidth-bound.
// Each S component is its own IEEE FMA chain in ascending atom order: the same
// bits as every other kernel here, whatever T.
typedef const float __attribute__((address_space(4))) *ek_cfp;

#define EK_PASS2_DIST 2         // trips (of 4 atoms) the row loads run ahead

static inline __host__ __device__ size_t ek_ctile_floats(int A)
{
    return (size_t)ek_ctile_atoms(A) * 3 * EK_MAX_CANDS;
}

// acc += x.lo * c  /  x.hi * c  (per component; c = an SGPR pair)
__device__ __forceinline__ void ek_pkfma_lo(ek_v2f &acc, ek_v2f x, ek_v2f c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
        : "+v"(acc) : "v"(x), "s"(c));
}
__device__ __forceinline__ void ek_pkfma_hi(ek_v2f &acc, ek_v2f x, ek_v2f c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(acc) : "v"(x), "s"(c));
}

// one atom of P candidate pairs: [pair][xyz][2]
template <int P> struct EkCChunk { float v[6 * P]; };

template <int P>
__device__ __forceinline__ void ek_ld_chunk(EkCChunk<P> &o, ek_cfp p)
{
#pragma unroll
    for (int i = 0; i < 6 * P; ++i)
        o.v[i] = p[i];
}

// FMAs [from, to) of the 9 P of one chunk (order: pair, then S row-major) into
// the accumulators of pairs p0 .. p0 + P - 1
template <int NP, int P, bool HI>
__device__ __forceinline__ void ek_chunk_fma(ek_v2f (&s2)[NP][9], int p0, ek_v2f X,
                                             ek_v2f Y, ek_v2f Z,
                                             const EkCChunk<P> &c, int from, int to)
{
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const ek_v2f cx = (ek_v2f){c.v[6 * p + 0], c.v[6 * p + 1]};
        const ek_v2f cy = (ek_v2f){c.v[6 * p + 2], c.v[6 * p + 3]};
        const ek_v2f cz = (ek_v2f){c.v[6 * p + 4], c.v[6 * p + 5]};
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = 9 * p + j;
            if (k < from || k >= to)
                continue;
            const ek_v2f r = (j / 3 == 0) ? X : (j / 3 == 1 ? Y : Z);
            const ek_v2f cc = (j % 3 == 0) ? cx : (j % 3 == 1 ? cy : cz);
            if (HI)
                ek_pkfma_hi(s2[p0 + p][j], r, cc);
            else
                ek_pkfma_lo(s2[p0 + p][j], r, cc);
        }
    }
}

// the round's candidates, interleaved for the kernel below; zeros for unused
// slots and for the EK_CTILE_PAD atoms after the last
template <int T>
__global__ void __launch_bounds__(EK_BLOCK)
ek_ctile_kernel(const unsigned char *__restrict__ recs,
                const EkPlan *__restrict__ plan, int A,
                float *__restrict__ ctile, double *__restrict__ ctrace)
{
    if (!plan->go)
        return;
    const int teff = plan->teff;
    const size_t rstride = ek_rec_bytes(A);
    const int total = ek_ctile_atoms(A) * 3 * T;
    for (int j = blockIdx.x * EK_BLOCK + threadIdx.x; j < total;
         j += gridDim.x * EK_BLOCK) {
        // source order (atom, candidate, xyz); destination: ek_ctile_index
        const int a = j / (3 * T), c = (j % (3 * T)) / 3, k = j % 3;
        float v = 0.f;
        if (a < A && c < teff)
            v = ((const float *)(recs + (size_t)plan->src[c] * rstride +
                                 sizeof(EkRecHdr)))[3 * a + k];
        ctile[ek_ctile_index(T, a, c, k)] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < T) {
        const int c = threadIdx.x;
        ctrace[c] = c < teff ? ((const EkRecHdr *)(recs + (size_t)plan->src[c] *
                                                   rstride))->trace
                             : 0.0;
    }
}

// UPD = true: a k-centers round (candidate 0 updates the state, the others'
// distances go to vecs[0..T-2]).  UPD = false: distances only, all T of them to
// vecs[0..T-1] (PAM proposal prefetch); dist / assign / blockmax are not touched.
// FUSE (the three- and four-launch rounds, ek_round.hip / ek_mshard.hip): the
// accepted chain of the previous round (`pend`) is applied to the frame's state
// on the way in instead of by a pass of its own, a distance vector is stored
// only where a wave holds a finite value, and -- if fz.ord is given (single
// shard) -- the last workgroup to finish works out the presumed acceptance
// order of this round's candidates.
template <int T, bool UPD, bool FUSE>
__global__ void __launch_bounds__(EK_BLOCK, (T <= 4) ? 5 : 4)
ek_pass2_kernel(const float *__restrict__ tiles, const double *__restrict__ G,
                float *__restrict__ dist, int32_t *__restrict__ assign,
                float *__restrict__ vecs, int64_t n, int64_t n_pad, int A,
                const float *__restrict__ ctile,
                const double *__restrict__ ctrace,
                const EkPlan *__restrict__ plan,
                EkBlockMax *__restrict__ blockmax, EkFuse fz)
{
    const EkPend *__restrict__ pend = fz.pend;
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    if (!plan->go)
        return;
    const int tid = threadIdx.x;
    const int teff = plan->teff;
    const int label = plan->label;
    static_assert(EK_BLOCK == EK_TILE, "one workgroup per tile");
    // distances only (!UPD): nothing is shared between the waves of a
    // workgroup, so a short frame list may be launched one wave per workgroup
    // (four workgroups per tile) and spread over four times the CUs
    const int64_t f0 = UPD ? (int64_t)blockIdx.x * EK_BLOCK
                           : (int64_t)blockIdx.x * blockDim.x;
    const int64_t f = f0 + tid;
    // FUSE with an order to work out: the last workgroup does a little more at
    // the end (see there), and "last" is decided by arrival tickets.  Only the
    // workgroups that own one of the round's guesses produce something it reads;
    // every other workgroup draws its ticket right here, where the atomic's
    // latency costs nothing -- the one that draws the last ticket does so after
    // every owner has finished.
    const bool order = UPD && FUSE && fz.ord != nullptr;
    bool owner_blk = false;
    unsigned int ticket = 0;
    if (order) {
#pragma unroll
        for (int j = 1; j < T; ++j) {
            const int64_t l = plan->gidx[j] - fz.goff - (int64_t)blockIdx.x * EK_BLOCK;
            owner_blk |= j < teff && l >= 0 && l < EK_BLOCK;
        }
        if (!owner_blk && tid == 0)
            ticket = __hip_atomic_fetch_add(fz.tick, 1u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    }
    // what the epilogue needs from memory is requested now, not after the loop
    double Gf = 0.0;
    float cur0 = 0.f;
    int32_t lab = -1;           // >= 0: the frame's state changes in this pass
    int own = 0;                // order: this frame is candidate `own` (>= 1)
    if (f < n) {
        Gf = G[f];
        if (UPD)
            cur0 = dist[f];
        if (UPD && FUSE) {
            // is this frame one of the round's guesses?  (its row of the new
            // distance vectors decides the presumed order, see the end)
            if (order) {
#pragma unroll
                for (int j = 1; j < T; ++j)
                    if (j < teff && plan->gidx[j] - fz.goff == f)
                        own = j;
            }
            // kcenters.py:304-306 for the pending chain, in order.  A vector is
            // only stored where some frame of the wave got a finite distance
            // (see the end): elsewhere it is +inf and changes nothing.
            const int pn = pend->n;
            const uint32_t vm = pn > 0 ? fz.vmask[f >> 6] : 0u;
            for (int k = 0; k < pn; ++k) {
                const int slot = pend->slot[k];
                if (!((vm >> (slot + 1)) & 1u))
                    continue;
                const float d = vecs[(size_t)slot * n_pad + f];
                if (d < cur0) {
                    cur0 = d;
                    lab = pend->label0 + k;
                }
            }
        }
    }

    // triangle inequality (ek_round_ti_tiles_kernel, round 5): the candidates that
    // can still change a frame of this tile.  None: the tile's buffer descriptor
    // gets a length of zero -- every row load then returns zeros without touching
    // memory -- and no trip runs; masked candidates of a partly masked tile skip
    // their solves.
    uint32_t tm = 0xffffffffu;
    if (UPD && FUSE && fz.tmask)
        tm = __builtin_amdgcn_readfirstlane(fz.tmask[blockIdx.x]) | (T < 32 ? ~((1u << T) - 1u) : 0u);
    const bool tile_off = UPD && FUSE && (tm & ((1u << T) - 1u)) == 0u;
    const float *tb = tiles + (size_t)(f0 / EK_TILE) * 3 * (size_t)A * EK_TILE;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)tb, 0, tile_off ? 0 : 3 * A * EK_TILE * 4, 0x00020000);
    const int vo = ((int)(f0 % EK_TILE) + tid) * 4;
    // non-temporal (aux bit 1): the frame stream is read once per pass
#define EK_LD(SO, K)                                                           \
    __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(            \
                                  rs, vo + (K) * (EK_TILE * 4), (SO), 2))
    constexpr int NP = T / 2;               // candidate pairs
    constexpr int P = NP < 4 ? NP : 4;      // pairs per chunk
    constexpr int H = NP / P;               // chunks per atom
    constexpr int CH = 6 * P;               // floats per chunk
    constexpr int NF = 9 * P;               // packed FMAs per chunk
    static_assert(NP % P == 0 && (4 * H) % 2 == 0, "whole chunks, even per trip");
    ek_v2f s2[NP][9];
#pragma unroll
    for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            s2[c][j] = (ek_v2f){0.f, 0.f};

    constexpr int DIST = EK_PASS2_DIST;
    constexpr int NB = DIST + 1;
    // X[b][h] = (x of atom 4t + 2h, x of atom 4t + 2h + 1) of the trip in buffer b
    ek_v2f X[NB][2], Y[NB][2], Z[NB][2];
#define EK_ROWS2(B, TR)                                                        \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                              \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                            \
        const int so = ((TR) * 4 + 2 * h + e) * (3 * EK_TILE * 4);             \
        X[B][h][e] = EK_LD(so, 0);                                             \
        Y[B][h][e] = EK_LD(so, 1);                                             \
        Z[B][h][e] = EK_LD(so, 2);                                             \
        __builtin_amdgcn_sched_barrier(0);  /* same issue order everywhere */  \
    }
    const int n_trip = tile_off ? 0 : A / 4;    // whole trips; A % 4 atoms follow
    const ek_cfp cp = (ek_cfp)ctile;
#pragma unroll
    for (int k = 0; k < DIST; ++k)
        EK_ROWS2(k, k)
    EkCChunk<P> cb[2];
    ek_ld_chunk<P>(cb[0], cp);
    // One trip: request the rows of trip t + DIST, then the 4 H chunks of trip t.
    // The first FMA of a chunk waits for that chunk's scalar loads (they return
    // out of order, so the wait is for all of them); the next chunk's are issued
    // right after it and have the other NF - 1 FMAs to arrive.  Chunk q of the
    // trip is atom q / H, candidate pairs (q % H) P ..; chunks alternate between
    // the two buffers (4 H is even: the parity carries over from trip to trip).
#define EK_TRIP2(K, TT)                                                        \
    {                                                                          \
        EK_ROWS2(((K) + DIST) % NB, (TT) + DIST)                               \
        const ek_cfp ca = cp + (size_t)(4 * (TT)) * (H * CH);                  \
        _Pragma("unroll") for (int q = 0; q < 4 * H; ++q) {                    \
            const int aa = q / H, p0 = (q % H) * P;                            \
            if (aa & 1)                                                        \
                ek_chunk_fma<NP, P, true>(s2, p0, X[K][aa / 2], Y[K][aa / 2],  \
                                          Z[K][aa / 2], cb[q & 1], 0, 1);      \
            else                                                               \
                ek_chunk_fma<NP, P, false>(s2, p0, X[K][aa / 2], Y[K][aa / 2], \
                                           Z[K][aa / 2], cb[q & 1], 0, 1);     \
            __builtin_amdgcn_sched_barrier(0);                                 \
            ek_ld_chunk<P>(cb[(q + 1) & 1], ca + (q + 1) * CH);                \
            __builtin_amdgcn_sched_barrier(0);                                 \
            if (aa & 1)                                                        \
                ek_chunk_fma<NP, P, true>(s2, p0, X[K][aa / 2], Y[K][aa / 2],  \
                                          Z[K][aa / 2], cb[q & 1], 1, NF);     \
            else                                                               \
                ek_chunk_fma<NP, P, false>(s2, p0, X[K][aa / 2], Y[K][aa / 2], \
                                           Z[K][aa / 2], cb[q & 1], 1, NF);    \
        }                                                                      \
    }
    // Whole groups of NB trips run without a branch inside, and the first group
    // is peeled: the loop is then entered in the very state it leaves at its
    // end, so the load-counter waits the compiler derives at the loop head are
    // the exact ones (rows two trips ahead stay in flight) instead of the
    // cautious join of prologue and loop states.
    const int n_grp = n_trip / NB;
    int t0 = 0;
    if (n_grp > 0) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
            EK_TRIP2(k, k)
        t0 = NB;
        for (int g = 1; g < n_grp; ++g) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                EK_TRIP2(k, t0 + k)
            t0 += NB;
        }
    }
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
        if (t0 + k < n_trip)            // wave-uniform
            EK_TRIP2(k, t0 + k)
    }
#undef EK_TRIP2
    // the A % 4 atoms after the last whole trip
    for (int a = tile_off ? A : 4 * n_trip; a < A; ++a) {
        const int so = a * (3 * EK_TILE * 4);
        ek_v2f x, y, z;
        x[0] = EK_LD(so, 0);
        y[0] = EK_LD(so, 1);
        z[0] = EK_LD(so, 2);
        x[1] = 0.f;
        y[1] = 0.f;
        z[1] = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            ek_ld_chunk<P>(cb[0], cp + ((size_t)a * H + h) * CH);
            ek_chunk_fma<NP, P, false>(s2, h * P, x, y, z, cb[0], 0, NF);
        }
    }
#undef EK_ROWS2
#undef EK_LD

    if (!UPD) {
        if (f < n) {
#pragma unroll
            for (int c = 0; c < T; ++c) {
                __builtin_amdgcn_sched_barrier(0);
                if (c < teff) {
                    float S[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j)
                        S[j] = s2[c / 2][j][c & 1];
                    vecs[(size_t)c * n_pad + f] =
                        ek_rmsd_from_S(S, Gf, ctrace[c], A);
                }
            }
        }
        return;
    }
    float bestv = -__builtin_inff();
    uint32_t besti = 0xffffffffu;
    if (f < n) {
        // candidate 0: the new center of this iteration (kcenters.py:298-306)
        float S0[9];
#pragma unroll
        for (int j = 0; j < 9; ++j)
            S0[j] = s2[0][j][0];
        float cur = cur0;
        const float d0 = (tm & 1u) ? ek_rmsd_from_S_below(S0, Gf, ctrace[0], A, cur)
                                   : __builtin_inff();
        if (d0 < cur) {
            cur = d0;
            lab = label;
        }
        if (lab >= 0) {
            dist[f] = cur;
            assign[f] = lab;
        }
        bestv = cur;
        besti = (uint32_t)f;
        if (order && own) {
            ek_coh_store(&fz.rows[own].cur, cur);
            ek_coh_store(&fz.rows[own].valid, 1);
        }
        // FUSE: almost every kept distance is +inf (abandoned: it cannot be below
        // the frame's own), and 4-byte-per-frame stores trickling into T - 1
        // arrays cost the pass 12 % (28 MB of writes against 3.6 GB of reads:
        // every small write burst turns the HBM bus around).  So a wave stores a
        // vector only if one of its frames has a finite value, and one word per
        // wave says which it stored; readers take the others as +inf.
        uint32_t wmask = 0;
#pragma unroll
        for (int c = 1; c < T; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            if (c < teff) {
                float S[9];
#pragma unroll
                for (int j = 0; j < 9; ++j)
                    S[j] = s2[c / 2][j][c & 1];
                const float dc = ((tm >> c) & 1u)
                                     ? ek_rmsd_from_S_below(S, Gf, ctrace[c], A, cur)
                                     : __builtin_inff();
                if (FUSE) {
                    if (__ballot(dc != __builtin_inff())) {     // wave-uniform
                        vecs[(size_t)(c - 1) * n_pad + f] = dc;
                        wmask |= 1u << c;
                    }
                } else {
                    vecs[(size_t)(c - 1) * n_pad + f] = dc;
                }
                if (order && own)
                    ek_coh_store(&fz.rows[own].d[c], dc);
            }
        }
        if (FUSE && (f & (EK_WAVE - 1)) == 0)
            fz.vmask[f >> 6] = wmask;
    }
    ek_wave_argmax(bestv, besti);
    const int lane = tid & (EK_WAVE - 1), wave = tid / EK_WAVE;
    if (lane == 0) {
        red_v[wave] = bestv;
        red_i[wave] = besti;
    }
    __syncthreads();
    if (tid == 0) {
        float v = red_v[0];
        uint32_t i = red_i[0];
#pragma unroll
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[w], red_i[w], v, i)) {
                v = red_v[w];
                i = red_i[w];
            }
        blockmax[blockIdx.x].val = v;
        blockmax[blockIdx.x].idx = i;
    }
    if (order) {
        // The last workgroup to get here works out the order in which the
        // candidates would be accepted (ek_chain.hip, step 1) from the
        // candidate frames' own rows of the new distance vectors (written by
        // the workgroups that own the candidate frames, as coherent stores:
        // ek_reduce.h)
        __shared__ EkChainRow rows[EK_MAX_CANDS];
        __shared__ int s_chain[EK_MAX_CANDS];
        __shared__ int s_cn;
        __shared__ bool early_last;
        bool last;
        if (owner_blk) {            // after its rows are in place
            last = ek_arrive_last(fz.tick);
        } else {
            if (tid == 0)
                early_last = ticket == gridDim.x - 1;
            __syncthreads();
            last = early_last;
        }
        if (!last)
            return;
        for (int e = tid; e < T * (T + 2); e += EK_BLOCK) {
            const int j = e / (T + 2), u = e % (T + 2);
            const bool live = j >= 1 && j < teff;
            if (u == 0)
                rows[j].cur = live ? ek_coh_load(&fz.rows[j].cur) : 0.f;
            else if (u == 1)
                rows[j].valid = live ? ek_coh_load(&fz.rows[j].valid) : 0;
            else
                rows[j].d[u - 2] = (live && u - 2 >= 1 && u - 2 < teff)
                                       ? ek_coh_load(&fz.rows[j].d[u - 2])
                                       : 0.f;
        }
        __syncthreads();
        if (tid < EK_MAX_CANDS)     // the rows are one round's: clear them
            fz.rows[tid].valid = 0;
        if (tid < EK_WAVE)
            ek_chain_simulate_wave(plan, rows, s_chain, &s_cn);
        __syncthreads();
        if (tid < EK_MAX_CANDS)
            fz.ord->cand[tid] = tid < s_cn ? s_chain[tid] : 0;
        if (tid == 0) {
            fz.ord->n = s_cn;
            // candidate 0 is a center now (kcenters.py:306-309): the count and
            // the history move when its distances are in, not when it was planned
            fz.hist[label].gidx = plan->gidx[0];
            fz.hist[label].dist = plan->maxdist[0];
            fz.hist[label].set = 1;
            fz.ctl->n_done = label + 1;
            fz.ctl->n_rounds = fz.ctl->n_rounds + 1;
            *fz.tick = 0;
        }
    }
}

size_t ek_ctile_bytes(int A) { return ek_ctile_floats(A) * sizeof(float); }

#define EK_BY_T(TT, CALL)                                                      \
    do {                                                                       \
        if ((TT) == 8) {                                                       \
            constexpr int T_ = 8;                                              \
            CALL;                                                              \
        } else {                                                               \
            constexpr int T_ = 4;                                              \
            CALL;                                                              \
        }                                                                      \
    } while (0)

// with_order: single shard, the last workgroup works out the presumed order
void ek_launch_round_pass(const EkRound &r, hipStream_t s, bool with_order)
{
    if (r.n <= 0)
        return;
    const unsigned blocks = (unsigned)((r.n + EK_BLOCK - 1) / EK_BLOCK);
    EkFuse fz;
    // (round 6: a round of 16 takes its per-prefix maxima in the pass; the presumed
    // order is then the candidates' own -- ek_round_next_kernel / ek_ms_plan_kernel
    // write it -- and no workgroup works one out at the end of the pass)
    const bool sweep = r.sweep && r.T == 16;
    fz.pend = r.pend;
    fz.ord = (with_order && !sweep) ? r.ord : nullptr;
    if (sweep) {
        fz.sweep_pm = r.pm;
        fz.sweep_nb = (int)blocks;
        // (the maxima per 64 frames: where the pick reads them, ek_round_chain_kernel)
        fz.sweep_fm = (r.fm && 4 * (int64_t)blocks <= 8 * 1024) ? r.fm : nullptr;
    }
    fz.tick = r.tick + 0;
    fz.goff = r.goff;
    fz.hist = r.hist;
    fz.ctl = r.ctl;
    fz.rows = r.rows;
    fz.vmask = r.vmask;
    fz.tmask = r.tmask;
    if (r.T >= 16) {        // (32: the stream twice, ek_pass16.hip)
        ek_launch_pass16(true, r.qtiles, r.G, r.dist, r.assign, r.vecs, r.n, r.n_pad,
                         r.A, r.ctile, r.ctrace, r.plan, r.blockmax, fz, s, r.T == 32);
        return;
    }
    EK_BY_T(r.T, hipLaunchKernelGGL((ek_pass2_kernel<T_, true, true>), dim3(blocks),
                                    dim3(EK_BLOCK), 0, s, r.tiles, r.G, r.dist,
                                    r.assign, r.vecs, r.n, r.n_pad, r.A, r.ctile,
                                    r.ctrace, r.plan, r.blockmax, fz));
}

// the one-launch-per-step form: lay the candidates out, then stream the frames
void ek_launch_pass(int T, const float *tiles, const float *qtiles, const double *G,
                    float *dist, int32_t *assign, float *vecs, int64_t n,
                    int64_t n_pad, int A, const unsigned char *recs,
                    const EkPlan *plan, EkBlockMax *blockmax, float *ctile,
                    double *ctrace, hipStream_t s)
{
    if (n <= 0)
        return;
    const unsigned blocks = (unsigned)((n + EK_BLOCK - 1) / EK_BLOCK);
    const unsigned cb = (unsigned)((ek_ctile_atoms(A) * 3 * T + EK_BLOCK - 1) /
                                   EK_BLOCK);
    if (T == 16) {
        hipLaunchKernelGGL((ek_ctile_kernel<16>), dim3(cb), dim3(EK_BLOCK), 0, s,
                           recs, plan, A, ctile, ctrace);
        ek_launch_pass16(false, qtiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                         ctrace, plan, blockmax, EkFuse(), s);
        return;
    }
    EK_BY_T(T, {
        hipLaunchKernelGGL((ek_ctile_kernel<T_>), dim3(cb), dim3(EK_BLOCK), 0, s,
                           recs, plan, A, ctile, ctrace);
        hipLaunchKernelGGL((ek_pass2_kernel<T_, true, false>), dim3(blocks),
                           dim3(EK_BLOCK), 0, s, tiles, G, dist, assign, vecs, n,
                           n_pad, A, ctile, ctrace, plan, blockmax, EkFuse());
    });
}

// distances of every frame to `count` records (count <= 8), nothing else:
// vecs[j][f] = rmsd(frame f, record j).  Used to prefetch PAM proposals.
__global__ void ek_plan_fixed_kernel(EkPlan *__restrict__ plan, int count)
{
    if (threadIdx.x != 0)
        return;
    plan->go = 1;
    plan->teff = count;
    plan->label = 0;
    for (int j = 0; j < EK_MAX_CANDS; ++j)
        plan->src[j] = j;
}

void ek_launch_pass_dist(int count, const float *tiles, const double *G,
                         float *vecs, int64_t n, int64_t n_pad, int A,
                         const unsigned char *recs, EkPlan *plan,
                         float *ctile, double *ctrace, hipStream_t s, bool prepared)
{
    if (n <= 0 || count <= 0)
        return;
    const int T = ek_pass_dist_T(count);
    // prepared: plan, ctile and ctrace already hold these records
    // (ek_launch_pam_setup)
    if (!prepared)
        hipLaunchKernelGGL(ek_plan_fixed_kernel, dim3(1), dim3(EK_WAVE), 0, s, plan,
                           count);
    const unsigned blocks = (unsigned)((n + EK_BLOCK - 1) / EK_BLOCK);
    const unsigned cb = (unsigned)((ek_ctile_atoms(A) * 3 * T + EK_BLOCK - 1) /
                                   EK_BLOCK);
    // a list too short to fill the chip (PAM's touched frames): one wave
    // per workgroup
    const bool thin = blocks < 1024;
    const dim3 pg(thin ? (unsigned)((n + EK_WAVE - 1) / EK_WAVE) : blocks);
    const dim3 pb(thin ? EK_WAVE : EK_BLOCK);
    EK_BY_T(T, {
        if (!prepared)
            hipLaunchKernelGGL((ek_ctile_kernel<T_>), dim3(cb), dim3(EK_BLOCK), 0, s,
                               recs, plan, A, ctile, ctrace);
        hipLaunchKernelGGL((ek_pass2_kernel<T_, false, false>), pg, pb, 0, s, tiles,
                           G, nullptr, nullptr, vecs, n, n_pad, A, ctile, ctrace,
                           plan, nullptr, EkFuse());
    });
}

// ---------------------------------------------------------------------------
// per-workgroup maxima of the current distances (used to seed a run)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_blockmax_kernel(const float *__restrict__ dist, int64_t n,
                   EkBlockMax *__restrict__ blockmax)
{
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (f < n) {
        v = dist[f];
        i = (uint32_t)f;
    }
    ek_wave_argmax(v, i);
    if ((tid & 63) == 0) {
        red_v[tid / 64] = v;
        red_i[tid / 64] = i;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[w], red_i[w], v, i)) {
                v = red_v[w];
                i = red_i[w];
            }
        blockmax[blockIdx.x].val = v;
        blockmax[blockIdx.x].idx = i;
    }
}

void ek_launch_blockmax(const float *dist, int64_t n, EkBlockMax *blockmax,
                        hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_blockmax_kernel,
                       dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, dist, n, blockmax);
}

// ---------------------------------------------------------------------------
// single-workgroup reductions over the per-workgroup maxima
// ---------------------------------------------------------------------------
// local farthest point -> 16-byte header (what a rank contributes to the
// per-center exchange)
__global__ void __launch_bounds__(EK_RED_THREADS)
ek_localmax_kernel(const EkBlockMax *__restrict__ blockmax, int nb,
                   int64_t global_offset, EkMaxHdr *__restrict__ out)
{
    float v;
    uint32_t i;
    int b;
    ek_block_argmax(blockmax, nb, nullptr, v, i, b);
    if (threadIdx.x == 0) {
        out->maxdist = v;
        out->valid = (i != 0xffffffffu) ? 1 : 0;
        out->gidx = (i != 0xffffffffu) ? global_offset + (int64_t)i : -1;
    }
}

void ek_launch_localmax(const EkBlockMax *blockmax, int nb,
                        int64_t global_offset, EkMaxHdr *out, hipStream_t s)
{
    hipLaunchKernelGGL(ek_localmax_kernel, dim3(1), dim3(EK_RED_THREADS), 0, s,
                       blockmax, nb, global_offset, out);
}

// decide whether the global farthest point is a stored candidate
__global__ void __launch_bounds__(EK_WAVE)
ek_check_kernel(const EkMaxHdr *__restrict__ hdrs, int n_hdrs, double cutoff,
                EkPlan *__restrict__ plan, EkHist *__restrict__ hist,
                EkCtl *__restrict__ ctl)
{
    if (threadIdx.x != 0)
        return;
    plan->apply = -1;
    if (!plan->go || plan->miss)
        return;
    plan->miss = 1;                      // until proven a hit
    if (ctl->stopped || ctl->n_done >= ctl->limit)
        return;
    int best = -1;
    float bv = 0.f;
    long long bg = 0;
    for (int r = 0; r < n_hdrs; ++r) {
        if (!hdrs[r].valid)
            continue;
        if (best < 0 || hdrs[r].maxdist > bv ||
            (hdrs[r].maxdist == bv && hdrs[r].gidx < bg)) {
            best = r;
            bv = hdrs[r].maxdist;
            bg = hdrs[r].gidx;
        }
    }
    if (best < 0)
        return;
    ctl->last_max = bv;
    if (!((double)bv > cutoff)) {        // kcenters.py:217
        ctl->stopped = 1;
        return;
    }
    for (int j = 1; j < plan->teff; ++j) {
        if (plan->gidx[j] == bg && !(plan->used & (1u << j))) {
            const int label = ctl->n_done;
            plan->apply = j;
            plan->apply_label = label;
            plan->used |= 1u << j;
            plan->miss = 0;
            hist[label].gidx = bg;
            hist[label].dist = bv;
            hist[label].set = 1;
            ctl->n_done = label + 1;
            return;
        }
    }
}

// single shard: local arg-max and the decision in one launch
__global__ void __launch_bounds__(EK_RED_THREADS)
ek_localmax_check_kernel(const EkBlockMax *__restrict__ blockmax, int nb,
                         int64_t global_offset, double cutoff,
                         EkPlan *__restrict__ plan, EkHist *__restrict__ hist,
                         EkCtl *__restrict__ ctl)
{
    // nothing to decide: the round did not run or already missed
    if (!plan->go || plan->miss) {
        if (threadIdx.x == 0)
            plan->apply = -1;
        return;
    }
    float v;
    uint32_t i;
    int b;
    ek_block_argmax(blockmax, nb, nullptr, v, i, b);
    if (threadIdx.x != 0)
        return;
    plan->apply = -1;
    plan->miss = 1;
    if (ctl->stopped || ctl->n_done >= ctl->limit || i == 0xffffffffu)
        return;
    ctl->last_max = v;
    if (!((double)v > cutoff)) {
        ctl->stopped = 1;
        return;
    }
    const long long bg = global_offset + (long long)i;
    for (int j = 1; j < plan->teff; ++j) {
        if (plan->gidx[j] == bg && !(plan->used & (1u << j))) {
            const int label = ctl->n_done;
            plan->apply = j;
            plan->apply_label = label;
            plan->used |= 1u << j;
            plan->miss = 0;
            hist[label].gidx = bg;
            hist[label].dist = v;
            hist[label].set = 1;
            ctl->n_done = label + 1;
            return;
        }
    }
}

void ek_launch_localmax_check(const EkBlockMax *blockmax, int nb,
                              int64_t global_offset, double cutoff,
                              EkPlan *plan, EkHist *hist, EkCtl *ctl,
                              hipStream_t s)
{
    hipLaunchKernelGGL(ek_localmax_check_kernel, dim3(1), dim3(EK_RED_THREADS),
                       0, s, blockmax, nb, global_offset, cutoff, plan, hist,
                       ctl);
}

void ek_launch_check(const EkMaxHdr *hdrs, int n_hdrs, double cutoff,
                     EkPlan *plan, EkHist *hist, EkCtl *ctl, hipStream_t s)
{
    hipLaunchKernelGGL(ek_check_kernel, dim3(1), dim3(EK_WAVE), 0, s, hdrs,
                       n_hdrs, cutoff, plan, hist, ctl);
}

// apply a stored distance vector (kcenters.py:304-306 on kept distances)
__global__ void __launch_bounds__(EK_BLOCK)
ek_apply_kernel(const float *__restrict__ vecs, const double *__restrict__ G,
                int64_t n, int64_t n_pad, int A, float *__restrict__ dist,
                int32_t *__restrict__ assign, const EkPlan *__restrict__ plan,
                EkBlockMax *__restrict__ blockmax)
{
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    const int j = plan->apply;
    if (j < 1)
        return;
    const int label = plan->apply_label;
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (f < n) {
        const float d = vecs[(size_t)(j - 1) * n_pad + f];
        float cur = dist[f];
        if (d < cur) {
            cur = d;
            dist[f] = d;
            assign[f] = label;
        }
        v = cur;
        i = (uint32_t)f;
    }
    ek_wave_argmax(v, i);
    if ((tid & 63) == 0) {
        red_v[tid / 64] = v;
        red_i[tid / 64] = i;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[w], red_i[w], v, i)) {
                v = red_v[w];
                i = red_i[w];
            }
        blockmax[blockIdx.x].val = v;
        blockmax[blockIdx.x].idx = i;
    }
}

void ek_launch_apply(const float *vecs, const double *G, int64_t n,
                     int64_t n_pad, int A, float *dist, int32_t *assign,
                     const EkPlan *plan, EkBlockMax *blockmax, hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_apply_kernel,
                       dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, vecs, G, n, n_pad, A, dist, assign,
                       plan, blockmax);
}

// ---------------------------------------------------------------------------
// the shard's T candidate records for the next round
// ---------------------------------------------------------------------------
// Any far frames would do -- correctness never depends on the guesses -- but the
// more of the following farthest points are among them, the fewer passes the
// fit needs.  So the pick plays the k-centers game ahead on the shard's
// EK_TOP_M farthest workgroup maxima: their pairwise distances are computed
// (EK_TOP_M^2/2 pairs, nothing next to a pass), and the candidates are taken
// greedily -- the frame with the largest remaining distance, after which every
// other one's distance is lowered by its distance to it -- exactly the order in
// which the fit itself would take them if no frame outside the set interfered.
// Record 0 is therefore the shard's first-index arg-max, as the protocol needs.
// (In a CPU model of the rounds this raised the centers per pass from 5.1 to 6.6
// with 32 maxima and 6.8 with 64, against picking the top maxima with distinct
// labels.)
size_t ek_top_scratch_bytes(int A)
{
    // EkTop | EkMsPub | coords [M][3A] f32 | traces [M] f64 | D [L][L] f32 (L = the
    // fused round's list, >= M)
    return EK_TOP_HEAD + (size_t)EK_TOP_M * 3 * A * sizeof(float) +
           EK_TOP_M * sizeof(double) + (size_t)EK_LIST_M * EK_LIST_M * sizeof(float);
}

__global__ void __launch_bounds__(EK_RED_THREADS)
ek_pick_top_kernel(const EkBlockMax *__restrict__ blockmax, int nb,
                   EkTop *__restrict__ top, const int32_t *__restrict__ assign)
{
    extern __shared__ uint32_t skip[];       // bitmap over workgroups
    ek_pick_top_body(blockmax, nb, top, skip, assign);
}

// their centred coordinates and traces; every read is its own cache line, so
// they are spread over many workgroups (one thread per value)
__global__ void __launch_bounds__(EK_BLOCK)
ek_top_gather_kernel(const float *__restrict__ tiles,
                     const double *__restrict__ G, int A,
                     unsigned char *__restrict__ scr)
{
    const EkTop *top = (const EkTop *)scr;
    const int j = blockIdx.y;
    if (j >= top->n)
        return;
    const int r = blockIdx.x * EK_BLOCK + threadIdx.x;
    const uint32_t i = top->idx[j];
    if (r == 0)
        ek_top_traces(scr, A)[j] = G[i];
    if (r >= 3 * A)
        return;
    const float *p = tiles + (size_t)(i / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (i % EK_TILE);
    ek_top_coords(scr)[(size_t)j * 3 * A + r] = p[(size_t)r * EK_TILE];
}

// their pairwise distances, one wave per pair (lanes strided over the atoms: the
// values only steer the guesses, their summation order is free)
__global__ void __launch_bounds__(EK_BLOCK)
ek_top_pair_kernel(int A, unsigned char *__restrict__ scr)
{
    const EkTop *top = (const EkTop *)scr;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int w = blockIdx.x * (EK_BLOCK / EK_WAVE) + threadIdx.x / EK_WAVE;
    const int i = w / EK_TOP_M, j = w % EK_TOP_M;
    if (i >= j || j >= top->n)
        return;
    const float *x = ek_top_coords(scr) + (size_t)i * 3 * A;
    const float *y = ek_top_coords(scr) + (size_t)j * 3 * A;
    float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = lane; a < A; a += EK_WAVE) {
        const float x0 = x[3 * a], x1 = x[3 * a + 1], x2 = x[3 * a + 2];
        const float y0 = y[3 * a], y1 = y[3 * a + 1], y2 = y[3 * a + 2];
        S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
        S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
        S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            S[q] += __shfl_xor(S[q], off, 64);
    if (lane == 0) {
        const double *Gt = ek_top_traces(scr, A);
        const float d = ek_rmsd_from_S(S, Gt[i], Gt[j], A);
        float *D = ek_top_D(scr, A);
        D[i * EK_TOP_M + j] = d;
        D[j * EK_TOP_M + i] = d;
    }
}

__global__ void __launch_bounds__(EK_BLOCK)
ek_top_records_kernel(int A, int T, int64_t global_offset,
                      unsigned char *__restrict__ scr,
                      unsigned char *__restrict__ recs, EkCtl *__restrict__ ctl)
{
    ek_top_records_body(A, T, global_offset, scr, recs, ctl);
}

void ek_launch_pickT(const EkBlockMax *blockmax, int nb, const float *tiles,
                     const double *G, const int32_t *assign, int A, int T,
                     int64_t global_offset, unsigned char *recs, EkCtl *ctl,
                     unsigned char *scratch, hipStream_t s)
{
    const size_t lds = (size_t)((nb + 31) / 32 + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL(ek_pick_top_kernel, dim3(1), dim3(EK_RED_THREADS), lds, s,
                       blockmax, nb, (EkTop *)scratch, assign);
    hipLaunchKernelGGL(ek_top_gather_kernel,
                       dim3((unsigned)((3 * A + EK_BLOCK - 1) / EK_BLOCK),
                            (unsigned)EK_TOP_M),
                       dim3(EK_BLOCK), 0, s, tiles, G, A, scratch);
    hipLaunchKernelGGL(ek_top_pair_kernel,
                       dim3((unsigned)(EK_TOP_M * EK_TOP_M /
                                       (EK_BLOCK / EK_WAVE))),
                       dim3(EK_BLOCK), 0, s, A, scratch);
    hipLaunchKernelGGL(ek_top_records_kernel, dim3(1), dim3(EK_BLOCK), 0, s, A, T,
                       global_offset, scratch, recs, ctl);
}
