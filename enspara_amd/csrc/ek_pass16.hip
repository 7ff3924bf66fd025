// ek_pass16.hip -- the pass of a k-centers round with 16 candidate centers,
// through the matrix cores.
//
// One frame against ONE center is a matrix-vector product (1.5 flop per byte:
// HBM-bound, ek_kcenters.hip).  Every frame against SIXTEEN candidates is a
// dense contraction -- S[frame][cand][i][j] = sum_a x[frame][a][i] y[cand][a][j],
// M = frames x 3, N = candidates x 3, K = atoms -- with 24 flop per byte: its
// vector-FMA form (ek_pass2_kernel, 8 candidates as scalar operands) already
// runs at the packed-FMA rate's edge at 16, starved by its scalar loads (two
// waves per SIMD cannot hide a scalar-cache miss every 36 FMAs: 1.08 ms per
// 10^6 x 300 pass measured, against 0.64 ms for 8 candidates).  Here the same
// sums run on v_mfma_f32_16x16x1_f32:
//
//   * 4 blocks of 16 x 16 per instruction, ONE k per instruction: the A
//     operand is a frame row exactly as the tile layout delivers it (lane =
//     frame, 64 frames = 4 blocks of 16), the B operand 16 candidates'
//     coordinate j of the same atom (replicated over the blocks); nine
//     instructions per atom (i, j = x, y, z), 144 accumulators per lane.
//   * every accumulator is the IEEE FMA chain over the atoms in ascending
//     order -- one k per instruction, a fused multiply-add per element
//     (tools/probes/mfma16_probe.hip: bit-identical to fmaf in k order) -- so
//     the distances are the bits the one-center kernel and the CPU checker
//     produce.
//   * a lane ends up with 16 (frame, candidate) pairs, all of ONE candidate
//     (l % 16) and 16 frames: it solves their quartics (early stop against the
//     frame's distance BEFORE this pass), the results cross an LDS transposition
//     into the frame-per-lane domain, and from there on the epilogue is
//     ek_pass2_kernel's: candidate 0 updates the state (kcenters.py:298-306),
//     the others' distances are kept (where a wave holds a finite one), the
//     per-workgroup arg-max, and -- single shard -- the presumed order.
//
// What bounds it (10^6 x 300, builds without one part each): the matrix
// instructions alone 0.64 ms -- 86 GFLOP at the 135-145 TFLOP/s the f32 matrix
// pipe sustains --, streaming the rows beside them +0.18 ms (3.7 GB: both HBM
// and the matrix pipe would have to run at ~90 % at once), the 16 quartic
// solves per frame +0.14 ms (they do not overlap with another wave's matrix
// instructions, see the stagger below): 0.95 ms per pass, 59 us per candidate
// against 80 us for the 8-candidate vector form.
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"
#include "ek_chain_dev.h"

typedef float ek_v16f __attribute__((ext_vector_type(16)));
typedef float ek_v4f __attribute__((ext_vector_type(4)));

#ifndef EK_P16_DIST
#define EK_P16_DIST 2           // trips (of 4 atoms) the row loads run ahead
#endif

// LDS per wave: the frames' traces and current distances, then the 16 x 64
// table of new distances (row stride 65: the writers of a register are 4
// frames apart)
#define EK_P16_DSTRIDE 65
// the workgroups resident at the start of a launch (256 CUs x 2), and how long
// the second of a CU waits before it starts: ~20 us in s_sleep(127) units of
// 64 x 127 cycles
#define EK_P16_FIRST_WGS 512
#ifndef EK_P16_STAGGER
#define EK_P16_STAGGER 6
#endif

template <bool FUSE>
__global__ void __launch_bounds__(EK_BLOCK, 2)
ek_pass16_kernel(const float *__restrict__ tiles, const double *__restrict__ G,
                 float *__restrict__ dist, int32_t *__restrict__ assign,
                 float *__restrict__ vecs, int64_t n, int64_t n_pad, int A,
                 const float *__restrict__ ctile,
                 const double *__restrict__ ctrace,
                 const EkPlan *__restrict__ plan,
                 EkBlockMax *__restrict__ blockmax, EkFuse fz)
{
    constexpr int T = 16;
    const EkPend *__restrict__ pend = fz.pend;
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    __shared__ double s_G[EK_BLOCK];
    __shared__ float s_cur[EK_BLOCK];
    __shared__ float s_D[EK_BLOCK / EK_WAVE][T * EK_P16_DSTRIDE];
    if (!plan->go)
        return;
    // Two workgroups share a CU (two waves per SIMD).  Launched together and
    // equally long they stay in step for the whole pass: both ask HBM for their
    // rows in the same microseconds, then both solve.  Of the first workgroups
    // to arrive, the one that got the second wave slot of its SIMDs therefore
    // starts ~20 us late (every later workgroup inherits the offset from the
    // one whose slot it takes): 147 -> 133 us per pass at 125 k frames, 2 % at
    // 10^6.  (What this does NOT buy: f32 matrix instructions and another
    // wave's vector instructions do not overlap on a SIMD -- tools/probes/
    // coexec.hip: phases of 2700 MFMAs and 6000 FMAs of two waves take the sum
    // of both, staggered or not -- so the quartic solves are paid in full.)
    if (gridDim.x > EK_P16_FIRST_WGS / 2 && blockIdx.x < EK_P16_FIRST_WGS) {
        // HW_REG_HW_ID (4), bits [3:0]: the wave's slot on its SIMD
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (slot & 1u)
            for (int q = 0; q < EK_P16_STAGGER; ++q)
                __builtin_amdgcn_s_sleep(127);
    }
    const int tid = threadIdx.x;
    const int lane = tid & (EK_WAVE - 1), wave = tid / EK_WAVE;
    const int teff = plan->teff;
    const int label = plan->label;
    static_assert(EK_BLOCK == EK_TILE, "one workgroup per tile");
    const int64_t f0 = (int64_t)blockIdx.x * EK_BLOCK;
    const int64_t f = f0 + tid;
    // the arrival tickets of a fused single-shard round: see ek_pass2_kernel
    const bool order = FUSE && fz.ord != nullptr;
    bool owner_blk = false;
    unsigned int ticket = 0;
    if (order) {
#pragma unroll
        for (int j = 1; j < T; ++j) {
            const int64_t l = plan->gidx[j] - fz.goff - f0;
            owner_blk |= j < teff && l >= 0 && l < EK_BLOCK;
        }
        if (!owner_blk && tid == 0)
            ticket = __hip_atomic_fetch_add(fz.tick, 1u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- this frame's state on the way in ---------------------------------------
    float cur0 = 0.f;
    int32_t lab = -1;           // >= 0: the frame's state changes in this pass
    int own = 0;                // order: this frame is candidate `own` (>= 1)
    {
        double Gf = 0.0;
        if (f < n) {
            Gf = G[f];
            cur0 = dist[f];
            if (FUSE) {
                if (order) {
#pragma unroll
                    for (int j = 1; j < T; ++j)
                        if (j < teff && plan->gidx[j] - fz.goff == f)
                            own = j;
                }
                // kcenters.py:304-306 for the pending chain, in order (a vector
                // is stored only where a wave holds a finite value)
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
        s_G[tid] = Gf;
        s_cur[tid] = cur0;      // read back after the loop, by this wave only
    }

    const float *tb = tiles + (size_t)(f0 / EK_TILE) * 3 * (size_t)A * EK_TILE;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)tb, 0, 3 * A * EK_TILE * 4, 0x00020000);
    const int vo = tid * 4;
    // non-temporal (aux bit 1): the frame stream is read once per pass
#define EK_LD(SO, K)                                                           \
    __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(            \
                                  rs, vo + (K) * (EK_TILE * 4), (SO), 2))
    ek_v16f acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[q][r] = 0.f;

    // rows (the HBM stream) run DR trips of 4 atoms ahead of the matrix
    // instructions, the candidates (a few cache lines out of L2) DC = DR - 1
    constexpr int DR = EK_P16_DIST, DC = DR - 1;
    constexpr int NR = DR + 1, NC = DC + 1;
    constexpr int GRP = NR * NC;            // buffers are back in phase (NR, NC coprime)
    float X[NR][4], Y[NR][4], Z[NR][4];     // the rows of a trip of 4 atoms
    ek_v4f Cq[NC][3];                       // its candidates: [xyz] -> 4 atoms
    // candidate tile: [trip][candidate][xyz][atom of the trip] (ek_ctile_index)
    const ek_v4f *cp = (const ek_v4f *)ctile + (size_t)(lane & 15) * 3;
#define EK_ROWS16(B, TR)                                                       \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                            \
        const int so = ((TR) * 4 + e) * (3 * EK_TILE * 4);                     \
        X[B][e] = EK_LD(so, 0);                                                \
        Y[B][e] = EK_LD(so, 1);                                                \
        Z[B][e] = EK_LD(so, 2);                                                \
    }
#define EK_CAND16(B, TR)                                                       \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                              \
        Cq[B][j] = cp[(size_t)(TR) * (16 * 3) + j];
    // three of the nine matrix instructions of one atom: S_ij += x_i * y_j, j fixed
#define EK_MFMA3(J, XX, YY, ZZ, CC)                                            \
    {                                                                          \
        acc[0 + J] = __builtin_amdgcn_mfma_f32_16x16x1f32(XX, CC, acc[0 + J], 0, 0, 0); \
        acc[3 + J] = __builtin_amdgcn_mfma_f32_16x16x1f32(YY, CC, acc[3 + J], 0, 0, 0); \
        acc[6 + J] = __builtin_amdgcn_mfma_f32_16x16x1f32(ZZ, CC, acc[6 + J], 0, 0, 0); \
    }
    const int n_trip = A / 4;           // whole trips; A % 4 atoms follow
    // (trip t is in row buffer t % NR and candidate buffer t % NC)
#pragma unroll
    for (int k = 0; k < DR; ++k) {
        if (k < DC)
            EK_CAND16(k, k)
        __builtin_amdgcn_sched_barrier(0);
        EK_ROWS16(k, k)
        __builtin_amdgcn_sched_barrier(0);
    }
    // One trip: the 36 matrix instructions of trip t with the requests for the
    // candidates of trip t + DC (first) and the rows of trip t + DR spread
    // among them, one load behind every three matrix instructions: a wave
    // issues in order, so a burst of fifteen loads that meets a full memory
    // queue holds the matrix instructions behind it back; spread out, a load
    // that has to wait costs the matrix pipe at most the slack of one
    // instruction.  The order of the requests matters too: vector loads are
    // counted in order, so waiting for a trip's candidates (a few cache lines
    // out of L2) also waits for every load issued before them.  Asked for first
    // and one trip later than the rows of the same trip, they are only behind
    // rows that are needed before they are.
#define EK_TRIP16(K, TT)                                                       \
    {                                                                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                        \
            const int so = (((TT) + DR) * 4 + e) * (3 * EK_TILE * 4);          \
            EK_MFMA3(0, X[(K) % NR][e], Y[(K) % NR][e], Z[(K) % NR][e],        \
                     Cq[(K) % NC][0][e])                                       \
            __builtin_amdgcn_sched_barrier(0);                                 \
            if (e == 0) {                                                      \
                EK_CAND16(((K) + DC) % NC, (TT) + DC)                          \
            }                                                                  \
            X[((K) + DR) % NR][e] = EK_LD(so, 0);                              \
            __builtin_amdgcn_sched_barrier(0);                                 \
            EK_MFMA3(1, X[(K) % NR][e], Y[(K) % NR][e], Z[(K) % NR][e],        \
                     Cq[(K) % NC][1][e])                                       \
            __builtin_amdgcn_sched_barrier(0);                                 \
            Y[((K) + DR) % NR][e] = EK_LD(so, 1);                              \
            __builtin_amdgcn_sched_barrier(0);                                 \
            EK_MFMA3(2, X[(K) % NR][e], Y[(K) % NR][e], Z[(K) % NR][e],        \
                     Cq[(K) % NC][2][e])                                       \
            __builtin_amdgcn_sched_barrier(0);                                 \
            Z[((K) + DR) % NR][e] = EK_LD(so, 2);                              \
            __builtin_amdgcn_sched_barrier(0);                                 \
        }                                                                      \
    }
    int t0 = 0;
    for (; t0 + GRP <= n_trip; t0 += GRP) {
#pragma unroll
        for (int k = 0; k < GRP; ++k)
            EK_TRIP16(k, t0 + k)
    }
#pragma unroll
    for (int k = 0; k < GRP - 1; ++k) {
        if (t0 + k < n_trip)            // wave-uniform
            EK_TRIP16(k, t0 + k)
    }
    // the A % 4 atoms after the last whole trip, one at a time (their rows and
    // candidates fetched for themselves: a product with the zeros past the end
    // is never issued, the chain of an accumulator is exactly the A atoms)
    for (int a = 4 * n_trip; a < A; ++a) {
        const int so = a * (3 * EK_TILE * 4);
        const float x = EK_LD(so, 0), y = EK_LD(so, 1), z = EK_LD(so, 2);
        const float c0 = ctile[ek_ctile_index(16, a, lane & 15, 0)];
        const float c1 = ctile[ek_ctile_index(16, a, lane & 15, 1)];
        const float c2 = ctile[ek_ctile_index(16, a, lane & 15, 2)];
        EK_MFMA3(0, x, y, z, c0)
        EK_MFMA3(1, x, y, z, c1)
        EK_MFMA3(2, x, y, z, c2)
    }
#undef EK_TRIP16
#undef EK_MFMA3
#undef EK_CAND16
#undef EK_ROWS16
#undef EK_LD

    // ---- the lane's 16 pairs: candidate lane % 16, frames 16 b + 4 (lane / 16) + r ----
    {
        const int cand = lane & 15;
        const double Gc = ctrace[cand];
        float *Dw = s_D[wave];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            const int fr = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
            float S[9];
#pragma unroll
            for (int q = 0; q < 9; ++q)
                S[q] = acc[q][r];
            // (the bound is the frame's distance before this pass: what candidate 0
            // makes of it is only known in the frame's own lane, below)
            float d = __builtin_inff();
            // (a frame of the last tile's padding has S = 0: a quadruple root,
            // fifty Newton steps for nothing)
            if (cand < teff && f0 + wave * EK_WAVE + fr < n)
                d = ek_rmsd_from_S_below(S, s_G[wave * EK_WAVE + fr], Gc, A,
                                         s_cur[wave * EK_WAVE + fr]);
            Dw[cand * EK_P16_DSTRIDE + fr] = d;
        }
    }
    __syncthreads();
    // ---- lane = frame again -------------------------------------------------------
    float bestv = -__builtin_inff();
    uint32_t besti = 0xffffffffu;
    if (f < n) {
        const float *Dw = s_D[wave];
        // candidate 0: the new center of this iteration (kcenters.py:298-306)
        float cur = cur0;
        const float d0 = Dw[lane];
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
        // a kept distance that is not below the frame's own can never be used
        // (strict <, against a value that only shrinks): it is +inf, and a wave
        // stores a vector only if one of its frames has a finite value -- one
        // word per wave says which it stored, readers take the others as +inf
        uint32_t wmask = 0;
#pragma unroll
        for (int c = 1; c < T; ++c) {
            if (c < teff) {
                float dc = Dw[c * EK_P16_DSTRIDE + lane];
                if (!(dc < cur))
                    dc = __builtin_inff();
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
        // the last workgroup: the presumed acceptance order (see ek_pass2_kernel)
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
        for (int e = tid; e < EK_MAX_CANDS * (EK_MAX_CANDS + 2); e += EK_BLOCK) {
            const int j = e / (EK_MAX_CANDS + 2), u = e % (EK_MAX_CANDS + 2);
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
            // candidate 0 is a center now (kcenters.py:306-309)
            fz.hist[label].gidx = plan->gidx[0];
            fz.hist[label].dist = plan->maxdist[0];
            fz.hist[label].set = 1;
            fz.ctl->n_done = label + 1;
            fz.ctl->n_rounds = fz.ctl->n_rounds + 1;
            *fz.tick = 0;
        }
    }
}

void ek_launch_pass16(bool fuse, const float *tiles, const double *G, float *dist,
                      int32_t *assign, float *vecs, int64_t n, int64_t n_pad, int A,
                      const float *ctile, const double *ctrace, const EkPlan *plan,
                      EkBlockMax *blockmax, const EkFuse &fz, hipStream_t s)
{
    if (n <= 0)
        return;
    const unsigned blocks = (unsigned)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (fuse)
        hipLaunchKernelGGL((ek_pass16_kernel<true>), dim3(blocks), dim3(EK_BLOCK), 0,
                           s, tiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                           ctrace, plan, blockmax, fz);
    else
        hipLaunchKernelGGL((ek_pass16_kernel<false>), dim3(blocks), dim3(EK_BLOCK), 0,
                           s, tiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                           ctrace, plan, blockmax, fz);
}
