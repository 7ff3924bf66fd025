// ek_pass16.hip -- the pass of a k-centers round with 16 candidate centers,
// through the matrix cores.
//
// One frame against ONE center is a matrix-vector product (1.5 flop per byte:
// HBM-bound, ek_kcenters.hip).  Every frame against SIXTEEN candidates is a
// dense contraction -- S[frame][cand][i][j] = sum_a x[frame][a][i] y[cand][a][j],
// M = frames x 3, N = candidates x 3, K = atoms -- with 24 flop per byte.  Here
// the sums run on v_mfma_f32_16x16x4_f32:
//
//   * 16 frames x 16 candidates x FOUR atoms per instruction: the A operand
//     has (frame l % 16 of a group of 16, atom l / 16 of the trip) in lane l,
//     the B operand (candidate l % 16, atom l / 16); a wave's 64 frames are
//     four groups, so a trip of 4 atoms is 4 groups x 3 x 3 coordinate pairs
//     = 36 instructions, 36 accumulators of 4 registers = 144 per lane.
//   * every accumulator is the IEEE FMA chain over the atoms in ascending
//     order -- a fused multiply-add per element and atom, the four of an
//     instruction in order (tools/probes/mfma16x4_probe.hip: bit-identical to
//     fmaf in k order) -- so the distances are the bits the one-center kernel
//     and the CPU checker produce.  Round 3's form, 16x16x1 in 4 blocks, took
//     ONE atom per instruction and read and wrote all 16 accumulator registers
//     of a coordinate pair for it: four times the register traffic for the
//     same products.  The pass is POWER-limited (profiles/r04/README.md): with
//     a quarter of that traffic the chip holds 2.30 GHz under it instead of
//     2.08, 0.847 -> 0.780 ms.
//   * rows: the frames' QUAD copy (ek_quad_tiles_kernel below) -- one 16-byte
//     load per lane is the lane's A operand for the wave's four groups: three
//     loads of 1 KB per wave and trip of 4 atoms where the frame-minor tiles
//     took twelve of 256 B.  The per-wave queue of outstanding loads was what
//     bounded the bytes in flight (profiles/r03/README.md).
//   * candidates: one 16-byte load per lane is the lane's B operand of one
//     coordinate for FOUR trips (ek_ctile_index) -- three loads per 16 atoms,
//     every byte of them used.
//   * a lane ends up with 16 (frame, candidate) pairs, all of ONE candidate
//     (l % 16) and 16 frames (16 g + 4 (l / 16) + r): the float32 certificate
//     settles the far ones, the rest are solved (early stop against the
//     frame's distance BEFORE this pass), the results cross an LDS transposition
//     into the frame-per-lane domain, and from there on the epilogue is
//     ek_pass2_kernel's: candidate 0 updates the state (kcenters.py:298-306),
//     the others' distances are kept (where a wave holds a finite one), the
//     per-workgroup arg-max, and -- single shard -- the presumed order.
#include <algorithm>
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"
#include "ek_chain_dev.h"

typedef float ek_v4f __attribute__((ext_vector_type(4)));

// LDS per wave: the frames' traces and current distances, then the 16 x 64
// table of new distances (row stride 65: the writers of a register are 4
// frames apart)
#define EK_P16_DSTRIDE 65
// the queue of pairs the float32 certificate does not settle: entries of nine
// floats + (candidate, frame) per wave, 48 bytes apart (three 16-byte stores);
// beyond the capacity a wave solves all its pairs the old way
#ifndef EK_P16_QUEUE
#define EK_P16_QUEUE 1
#endif
#define EK_P16_QCAP 256
#define EK_P16_QSTRIDE 12
// -DEK_P16_LEVEL2=1 (round 5, measured and not kept): the queue's entries take a
// second float32 test before the float64 solve (ek_far_certified2_f32: two Newton
// steps from the closed form's bound, sound on the same adversarial families as
// the first level, host and device).  Pass 0.7906 ms with it, 0.7888 without
// (profiles/r05/level2_ab_1m.log): the dense float64 solve of a wave's handful of
// queued pairs is not what the pass waits for.
#ifndef EK_P16_LEVEL2
#define EK_P16_LEVEL2 0
#endif
// the workgroups resident at the start of a launch (256 CUs x 2), and how long
// the second of a CU waits before it starts: ~20 us in s_sleep(127) units of
// 64 x 127 cycles
#define EK_P16_FIRST_WGS 512
#ifndef EK_P16_STAGGER
#define EK_P16_STAGGER 6
#endif

// LDS traffic between the lanes of ONE wave is in order; this only keeps the
// compiler from moving accesses across the point
__device__ __forceinline__ void ek_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The last workgroup of a fused single-shard pass: the presumed acceptance order
// (see ek_pass2_kernel), by NT threads -- the whole workgroup (t = threadIdx.x)
// or one wave of it (t = lane).
template <int NT, int NC>
__device__ __forceinline__ void ek_p16_order_tail(int t, const EkPlan *plan, const EkFuse &fz,
                                                  EkChainRow *rows, int *s_chain, int *s_cn,
                                                  int teff, int label)
{
    // (all of a thread's loads first, then its LDS stores: every one is a trip to
    // the L2 of ~2.5 us, and behind one another that is what they would cost)
    constexpr int ENT = NC * (NC + 2), PER = (ENT + NT - 1) / NT;
    float val[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = t + q * NT;
        const int j = e / (NC + 2), u = e % (NC + 2);
        const bool live = e < ENT && j >= 1 && j < teff;
        val[q] = 0.f;
        if (u == 0) {
            if (live)
                val[q] = ek_coh_load(&fz.rows[j].cur);
        } else if (u == 1) {
            if (live)
                val[q] = __int_as_float(ek_coh_load(&fz.rows[j].valid));
        } else if (live && u - 2 >= 1 && u - 2 < teff)
            val[q] = ek_coh_load(&fz.rows[j].d[u - 2]);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = t + q * NT;
        const int j = e / (NC + 2), u = e % (NC + 2);
        if (e >= ENT)
            continue;
        if (u == 0)
            rows[j].cur = val[q];
        else if (u == 1)
            rows[j].valid = __float_as_int(val[q]);
        else
            rows[j].d[u - 2] = val[q];
    }
    if (NT == EK_WAVE) ek_wave_sync(); else __syncthreads();
    if (t < EK_MAX_CANDS)       // the rows are one round's: clear them
        fz.rows[t].valid = 0;
    if (t < EK_WAVE)
        ek_chain_simulate_wave(plan, rows, s_chain, s_cn);
    if (NT == EK_WAVE) ek_wave_sync(); else __syncthreads();
    if (t < EK_MAX_CANDS)
        fz.ord->cand[t] = t < *s_cn ? s_chain[t] : 0;
    if (t == 0) {
        fz.ord->n = *s_cn;
        // candidate 0 is a center now (kcenters.py:306-309)
        fz.hist[label].gidx = plan->gidx[0];
        fz.hist[label].dist = plan->maxdist[0];
        fz.hist[label].set = 1;
        fz.ctl->n_done = label + 1;
        fz.ctl->n_rounds = fz.ctl->n_rounds + 1;
        *fz.tick = 0;
    }
}

// MODE 0: a round of 16.  A round of 32 candidates (round 5) is two launches over
// the frames behind one plan and one chain: MODE 1 takes candidates 0 .. 15 --
// everything a round of 16 does except the presumed order, which needs the
// other sixteen columns of the candidate frames' rows --, MODE 2 candidates
// 16 .. 31 against the state MODE 1 left: no pending chain (applied), no update
// (candidate 0 was MODE 1's), sixteen kept vectors more (slots 15 .. 30, mask
// bits 16 .. 31), the rows' other half, and the presumed order at its end.
template <bool FUSE, int MODE>
__global__ void __launch_bounds__(EK_BLOCK, 2)
ek_pass16_kernel(const float *__restrict__ qtiles, const double *__restrict__ G,
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
    __shared__ unsigned int s_arrive, s_ticket;
    __shared__ double s_G[EK_BLOCK];
    __shared__ float s_cur[EK_BLOCK];
    __shared__ float s_D[EK_BLOCK / EK_WAVE][T * EK_P16_DSTRIDE];
#if EK_P16_QUEUE
    __shared__ float s_t[EK_BLOCK];     // the frame's share of the certificate's far test
    __shared__ __attribute__((aligned(16)))
    uint32_t s_Q[EK_BLOCK / EK_WAVE][EK_P16_QCAP * EK_P16_QSTRIDE];
#endif
    if (!plan->go)
        return;
    constexpr int CB = MODE == 2 ? 16 : 0;      // the first candidate of this launch
    constexpr int NC = MODE == 0 ? 16 : EK_MAX_CANDS;   // candidates of the round, at most
    if (MODE == 2) {
        if (plan->teff <= 16)
            return;             // (the round has no second half)
        ctile += ek_ctile_half_floats(A);
        ctrace += 16;
    }
    // Two workgroups share a CU (two waves per SIMD).  Launched together and
    // equally long they stay in step for the whole pass: both ask HBM for their
    // rows in the same microseconds, then both solve.  Of the first workgroups
    // to arrive, the one that got the second wave slot of its SIMDs therefore
    // starts ~20 us late (every later workgroup inherits the offset from the
    // one whose slot it takes): 147 -> 133 us per pass at 125 k frames, 2 % at
    // 10^6.  (What the wave stamps of the measurement build say about the two
    // waves of a SIMD, -DEK_P16_STATS, profiles/r04/README.md: a wave's 2700
    // matrix instructions take 88.5 k cycles -- 32.8 each, the pipe's rate --
    // while the other wave is outside its loop, whatever that one executes; the
    // phases outside the loop are waits, ~100 k cycles a tile, and the pipe idles
    // whenever both waves are in one.)
    if (gridDim.x > EK_P16_FIRST_WGS / 2 && blockIdx.x < EK_P16_FIRST_WGS) {
        // HW_REG_HW_ID (4), bits [3:0]: the wave's slot on its SIMD
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (slot & 1u)
            for (int q = 0; q < EK_P16_STAGGER; ++q)
                __builtin_amdgcn_s_sleep(127);
    }
    const int tid = threadIdx.x;
    const int lane = tid & (EK_WAVE - 1), wave = tid / EK_WAVE;
#ifdef EK_P16_STATS     // (measurement build: where a wave's cycles go, s_memtime)
    unsigned long long stamp[6];
#define EK_P16_STAMP(k) stamp[k] = __builtin_amdgcn_s_memtime()
#else
#define EK_P16_STAMP(k)
#endif
#ifndef EK_P16_PRIO
#define EK_P16_PRIO 0
#endif
    if (EK_P16_PRIO)
        __builtin_amdgcn_s_setprio(EK_P16_PRIO);
    EK_P16_STAMP(0);
#ifdef EK_P16_STATS
    const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    __shared__ unsigned int s_endt[EK_BLOCK / EK_WAVE];
    // this CU's entry of the table of workgroup ends: XCC_ID (register 20), and
    // SE_ID [15:13], SH_ID [12], CU_ID [11:8] of HW_ID (register 4)
    const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    unsigned int *cu_end = fz.tick + 256 + 64 * 16 + ((xcc & 7u) << 8 | ((hwid >> 8) & 255u));
    if (FUSE && tid == 0) {
        const unsigned gap = (unsigned)real0 - *(volatile unsigned int *)cu_end;
        if (gap < 6000u) {      // (else: the launch's first workgroups)
            atomicAdd(fz.tick + 256 + 16 * (blockIdx.x & 63) + 11, gap);
            atomicAdd(fz.tick + 256 + 16 * (blockIdx.x & 63) + 12, 1u);
        }
    }
#endif
    const int teff_all = plan->teff;            // candidates of the round
    const int teff = teff_all - CB < 16 ? teff_all - CB : 16;   // ... of this launch
    const int label = plan->label;
    static_assert(EK_BLOCK == EK_TILE, "one workgroup per tile");
    const int64_t f0 = (int64_t)blockIdx.x * EK_BLOCK;
    const int64_t f = f0 + tid;
    // the arrival tickets of a fused single-shard round: see ek_pass2_kernel
    // rows_wr: the candidate frames' rows are wanted (single shard); order: this
    // launch ends with the presumed order (the round's last pass)
    const bool rows_wr = FUSE && fz.ord != nullptr;
    const bool order = rows_wr && (MODE != 1 || teff_all <= 16);
    bool owner_blk = false;
    unsigned int ticket = 0;
    if (rows_wr) {
#pragma unroll
        for (int j = 1; j < NC; ++j) {
            const int64_t l = plan->gidx[j] - fz.goff - f0;
            owner_blk |= j < teff_all && l >= 0 && l < EK_BLOCK;
        }
        if (order && !owner_blk && tid == 0)
            ticket = __hip_atomic_fetch_add(fz.tick, 1u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the first requests of the contraction, before anything else waits --------
    // rows: this tile of the quad copy; a trip of 4 atoms = 3 loads (x, y, z),
    // 16 bytes per lane; candidates: [16 atoms][3 loads][lane group][candidate][4]
    const int NQ = (A + 3) / 4;
    const float *tb = qtiles + (size_t)(f0 / EK_TILE) * (size_t)NQ * (3 * EK_TILE * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)tb, 0, NQ * (3 * EK_TILE * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)ctile, 0, ek_ctile_atoms(A) * (3 * 16 * 4), 0x00020000);
    const int vo = tid * 16, co = lane * 16;
    // (aux 2: non-temporal, the frame stream is read once per pass; loads past
    // the end of either buffer return zeros and are never multiplied)
    // (-DEK_P16_ABLATE=bits, measurement builds only: 1 no quartic solves, 2 no
    // row loads, 4 no candidate loads; results are then meaningless)
#ifndef EK_P16_ABLATE
#define EK_P16_ABLATE 0
#endif
    const ek_v4f ablate_v = {1.0f + (float)tid, 0.5f, -0.25f, 2.0f};
#define EK_LDR(TR, K)                                                          \
    ((EK_P16_ABLATE & 2) ? ablate_v :                                          \
    __builtin_bit_cast(ek_v4f, __builtin_amdgcn_raw_buffer_load_b128(          \
                                   rs, vo, ((TR) * 3 + (K)) * (EK_TILE * 16), 2)))
#define EK_LDC(SS, I)                                                          \
    ((EK_P16_ABLATE & 4) ? ablate_v :                                          \
    __builtin_bit_cast(ek_v4f, __builtin_amdgcn_raw_buffer_load_b128(          \
                                   cs, co, ((SS) * 3 + (I)) * (EK_WAVE * 16), 0)))
    constexpr int DR = 3;               // trips the row loads run ahead; DR + 1 buffers
    ek_v4f R[DR + 1][3];                // [trip % 4][xyz] -> 4 atoms
    ek_v4f Cq[2][3];                    // [super-trip % 2][load] -> 4 atoms
    // The frame's state first (vector loads return in order: what is asked for
    // before it is waited for with it), then the first rows: everything that does
    // not depend on something loaded is on its way before the first wait.
    double Gf = 0.0;
    float cur0 = 0.f;
    uint32_t vm = 0u;           // FUSE: which distance vectors this wave stored
    // triangle inequality (ek_round_ti_tiles_kernel): the candidates that can still
    // change a frame of this tile; none: the tile is not read
    uint32_t tm_v = 0xffffffffu;
    if (FUSE && fz.tmask)
        tm_v = fz.tmask[blockIdx.x];
    if (f < n) {
        Gf = G[f];
        cur0 = dist[f];
        if (FUSE)
            vm = fz.vmask[f >> 6];
    }
    int pn_v = 0, label0_v = 0; // (the pending chain's length and first label: with them)
    if (FUSE && MODE != 2) {
        pn_v = pend->n;
        label0_v = pend->label0;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        Cq[0][i] = EK_LDC(0, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < DR; ++k) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
            R[k][x] = EK_LDR(k, x);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (the only point where the four waves meet, while all of them wait for memory
    // anyway: from here on each wave runs to its end on its own -- its tables in
    // LDS are its own, and the workgroup's business at the end is done by
    // whichever wave arrives last)
    if (tid == 0)
        s_arrive = 0;
    __syncthreads();
    // (this launch's sixteen candidates of the mask; a tile that holds a candidate
    // frame is never without: a frame is at distance 0 from itself)
    const uint32_t tm = (__builtin_amdgcn_readfirstlane(tm_v) >> CB) & 0xffffu;
    const bool tile_off = FUSE && tm == 0u;
    // ---- this frame's state on the way in ---------------------------------------
    int32_t lab = -1;           // >= 0: the frame's state changes in this pass
    int own = 0;                // order: this frame is candidate `own` (>= 1)
    {
        if (rows_wr && owner_blk && f < n) {    // (31 workgroups of a pass at most)
#pragma unroll
            for (int j = 1; j < NC; ++j)
                if (j < teff_all && plan->gidx[j] - fz.goff == f)
                    own = j;
        }
        // kcenters.py:304-306 for the pending chain, in order.  A vector is stored
        // only where a wave holds a finite value: a wave that stored none (lane
        // 0's word is the wave's) skips the walk over the chain -- fifteen
        // dependent trips to memory otherwise.
        const int pn = __builtin_amdgcn_readfirstlane(pn_v);
        if (FUSE && MODE != 2 && pn > 0 && __builtin_amdgcn_readfirstlane(vm) != 0u) {
            const int label0 = __builtin_amdgcn_readfirstlane(label0_v);
            for (int k = 0; k < pn; ++k) {
                const int slot = pend->slot[k];
                if (f >= n || !((vm >> (slot + 1)) & 1u))
                    continue;
                const float d = vecs[(size_t)slot * n_pad + f];
                if (d < cur0) {
                    cur0 = d;
                    lab = label0 + k;
                }
            }
        }
        s_G[tid] = Gf;
        s_cur[tid] = cur0;      // read back after the loop, by this wave only
#if EK_P16_QUEUE
        s_t[tid] = ek_far_t_frame((float)Gf, A, cur0);
#endif
    }

    // ---- the contraction ------------------------------------------------------------
    // 9 x 4 accumulators of 16 frames x 16 candidates: acc[3 i + j][g] is S_ij of the
    // frames 16 g .. 16 g + 15 of the wave (lane l, register r: frame 16 g + 4 (l / 16)
    // + r, candidate l % 16)
    ek_v4f acc[9][4];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            acc[q][g] = ek_v4f{0.f, 0.f, 0.f, 0.f};
    // One trip K (0 .. 7 of the unrolled pair of super-trips) at trip index TT:
    // the 36 matrix instructions of its 4 atoms -- frames' group g, coordinates i
    // (rows) and j (candidates) -- with the requests for the candidates of the
    // NEXT 16 atoms (first trip of a super-trip, asked for before the rows of the
    // same trip: vector loads are counted in order) and for the rows of trip
    // TT + DR spread among them.
#define EK_TRIP16(K, TT)                                                       \
    {                                                                          \
        constexpr int SB = ((K) / 4) % 2;                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                        \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) {                    \
                _Pragma("unroll") for (int i = 0; i < 3; ++i)                  \
                    acc[3 * i + j][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(  \
                        R[(K) % 4][i][g], Cq[SB][j][(K) % 4], acc[3 * i + j][g], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                             \
                if ((K) % 4 == 0 && g == 0) {                                  \
                    Cq[1 - SB][j] = EK_LDC((TT) / 4 + 1, j);                   \
                    __builtin_amdgcn_sched_barrier(0);                         \
                }                                                              \
                if (g == 1) {                                                  \
                    R[((K) + DR) % 4][j] = EK_LDR((TT) + DR, j);               \
                    __builtin_amdgcn_sched_barrier(0);                         \
                }                                                              \
            }                                                                  \
        }                                                                      \
    }
    __builtin_amdgcn_sched_barrier(0);
    EK_P16_STAMP(1);
    if (EK_P16_PRIO)
        __builtin_amdgcn_s_setprio(0);
    // (the last trip's atoms past A are zeros in rows and candidates alike: a
    // product +0 added to an accumulator leaves its bits -- an accumulator is never
    // -0, it starts at +0 and +0 + -0 = +0 -- so every chain is the A atoms' own)
    const int n_trip = tile_off ? 0 : NQ;   // (no candidate can change this tile: nothing
                                            // is multiplied, nothing more is asked for)
    int t0 = 0;
    for (; t0 + 8 <= n_trip; t0 += 8) {
        EK_TRIP16(0, t0 + 0)
        EK_TRIP16(1, t0 + 1)
        EK_TRIP16(2, t0 + 2)
        EK_TRIP16(3, t0 + 3)
        EK_TRIP16(4, t0 + 4)
        EK_TRIP16(5, t0 + 5)
        EK_TRIP16(6, t0 + 6)
        EK_TRIP16(7, t0 + 7)
    }
    // up to seven whole trips more (wave-uniform branches)
#define EK_REST16(K)                                                           \
    if (t0 + (K) < n_trip)                                                     \
        EK_TRIP16(K, t0 + (K))
    EK_REST16(0)
    EK_REST16(1)
    EK_REST16(2)
    EK_REST16(3)
    EK_REST16(4)
    EK_REST16(5)
    EK_REST16(6)
#undef EK_REST16
#undef EK_TRIP16
#undef EK_LDC
#undef EK_LDR

    // ---- the lane's 16 pairs: candidate lane % 16, frames 16 b + 4 (lane / 16) + r ----
    // Most of these distances are never used -- the pair is far, all a strict "<"
    // asks -- and the float64 solve costs a wave whatever its SLOWEST lane needs.
    // So (round 4, EK_P16_QUEUE): every pair first takes the float32 certificate
    // (ek_far_certified_f32: sound, ~99 % of the far pairs, no float64), straight
    // line, no lane waits for another; the pairs it does not certify go into a
    // queue in LDS and are solved afterwards DENSELY, one per lane: the float64
    // path then runs once or twice per wave instead of sixteen times.  A wave
    // whose queue overflows (the first passes of a fit, where little is far yet)
    // takes the old path for all its pairs.  Same results either way: a
    // certified pair is +inf for every consumer, the others are solved as before.
    EK_P16_STAMP(2);
    if (EK_P16_PRIO)
        __builtin_amdgcn_s_setprio(EK_P16_PRIO);
    {
        const int cand = lane & 15;
        const double Gc = ctrace[cand];
        float *Dw = s_D[wave];
#if EK_P16_QUEUE
        uint32_t *Q = s_Q[wave];
        int qn = 0;                     // wave-uniform
        // (two pairs at a time in one basic block; a frame outside the shard or
        // a candidate outside the plan asks for nothing)
        const float tc = ek_far_t_center((float)Gc);
        const int n_here = (int)std::min<int64_t>(n - f0 - wave * EK_WAVE, EK_WAVE);
        const int fr_lim = (cand < teff && ((tm >> cand) & 1u)) ? n_here : 0;
#ifndef EK_P16_W
#define EK_P16_W 2
#endif
        constexpr int W = EK_P16_W;     // pairs whose certificates share a basic block
        if (tile_off) {         // (triangle inequality: every pair of the tile is far)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Dw[cand * EK_P16_DSTRIDE + 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3)] =
                    __builtin_inff();
        } else {
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += W) {
            float S[W][9], t[W];
            int fr[W];
            bool far[W];
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const int r = r0 + u;
                fr[u] = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
#pragma unroll
                for (int q = 0; q < 9; ++q)
                    S[u][q] = acc[q][r >> 2][r & 3];
                t[u] = s_t[wave * EK_WAVE + fr[u]] + tc;
            }
            ek_far_certified_f32_w<W>(S, t, far);
            // (all W verdicts before the first of them is acted on: the compiler
            // otherwise sinks each pair's chain behind the branch of the pair
            // before, one dependent chain after the other)
            {
                int fi[W];
#pragma unroll
                for (int u = 0; u < W; ++u)
                    fi[u] = far[u];
                if constexpr (W == 4)
                    asm volatile("" : "+v"(fi[0]), "+v"(fi[1]), "+v"(fi[2]), "+v"(fi[3]));
                else if constexpr (W == 2)
                    asm volatile("" : "+v"(fi[0]), "+v"(fi[1]));
#pragma unroll
                for (int u = 0; u < W; ++u)
                    far[u] = fi[u] != 0;
            }
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const bool need = fr[u] < fr_lim && !far[u];
                Dw[cand * EK_P16_DSTRIDE + fr[u]] = __builtin_inff();
                const unsigned long long m = __ballot(need);
                if (m) {                // wave-uniform
                    const int pos = qn + __popcll(m & ((1ull << lane) - 1ull));
                    if (need && pos < EK_P16_QCAP) {
                        uint4 *e = (uint4 *)(Q + pos * EK_P16_QSTRIDE);
                        e[0] = make_uint4(__float_as_uint(S[u][0]), __float_as_uint(S[u][1]),
                                          __float_as_uint(S[u][2]), __float_as_uint(S[u][3]));
                        e[1] = make_uint4(__float_as_uint(S[u][4]), __float_as_uint(S[u][5]),
                                          __float_as_uint(S[u][6]), __float_as_uint(S[u][7]));
                        e[2] = make_uint4(__float_as_uint(S[u][8]),
                                          (uint32_t)(cand << 8 | fr[u]),
                                          __float_as_uint(t[u]), 0u);
                    }
                    qn += __popcll(m);
                }
            }
        }
        }
        EK_P16_STAMP(3);
#ifdef EK_P16_STATS     // (measurement build: waves, overflowing waves, queued pairs)
        if (FUSE && lane == 0) {
            unsigned int *t = fz.tick + 256 + 16 * (blockIdx.x & 63);
            atomicAdd(t + 0, 1u);
            atomicAdd(t + 1, qn > EK_P16_QCAP ? 1u : 0u);
            atomicAdd(t + 2, (unsigned)qn);
            atomicAdd(t + 3, (unsigned)((qn + 63) / 64));
        }
#endif
        if (qn <= EK_P16_QCAP) {
            // (one wave: its LDS accesses are in order, no barrier)
            for (int base = 0; base < qn; base += EK_WAVE) {
                const int e = base + lane;
                // (EK_P16_LEVEL2: a second float32 level first, and the float64 path
                // only if a lane of the wave is still undecided; a certified pair is
                // +inf for every consumer, and Dw holds +inf already)
                float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int c2 = 0, fr = 0;
                bool open = false;
                if (e < qn) {
                    const uint4 *ent = (const uint4 *)(Q + e * EK_P16_QSTRIDE);
                    const uint4 e0 = ent[0], e1 = ent[1], e2 = ent[2];
                    S[0] = __uint_as_float(e0.x); S[1] = __uint_as_float(e0.y);
                    S[2] = __uint_as_float(e0.z); S[3] = __uint_as_float(e0.w);
                    S[4] = __uint_as_float(e1.x); S[5] = __uint_as_float(e1.y);
                    S[6] = __uint_as_float(e1.z); S[7] = __uint_as_float(e1.w);
                    S[8] = __uint_as_float(e2.x);
                    c2 = (int)(e2.y >> 8);
                    fr = (int)(e2.y & 255u);
#if EK_P16_LEVEL2
                    open = !ek_far_certified2_f32(S, __uint_as_float(e2.z));
#else
                    open = true;
#endif
                }
                if (__ballot(open)) {           // wave-uniform
                    if (open)
                        Dw[c2 * EK_P16_DSTRIDE + fr] =
                            ek_rmsd_from_S_below(S, s_G[wave * EK_WAVE + fr], ctrace[c2], A,
                                                 s_cur[wave * EK_WAVE + fr]);
                }
            }
        } else
#endif
        {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            const int fr = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
            float S[9];
#pragma unroll
            for (int q = 0; q < 9; ++q)
                S[q] = acc[q][r >> 2][r & 3];
            // (the bound is the frame's distance before this pass: what candidate 0
            // makes of it is only known in the frame's own lane, below)
            float d = __builtin_inff();
            // (a frame of the last tile's padding has S = 0: a quadruple root,
            // fifty Newton steps for nothing)
            if (EK_P16_ABLATE & 1) {    // (keeps the accumulators alive, costs ~10 adds)
                float t = S[0];
#pragma unroll
                for (int q = 1; q < 9; ++q)
                    t += S[q];
                if (t > 3e38f)
                    d = 0.f;
            } else if (cand < teff && ((tm >> cand) & 1u) && f0 + wave * EK_WAVE + fr < n)
                d = ek_rmsd_from_S_below(S, s_G[wave * EK_WAVE + fr], Gc, A,
                                         s_cur[wave * EK_WAVE + fr]);
            Dw[cand * EK_P16_DSTRIDE + fr] = d;
        }
        }
    }
    EK_P16_STAMP(4);
    ek_wave_sync();             // (s_D[wave] is written and read by this wave alone)
    // ---- lane = frame again -------------------------------------------------------
    float bestv = -__builtin_inff();
    uint32_t besti = 0xffffffffu;
    // (the frame's sixteen new distances in one go: read one by one behind the
    // branches below, each is a trip to LDS of its own)
    float dcs[T];
#pragma unroll
    for (int c = 0; c < T; ++c)
        dcs[c] = s_D[wave][c * EK_P16_DSTRIDE + lane];
    float run0 = -__builtin_inff();     // (sweep) the frame's distance after candidate 0
    uint32_t vm_out = 0u;               // (sweep) the wave's mask of vectors with a finite value
    if (f < n) {
        // candidate 0: the new center of this iteration (kcenters.py:298-306)
        float cur = cur0;
        if (MODE != 2) {
            const float d0 = dcs[0];
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
            if (rows_wr && own) {
                ek_coh_store(&fz.rows[own].cur, cur);
                ek_coh_store(&fz.rows[own].valid, 1);
            }
        }
        // a kept distance that is not below the frame's own can never be used
        // (strict <, against a value that only shrinks): it is +inf, and a wave
        // stores a vector only if one of its frames has a finite value -- one
        // word per wave says which it stored, readers take the others as +inf
        uint32_t wmask = MODE == 2 ? vm : 0u;
#pragma unroll
        for (int c = (MODE == 2 ? 0 : 1); c < T; ++c) {
            if (c < teff) {
                float dc = dcs[c];
                if (!(dc < cur))
                    dc = __builtin_inff();
                if (FUSE) {
                    if (__ballot(dc != __builtin_inff())) {     // wave-uniform
                        vecs[(size_t)(CB + c - 1) * n_pad + f] = dc;
                        wmask |= 1u << (CB + c);
                    }
                } else {
                    vecs[(size_t)(c - 1) * n_pad + f] = dc;
                }
                if (rows_wr && own)
                    ek_coh_store(&fz.rows[own].d[CB + c], dc);
            }
        }
        if (FUSE && (f & (EK_WAVE - 1)) == 0)
            fz.vmask[f >> 6] = wmask;
        run0 = cur;
        vm_out = wmask;
    }
    ek_wave_argmax(bestv, besti);
    // ---- round 6: the states the prefixes of the round's chain would leave ------------
    // With the presumed order = the order of the candidates (what the greedy choice of
    // the plan took them in), state k = min(cur, d_1 .. d_k) is at hand here, in
    // registers: its first-index arg-max per wave costs one wave arg-max per candidate
    // for which SOME frame of the wave holds a finite kept distance (a handful) -- the
    // chain kernel's sweep over dist and fifteen vectors, and its ticket, go away.
    // Only the states that DIFFER from the one before are written (a wave's word of
    // finite kept vectors says which: bit c = some frame of the wave holds a kept
    // distance to candidate c below its own); the workgroup's last wave reads, for
    // state k, every wave's last entry at or below k.
    __shared__ float s_pv[EK_BLOCK / EK_WAVE][T];
    __shared__ uint32_t s_pi[EK_BLOCK / EK_WAVE][T];
    __shared__ uint32_t s_fin[EK_BLOCK / EK_WAVE];
    const bool sweep = FUSE && MODE == 0 && fz.sweep_pm != nullptr;
    if (sweep) {
        // (bit 0: state 0 is always there; bits at or above teff never are)
        const uint32_t fin = (__builtin_amdgcn_readfirstlane(vm_out) & ~1u) | 1u;
        if (lane == 0) {
            s_pv[wave][0] = bestv;
            s_pi[wave][0] = besti;
            s_fin[wave] = fin;
        }
        float run = run0;
        // (most waves hold no finite kept distance at all: one test instead of fifteen)
        if (fin != 1u)
#pragma unroll
        for (int c = 1; c < T; ++c) {
            if ((fin >> c) & 1u) {              // uniform (and c < teff: a kept vector)
                if (dcs[c] < run)               // kcenters.py:304
                    run = dcs[c];
                float sv = f < n ? run : -__builtin_inff();
                uint32_t si = f < n ? (uint32_t)f : 0xffffffffu;
                ek_wave_argmax(sv, si);
                if (lane == 0) {
                    s_pv[wave][c] = sv;
                    s_pi[wave][c] = si;
                }
            }
        }
    }
#ifdef EK_P16_STATS
    EK_P16_STAMP(5);
    if (FUSE && lane == 0)
        for (int k = 0; k < 5; ++k)     // units of 64 cycles
            atomicAdd(fz.tick + 256 + 16 * (blockIdx.x & 63) + 4 + k,
                      (unsigned)((stamp[k + 1] - stamp[k]) >> 6));
    if (FUSE && lane == 0)      // the wave's life in 10 ns ticks
        atomicAdd(fz.tick + 256 + 16 * (blockIdx.x & 63) + 9,
                  (unsigned)(__builtin_amdgcn_s_memrealtime() - real0));
#endif
    __shared__ EkChainRow rows[EK_MAX_CANDS];
    __shared__ int s_chain[EK_MAX_CANDS];
    __shared__ int s_cn;
    if (lane == 0) {
        red_v[wave] = bestv;
        red_i[wave] = besti;
    }
    if (order && owner_blk) {
        // The few workgroups that hold a candidate frame: their rows have to be in
        // place before they count as arrived, all four waves go on together.
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
            if (MODE != 2) {    // (the state is the first pass's)
                blockmax[blockIdx.x].val = v;
                blockmax[blockIdx.x].idx = i;
            }
        }
        if (ek_arrive_last(fz.tick))
            ek_p16_order_tail<EK_BLOCK, NC>(tid, plan, fz, rows, s_chain, &s_cn, teff_all,
                                            label);
        return;
    }
    // Every other workgroup: no wave waits for another (a barrier here held each
    // wave for ~20 us, the matrix pipe of its SIMD idle whenever the other
    // resident wave was not in its loop either: profiles/r04/README.md).  A wave
    // leaves its maximum in LDS and counts itself; the one that counts last
    // finishes the workgroup's business alone.
    if (tid == 0)
        s_ticket = ticket;      // (before wave 0 counts itself: LDS is in order)
#ifdef EK_P16_STATS
    if (lane == 0)
        s_endt[wave] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    unsigned int arrived = 0;
    if (lane == 0)
        arrived = __hip_atomic_fetch_add(&s_arrive, 1u, __ATOMIC_ACQ_REL,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(arrived) != EK_BLOCK / EK_WAVE - 1)
        return;
#ifdef EK_P16_STATS
    if (FUSE && lane == 0) {    // first to last wave of the workgroup, 10 ns ticks
        unsigned lo = s_endt[0], hi = s_endt[0];
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w) {
            lo = min(lo, s_endt[w]);
            hi = max(hi, s_endt[w]);
        }
        atomicAdd(fz.tick + 256 + 16 * (blockIdx.x & 63) + 10, hi - lo);
        *(volatile unsigned int *)cu_end = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
#endif
    // the tile's maximum after candidate 0 (every lane works it out: four LDS words)
    float v0 = red_v[0];
    uint32_t i0 = red_i[0];
#pragma unroll
    for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
        if (ek_better(red_v[w], red_i[w], v0, i0)) {
            v0 = red_v[w];
            i0 = red_i[w];
        }
    if (MODE != 2 && lane == 0) {
        blockmax[blockIdx.x].val = v0;
        blockmax[blockIdx.x].idx = i0;
    }
    if (sweep && lane < teff) {
        // state `lane` of this tile from the four waves' entries (read by the next
        // launch).  A tile none of whose waves holds a finite kept distance: every
        // state is state 0, no walk over the waves' tables.  (Measured at 10^6 x 300:
        // the pass is 14-17 us = 2 % slower with the sweep in it, with or without this
        // shortcut and the one above -- as much as the chain kernel's own sweep cost
        // there, hence EK_OPT_PASS_SWEEP's rule by the shard's size;
        // profiles/r06/sweep_ab_1m.log.)
        uint32_t fw[EK_BLOCK / EK_WAVE];
#pragma unroll
        for (int w = 0; w < EK_BLOCK / EK_WAVE; ++w)
            fw[w] = s_fin[w];
        float v = v0;
        uint32_t i = i0;
        if ((fw[0] | fw[1] | fw[2] | fw[3]) != 1u) {        // uniform
            v = -__builtin_inff();
            i = 0xffffffffu;
#pragma unroll
            for (int w = 0; w < EK_BLOCK / EK_WAVE; ++w) {
                const uint32_t m = fw[w] & ((2u << lane) - 1u);
                const int e = 31 - __builtin_clz(m);        // (bit 0 is always set)
                const float ev = s_pv[w][e];
                const uint32_t ei = s_pi[w][e];
                if (ek_better(ev, ei, v, i)) {
                    v = ev;
                    i = ei;
                }
            }
        }
        if (lane >= 1) {
            EkBlockMax *o = fz.sweep_pm + (size_t)(lane - 1) * fz.sweep_nb + blockIdx.x;
            o->val = v;
            o->idx = i;
        }
    }
    if (sweep && fz.sweep_fm && lane < EK_BLOCK / EK_WAVE) {
        // (whatever teff is: a round of one candidate has these four entries too)
        const int e = 31 - __builtin_clz(s_fin[lane]);
        EkBlockMax *o = fz.sweep_fm + 4 * (size_t)blockIdx.x + lane;
        o->val = s_pv[lane][e];
        o->idx = s_pi[lane][e];
    }
    // the workgroup that drew the last ticket (at its start: every owner had
    // finished by then) works out the presumed order
    if (order && s_ticket == gridDim.x - 1)
        ek_p16_order_tail<EK_WAVE, NC>(lane, plan, fz, rows, s_chain, &s_cn, teff_all, label);
}

// wide: a round of 32 candidates -- the same stream twice, candidates 0 .. 15 and
// 16 .. 31 (the second launch returns at once when the plan holds 16 or fewer)
void ek_launch_pass16(bool fuse, const float *qtiles, const double *G, float *dist,
                      int32_t *assign, float *vecs, int64_t n, int64_t n_pad, int A,
                      const float *ctile, const double *ctrace, const EkPlan *plan,
                      EkBlockMax *blockmax, const EkFuse &fz, hipStream_t s, bool wide)
{
    if (n <= 0)
        return;
    const unsigned blocks = (unsigned)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (wide) {
        hipLaunchKernelGGL((ek_pass16_kernel<true, 1>), dim3(blocks), dim3(EK_BLOCK), 0,
                           s, qtiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                           ctrace, plan, blockmax, fz);
        hipLaunchKernelGGL((ek_pass16_kernel<true, 2>), dim3(blocks), dim3(EK_BLOCK), 0,
                           s, qtiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                           ctrace, plan, blockmax, fz);
    } else if (fuse)
        hipLaunchKernelGGL((ek_pass16_kernel<true, 0>), dim3(blocks), dim3(EK_BLOCK), 0,
                           s, qtiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                           ctrace, plan, blockmax, fz);
    else
        hipLaunchKernelGGL((ek_pass16_kernel<false, 0>), dim3(blocks), dim3(EK_BLOCK), 0,
                           s, qtiles, G, dist, assign, vecs, n, n_pad, A, ctile,
                           ctrace, plan, blockmax, fz);
}


// ---- the quad copy of the frames ----------------------------------------------------
// For tile t, trip q (atoms 4 q .. 4 q + 3), coordinate k: 256 slots of 16 bytes,
// slot 64 w + 16 e + f16 = the A operands of wave w of the tile for atom 4 q + e:
// the four floats are frames 64 w + 16 g + f16, g = 0 .. 3 (one per group of 16
// frames) -- what lane 16 e + f16 of that wave hands v_mfma_f32_16x16x4_f32 for
// the groups' four matrix instructions.  Zeros for the atoms past the last.  Made
// once per loaded shard from the frame-minor tiles.
__global__ void __launch_bounds__(EK_BLOCK)
ek_quad_tiles_kernel(const float *__restrict__ tiles, int A, int NQ,
                     float *__restrict__ qtiles)
{
    const int l = threadIdx.x;
    const int w = l >> 6, e = (l >> 4) & 3, f16 = l & 15;
    const size_t tile = blockIdx.x;
    const float *src = tiles + tile * 3 * (size_t)A * EK_TILE + 64 * w + f16;
    ek_v4f *dst = (ek_v4f *)qtiles + tile * (size_t)NQ * 3 * EK_TILE + l;
    for (int q = blockIdx.y; q < NQ; q += gridDim.y) {
        const int a = 4 * q + e;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ek_v4f v;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                v[g] = a < A ? src[(size_t)(3 * a + k) * EK_TILE + 16 * g] : 0.f;
            dst[(size_t)(q * 3 + k) * EK_TILE] = v;
        }
    }
}

size_t ek_quad_tiles_bytes(int64_t n_tiles, int A)
{
    return (size_t)n_tiles * (size_t)((A + 3) / 4) * 3 * EK_TILE * 16;
}

void ek_launch_quad_tiles(const float *tiles, int64_t n_tiles, int A, float *qtiles,
                          hipStream_t s)
{
    if (n_tiles <= 0)
        return;
    const int NQ = (A + 3) / 4;
    // (grid.y: a few workgroups per tile while the shard is small)
    const unsigned gy = (unsigned)std::max<int64_t>(
        1, std::min<int64_t>(NQ, 2048 / std::max<int64_t>(n_tiles, 1)));
    int64_t done = 0;
    while (done < n_tiles) {            // grid.x is limited to 2^31 - 1
        const int64_t cnt = std::min<int64_t>(n_tiles - done, 1 << 30);
        hipLaunchKernelGGL(ek_quad_tiles_kernel, dim3((unsigned)cnt, gy),
                           dim3(EK_BLOCK), 0, s,
                           tiles + (size_t)done * 3 * (size_t)A * EK_TILE, A, NQ,
                           qtiles + (size_t)done * (size_t)NQ * 3 * EK_TILE * 4);
        done += cnt;
    }
}
