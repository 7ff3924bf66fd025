// ek_common.h -- shared declarations of the gfx950 k-centers / RMSD library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/enspara_hip.h"

#define EK_BLOCK 256          // threads per workgroup of the streaming kernels
#define EK_WAVE 64
#define EK_MAX_ATOMS 4096     // center staged in LDS: 12 B/atom (48 KiB)

// ---- candidate record (see enspara_hip.h) ---------------------------------
struct EkRecHdr {
    float maxdist;
    int32_t valid;
    int64_t gidx;
    double trace;
    int64_t reserved;
};
static_assert(sizeof(EkRecHdr) == 32, "record header is 32 bytes");

static inline __host__ __device__ size_t ek_rec_bytes(int A)
{
    return (sizeof(EkRecHdr) + (size_t)12 * (size_t)A + 15) & ~(size_t)15;
}

// per-workgroup partial of the arg-max reduction
struct EkBlockMax {
    float val;
    uint32_t idx;   // frame index local to the shard
};

// accepted centers, one slot per label
struct EkHist {
    int64_t gidx;
    float dist;
    int32_t set;
};

// device-side control words
struct EkCtl {
    int32_t n_done;     // 1 + highest accepted label
    int32_t stopped;    // a step saw maxdist <= cutoff
    float last_max;     // maxdist carried by the most recent own record
    int32_t limit;      // multi-candidate rounds: stop once n_done == limit
    int32_t n_rounds;   // rounds (passes over the frames) that really ran
    int32_t pad[3];
};

// what a shard contributes to the per-center exchange between passes
struct EkMaxHdr {
    float maxdist;
    int32_t valid;
    int64_t gidx;
};
static_assert(sizeof(EkMaxHdr) == 16, "max header is 16 bytes");

typedef float ek_v2f __attribute__((ext_vector_type(2)));
// candidates per round of k-centers: 4, 8, 16 or 32 (a run-time choice,
// ek_run_rounds / ek_ms_run); the structures below are sized for the most.  A
// round of 32 is TWO passes of 16 over the frames (ek_pass16.hip) behind one
// plan, one chain and -- across shards -- one exchange (round 5).
#define EK_MAX_CANDS 32
// the widest form of the one-launch-per-step kernels (ek_chain.hip, the ek_spec_*
// protocol, option key 10 = 0): their per-thread arrays are sized for this
#define EK_LEGACY_CANDS 16
// PAM: proposals per prefetch pass / columns per workgroup of the pairs kernel
#define EK_PAM_GROUP 8
// plan of one multi-candidate round (ek_spec.hip); written only by the
// single-workgroup plan/check kernels, read by the kernels that follow
struct EkPlan {
    int32_t go;            // the round runs
    int32_t teff;          // candidates in use
    int32_t label;         // label of candidate 0
    int32_t apply;         // candidate to apply next (>= 1) or -1
    int32_t apply_label;
    int32_t miss;          // the farthest point was not a stored candidate
    uint32_t used;         // bitmask of candidates already applied
    int32_t n_rec;         // records chosen for the round (ek_round_ctile16_kernel)
    int32_t src[EK_MAX_CANDS];      // record index of each candidate
    int64_t gidx[EK_MAX_CANDS];
    float maxdist[EK_MAX_CANDS];
    double trace[EK_MAX_CANDS];
    // chained rounds (ek_chain.hip): the order in which candidates 1.. would be
    // accepted if each next farthest point is a stored candidate, how many of
    // them really were, and the label of the first
    int32_t chain_n;
    int32_t napply;
    int32_t chain_label0;
    int32_t pad2;
    int32_t chain[EK_MAX_CANDS];
    // multi-shard rounds: which of the offered records each candidate is
    // (ek_ms_ctile16_kernel reads their coordinates out of the mailboxes)
    int32_t offer[EK_MAX_CANDS];
};

// one candidate frame as seen by the shard that owns it: its current distance
// and its distance to every candidate of the round (ek_chain.hip)
struct EkChainRow {
    float cur;
    int32_t valid;
    float d[EK_MAX_CANDS];
};
static_assert(sizeof(EkChainRow) == 8 + 4 * EK_MAX_CANDS, "EkChainRow layout");

// ---- fused single-shard rounds (ek_round.hip) ----------------------------------
// the accepted part of a round's chain, not yet applied to dist / assign: the
// next pass applies it on the fly (its labels are label0, label0 + 1, ..)
struct EkPend {
    int32_t n;
    int32_t label0;
    int32_t slot[EK_MAX_CANDS];     // stored distance vector of each
};
// the presumed acceptance order of a round's candidates 1.. (ek_chain.hip, step 1)
struct EkChainOrd {
    int32_t n;
    int32_t cand[EK_MAX_CANDS];
};
// what the pass kernel needs for its part of a fused round
struct EkFuse {
    const EkPend *pend = nullptr;
    EkChainOrd *ord = nullptr;
    unsigned int *tick = nullptr;
    int64_t goff = 0;
    EkHist *hist = nullptr;
    EkCtl *ctl = nullptr;
    EkChainRow *rows = nullptr;     // [EK_MAX_CANDS] the candidate frames' rows
    uint32_t *vmask = nullptr;      // per 64 frames: bit c = vector c - 1 stored
    const uint32_t *tmask = nullptr;    // triangle inequality (ek_round_ti_*): per tile,
                                    //   bit c = candidate c can still change a frame of it
    // round 6: the per-prefix maxima of the round's chain taken by the pass itself
    // (ek_pass16_kernel<true, 0>): the presumed order is then the order of the
    // candidates (the greedy choice's own), known before the pass; sweep_pm[(k - 1) * nb
    // + tile] = first-index arg-max of the tile in the state candidates 0 .. k leave,
    // sweep_fm (may be null): per 64 frames, the state all of them leave
    EkBlockMax *sweep_pm = nullptr;
    EkBlockMax *sweep_fm = nullptr;
    int sweep_nb = 0;
};
// everything a fused round works on besides the frames
struct EkRound {
    float *dist;
    int32_t *assign;
    float *vecs;
    int64_t n, n_pad, goff;
    int A, T;
    const float *tiles;
    const float *qtiles;        // quad copy (16-candidate pass; null until one runs)
    const float *aos;           // the centred frames, frame-major [n][3A]
    const double *G;
    unsigned char *recs;        // T records (kept for the other entry points)
    EkPlan *plan;
    EkPend *pend;
    EkChainOrd *ord;
    EkBlockMax *blockmax;       // per 256 frames, state after the pass
    EkBlockMax *pm;             // [EK_MAX_CANDS][nb] states after each chain prefix
    EkBlockMax *fm;             // [4 nb] per 64 frames: the state after the whole chain
    unsigned char *top;         // scratch of the candidate pick
    float *ctile;
    double *ctrace;
    EkHist *hist;
    EkCtl *ctl;
    EkChainRow *rows;           // [EK_MAX_CANDS], see EkFuse
    uint32_t *vmask;            // [n_pad / 64], see EkFuse
    unsigned int *tick;         // [3] arrival counters of the three kernels
    double cutoff;
    // triangle inequality for rounds (option key 11, ek_round.hip "ek_round_ti_*"):
    // null = off
    float *ti_tab = nullptr;    // [labels][EK_MAX_CANDS] center-to-candidate distances
    uint32_t *tmask = nullptr;  // [tiles] candidates that can still change the tile
    unsigned long long *ti_stats = nullptr; // [0] (tile, candidate) pairs looked at, [1] left out
    int pick_cap = 4;           // far frames kept per label by the candidate pick (ek_top_dev.h)
    int sweep = 0;              // the pass takes the per-prefix maxima itself (EkFuse::sweep_pm)
};
// the masks of the round the plan describes (after ek_launch_round_next)
void ek_launch_round_ti(const EkRound &r, int max_labels, hipStream_t s);
// with_order: the pass's last workgroup works out the presumed order (single shard)
void ek_launch_round_pass(const EkRound &r, hipStream_t s, bool with_order = true);
// bootstrap != 0: no chain to decide, candidates from blockmax as it is
void ek_launch_round_chain(const EkRound &r, int bootstrap, hipStream_t s);
void ek_launch_round_next(const EkRound &r, int bootstrap, hipStream_t s);
// apply what is pending (end of a run, or before leaving the fused form)
void ek_launch_round_flush(const EkRound &r, hipStream_t s);

// ---- rounds across shards (ek_mshard.hip) ------------------------------------------
// One exchange per round: every shard's MESSAGE = the (max distance, global
// index) of its frames in the state every prefix of the round's chain would
// leave + its `offer` farthest frames of the state the whole chain would leave
// (records).  EkMsMsg | EkMaxHdr[EK_MAX_CANDS] | offer records.
#define EK_MS_MAX_WORLD 64
// records on offer in an exchange, over all shards: 128 (128 / world per shard, at most the 64
// a shard's pick lists), of which the plan kernel lets the 64 farthest compete -- the far
// frames of a state are not spread evenly over the shards (64 of them over 8 shards: 8 +- 2.6
// per shard), and with 64 / world offers per shard the ones beyond a shard's quota were on no
// list: 420 -> 407 passes for the headline case split 8 ways.  (For half a day of round 6 this
// stood accused of breaking runs of three and more shards in long-lived processes; what broke
// them was hipFree of uncached memory -- ek_uncached_alloc in ek_api.hip --, which larger
// mailboxes provoked sooner.  -DEK_MS_SLOTS=64 builds round 5's quota.)
#ifndef EK_MS_SLOTS
#define EK_MS_SLOTS 128
#endif
struct EkMsMsg {
    int32_t n_recs;         // valid records offered
    int32_t cn;             // states with a header
    int32_t state;          // the state the offered records are the far frames of
                            // (0 .. cn; -1: no chain, as the shard's state stands)
    int32_t pad;
};
static inline __host__ __device__ size_t ek_ms_msg_bytes(int A, int offer)
{
    return sizeof(EkMsMsg) + EK_MAX_CANDS * sizeof(EkMaxHdr) +
           (size_t)offer * ek_rec_bytes(A);
}
// device-side state of the rounds (one per context)
struct EkMsState {
    int32_t mode;           // 0 the run is over, 1 the pass runs, 2 no pass: offer
                            // records of state `pick_state` (the chain broke there)
    int32_t pick_state;
    uint32_t seq;           // exchanges completed (mailbox sequence number)
    int32_t err;            // a peer's message did not arrive
    uint32_t err_seq;       // ... in this exchange
    // what a run across shards spent waiting (ek_ms_diag; bench.py reports them per
    // rank): 10 ns ticks the last helper workgroup of the chain kernel waited for
    // the peers' flags (summed over the peers' waits, which overlap: the longest
    // of an exchange is what counts -- kept apart), exchanges, exchanges without a
    // pass (a chain that broke is offered again)
    uint32_t n_reoffer;
    unsigned long long wait_ticks_max;      // sum over exchanges of the longest wait
    unsigned long long wait_ticks_own;      // ... of the wait for this shard's OWN flag
};
// where a round's messages go and come from
struct EkMsXchg {
    int32_t world = 1, rank = 0, offer = 0;
    int32_t sys = 0;        // 1: peer mailboxes (system-scope stores / loads, flags);
                            // 0: dst[0] is a local buffer, src the gathered messages
    size_t msg_bytes = 0;
    unsigned char *dst[EK_MS_MAX_WORLD] = {};   // peer p's mailbox area [2][world][msg]
    uint32_t *dflag[EK_MS_MAX_WORLD] = {};      // peer p's flags [2][world][16]
    const unsigned char *src = nullptr;         // own mailbox area / gathered messages
    const uint32_t *sflag = nullptr;            // own flags
    // round 6 (mailbox transport only): the per-prefix headers travel FIRST, every shard
    // decides the chain inside its chain kernel and offers the far frames of the state the
    // chain really left -- a broken chain costs no exchange of its own (ek_ms_chain_kernel)
    int32_t two_phase = 0;
};
// chain: per-prefix maxima (presumed order = pick order), this shard's headers,
// its farthest frames of the speculated state, the message out
void ek_launch_ms_chain(const EkRound &r, EkMsState *ms, const EkMsXchg &x,
                        hipStream_t s);
// plan: all shards' messages in -> decide the chain, choose the next round's
// candidates (or ask for the records of the state the chain broke at)
void ek_launch_ms_plan(const EkRound &r, EkMsState *ms, const EkMsXchg &x, float *D,
                       hipStream_t s);

// ---- kernel launchers (defined in the .hip files) ---------------------------
// centring + trace + frame-minor transposition of `count` AoS frames
// aos_copy (optional): the centred coordinates once more, frame-major [n][3A]
void ek_launch_prepare_tiles(const float *src_aos, int64_t count, int A,
                             float *tiles, double *G, int64_t first_frame,
                             int64_t n_total, float *aos_copy, hipStream_t s);
// centring + trace of `count` AoS structures into center-major AoS
void ek_launch_prepare_centers(const float *src_aos, int32_t count, int A,
                               float *out_aos, double *Gc, hipStream_t s);

// one-center-vs-all distance pass.
//   mode 0: fused k-centers step (update dist/assign, per-block arg-max)
//   mode 1: distances only -> out_dist
void ek_launch_step(int fpl, int mode, int nt, const float *tiles,
                    const double *G,
                    float *dist, int32_t *assign, float *out_dist,
                    const unsigned char *recs, int n_recs, int64_t n, int A,
                    int label, double cutoff, EkBlockMax *blockmax,
                    EkHist *hist, EkCtl *ctl, hipStream_t s,
                    const uint8_t *tile_skip = nullptr);
// triangle inequality for the one-center step: distances of the k existing
// centers (hist[0..k)) to the new one (rec) -> Dnew[k]; tile_skip[t] = 1 for the
// tiles no frame of which can move; stats[0] += tiles looked at, [1] += skipped
void ek_launch_ti(const float *aos, const double *G, int A, const EkHist *hist,
                  int k, int64_t goff, const unsigned char *rec, float *Dnew,
                  const float *dist, const int32_t *assign, int64_t n,
                  const EkCtl *ctl, uint8_t *tile_skip, unsigned long long *stats,
                  hipStream_t s);
// the sharded form: centers from the table of accepted centers (row k is filled
// with the winner of `recs` here), tile marks as above
void ek_launch_ti_tab(float *tab, double *tabG, int A, int k,
                      const unsigned char *recs, int n_recs, float *Dnew,
                      const float *dist, const int32_t *assign, int64_t n,
                      const EkCtl *ctl, uint8_t *tile_skip, unsigned long long *stats,
                      hipStream_t s);
int ek_step_blocks(int fpl, int64_t n);

// reduce block partials (or, if blockmax == nullptr, the dist array itself)
// to the shard's (max, first index), gather that frame into `rec`
void ek_launch_pick(const EkBlockMax *blockmax, int n_blocks,
                    const float *dist, const float *tiles, const double *G,
                    int64_t n, int A, int64_t global_offset,
                    unsigned char *rec, EkCtl *ctl, hipStream_t s);

// record from a given local frame (center = frame index; idx_dev != nullptr:
// the index is read from device memory instead)
void ek_launch_record_from_frame(const float *tiles, const double *G, int A,
                                 int64_t local_idx, const int64_t *idx_dev,
                                 int64_t global_offset, unsigned char *rec,
                                 hipStream_t s);
// record from centred center-major coordinates (center = external structure)
void ek_launch_record_from_center(const float *center_aos, const double *Gc,
                                  int A, unsigned char *rec, hipStream_t s);

void ek_launch_fill_state(float *dist, int32_t *assign, int64_t n, float d,
                          int32_t a, hipStream_t s);

// all frames x K centers, strict-< in ascending center order
void ek_launch_assign(const float *tiles, const double *G, int64_t n, int A,
                      const float *centers_aos, const double *Gc, int32_t K,
                      float *dist, int32_t *assign, hipStream_t s);

// same result through the matrix cores; ctiles: centers in the frame-minor tile
// layout (as produced by ek_launch_prepare_tiles)
void ek_launch_assign_mfma(const float *tiles, const double *G, int64_t n, int A,
                           const float *ctiles, const double *Gc, int32_t K,
                           float *dist, int32_t *assign, hipStream_t s);

// ---- PAM (ek_pam.hip) ----------------------------------------------------------
void ek_launch_gather_frames(const float *tiles, const double *G, int A,
                             const int64_t *idx_dev, int count, int first_row,
                             float *out_aos, double *outG, hipStream_t s);
void ek_launch_count_members(const int32_t *assign, int64_t n, int32_t cid,
                             int32_t *blockcnt, int64_t *scan, int64_t *total,
                             hipStream_t s);
void ek_launch_select_member(const int32_t *assign, int64_t n, int32_t cid,
                             const int64_t *scan, int64_t j, int64_t *out,
                             hipStream_t s);
void ek_launch_count_members_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, int32_t *blockcnt, int64_t *scan,
                                   int64_t *total, hipStream_t s);
void ek_launch_select_member_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, const int64_t *scan,
                                   const int64_t *js_dev, int64_t *out,
                                   hipStream_t s);
void ek_launch_pam_classify(const float *dist, const int32_t *assign,
                            const float *newd, int64_t n, int32_t cid,
                            float *ndist, int32_t *nassign, uint32_t *amb,
                            unsigned long long *amb_best,
                            unsigned int *amb_count, unsigned int *reach,
                            hipStream_t s, int mark = 0);
// the classification of a window's slot (ek_pam.hip, ek_pam_classify_window_kernel)
struct EkPamClsWin {
    const int32_t *prev_accept;
    const float *frames_aos;    // frame-major copy of the shard
    const double *G;
    int32_t A;
    float *ambt;                // [3A][cap]
    double *ambG;
    int64_t cap;
    // pruning from the tables (O == nullptr: ek_pam_prune_kernel follows)
    const float *O, *T;         // O: this slot's row; T: the whole table [.][K]
    const int32_t *accepted;    // the window's verdicts so far, [slot]
    int32_t K, cid0, slot;
    int32_t *list;
    unsigned int *tick;         // [0] top, [1 ..] EK_ARRIVE_G leaves
};

void ek_launch_pam_classify_window(float *dist, int32_t *assign, const float *newd,
                                   int64_t n, int32_t cid, float *ndist,
                                   int32_t *nassign, uint32_t *amb,
                                   unsigned long long *amb_best,
                                   unsigned int *amb_count, const EkPamClsWin &w,
                                   hipStream_t s);
// medoids within reach of the ambiguous members -> list / n_list
void ek_launch_pam_prune(const float *aos, const double *Gm, int A, int K, int cid,
                         const unsigned int *reach, int32_t *list,
                         unsigned int *n_list, hipStream_t s);
// n_amb is read on the device; max_amb (a host-side upper bound) sizes the grid
// ambt [3A][cap] / ambG [cap]: scratch for the compacted frames, cap >= max_amb
void ek_launch_subset_assign(const float *tiles, const double *G, int A,
                             const uint32_t *amb, const unsigned int *n_amb,
                             int64_t max_amb, float *ambt, double *ambG,
                             int64_t cap, const float *centers,
                             const double *Gc, int K, const int32_t *list,
                             const unsigned int *n_list, const float *newd,
                             int cid, unsigned long long *amb_best,
                             hipStream_t s, bool gathered = false);
void ek_launch_pam_scatter(const uint32_t *amb,
                           const unsigned long long *amb_best,
                           const unsigned int *n_amb, int64_t max_amb,
                           float *ndist, int32_t *nassign, hipStream_t s);
// one shard's share of a proposal's outcome (ek_pam_propose_center)
struct EkPamOut {
    double sum_old;      // sum of squared distances, current state
    double sum_new;      // ... trial state
    int64_t n_frames;    // frames of this shard
    uint32_t n_amb;      // members of the cluster that stayed with it or left
    uint32_t moved;      // bit i: membership of cluster win_lo + i would change
};
static_assert(sizeof(EkPamOut) == 32, "EkPamOut layout");
// a window of proposals decided on the device (ek_pam_window_run): up to
// EK_PAM_WIN consecutive clusters, their proposals drawn, prefetched (in groups
// of EK_PAM_GROUP columns) and decided without a host round trip in between
#ifndef EK_PAM_WIN
#define EK_PAM_WIN 32     // (16 in round 2: the set-up of a window is ~240 us whatever its width)
#endif
static_assert(EK_PAM_WIN % EK_PAM_GROUP == 0 && EK_PAM_WIN <= 32,
              "whole column groups; the stale mask is 32 bits");
struct EkPamWin {
    int32_t stop;       // slots [0, stop) were decided
    uint32_t stale;     // clusters of the window whose membership changed
    int32_t err;        // 1 + slot of a proposal with more ambiguous members than declared
    int32_t pad;
    int32_t accept[EK_PAM_WIN];
    EkPamOut out[EK_PAM_WIN];
};
// what the last workgroup of a proposal's cost-sum launch needs to decide the
// proposal and to set up the next one (ek_pam.hip, "a window of proposals")
struct EkPamDecide {
    EkPamWin *win;
    int32_t slot;
    double n_total;             // frames of all shards: the means' divisor
    float *aos;                 // trial medoid table [K + 1][3A], row K = saved row
    double *Gm;
    int32_t A, K, cid;
    int64_t *med_idx;           // medoid frame indices (may be nullptr)
    int64_t frame;              // the frame proposed for cluster cid
    int64_t max_amb;
    int32_t next_cid;           // -1: last slot of the window
    int64_t next_frame;
    const float *frames_aos;    // frame-major copy of the shard
    const double *G;
    unsigned int *amb_count;
    unsigned int *moved;
};
// the state takes over the trial state if *flag != 0 (after a window's last slot)
void ek_launch_pam_apply(const int32_t *flag, float *dist, const float *ndist,
                         int32_t *assign, const int32_t *nassign, int64_t n,
                         hipStream_t s);
// active-set proposal prefetch (ek_pam.hip)
// T [n_prop][K]: medoid-to-proposal distances, dmin [groups of EK_PAM_GROUP
// proposals][K] their minimum per medoid and group, O [n_old][K]: distances to
// the medoids of clusters old_lo ..
void ek_launch_pam_tables(const float *aos, const double *Gm, int A, int K, int held,
                          const unsigned char *recs, int n_prop, int old_lo,
                          int n_old, float *T, float *O, float *dmin,
                          hipStream_t s, const float *dprop = nullptr);
// vecs[j * n_pad + list[i]] = rmsd(frame list[i], record j): the listed frames
// straight from the frame-major copy, results scattered into the full vectors
void ek_launch_pam_list_dist(const float *aos, const double *G, int A,
                             const uint32_t *list, int64_t n_rows,
                             const unsigned char *recs, int count, float *vecs,
                             int64_t n_pad, hipStream_t s);
void ek_launch_pam_active(const float *dist, const int32_t *assign, int64_t n,
                          const float *dmin, int n_groups, int K, int32_t win_lo,
                          int32_t win_count, uint32_t *list, unsigned int *n_list,
                          hipStream_t s, bool cleared = false);
// records, candidate tile + traces, fixed plan and a cleared counter for a
// window's proposals (local frames), one launch; the pass launcher is then
// called with prepared = true
#define EK_CTILE_PAD 8      // atoms of zeros after a candidate tile's last (read-ahead)
// Where (atom a, candidate c, coordinate k) of a round's T candidates sits in the
// candidate tile the pass kernel reads: T <= 8 [atom][pair][xyz][2] (pairs of
// candidates as scalar operands of v_pk_fma_f32, ek_spec.hip); T = 16
// [16 atoms][xyz][lane][trip]: the B operand of v_mfma_f32_16x16x4_f32 for one
// trip of 4 atoms and one coordinate has atom (lane / 16) of the trip and
// candidate (lane % 16) in each lane; one 16-byte load per lane fetches that
// operand for the four trips of 16 atoms (ek_pass16.hip).  The tile holds A +
// EK_CTILE_PAD atoms rounded up to whole groups of 16, zeros past the last.
// T = 32: two such tiles, candidate c in tile c / 16 (ek_ctile_half_floats).
static inline __host__ __device__ size_t ek_ctile_index(int T, int a, int c, int k)
{
    if (T >= 16)    // [16 atoms][xyz][lane = (a % 4) * 16 + c][trip of 4 atoms]
        return ((((size_t)(a >> 4) * 3 + k) * 4 + (a & 3)) * 16 + (c & 15)) * 4 + ((a >> 2) & 3);
    return (size_t)a * (3 * T) + (c / 2) * 6 + k * 2 + (c & 1);
}
static inline __host__ __device__ int ek_ctile_atoms(int A)  // incl. padding
{
    return (A + EK_CTILE_PAD + 15) / 16 * 16;
}
// a round of 32: candidates 16 .. 31 in a second tile of the 16-form, this many
// floats behind the first (what the round's second pass reads)
static inline __host__ __device__ size_t ek_ctile_half_floats(int A)
{
    return (size_t)ek_ctile_atoms(A) * 3 * 16;
}
void ek_launch_pam_setup(const float *aos, const double *G, int A,
                         const int64_t *frames, int count, int64_t global_offset,
                         unsigned char *recs, float *ctile, double *ctrace,
                         EkPlan *plan, unsigned int *counter, hipStream_t s,
                         const float *dist = nullptr, float *dprop = nullptr);
static inline int ek_pass_dist_T(int count)     // the pass width ek_launch_pass_dist picks
{
    return (count <= 4) ? 4 : 8;
}
// (aos: the frame-major copy of the shard)
void ek_launch_records_from_frames(const float *aos, const double *G, int A,
                                   const int64_t *frames, int count,
                                   int64_t global_offset, unsigned char *recs,
                                   hipStream_t s);
void ek_launch_pam_trial(const float *tiles, const double *G, int A, float *aos,
                         double *Gm, int K, int cid, int restore_cid,
                         int64_t frame_index, const int64_t *idx_dev,
                         const float *ext_aos, const double *ext_G,
                         unsigned int *amb_count, unsigned int *moved,
                         hipStream_t s, EkPamWin *win = nullptr, int win_slots = 0);
// numpy's summation order (ek_pam.hip, "cost sums in numpy's order")
#define EK_PW_CHUNK 8192            // numpy's reduction buffer, in elements
#define EK_PW_FULL_LEAVES 64        // leaves of a full chunk (128 elements each)
#define EK_PW_MAX_LEAVES 128        // leaves (and internal nodes) of any chunk
struct EkPwShape {
    int32_t n_leaves, n_nodes, n_levels, pad;
    int32_t leaf_off[EK_PW_MAX_LEAVES];   // relative to the chunk
    int32_t leaf_len[EK_PW_MAX_LEAVES];
    int32_t node_l[EK_PW_MAX_LEAVES];     // children: < n_leaves a leaf, else
    int32_t node_r[EK_PW_MAX_LEAVES];     //   n_leaves + node index
    int32_t level_start[16];              // nodes of level i: [start[i], start[i+1])
};
void ek_pw_build_shape(int len, EkPwShape *sh);      // host
void ek_launch_sumsq_pack(const float *a, float *b, const int32_t *assign,
                          int32_t *nassign, int64_t n, int32_t win_lo,
                          int32_t win_count, const EkPwShape *shapes, int n_full,
                          int n_leaves_total, int n_chunks, double *part,
                          const unsigned int *n_amb, unsigned int *moved,
                          EkPamOut *out, hipStream_t s,
                          const unsigned long long *amb_best = nullptr,
                          unsigned int *tick = nullptr,
                          const EkPamDecide *decide = nullptr);
void ek_launch_pam_vecs_reset(const uint32_t *list, int64_t n_rows, int cols,
                              int64_t n_pad, float *vecs, hipStream_t s);
void ek_launch_pw_tree(const float *dist, const int32_t *assign, int64_t n,
                       const EkPwShape *shapes, int n_full, int n_leaves_total,
                       int n_chunks, double *part, unsigned int *mask_scratch,
                       hipStream_t s);
void ek_launch_gather_rows(const float *tiles, const double *G, int A,
                           const int64_t *idx_dev, const int64_t *rows_dev,
                           int count, float *out_aos, double *outG, hipStream_t s);

// ---- multi-candidate rounds (ek_spec.hip) ---------------------------------------
void ek_launch_plan(const unsigned char *recs, int n_recs, int A, int T,
                    double cutoff, float *D, EkPlan *plan, EkHist *hist,
                    EkCtl *ctl, hipStream_t s);
// the candidates are laid out in `ctile` (ek_ctile_bytes) / `ctrace`
// ([EK_MAX_CANDS] f64) and read as scalar operands
size_t ek_ctile_bytes(int A);
// (qtiles: read instead of tiles when T == 16)
void ek_launch_pass(int T, const float *tiles, const float *qtiles, const double *G,
                    float *dist, int32_t *assign, float *vecs, int64_t n,
                    int64_t n_pad, int A, const unsigned char *recs,
                    const EkPlan *plan, EkBlockMax *blockmax, float *ctile,
                    double *ctrace, hipStream_t s);
// T = 16 through the matrix cores (ek_pass16.hip); fuse: the fused rounds' form
// (qtiles: the quad copy of the frames, ek_launch_quad_tiles)
size_t ek_quad_tiles_bytes(int64_t n_tiles, int A);
void ek_launch_quad_tiles(const float *tiles, int64_t n_tiles, int A, float *qtiles,
                          hipStream_t s);
void ek_launch_pass16(bool fuse, const float *qtiles, const double *G, float *dist,
                      int32_t *assign, float *vecs, int64_t n, int64_t n_pad, int A,
                      const float *ctile, const double *ctrace, const EkPlan *plan,
                      EkBlockMax *blockmax, const EkFuse &fz, hipStream_t s,
                      bool wide = false);
// distances only: vecs[j][f] = rmsd(frame f, record j), j < count <= 8
void ek_launch_pass_dist(int count, const float *tiles, const double *G,
                         float *vecs, int64_t n, int64_t n_pad, int A,
                         const unsigned char *recs, EkPlan *plan,
                         float *ctile, double *ctrace, hipStream_t s, bool prepared = false);
// ---- chained rounds (ek_chain.hip) --------------------------------------------------
// rows_out[EK_MAX_CANDS]: this shard's view of the candidate frames it owns
void ek_launch_chain_rows(const EkPlan *plan, const float *dist,
                          const float *vecs, int64_t n, int64_t n_pad,
                          int64_t global_offset, EkChainRow *rows_out,
                          hipStream_t s);
// presumed acceptance order from all shards' rows; rows_all == nullptr: single
// shard, the rows are computed in place
// per-workgroup maxima of the states after applying chain[0..k-1], k = 1..
void ek_launch_chain_max(const float *dist, const float *vecs, int64_t n,
                         int64_t n_pad, EkPlan *plan, EkBlockMax *pm,
                         int local_order, int64_t global_offset, hipStream_t s);
// this shard's (max, global index) for each of those states -> hdrs_out[8]
int ek_chain_max_blocks(int64_t n);     // entries per prefix in pm
// order + per-prefix maxima (one pm entry per 256 frames) + this shard's headers,
// one launch (the multi-shard round); tick: an arrival counter, left at 0
void ek_launch_chain_max2(const float *dist, const float *vecs, int64_t n,
                          int64_t n_pad, EkPlan *plan, const EkChainRow *rows_all,
                          int n_shards, const EkBlockMax *blockmax, EkBlockMax *pm,
                          int64_t global_offset, EkMaxHdr *hdrs_out,
                          unsigned int *tick, hipStream_t s);
// accept the longest verified prefix of the chain (hdrs_all[shard][8])
void ek_launch_chain_decide(const EkMaxHdr *hdrs_all, int n_shards, double cutoff,
                            EkPlan *plan, EkHist *hist, EkCtl *ctl, hipStream_t s);
// single shard: the two above in one launch
void ek_launch_chain_decide_local(const EkBlockMax *blockmax, const EkBlockMax *pm,
                                  int nb, int nbp, int64_t global_offset,
                                  double cutoff, EkPlan *plan, EkHist *hist,
                                  EkCtl *ctl, hipStream_t s);
void ek_launch_chain_apply(const float *vecs, int64_t n, int64_t n_pad, float *dist,
                           int32_t *assign, const EkPlan *plan,
                           EkBlockMax *blockmax, hipStream_t s);
void ek_launch_blockmax(const float *dist, int64_t n, EkBlockMax *blockmax,
                        hipStream_t s);
void ek_launch_localmax(const EkBlockMax *blockmax, int nb,
                        int64_t global_offset, EkMaxHdr *out, hipStream_t s);
void ek_launch_check(const EkMaxHdr *hdrs, int n_hdrs, double cutoff,
                     EkPlan *plan, EkHist *hist, EkCtl *ctl, hipStream_t s);
void ek_launch_apply(const float *vecs, const double *G, int64_t n,
                     int64_t n_pad, int A, float *dist, int32_t *assign,
                     const EkPlan *plan, EkBlockMax *blockmax, hipStream_t s);
size_t ek_top_scratch_bytes(int A);
void ek_launch_pickT(const EkBlockMax *blockmax, int nb, const float *tiles,
                     const double *G, const int32_t *assign, int A, int T,
                     int64_t global_offset, unsigned char *recs, EkCtl *ctl,
                     unsigned char *scratch, hipStream_t s);
void ek_launch_localmax_check(const EkBlockMax *blockmax, int nb,
                              int64_t global_offset, double cutoff,
                              EkPlan *plan, EkHist *hist, EkCtl *ctl,
                              hipStream_t s);
