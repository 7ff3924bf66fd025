// ek_pam_sparse.h -- a window of PAM proposals in one workgroup (ek_pam_sparse.hip)
#pragma once
#include "ek_common.h"

#define EK_SP_THREADS 512
// frames on a window's list (ek_pam_active_kernel) up to which the one-workgroup
// form is used; above it the three launches per proposal of ek_pam.hip are
#define EK_SP_CAP 65536
// (ambiguous member, medoid within reach) pairs one workgroup searches itself;
// a proposal with more ends the window and goes through the launches
#define EK_SP_MAX_PAIRS 16384
// chunks of 8192 frames whose sums the workgroup keeps in LDS (shards of up to
// 8.4 million frames)
#define EK_SP_MAX_CHUNKS 1024
// frames a proposal may change (a proposal with more ends the window)
#define EK_SP_CAP_CHG 2048

// What the speculative evaluation of a slot leaves (ek_sp_spec_kernel: every slot
// of the window at once, each on the state the window opens with); the window's
// workgroup takes it over where no accepted earlier slot touched what it read.
struct EkSpSpecRec {
    uint32_t n_chg, n_amb;
    uint32_t status;            // 0: lists below are complete; else the slot is evaluated in turn
    uint32_t tabconf;           // earlier slots whose old or new medoid is within the members' reach
    uint32_t conf_in;           // earlier slots that would change a frame of this slot's bucket,
    uint32_t moved;             // clusters of the window that would lose or gain a frame
    uint32_t pad;
    uint32_t conf_out;          // later slots whose bucket holds a frame this one would change
                                // (of every such pair at least one of the two has the bit: the
                                // one that marked the frame second, see ek_sp_spec_kernel)
    double delta, dab;          // sum and sum of magnitudes of new^2 - old^2 over the changes
};

struct EkSpArgs {
    float *dist;                // the state; trial values are written into it and
    int32_t *assign;            //   taken back if the proposal is rejected
    int64_t n;
    double n_total;             // the means' divisor
    int32_t A, K, cid0, count, win_count;
    const uint2 *bucket;        // [slot][bcap] the frames a slot looks at, with their
                                //   distance to the slot's proposal
    const unsigned int *bcnt;
    int64_t bcap;
    int64_t frames[EK_PAM_WIN]; // the proposed frames
    int64_t max_amb[EK_PAM_WIN];// members declared for the slots' clusters
    const float *O, *T;         // the window's tables (ek_pam_pairs_kernel<0>)
    float *med_aos;             // medoid table [K + 1][3A]
    double *med_G;
    int64_t *med_idx;
    int32_t restore;            // row a rejected proposal still occupies, or -1
    const float *frames_aos;    // frame-major copy of the shard
    const double *G;
    double *leaf;               // [2 g] leaf sum of the state, [2 g + 1] the value before
    double *chunk;              // same per chunk
    const EkPwShape *shapes;
    int32_t n_full, n_leaves, n_chunks;
    int64_t max_pairs;
    int32_t exact_always;       // take both cost sums for every proposal (ek_set_option key 14)
    EkPamWin *win;
    EkPamWin *win_host;         // mapped host memory that receives the record too (or nullptr)
    // speculative evaluation (round 5; use_spec = 0: every slot in turn, as before)
    int32_t use_spec;
    EkSpSpecRec *spec;          // [EK_PAM_WIN]
    uint32_t *spec_lists;       // [slot][5][EK_SP_CAP_CHG]: frame, old d, new d, old label, new label
    unsigned long long *marks;  // [n] bit j: the frame is in slot j's bucket, bit 32 + j: slot j
                                //   would change it (zero between windows)
    int32_t finish_later;       // the accepted proposals' rows of the medoid table and the marks
                                //   are left to ek_launch_sp_finish
    // (for ek_sp_finish_kernel: the window's distance vectors go back to +inf at the
    // listed frames once the window has run to its end)
    const uint32_t *act_list;
    int64_t n_act, n_pad;
    float *vecs;
    unsigned long long *prof;   // measurement builds (EK_SP_PROF): 10 ns ticks per step
};

void ek_launch_sp_bucket(const uint32_t *list, int64_t n_act, const float *dist,
                         const int32_t *assign, const float *vecs, int64_t n_pad,
                         int32_t cid0, int count, uint2 *bucket,
                         unsigned int *bcnt, int64_t bcap, hipStream_t s,
                         bool cleared = false);
void ek_launch_sp_spec(const EkSpArgs &p, hipStream_t s);
void ek_launch_sp_window(const EkSpArgs &p, hipStream_t s);
void ek_launch_sp_finish(const EkSpArgs &p, hipStream_t s);
size_t ek_sp_spec_bytes();      // of EkSpArgs::spec + ::spec_lists
