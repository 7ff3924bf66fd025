/*
 * enspara_hip.h -- C ABI of the MI355X (gfx950) k-centers / RMSD hot path.
 *
 * Drop-in boundary.  The reference (bowman-lab/enspara) has no FFI for this
 * path: its plug-in point is a Python callable `metric(X, y) -> distances`
 * (enspara/cluster/kcenters.py:132-137, resolved from the string 'rmsd' to
 * mdtraj.rmsd at enspara/cluster/util.py:289-291), driven by pure-Python /
 * numpy loops.  This header is the binding a maintainer would add underneath
 * those loops (INTEGRATION.md shows the ctypes stub).  Each entry point cites
 * the reference code it replaces.  Paths are relative to /root/reference.
 *
 * Conventions
 *   - every function returns 0 on success, a negative EK_E* code otherwise;
 *     ek_last_error() returns a thread-local human readable message;
 *   - plain pointers and sizes only; "host" pointers are ordinary memory,
 *     "dev" pointers are HIP device memory of the context's device;
 *   - a context owns one shard of frames resident in HBM, its k-centers state
 *     (float32 distance and int32 assignment per frame) and one HIP stream;
 *     all work is enqueued on that stream; functions that return data to host
 *     memory synchronise the stream, the others do not;
 *   - a context is not thread-safe; different contexts are independent.
 *
 * Data layout in HBM: centred coordinates in tiles of
 * EK_TILE frames, frame-minor: element (frame f, atom a, axis k) lives at
 * float index ((f / EK_TILE) * 3A + 3a + k) * EK_TILE + f % EK_TILE.
 */
#ifndef ENSPARA_HIP_H
#define ENSPARA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EK_ABI_VERSION 1
#define EK_TILE 256

#define EK_OK 0
#define EK_EARG (-1)     /* bad argument */
#define EK_EHIP (-2)     /* HIP runtime error (message has the hipError_t) */
#define EK_ESTATE (-3)   /* call sequence error (e.g. frames not loaded) */
#define EK_ENOMEM (-4)

typedef struct ek_ctx ek_ctx;

int ek_abi_version(void);
const char *ek_last_error(void);
/* number of visible HIP devices, or a negative error */
int ek_device_count(void);

/* ---- context ----------------------------------------------------------- */
/* n_frames frames of n_atoms atoms will live on `device`.  `global_offset`
 * is the index of this shard's first frame in the whole data set (0 on one
 * GPU); candidate records carry global indices.  `stream` is a hipStream_t to
 * enqueue on (e.g. torch.cuda.current_stream().cuda_stream) or NULL to let
 * the context create its own. */
int ek_ctx_create(int device, int64_t n_frames, int32_t n_atoms,
                  int64_t global_offset, void *stream, ek_ctx **out);
int ek_ctx_destroy(ek_ctx *ctx);
int ek_ctx_sync(ek_ctx *ctx);
/* the hipStream_t the context enqueues on */
void *ek_ctx_stream(ek_ctx *ctx);

/* ---- loading frames ---------------------------------------------------- */
/* Replaces the per-call centring mdtraj.rmsd does (precentered=False) and
 * the md.Trajectory.center_coordinates() enspara does once before
 * reassignment (enspara/cluster/util.py:624-629).
 * xyz: float32 [count][n_atoms][3] (the md.Trajectory.xyz layout,
 * enspara/util/load.py:211-216), host or device memory.  Frames
 * [first, first+count) of the shard are centred (float64 mean), their traces
 * computed, and stored frame-minor.  `first` must be a multiple of EK_TILE
 * unless it is 0.
 * Host memory (any: pageable numpy arrays) goes through two pinned buffers of
 * <= 256 MiB filled by a few host threads and a DMA behind each (16 -> 45 GB/s
 * measured; EK_UPLOAD_THREADS, EK_UPLOAD_CHUNK_MB).  On return the caller's
 * array has been read completely; the centring / layout kernels may still be
 * running on the context's stream, where everything that follows is ordered
 * behind them (ek_ctx_sync to wait). */
int ek_load_frames(ek_ctx *ctx, const float *xyz, int64_t first,
                   int64_t count, int src_is_device);

/* ---- metric parity: one center vs all frames --------------------------- */
/* Replaces `distance_method(traj, new_center)` (kcenters.py:298,
 * kmedoids.py:637) for metric 'rmsd'.  The center is either frame
 * `frame_index` of this shard (>= 0) or, if frame_index < 0, the float32
 * [n_atoms][3] host coordinates `center_xyz` (centred here).  Writes float32
 * [n_frames] RMSDs to out_host.  Does not touch the k-centers state. */
int ek_rmsd_to_center(ek_ctx *ctx, int64_t frame_index,
                      const float *center_xyz, float *out_host);

/* ---- k-centers state ---------------------------------------------------- */
/* kcenters.py:198-199: assignments = -1, distances = +inf */
int ek_state_reset(ek_ctx *ctx);
/* float32 distances / int32 assignments, [n_frames] each, host memory */
int ek_state_download(ek_ctx *ctx, float *dist_host, int32_t *assign_host);
int ek_state_upload(ek_ctx *ctx, const float *dist_host,
                    const int32_t *assign_host);

/* ---- candidate records --------------------------------------------------
 * A record describes a shard's farthest frame: what one rank contributes to
 * the exchange of kcenters.py:332-348 (two allgathers + Bcast of the frame)
 * folded into one message.  Layout (little endian):
 *   float  maxdist;  int32 valid;  int64 global_index;  double trace;
 *   int64  reserved;  float centred_xyz[3 * n_atoms];   padded to 16 bytes.
 */
size_t ek_record_bytes(int32_t n_atoms);
/* Reduce the shard's distances to (max, first index) -- np.argmax semantics,
 * kcenters.py:282 -- gather that frame, write the record to rec_dev (device
 * memory, ek_record_bytes long; NULL = the context's own slot). */
int ek_local_candidate(ek_ctx *ctx, void *rec_dev);
/* device address of the context's own record slot */
void *ek_own_record(ek_ctx *ctx);

/* One k-centers iteration on this shard (kcenters.py:243-311; the MPI
 * variant :314-378): pick the winning record among n_recs contiguous records
 * at recs_dev (largest maxdist, lowest position on ties = lowest rank,
 * kcenters.py:337); if winner.maxdist <= dist_cutoff (stop rule,
 * kcenters.py:217) do nothing; otherwise distance pass against the winner's
 * frame, strict-< update of distances/assignments with `label`
 * (kcenters.py:304-306), history[label] = winner, then refresh the context's
 * own record (as ek_local_candidate(ctx, own_rec_out)).  recs_dev == NULL
 * means "the context's own record" (single shard).  own_rec_out == NULL means
 * the context's own slot.  Entirely asynchronous. */
int ek_kcenters_step(ek_ctx *ctx, const void *recs_dev, int32_t n_recs,
                     int32_t label, double dist_cutoff, void *own_rec_out);

/* Single-shard driver: the whole loop of kcenters.py:217-231 on the device.
 * Starts at label `first_label`, adds at most `max_new` centers, stops early
 * when the maximum distance is <= dist_cutoff.  Returns the number of centers
 * added in *n_added, their frame indices in center_index_out[0..n_added) and
 * the distance each had to its nearest earlier center in
 * center_dist_out[0..n_added) (either may be NULL).  *final_maxdist receives
 * distances.max() after the last update (kcenters.py:226). */
int ek_kcenters_run(ek_ctx *ctx, int32_t first_label, int32_t max_new,
                    double dist_cutoff, int32_t *n_added,
                    int64_t *center_index_out, float *center_dist_out,
                    float *final_maxdist);

/* ---- several candidate centers per pass ("speculative" rounds) ------------------
 * The same sequential algorithm (identical centers, labels, distances) with
 * the frames streamed once per ROUND against T candidates (DESIGN.md
 * section 4a).  ek_kcenters_run uses it on one shard automatically.  Across
 * shards the caller moves two kinds of messages:
 *   per round:   T candidate records per rank (T * ek_record_bytes, T =
 *                ek_spec_candidates) -> all-gather -> ek_spec_round
 *   per center:  one 16-byte header {float maxdist; int32 valid; int64 global
 *                index} per rank (ek_spec_localmax) -> all-gather ->
 *                ek_spec_apply, T-1 times per round
 * ek_spec_begin(first_label, limit): accept centers first_label..limit-1, then
 * make every further call a no-op; writes this shard's first T records.
 * ek_spec_round_end writes the records for the next round.  ek_spec_progress
 * synchronises and reports how many centers exist and whether the stop rule
 * (maximum distance <= dist_cutoff, kcenters.py:217) fired. */
int ek_spec_candidates(ek_ctx *ctx);
/* the widest round ek_kcenters_run / ek_ms_run may use (option key 4; 16 by
 * default, 32 on request): ek_spec_candidates is the same for the ek_spec_*
 * protocol (<= 16) */
int ek_round_candidates(ek_ctx *ctx);
/* Rounds of 16 / 32 candidates stream a third copy of the frames (the quad copy,
 * 12 * n_atoms bytes per frame), made when first needed.  1: it exists or could
 * be made now; 0: no memory for it (a single shard then runs rounds of 8 on its
 * own).  Shards of a GROUP must all run the same form: the multi-shard entry
 * points (ek_ms_*, ek_spec_*) never narrow their rounds on their own -- they fail
 * with EK_ENOMEM -- and the caller agrees on the form first: every rank asks
 * this, and if any says 0 all set option key 4 to 8 (sharded.kcenters_sharded). */
int ek_quad_copy_ready(ek_ctx *ctx);
int ek_spec_begin(ek_ctx *ctx, int32_t first_label, int32_t limit,
                  void *recs_out);
int ek_spec_round(ek_ctx *ctx, const void *recs_all, int32_t n_recs,
                  double dist_cutoff);
int ek_spec_localmax(ek_ctx *ctx, void *hdr_out);
int ek_spec_apply(ek_ctx *ctx, const void *hdrs_all, int32_t n_hdrs,
                  double dist_cutoff);
int ek_spec_round_end(ek_ctx *ctx, void *recs_out);
int ek_spec_progress(ek_ctx *ctx, int32_t *n_done, int32_t *stopped);
/*
 * The cheap steps of a round can also be taken all at once ("chained",
 * csrc/ek_chain.hip): between ek_spec_round and ek_spec_round_end, instead of
 * up to T-1 times (ek_spec_localmax, all-gather, ek_spec_apply),
 *   ek_spec_chain_rows(ctx, my_rows)      this shard's view of the candidate
 *                                         frames it owns: current distance and
 *                                         distance to every candidate
 *   all-gather my_rows -> all_rows        (rows_bytes per shard)
 *   ek_spec_chain_max(ctx, all_rows, n_shards, my_hdrs)
 *                                         the order in which the candidates
 *                                         would be accepted, and this shard's
 *                                         (max distance, global index) of the
 *                                         state each prefix of it would leave
 *   all-gather my_hdrs -> all_hdrs        (hdrs_bytes per shard)
 *   ek_spec_chain_apply(ctx, all_hdrs, n_shards, dist_cutoff)
 *                                         accept the longest prefix whose every
 *                                         member is the farthest point when its
 *                                         turn comes, and apply it
 * -- two exchanges per round instead of one per accepted center, same centers,
 * labels and distances.  Buffers are device memory; ek_spec_chain_bytes gives
 * their per-shard sizes (320 and 128 bytes). */
int ek_spec_chain_bytes(int32_t *rows_bytes, int32_t *hdrs_bytes);
int ek_spec_chain_rows(ek_ctx *ctx, void *rows_out);
int ek_spec_chain_max(ek_ctx *ctx, const void *rows_all, int32_t n_shards,
                      void *hdrs_out);
int ek_spec_chain_apply(ek_ctx *ctx, const void *hdrs_all, int32_t n_shards,
                        double dist_cutoff);
/* ---- rounds across shards with ONE exchange per round (csrc/ek_mshard.hip) -------
 * Replaces, per ROUND of up to 16 candidates (~15 accepted centers), what the
 * reference's MPI iteration moves per CENTER: two pickled allgathers
 * (kcenters.py:332-335), the owner's frame (mpi/ops.py:169-212) and the
 * allreduce of the stop test (mpi/ops.py:128-140).  A shard's message holds its
 * (max distance, global index) in the state every prefix of the round's chain
 * would leave and its farthest frames of the state the whole chain would leave
 * (16 + 32 * 16 + offer * ek_record_bytes bytes, offer = min(64, 128 / world)).  Results are
 * those of the single-shard run bit for bit: every accepted center is the
 * global first-index arg-max of the state before it (lowest rank on ties,
 * kcenters.py:337).
 *
 *   ek_ms_setup(ctx, world, rank, &message_bytes)      once per group
 * Exchange by the caller (any all-gather: RCCL, gloo):
 *   ek_ms_begin(ctx, first_label, limit)
 *   repeat:  ek_ms_local(ctx, cutoff, my_message)       pass + chain -> message
 *            all-gather my_message -> all_messages      (message_bytes per shard)
 *            ek_ms_global(ctx, cutoff, all_messages)    decision + next plan
 *            (ek_spec_progress now and then; calls past the end are no-ops)
 *   ek_ms_end(ctx)                                      pending chain applied
 * Exchange on the device (peer mailboxes, no host and no collective in a
 * round): every shard publishes its mailbox (ek_ms_mailbox: addresses for
 * contexts of one process, hipIpc handles of 64 bytes each across processes),
 * connects every shard's including its own (ek_ms_connect), then
 *   ek_ms_run(ctx, first_label, max_new, cutoff, ...)   as ek_kcenters_run
 * on every shard at the same time.  A message that does not arrive within 10 s
 * (of the device's constant 100 MHz clock) is reported as an error (a peer
 * died), not waited for.  With the default option key 4 = -1 the run moves
 * between rounds of 8 and 16 (with key 4 = 32: and 32) candidates by the centers the rounds of a batch
 * accepted -- numbers every shard sees alike, so every shard takes the same
 * decision at the same round; ek_run_stats reports the mix.  A round of 32 is
 * two passes of 16 over the frames behind ONE plan, chain and exchange (the
 * message carries EK_MAX_CANDS = 32 per-prefix headers).  ek_ms_begin's loop
 * (exchange by the caller) runs one form: 16 unless key 4 names another. */
int ek_ms_setup(ek_ctx *ctx, int32_t world, int32_t rank, size_t *message_bytes);
int ek_ms_mailbox(ek_ctx *ctx, void **mbox, void **flags, void *ipc_mbox,
                  void *ipc_flags);
int ek_ms_connect(ek_ctx *ctx, int32_t peer, void *mbox, void *flags,
                  const void *ipc_mbox, const void *ipc_flags);
/* Allocates, NOW, everything a run of the rounds to n_centers centers needs (the
 * accepted centers' history, the rounds' working set, the third copy of the
 * frames for rounds of 16): an allocation inside ek_ms_run waits for the whole
 * device, and with peers already polling for this shard's message on the same
 * GPU that is a deadlock until the mailbox time-out.  Call it on every shard
 * before the first of them enters ek_ms_run.  No reference counterpart. */
int ek_reserve_centers(ek_ctx *ctx, int32_t n_centers);
int ek_ms_begin(ek_ctx *ctx, int32_t first_label, int32_t limit);
int ek_ms_local(ek_ctx *ctx, double dist_cutoff, void *message_out);
int ek_ms_global(ek_ctx *ctx, double dist_cutoff, const void *messages_all);
int ek_ms_end(ek_ctx *ctx);
/* What the last ek_ms_run spent where -- counts[5]: exchanges, exchanges without a
 * pass (a broken chain offered again), 10 ns ticks waited for the peers' messages
 * (per exchange the longest wait), ... for the shard's own flag (the floor),
 * rounds sampled; ms[3]: mean milliseconds of a sampled round's pass, chain
 * kernel (the exchange's wait inside) and plan kernels.  No reference counterpart
 * (its MPI iteration, kcenters.py:314-378, is not instrumented). */
int ek_ms_diag(ek_ctx *ctx, int64_t *counts, double *ms);
/* mode (0: the run is over), exchanges completed since ek_ms_setup, error */
int ek_ms_state(ek_ctx *ctx, int32_t *mode, int32_t *exchanges, int32_t *err);
int ek_ms_run(ek_ctx *ctx, int32_t first_label, int32_t max_new,
              double dist_cutoff, int32_t *n_added, int64_t *center_index_out,
              float *center_dist_out, float *final_maxdist);
/* rounds (passes over the frames) that really ran since ek_spec_begin /
 * ek_kcenters_run started */
int ek_spec_rounds(ek_ctx *ctx, int32_t *rounds);
/* How the last ek_kcenters_run / ek_ms_run spent its rounds:
 * passes[i] / centers[i] = rounds run with 1, 4, 8, 16, 32 candidate centers
 * (i = 0 .. 4: both arrays hold FIVE entries since round 5) and the centers
 * those rounds accepted.  A round of 32 streams the frames twice (two passes of
 * 16 behind one plan, one chain and one exchange); every other form once.  The loop of
 * kcenters.py:217-231 has no such notion (one metric call per center); the
 * forms give identical centers, labels and distances, and the run moves
 * between them by the centers per millisecond each achieves. */
int ek_run_stats(ek_ctx *ctx, int64_t *passes, int64_t *centers);
/* Triangle inequality (ek_set_option key 11; the reference's
 * `use_triangle_inequality`, kcenters.py:287-296 / :351-364): how many
 * (center, tile of 256 frames) pairs the last ek_kcenters_run looked at and
 * how many of them it did not have to read because no frame of the tile could
 * come closer to the new center than it is to its own. */
int ek_ti_stats(ek_ctx *ctx, int64_t *tiles, int64_t *skipped);

/* history written by ek_kcenters_step: for labels [first, first+count) the
 * global frame index and pre-update distance of each accepted center;
 * *n_done = 1 + the highest label accepted so far (0 if none). */
int ek_history_download(ek_ctx *ctx, int32_t first, int32_t count,
                        int64_t *center_index_out, float *center_dist_out,
                        int32_t *n_done);
int ek_history_reset(ek_ctx *ctx);

/* ---- nearest-center assignment ----------------------------------------- */
/* Replaces assign_to_nearest_center (enspara/cluster/util.py:159-205,
 * center-major branch :199-203; strict <, lowest center index wins ties,
 * assignments start at 0) for metric 'rmsd': centers_xyz is float32
 * [n_centers][n_atoms][3] host memory.  Overwrites the context's k-centers
 * state with the result (so predict / warm start / reassign read it back
 * with ek_state_download). */
int ek_assign_nearest(ek_ctx *ctx, const float *centers_xyz,
                      int32_t n_centers);

/* ---- PAM (k-medoids) proposals ------------------------------------------------
 * Replace the O(n) passes of one proposal of _kmedoids_pam_update
 * (enspara/cluster/kmedoids.py:610-690) for metric 'rmsd'.  The host keeps the
 * random stream (RandomState.choice, :514) and the accept test (:683).
 *
 * ek_pam_begin: medoid_frames[K] are the current medoids (frame indices of
 * this shard, :607); builds the device-side medoid table.  The k-centers
 * state (distances / assignments) must already be consistent with them.
 * ek_pam_count_members: len(where(assignments == cid)) (:611).
 * ek_pam_select_member: the j-th such frame in ascending order, so that
 * `state_inds[random_state.choice(len(state_inds))]` == the reference's
 * `random_state.choice(state_inds)`; call right after count_members(cid).
 * ek_pam_propose: distances to frame `frame_index` (:637), the three masks
 * (:644-658), nearest-medoid search for the ambiguous frames against the
 * trial medoid set (:666), and both costs mean(d^2) in float64 (:478,
 * :680-681).  The new state is held aside until
 * ek_pam_commit(accept != 0): adopt it (:687-689); accept == 0: drop it. */
int ek_pam_begin(ek_ctx *ctx, const int64_t *medoid_frames, int32_t n_medoids);
int ek_pam_count_members(ek_ctx *ctx, int32_t cid, int64_t *count);
int ek_pam_select_member(ek_ctx *ctx, int32_t cid, int64_t j,
                         int64_t *frame_index);
int ek_pam_propose(ek_ctx *ctx, int32_t cid, int64_t frame_index,
                   double *old_cost, double *new_cost, int64_t *n_ambiguous);
/* ek_pam_select_member + ek_pam_propose in one call (the selected frame index
 * never leaves the device before the final read-back): propose the j-th member
 * of cluster cid; needs ek_pam_count_members(cid) immediately before. */
int ek_pam_propose_member(ek_ctx *ctx, int32_t cid, int64_t j,
                          int64_t *frame_index, double *old_cost,
                          double *new_cost, int64_t *n_ambiguous);
int ek_pam_commit(ek_ctx *ctx, int accept);

/* Proposal prefetch.  A sweep (kmedoids.py:575-699) visits clusters 0..K-1 in
 * order, and an accepted proposal rarely changes the clusters visited next, so
 * the host may draw the next `count` (<= ek_pam_window_max(), 16) proposals
 * ahead of time and have their distance vectors computed together (one pass
 * over the frames per 8 of them, or -- ek_pam_prefetch_window -- only the
 * frames they can touch) instead of by one pass each; every guess is verified when its turn comes, so the sweep's
 * results do not change.
 *  ek_pam_count_members_batch: member counts of clusters cid0..cid0+count-1 in
 *    the current state (one read-back);
 *  ek_pam_select_members_batch: frames[i] = the js[i]-th member of cluster
 *    cid0+i, from the scans the batch count left (call it right after);
 *    js[i] < 0 skips cluster cid0+i (frames[i] = -1);
 *  ek_pam_prefetch: distances of every frame to frames[0..count) are computed
 *    and kept until the next ek_pam_prefetch / ek_pam_begin (count = 0 drops
 *    them); no read-back;
 *  ek_pam_propose_ex: ek_pam_propose with the member count of cluster cid
 *    supplied by the caller (n_members; it bounds the ambiguous set, the call
 *    fails if it was too small) and, when frame_index was prefetched, without
 *    the distance pass.  With win_count > 0 (<= 32), bit i of *moved_mask tells
 *    whether accepting the proposal changes the membership of cluster
 *    win_lo + i, i.e. whether member lists obtained earlier for that cluster
 *    are stale after ek_pam_commit(ctx, 1).
 *  ek_pam_prefetch_stats: proposals served from / not from a prefetched vector
 *    since the context was created. */
int ek_pam_count_members_batch(ek_ctx *ctx, int32_t cid0, int32_t count,
                               int64_t *counts);
int ek_pam_select_members_batch(ek_ctx *ctx, int32_t cid0, int32_t count,
                                const int64_t *js, int64_t *frames);
int ek_pam_prefetch(ek_ctx *ctx, const int64_t *frames, int32_t count);
/* The same when the caller works through clusters win_lo..win_lo+win_count-1
 * before the next prefetch (the sweep's window; the prefetched frames are
 * their proposals).  While the state is exact (ek_set_option key 6/7) exact
 * distances are then computed only for the frames a proposal can touch -- the
 * members of the window's clusters and every frame f with
 * D(its medoid, proposal) < 2 d(f) for some proposal; for the rest "not closer
 * than now" follows from the triangle inequality and the vector holds +inf --
 * instead of for all of them.  Same sweep results.  ek_pam_prefetch_passes
 * counts the prefetches of either kind. */
int ek_pam_prefetch_window(ek_ctx *ctx, const int64_t *frames, int32_t count,
                           int32_t win_lo, int32_t win_count);
int ek_pam_prefetch_passes(ek_ctx *ctx, int64_t *restricted, int64_t *full);
int ek_pam_propose_ex(ek_ctx *ctx, int32_t cid, int64_t frame_index,
                      int64_t n_members, int32_t win_lo, int32_t win_count,
                      double *old_cost, double *new_cost, int64_t *n_ambiguous,
                      uint32_t *moved_mask);
int ek_pam_prefetch_stats(ek_ctx *ctx, int64_t *hits, int64_t *misses);
/* A window of proposals without a host round trip each: the loop body of
 * kmedoids.py:597-699 for clusters cid0 .. cid0 + count - 1 (count <=
 * ek_pam_window_max()), in order.  frames[i] is the frame proposed for cluster cid0 + i, drawn by the
 * caller from that cluster's member list as it stood when the window was
 * opened, n_members[i] that list's length; every frames[i] must have been
 * prefetched with ek_pam_prefetch_window(ctx, frames, count, cid0, win_count).
 * All kernels are enqueued at once.  The device takes each decision --
 * mean(new^2) < mean(old^2) in float64, kmedoids.py:478-479, :683 --, commits
 * (:684-690) or undoes the proposal, and STOPS the window at the first cluster
 * whose member list an accepted proposal changed (its proposal was drawn from a
 * list that no longer holds, :611-614).  *n_done slots were decided; the caller
 * handles cluster cid0 + *n_done one proposal at a time and opens a new window
 * after it.  accept[i] (i < *n_done): 1 if proposal i was accepted; the costs
 * and ambiguous-member counts are returned for logging. */
/* windows ek_pam_window_run worked through in one workgroup (ek_set_option key
 * 12), and how many of them ended early because a proposal's ambiguous members
 * had more medoids within reach than one workgroup should search */
int ek_pam_sparse_stats(ek_ctx *ctx, int64_t *windows, int64_t *ended_early);
/* slots of those windows whose evaluation AHEAD of their turn was taken over
 * (ek_set_option key 19): every slot of a window is evaluated at once, one
 * workgroup each, on the state the window opens with; the window's workgroup
 * then goes through the slots in order (kmedoids.py:575-699 is sequential: a
 * proposal sees what the accepted ones before it changed) and takes a slot's
 * evaluation as it is unless an earlier slot it accepted changed a frame that
 * slot read or a medoid within its members' reach -- then the slot is
 * evaluated in its turn, as all were before round 5.  Same results either way. */
int ek_pam_ahead_stats(ek_ctx *ctx, int64_t *slots_taken_over);
/* the most proposals a window / a local-frame prefetch may hold */
int32_t ek_pam_window_max(void);
int ek_pam_window_run(ek_ctx *ctx, int32_t cid0, int32_t count,
                      const int64_t *frames, const int64_t *n_members,
                      int32_t win_lo, int32_t win_count, int32_t *n_done,
                      int32_t *accept, double *old_cost, double *new_cost,
                      int64_t *n_ambiguous);

/* The window loop of a whole sweep (kmedoids.py:575-699 for clusters *cid ..
 * n_medoids - 1, after ek_pam_begin): windows of up to `width` (2 ..
 * ek_pam_window_max()) proposals -- counted, drawn, selected, prefetched, decided
 * by ek_pam_window_run --, the cluster a window stops at one proposal at a time.
 * What the Python driver does call by call, without the interpreter between the
 * calls.  proposals == NULL: the draws are numpy's RandomState.choice(m) made on
 * `raw`, the next 32-bit outputs of the caller's RandomState
 * (randint(0, 2**32, dtype=uint32)); *pos counts the outputs consumed.
 * medoids[i] is replaced where proposal i was accepted; accept / old_cost /
 * new_cost / n_ambiguous [n_medoids] report every proposal.
 * *status: 0 done (*cid == n_medoids); 1 `raw` ran out -- call again with more
 * (everything returned so far stands); 2 cluster *cid is empty (choice raises). */
/* ek_pam_sweep's draws alone (host only, for tests): out[i] =
 * RandomState.choice(m[i]) -- 32-bit outputs masked to the bits of m - 1, values
 * above it rejected, m == 1 consumes nothing -- taken from `raw` from *pos on.
 * Returns how many draws were made: fewer than `count` if the outputs ran out or
 * an m[i] is < 1; -1 on a NULL argument. */
int64_t ek_np_choice_draws(const uint32_t *raw, int64_t n_raw, int64_t *pos,
                           const int64_t *m, int64_t count, int64_t *out);
int ek_pam_sweep(ek_ctx *ctx, int32_t n_medoids, int32_t width, const uint32_t *raw,
                 int64_t n_raw, int64_t *pos, const int64_t *proposals, int32_t *cid,
                 int64_t *medoids, int32_t *accept, double *old_cost,
                 double *new_cost, int64_t *n_ambiguous, int32_t *status);

/* PAM when the frames are sharded over several contexts (one per GPU; the
 * reference's MPI branch of kmedoids.py:575-699 with mpi/ops.py:143-212).  A
 * medoid or a proposal may then be a frame of another shard, so centers are
 * passed as centred center-major coordinates [3 * n_atoms] float32 + trace
 * (float64) in DEVICE memory -- the form ek_centered_frames produces and a
 * collective can move between ranks.
 *  ek_centered_frames: for i < count, row rows[i] of aos_dev / G_dev := centred
 *    coordinates and trace of LOCAL frame local_frames[i];
 *  ek_pam_begin_table: like ek_pam_begin, the medoid table given as
 *    n_medoids rows of aos_dev / G_dev;
 *  ek_pam_prefetch_centers: like ek_pam_prefetch for `count` centers given as
 *    rows of aos_dev / G_dev; the vectors are addressed by slot = row;
 *  ek_pam_propose_center: this shard's part of a proposal (distances of its
 *    frames to the center -- from prefetch slot `slot`, or computed when
 *    slot < 0 --, classification, ambiguous members against the trial table,
 *    cost sums).  n_members_local: members of cluster cid among this shard's
 *    frames.  No read-back: the 32-byte result
 *      { double sum_old, sum_new; int64 n_frames; uint32 n_ambiguous, moved }
 *    is written to out_dev (device memory) for the caller to exchange; the
 *    caller adds the sums over shards (cost = sum / total frames, kmedoids.py:
 *    478-479 with mpi/ops.py:143-166), ORs `moved`, checks n_ambiguous <=
 *    n_members_local, and then calls ek_pam_commit on every shard. */
int ek_centered_frames(ek_ctx *ctx, const int64_t *local_frames,
                       const int32_t *rows, int32_t count, float *aos_dev,
                       double *G_dev);
int ek_pam_begin_table(ek_ctx *ctx, const float *aos_dev, const double *G_dev,
                       int32_t n_medoids);
int ek_pam_prefetch_centers(ek_ctx *ctx, const float *aos_dev,
                            const double *G_dev, int32_t count);
int ek_pam_prefetch_centers_window(ek_ctx *ctx, const float *aos_dev,
                                   const double *G_dev, int32_t count,
                                   int32_t win_lo, int32_t win_count);
int ek_pam_propose_center(ek_ctx *ctx, int32_t cid, int32_t slot,
                          const float *center_aos_dev,
                          const double *center_G_dev, int64_t n_members_local,
                          int32_t win_lo, int32_t win_count, void *out_dev);

/* ---- MSM construction (secondary kernel) -------------------------------------
 * ek_msm_counts replaces assigns_to_counts
 * (enspara/msm/transition_matrices.py:113-170 with _transitions_helper
 * :310-321).  assigns: int32 state index per frame, trajectories concatenated
 * (host memory); lengths[n_trj]: frames per trajectory.  Frames equal to -1
 * are removed from their trajectory first (:156), then state[t] is paired
 * with state[t+lag_time] (sliding_window != 0) or every lag_time-th frame
 * with the next (:316-319).  Every other state must lie in [0, n_states).
 * Output: the count matrix as COO sorted by (row, col) with duplicates summed,
 * at most `capacity` entries (the number of frames is always enough). */
int ek_msm_counts(int device, const int32_t *assigns, const int64_t *lengths,
                  int64_t n_trj, int32_t lag_time, int32_t sliding_window,
                  int32_t n_states, int64_t capacity, int32_t *rows_out,
                  int32_t *cols_out, int64_t *counts_out, int64_t *nnz_out);
/* The same count over the labels a fit left in the context's HBM (the
 * reference pipes result.assignments into assigns_to_counts,
 * transition_matrices.py:113: here they need not leave the device in between).
 * lengths[n_trj]: how the context's frames, in order, split into trajectories
 * (they must add up to its frame count). */
int ek_msm_counts_ctx(ek_ctx *ctx, const int64_t *lengths, int64_t n_trj,
                      int32_t lag_time, int32_t sliding_window, int32_t n_states,
                      int64_t capacity, int32_t *rows_out, int32_t *cols_out,
                      int64_t *counts_out, int64_t *nnz_out);
/* ek_msm_row_normalize replaces _row_normalize's sparse branch
 * (enspara/msm/builders.py:188-196) on a CSR matrix (host arrays):
 * probs = diag(1/rowsum) * data, empty rows stay zero.  rowsum_out may be
 * NULL. */
int ek_msm_row_normalize(int device, const int64_t *indptr, const double *data,
                         int64_t n_rows, double *probs_out, double *rowsum_out);

/* ---- leading eigenpairs of a sparse transition matrix ---------------------------
 * Device primitives of an Arnoldi / Krylov-Schur solver replacing the ARPACK /
 * LAPACK calls of eigenspectrum (enspara/msm/transition_matrices.py:173-233).
 * The operator A is given in CSR (host arrays, copied to the device); the
 * basis V[0..m_max] lives on the device.  The host keeps the projected
 * (m x m) problem only.
 *   ek_krylov_step(j, apply, h_col): w = A V[j] (apply != 0) or w = V[j+1]
 *     (apply == 0); orthogonalise w against V[0..j] (classical Gram-Schmidt,
 *     twice); h_col[0..j] = coefficients, h_col[j+1] = ||w||, V[j+1] = w/||w||.
 *   ek_krylov_rotate(m, kk, Q, move_last): V[0..kk) = V[0..m) Q, Q column-major
 *     m x kk; move_last != 0 also moves V[m] to V[kk] (restart). */
typedef struct ek_krylov ek_krylov;
int ek_krylov_create(int device, int64_t n, const int64_t *indptr,
                     const int32_t *indices, const double *data, int32_t m_max,
                     ek_krylov **out);
int ek_krylov_destroy(ek_krylov *k);
int ek_krylov_set_vector(ek_krylov *k, int32_t j, const double *vec);
int ek_krylov_get_vector(ek_krylov *k, int32_t j, double *vec);
int ek_krylov_step(ek_krylov *k, int32_t j, int32_t apply, double *h_col);
int ek_krylov_rotate(ek_krylov *k, int32_t m, int32_t kk, const double *Q,
                     int32_t move_last);
/* Arnoldi steps j0 .. m-1 enqueued back to back (one synchronisation at the
 * end).  H_out: column-major, leading dimension ldh >= m + 1; column j gets
 * h[0..j+1].  The caller checks the sub-diagonal for breakdowns. */
int ek_krylov_expand(ek_krylov *k, int32_t j0, int32_t m, double *H_out,
                     int32_t ldh);
/* The operator of ek_krylov_step / _expand from here on: A itself (degree 0) or
 * T_degree((A - c) / e), the Chebyshev polynomial on [a, b] = [c - e, c + e] (degree
 * >= 2: `degree` sparse products per step by the three-term recurrence).  What
 * eigenspectrum's restarted iteration uses once it has seen where the wanted
 * eigenvalues end (reference transition_matrices.py:173-233 leaves this to ARPACK):
 * eigenvalues above b are amplified and pulled apart, the rest stay within [-1, 1],
 * so the restarts -- and with them the orthogonalisations, most of the launches --
 * drop by an order of magnitude where the leading eigenvalues cluster at 1.  The
 * eigenpairs returned are A's: Rayleigh-Ritz on the converged subspace (host). */
int ek_krylov_set_filter(ek_krylov *k, int32_t degree, double a, double b);
/* out_host[c][0..n) = V[0..m) Q[:, c], c < kk <= m_max + 1; basis unchanged */
int ek_krylov_combine(ek_krylov *k, int32_t m, int32_t kk, const double *Q,
                      double *out_host);

/* ---- feature-space metrics --------------------------------------------------------
 * Replace the reference's native distance kernels, enspara/geometry/libdist.pyx
 * (_euclidean :122-145, _manhattan :100-117, _hamming :77-95; bound as metrics
 * 'euclidean' / 'manhattan' at enspara/cluster/util.py:292-295).
 * Samples: row-major [n_samples][n_features] host array of elem_kind
 * 0 = float32, 1 = float64, 2 = int64 (hamming only).  ek_feat_distance
 * computes metric 0 = euclidean, 1 = manhattan, 2 = hamming between every
 * sample and the point y (n_features elements of the same kind) into
 * float64 out_host[n_samples]; float32 samples use float32 differences and
 * squares and a float64 running sum in feature order, exactly as the
 * reference's generated C does. */
typedef struct ek_feat ek_feat;
int ek_feat_create(int device, int64_t n_samples, int32_t n_features,
                   int32_t elem_kind, ek_feat **out);
int ek_feat_destroy(ek_feat *k);
int ek_feat_load(ek_feat *k, const void *X, int64_t first, int64_t count);
int ek_feat_distance(ek_feat *k, int32_t metric, const void *y,
                     double *out_host);
/* The k-centers loop itself (kcenters.py:217-231 with the iteration of
 * :243-311) for a feature metric, resident on the device: per center one
 * launch computes metric(X, X[argmax]) with the arithmetic of
 * ek_feat_distance, applies the strict-< update to float64 distances / labels
 * kept in HBM and leaves arg-max partials; a single-workgroup launch reduces
 * them (first index of the maximum), applies the stop rule
 * `distances.max() > dist_cutoff` and fetches the next center's features.
 * dist_io / assign_io: the state on entry (float64 [n], int32 [n]; a fresh run
 * passes +inf / -1) and on return; labels first_label, first_label + 1, ..;
 * at most max_new centers; centers_out[0..*n_added) = the samples chosen;
 * *final_max = distances.max() after the last update. */
int ek_feat_kcenters(ek_feat *k, int32_t metric, int32_t first_label,
                     int32_t max_new, double dist_cutoff, double *dist_io,
                     int32_t *assign_io, int64_t *centers_out, int32_t *n_added,
                     double *final_max);
/* One PAM sweep (reference kmedoids.py:575-699, serial branch) over clusters
 * *cid .. n_medoids - 1 for metric 0 (euclidean) / 1 (manhattan), with the
 * float64 distances, the labels and the medoids' features resident on the device:
 * per proposal the member count, the draw, the distance of every sample to the
 * proposal, the three masks (:644-658), the ambiguous members against all
 * medoids (strict <, ascending medoid index: util.py:199-203), and both costs
 * mean(d**2) in float64 with numpy's pairwise summation (:478-479); the host
 * compares them (:683).  dist_io (float64) / assign_io (int32) are the state:
 * uploaded when *cid == 0, written back when the sweep is through.  proposals ==
 * NULL: proposals are drawn like RandomState.choice(state_inds) from `raw`, the
 * next 32-bit outputs of the caller's RandomState (*pos: outputs consumed).
 * medoids[c] is replaced and accept[c] = 1 where proposal c was accepted.
 * *status: 0 done; 1 `raw` ran out at cluster *cid (call again with more; the
 * state stays on the device); 2 cluster *cid has no member (choice raises).
 * Round 5: for samples beyond 64 MB the proposals go in windows of up to 32 --
 * drawn when the window opens, every sample's distance to each of them in ONE
 * pass; a drawn proposal's window ends where an accepted earlier one moved a
 * sample into or out of its cluster (:611-614 draws from the member list of the
 * moment).  Environment EK_FEAT_PAM_WINDOWS=1 / 0 forces / forbids the form,
 * EK_FEAT_PAM_SYNC=1 is round 3's loop with two waits per proposal; same
 * results in all of them. */
int ek_feat_pam_sweep(ek_feat *k, int32_t metric, int32_t n_medoids, int64_t *medoids,
                      const int64_t *proposals, const uint32_t *raw, int64_t n_raw,
                      int64_t *pos, double *dist_io, int32_t *assign_io,
                      int32_t *accept, int32_t *cid, int32_t *status);
/* frees what ek_feat_pam_sweep keeps between calls (ek_feat_destroy does it too) */
void ek_feat_pam_release(ek_feat *k);

/* ---- tuning knobs (benchmarks only) -------------------------------------- */
/* frames per lane of the distance kernel: 1, 2 or 4; 0 = choose from the
 * shard size */
int ek_set_frames_per_lane(ek_ctx *ctx, int fpl);
/* Options of a context: ek_set_option(ctx, key, value) / ek_get_option.  Every
 * option leaves centers, labels and distances unchanged (each form is tested
 * against the oracle); the ones marked MEASUREMENT exist so that two forms can
 * be timed against each other in one process and are not meant for callers. */
enum ek_option {
    /* non-temporal loads of the frame stream: 0 / 1; -1 = automatic (on when
     * the shard is larger than the Infinity Cache).  MEASUREMENT */
    EK_OPT_NONTEMPORAL = 1,
    /* nearest-center kernel (ek_assign_nearest): 0 automatic, 1 vector FMA, 2
     * MFMA 32x32x2, 3 MFMA 16x16x4 on the quad copy of the frames.
     * MEASUREMENT */
    EK_OPT_ASSIGN_KERNEL = 2,
    /* candidate centers per round of ek_kcenters_run / ek_ms_* / ek_spec_*: -1
     * automatic (up to 16, see EK_OPT_ADAPTIVE), 1 = one-center passes, 4, 8,
     * 16, 32 (32 only on request -- measured, its second sixteen guesses are
     * accepted too rarely --: the frames streamed twice per round, candidates
     * 0..15 and 16..31; the one-launch-per-step forms -- ek_spec_*,
     * EK_OPT_CHAINED = 0, EK_OPT_FUSED_ROUNDS = 0 -- stop at 16).  Every rank
     * of a group must hold the same value (sharded._agree_on_form) */
    EK_OPT_CANDIDATES = 4,
    /* cheap steps of a round in ek_kcenters_run: 1 chained (default), 0 one
     * launch pair per accepted center.  MEASUREMENT */
    EK_OPT_CHAINED = 5,
    /* PAM proposals search the ambiguous members' new medoid only among the
     * medoids within their reach (triangle inequality): 1 (default) / 0.  Used
     * only while the state is known to hold, for every frame, the distance to
     * the medoid its label names: true after ek_state_reset + k-centers, false
     * after ek_state_upload / ek_assign_nearest */
    EK_OPT_PAM_PRUNE = 6,
    /* assert (1) or withdraw (0) that property, e.g. after ek_assign_nearest
     * with the medoid frames themselves as centers */
    EK_OPT_STATE_EXACT = 7,
    /* with EK_OPT_CANDIDATES = -1, let ek_kcenters_run move between 1, 8 and 16
     * candidates per round by measured centers per millisecond: 1 (default) /
     * 0 (always the widest form) */
    EK_OPT_ADAPTIVE = 8,
    /* retired (round 1's pass kernel with the candidates staged in LDS); only
     * the value 1 is accepted */
    EK_OPT_PASS_FORM = 9,
    /* ek_kcenters_run's rounds in three launches (the single-workgroup steps
     * ride at the end of the launch that produces their input): 1 (default) /
     * 0 (one launch per step).  MEASUREMENT */
    EK_OPT_FUSED_ROUNDS = 10,
    /* the reference's `use_triangle_inequality` (kcenters.py:287-296 /
     * :351-364), 0 default / 1: per round the distances of the existing
     * centers to the candidates mark, per tile of 256 frames, the candidates
     * none of whose frames can move (own center at least twice the frame's
     * distance away, with a margin); such (tile, candidate) pairs are not
     * computed and a tile without any is not read.  From a fresh state
     * (ek_state_reset) and for >= 3 atoms.  Also in the sharded one-center
     * iteration (ek_kcenters_step with gathered records): every shard keeps a
     * table of the accepted centers filled from the winning records */
    EK_OPT_TRIANGLE = 11,
    /* ek_pam_window_run works through a window whose prefetch was restricted
     * to a list of frames in ONE workgroup, no launch between two proposals
     * (1, default) or with three launches per proposal (0).  MEASUREMENT */
    EK_OPT_PAM_ONE_WORKGROUP = 12,
    /* the (ambiguous member, medoid within reach) pairs such a workgroup
     * searches itself (default 16384); a proposal with more ends the window
     * and goes through the launches.  0 makes every proposal whose members
     * have another medoid within reach do so (tests) */
    EK_OPT_PAM_MAX_PAIRS = 13,
    /* such a workgroup takes both cost sums (numpy's order) for every proposal
     * (1) or only where the sum of the changes, new^2 - old^2 over the frames
     * the proposal moves, does not decide kmedoids.py:683's comparison by four
     * orders of magnitude more than the rounding of the sums can amount to (0,
     * default).  MEASUREMENT */
    EK_OPT_PAM_BOTH_SUMS = 14,
    /* the next round's candidates are chosen among the farthest frames per 64
     * frames of the state the whole chain leaves (1, default) or per 256 (0).
     * MEASUREMENT */
    EK_OPT_FINE_PICK = 15,
    /* ek_pam_sweep's windows of drawn proposals take the proposal-to-medoid
     * distance table as the lower bounds the medoid-to-medoid table and the
     * proposals' own distances give (triangle inequality; half the pairs of a
     * window's tables): 1 (default) / 0 exact distances.  MEASUREMENT */
    EK_OPT_PAM_BOUNDS = 16,
    /* how many of a label's farthest frames the candidate pick of a round may
     * list (the list of 64 the guesses are chosen from): 0 (default) = 4 or 16
     * by the share of its guesses the run sees accepted (4 suits frames in
     * clouds around templates, 16 a continuous landscape), 1 .. 16 fixed */
    EK_OPT_PICK_CAP = 17,
    /* ek_ms_run's ladder moves from rounds of 8 to 16 once they accept 4.5 (1)
     * or 6.5 (0, default) centers: 1 suits shards of up to ~300 000 frames;
     * every rank of a group must set the same value (sharded.kcenters_sharded
     * does, from the gathered shard sizes) */
    EK_OPT_SMALL_SHARDS = 18,
    /* a PAM window's slots evaluated at once ahead of their turn: 1 (default)
     * / 0 (each in its turn; ek_pam_ahead_stats).  MEASUREMENT */
    EK_OPT_PAM_AHEAD = 19,
    /* a PAM window's record, the next window's member counts and the drawn
     * frames written into mapped host memory by the kernels that make them (1,
     * default) or copied back behind them (0).  MEASUREMENT */
    EK_OPT_PAM_ZERO_COPY = 20,
    /* the distance kernels of a PAM window (tables, listed frames) on the
     * matrix cores, 16 rows x 16 columns per wave (1, default) or LDS-staged
     * 64 x 8 per workgroup (0); process-wide.  MEASUREMENT */
    EK_OPT_PAM_PAIRS_MFMA = 21,
    /* rounds of 16 candidates of ek_kcenters_run: the states every prefix of the
     * round's chain would leave are reduced to their per-tile arg-max by the pass
     * itself, from registers (the chain kernel is then one workgroup that decides
     * and picks, and the presumed order is the order the plan's greedy choice took
     * the candidates in) or by a sweep over the kept distance vectors in the chain
     * kernel (rounds 3-5): 1 (default) the former on shards of up to 524 288 frames
     * -- where it pays: 10 % of a fit at 125 000 frames, nothing at 10^6 --, 2
     * always, 0 never.  MEASUREMENT */
    EK_OPT_PASS_SWEEP = 22,
    /* ek_ms_run (peer mailboxes): an exchange in two steps inside the chain kernel --
     * every shard's per-prefix maxima first; every shard then walks the chain itself and
     * offers the far frames of the state the chain REALLY left -- (1, default) or in one,
     * the offers speculating that the whole chain holds and a chain that broke offered
     * again in an exchange of its own (0: rounds 3-5; what ek_ms_local / ek_ms_global,
     * whose exchange is the caller's collective, always do).  Every rank of a group must
     * hold the same value.  MEASUREMENT */
    EK_OPT_MS_TWO_PHASE = 23
};
int ek_set_option(ek_ctx *ctx, int32_t key, int32_t value);
/* the value an option holds (what ek_set_option stored, or its default) */
int ek_get_option(ek_ctx *ctx, int32_t key, int32_t *value);
/* time of the last ek_kcenters_run loop measured with HIP events on the
 * context's stream, milliseconds, and the number of distance-kernel launches
 * it covered */
int ek_last_run_timing(ek_ctx *ctx, float *ms, int32_t *launches);

/* Sampled timing of the distance kernel alone: after ek_timing_begin, every
 * `sample_every`-th ek_kcenters_step brackets its distance-kernel launch with
 * a HIP event pair on the context's stream (at most max_samples pairs).
 * ek_timing_end synchronises and returns the mean elapsed time per sampled
 * launch in milliseconds -- of the launches with the number of candidates per
 * pass that was sampled most, which ek_timing_form then reports (a run moves
 * between 1, 8 and 16 candidates per pass). */
int ek_timing_begin(ek_ctx *ctx, int32_t sample_every, int32_t max_samples);
int ek_timing_end(ek_ctx *ctx, float *avg_ms, int32_t *n_samples);
int ek_timing_form(ek_ctx *ctx, int32_t *candidates);
/* gbytes_per_s[0]: read + write rate (GB/s) of a 16-byte-per-lane non-temporal
 * copy of `bytes` bytes on this GPU; [1]: the rate of only reading them (what the
 * distance kernels do); best of four each: the measured ceilings bench.py quotes
 * beside the nominal HBM peak */
int ek_hbm_copy_rate(int device, size_t bytes, double *gbytes_per_s);
/* Test entry: the QCP arithmetic of csrc/ek_qcp.h evaluated by a KERNEL on m
 * inner-product matrices handed over by the caller (host arrays: S [m][9] f32,
 * Gx / Gy [m] f64 traces, cur [m] f32 the distance each solve may stop above) --
 * full[i] = ek_rmsd_from_S, below[i] = ek_rmsd_from_S_below(.., cur[i]),
 * cert[i] = ek_far_certified_f32(S, (float)(Gx + Gy), n_atoms, cur[i]) in bit 0 and
 * ek_far_certified2_f32 (the second level) in bit 1 -- so that
 * the soundness tests of the early-stopped solve and of the float32 certificate
 * run through the instructions that ship (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32:
 * 1 ulp, where the host build of the header rounds correctly).  Stands where
 * mdtraj.rmsd's own per-pair solve would (cluster/util.py:289-291); not used by
 * the product path. */
int ek_qcp_probe(int device, const float *S, const double *Gx, const double *Gy,
                 int32_t n_atoms, const float *cur, int64_t m, float *full,
                 float *below, unsigned char *cert);

/* DEBUG (tools/fuzz_*.py).  With EK_POISON=1 in the environment when the library starts,
 * every working buffer of a context starts as 0x5a bytes (EK_POISON_BYTE) and is followed by a guard page;
 * this returns the number of buffers something wrote past the end of (their names go to
 * stderr), 0 without EK_POISON. */
int ek_debug_guards(ek_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ENSPARA_HIP_H */
