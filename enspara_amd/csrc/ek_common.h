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
    int32_t pad;
};

// ---- kernel launchers (defined in the .hip files) ---------------------------
// centring + trace + frame-minor transposition of `count` AoS frames
void ek_launch_prepare_tiles(const float *src_aos, int64_t count, int A,
                             float *tiles, double *G, int64_t first_frame,
                             int64_t n_total, hipStream_t s);
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
                    EkHist *hist, EkCtl *ctl, hipStream_t s);
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
                           float *dist, int32_t *assign, int ablate,
                           hipStream_t s);

// ---- PAM (ek_pam.hip) ----------------------------------------------------------
void ek_launch_gather_frames(const float *tiles, const double *G, int A,
                             const int64_t *idx_dev, int count, int first_row,
                             float *out_aos, double *outG, hipStream_t s);
void ek_launch_copy_row(float *aos, double *Gm, int A, int src, int dst,
                        hipStream_t s);
void ek_launch_count_members(const int32_t *assign, int64_t n, int32_t cid,
                             int32_t *blockcnt, int64_t *scan, int64_t *total,
                             hipStream_t s);
void ek_launch_select_member(const int32_t *assign, int64_t n, int32_t cid,
                             const int64_t *scan, int64_t j, int64_t *out,
                             hipStream_t s);
void ek_launch_pam_classify(const float *dist, const int32_t *assign,
                            const float *newd, int64_t n, int32_t cid,
                            float *ndist, int32_t *nassign, uint32_t *amb,
                            unsigned long long *amb_best,
                            unsigned int *amb_count, hipStream_t s);
// n_amb is read on the device; max_amb (a host-side upper bound) sizes the grid
// ambt [3A][cap] / ambG [cap]: scratch for the compacted frames, cap >= max_amb
void ek_launch_subset_assign(const float *tiles, const double *G, int A,
                             const uint32_t *amb, const unsigned int *n_amb,
                             int64_t max_amb, float *ambt, double *ambG,
                             int64_t cap, const float *centers,
                             const double *Gc, int K,
                             unsigned long long *amb_best, hipStream_t s);
void ek_launch_pam_scatter(const uint32_t *amb,
                           const unsigned long long *amb_best,
                           const unsigned int *n_amb, int64_t max_amb,
                           float *ndist, int32_t *nassign, hipStream_t s);
#define EK_SUMSQ_PART_DOUBLES 2048
void ek_launch_sumsq2(const float *a, const float *b, int64_t n, double *part,
                      double *out, hipStream_t s);
