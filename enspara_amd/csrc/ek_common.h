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
void ek_launch_step(int fpl, int mode, const float *tiles, const double *G,
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

// record from a given local frame (center = frame index)
void ek_launch_record_from_frame(const float *tiles, const double *G, int A,
                                 int64_t local_idx, int64_t global_offset,
                                 unsigned char *rec, hipStream_t s);
// record from centred center-major coordinates (center = external structure)
void ek_launch_record_from_center(const float *center_aos, const double *Gc,
                                  int A, unsigned char *rec, hipStream_t s);

void ek_launch_fill_state(float *dist, int32_t *assign, int64_t n, float d,
                          int32_t a, hipStream_t s);

// all frames x K centers, strict-< in ascending center order
void ek_launch_assign(const float *tiles, const double *G, int64_t n, int A,
                      const float *centers_aos, const double *Gc, int32_t K,
                      float *dist, int32_t *assign, hipStream_t s);
