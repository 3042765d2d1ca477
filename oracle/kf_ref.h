/*
 * kf_ref — CPU restatement of the SLAMBench `kfusion` C++ back-end (KinectFusion).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of the kfusion use-case is not vendored in the
 * reference tree (it is cloned at build time, benchmarks/benchmarks.repos:3 and
 * framework/makefiles/benchmarks.makefile:162-174 of the reference; TooN pinned at
 * framework/makefiles/toon.make:8) and the reference holds no per-kernel golden
 * vectors.  The restatement follows the published KinectFusion algorithm as
 * restructured by the PAMELA project (pamela-project/kfusion, branch
 * update-master; TooN SE3/GR_SVD semantics).  What the reference tree does pin
 * is checked in tests/: parameter names and defaults, phase scheduling
 * (icra2018_results/1080/violons_libkfusion-cpp_living_room_traj0_loop.log:20-46)
 * and the harness-side ATE definition (framework/shared/src/metrics/ATEMetric.cpp:51-167).
 */
#ifndef KF_REF_H_
#define KF_REF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order and types as kfh_config (include/kfusion_hip.h). */
typedef struct kfr_config {
    int32_t input_width, input_height;  /* sensor frame size, e.g. 640x480          */
    int32_t compute_size_ratio;         /* 1, 2, 4 or 8                             */
    float   fx, fy, cx, cy;             /* pinhole intrinsics in INPUT pixels       */
    float   depth_divisor;              /* raw depth units per metre (1000 for mm)  */
    int32_t vol_res[3];                 /* voxels per side                          */
    float   vol_dim[3];                 /* metres per side                          */
    float   init_pos[3];                /* initial camera position in the volume    */
    float   mu;                         /* TSDF truncation distance, metres         */
    float   icp_threshold;              /* convergence bound on |twist|             */
    int32_t pyramid[3];                 /* ICP iterations, [0] = finest level       */
    int32_t integration_rate, tracking_rate;
    int32_t integrate_filtered;         /* 0: fuse raw metres depth, 1: bilateral-filtered */
    int32_t slab_z0, slab_z1;           /* unused by the oracle (full volume)       */
} kfr_config;

typedef struct kfr_track_data {
    int32_t result;
    float   error;
    float   J[6];
} kfr_track_data;

/* ---- stateless kernels (host pointers; float3 arrays are packed 12-byte) ---- */
void kfr_mm2meters(float *out, int ow, int oh, const uint16_t *in, int iw, int ih, float divisor);
void kfr_bilateral(float *out, const float *in, int w, int h, int radius, float delta, float e_d);
void kfr_half_sample(float *out, const float *in, int iw, int ih, float e_d, int r);
void kfr_depth2vertex(float *vertex3, const float *depth, int w, int h, const float invK[16]);
void kfr_vertex2normal(float *normal3, const float *vertex3, int w, int h);
void kfr_track(kfr_track_data *out, int out_stride,
               const float *in_vertex3, const float *in_normal3, int iw, int ih,
               const float *ref_vertex3, const float *ref_normal3, int rw, int rh,
               const float Ttrack[16], const float view[16],
               float dist_threshold, float normal_threshold);
void kfr_reduce(float out[8 * 32], const kfr_track_data *J, int j_stride, int w, int h);
void kfr_solve(double x[6], const float vals[27]);
int  kfr_update_pose(float pose[16], const float sums[32], float icp_threshold);
int  kfr_check_pose(float pose[16], const float old_pose[16], const float sums[32],
                    int w, int h, float track_threshold);
void kfr_integrate(int16_t *vol, const int32_t res[3], const float dim[3],
                   const float *depth, int w, int h,
                   const float invTrack[16], const float K[16], float mu, float maxweight);
void kfr_raycast(float *vertex3, float *normal3, int w, int h,
                 const int16_t *vol, const int32_t res[3], const float dim[3],
                 const float view[16], float near_plane, float far_plane,
                 float step, float largestep);
/* debug images of the use-case (RGBA8, alpha 0): depth colour map, tracking status, shaded second raycast */
void kfr_render_depth(uint8_t *rgba, const float *depth, int w, int h, float near_plane, float far_plane);
void kfr_render_track(uint8_t *rgba, const kfr_track_data *data, int w, int h);
void kfr_render_volume(uint8_t *rgba, int w, int h, const int16_t *vol, const int32_t res[3], const float dim[3],
                       const float view[16], float near_plane, float far_plane, float step, float largestep,
                       const float light[3], const float ambient[3]);
void kfr_camera_matrix(float K[16], float fx, float fy, float cx, float cy);
void kfr_inverse_camera_matrix(float invK[16], float fx, float fy, float cx, float cy);
void kfr_mat_inverse(float out[16], const float in[16]);
void kfr_mat_mul(float out[16], const float a[16], const float b[16]);
void kfr_se3_exp(float out[16], const double x[6]);

/* ---- whole pipeline ---- */
typedef struct kfr_context kfr_context;
kfr_context *kfr_create(const kfr_config *cfg);
void kfr_destroy(kfr_context *c);
/* One frame: preprocess -> track -> integrate -> raycast, in that order.
 * flags bit0 tracked, bit1 integrated, bit2 raycasted.  phase_seconds[4] (may be NULL)
 * receives wall time of the four phases. */
int  kfr_process_frame(kfr_context *c, const uint16_t *depth, double *phase_seconds);
void kfr_get_pose(const kfr_context *c, float pose[16]);
void kfr_set_pose(kfr_context *c, const float pose[16]);
int  kfr_frame_index(const kfr_context *c);
int  kfr_set_threads(int n);   /* OpenMP worker threads (n <= 0: query only); returns the count in force */
/* buffer accessors (pointers stay owned by the context) */
const float   *kfr_float_depth(const kfr_context *c);
const float   *kfr_scaled_depth(const kfr_context *c, int level);
const float   *kfr_input_vertex(const kfr_context *c, int level);
const float   *kfr_input_normal(const kfr_context *c, int level);
const float   *kfr_model_vertex(const kfr_context *c);
const float   *kfr_model_normal(const kfr_context *c);
const int16_t *kfr_volume(const kfr_context *c);
const float   *kfr_reduction(const kfr_context *c);
int kfr_icp_iterations(const kfr_context *c);                       /* track + reduce passes of the last tracked-rate frame */
double kfr_cutoff_margin(const kfr_context *c);                    /* smallest |ln(|w_i| * 1e6 / max|w|)| over the solves of that frame: how close a keep/drop decision of the singular-value cut-off came to its limit */
double kfr_last_cutoff_margin(void);                               /* the same for the last kfr_solve of the calling thread */
void kfr_set_reduce_rows(int rows);                                /* test aid: kfr_reduce sums over `rows` strided partial rows (1..8) instead of the reference's 8: the same sums in another order */
const kfr_track_data *kfr_tracking_result(const kfr_context *c);   /* [h][w] of the compute size, last iteration run */

#ifdef __cplusplus
}
#endif
#endif
