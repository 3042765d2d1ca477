// Internal launch interface between the kernel translation unit and the context / C ABI.
#ifndef KFH_KERNELS_H_
#define KFH_KERNELS_H_

#include "kfh_device.h"
#include "../../include/kfusion_hip.h"

namespace kfh {

// Device-resident control block of one pipeline context.  The host never needs to read it inside a
// frame: every decision that depends on tracking results is taken by the kernels themselves.
struct Ctrl {
    Mat4 pose;          // camera-to-world, current estimate
    Mat4 old_pose;      // pose saved at the start of tracking
    Mat4 raycast_pose;  // pose the model maps were rendered from
    Mat4 project_ref;   // K * inverse(raycast_pose)
    Mat4 inv_pose;      // inverse(pose), input of integrate
    Mat4 view;          // pose * inverse(K), input of raycast
    float sums[32];     // last ICP reduction
    double x[6];        // last twist
    int level_done[4];  // frame + 1 of the frame in which the ICP level finished (converged or exhausted)
    int tracked;        // result of the pose check of this frame
    int integrate_gate; // tracked, or frame <= 3
    int frame;          // frame counter (device copy)
    int pad_;
};

struct IcpLevel {
    const float4 *in_vertex;
    const float4 *in_normal;
    int w, h;
};

constexpr int KFH_TRACK_THREADS = 256;
constexpr int KFH_TRACK_PIX_PER_THREAD = 4;
constexpr int KFH_STATUS_RING = 4096;

inline int track_blocks(int w, int h) {
    const int per_block = KFH_TRACK_THREADS * KFH_TRACK_PIX_PER_THREAD;
    return (w * h + per_block - 1) / per_block;
}

hipError_t launch_preprocess(float *raw, float *filtered, int ow, int oh, const void *in, bool in_is_u16,
                             int iw, int ratio, float divisor, hipStream_t s);
hipError_t launch_half_sample(float *out, const float *in, int iw, int ih, float e_d, hipStream_t s);
hipError_t launch_vertex_normal(float4 *vertex, float4 *normal, const float *depth, int w, int h, float fx,
                                float fy, float cx, float cy, hipStream_t s);
// one ICP iteration, part 1: residuals + per-block partial sums.  `done_flag` (may be null) makes the
// kernel a no-op when *done_flag == done_value.
hipError_t launch_track_reduce(float *partials, kfh_track_data *track_out, int track_stride,
                               const float4 *in_vertex, const float4 *in_normal, int iw, int ih,
                               const float4 *ref_vertex, const float4 *ref_normal, int rw, int rh,
                               const Mat4 *Ttrack, const Mat4 *view, float dist_threshold,
                               float normal_threshold, const int *done_flag, int done_value, hipStream_t s);
// part 2: fixed-order sum of the partials into sums[32]
hipError_t launch_sum_partials(float *sums, const float *partials, int nblocks, hipStream_t s);
// part 2 of a pipeline iteration: sum, solve, update pose, convergence, optional pose check + derived matrices
struct SolveArgs {
    Ctrl *ctrl;
    const float *partials;
    int nblocks;
    int level;
    int last_iter_of_level;   // set level_done after this iteration regardless of convergence
    int final_level;          // this level ends tracking: run the pose check when the level finishes
    int w, h;                 // compute size (pose check normalisation)
    float icp_threshold;
    float track_threshold;
    float k[4];               // intrinsics at compute size
    int will_raycast;         // frame > 2: the model is re-rendered from the checked pose
    int frame;
    int integration_rate;
    kfh_frame_status *status_ring;
};
hipError_t launch_icp_solve(const SolveArgs &a, hipStream_t s);
// tracking skipped (tracking-rate) or no iterations configured: record "not tracked" and derive matrices
hipError_t launch_track_skip(const SolveArgs &a, hipStream_t s);
hipError_t launch_update_pose(float *pose16, double *x6, int *converged, const float *sums, float thr,
                              hipStream_t s);
hipError_t launch_pose_derive(Ctrl *ctrl, const float k[4], int set_raycast_pose, hipStream_t s);
hipError_t launch_integrate(short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                            const float *depth, int w, int h, const Mat4 *inv_track, Mat4 K, float mu,
                            float maxweight, const int *gate, hipStream_t s);
hipError_t launch_raycast(float4 *vertex, float4 *normal, float *ray_t, int w, int h, const short2 *vol,
                          const int res[3], const float dim[3], int z0, int z1, const Mat4 *view,
                          float near_plane, float far_plane, float step, float largestep, hipStream_t s);
hipError_t launch_composite(float4 *vertex, float4 *normal, float *ray_t, const float *t_parts,
                            const float4 *v_parts, const float4 *n_parts, int n_parts_count, int npix,
                            hipStream_t s);
hipError_t launch_fill_volume(short2 *vol, size_t nvox, hipStream_t s);

}  // namespace kfh
#endif
