// Internal launch interface between the kernel translation unit and the context / C ABI.
#ifndef KFH_KERNELS_H_
#define KFH_KERNELS_H_

#include <cstdio>
#include <cstdlib>

#include "kfh_device.h"
#include "../../include/kfusion_hip.h"

namespace kfh {

// State of the ICP loop between two iteration launches (see k_icp_iteration).
struct IcpState {
    Mat4 pose;            // estimate after the last finalised iteration
    float sums[32];       // reduction of the last finalised iteration
    int pending_nblocks;  // partial rows written by the previous launch that still await their solve
    int pending_level;    // pyramid level of that iteration
    int pending_last;     // it was the last iteration configured for its level
    int done_mask;        // bit l: level l finished (converged or exhausted) in this frame
    int iterations;       // iterations finalised in this frame
    int slow_solves;      // of those, solves that needed the eigen-decomposition path
    int pad_[2];
};

// Device-resident control block of one pipeline context.  The host never needs to read it inside a
// frame: every decision that depends on tracking results is taken by the kernels themselves.
struct Ctrl {
    Mat4 pose;          // camera-to-world, current estimate
    Mat4 old_pose;      // pose saved at the start of tracking
    Mat4 raycast_pose;  // pose the model maps were rendered from
    Mat4 project_ref;   // K * inverse(raycast_pose)
    Mat4 inv_pose;      // inverse(pose), input of integrate
    Mat4 view;          // pose * inverse(K), input of raycast
    Mat4 raycast_view;  // the `view` the model maps were rendered with (rays are rebuilt from it in tracking)
    float sums[32];     // last ICP reduction
    int tracked;        // result of the pose check of this frame
    int integrate_gate; // this frame fuses its depth map: (tracked and rate hit) or frame <= 3
    int frame;          // frame counter (device copy)
    unsigned ray_task_count;  // rays the first raycast pass handed to the second (reset by the tracking finish)
    unsigned integrate_box_count[2];  // work items the integrate plan launch made: [0] the long ones, appended from the
                                      //   front of the list, [1] the short ones, from its back (reset by the tracking finish)
    int pad_[2];
    // running totals since the context was created or reset (never cleared by the frame kernels): boxes the plan
    // launches kept, and plan launches whose gate was open.  A benchmark reads them before and after a run.
    unsigned long long integrate_boxes_total;
    unsigned long long integrate_launches_total;
    unsigned long long integrate_boxes_before;   // integrate_boxes_total when the last integration started
    unsigned long long integrate_boxes_last;     // boxes the last integration kept: sizes the next one's work items
    IcpState icp[2];    // alternating records of the ICP loop
};

// Tracking block: wavefront 0 finalises the previous iteration (row sum, 6x6 solve, pose update) while
// PW pixel wavefronts (3 to 8) prepare and then evaluate the pixels of a 64 x (PW * PPT) tile, PPT rows per
// thread.  A SIMD issues one vector instruction every 4 cycles for a lone wavefront but every 2 cycles with
// two, so 8 pixel wavefronts (two per SIMD) evaluate twice the tile in the same time.
constexpr int KFH_TRACK_TILE_W = 64;
constexpr int KFH_TRACK_MAX_THREADS = 64 * (1 + 8);
constexpr int KFH_SUM_THREADS = 256;      // block size of the stand-alone row sum / finish kernels
constexpr int KFH_STATUS_RING = 4096;
// diagnostic time stamps: [launch][block][slot]
constexpr int KFH_STAMP_LAUNCHES = 64, KFH_STAMP_BLOCKS = 320, KFH_STAMP_SLOTS = 14;

inline int track_blocks(int w, int h, int pw, int ppt) {
    const int tile_h = pw * ppt;
    return ((w + KFH_TRACK_TILE_W - 1) / KFH_TRACK_TILE_W) * ((h + tile_h - 1) / tile_h);
}
// Tile shape for an image: the instantiated (pixel wavefronts, rows per thread) pair with the fewest rows per
// thread whose grid still fits on `max_blocks` compute units at one block each (no unit gets a second tile);
// ties go to the smaller block.
inline void track_shape(int w, int h, int max_blocks, int *pw, int *ppt) {
    static const int shapes[][2] = {{3, 1}, {4, 1}, {5, 1}, {6, 1}, {8, 1}, {4, 2}, {8, 2}, {4, 3}, {7, 3}, {8, 3}, {4, 5}, {8, 5}, {4, 8}, {8, 8}};
    // experiments: KFH_TRACK_SHAPE_<image width>=pw,ppt forces an (instantiated) shape for images of that width
    {
        char name[40];
        std::snprintf(name, sizeof(name), "KFH_TRACK_SHAPE_%d", w);
        const char *v = std::getenv(name);
        int fp = 0, fq = 0;
        if (v && std::sscanf(v, "%d,%d", &fp, &fq) == 2 && fp > 0 && fq > 0) {
            *pw = fp;
            *ppt = fq;
            return;
        }
    }
    for (const auto &sh : shapes)
        if (track_blocks(w, h, sh[0], sh[1]) <= max_blocks) {
            *pw = sh[0];
            *ppt = sh[1];
            return;
        }
    *pw = 8;
    *ppt = 8;
}

struct IcpIterArgs {
    Ctrl *ctrl;
    const float *partials_in;   // rows of the previous iteration launch
    float *partials_out;        // rows of this one
    int pending_rows;           // rows the previous launch wrote if it evaluated anything (its grid size; 0 for seq 0)
    int seq;                    // position of this launch in the frame's iteration sequence
    int level, last_iter;       // pyramid level; last iteration configured for the level
    const float *in_depth;      // pyramid level of the filtered depth (the input vertex is recomputed from it)
    const float4 *in_normal;    // input normal map; null: normals are recomputed from in_depth as well
    float k[4];                 // intrinsics of this level
    float ik[4];                // set by the launcher: 1/fx, -cx/fx, 1/fy, -cy/fy
    int iw, ih;
    const float4 *ref_vertex, *ref_normal;   // model maps, or
    const float4 *ref_tn;       // (ref_vertex == null) packed model map {hit distance, normal}; the model vertex
                                // is rebuilt as origin + direction * t from ctrl->raycast_view
    int rw, rh;
    float dist_threshold, normal_threshold, icp_threshold;
    float dist2_max;            // set by the launcher: largest d2 with sqrtf(d2) <= dist_threshold
    int pw, ppt;                // pixel wavefronts per block and rows per thread; 0 = chosen by the launcher
    int max_blocks;             // compute units of the device (for that choice)
    kfh_track_data *track_out;  // per-pixel TrackData for parity tests, may be null
    int track_stride;
    signed char *result_out;    // per-pixel result code of trackKernel (1, -1 .. -5) for the tracking render, may be null
    int result_stride;
    unsigned long long *stamps; // diagnostic time stamps [launch][block][slot], may be null
    // A device that evaluates only a share of the image (row-sharded tracking across devices): first tile and
    // number of tiles of its share (0 = all from block0 on), and whether a launch between two iterations folds
    // the partial rows of all devices into ONE row (then the next launch finds pending_nblocks == 1).
    int block0, block_count, rows_reduced;
};

struct FinishArgs {
    Ctrl *ctrl;
    const float *partials_in;
    int pending_rows;         // grid size of the last iteration launch
    int seq;                  // number of iteration launches of this frame
    int w, h;                 // compute size (pose check normalisation)
    float icp_threshold;
    float track_threshold;
    float k[4];               // intrinsics at compute size
    int will_raycast;         // frame > 2: the model is re-rendered from the checked pose
    int frame;
    int integration_rate;
    kfh_frame_status *status_ring;
    unsigned *mail;           // host-visible record {pose[16], tracked, ..., [31] = mail_seq}, or null: see PoseMail
    unsigned mail_seq;
};

// The accepted pose of a frame in host memory the device can write (pinned, mapped, coherent): a caller that waits
// for the pose after every frame reads it from here once the stream has drained, instead of paying for a
// device-to-host copy of 64 bytes (the copy engine's hand-overs cost more than a frame's integration).
struct PoseMail {
    float pose[16];
    int tracked;
    int pad_[14];
    unsigned seq;             // written last: the launch number the record belongs to
};
static_assert(sizeof(PoseMail) == 128, "PoseMail is addressed as 32 dwords");

// level1 / level2 (both or neither): also write the two robust half-samplings of the filtered image
hipError_t launch_preprocess(float *raw, float *filtered, int ow, int oh, const void *in, bool in_is_u16,
                             int iw, int ratio, float divisor, float *level1, float *level2, float pyramid_e_d,
                             float *tile_max, hipStream_t s);
// floats of the per-tile depth maxima launch_preprocess writes (tile_max != null) and launch_integrate reads
size_t depth_tile_count(int w, int h);
hipError_t launch_half_sample(float *out, const float *in, int iw, int ih, float e_d, hipStream_t s);
hipError_t launch_vertex_normal(float4 *vertex, float4 *normal, const float *depth, int w, int h, float fx,
                                float fy, float cx, float cy, hipStream_t s);
hipError_t launch_icp_iteration(const IcpIterArgs &a, hipStream_t s);
bool icp_stamps_compiled();   // the library was built with its diagnostics: ICP time stamps, per-wavefront raycast / integrate records (make STAMPS=1)
hipError_t launch_icp_finish(const FinishArgs &a, hipStream_t s);
// tracking skipped (tracking-rate) or no iterations configured: record "not tracked" and derive matrices
hipError_t launch_track_skip(const FinishArgs &a, hipStream_t s);
// one stand-alone iteration on caller-provided matrices + fixed-order sum of its partial rows (parity tests)
hipError_t launch_track_reduce(Ctrl *scratch_ctrl, float *partials, float *sums, kfh_track_data *track_out,
                               int track_stride, const float *in_depth, const float k[4], const float4 *in_normal,
                               int iw, int ih,
                               const float4 *ref_vertex, const float4 *ref_normal, int rw, int rh,
                               const Mat4 &Ttrack, const Mat4 &view, float dist_threshold, float normal_threshold,
                               hipStream_t s);
hipError_t launch_sum_partials(float *sums, const float *partials, int nrows, hipStream_t s);
hipError_t launch_update_pose(float *pose16, double *x6, int *converged, const float *sums, float thr,
                              hipStream_t s);
hipError_t launch_update_pose_chain(float *pose16, const float *sums, float thr, int n, hipStream_t s);
hipError_t launch_pose_derive(Ctrl *ctrl, const float k[4], int set_raycast_pose, hipStream_t s);
// Two launches: a plan launch lists the 64 x 4 x zchunk boxes that can touch the camera frustum, the work launch
// fuses the depth map into those.  `boxes` holds integrate_box_list_bytes() bytes and box_count[0], box_count[1] must be zero
// when the first launch starts.
size_t integrate_box_list_bytes(const int res[3], int z1);
hipError_t launch_integrate(short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                            const float *depth, int w, int h, const Mat4 *inv_track, Mat4 K, float mu,
                            float maxweight, const int *gate, void *boxes, unsigned *box_count,
                            unsigned char *bricks, const float *tile_max, hipStream_t s,
                            unsigned long long *totals = nullptr);   // &Ctrl::integrate_boxes_total (the four running totals)
// voxels of one box of the integrate work list for a volume whose last stored plane is z1 - 1
int integrate_box_voxels(int z1);
// One byte per 8^3-voxel brick ("nothing a coarse raycast step could notice in here"): integrate clears marks,
// raycast skips marked bricks.  Optional (null) in both launches; indexed by the whole volume's brick grid (a slab's
// owner only marks bricks whose planes, and the plane above them, it stores).
inline size_t brick_mark_bytes(const int res[3]) {
    return (size_t)((res[0] + 7) / 8) * ((res[1] + 7) / 8) * ((res[2] + 7) / 8);
}
// Two launches: per-pixel march, then the rays it left unfinished.  `tasks` holds raycast_task_bytes(w, h)
// bytes and *task_count must be zero when the first launch starts.  Of the four output maps only those given
// (non-null) are written; launch_model_maps rebuilds vertex, normal and ray_t from ref_tn and the view matrix the
// raycast used.
size_t raycast_task_bytes(int w, int h);
hipError_t launch_raycast(float4 *vertex, float4 *normal, float *ray_t, float4 *ref_tn, int w, int h,
                          const short2 *vol, const int res[3], const float dim[3], int z0, int z1, const Mat4 *view,
                          float near_plane, float far_plane, float step, float largestep, void *tasks,
                          unsigned *task_count, const unsigned char *bricks, hipStream_t s);
// The two launches separately, for a caller that postpones the second pass: nothing reads the maps between the
// end of one frame and the tracking of the next, so the second pass may wait for the next frame's preprocessing
// and share its launch (RAYCAST_TAIL with `with_preprocess`: what launch_preprocess would be given, u16 input
// and both pyramid levels required).
enum { RAYCAST_BOTH = 0, RAYCAST_MARCH = 1, RAYCAST_TAIL = 2 };
struct PreprocessLaunch {
    float *raw, *filtered;
    int ow, oh;
    const void *in;
    int iw, ratio;
    float divisor;
    float *level1, *level2;
    float pyramid_e_d;
    float *tile_max;
};
hipError_t launch_raycast_part(float4 *vertex, float4 *normal, float *ray_t, float4 *ref_tn, int w, int h,
                               const short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                               const Mat4 *view, float near_plane, float far_plane, float step, float largestep,
                               void *tasks, unsigned *task_count, const unsigned char *bricks, int part,
                               const PreprocessLaunch *with_preprocess, hipStream_t s);
hipError_t launch_model_maps(float4 *vertex, float4 *normal, float *ray_t, const float4 *ref_tn, const Mat4 *view,
                             int w, int h, hipStream_t s);
hipError_t launch_composite(float4 *ref_tn, const float4 *tn_parts, int n_parts_count, int npix, hipStream_t s);
// the same with one pointer per slab (own or peer memory); at most KFH_GROUP_MAX slabs
constexpr int KFH_GROUP_MAX = 16;
hipError_t launch_composite_parts(float4 *ref_tn, const float4 *const *parts, int n, int npix, hipStream_t s);
hipError_t launch_composite_share(float4 *const *outs, const float4 *const *parts, int n, int first, int count,
                                  hipStream_t s);
// diagnostics: per-wavefront records of every following raycast launch go to d_buffer (null: off)
void set_raycast_debug(unsigned long long *d_buffer);
void set_integrate_debug(unsigned long long *d_buffer);
hipError_t launch_fill_volume(short2 *vol, size_t nvox, hipStream_t s);
// copy `bytes` (a multiple of 16) from pinned host memory to device memory with a kernel
hipError_t launch_stage_copy(void *dst, const void *pinned_src, size_t bytes, hipStream_t s);
// The three images of the reference GUI (renderDepthKernel, renderTrackKernel, the shading half of
// renderVolumeKernel; its raycast half is launch_raycast into scratch maps).  RGBA, alpha 0.
hipError_t launch_render_depth(uchar4 *out, const float *depth, int npix, float near_plane, float far_plane,
                               hipStream_t s);
// exactly one of data (row stride `stride` elements) and codes (row stride `stride` bytes) is non-null
hipError_t launch_render_track(uchar4 *out, const kfh_track_data *data, const signed char *codes, int stride, int w,
                               int h, hipStream_t s);
hipError_t launch_render_shade(uchar4 *out, const float4 *vertex, const float4 *normal, int npix,
                               const float light[3], const float ambient[3], hipStream_t s);
// a matrix handed over by value (no host-to-device copy) and, optionally, one counter set to zero
hipError_t launch_store_mat4(Mat4 *d_out, const Mat4 &m, unsigned *d_zeroed, hipStream_t s);
// view = pose * inverse camera matrix, written to device memory for launch_raycast
hipError_t launch_view_from_pose(Mat4 *d_view, const Mat4 &pose, const float k[4], hipStream_t s);

}  // namespace kfh
#endif
