/*
 * kfusion_hip.h — C ABI of the gfx950 KinectFusion core (libkfusion_hip_core.so).
 *
 * This is the inner boundary of the drop-in: the SLAMBench plugin adapter
 * (slambench_amd/plugin/kfusion_hip_library.cpp) implements the reference's six plugin entry
 * points on top of it, and tests / the benchmark call it directly through ctypes.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   - plugin entry points  framework/shared/include/SLAMBenchAPI.h:30-54  (sb_new_slam_configuration,
 *     sb_init_slam_system, sb_update_frame, sb_process_once, sb_update_outputs, sb_clean_slam_system)
 *     and their C wrappers framework/shared/src/library_wrapper.cpp:15-44;
 *   - the kfusion use-case the harness clones at build time (benchmarks/benchmarks.repos:3):
 *     class Kfusion {preprocessing, tracking, integration, raycasting} and its kernels
 *     (mm2metersKernel, bilateralFilterKernel, halfSampleRobustImageKernel, depth2vertexKernel,
 *     vertex2normalKernel, trackKernel, reduceKernel, updatePoseKernel, checkPoseKernel,
 *     integrateKernel, raycastKernel).  That source is not vendored in the reference tree, so
 *     those names are cited without line numbers.
 *
 * Conventions
 *   - every function returns KFH_OK (0) or a KFH_ERR_* code; kfh_last_error() gives the text;
 *   - matrices are 4x4 float, row-major (camera-to-world for poses);
 *   - images are row-major; vertex / normal maps are float4-strided (x, y, z, unused) in device
 *     memory; an invalid normal has x == -2;
 *   - the TSDF volume is short2 per voxel (x: tsdf * 32766, y: weight), index x + y*X + z*X*Y;
 *   - pointers named d_* are device pointers, h_* host pointers; `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream);
 *   - a context belongs to the device that was current in kfh_create; every kfh_* call on it makes
 *     that device current for its own duration and restores the caller's, so contexts of several
 *     devices can be driven from one thread or from one thread each.  Calls on ONE context must not
 *     overlap; calls on different contexts may (kfh_last_error() is per thread);
 *   - the staged calls follow Kfusion's order (kfh_preprocess, kfh_track, kfh_integrate, kfh_raycast,
 *     kfh_end_frame); kfh_integrate and kfh_raycast may also be repeated or called after kfh_set_pose without
 *     a kfh_track in between (a second fusion of the same depth map, a re-render from another pose): each launch
 *     then empties its own work list first, and the raycast makes its pose the one tracking will read the
 *     model maps with;
 *   - everything is queued on the context's stream and nothing waits unless said so.  The second
 *     pass of a frame's raycast may stay unqueued until the next frame's preprocessing (it then
 *     shares that launch) or until a call that reads the model maps, waits for the stream, returns
 *     a device pointer or the stream, or changes pose or volume: all of these queue it first, so
 *     results never depend on it.  Work queued on the stream from kfh_stream() by the caller
 *     after further kfh_* calls must ask for the stream again.
 */
#ifndef KFUSION_HIP_H_
#define KFUSION_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KFH_OK               0
#define KFH_ERR_INVALID_ARG  1
#define KFH_ERR_HIP          2
#define KFH_ERR_STATE        3
#define KFH_ERR_NO_DEVICE    4

#define KFH_INVALID_NORMAL  (-2.0f)

/* Plugin parameters (names/defaults: icra2018_results/1080/violons_libkfusion-cpp_living_room_traj0_loop.log:20-31)
 * plus the sensor description the plugin reads in sb_init_slam_system. */
typedef struct kfh_config {
    int32_t input_width, input_height;  /* sensor frame size                                  */
    int32_t compute_size_ratio;         /* compute-size-ratio: 1, 2, 4 or 8                   */
    float   fx, fy, cx, cy;             /* intrinsics in INPUT pixels (normalised value * W/H) */
    float   depth_divisor;              /* raw depth units per metre (1000 for millimetres)   */
    int32_t vol_res[3];                 /* volume-resolution                                  */
    float   vol_dim[3];                 /* volume-size, metres                                */
    float   init_pos[3];                /* volume-direction: initial camera position          */
    float   mu;                         /* mu                                                 */
    float   icp_threshold;              /* icp-threshold                                      */
    int32_t pyramid[3];                 /* pyramid-level1..3, [0] = finest                    */
    int32_t integration_rate, tracking_rate;
    int32_t integrate_filtered;         /* 0: fuse raw metres depth, 1: bilateral-filtered    */
    int32_t slab_z0, slab_z1;           /* z range of the volume held by this device          */
} kfh_config;

typedef struct kfh_track_data {         /* TrackData of the kfusion use-case */
    int32_t result;
    float   error;
    float   J[6];
} kfh_track_data;

typedef struct kfh_frame_status {
    int32_t frame;        /* index of the frame the record belongs to (0-based) */
    int32_t tracked;      /* checkPose accepted the ICP result                  */
    int32_t integrated;   /* the TSDF was updated                               */
    int32_t raycasted;    /* the model maps were re-rendered                    */
    float   pose[16];     /* pose after the frame                               */
    float   sums[32];     /* last ICP reduction (layout of reduceKernel)        */
    int32_t icp_iterations; /* ICP iterations executed in this frame (all levels) */
    int32_t slow_solves;    /* of those, solves that needed the eigen-decomposition path */
    int32_t examined_solves; /* of those, solves whose conditioning was too close to the reference's 1e6 cut-off to
                               certify cheaply: the smallest singular value was computed (eigen-decompositions included) */
    int32_t reserved_;
} kfh_frame_status;

typedef struct kfh_context kfh_context;

/* buffer identifiers for kfh_download / kfh_device_ptr */
enum kfh_buffer {
    KFH_BUF_FLOAT_DEPTH = 0,   /* float[h][w]            raw metres                    */
    KFH_BUF_SCALED_DEPTH = 1,  /* float[h>>l][w>>l]      level l of the filtered pyramid */
    KFH_BUF_INPUT_VERTEX = 2,  /* float4[h>>l][w>>l]                                    */
    KFH_BUF_INPUT_NORMAL = 3,  /* float4[h>>l][w>>l]                                    */
    KFH_BUF_MODEL_VERTEX = 4,  /* float4[h][w]           the raycast's vertex map.  The raycast itself only writes the packed map
                                  (KFH_BUF_MODEL_TN); this map, KFH_BUF_MODEL_NORMAL and KFH_BUF_RAY_T are built from it
                                  inside kfh_device_ptr / kfh_download, for the raycast that was last when that call was
                                  made: a pointer kept across frames shows old maps; ask again after each raycast (a
                                  per-frame device consumer should read KFH_BUF_MODEL_TN) */
    KFH_BUF_MODEL_NORMAL = 5,  /* float4[h][w]           (see KFH_BUF_MODEL_VERTEX)     */
    KFH_BUF_VOLUME = 6,        /* short2[slab_z1-slab_z0][Y][X]                         */
    KFH_BUF_RAY_T = 7,         /* float[h][w]            hit distance of the last raycast (see KFH_BUF_MODEL_VERTEX) */
    KFH_BUF_BRICK_MARKS = 8,   /* uint8[Z/8][Y/8][X/8]   1: no voxel of the 8^3 brick (and one voxel past it per axis) has
                                  been below 26300/32766 since the last reset; the raycast skips such bricks with its
                                  coarse step.  Unsharded contexts only. */
    KFH_BUF_MODEL_TN = 9       /* float4[h][w]           the raycast output as tracking reads it: {hit distance t along the
                                  pixel's ray (0 = miss), normal}; the model vertex is origin + direction * t of the
                                  raycast pose.  After kfh_raycast_slab: this device's slab segment. */
};

/* image identifiers for kfh_render: the three pictures of the kfusion plugin's sb_update_outputs */
enum kfh_render_image {
    KFH_RENDER_DEPTH = 0,      /* renderDepthKernel: raw metres depth as a hue ramp between the near and far plane */
    KFH_RENDER_TRACK = 1,      /* renderTrackKernel: result code of the last trackKernel execution per pixel     */
    KFH_RENDER_VOLUME = 2      /* renderVolumeKernel: the TSDF raycast from a pose and shaded with a point light  */
};

const char *kfh_last_error(void);
int kfh_version(void);
int kfh_device_count(int *count);

/* ---- lifecycle (sb_init_slam_system / sb_clean_slam_system) ---- */
int kfh_create(const kfh_config *cfg, kfh_context **out);
int kfh_destroy(kfh_context *ctx);
int kfh_reset(kfh_context *ctx);                 /* empty volume, initial pose, frame 0 */

/* ---- frame input (sb_update_frame) ---- */
int kfh_upload_depth(kfh_context *ctx, const uint16_t *h_depth);      /* async H2D via pinned staging */
int kfh_set_depth_device(kfh_context *ctx, const uint16_t *d_depth);  /* use a frame already in HBM   */

/* ---- per-frame processing (sb_process_once); all asynchronous on the context's stream ---- */
int kfh_preprocess(kfh_context *ctx);            /* Kfusion::preprocessing */
int kfh_track(kfh_context *ctx);                 /* Kfusion::tracking      */
int kfh_integrate(kfh_context *ctx);             /* Kfusion::integration   */
int kfh_raycast(kfh_context *ctx);               /* Kfusion::raycasting    */
int kfh_end_frame(kfh_context *ctx);             /* frame counter ++       */
int kfh_process_frame(kfh_context *ctx);         /* the five calls above   */
int kfh_sync(kfh_context *ctx);

/* ---- results (sb_update_outputs) ---- */
int kfh_get_pose(kfh_context *ctx, float pose[16]);                    /* synchronises */
int kfh_set_pose(kfh_context *ctx, const float pose[16]);
int kfh_get_status(kfh_context *ctx, int frame, kfh_frame_status *out); /* synchronises; last 4096 frames kept */
int kfh_frame_index(kfh_context *ctx, int *frame);
/* Running totals of the integration's work list since kfh_create / kfh_reset: boxes of voxels_per_box voxels
 * (64 x 4 x a z extent chosen from the volume height) that the plan launches kept because they can hold a voxel
 * integrateKernel would update, and integrations that actually ran.  boxes * voxels_per_box * 8 bytes is what the
 * work launches had to read and write at most; a benchmark takes the difference of two calls.  Synchronises. */
int kfh_integrate_totals(kfh_context *ctx, uint64_t *boxes, uint64_t *launches, int32_t *voxels_per_box);
int kfh_compute_size(kfh_context *ctx, int *w, int *h);
int kfh_download(kfh_context *ctx, int buffer, int level, void *h_dst, size_t bytes);
int kfh_device_ptr(kfh_context *ctx, int buffer, int level, void **d_ptr, size_t *bytes);
int kfh_stream(kfh_context *ctx, void **stream);
/* HIP-event timing on the context's stream.  mode 0: off; 1: all four phases of every frame;
 * 2: the integration phase only (two events per frame); 3: as 2 but only on frames whose index is a
 * multiple of 16 (an event record occupies the stream for a few microseconds).  Records of the last 512 frames are kept;
 * ms = {preprocessing, tracking, integration, raycasting}; both getters synchronise. */
int kfh_set_timing(kfh_context *ctx, int mode);
int kfh_phase_ms(kfh_context *ctx, float ms[4]);                 /* most recent frame */
int kfh_phase_ms_frame(kfh_context *ctx, int frame, float ms[4]);

/* ---- GUI images (the FrameValue outputs of sb_update_outputs) ---- */
/* RGBA8 images of the compute size (kfh_compute_size), alpha 0.  view_pose is only used by
 * KFH_RENDER_VOLUME: camera-to-world pose to render from, NULL = the current pose (Kfusion::getViewPose);
 * the march uses the plugin's parameters (near plane 0.4, far plane 2 * 4.0, large step 0.75 * mu, light at
 * (1, 1, -1), ambient 0.1).  kfh_render copies to host memory and synchronises; kfh_render_device writes
 * w * h * 4 bytes of device memory asynchronously on the context's stream.  Not available on z-slab contexts
 * (KFH_RENDER_VOLUME). */
int kfh_render(kfh_context *ctx, int image, const float view_pose[16], uint8_t *h_rgba, size_t bytes);
int kfh_render_device(kfh_context *ctx, int image, const float view_pose[16], uint8_t *d_rgba);
/* KFH_RENDER_TRACK needs the per-pixel result codes, which tracking only stores (one byte per pixel and
 * executed iteration) after this call with enabled != 0; codes are zero ("unknown" colour) until tracking runs. */
int kfh_enable_track_render(kfh_context *ctx, int enabled);

/* ---- z-slab sharding across devices (one context per device); no counterpart in the reference ---- */
/* March only this context's slab; writes, per pixel, the hit distance t of the nearest zero crossing inside the
 * slab (0 = none) and its normal into the packed model map (KFH_BUF_MODEL_TN).  16 bytes per pixel are all that
 * has to be exchanged: the vertex is a function of the pixel and t. */
int kfh_raycast_slab(kfh_context *ctx);
/* Keep, per pixel, the candidate with the smallest positive t among n_parts gathered slab results.
 * d_tn_parts: float4[n_parts][h][w] in the layout of KFH_BUF_MODEL_TN; the winners become the context's model map. */
int kfh_composite_slabs(kfh_context *ctx, const void *d_tn_parts, int n_parts);
/* The same selection on a range of pixels only: d_tn_parts is float4[n_parts][pixel_count], the candidates of
 * pixels [first_pixel, first_pixel + pixel_count) (row-major pixel numbers); the winners go to d_out
 * (float4[pixel_count]), or into the context's own packed map at first_pixel when d_out is NULL.  For exchanges
 * that split the image among the devices instead of giving every device every slab: an all-to-all hands each
 * device the candidates of its share (1/N of the bytes of the all-gather above), this call selects, and an
 * all-gather of the selected shares into KFH_BUF_MODEL_TN completes the map on every device. */
int kfh_composite_tile(kfh_context *ctx, const void *d_tn_parts, int n_parts, int first_pixel, int pixel_count,
                       void *d_out);
/* Kfusion::tracking with the pixels of every pyramid level split by rows over share_count devices that all hold the
 * same model map and depth frame (what z-slab ranks do).  This context evaluates share `share`.  After every
 * iteration launch the library folds this device's partial sums into one row of 32 floats (reduceKernel layout) and
 * calls allreduce(user, d_sums32, stream): the callback must queue, in stream order on `stream`, an in-place SUM of
 * those 32 floats over all shares (ncclAllReduce on that stream, or wait + host exchange + copy back) and return 0.
 * It is called once per configured ICP iteration (19 with the default pyramid), from the calling thread.  Every
 * device then solves the same 6x6 system.  allreduce == NULL is only allowed with share_count == 1 and is
 * kfh_track; a single share with a callback still calls it (an identity then). */
typedef int (*kfh_allreduce_fn)(void *user, float *d_sums32, void *stream);
int kfh_track_rows(kfh_context *ctx, int share, int share_count, kfh_allreduce_fn allreduce, void *user);

/* ---- a group: z-slab sharding inside ONE process, driven by ONE thread; no counterpart in the reference ---- */
/* The harness this library plugs into is a single process that calls the plugin frame by frame
 * (framework/shared/src/SLAMBenchConfiguration.cpp:94-133 loads the libraries, :230-342 is the frame loop), so a
 * plugin that wants to spread a large volume over the GPUs of a node cannot rely on one process per GPU.  A group
 * holds one context per entry of `devices` (HIP device indices; an index may repeat, which puts several slabs on one
 * device), gives member i the z-slab i of n (contiguous, nearly equal; kfh_config.slab_z0 / slab_z1 of `cfg` are
 * ignored), and runs one sequence on all of them: every member preprocesses and tracks the whole frame (same inputs,
 * same poses), integrates and marches only its slab, and composites the nearest hit per pixel straight out of every
 * member's segment map - its own memory and its peers', read over the links after hipDeviceEnablePeerAccess; nothing
 * is copied and no collective library is involved.  The devices order themselves by events; the calling thread only
 * queues.  Results are those of kfh_raycast_slab + an all-gather + kfh_composite_slabs, bit for bit.
 * A group of one member is a plain context.  At most 16 members; calls on one group must not overlap.
 * Up to three members every member composites the whole image (it reads the other members' whole segment maps); from
 * four members on the image is split: a member selects the nearest hit for its 1/N of the pixels and stores the result
 * into every member's map (two maps' worth of bytes over the links per member, whatever N).  KFH_GROUP_EXCHANGE=gather /
 * share in the environment at creation decides otherwise.
 * When the members sit on more than one device, kfh_group_process_frame queues each member's launches from a host
 * thread of that member (created with the group, joined twice per frame); KFH_GROUP_THREADS=0 / 1 in the
 * environment at creation decides otherwise. */
typedef struct kfh_group kfh_group;
int kfh_group_create(const kfh_config *cfg, const int *devices, int n, kfh_group **out);
int kfh_group_destroy(kfh_group *group);
int kfh_group_reset(kfh_group *group);
int kfh_group_size(kfh_group *group, int *n);
/* member `index` as a context, for everything a group has no call of its own for (kfh_get_status, kfh_download,
 * kfh_set_timing, ...).  Frame-stepping calls on a member (kfh_process_frame, kfh_end_frame, ...) break the group. */
int kfh_group_member(kfh_group *group, int index, kfh_context **ctx);
int kfh_group_upload_depth(kfh_group *group, const uint16_t *h_depth);      /* the same host frame to every member */
/* d_depth_per_member[i]: the frame in the memory of member i's device */
int kfh_group_set_depth_device(kfh_group *group, const uint16_t *const *d_depth_per_member);
/* Kfusion's four stages + the exchange, queued on every member.  When it returns an error some members may have queued
 * more of the frame than others: the group must be reset (kfh_group_reset) or destroyed before it is used again. */
int kfh_group_process_frame(kfh_group *group);
int kfh_group_sync(kfh_group *group);
int kfh_group_get_pose(kfh_group *group, float pose[16]);                   /* waits for every member */

/* ---- stateless kernels on caller-owned device memory (used by the parity tests) ---- */
/* mm2metersKernel (+ bilateralFilterKernel when d_filtered != NULL) */
int kfh_k_preprocess(float *d_raw, float *d_filtered, int ow, int oh, const uint16_t *d_in, int iw, int ih,
                     float depth_divisor, void *stream);
/* bilateralFilterKernel on a float image (radius 2, delta 4.0, e_d 0.1) */
int kfh_k_bilateral(float *d_out, const float *d_in, int w, int h, void *stream);
/* halfSampleRobustImageKernel (r = 1) */
int kfh_k_half_sample(float *d_out, const float *d_in, int iw, int ih, float e_d, void *stream);
/* depth2vertexKernel + vertex2normalKernel, k = intrinsics of this level */
int kfh_k_vertex_normal(void *d_vertex4, void *d_normal4, const float *d_depth, int w, int h,
                        float fx, float fy, float cx, float cy, void *stream);
/* trackKernel + reduceKernel fused.  The input vertex map is not read: the kernel re-evaluates
 * depth2vertexKernel from d_in_depth (float[ih][iw]) and the level's intrinsics k = {fx, fy, cx, cy}.
 * d_sums receives 32 floats (reduceKernel layout); d_track (may be NULL) receives the per-pixel TrackData
 * with row stride `track_stride`. */
int kfh_k_track_reduce(float *d_sums, kfh_track_data *d_track, int track_stride,
                       const float *d_in_depth, const float k[4], const void *d_in_normal4, int iw, int ih,
                       const void *d_ref_vertex4, const void *d_ref_normal4, int rw, int rh,
                       const float Ttrack[16], const float view[16],
                       float dist_threshold, float normal_threshold, void *stream);
/* solve + updatePoseKernel on the device: pose <- exp(x) * pose; d_x6 receives the twist (6 doubles),
 * d_converged two ints: {|x| < icp_threshold, solve path: 0 plain solution (conditioning below the 1e6 limit),
 * 1 full eigen-decomposition, 2 only the smallest singular value dropped (found by power iteration)}. */
int kfh_k_update_pose(float *d_pose16, double *d_x6, int *d_converged, const float *d_sums,
                      float icp_threshold, void *stream);
/* integrateKernel over z in [z0, z1) of a volume whose storage starts at plane z0 */
int kfh_k_integrate(void *d_volume, const int32_t res[3], const float dim[3], int z0, int z1,
                    const float *d_depth, int w, int h, const float invTrack[16], const float K[16],
                    float mu, float maxweight, void *stream);
/* kfh_k_integrate for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments and the same voxels as
 * kfh_k_integrate, but the work list of the two launches lives in a fixed 4 MiB + 512 B array in device memory
 * (one copy per device, kept until the process ends) instead of a per-call hipMalloc, the pose travels as a kernel
 * argument instead of a host-to-device copy, and the call returns without waiting: everything is queued on
 * `stream`.  The array holds 524288 work items of 8 bytes; a volume whose work list needs more is refused with
 * KFH_ERR_INVALID_ARG (a 2048^3 volume with the default settings takes 1 MiB of the 4; the experiment settings
 * KFH_INTEGRATE_GROUP and KFH_INTEGRATE_ZCHUNK change what a volume needs: a group of 1 at 2048^3 needs 8 MiB).  All calls on one device share that device's array: calls
 * that overlap on one device (different streams or threads) race; calls on one stream are safe back to back.
 * The first call on a device looks the array's address up (hipGetSymbolAddress); make it outside a capture. */
int kfh_k_integrate_noalloc(void *d_volume, const int32_t res[3], const float dim[3], int z0, int z1,
                            const float *d_depth, int w, int h, const float invTrack[16], const float K[16],
                            float mu, float maxweight, void *stream);
/* raycastKernel */
int kfh_k_raycast(void *d_vertex4, void *d_normal4, int w, int h, const void *d_volume,
                  const int32_t res[3], const float dim[3], const float view[16],
                  float near_plane, float far_plane, float step, float largestep, void *stream);

/* renderDepthKernel */
int kfh_k_render_depth(uint8_t *d_rgba, const float *d_depth, int w, int h, float near_plane, float far_plane,
                       void *stream);
/* renderTrackKernel on TrackData as kfh_k_track_reduce writes it */
int kfh_k_render_track(uint8_t *d_rgba, const kfh_track_data *d_track, int track_stride, int w, int h, void *stream);
/* renderVolumeKernel: raycast from `view` (pose * inverse camera matrix) + shading */
int kfh_k_render_volume(uint8_t *d_rgba, int w, int h, const void *d_volume, const int32_t res[3], const float dim[3],
                        const float view[16], float near_plane, float far_plane, float step, float largestep,
                        const float light[3], const float ambient[3], void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KFUSION_HIP_H_ */
