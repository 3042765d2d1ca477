// C ABI of libkfusion_hip_core.so: pipeline context (the Kfusion class of the kfusion use-case re-cast as
// a single HIP stream of kernels with a device-resident control block) and the stateless kernel entry points.
#include "kfh_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <condition_variable>
#include <functional>
#include <thread>
#include <vector>

using namespace kfh;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define KFH_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(KFH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                   \
    } while (0)

#define KFH_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return fail(KFH_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

constexpr float DIST_THRESHOLD = 0.1f;     // Kfusion dist_threshold
constexpr float NORMAL_THRESHOLD = 0.8f;   // Kfusion normal_threshold
constexpr float TRACK_THRESHOLD = 0.15f;   // Kfusion track_threshold
constexpr float MAX_WEIGHT = 100.0f;       // Kfusion maxweight
constexpr float NEAR_PLANE = 0.4f, FAR_PLANE = 4.0f;
constexpr float E_DELTA = 0.1f;

}  // namespace

struct kfh_context {
    kfh_config cfg;
    int device = 0;
    int w = 0, h = 0;          // compute size
    float k[4] = {0, 0, 0, 0}; // intrinsics at compute size
    int frame = 0;
    int zs0 = 0, zs1 = 0;      // stored planes (slab + halo)
    bool sharded = false;
    int compute_units = 256;
    hipStream_t stream = nullptr;

    // Two staging slots (pinned host + device) so that the host fills one while the device still reads the other
    uint16_t *h_depth_pinned[2] = {nullptr, nullptr};
    uint16_t *d_depth_in[2] = {nullptr, nullptr};
    hipEvent_t ev_slot_free[2] = {nullptr, nullptr};   // recorded after the preprocessing launch that consumed the slot
    bool slot_unread[2] = {false, false};  // uploaded, but no preprocessing launch has been queued on it yet
    int upload_slot = 0;
    const uint16_t *d_depth_cur = nullptr; // frame to process (staging or caller-owned)
    // Depth images of the frame being processed: aliases of one of two buffer sets.  With `overlap` the
    // preprocessing of frame n + 1 runs on its own stream and writes the other set while frame n is still
    // tracking / integrating.
    float *d_float_depth = nullptr;
    float *d_scaled[3] = {nullptr, nullptr, nullptr};
    float *set_float_depth[2] = {nullptr, nullptr};
    float *set_scaled[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    float *d_tile_max = nullptr;           // largest depth per preprocessing tile of the current set
    float *set_tile_max[2] = {nullptr, nullptr};
    int next_set = 0;
    hipStream_t pre_stream = nullptr;
    hipEvent_t ev_pre_done = nullptr;      // recorded on pre_stream after each preprocessing launch
    hipEvent_t ev_set_free[2] = {nullptr, nullptr};   // recorded on the main stream when a frame is done with its set
    int cur_set = 0;
    bool pre_pending = false;              // the main stream has not yet waited for ev_pre_done
    bool overlap = false;
    bool tail_pending = false;             // the second pass of the last raycast has not been launched yet
    // The tracking finish kernel is what empties the two device work lists (integrate boxes, unfinished rays) and
    // points the raycast matrices at the pose it accepted.  A staged call made without a kfh_track in between
    // (a second kfh_integrate, kfh_set_pose + kfh_raycast to re-render) finds these false and does both itself.
    // The unsharded raycast only writes the packed map tracking reads; the float4 vertex / normal maps and the hit
    // distances are built from it when a caller asks for them (model_maps_stale: not built since the last raycast).
    bool model_maps_stale = false;
    bool box_list_fresh = false;
    bool ray_list_fresh = false;
    float4 *d_in_vertex[3] = {nullptr, nullptr, nullptr};
    float4 *d_in_normal[3] = {nullptr, nullptr, nullptr};
    float4 *d_vertex = nullptr, *d_normal = nullptr;
    unsigned char *d_bricks = nullptr;     // brick marks for the raycast (the whole volume's grid, also for a slab)
    void *d_integrate_boxes = nullptr;     // work items the integrate plan launch made for the work launch
    void *d_ray_tasks = nullptr;           // rays handed from the first raycast pass to the second
    float4 *d_ref_tn = nullptr;            // packed model map {hit distance, normal}: what tracking gathers
    float *d_ray_t = nullptr;
    int depth_version = 0;                 // bumped whenever the depth pyramid changes
    int in_maps_version[3] = {-1, -1, -1}; // pyramid version the input vertex / normal maps were built from
    short2 *d_volume = nullptr;
    size_t volume_voxels = 0;
    float *d_reduced = nullptr;            // row-sharded tracking: two alternating rows of 32 sums (this device's share,
                                           //   then all devices' after the caller's all-reduce); allocated on first use
    float *d_partials = nullptr;           // two alternating buffers of per-block partial rows
    size_t partials_stride = 0;            // floats per buffer
    Ctrl *d_ctrl = nullptr;
    kfh_frame_status *d_status = nullptr;
    // Pose record in host memory the finish launch writes for callers that ask for the pose after every frame
    // (see PoseMail).  Written only while such a caller is there: the frame after a kfh_get_pose gets one.
    PoseMail *h_mail = nullptr;
    unsigned *d_mail = nullptr;            // the same record as the device addresses it
    unsigned mail_seq = 0;                 // number of the last finish launch that was given the record
    bool mail_current = false;             // that launch is the last finish launch queued, and no pose was set since
    bool pose_read_this_frame = false, pose_read_last_frame = false;
    unsigned long long *d_stamps = nullptr; // diagnostic ICP time stamps (KFH_ICP_STAMPS=1), [launch][block][slot]
    // GUI images: allocated when first asked for
    signed char *d_track_codes = nullptr;  // trackKernel result code per pixel of the compute size
    float4 *d_render_vertex = nullptr, *d_render_normal = nullptr;   // maps of the volume render's own raycast
    uchar4 *d_render_rgba = nullptr;
    char *d_render_state = nullptr;        // [view matrix | unfinished-ray counter at byte 256]

    // phase timing: a ring of event sets, one per frame (slot = frame % KFH_TIMING_RING)
    int timing = 0;            // 0 off, 1 all four phases, 2 integration only
    hipEvent_t (*ev)[5] = nullptr;
    int timed_first = -1;      // first frame recorded since timing was (re)enabled
};

constexpr int KFH_TIMING_RING = 512;

namespace {

// Entry points may be called from any thread and with any device current: the context's device is made
// current for the duration of the call (its streams, events and buffers belong to it) and the caller's put back.
struct DeviceScope {
    int previous = -1;
    bool switched = false;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&previous) == hipSuccess && previous != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(previous);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

int validate_config(const kfh_config *c) {
    KFH_REQUIRE(c != nullptr, "config is NULL");
    KFH_REQUIRE(c->input_width > 0 && c->input_height > 0, "input size must be positive");
    const int r = c->compute_size_ratio;
    KFH_REQUIRE(r == 1 || r == 2 || r == 4 || r == 8, "compute_size_ratio must be 1, 2, 4 or 8 (got %d)", r);
    KFH_REQUIRE(c->input_width % (4 * r) == 0 && c->input_height % (4 * r) == 0,
                "input size %dx%d must be divisible by 4*compute_size_ratio", c->input_width, c->input_height);
    KFH_REQUIRE(c->fx != 0.f && c->fy != 0.f, "focal lengths must be non-zero");
    KFH_REQUIRE(c->depth_divisor > 0.f, "depth_divisor must be positive");
    for (int i = 0; i < 3; ++i) {
        // two voxels per axis at least: a trilinear sample reads pairs of x-adjacent voxels
        KFH_REQUIRE(c->vol_res[i] >= 2 && c->vol_res[i] <= 4096, "vol_res[%d]=%d out of range [2, 4096]", i, c->vol_res[i]);
        KFH_REQUIRE(c->vol_dim[i] > 0.f, "vol_dim[%d] must be positive", i);
        KFH_REQUIRE(c->pyramid[i] >= 0 && c->pyramid[i] <= 64, "pyramid[%d]=%d out of range", i, c->pyramid[i]);
    }
    KFH_REQUIRE(c->mu > 0.f, "mu must be positive");
    KFH_REQUIRE(c->integration_rate >= 1 && c->tracking_rate >= 1, "rates must be >= 1");
    KFH_REQUIRE(c->slab_z0 >= 0 && c->slab_z0 < c->slab_z1 && c->slab_z1 <= c->vol_res[2],
                "slab [%d,%d) must lie inside [0,%d)", c->slab_z0, c->slab_z1, c->vol_res[2]);
    return KFH_OK;
}

template <typename T>
hipError_t dev_alloc(T **p, size_t count) {
    return hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
}

void free_all(kfh_context *c) {
    for (int i = 0; i < 2; ++i) {
        if (c->h_depth_pinned[i]) (void)hipHostFree(c->h_depth_pinned[i]);
        if (c->ev_slot_free[i]) (void)hipEventDestroy(c->ev_slot_free[i]);
    }
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    void *bufs[] = {c->set_tile_max[0], c->set_tile_max[1], c->d_depth_in[0], c->d_depth_in[1], c->set_float_depth[0], c->set_float_depth[1], c->set_scaled[0][0],
                    c->set_scaled[0][1], c->set_scaled[0][2], c->set_scaled[1][0], c->set_scaled[1][1],
                    c->set_scaled[1][2],
                    c->d_in_vertex[0], c->d_in_vertex[1], c->d_in_vertex[2], c->d_in_normal[0],
                    c->d_in_normal[1], c->d_in_normal[2], c->d_vertex, c->d_normal, c->d_ray_t, c->d_volume, c->d_ref_tn, c->d_ray_tasks, c->d_integrate_boxes, c->d_bricks,
                    c->d_partials, c->d_reduced, c->d_ctrl, c->d_status, c->d_stamps, c->d_track_codes, c->d_render_vertex,
                    c->d_render_normal, c->d_render_rgba, c->d_render_state};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->ev) {
        for (int i = 0; i < KFH_TIMING_RING; ++i)
            for (int j = 0; j < 5; ++j)
                if (c->ev[i][j]) (void)hipEventDestroy(c->ev[i][j]);
        delete[] c->ev;
    }
    if (c->ev_pre_done) (void)hipEventDestroy(c->ev_pre_done);
    for (int i = 0; i < 2; ++i)
        if (c->ev_set_free[i]) (void)hipEventDestroy(c->ev_set_free[i]);
    if (c->pre_stream) (void)hipStreamDestroy(c->pre_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

// Event records cost a few microseconds of stream time each; mode 3 only takes them on every sixteenth frame.
bool timed_frame(const kfh_context *c) { return c->timing != 0 && (c->timing != 3 || c->frame % 16 == 0); }

// The main stream must not touch the depth images before the preprocessing launch that fills them is done.
hipError_t ensure_pre_done(kfh_context *c) {
    if (!c->pre_pending) return hipSuccess;
    c->pre_pending = false;
    return hipStreamWaitEvent(c->stream, c->ev_pre_done, 0);
}

void select_set(kfh_context *c, int set) {
    c->cur_set = set;
    c->d_float_depth = c->set_float_depth[set];
    c->d_tile_max = c->set_tile_max[set];
    for (int l = 0; l < 3; ++l) c->d_scaled[l] = c->set_scaled[set][l];
}

int init_state(kfh_context *c) {
    const size_t npix = (size_t)c->w * c->h;
    KFH_HIP(hipStreamSynchronize(c->pre_stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    c->pre_pending = false;
    c->tail_pending = false;       // the maps it would complete are about to be cleared
    c->model_maps_stale = false;
    c->next_set = 0;
    select_set(c, 0);
    KFH_HIP(launch_fill_volume(c->d_volume, c->volume_voxels, c->stream));
    if (c->d_bricks) {
        // A mark vouches for the voxels of its brick and for the plane above it (the upper taps of a sample based in
        // the brick).  A slab context only sees its own planes and two more per side: bricks that reach outside
        // them never get a mark (the raycast then fetches their voxels); the volume's own faces need no neighbour.
        const int Z = c->cfg.vol_res[2], bgz = (Z + 7) / 8;
        const size_t layer = (size_t)((c->cfg.vol_res[0] + 7) / 8) * ((c->cfg.vol_res[1] + 7) / 8);
        const int b_lo = c->zs0 == 0 ? 0 : (c->zs0 + 7) / 8;
        const int b_hi = c->zs1 == Z ? bgz : (c->zs1 >= 9 ? (c->zs1 - 9) / 8 + 1 : 0);
        KFH_HIP(hipMemsetAsync(c->d_bricks, 0, brick_mark_bytes(c->cfg.vol_res), c->stream));
        if (b_hi > b_lo) KFH_HIP(hipMemsetAsync(c->d_bricks + (size_t)b_lo * layer, 1, (size_t)(b_hi - b_lo) * layer, c->stream));
    }
    KFH_HIP(hipMemsetAsync(c->d_vertex, 0, npix * sizeof(float4), c->stream));
    KFH_HIP(hipMemsetAsync(c->d_normal, 0, npix * sizeof(float4), c->stream));
    KFH_HIP(hipMemsetAsync(c->d_ray_t, 0, npix * sizeof(float), c->stream));
    KFH_HIP(hipMemsetAsync(c->d_ref_tn, 0, npix * sizeof(float4), c->stream));
    for (int l = 0; l < 3; ++l) c->in_maps_version[l] = -1;
    if (c->d_track_codes) KFH_HIP(hipMemsetAsync(c->d_track_codes, 0, npix, c->stream));
    KFH_HIP(hipMemsetAsync(c->d_status, 0, sizeof(kfh_frame_status) * KFH_STATUS_RING, c->stream));
    Ctrl init;
    std::memset(&init, 0, sizeof(init));
    const float p[16] = {1, 0, 0, c->cfg.init_pos[0], 0, 1, 0, c->cfg.init_pos[1],
                         0, 0, 1, c->cfg.init_pos[2], 0, 0, 0, 1};
    std::memcpy(init.pose.m, p, sizeof(p));
    KFH_HIP(hipMemcpyAsync(c->d_ctrl, &init, sizeof(Ctrl), hipMemcpyHostToDevice, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));   // `init` lives on this stack frame
    KFH_HIP(launch_pose_derive(c->d_ctrl, c->k, 1, c->stream));
    c->box_list_fresh = c->ray_list_fresh = true;
    c->mail_current = false;
    c->pose_read_this_frame = c->pose_read_last_frame = false;
    c->frame = 0;
    c->d_depth_cur = c->d_depth_in[0];
    c->upload_slot = 0;
    c->slot_unread[0] = c->slot_unread[1] = false;
    c->timed_first = -1;
    return KFH_OK;
}

FinishArgs finish_args(const kfh_context *c, int seq, int pending_rows) {
    FinishArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ctrl = c->d_ctrl;
    a.partials_in = c->d_partials + (size_t)(seq & 1) * c->partials_stride;
    a.pending_rows = pending_rows;
    a.seq = seq;
    a.w = c->w;
    a.h = c->h;
    a.icp_threshold = c->cfg.icp_threshold;
    a.track_threshold = TRACK_THRESHOLD;
    for (int i = 0; i < 4; ++i) a.k[i] = c->k[i];
    a.will_raycast = c->frame > 2 ? 1 : 0;
    a.frame = c->frame;
    a.integration_rate = c->cfg.integration_rate;
    a.status_ring = c->d_status;
    return a;
}

// The finish launch of this frame writes the host-visible pose record when the caller read the pose during the
// previous frame (one that does so every frame is served from the record from its second frame on).
void attach_mail(kfh_context *c, FinishArgs *f) {
    c->mail_current = false;
    if (c->d_mail && c->pose_read_last_frame) {
        f->mail = c->d_mail;
        f->mail_seq = ++c->mail_seq;
        c->mail_current = true;
    }
}

}  // namespace

extern "C" {

const char *kfh_last_error(void) { return g_last_error.c_str(); }
int kfh_version(void) { return 100; }

int kfh_device_count(int *count) {
    KFH_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return KFH_OK;
}

int kfh_create(const kfh_config *cfg, kfh_context **out) {
    KFH_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    if (int rc = validate_config(cfg)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(KFH_ERR_NO_DEVICE, "no HIP device is visible to this process");
    kfh_context *c = new (std::nothrow) kfh_context();
    if (!c) return fail(KFH_ERR_STATE, "out of host memory");
    c->cfg = *cfg;
    const int r = cfg->compute_size_ratio;
    c->w = cfg->input_width / r;
    c->h = cfg->input_height / r;
    c->k[0] = cfg->fx / (float)r; c->k[1] = cfg->fy / (float)r;
    c->k[2] = cfg->cx / (float)r; c->k[3] = cfg->cy / (float)r;
    c->sharded = !(cfg->slab_z0 == 0 && cfg->slab_z1 == cfg->vol_res[2]);
    c->zs0 = c->sharded ? (cfg->slab_z0 - 2 > 0 ? cfg->slab_z0 - 2 : 0) : 0;
    c->zs1 = c->sharded ? (cfg->slab_z1 + 2 < cfg->vol_res[2] ? cfg->slab_z1 + 2 : cfg->vol_res[2]) : cfg->vol_res[2];
    c->volume_voxels = (size_t)cfg->vol_res[0] * cfg->vol_res[1] * (size_t)(c->zs1 - c->zs0);

    const size_t npix = (size_t)c->w * c->h;
    const size_t nin = (size_t)cfg->input_width * cfg->input_height;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc(reinterpret_cast<void **>(&c->h_depth_pinned[i]), nin * sizeof(uint16_t), hipHostMallocDefault);
        if (e == hipSuccess) e = dev_alloc(&c->d_depth_in[i], nin);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_slot_free[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pre_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pre_done, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ev_set_free[i], hipEventDisableTiming);
        if (e == hipSuccess) e = dev_alloc(&c->set_float_depth[i], npix);
        if (e == hipSuccess) e = dev_alloc(&c->set_tile_max[i], depth_tile_count(c->w, c->h));
    }
    // Opt-in: on MI355X the cross-stream event waits this needs cost more per frame (about 20 us measured)
    // than the 13 us preprocessing launch they hide, so a single stream is the default.
    c->overlap = std::getenv("KFH_OVERLAP_PREPROCESS") != nullptr;
    for (int l = 0; l < 3 && e == hipSuccess; ++l) {
        const size_t nl = (size_t)(c->w >> l) * (c->h >> l);
        e = dev_alloc(&c->set_scaled[0][l], nl);
        if (e == hipSuccess) e = dev_alloc(&c->set_scaled[1][l], nl);
        if (e == hipSuccess) e = dev_alloc(&c->d_in_vertex[l], nl);
        if (e == hipSuccess) e = dev_alloc(&c->d_in_normal[l], nl);
    }
    if (e == hipSuccess) e = dev_alloc(&c->d_vertex, npix);
    if (e == hipSuccess) e = dev_alloc(&c->d_normal, npix);
    if (e == hipSuccess) e = dev_alloc(&c->d_ray_t, npix);
    if (e == hipSuccess) e = dev_alloc(&c->d_ref_tn, npix);
    if (e == hipSuccess) e = hipMalloc(&c->d_ray_tasks, raycast_task_bytes(c->w, c->h));
    if (e == hipSuccess)
        e = hipMalloc(&c->d_integrate_boxes, integrate_box_list_bytes(cfg->vol_res, c->zs1));
    if (e == hipSuccess && std::getenv("KFH_NO_BRICK_MARKS") == nullptr)
        e = hipMalloc(reinterpret_cast<void **>(&c->d_bricks), brick_mark_bytes(cfg->vol_res));
    if (e == hipSuccess) e = dev_alloc(&c->d_volume, c->volume_voxels);
    {
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, c->device);
        c->compute_units = e == hipSuccess ? prop.multiProcessorCount : 256;
        int rows = 1;
        for (int l = 0; l < 3; ++l) {
            const int wl = c->w >> l, hl = c->h >> l;
            int pw, ppt;
            track_shape(wl, hl, c->compute_units, &pw, &ppt);
            const int nb = track_blocks(wl, hl, pw, ppt);
            if (nb > rows) rows = nb;
        }
        c->partials_stride = (size_t)rows * 32;
    }
    if (e == hipSuccess) e = dev_alloc(&c->d_partials, 2 * c->partials_stride);
    if (e == hipSuccess) e = dev_alloc(&c->d_ctrl, 1);
    if (e == hipSuccess) e = dev_alloc(&c->d_status, KFH_STATUS_RING);
    if (e == hipSuccess && !std::getenv("KFH_NO_POSE_MAIL")) {
        // optional: without it kfh_get_pose copies the pose from device memory as before
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, sizeof(PoseMail), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                std::memset(h, 0, sizeof(PoseMail));
                c->h_mail = static_cast<PoseMail *>(h);
                c->d_mail = static_cast<unsigned *>(d);
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    if (e == hipSuccess && std::getenv("KFH_ICP_STAMPS")) {
        const size_t n = (size_t)KFH_STAMP_LAUNCHES * KFH_STAMP_BLOCKS * KFH_STAMP_SLOTS;
        e = dev_alloc(&c->d_stamps, n);
        if (e == hipSuccess) e = hipMemset(c->d_stamps, 0, n * sizeof(unsigned long long));
    }
    if (e != hipSuccess) {
        free_all(c);
        delete c;
        return fail(KFH_ERR_HIP, "allocation failed: %s", hipGetErrorString(e));
    }
    if (int rc = init_state(c)) {
        free_all(c);
        delete c;
        return rc;
    }
    *out = c;
    return KFH_OK;
}

int kfh_destroy(kfh_context *c) {
    if (!c) return KFH_OK;
    DeviceScope on_device(c->device);
    if (c->pre_stream) (void)hipStreamSynchronize(c->pre_stream);
    (void)hipStreamSynchronize(c->stream);
    free_all(c);
    delete c;
    return KFH_OK;
}

int kfh_reset(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    return init_state(c);
}

static int flush_tail(kfh_context *c);                                                    // below, with kfh_raycast
static hipError_t raycast_part(kfh_context *c, int part, const PreprocessLaunch *pre);

int kfh_upload_depth(kfh_context *c, const uint16_t *h_depth) {
    KFH_REQUIRE(c != nullptr && h_depth != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    const size_t bytes = (size_t)c->cfg.input_width * c->cfg.input_height * sizeof(uint16_t);
    // The copy goes on the stream that preprocesses.  The slot's pinned buffer fed the copy of the frame before
    // the previous one and its device buffer that frame's preprocessing launch: wait for that launch only, so the
    // host fills this slot while the device is still busy with the previous frame.
    hipStream_t ps = (c->overlap && c->timing != 1) ? c->pre_stream : c->stream;   // the stream kfh_preprocess will use
    const int slot = c->upload_slot;
    c->upload_slot ^= 1;
    if (c->slot_unread[slot]) KFH_HIP(hipStreamSynchronize(ps));   // two uploads ago nobody consumed this slot
    else KFH_HIP(hipEventSynchronize(c->ev_slot_free[slot]));
    c->slot_unread[slot] = true;
    std::memcpy(c->h_depth_pinned[slot], h_depth, bytes);
    static const bool use_copy_engine = std::getenv("KFH_UPLOAD_COPY_ENGINE") != nullptr;
    if (use_copy_engine || bytes % 16 != 0)
        KFH_HIP(hipMemcpyAsync(c->d_depth_in[slot], c->h_depth_pinned[slot], bytes, hipMemcpyHostToDevice, ps));
    else
        KFH_HIP(launch_stage_copy(c->d_depth_in[slot], c->h_depth_pinned[slot], bytes, ps));
    c->d_depth_cur = c->d_depth_in[slot];
    return KFH_OK;
}

int kfh_set_depth_device(kfh_context *c, const uint16_t *d_depth) {
    KFH_REQUIRE(c != nullptr && d_depth != nullptr, "NULL argument");
    c->d_depth_cur = d_depth;
    return KFH_OK;
}

int kfh_preprocess(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    // With phase timing on, everything runs on the main stream so that the four phases do not overlap.
    const bool side = c->overlap && c->timing != 1;
    hipStream_t ps = side ? c->pre_stream : c->stream;
    const int set = c->next_set;
    c->next_set ^= 1;
    KFH_HIP(ensure_pre_done(c));
    select_set(c, set);
    // frame n - 2 used this set: its tracking and integration must be done before it is overwritten
    if (side) KFH_HIP(hipStreamWaitEvent(ps, c->ev_set_free[set], 0));
    if (c->timing == 1) KFH_HIP(hipEventRecord(c->ev[c->frame % KFH_TIMING_RING][0], ps));
    // the depth pyramid of tracking (halfSampleRobustImageKernel twice) is produced by the same launch
    if (c->tail_pending && !side) {
        // together with the second pass of the previous frame's raycast
        PreprocessLaunch pre;
        pre.raw = c->d_float_depth; pre.filtered = c->d_scaled[0]; pre.ow = c->w; pre.oh = c->h; pre.in = c->d_depth_cur;
        pre.iw = c->cfg.input_width; pre.ratio = c->cfg.compute_size_ratio; pre.divisor = c->cfg.depth_divisor;
        pre.level1 = c->d_scaled[1]; pre.level2 = c->d_scaled[2]; pre.pyramid_e_d = E_DELTA * 3; pre.tile_max = c->d_tile_max;
        c->tail_pending = false;
        KFH_HIP(raycast_part(c, RAYCAST_TAIL, &pre));
    } else {
        if (int rc = flush_tail(c)) return rc;
        KFH_HIP(launch_preprocess(c->d_float_depth, c->d_scaled[0], c->w, c->h, c->d_depth_cur, true,
                                  c->cfg.input_width, c->cfg.compute_size_ratio, c->cfg.depth_divisor, c->d_scaled[1],
                                  c->d_scaled[2], E_DELTA * 3, c->d_tile_max, ps));
    }
    c->depth_version++;
    for (int i = 0; i < 2; ++i)      // a staging slot is free again once this launch has read it
        if (c->d_depth_cur == c->d_depth_in[i]) {
            KFH_HIP(hipEventRecord(c->ev_slot_free[i], ps));
            c->slot_unread[i] = false;
        }
    if (c->timing == 1) KFH_HIP(hipEventRecord(c->ev[c->frame % KFH_TIMING_RING][1], ps));
    if (side) {
        KFH_HIP(hipEventRecord(c->ev_pre_done, ps));
        c->pre_pending = true;
    }
    return KFH_OK;
}

// Kfusion::tracking.  share_count == 1: the whole image on this device, one launch per iteration.  Otherwise this
// device evaluates tile rows [share * T / share_count, (share + 1) * T / share_count) of every level's T tile rows;
// after every iteration launch its partial rows are folded into one row of 32 sums and `allreduce` must queue, in
// stream order, an in-place sum of that row over all devices, which the next launch (or the finish) then solves.
static int track_impl(kfh_context *c, int share, int share_count, kfh_allreduce_fn allreduce, void *user) {
    if (int rc = flush_tail(c)) return rc;
    KFH_HIP(ensure_pre_done(c));
    int total_iterations = 0;
    for (int l = 0; l < 3; ++l) total_iterations += c->cfg.pyramid[l];
    if (c->frame % c->cfg.tracking_rate != 0 || total_iterations == 0) {
        FinishArgs f = finish_args(c, 0, 0);
        attach_mail(c, &f);
        KFH_HIP(launch_track_skip(f, c->stream));
    } else {
        // depth2vertex / vertex2normal are evaluated inside the iteration kernel from the depth pyramid;
        // the vertex / normal maps themselves are only materialised when a caller asks for them
        IcpIterArgs a;
        std::memset(&a, 0, sizeof(a));
        a.ctrl = c->d_ctrl;
        a.ref_tn = c->d_ref_tn;
        a.rw = c->w;
        a.rh = c->h;
        a.dist_threshold = DIST_THRESHOLD;
        a.normal_threshold = NORMAL_THRESHOLD;
        a.icp_threshold = c->cfg.icp_threshold;
        a.stamps = c->d_stamps;
        a.result_out = c->d_track_codes;
        a.result_stride = c->w;
        a.max_blocks = c->compute_units;
        const bool sharded = allreduce != nullptr;      // a single share with a callback still takes the exchange path
        a.rows_reduced = sharded ? 1 : 0;
        if (sharded) {
            // every level must give every share at least one row of tiles: checked before anything is queued, so that
            // a refusal leaves the context as it was
            for (int level = 2; level >= 0; --level) {
                int pw = 0, ppt = 0;
                const int iw = c->w >> level, ih = c->h >> level;
                track_shape(iw, ih, c->compute_units, &pw, &ppt);
                const int tile_rows = track_blocks(iw, ih, pw, ppt) / ((iw + KFH_TRACK_TILE_W - 1) / KFH_TRACK_TILE_W);
                if (tile_rows < share_count)
                    return fail(KFH_ERR_INVALID_ARG, "pyramid level %d has %d tile rows, fewer than the %d shares", level,
                                tile_rows, share_count);
            }
        }
        int seq = 0, rows = 0;
        const float *pending = nullptr;      // sharded: the all-reduced row the next launch solves
        for (int level = 2; level >= 0; --level) {
            a.level = level;
            a.in_depth = c->d_scaled[level];
            for (int i = 0; i < 4; ++i) a.k[i] = c->k[i] / (float)(1 << level);
            a.iw = c->w >> level;
            a.ih = c->h >> level;
            track_shape(a.iw, a.ih, c->compute_units, &a.pw, &a.ppt);
            const int tiles_x = (a.iw + KFH_TRACK_TILE_W - 1) / KFH_TRACK_TILE_W;
            const int tile_rows = track_blocks(a.iw, a.ih, a.pw, a.ppt) / tiles_x;
            if (sharded) {
                const int r0 = (int)((long long)share * tile_rows / share_count);
                const int r1 = (int)((long long)(share + 1) * tile_rows / share_count);
                a.block0 = r0 * tiles_x;
                a.block_count = (r1 - r0) * tiles_x;
            }
            for (int i = 0; i < c->cfg.pyramid[level]; ++i, ++seq) {
                a.seq = seq;
                a.last_iter = (i == c->cfg.pyramid[level] - 1) ? 1 : 0;
                a.partials_in = sharded ? (pending ? pending : c->d_reduced) : c->d_partials + (size_t)(seq & 1) * c->partials_stride;
                a.partials_out = c->d_partials + (size_t)((seq + 1) & 1) * c->partials_stride;
                a.pending_rows = rows;
                KFH_HIP(launch_icp_iteration(a, c->stream));
                if (sharded) {
                    float *row = c->d_reduced + (size_t)((seq + 1) & 1) * 64;
                    KFH_HIP(launch_sum_partials(row, a.partials_out, a.block_count, c->stream));
                    if (int rc = allreduce(user, row, c->stream)) {
                        // The frame cannot be tracked any further.  The iteration launches queued so far have left a
                        // pending row in the ICP record; the skip launch ends the frame as "not tracked" (old pose
                        // kept, record cleared, work lists emptied), so the context stays usable.
                        FinishArgs skip = finish_args(c, 0, 0);
                        attach_mail(c, &skip);
                        (void)launch_track_skip(skip, c->stream);
                        c->box_list_fresh = c->ray_list_fresh = true;
                        return fail(KFH_ERR_STATE, "the all-reduce callback of kfh_track_rows failed with code %d; the "
                                    "frame was ended as not tracked", rc);
                    }
                    pending = row;
                    rows = 1;
                } else {
                    rows = track_blocks(a.iw, a.ih, a.pw, a.ppt);
                }
            }
        }
        FinishArgs f = finish_args(c, seq, rows);
        if (sharded) f.partials_in = pending;
        attach_mail(c, &f);
        KFH_HIP(launch_icp_finish(f, c->stream));
    }
    c->box_list_fresh = c->ray_list_fresh = true;      // both finish kernels empty the lists and set the raycast matrices
    if (timed_frame(c)) KFH_HIP(hipEventRecord(c->ev[c->frame % KFH_TIMING_RING][2], c->stream));
    return KFH_OK;
}

int kfh_track(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    return track_impl(c, 0, 1, nullptr, nullptr);
}

int kfh_track_rows(kfh_context *c, int share, int share_count, kfh_allreduce_fn allreduce, void *user) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    KFH_REQUIRE(share_count >= 1 && share >= 0 && share < share_count, "share %d of %d is out of range", share, share_count);
    KFH_REQUIRE(share_count == 1 || allreduce != nullptr, "more than one share needs an all-reduce callback");
    DeviceScope on_device(c->device);
    if (allreduce != nullptr && !c->d_reduced) {
        KFH_HIP(dev_alloc(&c->d_reduced, (size_t)128));
        KFH_HIP(hipMemsetAsync(c->d_reduced, 0, 128 * sizeof(float), c->stream));
    }
    return track_impl(c, share, share_count, allreduce, user);
}

// The per-tile depth maxima let the integration drop boxes behind the seen surface (KFH_NO_DEPTH_CULL: off).
static const float *tile_max_for_integrate(const kfh_context *c) {
    static const bool off = std::getenv("KFH_NO_DEPTH_CULL") != nullptr;
    return off ? nullptr : c->d_tile_max;
}

int kfh_integrate(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    KFH_HIP(ensure_pre_done(c));
    if ((c->frame % c->cfg.integration_rate) == 0 || c->frame <= 3) {
        const float *src = c->cfg.integrate_filtered ? c->d_scaled[0] : c->d_float_depth;
        if (!c->box_list_fresh)      // no kfh_track since the last integration: the list still holds that one's boxes
            KFH_HIP(hipMemsetAsync(c->d_ctrl->integrate_box_count, 0, 2 * sizeof(unsigned), c->stream));
        c->box_list_fresh = false;
        // planes [zs0, zs1) are stored; the kernel replays the z < zs0 steps of each column arithmetically
        KFH_HIP(launch_integrate(c->d_volume, c->cfg.vol_res, c->cfg.vol_dim, c->zs0, c->zs1, src, c->w, c->h,
                                 &c->d_ctrl->inv_pose, camera_matrix(c->k[0], c->k[1], c->k[2], c->k[3]),
                                 c->cfg.mu, MAX_WEIGHT, &c->d_ctrl->integrate_gate, c->d_integrate_boxes,
                                 c->d_ctrl->integrate_box_count, c->d_bricks, tile_max_for_integrate(c), c->stream,
                                 &c->d_ctrl->integrate_boxes_total));
    }
    if (timed_frame(c)) {
        KFH_HIP(hipEventRecord(c->ev[c->frame % KFH_TIMING_RING][3], c->stream));
        if (c->timing >= 2 && c->timed_first < 0) c->timed_first = c->frame;
    }
    return KFH_OK;
}

static float raycast_step(const kfh_config &cfg) {
    const float mind = fminf(cfg.vol_dim[0], fminf(cfg.vol_dim[1], cfg.vol_dim[2]));
    int maxr = cfg.vol_res[0];
    if (cfg.vol_res[1] > maxr) maxr = cfg.vol_res[1];
    if (cfg.vol_res[2] > maxr) maxr = cfg.vol_res[2];
    return mind / (float)maxr;
}

// The frame's raycast, whole or in parts.  The second pass (the rays the march left unfinished) may be postponed:
// nothing reads the model maps before the next frame's tracking, so it waits for the next frame's preprocessing
// and shares that launch.  Every entry point that reads the maps, waits for the stream, hands out pointers or
// changes what the pass reads (pose, volume, task list) launches a postponed pass first (flush_tail).
static hipError_t raycast_part(kfh_context *c, int part, const PreprocessLaunch *pre) {
    static const bool eager = std::getenv("KFH_EAGER_MODEL_MAPS") != nullptr;     // experiments: all four maps per raycast
    c->model_maps_stale = !eager;
    return launch_raycast_part(eager ? c->d_vertex : nullptr, eager ? c->d_normal : nullptr, eager ? c->d_ray_t : nullptr,
                               c->d_ref_tn, c->w, c->h, c->d_volume, c->cfg.vol_res,
                               c->cfg.vol_dim, 0, c->cfg.vol_res[2], &c->d_ctrl->view, NEAR_PLANE, FAR_PLANE,
                               raycast_step(c->cfg), 0.75f * c->cfg.mu, c->d_ray_tasks, &c->d_ctrl->ray_task_count,
                               c->d_bricks, part, pre, c->stream);
}

static bool tail_may_wait(const kfh_context *c) {
    static const bool off = std::getenv("KFH_NO_DEFERRED_TAIL") != nullptr;
    return !off && !c->sharded && !c->overlap && c->timing != 1;
}

static int flush_tail(kfh_context *c) {
    if (c->tail_pending) {
        c->tail_pending = false;
        KFH_HIP(raycast_part(c, RAYCAST_TAIL, nullptr));
    }
    return KFH_OK;
}

// Before a raycast launch: the list of unfinished rays must be empty and the matrices tracking rebuilds the model
// vertices from (raycast_pose, raycast_view, project_ref) must be those of the pose that is rendered.  The tracking
// finish kernel has done both for the first raycast after a kfh_track; any other gets them done here.
static int prepare_ray_list(kfh_context *c) {
    if (!c->ray_list_fresh) {
        KFH_HIP(hipMemsetAsync(&c->d_ctrl->ray_task_count, 0, sizeof(unsigned), c->stream));
        KFH_HIP(launch_pose_derive(c->d_ctrl, c->k, 1, c->stream));
    }
    c->ray_list_fresh = false;
    return KFH_OK;
}

int kfh_raycast(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    if (c->frame > 2 && !c->sharded) {
        if (int rc = prepare_ray_list(c)) return rc;
        if (tail_may_wait(c)) {
            KFH_HIP(raycast_part(c, RAYCAST_MARCH, nullptr));
            c->tail_pending = true;
        } else {
            KFH_HIP(raycast_part(c, RAYCAST_BOTH, nullptr));
        }
    }
    if (c->timing == 1) {
        KFH_HIP(hipEventRecord(c->ev[c->frame % KFH_TIMING_RING][4], c->stream));
        if (c->timed_first < 0) c->timed_first = c->frame;
    }
    return KFH_OK;
}

// The march of this context's slab into `segments` (a packed map of the compute size): the context's own packed map
// for kfh_raycast_slab, a map of its own for the members of a group (whose peers read it while the context already
// composites into its packed map).
static int raycast_slab_to(kfh_context *c, float4 *segments) {
    if (int rc = flush_tail(c)) return rc;
    if (c->frame > 2) {
        if (int rc = prepare_ray_list(c)) return rc;
        c->model_maps_stale = true;
        KFH_HIP(launch_raycast(nullptr, nullptr, nullptr, segments, c->w, c->h, c->d_volume,
                               c->cfg.vol_res, c->cfg.vol_dim, c->cfg.slab_z0, c->cfg.slab_z1, &c->d_ctrl->view, NEAR_PLANE,
                               FAR_PLANE, raycast_step(c->cfg), 0.75f * c->cfg.mu, c->d_ray_tasks,
                               &c->d_ctrl->ray_task_count, c->d_bricks, c->stream));
    }
    return KFH_OK;
}

int kfh_raycast_slab(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    return raycast_slab_to(c, c->d_ref_tn);
}

int kfh_composite_slabs(kfh_context *c, const void *d_tn_parts, int n_parts) {
    KFH_REQUIRE(c != nullptr && d_tn_parts != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    KFH_REQUIRE(n_parts >= 1 && n_parts <= 64, "n_parts=%d out of range", n_parts);
    if (c->frame > 2) {
        c->model_maps_stale = true;
        KFH_HIP(launch_composite(c->d_ref_tn, static_cast<const float4 *>(d_tn_parts), n_parts, c->w * c->h, c->stream));
    }
    return KFH_OK;
}

int kfh_composite_tile(kfh_context *c, const void *d_tn_parts, int n_parts, int first_pixel, int pixel_count,
                       void *d_out) {
    KFH_REQUIRE(c != nullptr && d_tn_parts != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    KFH_REQUIRE(n_parts >= 1 && n_parts <= 64, "n_parts=%d out of range", n_parts);
    KFH_REQUIRE(first_pixel >= 0 && pixel_count >= 1 && first_pixel <= c->w * c->h - pixel_count,
                "pixels [%d, %d + %d) do not lie inside the %d x %d image", first_pixel, first_pixel, pixel_count, c->w, c->h);
    if (c->frame > 2) {
        c->model_maps_stale = true;      // the caller completes the packed map behind this call
        float4 *out = d_out ? static_cast<float4 *>(d_out) : c->d_ref_tn + first_pixel;
        KFH_HIP(launch_composite(out, static_cast<const float4 *>(d_tn_parts), n_parts, pixel_count, c->stream));
    }
    return KFH_OK;
}

int kfh_enable_track_render(kfh_context *c, int enabled) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    const size_t npix = (size_t)c->w * c->h;
    if (enabled && !c->d_track_codes) {
        KFH_HIP(dev_alloc(&c->d_track_codes, npix));
        KFH_HIP(hipMemsetAsync(c->d_track_codes, 0, npix, c->stream));
    } else if (!enabled && c->d_track_codes) {
        KFH_HIP(hipStreamSynchronize(c->stream));   // an iteration launch may still be writing codes
        KFH_HIP(hipFree(c->d_track_codes));
        c->d_track_codes = nullptr;
    }
    return KFH_OK;
}

int kfh_render_device(kfh_context *c, int image, const float view_pose[16], uint8_t *d_rgba) {
    KFH_REQUIRE(c != nullptr && d_rgba != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    const int npix = c->w * c->h;
    uchar4 *out = reinterpret_cast<uchar4 *>(d_rgba);
    KFH_HIP(ensure_pre_done(c));
    switch (image) {
        case KFH_RENDER_DEPTH:
            KFH_HIP(launch_render_depth(out, c->d_float_depth, npix, NEAR_PLANE, FAR_PLANE, c->stream));
            break;
        case KFH_RENDER_TRACK:
            if (!c->d_track_codes)
                return fail(KFH_ERR_STATE, "tracking keeps no result codes: call kfh_enable_track_render first");
            KFH_HIP(launch_render_track(out, nullptr, c->d_track_codes, c->w, c->w, c->h, c->stream));
            break;
        case KFH_RENDER_VOLUME: {
            if (c->sharded) return fail(KFH_ERR_STATE, "a z-slab context holds only part of the volume");
            if (!c->d_render_vertex) KFH_HIP(dev_alloc(&c->d_render_vertex, (size_t)npix));
            if (!c->d_render_normal) KFH_HIP(dev_alloc(&c->d_render_normal, (size_t)npix));
            if (!c->d_render_state) KFH_HIP(dev_alloc(&c->d_render_state, (size_t)512));
            const Mat4 *d_view = &c->d_ctrl->view;             // current pose * inverse camera matrix
            if (view_pose) {
                Mat4 pose;
                std::memcpy(pose.m, view_pose, sizeof(pose.m));
                Mat4 *dv = reinterpret_cast<Mat4 *>(c->d_render_state);
                KFH_HIP(launch_view_from_pose(dv, pose, c->k, c->stream));
                d_view = dv;
            }
            unsigned *d_count = reinterpret_cast<unsigned *>(c->d_render_state + 256);
            KFH_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned), c->stream));
            // the frame's own raycast is over (stream order): its task list is free to reuse
            KFH_HIP(launch_raycast(c->d_render_vertex, c->d_render_normal, nullptr, nullptr, c->w, c->h, c->d_volume,
                                   c->cfg.vol_res, c->cfg.vol_dim, 0, c->cfg.vol_res[2], d_view, NEAR_PLANE,
                                   2.0f * FAR_PLANE, raycast_step(c->cfg), 0.75f * c->cfg.mu, c->d_ray_tasks, d_count,
                                   c->d_bricks, c->stream));
            const float light[3] = {1.f, 1.f, -1.f}, ambient[3] = {0.1f, 0.1f, 0.1f};
            KFH_HIP(launch_render_shade(out, c->d_render_vertex, c->d_render_normal, npix, light, ambient, c->stream));
            break;
        }
        default: return fail(KFH_ERR_INVALID_ARG, "unknown image id %d", image);
    }
    return KFH_OK;
}

int kfh_render(kfh_context *c, int image, const float view_pose[16], uint8_t *h_rgba, size_t bytes) {
    KFH_REQUIRE(c != nullptr && h_rgba != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    const size_t need = (size_t)c->w * c->h * 4;
    KFH_REQUIRE(bytes >= need, "image needs %zu bytes, destination holds %zu", need, bytes);
    if (!c->d_render_rgba) KFH_HIP(dev_alloc(&c->d_render_rgba, (size_t)c->w * c->h));
    if (int rc = kfh_render_device(c, image, view_pose, reinterpret_cast<uint8_t *>(c->d_render_rgba))) return rc;
    KFH_HIP(hipMemcpyAsync(h_rgba, c->d_render_rgba, need, hipMemcpyDeviceToHost, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    return KFH_OK;
}

int kfh_end_frame(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    KFH_HIP(ensure_pre_done(c));
    // this frame is done with its depth images: only the side stream of the overlap mode ever waits for that (an
    // event record occupies the stream for a few microseconds, so it is not made for nothing)
    if (c->overlap) KFH_HIP(hipEventRecord(c->ev_set_free[c->cur_set], c->stream));
    c->pose_read_last_frame = c->pose_read_this_frame;
    c->pose_read_this_frame = false;
    c->frame++;
    return KFH_OK;
}

int kfh_process_frame(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);          // the calls below then find the right device current
    if (int rc = kfh_preprocess(c)) return rc;
    if (int rc = kfh_track(c)) return rc;
    if (int rc = kfh_integrate(c)) return rc;
    if (int rc = kfh_raycast(c)) return rc;
    return kfh_end_frame(c);
}

int kfh_sync(kfh_context *c) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    KFH_HIP(hipStreamSynchronize(c->pre_stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    return KFH_OK;
}

int kfh_get_pose(kfh_context *c, float pose[16]) {
    KFH_REQUIRE(c != nullptr && pose != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    // the pose and the status record are final once tracking has finished: a postponed raycast pass may stay postponed
    c->pose_read_this_frame = true;
    if (c->mail_current) {
        // the last finish launch left the pose in host memory: wait for the queued work as before, then read it there
        KFH_HIP(hipStreamSynchronize(c->stream));
        if (__atomic_load_n(&c->h_mail->seq, __ATOMIC_ACQUIRE) == c->mail_seq) {
            std::memcpy(pose, c->h_mail->pose, sizeof(Mat4));
            return KFH_OK;
        }
    }
    KFH_HIP(hipMemcpyAsync(pose, &c->d_ctrl->pose, sizeof(Mat4), hipMemcpyDeviceToHost, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    return KFH_OK;
}

int kfh_set_pose(kfh_context *c, const float pose[16]) {
    KFH_REQUIRE(c != nullptr && pose != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    KFH_HIP(hipMemcpyAsync(&c->d_ctrl->pose, pose, sizeof(Mat4), hipMemcpyHostToDevice, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    c->mail_current = false;       // the record in host memory is the tracked pose, not this one
    KFH_HIP(launch_pose_derive(c->d_ctrl, c->k, 0, c->stream));
    c->ray_list_fresh = false;     // a raycast that follows renders from this pose, not from the tracked one
    return KFH_OK;
}

int kfh_get_status(kfh_context *c, int frame, kfh_frame_status *out) {
    KFH_REQUIRE(c != nullptr && out != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    // the pose and the status record are final once tracking has finished: a postponed raycast pass may stay postponed
    KFH_REQUIRE(frame >= 0 && frame < c->frame && frame >= c->frame - KFH_STATUS_RING,
                "frame %d is not in the retained window (%d frames processed)", frame, c->frame);
    KFH_HIP(hipMemcpyAsync(out, c->d_status + (frame % KFH_STATUS_RING), sizeof(kfh_frame_status),
                           hipMemcpyDeviceToHost, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    return KFH_OK;
}

int kfh_integrate_totals(kfh_context *c, uint64_t *boxes, uint64_t *launches, int32_t *voxels_per_box) {
    KFH_REQUIRE(c != nullptr && boxes != nullptr && launches != nullptr && voxels_per_box != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    unsigned long long t[2] = {0, 0};
    KFH_HIP(hipMemcpyAsync(t, &c->d_ctrl->integrate_boxes_total, sizeof(t), hipMemcpyDeviceToHost, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    *boxes = t[0];
    *launches = t[1];
    *voxels_per_box = integrate_box_voxels(c->zs1);
    return KFH_OK;
}

int kfh_frame_index(kfh_context *c, int *frame) {
    KFH_REQUIRE(c != nullptr && frame != nullptr, "NULL argument");
    *frame = c->frame;
    return KFH_OK;
}

int kfh_compute_size(kfh_context *c, int *w, int *h) {
    KFH_REQUIRE(c != nullptr && w != nullptr && h != nullptr, "NULL argument");
    *w = c->w;
    *h = c->h;
    return KFH_OK;
}

int kfh_device_ptr(kfh_context *c, int buffer, int level, void **d_ptr, size_t *bytes) {
    KFH_REQUIRE(c != nullptr && d_ptr != nullptr && bytes != nullptr, "NULL argument");
    KFH_REQUIRE(level >= 0 && level < 3, "level %d out of range", level);
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;     // the caller is about to read or copy the buffer
    const size_t npix = (size_t)c->w * c->h;
    const size_t nl = (size_t)(c->w >> level) * (c->h >> level);
    KFH_HIP(ensure_pre_done(c));     // the depth images may still be in the making on the other stream
    switch (buffer) {
        case KFH_BUF_FLOAT_DEPTH: *d_ptr = c->d_float_depth; *bytes = npix * sizeof(float); break;
        case KFH_BUF_SCALED_DEPTH: *d_ptr = c->d_scaled[level]; *bytes = nl * sizeof(float); break;
        case KFH_BUF_INPUT_VERTEX:
        case KFH_BUF_INPUT_NORMAL:
            // not needed by the pipeline itself: built from the depth pyramid of the last tracked frame on request
            if (c->in_maps_version[level] != c->depth_version) {
                const float s = (float)(1 << level);
                KFH_HIP(launch_vertex_normal(c->d_in_vertex[level], c->d_in_normal[level], c->d_scaled[level],
                                             c->w >> level, c->h >> level, c->k[0] / s, c->k[1] / s, c->k[2] / s,
                                             c->k[3] / s, c->stream));
                c->in_maps_version[level] = c->depth_version;
            }
            *d_ptr = buffer == KFH_BUF_INPUT_VERTEX ? (void *)c->d_in_vertex[level] : (void *)c->d_in_normal[level];
            *bytes = nl * sizeof(float4);
            break;
        case KFH_BUF_MODEL_VERTEX:
        case KFH_BUF_MODEL_NORMAL:
        case KFH_BUF_RAY_T:
            if (c->model_maps_stale) {
                KFH_HIP(launch_model_maps(c->d_vertex, c->d_normal, c->d_ray_t, c->d_ref_tn, &c->d_ctrl->raycast_view, c->w,
                                          c->h, c->stream));
                c->model_maps_stale = false;
            }
            if (buffer == KFH_BUF_RAY_T) { *d_ptr = c->d_ray_t; *bytes = npix * sizeof(float); }
            else { *d_ptr = buffer == KFH_BUF_MODEL_VERTEX ? c->d_vertex : c->d_normal; *bytes = npix * sizeof(float4); }
            break;
        case KFH_BUF_MODEL_TN: *d_ptr = c->d_ref_tn; *bytes = npix * sizeof(float4); break;
        case KFH_BUF_VOLUME: *d_ptr = c->d_volume; *bytes = c->volume_voxels * sizeof(short2); break;
        case KFH_BUF_BRICK_MARKS:
            if (!c->d_bricks) return fail(KFH_ERR_STATE, "this context keeps no brick marks (KFH_NO_BRICK_MARKS)");
            *d_ptr = c->d_bricks;
            *bytes = brick_mark_bytes(c->cfg.vol_res);
            break;
        default: return fail(KFH_ERR_INVALID_ARG, "unknown buffer id %d", buffer);
    }
    return KFH_OK;
}

int kfh_download(kfh_context *c, int buffer, int level, void *h_dst, size_t bytes) {
    KFH_REQUIRE(h_dst != nullptr, "destination is NULL");
    void *src = nullptr;
    size_t have = 0;
    if (int rc = kfh_device_ptr(c, buffer, level, &src, &have)) return rc;
    KFH_REQUIRE(bytes <= have, "requested %zu bytes, buffer holds %zu", bytes, have);
    DeviceScope on_device(c->device);
    KFH_HIP(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    KFH_HIP(hipStreamSynchronize(c->stream));
    return KFH_OK;
}

int kfh_stream(kfh_context *c, void **stream) {
    KFH_REQUIRE(c != nullptr && stream != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;     // work the caller queues must find the frame complete
    *stream = c->stream;
    return KFH_OK;
}

int kfh_set_timing(kfh_context *c, int mode) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    KFH_REQUIRE(mode >= 0 && mode <= 3, "timing mode must be 0, 1, 2 or 3");
    if (mode != 0 && !c->ev) {
        c->ev = new (std::nothrow) hipEvent_t[KFH_TIMING_RING][5]();
        if (!c->ev) return fail(KFH_ERR_STATE, "out of host memory");
        for (int i = 0; i < KFH_TIMING_RING; ++i)
            for (int j = 0; j < 5; ++j) KFH_HIP(hipEventCreate(&c->ev[i][j]));
    }
    c->timing = mode;
    c->timed_first = -1;
    return KFH_OK;
}

int kfh_phase_ms_frame(kfh_context *c, int frame, float ms[4]) {
    KFH_REQUIRE(c != nullptr && ms != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (!c->timing || c->timed_first < 0)
        return fail(KFH_ERR_STATE, "no frame has been processed with timing enabled");
    KFH_REQUIRE(frame >= c->timed_first && frame < c->frame && frame >= c->frame - KFH_TIMING_RING &&
                    (c->timing != 3 || frame % 16 == 0),
                "frame %d has no retained timing record", frame);
    KFH_HIP(hipStreamSynchronize(c->stream));
    hipEvent_t *e = c->ev[frame % KFH_TIMING_RING];
    for (int i = 0; i < 4; ++i) ms[i] = 0.f;
    if (c->timing == 1) {
        for (int i = 0; i < 4; ++i) KFH_HIP(hipEventElapsedTime(&ms[i], e[i], e[i + 1]));
    } else {
        KFH_HIP(hipEventElapsedTime(&ms[2], e[2], e[3]));
    }
    return KFH_OK;
}

int kfh_phase_ms(kfh_context *c, float ms[4]) {
    KFH_REQUIRE(c != nullptr, "context is NULL");
    DeviceScope on_device(c->device);
    return kfh_phase_ms_frame(c, c->frame - 1, ms);
}

// Diagnostic (not part of include/kfusion_hip.h): microseconds per pose update (6x6 solve, exponential, pose
// product) in a dependent chain on one wavefront, from two runs of different length.
int kfh_debug_solver_us(const float *h_sums32, float *us_per_update) {
    float *d = nullptr;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&d), 48 * sizeof(float)));
    float host[48] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(host + 16, h_sums32, 32 * sizeof(float));
    hipEvent_t e0, e1;
    KFH_HIP(hipEventCreate(&e0));
    KFH_HIP(hipEventCreate(&e1));
    float ms[2] = {0, 0};
    const int n[2] = {2000, 4000};
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < 2; ++k) {
            KFH_HIP(hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice));
            KFH_HIP(hipEventRecord(e0, nullptr));
            KFH_HIP(launch_update_pose_chain(d, d + 16, 0.f, n[k], nullptr));
            KFH_HIP(hipEventRecord(e1, nullptr));
            KFH_HIP(hipEventSynchronize(e1));
            KFH_HIP(hipEventElapsedTime(&ms[k], e0, e1));
        }
    *us_per_update = (ms[1] - ms[0]) * 1e3f / (float)(n[1] - n[0]);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return KFH_OK;
}

// Diagnostic (not part of include/kfusion_hip.h): fuse the current depth map once more (this changes the
// volume) with per-wavefront records {start tick, end tick, candidate voxels, voxels written}.
// h_out holds n_values = boxes * 4 wavefronts * 4 values + 1; the last one receives the number of boxes kept.
int kfh_debug_integrate_stats(kfh_context *c, unsigned long long *h_out, size_t n_values) {
    KFH_REQUIRE(c != nullptr && h_out != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    if (!icp_stamps_compiled()) return fail(KFH_ERR_STATE, "this library was built without diagnostics (make -C slambench_amd/csrc clean && make -C slambench_amd/csrc STAMPS=1)");
    unsigned long long *d = nullptr;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&d), n_values * sizeof(unsigned long long)));
    KFH_HIP(hipMemset(d, 0, n_values * sizeof(unsigned long long)));
    KFH_HIP(hipMemsetAsync(c->d_ctrl->integrate_box_count, 0, 2 * sizeof(unsigned), c->stream));
    set_integrate_debug(d);
    c->box_list_fresh = false;
    const float *src = c->cfg.integrate_filtered ? c->d_scaled[0] : c->d_float_depth;
    hipError_t e = launch_integrate(c->d_volume, c->cfg.vol_res, c->cfg.vol_dim, c->zs0, c->zs1, src, c->w, c->h,
                                    &c->d_ctrl->inv_pose, camera_matrix(c->k[0], c->k[1], c->k[2], c->k[3]),
                                    c->cfg.mu, MAX_WEIGHT, nullptr, c->d_integrate_boxes,
                                    c->d_ctrl->integrate_box_count, c->d_bricks, tile_max_for_integrate(c), c->stream);
    set_integrate_debug(nullptr);
    unsigned counts[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(counts, c->d_ctrl->integrate_box_count, sizeof(counts), hipMemcpyDeviceToHost,
                                            c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    // items of the front part take the record slots [0, counts[0]), those of the back part follow
    h_out[n_values - 1] = (unsigned long long)counts[0] + counts[1];
    if (e == hipSuccess) e = hipMemcpy(h_out, d, (n_values - 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    KFH_HIP(e);
    return KFH_OK;
}

// Diagnostic (not part of include/kfusion_hip.h): run the raycast of the current state once more with
// per-wavefront records {start tick, end tick, batches, tail rounds} (100 MHz ticks) and copy them out.
// h_out holds ((w+15)/16) * ((h+15)/16) * 4 wavefronts * 4 values, then one more: the number of rays the
// second pass received.
int kfh_debug_raycast_stats(kfh_context *c, unsigned long long *h_out) {
    KFH_REQUIRE(c != nullptr && h_out != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    if (!icp_stamps_compiled()) return fail(KFH_ERR_STATE, "this library was built without diagnostics (make -C slambench_amd/csrc clean && make -C slambench_amd/csrc STAMPS=1)");
    const size_t n = (size_t)((c->w + 15) / 16) * ((c->h + 15) / 16) * 16;
    unsigned long long *d = nullptr;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(unsigned long long)));
    KFH_HIP(hipMemset(d, 0, n * sizeof(unsigned long long)));
    KFH_HIP(hipMemsetAsync(&c->d_ctrl->ray_task_count, 0, sizeof(unsigned), c->stream));
    set_raycast_debug(d);
    c->ray_list_fresh = false;
    hipError_t e = launch_raycast(c->d_vertex, c->d_normal, c->d_ray_t, c->d_ref_tn, c->w, c->h, c->d_volume,
                                  c->cfg.vol_res, c->cfg.vol_dim, 0, c->cfg.vol_res[2], &c->d_ctrl->view, NEAR_PLANE,
                                  FAR_PLANE, raycast_step(c->cfg), 0.75f * c->cfg.mu, c->d_ray_tasks,
                                  &c->d_ctrl->ray_task_count, c->d_bricks, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_out + n, &c->d_ctrl->ray_task_count, sizeof(unsigned),
                                            hipMemcpyDeviceToHost, c->stream);
    set_raycast_debug(nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(h_out, d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    KFH_HIP(e);
    return KFH_OK;
}

// Diagnostic (not part of include/kfusion_hip.h): copy the ICP time stamps of the last frame,
// [64 launches][320 blocks][8 slots] ticks of the 100 MHz real-time counter.  Only filled when the context
// was created with KFH_ICP_STAMPS set.
int kfh_debug_icp_stamps(kfh_context *c, unsigned long long *h_out) {
    KFH_REQUIRE(c != nullptr && h_out != nullptr, "NULL argument");
    DeviceScope on_device(c->device);
    if (int rc = flush_tail(c)) return rc;
    if (!c->d_stamps) return fail(KFH_ERR_STATE, "context was created without KFH_ICP_STAMPS");
    if (!icp_stamps_compiled()) return fail(KFH_ERR_STATE, "this library was built without diagnostics (make -C slambench_amd/csrc clean && make -C slambench_amd/csrc STAMPS=1)");
    KFH_HIP(hipStreamSynchronize(c->stream));
    KFH_HIP(hipMemcpy(h_out, c->d_stamps,
                      (size_t)KFH_STAMP_LAUNCHES * KFH_STAMP_BLOCKS * KFH_STAMP_SLOTS * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
    return KFH_OK;
}

// ------------------------------------------------------------------------------------------------------
// stateless kernels
// ------------------------------------------------------------------------------------------------------
int kfh_k_preprocess(float *d_raw, float *d_filtered, int ow, int oh, const uint16_t *d_in, int iw, int ih,
                     float depth_divisor, void *stream) {
    KFH_REQUIRE(d_in != nullptr && (d_raw || d_filtered), "NULL argument");
    KFH_REQUIRE(ow > 0 && oh > 0 && iw >= ow && ih >= oh, "bad sizes");
    KFH_REQUIRE(iw % ow == 0 && ih % oh == 0 && iw / ow == ih / oh, "input size must be an integer multiple of the output size");
    KFH_REQUIRE(depth_divisor > 0.f, "depth_divisor must be positive");
    KFH_HIP(launch_preprocess(d_raw, d_filtered, ow, oh, d_in, true, iw, iw / ow, depth_divisor, nullptr, nullptr, 0.f,
                              nullptr, static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_bilateral(float *d_out, const float *d_in, int w, int h, void *stream) {
    KFH_REQUIRE(d_out && d_in && w > 0 && h > 0, "bad argument");
    KFH_HIP(launch_preprocess(nullptr, d_out, w, h, d_in, false, w, 1, 1.0f, nullptr, nullptr, 0.f, nullptr,
                              static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_half_sample(float *d_out, const float *d_in, int iw, int ih, float e_d, void *stream) {
    KFH_REQUIRE(d_out && d_in && iw >= 2 && ih >= 2 && iw % 2 == 0 && ih % 2 == 0, "bad argument");
    KFH_HIP(launch_half_sample(d_out, d_in, iw, ih, e_d, static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_vertex_normal(void *d_vertex4, void *d_normal4, const float *d_depth, int w, int h, float fx, float fy,
                        float cx, float cy, void *stream) {
    KFH_REQUIRE(d_vertex4 && d_normal4 && d_depth && w > 0 && h > 0, "bad argument");
    KFH_HIP(launch_vertex_normal(static_cast<float4 *>(d_vertex4), static_cast<float4 *>(d_normal4), d_depth, w, h,
                                 fx, fy, cx, cy, static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_track_reduce(float *d_sums, kfh_track_data *d_track, int track_stride, const float *d_in_depth,
                       const float k[4], const void *d_in_normal4, int iw, int ih, const void *d_ref_vertex4,
                       const void *d_ref_normal4, int rw, int rh, const float Ttrack[16], const float view[16],
                       float dist_threshold, float normal_threshold, void *stream) {
    KFH_REQUIRE(d_sums && d_in_depth && k && d_in_normal4 && d_ref_vertex4 && d_ref_normal4 && Ttrack && view,
                "NULL argument");
    KFH_REQUIRE(iw > 0 && ih > 0 && rw > 0 && rh > 0, "bad sizes");
    KFH_REQUIRE(k[0] != 0.f && k[1] != 0.f, "focal lengths must be non-zero");
    KFH_REQUIRE(!d_track || track_stride >= iw, "track_stride must be >= iw");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int tpw, tppt;
    track_shape(iw, ih, 256, &tpw, &tppt);
    const int nb = track_blocks(iw, ih, tpw, tppt);
    char *scratch = nullptr;   // [control block | partial rows]
    const size_t ctrl_bytes = (sizeof(Ctrl) + 255) & ~(size_t)255;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&scratch), ctrl_bytes + (size_t)nb * 32 * sizeof(float)));
    Mat4 T, V;
    std::memcpy(T.m, Ttrack, sizeof(T.m));
    std::memcpy(V.m, view, sizeof(V.m));
    hipError_t e = launch_track_reduce(reinterpret_cast<Ctrl *>(scratch), reinterpret_cast<float *>(scratch + ctrl_bytes),
                                       d_sums, d_track, track_stride, d_in_depth, k,
                                       static_cast<const float4 *>(d_in_normal4), iw, ih,
                                       static_cast<const float4 *>(d_ref_vertex4),
                                       static_cast<const float4 *>(d_ref_normal4), rw, rh, T, V, dist_threshold,
                                       normal_threshold, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    KFH_HIP(e);
    return KFH_OK;
}

int kfh_k_update_pose(float *d_pose16, double *d_x6, int *d_converged, const float *d_sums, float icp_threshold,
                      void *stream) {
    KFH_REQUIRE(d_pose16 && d_x6 && d_converged && d_sums, "NULL argument");
    KFH_HIP(launch_update_pose(d_pose16, d_x6, d_converged, d_sums, icp_threshold, static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_integrate(void *d_volume, const int32_t res[3], const float dim[3], int z0, int z1, const float *d_depth,
                    int w, int h, const float invTrack[16], const float K[16], float mu, float maxweight,
                    void *stream) {
    KFH_REQUIRE(d_volume && res && dim && d_depth && invTrack && K, "NULL argument");
    KFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && 0 <= z0 && z0 < z1 && z1 <= res[2], "bad volume range");
    KFH_REQUIRE(w > 0 && h > 0 && mu > 0.f, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch: [inverse pose | box counter | box list]
    char *scratch = nullptr;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&scratch), 512 + integrate_box_list_bytes(res, z1)));
    Mat4 *d_inv = reinterpret_cast<Mat4 *>(scratch);
    unsigned *d_count = reinterpret_cast<unsigned *>(scratch + 256);
    Mat4 Km;
    std::memcpy(Km.m, K, sizeof(Km.m));
    hipError_t e = hipMemcpyAsync(d_inv, invTrack, sizeof(Mat4), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned), s);
    if (e == hipSuccess)
        e = launch_integrate(static_cast<short2 *>(d_volume), res, dim, z0, z1, d_depth, w, h, d_inv, Km, mu,
                             maxweight, nullptr, scratch + 512, d_count, nullptr, nullptr, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    KFH_HIP(e);
    return KFH_OK;
}

// ------------------------------------------------------------------------------------------------------
// A group: ONE sequence whose volume is split into z-slabs over several contexts (one per listed device; a device
// may be listed more than once), all driven from the calling thread.  This is the z-slab path for callers that are
// a single process with a single thread of control - the SLAMBench harness loads a plugin into one process and
// calls it frame by frame (framework/shared/src/SLAMBenchConfiguration.cpp:230-342) - where one process per GPU
// and a collective library are not to be had.  Per frame every member preprocesses the frame, tracks on its own
// copy of the model map (identical inputs: identical poses), integrates its slab and marches its slab's ray
// segments into a map of its own; then every member composites the nearest hit per pixel straight out of all
// members' segment maps (its own memory and its peers', read over the links: no copy, no collective).  Ordering
// between the devices is by events: a member composites when all marches of the frame are done, and marches the
// next frame's segments when all composites of this frame are done.
// ------------------------------------------------------------------------------------------------------
namespace {
// One host thread per group member, so that the members' launches of a frame (25 each) are queued side by side instead
// of one member after the other.  run(job) hands job(i) to thread i and returns when all have finished: the first
// non-zero result, with that thread's error text as the caller's.
struct GroupWorkers {
    std::vector<std::thread> threads;
    std::mutex lock;
    std::condition_variable go, done;
    std::function<int(int)> job;
    std::vector<int> result;
    std::vector<std::string> text;
    std::atomic<unsigned> generation{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};
    // The rounds of a frame follow each other within microseconds, and waking a sleeping thread costs more than a
    // round's launches: both sides poll for this long before they go to sleep on the condition variable.
    static constexpr std::chrono::microseconds SPIN{200};

    static bool spin_until(const std::function<bool()> &done_now) {
        const auto t0 = std::chrono::steady_clock::now();
        while (!done_now()) {
            if (std::chrono::steady_clock::now() - t0 > SPIN) return false;
            __builtin_ia32_pause();
        }
        return true;
    }
    void start(int n, const std::vector<int> &device) {
        result.assign(n, 0);
        text.assign(n, std::string());
        for (int i = 0; i < n; ++i) threads.emplace_back([this, i, dev = device[i]] { loop(i, dev); });
    }
    void loop(int i, int device) {
        (void)hipSetDevice(device);
        unsigned seen = 0;
        for (;;) {
            std::function<int(int)> mine;
            spin_until([&] { return stop.load(std::memory_order_acquire) || generation.load(std::memory_order_acquire) != seen; });
            {
                std::unique_lock<std::mutex> hold(lock);
                go.wait(hold, [&] { return stop.load() || generation.load() != seen; });
                if (stop.load()) return;
                seen = generation.load();
                mine = job;
            }
            const int rc = mine(i);
            std::string why = rc != KFH_OK ? g_last_error : std::string();
            std::lock_guard<std::mutex> hold(lock);
            result[i] = rc;
            text[i].swap(why);
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) done.notify_one();
        }
    }
    int run(const std::function<int(int)> &f) {
        {
            std::lock_guard<std::mutex> hold(lock);
            job = f;
            pending.store((int)threads.size(), std::memory_order_release);
            generation.fetch_add(1, std::memory_order_acq_rel);
        }
        go.notify_all();
        if (!spin_until([&] { return pending.load(std::memory_order_acquire) == 0; })) {
            std::unique_lock<std::mutex> hold(lock);
            done.wait(hold, [&] { return pending.load() == 0; });
        }
        std::lock_guard<std::mutex> hold(lock);            // the last worker has left its critical section
        for (size_t i = 0; i < result.size(); ++i)
            if (result[i] != KFH_OK) return fail(result[i], "group member %zu: %s", i, text[i].c_str());
        return KFH_OK;
    }
    void finish() {
        {
            std::lock_guard<std::mutex> hold(lock);
            stop.store(true);
        }
        go.notify_all();
        for (std::thread &t : threads) t.join();
        threads.clear();
    }
};
constexpr std::chrono::microseconds GroupWorkers::SPIN;
}  // namespace

struct kfh_group {
    std::vector<kfh_context *> member;
    std::vector<float4 *> segments;          // per member: the packed map its slab march writes (on its device)
    std::vector<hipEvent_t> marched, composited;
    std::vector<float4 *> maps;              // per member: its packed model map (what its tracking reads)
    std::vector<hipEvent_t> shared;          // per member: its share of the image is in every member's map
    bool by_share = false;                   // the image is split among the members (from four members on)
    bool exchanged = false;                  // a composite has been queued since the group was created or reset
    GroupWorkers *workers = nullptr;         // null: the calling thread queues every member's launches itself
};

namespace {
// planes [z0, z1) of slab `index` out of `count`: contiguous, nearly equal, the first res % count one plane larger
void slab_range(int res, int count, int index, int *z0, int *z1) {
    const int base = res / count, extra = res % count;
    *z0 = index * base + (index < extra ? index : extra);
    *z1 = *z0 + base + (index < extra ? 1 : 0);
}

void group_free(kfh_group *g) {
    if (g->workers) {
        g->workers->finish();
        delete g->workers;
        g->workers = nullptr;
    }
    for (size_t i = 0; i < g->member.size(); ++i) {
        kfh_context *c = g->member[i];
        if (!c) continue;
        DeviceScope on_device(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (i < g->segments.size() && g->segments[i]) (void)hipFree(g->segments[i]);
        if (i < g->marched.size() && g->marched[i]) (void)hipEventDestroy(g->marched[i]);
        if (i < g->composited.size() && g->composited[i]) (void)hipEventDestroy(g->composited[i]);
        if (i < g->shared.size() && g->shared[i]) (void)hipEventDestroy(g->shared[i]);
    }
    for (kfh_context *c : g->member)
        if (c) (void)kfh_destroy(c);
    delete g;
}
}  // namespace

int kfh_group_create(const kfh_config *cfg, const int *devices, int n, kfh_group **out) {
    KFH_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    KFH_REQUIRE(cfg != nullptr && devices != nullptr, "NULL argument");
    KFH_REQUIRE(n >= 1 && n <= KFH_GROUP_MAX, "a group has 1 to %d members (got %d)", KFH_GROUP_MAX, n);
    KFH_REQUIRE(cfg->vol_res[2] >= n, "more members (%d) than z planes (%d)", n, cfg->vol_res[2]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(KFH_ERR_NO_DEVICE, "no HIP device is visible to this process");
    for (int i = 0; i < n; ++i)
        KFH_REQUIRE(devices[i] >= 0 && devices[i] < ndev, "device %d of member %d is not one of the %d visible devices",
                    devices[i], i, ndev);
    int caller_device = 0;
    KFH_HIP(hipGetDevice(&caller_device));
    kfh_group *g = new (std::nothrow) kfh_group();
    if (!g) return fail(KFH_ERR_STATE, "out of host memory");
    g->member.assign(n, nullptr);
    g->segments.assign(n, nullptr);
    g->marched.assign(n, nullptr);
    g->composited.assign(n, nullptr);
    g->shared.assign(n, nullptr);
    g->maps.assign(n, nullptr);
    // Every member compositing the whole image reads N - 1 whole segment maps from its peers; with the image split
    // among the members it reads N - 1 shares and writes its own share N - 1 times (two maps' worth in all), at the
    // price of a second round of events.  KFH_GROUP_EXCHANGE=gather / share decides otherwise.
    g->by_share = n >= 4;
    if (const char *e = std::getenv("KFH_GROUP_EXCHANGE")) {
        if (std::strcmp(e, "share") == 0) g->by_share = n > 1;
        else if (std::strcmp(e, "gather") == 0) g->by_share = false;
    }
    int rc = KFH_OK;
    for (int i = 0; i < n && rc == KFH_OK; ++i) {
        // every member reads every other member's segment map: the devices must see each other's memory
        for (int j = 0; j < n && rc == KFH_OK; ++j) {
            if (devices[j] == devices[i]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can)
                rc = fail(KFH_ERR_STATE, "device %d cannot read the memory of device %d (no peer access)", devices[i], devices[j]);
        }
    }
    for (int i = 0; i < n && rc == KFH_OK; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) { rc = fail(KFH_ERR_HIP, "hipSetDevice(%d) failed", devices[i]); break; }
        for (int j = 0; j < n; ++j) {
            if (devices[j] == devices[i]) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess) { rc = fail(KFH_ERR_HIP, "hipDeviceEnablePeerAccess(%d) on device %d: %s", devices[j], devices[i], hipGetErrorString(e)); break; }
        }
        if (rc != KFH_OK) break;
        kfh_config member_cfg = *cfg;
        if (n > 1) slab_range(cfg->vol_res[2], n, i, &member_cfg.slab_z0, &member_cfg.slab_z1);
        rc = kfh_create(&member_cfg, &g->member[i]);
        if (rc != KFH_OK) break;
        kfh_context *c = g->member[i];
        hipError_t e = dev_alloc(&g->segments[i], (size_t)c->w * c->h);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->marched[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->composited[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->shared[i], hipEventDisableTiming);
        g->maps[i] = c->d_ref_tn;
        if (e != hipSuccess) rc = fail(KFH_ERR_HIP, "group member %d: %s", i, hipGetErrorString(e));
    }
    (void)hipSetDevice(caller_device);
    if (rc != KFH_OK) {
        const std::string text = kfh_last_error();        // the clean-up below must not replace the reason
        group_free(g);
        return fail(rc, "%s", text.c_str());
    }
    // A thread per member when the members sit on more than one device (on one device the launches share the
    // device's command processor anyway); KFH_GROUP_THREADS=0 / 1 decides otherwise.
    bool threaded = false;
    for (int i = 1; i < n; ++i) threaded = threaded || devices[i] != devices[0];
    if (const char *e = std::getenv("KFH_GROUP_THREADS")) threaded = std::atoi(e) != 0;
    if (threaded && n > 1) {
        g->workers = new (std::nothrow) GroupWorkers();
        if (!g->workers) {
            group_free(g);
            return fail(KFH_ERR_STATE, "out of host memory");
        }
        g->workers->start(n, std::vector<int>(devices, devices + n));
    }
    *out = g;
    return KFH_OK;
}

int kfh_group_destroy(kfh_group *g) {
    if (!g) return KFH_OK;
    group_free(g);
    return KFH_OK;
}

int kfh_group_size(kfh_group *g, int *n) {
    KFH_REQUIRE(g != nullptr && n != nullptr, "NULL argument");
    *n = (int)g->member.size();
    return KFH_OK;
}

int kfh_group_member(kfh_group *g, int index, kfh_context **ctx) {
    KFH_REQUIRE(g != nullptr && ctx != nullptr, "NULL argument");
    KFH_REQUIRE(index >= 0 && index < (int)g->member.size(), "member %d of %zu", index, g->member.size());
    *ctx = g->member[index];
    return KFH_OK;
}

int kfh_group_reset(kfh_group *g) {
    KFH_REQUIRE(g != nullptr, "group is NULL");
    for (kfh_context *c : g->member) {
        if (int rc = kfh_sync(c)) return rc;       // no member may still read a segment map
    }
    for (kfh_context *c : g->member)
        if (int rc = kfh_reset(c)) return rc;
    g->exchanged = false;
    return KFH_OK;
}

int kfh_group_upload_depth(kfh_group *g, const uint16_t *h_depth) {
    KFH_REQUIRE(g != nullptr && h_depth != nullptr, "NULL argument");
    for (kfh_context *c : g->member)
        if (int rc = kfh_upload_depth(c, h_depth)) return rc;
    return KFH_OK;
}

int kfh_group_set_depth_device(kfh_group *g, const uint16_t *const *d_depth_per_member) {
    KFH_REQUIRE(g != nullptr && d_depth_per_member != nullptr, "NULL argument");
    for (size_t i = 0; i < g->member.size(); ++i)
        if (int rc = kfh_set_depth_device(g->member[i], d_depth_per_member[i])) return rc;
    return KFH_OK;
}

namespace {
// What member i queues before the exchange: the frame up to its slab's ray segments.
int group_member_first_half(kfh_group *g, int i, bool raycasts) {
    const int n = (int)g->member.size();
    kfh_context *c = g->member[i];
    DeviceScope on_device(c->device);
    if (int rc = kfh_preprocess(c)) return rc;
    if (int rc = kfh_track(c)) return rc;
    if (int rc = kfh_integrate(c)) return rc;
    if (raycasts) {
        // the march overwrites the map the peers composited from in the frame before (with the image split among the
        // members this member has already waited for their shares, which they wrote after reading its map)
        if (g->exchanged && !g->by_share)
            for (int j = 0; j < n; ++j)
                if (j != i) KFH_HIP(hipStreamWaitEvent(c->stream, g->composited[j], 0));
        if (int rc = raycast_slab_to(c, g->segments[i])) return rc;
        KFH_HIP(hipEventRecord(g->marched[i], c->stream));
    }
    return KFH_OK;
}

// pixels [first, first + count) of the image are member i's share
void group_share(const kfh_group *g, int i, int *first, int *count) {
    const int n = (int)g->member.size(), npix = g->member[0]->w * g->member[0]->h;
    const int share = ((npix + n - 1) / n + 63) / 64 * 64;
    *first = std::min(i * share, npix);
    *count = std::min(share, npix - *first);
}

// What member i queues once every member's `marched` event of this frame has been recorded on the host side.
int group_member_second_half(kfh_group *g, int i, bool raycasts) {
    const int n = (int)g->member.size();
    kfh_context *c = g->member[i];
    DeviceScope on_device(c->device);
    if (raycasts) {
        for (int j = 0; j < n; ++j)
            if (j != i) KFH_HIP(hipStreamWaitEvent(c->stream, g->marched[j], 0));
        c->model_maps_stale = true;
        if (g->by_share) {
            int first = 0, count = 0;
            group_share(g, i, &first, &count);
            KFH_HIP(launch_composite_share(g->maps.data(), g->segments.data(), n, first, count, c->stream));
            KFH_HIP(hipEventRecord(g->shared[i], c->stream));
            return KFH_OK;                  // the frame ends in the third round, behind the peers' shares
        }
        KFH_HIP(launch_composite_parts(c->d_ref_tn, g->segments.data(), n, c->w * c->h, c->stream));
        KFH_HIP(hipEventRecord(g->composited[i], c->stream));
    }
    return kfh_end_frame(c);
}

// Split image only: member i's map is complete when every peer has stored its share into it.
int group_member_third_half(kfh_group *g, int i) {
    const int n = (int)g->member.size();
    kfh_context *c = g->member[i];
    DeviceScope on_device(c->device);
    for (int j = 0; j < n; ++j)
        if (j != i) KFH_HIP(hipStreamWaitEvent(c->stream, g->shared[j], 0));
    return kfh_end_frame(c);
}
}  // namespace

int kfh_group_process_frame(kfh_group *g) {
    KFH_REQUIRE(g != nullptr, "group is NULL");
    const int n = (int)g->member.size();
    if (n == 1) return kfh_process_frame(g->member[0]);
    const bool raycasts = g->member[0]->frame > 2;
    // Two rounds with a host-side join between them: a stream can only be told to wait for an event that has been
    // recorded already, so every member's march must be queued before any member queues its composite.
    const bool third = raycasts && g->by_share;
    if (g->workers) {
        if (int rc = g->workers->run([g, raycasts](int i) { return group_member_first_half(g, i, raycasts); })) return rc;
        if (int rc = g->workers->run([g, raycasts](int i) { return group_member_second_half(g, i, raycasts); })) return rc;
        if (third)
            if (int rc = g->workers->run([g](int i) { return group_member_third_half(g, i); })) return rc;
    } else {
        for (int i = 0; i < n; ++i)
            if (int rc = group_member_first_half(g, i, raycasts)) return rc;
        for (int i = 0; i < n; ++i)
            if (int rc = group_member_second_half(g, i, raycasts)) return rc;
        if (third)
            for (int i = 0; i < n; ++i)
                if (int rc = group_member_third_half(g, i)) return rc;
    }
    if (raycasts) g->exchanged = true;
    return KFH_OK;
}

int kfh_group_sync(kfh_group *g) {
    KFH_REQUIRE(g != nullptr, "group is NULL");
    for (kfh_context *c : g->member)
        if (int rc = kfh_sync(c)) return rc;
    return KFH_OK;
}

int kfh_group_get_pose(kfh_group *g, float pose[16]) {
    KFH_REQUIRE(g != nullptr && pose != nullptr, "NULL argument");
    // every member holds the same pose (same inputs, same kernels); the others are waited for as well, so that the
    // frame is over on every device when this returns
    for (size_t i = 1; i < g->member.size(); ++i)
        if (int rc = kfh_sync(g->member[i])) return rc;
    return kfh_get_pose(g->member[0], pose);
}

// ------------------------------------------------------------------------------------------------------
// kfh_k_integrate for callers that cannot allocate.  The work list of the two integrate launches (what
// kfh_k_integrate obtains from hipMalloc per call) lives in a fixed array in device memory; every device the
// library runs on holds its own copy until the process ends.
// ------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
constexpr size_t KFH_NOALLOC_LIST_BYTES = (size_t)4 << 20;            // 524288 work items of 8 bytes: up to 2048^3 voxels
constexpr size_t KFH_NOALLOC_SCRATCH_BYTES = 512 + KFH_NOALLOC_LIST_BYTES;    // [inverse pose | box counter | box list]
constexpr int KFH_NOALLOC_MAX_DEVICES = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[KFH_NOALLOC_SCRATCH_BYTES];

namespace {
// Address of `scratch` on the current device.  In host code the array's name is only a handle: the address a
// kernel can use comes from hipGetSymbolAddress, asked once per device with that device current.
int noalloc_scratch(char **out) {
    static std::mutex lock;
    static char *address[KFH_NOALLOC_MAX_DEVICES] = {};
    int device = -1;
    KFH_HIP(hipGetDevice(&device));
    if (device < 0 || device >= KFH_NOALLOC_MAX_DEVICES)
        return fail(KFH_ERR_STATE, "device index %d is beyond the %d devices this entry point keeps scratch for", device,
                    KFH_NOALLOC_MAX_DEVICES);
    std::lock_guard<std::mutex> guard(lock);
    if (!address[device]) {
        void *p = nullptr;
        KFH_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(KFH_ERR_HIP, "hipGetSymbolAddress returned a null address for the scratch array");
        address[device] = static_cast<char *>(p);
    }
    *out = address[device];
    return KFH_OK;
}
}  // namespace

extern "C" {

int kfh_k_integrate_noalloc(void *d_volume, const int32_t res[3], const float dim[3], int z0, int z1,
                            const float *d_depth, int w, int h, const float invTrack[16], const float K[16], float mu,
                            float maxweight, void *stream) {
    KFH_REQUIRE(d_volume && res && dim && d_depth && invTrack && K, "NULL argument");
    KFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && 0 <= z0 && z0 < z1 && z1 <= res[2], "bad volume range");
    KFH_REQUIRE(w > 0 && h > 0 && mu > 0.f, "bad argument");
    const size_t list_bytes = integrate_box_list_bytes(res, z1);
    if (list_bytes > KFH_NOALLOC_LIST_BYTES)
        return fail(KFH_ERR_INVALID_ARG,
                    "a %dx%dx%d volume up to plane %d needs a work list of %zu bytes; kfh_k_integrate_noalloc holds %zu "
                    "(use kfh_k_integrate, which allocates per call)", res[0], res[1], res[2], z1, list_bytes,
                    KFH_NOALLOC_LIST_BYTES);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ptr = nullptr;
    if (int rc = noalloc_scratch(&ptr)) return rc;
    Mat4 *d_inv = reinterpret_cast<Mat4 *>(ptr);
    unsigned *d_count = reinterpret_cast<unsigned *>(ptr + 256);
    Mat4 inv, Km;
    std::memcpy(inv.m, invTrack, sizeof(inv.m));
    std::memcpy(Km.m, K, sizeof(Km.m));
    // the matrix travels as a kernel argument and the same launch empties the list left by the previous call
    KFH_HIP(launch_store_mat4(d_inv, inv, d_count, s));
    KFH_HIP(launch_integrate(static_cast<short2 *>(d_volume), res, dim, z0, z1, d_depth, w, h, d_inv, Km, mu, maxweight,
                             nullptr, ptr + 512, d_count, nullptr, nullptr, s));
    return KFH_OK;
}

int kfh_k_raycast(void *d_vertex4, void *d_normal4, int w, int h, const void *d_volume, const int32_t res[3],
                  const float dim[3], const float view[16], float near_plane, float far_plane, float step,
                  float largestep, void *stream) {
    KFH_REQUIRE(d_vertex4 && d_normal4 && d_volume && res && dim && view, "NULL argument");
    KFH_REQUIRE(w > 0 && h > 0 && res[0] >= 2 && res[1] >= 2 && res[2] >= 2, "bad sizes (a volume side has at least two voxels)");
    KFH_REQUIRE(step > 0.f && largestep > 0.f, "steps must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch: [view matrix | task counter | task list]
    char *scratch = nullptr;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&scratch), 512 + raycast_task_bytes(w, h)));
    Mat4 *d_view = reinterpret_cast<Mat4 *>(scratch);
    unsigned *d_count = reinterpret_cast<unsigned *>(scratch + 256);
    hipError_t e = hipMemcpyAsync(d_view, view, sizeof(Mat4), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, sizeof(unsigned), s);
    if (e == hipSuccess)
        e = launch_raycast(static_cast<float4 *>(d_vertex4), static_cast<float4 *>(d_normal4), nullptr, nullptr, w, h,
                           static_cast<const short2 *>(d_volume), res, dim, 0, res[2], d_view, near_plane,
                           far_plane, step, largestep, scratch + 512, d_count, nullptr, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    KFH_HIP(e);
    return KFH_OK;
}

int kfh_k_render_depth(uint8_t *d_rgba, const float *d_depth, int w, int h, float near_plane, float far_plane,
                       void *stream) {
    KFH_REQUIRE(d_rgba && d_depth, "NULL argument");
    KFH_REQUIRE(w > 0 && h > 0, "bad sizes");
    KFH_HIP(launch_render_depth(reinterpret_cast<uchar4 *>(d_rgba), d_depth, w * h, near_plane, far_plane,
                                static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_render_track(uint8_t *d_rgba, const kfh_track_data *d_track, int track_stride, int w, int h, void *stream) {
    KFH_REQUIRE(d_rgba && d_track, "NULL argument");
    KFH_REQUIRE(w > 0 && h > 0 && track_stride >= w, "bad sizes");
    KFH_HIP(launch_render_track(reinterpret_cast<uchar4 *>(d_rgba), d_track, nullptr, track_stride, w, h,
                                static_cast<hipStream_t>(stream)));
    return KFH_OK;
}

int kfh_k_render_volume(uint8_t *d_rgba, int w, int h, const void *d_volume, const int32_t res[3], const float dim[3],
                        const float view[16], float near_plane, float far_plane, float step, float largestep,
                        const float light[3], const float ambient[3], void *stream) {
    KFH_REQUIRE(d_rgba && d_volume && res && dim && view && light && ambient, "NULL argument");
    KFH_REQUIRE(w > 0 && h > 0 && res[0] >= 2 && res[1] >= 2 && res[2] >= 2, "bad sizes (a volume side has at least two voxels)");
    KFH_REQUIRE(step > 0.f && largestep > 0.f, "steps must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch: [view matrix | task counter | vertex map | normal map | task list]
    const size_t map_bytes = (size_t)w * h * sizeof(float4);
    char *scratch = nullptr;
    KFH_HIP(hipMalloc(reinterpret_cast<void **>(&scratch), 512 + 2 * map_bytes + raycast_task_bytes(w, h)));
    Mat4 *d_view = reinterpret_cast<Mat4 *>(scratch);
    unsigned *d_count = reinterpret_cast<unsigned *>(scratch + 256);
    float4 *d_vertex = reinterpret_cast<float4 *>(scratch + 512);
    float4 *d_normal = reinterpret_cast<float4 *>(scratch + 512 + map_bytes);
    hipError_t e = hipMemcpyAsync(d_view, view, sizeof(Mat4), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, sizeof(unsigned), s);
    if (e == hipSuccess)
        e = launch_raycast(d_vertex, d_normal, nullptr, nullptr, w, h, static_cast<const short2 *>(d_volume), res, dim, 0,
                           res[2], d_view, near_plane, far_plane, step, largestep, scratch + 512 + 2 * map_bytes, d_count,
                           nullptr, s);
    if (e == hipSuccess)
        e = launch_render_shade(reinterpret_cast<uchar4 *>(d_rgba), d_vertex, d_normal, w * h, light, ambient, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    KFH_HIP(e);
    return KFH_OK;
}

}  // extern "C"
