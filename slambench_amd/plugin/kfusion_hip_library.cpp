// SLAMBench plugin adapter of the gfx950 KinectFusion core: libkfusion-hip-library.so.
//
// Implements the plugin entry points the harness resolves with dlsym
// (reference framework/shared/include/SLAMBenchAPI.h:30-54; loader
// framework/shared/src/SLAMBenchConfiguration.cpp:107-124) on top of the C ABI in include/kfusion_hip.h.
// Only SLAMBench's own public headers are used, so the same source compiles
//   - against the real SLAMBench tree (+Eigen), giving the drop-in library, and
//   - against slambench_amd/plugin/sb_lite, giving the library the in-repo mini-harness loads.
// Parameter names, defaults and short options are those of the kfusion use-case as printed in the reference's
// logs (icra2018_results/1080/violons_libkfusion-cpp_living_room_traj0_loop.log:20-31); phase names are the
// Duration_* columns of the same log (:36).
#include <limits>      // SLAMBench's Parameters.h uses std::numeric_limits without including it (an error from GCC 11 on)

#include <SLAMBenchAPI.h>
#include <io/SLAMFrame.h>
#include <io/sensor/CameraSensor.h>
#include <io/sensor/CameraSensorFinder.h>
#include <io/sensor/DepthSensor.h>
#include <metrics/MetricManager.h>
#include <metrics/Phase.h>
#include <outputs/Output.h>
#include <outputs/OutputManager.h>
#include <values/Value.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/kfusion_hip.h"

namespace {

// ---- parameters (storage must outlive the run: the harness keeps pointers to it) ----
const int default_compute_size_ratio = 1, default_integration_rate = 2, default_tracking_rate = 1,
          default_rendering_rate = 4;
const float default_icp_threshold = 1e-5f, default_mu = 0.1f;
const sb_float3 default_volume_size = {8.f, 8.f, 8.f};
const sb_float3 default_volume_direction = {4.f, 4.f, 4.f};
const sb_uint3 default_volume_resolution = {256, 256, 256};
const int default_pyramid[3] = {10, 5, 4};
const int default_devices = 1;

int compute_size_ratio, integration_rate, tracking_rate, rendering_rate;
float icp_threshold, mu;
sb_float3 volume_size, volume_direction;
sb_uint3 volume_resolution;
int pyramid_level1, pyramid_level2, pyramid_level3;
int devices;           // not a parameter of the kfusion use-case: GPUs the volume is split over (z-slabs, kfh_group)

// ---- run state (sb_clean_slam_system takes no argument, so the state is file-scope like upstream's) ----
kfh_context *ctx = nullptr;       // the context; with --devices N > 1 the first member of `group`
kfh_group *group = nullptr;
slambench::io::DepthSensor *depth_sensor = nullptr;
slambench::outputs::Output *pose_output = nullptr;
slambench::outputs::Output *pointcloud_output = nullptr;
// the three pictures of the kfusion GUI (renderDepthKernel, renderTrackKernel, renderVolumeKernel)
slambench::outputs::Output *depth_frame_output = nullptr, *track_frame_output = nullptr, *render_frame_output = nullptr;
std::vector<uint8_t> depth_image, track_image, volume_image;   // RGBA of the compute size, filled in the Render phase
bool depth_image_fresh = false, track_image_fresh = false, volume_image_fresh = false;
bool track_codes_on = false;
bool render_always = false;    // KFH_RENDER_ALWAYS: render like the reference, whether or not anybody looks
bool log_schedule = false;     // KFH_LOG_SCHEDULE: one line per frame on stderr with what the frame did (tracked,
                               //   integrated, raycasted), for comparisons with the phase columns of a reference log
int image_w = 0, image_h = 0;
size_t frame_bytes = 0;
float frame_pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // pose after the last sb_process_once
bool depth_ready = false;
int frame_counter = 0;
const char *const phase_names[5] = {"Preprocessing", "Tracking", "Integration", "Raycasting", "Render"};

bool report(const char *what, int rc) {
    if (rc == KFH_OK) return true;
    std::fprintf(stderr, "kfusion-hip: %s failed: %s\n", what, kfh_last_error());
    return false;
}

}  // namespace

bool sb_new_slam_configuration(SLAMBenchLibraryHelper *slam_settings) {
    slam_settings->addParameter(TypedParameter<int>("c", "compute-size-ratio", "Compute ratio", &compute_size_ratio, &default_compute_size_ratio));
    slam_settings->addParameter(TypedParameter<int>("r", "integration-rate", "integration-rate", &integration_rate, &default_integration_rate));
    slam_settings->addParameter(TypedParameter<int>("t", "tracking-rate", "tracking-rate", &tracking_rate, &default_tracking_rate));
    slam_settings->addParameter(TypedParameter<int>("z", "rendering-rate", "rendering-rate", &rendering_rate, &default_rendering_rate));
    slam_settings->addParameter(TypedParameter<float>("l", "icp-threshold", "icp-threshold", &icp_threshold, &default_icp_threshold));
    slam_settings->addParameter(TypedParameter<float>("m", "mu", "mu", &mu, &default_mu));
    slam_settings->addParameter(TypedParameter<sb_float3>("s", "volume-size", "volume-size", &volume_size, &default_volume_size));
    slam_settings->addParameter(TypedParameter<sb_float3>("d", "volume-direction", "volume-direction", &volume_direction, &default_volume_direction));
    slam_settings->addParameter(TypedParameter<sb_uint3>("v", "volume-resolution", "volume-resolution", &volume_resolution, &default_volume_resolution));
    slam_settings->addParameter(TypedParameter<int>("y1", "pyramid-level1", "pyramid-level1", &pyramid_level1, &default_pyramid[0]));
    slam_settings->addParameter(TypedParameter<int>("y2", "pyramid-level2", "pyramid-level2", &pyramid_level2, &default_pyramid[1]));
    slam_settings->addParameter(TypedParameter<int>("y3", "pyramid-level3", "pyramid-level3", &pyramid_level3, &default_pyramid[2]));
    // One harness process drives one plugin (reference framework/shared/src/SLAMBenchConfiguration.cpp:230-342), so a
    // volume too large or too slow for one GPU is split inside the plugin: N z-slabs on the first N visible devices
    // (more slabs than devices wrap around).
    slam_settings->addParameter(TypedParameter<int>("n", "devices", "GPUs the volume is split over (z-slabs)", &devices, &default_devices));
    return true;
}

bool sb_init_slam_system(SLAMBenchLibraryHelper *slam_settings) {
    // The depth camera: pixel format D_I_16 (reference framework/shared/src/io/PixelFormat.cpp:73).
    slambench::io::CameraSensorFinder sensor_finder;
    depth_sensor = static_cast<slambench::io::DepthSensor *>(
        sensor_finder.FindOne(slam_settings->get_sensors(), {{"camera_type", "depth"}}));
    if (depth_sensor == nullptr) {
        std::fprintf(stderr, "kfusion-hip: no depth sensor in the input\n");
        return false;
    }
    if (depth_sensor->PixelFormat != slambench::io::pixelformat::D_I_16) {
        std::fprintf(stderr, "kfusion-hip: depth sensor must be 16-bit integer (D_I_16)\n");
        return false;
    }

    kfh_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.input_width = (int32_t)depth_sensor->Width;
    cfg.input_height = (int32_t)depth_sensor->Height;
    cfg.compute_size_ratio = compute_size_ratio;
    // intrinsics are stored normalised by the image size (reference framework/tools/dataset-tools/ICLNUIM.cpp:43-47)
    cfg.fx = depth_sensor->Intrinsics[0] * (float)depth_sensor->Width;
    cfg.fy = depth_sensor->Intrinsics[1] * (float)depth_sensor->Height;
    cfg.cx = depth_sensor->Intrinsics[2] * (float)depth_sensor->Width;
    cfg.cy = depth_sensor->Intrinsics[3] * (float)depth_sensor->Height;
    // raw unit -> metres comes from the sensor (DisparityParams[0], e.g. 0.001 for millimetres); the kernel
    // divides by the unit count per metre so that millimetre input reproduces upstream's `/ 1000.0f` bit for bit
    const float scale = depth_sensor->DisparityParams[0];
    cfg.depth_divisor = scale > 0.f ? std::round(1.0f / scale) : 1000.0f;
    cfg.vol_res[0] = (int32_t)volume_resolution.x; cfg.vol_res[1] = (int32_t)volume_resolution.y; cfg.vol_res[2] = (int32_t)volume_resolution.z;
    cfg.vol_dim[0] = volume_size.x; cfg.vol_dim[1] = volume_size.y; cfg.vol_dim[2] = volume_size.z;
    cfg.init_pos[0] = volume_direction.x; cfg.init_pos[1] = volume_direction.y; cfg.init_pos[2] = volume_direction.z;
    cfg.mu = mu;
    cfg.icp_threshold = icp_threshold;
    cfg.pyramid[0] = pyramid_level1; cfg.pyramid[1] = pyramid_level2; cfg.pyramid[2] = pyramid_level3;
    cfg.integration_rate = integration_rate;
    cfg.tracking_rate = tracking_rate;
    cfg.integrate_filtered = 0;
    cfg.slab_z0 = 0;
    cfg.slab_z1 = cfg.vol_res[2];
    if (devices > 1) {
        int visible = 0;
        if (!report("kfh_device_count", kfh_device_count(&visible)) || visible < 1) return false;
        std::vector<int> list((size_t)devices);
        for (int i = 0; i < devices; ++i) list[(size_t)i] = i % visible;
        if (!report("kfh_group_create", kfh_group_create(&cfg, list.data(), devices, &group))) return false;
        if (!report("kfh_group_member", kfh_group_member(group, 0, &ctx))) return false;
    } else if (!report("kfh_create", kfh_create(&cfg, &ctx))) {
        return false;
    }

    frame_bytes = (size_t)cfg.input_width * cfg.input_height * sizeof(uint16_t);
    depth_ready = false;
    frame_counter = 0;

    // Every declared phase must be bracketed in every frame: the harness reads each phase's duration at
    // EndFrame (reference framework/shared/src/metrics/MetricManager.cpp:104-108).
    for (const char *name : phase_names) slam_settings->GetMetricManager().AddPhase(name);

    pose_output = new slambench::outputs::Output("Pose", slambench::values::VT_POSE, true);
    slam_settings->GetOutputManager().RegisterOutput(pose_output);
    pointcloud_output = new slambench::outputs::Output("PointCloud", slambench::values::VT_POINTCLOUD, true);
    pointcloud_output->SetKeepOnlyMostRecent(true);
    slam_settings->GetOutputManager().RegisterOutput(pointcloud_output);

    if (!report("kfh_compute_size", kfh_compute_size(ctx, &image_w, &image_h))) return false;
    const size_t image_bytes = (size_t)image_w * image_h * 4;
    depth_image.assign(image_bytes, 0);
    track_image.assign(image_bytes, 0);
    volume_image.assign(image_bytes, 0);
    depth_image_fresh = track_image_fresh = volume_image_fresh = false;
    track_codes_on = false;
    render_always = std::getenv("KFH_RENDER_ALWAYS") != nullptr;
    log_schedule = std::getenv("KFH_LOG_SCHEDULE") != nullptr;
    slambench::outputs::Output **frame_outputs[3] = {&depth_frame_output, &track_frame_output, &render_frame_output};
    const char *const frame_names[3] = {"Depth Frame", "Tracking Frame", "Rendered frame"};
    for (int i = 0; i < 3; ++i) {
        *frame_outputs[i] = new slambench::outputs::Output(frame_names[i], slambench::values::VT_FRAME);
        (*frame_outputs[i])->SetKeepOnlyMostRecent(true);
        slam_settings->GetOutputManager().RegisterOutput(*frame_outputs[i]);
    }
    return true;
}

bool sb_update_frame(SLAMBenchLibraryHelper *, slambench::io::SLAMFrame *s) {
    if (s->FrameSensor != depth_sensor) return false; // RGB / grey / ground-truth frames are not used
    if (s->GetSize() != frame_bytes) return false;
    // The harness reuses one buffer for all frames, so the bytes are copied out before FreeData(): straight into
    // the core's pinned staging slot, from where the transfer to the device is queued at once and runs while the
    // harness is still between sb_update_frame and sb_process_once.
    const uint16_t *pixels = static_cast<const uint16_t *>(s->GetData());
    const bool ok = group ? report("kfh_group_upload_depth", kfh_group_upload_depth(group, pixels))
                          : report("kfh_upload_depth", kfh_upload_depth(ctx, pixels));
    s->FreeData();
    depth_ready = ok;
    return ok;
}

bool sb_process_once(SLAMBenchLibraryHelper *slam_settings) {
    auto &mm = slam_settings->GetMetricManager();
    slambench::metrics::Phase *phase[5];
    for (int i = 0; i < 5; ++i) phase[i] = mm.GetPhase(phase_names[i]);
    bool ok = true;

    if (group) {
        // the stages of all members are queued by one call (it also ends the members' frame); the phases are still
        // bracketed, as the harness reads every declared phase in every frame
        phase[0]->Begin();
        ok = ok && report("kfh_group_process_frame", kfh_group_process_frame(group));
        phase[0]->End();
        phase[1]->Begin(); phase[1]->End();
        phase[2]->Begin(); phase[2]->End();
        phase[3]->Begin();
    } else {
    phase[0]->Begin();
    ok = ok && report("kfh_preprocess", kfh_preprocess(ctx));
    phase[0]->End();
    phase[1]->Begin();
    ok = ok && report("kfh_track", kfh_track(ctx));
    phase[1]->End();
    phase[2]->Begin();
    ok = ok && report("kfh_integrate", kfh_integrate(ctx));
    phase[2]->End();
    phase[3]->Begin();
    ok = ok && report("kfh_raycast", kfh_raycast(ctx));
    }
    // The phases above only queue work; the frame's device time is accounted here, once, so that
    // Duration_Frame is honest (the reference's GPU builds show near-zero phases for the same reason:
    // host timers without a device synchronisation, framework/shared/src/metrics/DurationMetric.cpp:65-71).
    // Waiting for the pose waits for everything queued so far; the second pass of the raycast, which only the
    // next frame's tracking needs, stays postponed and shares the next frame's preprocessing launch.
    ok = ok && (group ? report("kfh_group_get_pose", kfh_group_get_pose(group, frame_pose))
                      : report("kfh_get_pose", kfh_get_pose(ctx, frame_pose)));
    phase[3]->End();
    // Render: the depth and tracking pictures every frame, the volume picture every rendering-rate frames, as
    // the Duration_Render column of the reference's logs shows (two levels, the high one on rows 1, 5, 9, ...).
    // Unlike the reference, a picture nobody has activated is not drawn (KFH_RENDER_ALWAYS=1 draws all three).
    phase[4]->Begin();
    {
        const bool want_depth = render_always || depth_frame_output->IsActive();
        const bool want_track = render_always || track_frame_output->IsActive();
        // (the volume picture needs the whole volume on one device: not drawn for a split volume)
        const bool want_volume = !group && (render_always || render_frame_output->IsActive()) && rendering_rate > 0 &&
                                 (frame_counter % rendering_rate) == 0;
        if (want_depth) {
            ok = ok && report("kfh_render(depth)", kfh_render(ctx, KFH_RENDER_DEPTH, nullptr, depth_image.data(), depth_image.size()));
            depth_image_fresh = ok;
        }
        if (want_track && track_codes_on) {
            ok = ok && report("kfh_render(track)", kfh_render(ctx, KFH_RENDER_TRACK, nullptr, track_image.data(), track_image.size()));
            track_image_fresh = ok;
        }
        if (want_track != track_codes_on) {          // tracking stores its result codes from the next frame on
            ok = ok && report("kfh_enable_track_render", kfh_enable_track_render(ctx, want_track ? 1 : 0));
            track_codes_on = want_track;
        }
        if (want_volume) {
            ok = ok && report("kfh_render(volume)", kfh_render(ctx, KFH_RENDER_VOLUME, nullptr, volume_image.data(), volume_image.size()));
            volume_image_fresh = ok;
        }
    }
    phase[4]->End();
    if (!group) ok = ok && report("kfh_end_frame", kfh_end_frame(ctx));
    if (log_schedule && ok) {
        kfh_frame_status st;
        ok = report("kfh_get_status", kfh_get_status(ctx, frame_counter, &st));
        if (ok) std::fprintf(stderr, "kfusion-hip: schedule frame %d tracked %d integrated %d raycasted %d\n", frame_counter,
                             st.tracked, st.integrated, st.raycasted);
    }
    depth_ready = false;
    frame_counter++;
    return ok;
}

bool sb_update_outputs(SLAMBenchLibraryHelper *lib, const slambench::TimeStamp *ts_p) {
    slambench::TimeStamp ts = *ts_p;
    std::lock_guard<FastLock> lock(lib->GetOutputManager().GetLock());
    if (pose_output->IsActive()) {
        Eigen::Matrix4f mat;                       // the pose sb_process_once waited for
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) mat(r, c) = frame_pose[r * 4 + c];
        pose_output->AddPoint(ts, new slambench::values::PoseValue(mat));
    }
    if (pointcloud_output->IsActive() && rendering_rate > 0 && (frame_counter % rendering_rate) == 0) {
        // the model vertex map of the last raycast, in the volume frame
        int w = 0, h = 0;
        if (!report("kfh_compute_size", kfh_compute_size(ctx, &w, &h))) return false;
        std::vector<float> vertex((size_t)w * h * 4);
        if (!report("kfh_download", kfh_download(ctx, KFH_BUF_MODEL_VERTEX, 0, vertex.data(), vertex.size() * sizeof(float))))
            return false;
        auto *cloud = new slambench::values::PointCloudValue();
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            const float *v = &vertex[i * 4];
            if (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f) cloud->AddPoint({v[0], v[1], v[2]});
        }
        pointcloud_output->AddPoint(ts, cloud);
    }
    struct { slambench::outputs::Output *out; std::vector<uint8_t> *image; bool *fresh; } pictures[3] = {
        {depth_frame_output, &depth_image, &depth_image_fresh},
        {track_frame_output, &track_image, &track_image_fresh},
        {render_frame_output, &volume_image, &volume_image_fresh}};
    for (auto &p : pictures) {
        if (!p.out->IsActive() || !*p.fresh) continue;
        p.out->AddPoint(ts, new slambench::values::FrameValue((uint32_t)image_w, (uint32_t)image_h,
                                                                slambench::io::pixelformat::RGBA_IIII_8888, p.image->data()));
        *p.fresh = false;
    }
    return true;
}

bool sb_clean_slam_system() {
    const bool ok = group ? report("kfh_group_destroy", kfh_group_destroy(group)) : report("kfh_destroy", kfh_destroy(ctx));
    ctx = nullptr;
    group = nullptr;
    depth_sensor = nullptr;
    return ok;
}

// The harness looks up these C names.  A plugin linked inside the SLAMBench tree gets them from the
// slambench-c-wrapper static library (reference framework/shared/src/library_wrapper.cpp:15-44); define
// KFH_USE_SLAMBENCH_C_WRAPPER in that case.  c_sb_relocalize is deliberately absent: without the C++ symbol
// sb_relocalize the harness aliases relocalisation to c_sb_process_once
// (reference framework/shared/src/SLAMBenchConfiguration.cpp:114-119).
#ifndef KFH_USE_SLAMBENCH_C_WRAPPER
extern "C" {
bool c_sb_new_slam_configuration(void *slam_settings) { return sb_new_slam_configuration((SLAMBenchLibraryHelper *)slam_settings); }
bool c_sb_init_slam_system(void *slam_settings) { return sb_init_slam_system((SLAMBenchLibraryHelper *)slam_settings); }
bool c_sb_update_frame(void *slam_settings, void *type) {
    return sb_update_frame((SLAMBenchLibraryHelper *)slam_settings, (slambench::io::SLAMFrame *)type);
}
bool c_sb_process_once(void *slam_settings) { return sb_process_once((SLAMBenchLibraryHelper *)slam_settings); }
bool c_sb_clean_slam_system() { return sb_clean_slam_system(); }
bool c_sb_update_outputs(void *lib, void *timestamp) {
    return sb_update_outputs((SLAMBenchLibraryHelper *)lib, (const slambench::TimeStamp *)timestamp);
}
}
#endif
