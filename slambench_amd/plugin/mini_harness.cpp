// kfh_mini_slambench — a lean stand-in for the `slambench` loader, for machines that cannot build the real one.
//
// It replays the call order of the reference harness around a plugin
// (reference framework/shared/src/SLAMBenchConfiguration.cpp: AddSLAMLibrary :94-133, InitAlgorithms :171-192,
// ComputeLoopAlgorithm :230-342): dlopen, dlsym of the six mandatory c_sb_* symbols (exit 1 if one is
// missing), c_sb_new_slam_configuration before the plugin's own options are parsed, c_sb_init_slam_system, then
// per sensor frame c_sb_update_frame and — once it returns true — BeginFrame, c_sb_process_once,
// c_sb_update_outputs, EndFrame, one TSV row; finally c_sb_clean_slam_system.  Any `false` from the plugin ends
// the run with exit code 1, as in the reference.  Three sensor streams (RGB, Depth, Grey) are fed per time
// step, as the ICL-NUIM / TUM .slam files do, so the plugin's "not my sensor" path is exercised.
//
// Input, by magic number:
//  * a .slam file, SLAMBench's own dataset container (header: reference
//    framework/shared/src/io/serialisation/SLAMFileHeaderSerialiser.cpp:34-44, sensors: io/sensor/Sensor.cpp:77-97,
//    CameraSensor.cpp:66-79, DepthSensor.cpp:38-56, GroundTruthSensor.cpp:23-37, frames:
//    io/serialisation/SLAMFrameSerialiser.cpp:21-55).  Every sensor type the reference registers is read
//    (Camera, Depth, GroundTruthTrajectory, Accelerometer, Gyro, IMU, Odom, Lidar, PointCloud, EventCamera; the
//    last three have a u32 byte count in front of every frame, io/deserialisation/SLAMFileDeserialiser.cpp:56-79);
//    ground-truth trajectory frames are filtered out before the plugin, as the reference's frame loop does
//    (SLAMBenchConfiguration.cpp:252), all other frames are handed to it, which is what makes a depth-only
//    plugin say "not mine" to the accelerometer frames of a TUM file.
//
// -fl N counts processed frame-sets and -s N skips the first N frame-sets.  With --reference-frame-limit both
// count the way the reference does (SLAMBenchConfiguration.cpp:251-256, :322-324): its counter goes up by one for
// every frame taken from the file, of whatever sensor, and by one more for every processed frame-set; frames are
// withheld from the plugin while the counter is below 3 * N of -s, and the run ends when it reaches -fl.  With
// RGB, Depth and Grey streams "-fl 30" so processes 7 or 8 frame-sets; a run that is to be compared with one of the
// reference's logs needs the switch.
//  * a .kfd file (this repository's raw depth container, written by slambench_amd/harness.py):
//    char magic[4] = "KFD1"; uint32 width, height, frames; float fx, fy, cx, cy (normalised by width/height);
//    float depth_scale (raw unit -> metres); then per frame: double timestamp_s; uint16 depth[height][width].
#include <SLAMBenchLibraryHelper.h>
#include <io/SLAMFrame.h>
#include <io/sensor/CameraSensor.h>
#include <io/sensor/DepthSensor.h>
#include <outputs/Output.h>
#include <values/Value.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace sio = slambench::io;

#define LOAD_FUNC2HELPER(handle, lib, f)                                                        \
    *(void **)(&lib->f) = dlsym(handle, #f);                                                    \
    if (const char *err = dlerror()) {                                                          \
        std::fprintf(stderr, "Cannot load symbol '%s': %s\n", #f, err);                         \
        dlclose(handle);                                                                        \
        std::exit(1);                                                                           \
    }

// The option table, built the way the reference builds it (framework/shared/src/ParameterManager.cpp:102-141): first
// the harness's own options (framework/shared/src/SLAMBenchConfiguration.cpp:67-76,
// framework/tools/loaders/slambench_app.cpp:26-29), then the library's in the order in which it declared them, each
// under "<id>-<name>" when the library was loaded as file.so=id (framework/shared/include/Parameters.h:152-153).  A
// long or short name that is already taken ends the registration of that library's options there and then: what it
// declared before stays, what it declared after is unknown to the command line.  The kfusion use-case's "-s"
// (volume-size) meets the harness's "-s" (start-frame) that way.
struct OptionTable {
    std::map<std::string, Parameter *> long_options, short_options;     // harness-owned names map to nullptr
    OptionTable() {
        for (const char *n : {"frame-limit", "start-frame", "log-file", "input", "load-slam-library", "save-groundtruth", "dse",
                              "help", "realtime-mode", "realtime-multiplier", "file-output", "alignment-technique", "gui", "lifelong"})
            long_options[n] = nullptr;
        for (const char *n : {"fl", "s", "o", "i", "load", "sgt", "dse", "h", "realtime", "realtime-mult", "fo", "a", "gui", "ll"})
            short_options[n] = nullptr;
    }
    bool add_library(SLAMBenchLibraryHelper *lib, const std::string &prefix) {
        for (Parameter *p : lib->getParameters()) {
            if (!p->getLongOption().empty()) {
                const std::string name = prefix + p->getLongOption();
                if (long_options.count(name)) { std::cerr << "*** Duplicated long option replaced: '" << name << "'" << std::endl; return false; }
                long_options[name] = p;
            }
            if (!p->getShortOption().empty()) {
                const std::string name = prefix + p->getShortOption();
                if (short_options.count(name)) { std::cerr << "*** Duplicated short option replaced: '" << name << "'" << std::endl; return false; }
                short_options[name] = p;
            }
        }
        return true;
    }
};

static bool parse_plugin_options(const OptionTable &table, const std::vector<std::string> &args) {
    for (size_t i = 0; i < args.size(); ++i) {
        const std::string &a = args[i];
        Parameter *hit = nullptr;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            auto it = table.long_options.find(a.substr(2));
            if (it != table.long_options.end()) hit = it->second;
        } else if (a.size() > 1 && a[0] == '-') {
            auto it = table.short_options.find(a.substr(1));
            if (it != table.short_options.end()) hit = it->second;
        }
        if (!hit) { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return false; }
        if (hit->requiresValue()) {
            if (i + 1 >= args.size()) { std::fprintf(stderr, "option %s needs a value\n", a.c_str()); return false; }
            hit->setValue(args[++i].c_str());
        } else {
            hit->setValue(nullptr);
        }
    }
    return true;
}

int main(int argc, char **argv) {
    std::string input, library, library_id, log_file, dump_frames;
    int frame_limit = 0, start_frame = 0;
    bool dse = false, reference_counting = false;
    std::vector<std::string> plugin_args;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *what) { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(1); } return std::string(argv[++i]); };
        if (a == "-i" || a == "--input") input = need("-i");
        else if (a == "-load" || a == "--load-library" || a == "--load-slam-library") {
            // file.so or file.so=identifier (framework/shared/include/SLAMBenchConfiguration.h:180-187)
            library = need("-load");
            const size_t eq = library.find('=');
            if (eq != std::string::npos) { library_id = library.substr(eq + 1); library = library.substr(0, eq); }
        }
        else if (a == "-o" || a == "--log-file") log_file = need("-o");
        else if (a == "-fl" || a == "--frame-limit") frame_limit = std::atoi(need("-fl").c_str());
        else if (a == "-s" || a == "--start-frame") start_frame = std::atoi(need("-s").c_str());
        else if (a == "--reference-frame-limit") reference_counting = true;
        else if (a == "--dump-frames") dump_frames = need("--dump-frames");
        else if (a == "-dse" || a == "--dse") dse = true;
        else plugin_args.push_back(a);
    }
    if (input.empty() || library.empty()) {
        std::fprintf(stderr, "usage: %s -i stream.{slam,kfd} -load libkfusion-hip-library.so [-fl N] [-s N] [--reference-frame-limit] [-o log] [--dump-frames DIR] [--dse] [plugin options]\n", argv[0]);
        return 1;
    }

    // ---- input file -> sensors ----
    std::ifstream in(input, std::ios::binary);
    char magic[5] = {0, 0, 0, 0, 0};
    in.read(magic, 4);
    const bool is_slam = in && std::memcmp(magic, "SLAM", 4) == 0;
    const bool is_kfd = in && std::memcmp(magic, "KFD1", 4) == 0;
    if (!is_slam && !is_kfd) { std::fprintf(stderr, "%s is neither a .slam nor a KFD1 file\n", input.c_str()); return 1; }

    sio::InputInterface interface;
    std::vector<sio::Sensor *> file_sensors;       // by file index (.slam)
    std::vector<size_t> frame_sizes;                // payload bytes; 0 with counted[] set: the count in front decides
    std::vector<bool> counted;                      // a u32 byte count precedes every frame (variable-size sensors)
    uint32_t w = 0, h = 0, n = 0;
    float intr[4] = {0, 0, 0, 0}, depth_scale = 0.f;
    sio::CameraSensor *rgb = nullptr, *grey = nullptr;
    sio::DepthSensor *depth = nullptr;
    if (is_kfd) {
        in.read((char *)&w, 4); in.read((char *)&h, 4); in.read((char *)&n, 4);
        in.read((char *)intr, 16); in.read((char *)&depth_scale, 4);
        if (!in) { std::fprintf(stderr, "%s: truncated header\n", input.c_str()); return 1; }
        rgb = new sio::CameraSensor("RGB");
        rgb->Width = w; rgb->Height = h; rgb->PixelFormat = sio::pixelformat::RGB_III_888;
        depth = new sio::DepthSensor("Depth");
        depth->Width = w; depth->Height = h; depth->PixelFormat = sio::pixelformat::D_I_16;
        depth->DisparityParams[0] = depth_scale; depth->DisparityParams[1] = 0.f;
        grey = new sio::CameraSensor("Grey");
        grey->Width = w; grey->Height = h; grey->PixelFormat = sio::pixelformat::G_I_8;
        for (sio::CameraSensor *s : {rgb, (sio::CameraSensor *)depth, grey}) {
            std::memcpy(s->Intrinsics, intr, sizeof(intr));
            interface.GetSensors().AddSensor(s);
        }
    } else {
        auto rd = [&](void *dst, size_t bytes) { in.read((char *)dst, (std::streamsize)bytes); return (bool)in; };
        char nul;
        uint32_t version = 0, count = 0;
        if (!rd(&nul, 1) || nul != 0 || !rd(&version, 4) || !rd(&count, 4) || version != 1) {
            std::fprintf(stderr, "%s: bad .slam header (version %u)\n", input.c_str(), version);
            return 1;
        }
        for (uint32_t i = 0; i < count; ++i) {
            uint8_t len = 0, index = 0;
            uint32_t desc_len = 0;
            float rate = 0.f, pose[16], delay = 0.f;
            if (!rd(&len, 1)) return 1;
            std::string name(len, '\0');
            rd(&name[0], len);
            name = name.c_str();
            if (!rd(&len, 1)) return 1;
            std::string type(len, '\0');
            rd(&type[0], len);
            type = type.c_str();
            rd(&index, 1);
            rd(&desc_len, 4);
            std::string desc(desc_len, '\0');
            rd(&desc[0], desc_len);
            rd(&rate, 4); rd(pose, 64); rd(&delay, 4);
            if (!in) { std::fprintf(stderr, "%s: truncated sensor block\n", input.c_str()); return 1; }
            sio::Sensor *sensor = nullptr;
            size_t frame_size = 0;
            bool counted_now = false;
            if (type == "Camera" || type == "Depth") {
                int32_t frame_format = 0, pixel_format = 0;
                uint32_t sw = 0, sh = 0;
                float k[4];
                rd(&frame_format, 4); rd(&pixel_format, 4); rd(&sw, 4); rd(&sh, 4); rd(k, 16);
                sio::CameraSensor *cam;
                if (type == "Depth") {
                    auto *d = new sio::DepthSensor(name);
                    int32_t disparity_type = 0, distortion_type = 0;
                    float distortion[5];
                    rd(d->DisparityParams, 8); rd(&disparity_type, 4); rd(&distortion_type, 4); rd(distortion, 20);
                    cam = d;
                } else {
                    cam = new sio::CameraSensor(name);
                    int32_t distortion_type = 0;
                    float distortion[15];
                    rd(&distortion_type, 4); rd(distortion, 60);
                }
                cam->Width = sw; cam->Height = sh;
                cam->PixelFormat = (sio::pixelformat::EPixelFormat)pixel_format;
                std::memcpy(cam->Intrinsics, k, sizeof(k));
                cam->Rate = rate;
                frame_size = cam->GetFrameSize();
                sensor = cam;
                interface.GetSensors().AddSensor(cam);
            } else {
                // the other sensor types: bytes of their own block, payload bytes of a frame (0 = the count in
                // front of the frame), whether such a count is there (the reference's "variable size" flag)
                static const struct { const char *type; size_t block, frame; bool has_count; } kinds[] = {
                    {"GroundTruthTrajectory", 0, 64, false},   // GroundTruthSensor.cpp:21-24
                    {"Accelerometer", 72, 12, false},          // AccelerometerSensor.cpp:23-35
                    {"Gyro", 72, 12, false},                   // GyroSensor.cpp:22-34
                    {"IMU", 32, 24, false},                    // IMUSensor.cpp:22-44
                    {"Odom", 0, 52, false},                    // OdomSensor.cpp:19-31
                    {"Lidar", 36, 0, true},                    // LidarSensor.cpp:25-53
                    {"PointCloud", 0, 0, true},                // PointCloudSensor.cpp:21-31
                    {"EventCamera", 8, 0, true},               // EventCameraSensor.cpp:25-42 (frame size below)
                };
                bool known = false;
                for (const auto &k : kinds) {
                    if (type != k.type) continue;
                    known = true;
                    unsigned char block[72];
                    rd(block, k.block);
                    frame_size = k.frame;
                    counted_now = k.has_count;
                    if (type == "EventCamera") {               // x, y, on/off; coordinates as narrow as the sensor allows
                        int32_t ew = 0, eh = 0;
                        std::memcpy(&ew, block, 4); std::memcpy(&eh, block + 4, 4);
                        const size_t coord = (ew <= 0xFF && eh <= 0xFF) ? 1 : (ew <= 0xFFFF && eh <= 0xFFFF) ? 2 : 4;
                        frame_size = 2 * coord + 1;
                    }
                }
                if (!known) {
                    std::fprintf(stderr, "%s: sensor '%s' has type '%s', which the reference does not register\n",
                                 input.c_str(), name.c_str(), type.c_str());
                    return 1;
                }
                sensor = new sio::Sensor(name, type);
                sensor->Rate = rate;
                if (type != "GroundTruthTrajectory") interface.GetSensors().AddSensor(sensor);
            }
            if (!in) { std::fprintf(stderr, "%s: truncated sensor block\n", input.c_str()); return 1; }
            file_sensors.push_back(sensor);
            frame_sizes.push_back(frame_size);
            counted.push_back(counted_now);
        }
    }

    // ---- AddSLAMLibrary ----
    void *handle = dlopen(library.c_str(), RTLD_LAZY);
    if (!handle) { std::fprintf(stderr, "Cannot open library: %s\n", dlerror()); return 1; }
    dlerror();
    size_t start = library.find_last_of('/') == std::string::npos ? 0 : library.find_last_of('/') + 1;
    std::string libname = library.substr(start);
    if (libname.compare(0, 3, "lib") == 0) libname = libname.substr(3);
    libname = libname.substr(0, libname.find('-'));
    auto *lib = new SLAMBenchLibraryHelper("", libname, std::cerr, &interface);
    LOAD_FUNC2HELPER(handle, lib, c_sb_init_slam_system);
    LOAD_FUNC2HELPER(handle, lib, c_sb_new_slam_configuration);
    LOAD_FUNC2HELPER(handle, lib, c_sb_update_frame);
    LOAD_FUNC2HELPER(handle, lib, c_sb_process_once);
    LOAD_FUNC2HELPER(handle, lib, c_sb_clean_slam_system);
    LOAD_FUNC2HELPER(handle, lib, c_sb_update_outputs);
    if (dlsym(handle, "_Z13sb_relocalizeP22SLAMBenchLibraryHelper")) {
        *(void **)(&lib->c_sb_relocalize) = dlsym(handle, "c_sb_relocalize");
    } else {
        lib->c_sb_relocalize = lib->c_sb_process_once;
    }
    dlerror();
    if (!lib->c_sb_new_slam_configuration(lib)) { std::fprintf(stderr, "Configuration construction failed.\n"); return 1; }
    const std::string option_prefix = library_id.empty() ? "" : library_id + "-";
    OptionTable options;
    options.add_library(lib, option_prefix);
    if (!parse_plugin_options(options, plugin_args)) return 1;
    if (dse) {                          // SLAMBenchConfiguration::PrintDse: the parameter space of the loaded library
        std::cout << "libs:" << (library_id.empty() ? lib->GetIdentifier() : library_id) << "\n";
        for (Parameter *p : lib->getParameters())
            std::cout << "argument:" << option_prefix << p->getLongOption() << "\n" << p->getStrDetails() << "\n";
        return 0;
    }

    std::ostream *log = &std::cout;
    std::ofstream log_stream;
    if (!log_file.empty()) { log_stream.open(log_file); log = &log_stream; }
    *log << "Properties:\n=================\n\n";
    *log << "frame-limit: " << frame_limit << "\ninput: " << input << "\nload-library: " << library << "\n";
    for (sio::Sensor *s : interface.GetSensors()) {
        if (s->GetType() != "Depth") continue;
        auto *d = static_cast<sio::DepthSensor *>(s);
        *log << d->getName() << "-intrinsics-parameters: " << d->Intrinsics[0] << "," << d->Intrinsics[1] << ","
             << d->Intrinsics[2] << "," << d->Intrinsics[3] << "\n";
        *log << d->getName() << "-disparity-params: " << d->DisparityParams[0] << "," << d->DisparityParams[1] << "\n";
    }
    for (Parameter *p : lib->getParameters()) *log << option_prefix << p->getLongOption() << ": " << p->getStrValue() << "\n";

    // ---- InitAlgorithms ----
    if (!lib->c_sb_init_slam_system(lib)) { std::fprintf(stderr, "Algorithm initialization failed.\n"); return 1; }
    auto *trajectory = lib->GetOutputManager().GetMainOutput(slambench::values::VT_POSE);
    if (!trajectory) { std::fprintf(stderr, "Algo does not provide a main pose output\n"); return 1; }
    trajectory->SetActive(true);
    // what the GUI does when a picture is ticked: activate the frame outputs; their last values are written at the end
    if (!dump_frames.empty())
        for (auto &kv : lib->GetOutputManager())
            if (kv.second->GetType() == slambench::values::VT_FRAME) kv.second->SetActive(true);

    *log << "\nStatistics:\n=================\n\nFrame Number\tTimestamp\tDuration_Frame";
    for (auto *p : lib->GetMetricManager().GetPhases()) *log << "\tDuration_" << p->GetName();
    *log << "\tX\tY\tZ\n";
    *log << std::fixed << std::setprecision(10);

    // ---- ComputeLoopAlgorithm ----
    unsigned processed = 0;
    auto feed = [&](sio::SLAMFrame *fr, const slambench::TimeStamp &ts) -> int {
        const bool ongoing = !lib->c_sb_update_frame(lib, fr);
        if (ongoing) return 0;
        lib->GetMetricManager().BeginFrame();
        if (!lib->c_sb_process_once(lib)) { std::fprintf(stderr, "Error after lib->c_sb_process_once.\n"); return -1; }
        slambench::TimeStamp out_ts = ts;
        if (!lib->c_sb_update_outputs(lib, &out_ts)) { std::fprintf(stderr, "Failed to get outputs\n"); return -1; }
        lib->GetMetricManager().EndFrame();
        ++processed;
        auto *out = static_cast<slambench::outputs::Output *>(trajectory);
        const auto &pose = static_cast<const slambench::values::PoseValue *>(out->GetMostRecentValue().second)->GetValue();
        *log << processed << "\t" << ts.ToS() << "\t" << lib->GetMetricManager().LastFrameDuration();
        for (auto *p : lib->GetMetricManager().GetPhases())
            *log << "\t" << lib->GetMetricManager().LastPhaseDurations().at(p->GetName());
        *log << "\t" << pose(0, 3) << "\t" << pose(1, 3) << "\t" << pose(2, 3) << "\n";
        return 1;
    };
    if (is_kfd) {
        std::vector<uint16_t> depth_buf((size_t)w * h);
        std::vector<unsigned char> dummy(4, 0);
        for (uint32_t f = 0; f < n; ++f) {
            double ts_s = 0.0;
            in.read((char *)&ts_s, 8);
            in.read((char *)depth_buf.data(), depth_buf.size() * 2);
            if (!in) { std::fprintf(stderr, "truncated input at frame %u\n", f); return 1; }
            const slambench::TimeStamp ts = slambench::TimeStamp::get((uint32_t)ts_s, (uint32_t)((ts_s - (uint32_t)ts_s) * 1e9 + 0.5));
            sio::SLAMInMemoryFrame frames[3];
            frames[0].FrameSensor = rgb;   frames[0].Data = dummy.data();     frames[0].Size = 0;
            frames[1].FrameSensor = depth; frames[1].Data = depth_buf.data(); frames[1].Size = depth_buf.size() * 2;
            frames[2].FrameSensor = grey;  frames[2].Data = dummy.data();     frames[2].Size = 0;
            for (auto &fr : frames) {
                fr.Timestamp = ts;
                if (feed(&fr, ts) < 0) return 1;
            }
            if (frame_limit > 0 && processed >= (unsigned)frame_limit) break;
        }
    } else {
        std::vector<unsigned char> payload;
        unsigned frame_count = 0;                 // the reference's counter (SLAMBenchConfiguration.cpp:239)
        unsigned sets_seen = 0;                   // frame-sets completed or skipped (the plain counting)
        while (true) {
            struct __attribute__((packed)) { uint32_t s, ns; uint8_t index; } head;
            in.read((char *)&head, sizeof(head));
            if (in.gcount() == 0) break;                      // clean end of file
            if (!in || head.index >= file_sensors.size()) { std::fprintf(stderr, "bad frame header\n"); return 1; }
            size_t size = frame_sizes[head.index];
            if (counted[head.index]) {
                uint32_t bytes = 0;
                in.read((char *)&bytes, 4);
                if (size == 0) size = bytes;
            }
            payload.resize(size);
            in.read((char *)payload.data(), (std::streamsize)payload.size());
            if (!in) { std::fprintf(stderr, "truncated frame payload\n"); return 1; }
            sio::Sensor *sensor = file_sensors[head.index];
            ++frame_count;
            if (sensor->GetType() == "GroundTruthTrajectory") continue;     // kept from the algorithms
            const slambench::TimeStamp ts = slambench::TimeStamp::get(head.s, head.ns);
            sio::SLAMInMemoryFrame fr;
            fr.FrameSensor = sensor;
            fr.Data = payload.data();
            fr.Size = payload.size();
            fr.Timestamp = ts;
            if (reference_counting) {
                if (frame_count < (unsigned)start_frame * 3) continue;
                const int fed = feed(&fr, ts);
                if (fed < 0) return 1;
                if (fed > 0) {
                    ++frame_count;
                    if (frame_limit > 0 && frame_count >= (unsigned)frame_limit) break;
                }
            } else {
                if (sets_seen < (unsigned)start_frame) {           // -s: whole frame-sets are skipped, told by their depth frame
                    if (sensor->GetType() == "Depth") ++sets_seen;
                    continue;
                }
                if (feed(&fr, ts) < 0) return 1;
                if (frame_limit > 0 && processed >= (unsigned)frame_limit) break;
            }
        }
    }
    if (!dump_frames.empty()) {
        for (auto &kv : lib->GetOutputManager()) {
            if (kv.second->GetType() != slambench::values::VT_FRAME) continue;
            auto *out = static_cast<slambench::outputs::Output *>(kv.second);
            if (out->Empty()) continue;
            const auto *fv = static_cast<const slambench::values::FrameValue *>(out->GetMostRecentValue().second);
            std::string name = kv.first;
            for (char &ch : name) if (ch == ' ') ch = '_';
            // raw RGBA rows, preceded by "width height\n"
            std::ofstream f(dump_frames + "/" + name + ".rgba", std::ios::binary);
            f << fv->GetWidth() << " " << fv->GetHeight() << "\n";
            f.write((const char *)fv->GetData(), (std::streamsize)fv->GetWidth() * fv->GetHeight() * 4);
        }
    }
    if (!lib->c_sb_clean_slam_system()) { std::fprintf(stderr, "Algorithm cleaning failed.\n"); return 1; }
    delete lib;
    return 0;
}
