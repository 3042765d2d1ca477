// sb_lite — the slice of the SLAMBench harness interface that a depth-only SLAM plugin touches,
// re-declared header-only so that the plugin adapter and the mini-harness of this repository build and
// run on a machine that has neither the SLAMBench tree nor Eigen / Boost / PCL.
//
// It is NOT binary compatible with a real `slambench` executable: the harness passes C++ objects
// (vtables, std::string, std::map, Eigen matrices) across the plugin boundary, so a drop-in
// libkfusion-hip-library.so must be compiled against the real headers (plugin/Makefile: SLAMBENCH_INCLUDE=...).  Names,
// members and behaviour follow the reference headers cited at each class
// (paths relative to framework/shared/include of the reference).
#ifndef SB_LITE_ALL_H_
#define SB_LITE_ALL_H_

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <utility>
#include <vector>

// ---------------------------------------------------------------------------------- Eigen (column-major 4x4 only)
namespace Eigen {
struct Matrix4f {
    float data_[16];
    Matrix4f() { std::memset(data_, 0, sizeof(data_)); }
    static Matrix4f Identity() { Matrix4f m; for (int i = 0; i < 4; ++i) m(i, i) = 1.f; return m; }
    float &operator()(int r, int c) { return data_[c * 4 + r]; }
    float operator()(int r, int c) const { return data_[c * 4 + r]; }
    const float *data() const { return data_; }
};
}  // namespace Eigen

// ---------------------------------------------------------------------------------- math_types.h
struct sb_uint3 { unsigned int x, y, z; };
struct sb_float3 { float x, y, z; };
struct sb_float4 { float x, y, z, w; };
inline sb_uint3 make_sb_uint3(unsigned a, unsigned b, unsigned c) { return {a, b, c}; }
inline sb_float3 make_sb_float3(float a, float b, float c) { return {a, b, c}; }

// ---------------------------------------------------------------------------------- utils.h:29-37 (FastLock)
class FastLock {
public:
    void lock() { while (flag_.test_and_set(std::memory_order_acquire)) {} }
    void unlock() { flag_.clear(std::memory_order_release); }
private:
    std::atomic_flag flag_ = ATOMIC_FLAG_INIT;
};

// ---------------------------------------------------------------------------------- TimeStamp.h:22-50
namespace slambench {
struct TimeStamp {
    uint32_t S;
    uint32_t Ns;
    uint64_t ToNs() const { return (S * 1000000000ULL) + Ns; }
    double ToS() const { return ToNs() / 1000000000.0; }
    static TimeStamp get(uint32_t s, uint32_t ns) { TimeStamp t; t.S = s; t.Ns = ns; return t; }
};
inline bool operator<(const TimeStamp &a, const TimeStamp &b) { return a.ToNs() < b.ToNs(); }
}  // namespace slambench

// ---------------------------------------------------------------------------------- ParameterComponent.h, Parameters.h
class ParameterComponent;
struct Parameter {
    Parameter(std::string short_option, std::string long_option, std::string description)
        : short_(std::move(short_option)), long_(std::move(long_option)), desc_(std::move(description)) {}
    virtual ~Parameter() = default;
    const std::string &getShortOption() const { return short_; }
    const std::string &getLongOption() const { return long_; }
    const std::string &getDescription() const { return desc_; }
    virtual bool requiresValue() = 0;
    virtual void setValue(const char *) = 0;
    virtual void resetValue() = 0;
    virtual const std::string getStrValue() const = 0;
    virtual const std::string getStrDefault() = 0;
    // what `--dse` prints per parameter (Parameters.h:161-167)
    virtual const std::string getStrDetails() {
        std::ostringstream res;
        res << "name:" << long_ << "\n" << "short-option:" << short_ << "\n" << "description:" << desc_ << "\n";
        return res.str();
    }
private:
    std::string short_, long_, desc_;
};

template <typename T> inline std::string sb_lite_to_string(const T &v) { std::ostringstream s; s << v; return s.str(); }
template <> inline std::string sb_lite_to_string(const sb_float3 &v) {
    std::ostringstream s; s << v.x << "," << v.y << "," << v.z; return s.str();
}
template <> inline std::string sb_lite_to_string(const sb_uint3 &v) {
    std::ostringstream s; s << v.x << "," << v.y << "," << v.z; return s.str();
}
template <typename T> inline void sb_lite_parse(const char *s, T *out) { std::istringstream(s) >> *out; }
template <> inline void sb_lite_parse(const char *s, sb_float3 *out) {
    std::istringstream in(s); std::string tok; float v[3] = {out->x, out->y, out->z}; int n = 0;
    while (n < 3 && std::getline(in, tok, ',')) v[n++] = (float)std::atof(tok.c_str());
    if (n == 1) v[1] = v[2] = v[0];                      // one value applies to the three axes
    out->x = v[0]; out->y = v[1]; out->z = v[2];
}
template <> inline void sb_lite_parse(const char *s, sb_uint3 *out) {
    std::istringstream in(s); std::string tok; unsigned v[3] = {out->x, out->y, out->z}; int n = 0;
    while (n < 3 && std::getline(in, tok, ',')) v[n++] = (unsigned)std::strtoul(tok.c_str(), nullptr, 10);
    if (n == 1) v[1] = v[2] = v[0];
    out->x = v[0]; out->y = v[1]; out->z = v[2];
}

// Parameters.h:201-238: value lives in caller-owned storage, reset to *default_v when registered
template <typename T>
struct TypedParameter : Parameter {
    TypedParameter(std::string short_option, std::string name, std::string description, T *ptr, const T *default_v)
        : Parameter(std::move(short_option), std::move(name), std::move(description)), default_(default_v), ptr_(ptr) {
        if (!ptr) throw std::logic_error("TypedParameter needs a valid storage pointer");
    }
    bool requiresValue() override { return true; }
    void setValue(const char *s) override { sb_lite_parse<T>(s, ptr_); }
    void resetValue() override { if (default_) *ptr_ = *default_; }
    const std::string getStrValue() const override { return sb_lite_to_string<T>(*ptr_); }
    const std::string getStrDefault() override { return default_ ? sb_lite_to_string<T>(*default_) : "nullptr"; }
    const std::string getStrDetails() override {          // Parameters.h:226-232
        std::ostringstream res;
        res << Parameter::getStrDetails() << "type:" << typeid(T).name() << "\n" << "default:" << getStrDefault() << "\n";
        return res.str();
    }
private:
    const T *default_;
    T *ptr_;
};

// Parameters.h:392-410: a parameter with an explicit list of admissible values
template <typename T>
struct DiscretParameter : TypedParameter<T> {
    DiscretParameter(std::vector<T> values, std::string short_option, std::string name, std::string description, T *ptr,
                     const T *default_v)
        : TypedParameter<T>(std::move(short_option), std::move(name), std::move(description), ptr, default_v),
          values_(std::move(values)) {}
    const std::string getStrDetails() override {
        std::ostringstream res;
        res << TypedParameter<T>::getStrDetails();
        for (const T &v : values_) res << "values:" << sb_lite_to_string<T>(v) << "\n";
        return res.str();
    }
private:
    std::vector<T> values_;
};

// Parameters.h:413-429: a parameter with a range
template <typename T>
struct BoundedParameter : TypedParameter<T> {
    BoundedParameter(T min, T max, std::string short_option, std::string name, std::string description, T *ptr,
                     const T *default_v)
        : TypedParameter<T>(std::move(short_option), std::move(name), std::move(description), ptr, default_v),
          min_(min), max_(max) {}
    const std::string getStrDetails() override {
        std::ostringstream res;
        res << TypedParameter<T>::getStrDetails() << "min:" << min_ << "\n" << "max:" << max_ << "\n";
        return res.str();
    }
private:
    T min_, max_;
};

// ParameterComponent.h:25-60
class ParameterComponent {
public:
    explicit ParameterComponent(std::string name) : name_(std::move(name)) {}
    virtual ~ParameterComponent() { for (auto *p : arguments_) delete p; }
    const std::vector<Parameter *> &getParameters() const { return arguments_; }
    std::string getName() const { return name_; }
    template <typename T> void addParameter(T p) {
        T *param = new T(p);
        arguments_.push_back(param);
        param->resetValue();
    }
private:
    std::string name_;
    std::vector<Parameter *> arguments_;
};

// ---------------------------------------------------------------------------------- io
namespace slambench {
namespace io {
namespace pixelformat {
enum EPixelFormat { UNKNOWN, G_I_8, RGB_III_888, D_I_8, D_F_32, D_I_16, D_F_64, RGBA_IIII_8888 };   // io/PixelFormat.h:20-31
inline size_t GetPixelSize(EPixelFormat f) {
    switch (f) { case G_I_8: case D_I_8: return 1; case D_I_16: return 2; case RGB_III_888: return 3;
                 case D_F_32: case RGBA_IIII_8888: return 4; case D_F_64: return 8; default: return 0; }
}
inline std::string TypeOf(EPixelFormat f) {
    switch (f) { case G_I_8: return "grey"; case RGB_III_888: case RGBA_IIII_8888: return "rgb";
                 case D_I_8: case D_F_32: case D_I_16: case D_F_64: return "depth"; default: return "unknown"; }
}
}  // namespace pixelformat

// io/sensor/Sensor.h
class Sensor : public ParameterComponent {
public:
    typedef std::string sensor_name_t;
    typedef std::string sensor_type_t;
    Sensor(const sensor_name_t &name, const sensor_type_t &type) : ParameterComponent(name), name_(name), type_(type) {}
    const sensor_name_t &getName() const { return name_; }
    const sensor_type_t &GetType() const { return type_; }
    uint8_t Index = 0;
    float Rate = 0.f;
private:
    sensor_name_t name_;
    sensor_type_t type_;
};

// io/sensor/CameraSensor.h:20-58
class CameraSensor : public Sensor {
public:
    typedef float intrinsics_t[4];
    CameraSensor(const sensor_name_t &name, const sensor_type_t &type = "Camera") : Sensor(name, type) {}
    uint32_t Width = 0, Height = 0;
    pixelformat::EPixelFormat PixelFormat = pixelformat::UNKNOWN;
    intrinsics_t Intrinsics = {0, 0, 0, 0};          // normalised: fx/W, fy/H, cx/W, cy/H
    size_t GetFrameSize() const { return (size_t)Width * Height * pixelformat::GetPixelSize(PixelFormat); }
};

// io/sensor/DepthSensor.h:17-35
class DepthSensor : public CameraSensor {
public:
    typedef float disparity_params_t[2];
    explicit DepthSensor(const sensor_name_t &name) : CameraSensor(name, "Depth") {}
    disparity_params_t DisparityParams = {0.001f, 0.f};   // raw unit -> metres scale, offset
};

// io/sensor/SensorCollection.h
class SensorCollection {
public:
    void AddSensor(Sensor *s) { s->Index = (uint8_t)sensors_.size(); sensors_.push_back(s); }
    std::vector<Sensor *>::const_iterator begin() const { return sensors_.begin(); }
    std::vector<Sensor *>::const_iterator end() const { return sensors_.end(); }
    size_t size() const { return sensors_.size(); }
private:
    std::vector<Sensor *> sensors_;
};

// io/sensor/CameraSensorFinder.h:25-40, src/io/sensor/CameraSensorFinder.cpp:22-69
class CameraSensorFinder {
public:
    typedef std::pair<std::string, std::string> CameraSensorFilter;
    CameraSensor *FindOne(const SensorCollection &sensors, const std::initializer_list<CameraSensorFilter> &filters) {
        for (Sensor *s : sensors) {
            if (s->GetType() != "Camera" && s->GetType() != "Depth") continue;
            CameraSensor *cam = static_cast<CameraSensor *>(s);
            bool ok = true;
            for (const auto &f : filters) ok = ok && test(cam, f);
            if (ok) return cam;
        }
        return nullptr;
    }
private:
    static bool test(const CameraSensor *c, const CameraSensorFilter &f) {
        if (f.first == "width") return c->Width == std::stoul(f.second);
        if (f.first == "height") return c->Height == std::stoul(f.second);
        if (f.first == "name") return c->getName() == f.second;
        if (f.first == "camera_type") return pixelformat::TypeOf(c->PixelFormat) == f.second;
        throw std::out_of_range("unknown camera filter " + f.first);
    }
};

// io/SLAMFrame.h:28-56
class SLAMFrame {
public:
    virtual ~SLAMFrame() = default;
    TimeStamp Timestamp;
    Sensor *FrameSensor = nullptr;
    virtual size_t GetSize() const = 0;
    virtual void *GetData() = 0;
    virtual void FreeData() = 0;
};
class SLAMInMemoryFrame : public SLAMFrame {
public:
    void *Data = nullptr;
    size_t Size = 0;
    size_t GetSize() const override { return Size; }
    void *GetData() override { return Data; }
    void FreeData() override {}
};

// io/InputInterface.h (only what SLAMBenchLibraryHelper::get_sensors needs)
class InputInterface {
public:
    SensorCollection &GetSensors() { return sensors_; }
private:
    SensorCollection sensors_;
};
}  // namespace io

// ---------------------------------------------------------------------------------- values/Value.h
namespace values {
enum ValueType { VT_UNKNOWN, VT_POSE, VT_POINTCLOUD, VT_FRAME, VT_DOUBLE };
class Value {
public:
    explicit Value(ValueType t) : type_(t) {}
    virtual ~Value() = default;
    ValueType GetType() const { return type_; }
private:
    ValueType type_;
};
class PoseValue : public Value {             // values/Value.h:141-149
public:
    explicit PoseValue(const Eigen::Matrix4f &pose) : Value(VT_POSE), pose_(pose) {}
    const Eigen::Matrix4f &GetValue() const { return pose_; }
private:
    Eigen::Matrix4f pose_;
};
struct Point3DF { float X, Y, Z; };
class PointCloudValue : public Value {       // values/Value.h:231-252
public:
    PointCloudValue() : Value(VT_POINTCLOUD) {}
    void AddPoint(const Point3DF &p) { points_.push_back(p); }
    const std::vector<Point3DF> &GetPoints() const { return points_; }
private:
    std::vector<Point3DF> points_;
};
class FrameValue : public Value {            // values/Value.h:318-335
public:
    FrameValue(uint32_t w, uint32_t h, io::pixelformat::EPixelFormat fmt, void *data)
        : Value(VT_FRAME), width_(w), height_(h), fmt_(fmt),
          data_((unsigned char *)data, (unsigned char *)data + (size_t)w * h * io::pixelformat::GetPixelSize(fmt)) {}
    const void *GetData() const { return data_.data(); }
    uint32_t GetWidth() const { return width_; }
    uint32_t GetHeight() const { return height_; }
private:
    uint32_t width_, height_;
    io::pixelformat::EPixelFormat fmt_;
    std::vector<unsigned char> data_;
};
}  // namespace values

// ---------------------------------------------------------------------------------- outputs/Output.h, OutputManager.h
namespace outputs {
class BaseOutput {
public:
    typedef std::map<TimeStamp, const values::Value *> value_map_t;
    BaseOutput(std::string name, values::ValueType type, bool main_output)
        : name_(std::move(name)), type_(type), main_(main_output) {}
    virtual ~BaseOutput() = default;
    const std::string &GetName() const { return name_; }
    values::ValueType GetType() const { return type_; }
    void SetKeepOnlyMostRecent(bool b) { only_most_recent_ = b; }
    bool GetKeepOnlyMostRecent() const { return only_most_recent_; }
    bool IsActive() const { return active_; }
    void SetActive(bool a) { active_ = a; }
    bool IsMainOutput() const { return main_; }
    virtual const value_map_t &GetValues() const = 0;
private:
    std::string name_;
    values::ValueType type_;
    bool only_most_recent_ = false, active_ = false, main_;
};
class Output : public BaseOutput {           // outputs/Output.h:76-90, src/outputs/Output.cpp:50-63
public:
    Output(const std::string &name, values::ValueType type, bool main_output = false) : BaseOutput(name, type, main_output) {}
    ~Output() override { for (auto &kv : values_) delete kv.second; }
    void AddPoint(TimeStamp time, const values::Value *value) {
        if (!IsActive()) throw std::logic_error("Output is not active");
        if (value->GetType() != GetType()) throw std::logic_error("value type does not match the output");
        if (GetKeepOnlyMostRecent()) { for (auto &kv : values_) delete kv.second; values_.clear(); }
        values_[time] = value;
    }
    const value_map_t &GetValues() const override { return values_; }
    const value_map_t::value_type &GetMostRecentValue() const { return *values_.rbegin(); }
    bool Empty() const { return values_.empty(); }
private:
    value_map_t values_;
};
class OutputManager {                        // outputs/OutputManager.h:30-55
public:
    ~OutputManager() { for (auto &kv : outputs_) delete kv.second; }
    void RegisterOutput(BaseOutput *o) {
        if (outputs_.count(o->GetName())) throw std::logic_error("output already registered: " + o->GetName());
        outputs_[o->GetName()] = o;
    }
    BaseOutput *GetOutput(const std::string &name) { return outputs_.at(name); }
    BaseOutput *GetMainOutput(values::ValueType t) {
        for (auto &kv : outputs_) if (kv.second->IsMainOutput() && kv.second->GetType() == t) return kv.second;
        return nullptr;
    }
    std::map<std::string, BaseOutput *>::iterator begin() { return outputs_.begin(); }
    std::map<std::string, BaseOutput *>::iterator end() { return outputs_.end(); }
    FastLock &GetLock() { return lock_; }
private:
    std::map<std::string, BaseOutput *> outputs_;
    FastLock lock_;
};
}  // namespace outputs

// ---------------------------------------------------------------------------------- metrics/MetricManager.h, Phase.h
namespace metrics {
class MetricManager;
class Phase {                                // metrics/Phase.h:26-37
public:
    Phase(MetricManager *mgr, std::string name) : name_(std::move(name)), manager_(mgr) {}
    void Begin();
    void End();
    const std::string &GetName() { return name_; }
private:
    std::string name_;
    MetricManager *manager_;
};
// Frame / phase bracketing with the wall-clock duration metric folded in (src/metrics/MetricManager.cpp:54-111,
// src/metrics/DurationMetric.cpp:30-71).  As in the reference, EndFrame asks for the duration of EVERY declared
// phase and throws std::out_of_range for one that was not bracketed in this frame.
class MetricManager {
public:
    ~MetricManager() { for (auto *p : phases_) delete p; }
    void AddPhase(const std::string &name) { phases_.push_back(new Phase(this, name)); }
    Phase *GetPhase(const std::string &name) {
        for (auto *p : phases_) if (p->GetName() == name) return p;
        throw std::logic_error("Could not find phase " + name);
    }
    const std::vector<Phase *> &GetPhases() const { return phases_; }
    void BeginPhase(Phase *p) { start_[p] = clock::now(); }
    void EndPhase(Phase *p) { duration_[p] = std::chrono::duration<double>(clock::now() - start_.at(p)).count(); }
    void BeginFrame() { duration_.clear(); frame_start_ = clock::now(); }
    void EndFrame() {
        last_.clear();
        last_frame_ = std::chrono::duration<double>(clock::now() - frame_start_).count();
        for (auto *p : phases_) last_[p->GetName()] = duration_.at(p);
    }
    double LastFrameDuration() const { return last_frame_; }
    const std::map<std::string, double> &LastPhaseDurations() const { return last_; }
private:
    typedef std::chrono::steady_clock clock;
    std::vector<Phase *> phases_;
    std::map<Phase *, clock::time_point> start_;
    std::map<Phase *, double> duration_;
    std::map<std::string, double> last_;
    clock::time_point frame_start_;
    double last_frame_ = 0.0;
};
inline void Phase::Begin() { manager_->BeginPhase(this); }
inline void Phase::End() { manager_->EndPhase(this); }
}  // namespace metrics
}  // namespace slambench

// ---------------------------------------------------------------------------------- SLAMBenchLibraryHelper.h:23-77
class SLAMBenchLibraryHelper : public ParameterComponent {
public:
    bool (*c_sb_new_slam_configuration)(SLAMBenchLibraryHelper *) = nullptr;
    bool (*c_sb_init_slam_system)(SLAMBenchLibraryHelper *) = nullptr;
    bool (*c_sb_update_frame)(SLAMBenchLibraryHelper *, slambench::io::SLAMFrame *) = nullptr;
    bool (*c_sb_process_once)(SLAMBenchLibraryHelper *) = nullptr;
    bool (*c_sb_clean_slam_system)() = nullptr;
    bool (*c_sb_update_outputs)(SLAMBenchLibraryHelper *, const slambench::TimeStamp *) = nullptr;
    bool (*c_sb_relocalize)(SLAMBenchLibraryHelper *) = nullptr;

    SLAMBenchLibraryHelper(const std::string &id, std::string lib, std::ostream &l, slambench::io::InputInterface *i)
        : ParameterComponent(id), identifier_(id), library_name_(std::move(lib)), log_stream_(l), input_interface_(i) {}
    const std::string &GetIdentifier() const { return identifier_; }
    const std::string &GetLibraryName() const { return library_name_; }
    std::ostream &GetLogStream() { return log_stream_; }
    slambench::metrics::MetricManager &GetMetricManager() { return metric_manager_; }
    slambench::outputs::OutputManager &GetOutputManager() { return output_manager_; }
    slambench::io::InputInterface *GetInputInterface() { return input_interface_; }
    const slambench::io::SensorCollection &get_sensors() { return input_interface_->GetSensors(); }
private:
    std::string identifier_, library_name_;
    slambench::metrics::MetricManager metric_manager_;
    std::ostream &log_stream_;
    slambench::io::InputInterface *input_interface_;
    slambench::outputs::OutputManager output_manager_;
};

// ---------------------------------------------------------------------------------- SLAMBenchAPI.h:30-71
bool sb_new_slam_configuration(SLAMBenchLibraryHelper *slam_settings);
bool sb_init_slam_system(SLAMBenchLibraryHelper *slam_settings);
bool sb_update_frame(SLAMBenchLibraryHelper *slam_settings, slambench::io::SLAMFrame *type);
bool sb_process_once(SLAMBenchLibraryHelper *slam_settings);
bool sb_update_outputs(SLAMBenchLibraryHelper *lib, const slambench::TimeStamp *latest_output);
bool sb_clean_slam_system();
extern "C" {
bool c_sb_new_slam_configuration(void *slam_settings);
bool c_sb_init_slam_system(void *slam_settings);
bool c_sb_update_frame(void *slam_settings, void *type);
bool c_sb_process_once(void *slam_settings);
bool c_sb_update_outputs(void *lib, void *timestamp);
bool c_sb_clean_slam_system();
}

#endif  // SB_LITE_ALL_H_
