"""Reader and writer of SLAMBench's `.slam` dataset container: every sensor type the reference registers is
parsed (a depth-only plugin uses Depth and the ground truth and steps over the frames of the others).

Layout (little endian, no padding), as the reference serialises it:
  header   "SLAM\\0" (5 bytes), u32 version = 1, u32 sensor count
           (reference framework/shared/src/io/serialisation/SLAMFileHeaderSerialiser.cpp:34-44,
            framework/shared/include/io/SLAMFile.h:33-39)
  sensor   u8 n + name (n bytes, NUL included), u8 n + type, u8 index, u32 n + description, f32 rate,
           16 x f32 pose (column-major), f32 delay, then the type's own block
           (reference framework/shared/src/io/sensor/Sensor.cpp:77-97,
            framework/shared/src/io/deserialisation/SensorCollectionDeserialiser.cpp:40-54)
    Camera                 i32 frame format, i32 pixel format, u32 width, u32 height, 4 x f32 intrinsics
                           (normalised by width / height), i32 distortion type, 3 x 5 x f32 distortion
                           (reference framework/shared/src/io/sensor/CameraSensor.cpp:66-79)
    Depth                  the first five Camera fields, 2 x f32 disparity parameters, i32 disparity type,
                           i32 distortion type, 5 x f32 distortion
                           (reference framework/shared/src/io/sensor/DepthSensor.cpp:38-56)
    GroundTruthTrajectory  nothing; its frames are 4x4 f32 poses, column-major
                           (reference framework/shared/src/io/sensor/GroundTruthSensor.cpp:23-37)
    Accelerometer, Gyro    12 x f32 intrinsic, 3 x f32 noise variances, 3 x f32 bias variances; frames 3 x f32
                           (reference framework/shared/src/io/sensor/AccelerometerSensor.cpp:23-35, GyroSensor.cpp:22-34;
                            the TUM generator adds an Accelerometer by default, tools/dataset-tools/TUM.cpp:340-389)
    IMU                    8 x f32 noise / saturation figures; frames 6 x f32 (IMUSensor.cpp:22-44)
    Odom                   nothing; frames 13 x f32 (OdomSensor.cpp:19-31)
    Lidar                  f32 points per cycle, f32 altitude, 2 x f32 + 2 x f32 fields of view, 2 x f32 angular
                           resolutions, i32 beams; variable-size frames (LidarSensor.cpp:25-53, :84)
    PointCloud             nothing; variable-size frames: u32 count + count x 3 x f32; a ground-truth sensor
                           (PointCloudSensor.cpp:21-53, io/format/PointCloud.cpp:17-48; the ICL-NUIM generator adds
                            one when given a .ply, tools/dataset-tools/ICLNUIM.cpp:388-406)
    EventCamera            i32 width, i32 height; registered as variable-size, payload 2 coordinates of 1, 2 or 4
                           bytes (by sensor size) + 1 byte (EventCameraSensor.cpp:25-42, :73)
  frames   until end of file: u32 seconds, u32 nanoseconds, u8 sensor index, then for a variable-size sensor a u32
           byte count, then the payload (the count for a variable-size sensor whose frame size follows it, else
           the sensor's fixed frame size)
           (reference framework/shared/src/io/serialisation/SLAMFrameSerialiser.cpp:21-55,
            framework/shared/src/io/deserialisation/SLAMFileDeserialiser.cpp:56-79)
"""
import struct
from dataclasses import dataclass, field

import numpy as np

MAGIC = b"SLAM\0"
VERSION = 1

# reference framework/shared/include/io/PixelFormat.h:20-31 and FrameFormat.h:19-24
PIXEL_FORMATS = {"UNKNOWN": 0, "G_I_8": 1, "RGB_III_888": 2, "D_I_8": 3, "D_F_32": 4, "D_I_16": 5, "D_F_64": 6,
                 "RGBA_IIII_8888": 7}
PIXEL_SIZE = {0: 0, 1: 1, 2: 3, 3: 1, 4: 4, 5: 2, 6: 8, 7: 4}
FRAME_FORMAT_RASTER = 1

CAMERA, DEPTH, GROUND_TRUTH = "Camera", "Depth", "GroundTruthTrajectory"
ACCELEROMETER, GYRO, IMU, ODOM, LIDAR, POINT_CLOUD, EVENT_CAMERA = ("Accelerometer", "Gyro", "IMU", "Odom", "Lidar",
                                                                    "PointCloud", "EventCamera")
# type -> (struct format of the type's own block, fixed frame size in bytes or None, variable-size flag of the
# reference's sensor database, ground-truth flag); Camera / Depth / EventCamera sizes depend on the block
_SIMPLE_SENSORS = {
    GROUND_TRUTH: ("", 64, False, True),
    ACCELEROMETER: ("<18f", 12, False, False),
    GYRO: ("<18f", 12, False, False),
    IMU: ("<8f", 24, False, False),
    ODOM: ("", 52, False, False),
    LIDAR: ("<8fi", None, True, False),
    POINT_CLOUD: ("", None, True, True),
    EVENT_CAMERA: ("<ii", None, True, False),
}


@dataclass
class Sensor:
    name: str
    type: str
    index: int = 0
    description: str = ""
    rate: float = 0.0
    pose: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    delay: float = 0.0
    # cameras
    frame_format: int = FRAME_FORMAT_RASTER
    pixel_format: int = 0
    width: int = 0
    height: int = 0
    intrinsics: tuple = (0.0, 0.0, 0.0, 0.0)          # fx / width, fy / height, cx / width, cy / height
    distortion_type: int = 0
    distortion: tuple = (0.0,) * 15                    # Camera: radial-tangential, equidistant, generic
    # depth
    disparity_params: tuple = (0.001, 0.0)             # raw unit -> metres: d * p[0] + p[1]
    disparity_type: int = 0
    # the other sensor types: the values of their own block, in file order (zeros when written without)
    block: tuple = ()

    @property
    def variable_size(self):
        """The reference's sensor database flag: a u32 byte count precedes every frame of the sensor."""
        return self.type in _SIMPLE_SENSORS and _SIMPLE_SENSORS[self.type][2]

    @property
    def is_ground_truth(self):
        return self.type in _SIMPLE_SENSORS and _SIMPLE_SENSORS[self.type][3]

    @property
    def frame_size(self):
        """Payload bytes of a frame; None when the count in front of each frame decides (Lidar, PointCloud)."""
        if self.type in (CAMERA, DEPTH):
            return self.width * self.height * PIXEL_SIZE[self.pixel_format]
        if self.type == EVENT_CAMERA:                  # x, y, on/off (EventCameraSensor.cpp:25-35)
            w, h = self.block[:2] if len(self.block) >= 2 else (self.width, self.height)
            coord = 1 if (w <= 0xFF and h <= 0xFF) else 2 if (w <= 0xFFFF and h <= 0xFFFF) else 4
            return 2 * coord + 1
        if self.type in _SIMPLE_SENSORS:
            return _SIMPLE_SENSORS[self.type][1]
        raise ValueError("sensor type %r is not one the reference registers" % self.type)


def _cstr(s):
    return s.encode("utf-8") + b"\0"


def _write_sensor(f, s):
    name, typ, desc = _cstr(s.name), _cstr(s.type), _cstr(s.description)
    f.write(struct.pack("<B", len(name)) + name)
    f.write(struct.pack("<B", len(typ)) + typ)
    f.write(struct.pack("<B", s.index))
    f.write(struct.pack("<I", len(desc)) + desc)
    f.write(struct.pack("<f", s.rate))
    f.write(np.asarray(s.pose, np.float32).T.tobytes())          # Eigen stores column-major
    f.write(struct.pack("<f", s.delay))
    if s.type == CAMERA:
        f.write(struct.pack("<iiII4fi", s.frame_format, s.pixel_format, s.width, s.height, *s.intrinsics,
                            s.distortion_type))
        f.write(struct.pack("<15f", *s.distortion))
    elif s.type == DEPTH:
        f.write(struct.pack("<iiII4f2fii", s.frame_format, s.pixel_format, s.width, s.height, *s.intrinsics,
                            *s.disparity_params, s.disparity_type, s.distortion_type))
        f.write(struct.pack("<5f", *s.distortion[:5]))
    elif s.type in _SIMPLE_SENSORS:
        fmt = _SIMPLE_SENSORS[s.type][0]
        if fmt:
            n = len(struct.unpack(fmt, bytes(struct.calcsize(fmt))))
            values = tuple(s.block) if s.block else (0,) * n
            if s.type == EVENT_CAMERA and not s.block:
                values = (s.width, s.height)
            f.write(struct.pack(fmt, *values))
    else:
        raise ValueError("sensor type %r is not one the reference registers" % s.type)


def write_slam(path, sensors, frames):
    """sensors: list of Sensor (index = position in the list); frames: iterable of (timestamp, sensor index,
    bytes-like payload) in file order; a timestamp is seconds as a float or a (seconds, nanoseconds) tuple."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, len(sensors)))
        for i, s in enumerate(sensors):
            s.index = i
            _write_sensor(f, s)
        for ts, idx, payload in frames:
            payload = bytes(payload)
            want = sensors[idx].frame_size
            if want is not None and len(payload) != want:
                raise ValueError("frame of sensor %d has %d bytes, expected %d" % (idx, len(payload), want))
            if isinstance(ts, tuple):                       # (seconds, nanoseconds), kept exactly
                sec, nsec = ts
            else:
                sec = int(ts)
                nsec = int(round((ts - sec) * 1e9))
            f.write(struct.pack("<IIB", sec, nsec, idx))
            if sensors[idx].variable_size:
                f.write(struct.pack("<I", len(payload)))
            f.write(payload)


def _read_exact(f, n):
    b = f.read(n)
    if len(b) != n:
        raise EOFError("truncated .slam file")
    return b


def _read_sensor(f):
    (n,) = struct.unpack("<B", _read_exact(f, 1))
    name = _read_exact(f, n).split(b"\0")[0].decode()
    (n,) = struct.unpack("<B", _read_exact(f, 1))
    typ = _read_exact(f, n).split(b"\0")[0].decode()
    (index,) = struct.unpack("<B", _read_exact(f, 1))
    (n,) = struct.unpack("<I", _read_exact(f, 4))
    desc = _read_exact(f, n).split(b"\0")[0].decode()
    (rate,) = struct.unpack("<f", _read_exact(f, 4))
    pose = np.frombuffer(_read_exact(f, 64), np.float32).reshape(4, 4).T.copy()
    (delay,) = struct.unpack("<f", _read_exact(f, 4))
    s = Sensor(name=name, type=typ, index=index, description=desc, rate=rate, pose=pose, delay=delay)
    if typ == CAMERA:
        v = struct.unpack("<iiII4fi", _read_exact(f, 36))
        s.frame_format, s.pixel_format, s.width, s.height = v[:4]
        s.intrinsics, s.distortion_type = tuple(v[4:8]), v[8]
        s.distortion = struct.unpack("<15f", _read_exact(f, 60))
    elif typ == DEPTH:
        v = struct.unpack("<iiII4f2fii", _read_exact(f, 48))
        s.frame_format, s.pixel_format, s.width, s.height = v[:4]
        s.intrinsics, s.disparity_params, s.disparity_type, s.distortion_type = tuple(v[4:8]), tuple(v[8:10]), v[10], v[11]
        s.distortion = struct.unpack("<5f", _read_exact(f, 20)) + (0.0,) * 10
    elif typ in _SIMPLE_SENSORS:
        fmt = _SIMPLE_SENSORS[typ][0]
        if fmt:
            s.block = struct.unpack(fmt, _read_exact(f, struct.calcsize(fmt)))
            if typ == EVENT_CAMERA:
                s.width, s.height = s.block
    else:
        raise ValueError("sensor %r has type %r, which the reference does not register" % (name, typ))
    return s


def read_slam_header(f):
    if _read_exact(f, 5) != MAGIC:
        raise ValueError("not a .slam file (bad magic number)")
    version, count = struct.unpack("<II", _read_exact(f, 8))
    if version != VERSION:
        raise ValueError(".slam version %d is not supported (expected %d)" % (version, VERSION))
    return [_read_sensor(f) for _ in range(count)]


def iter_frames(f, sensors):
    """Yields (timestamp_s, sensor, payload bytes) until the end of the file."""
    while True:
        head = f.read(9)
        if not head:
            return
        if len(head) != 9:
            raise EOFError("truncated frame header")
        sec, nsec, idx = struct.unpack("<IIB", head)
        if idx >= len(sensors):
            raise ValueError("frame refers to sensor %d of %d" % (idx, len(sensors)))
        size = sensors[idx].frame_size
        if sensors[idx].variable_size:
            (counted,) = struct.unpack("<I", _read_exact(f, 4))
            if size is None:
                size = counted
        yield sec + nsec * 1e-9, sensors[idx], _read_exact(f, size)


def read_depth_sequence(path):
    """Convenience for depth-only use: returns (depth sensor, uint16 depth [n, h, w], depth timestamps,
    ground-truth poses [m, 4, 4], ground-truth timestamps)."""
    with open(path, "rb") as f:
        sensors = read_slam_header(f)
        depth_sensor = next((s for s in sensors if s.type == DEPTH), None)
        if depth_sensor is None:
            raise ValueError("%s holds no depth sensor" % path)
        if depth_sensor.pixel_format != PIXEL_FORMATS["D_I_16"]:
            raise ValueError("depth frames must be D_I_16")
        depth, ts, gt, gt_ts = [], [], [], []
        for t, s, payload in iter_frames(f, sensors):
            if s is depth_sensor:
                depth.append(np.frombuffer(payload, np.uint16).reshape(s.height, s.width))
                ts.append(t)
            elif s.type == GROUND_TRUTH:
                gt.append(np.frombuffer(payload, np.float32).reshape(4, 4).T.astype(np.float64))
                gt_ts.append(t)
    return depth_sensor, np.stack(depth), np.asarray(ts), (np.stack(gt) if gt else np.zeros((0, 4, 4))), np.asarray(gt_ts)


def write_depth_sequence(path, depth, timestamps, intrinsics_px, gt_poses=None, disparity_scale=0.001, rate=30.0,
                         with_rgb_and_grey=True):
    """A .slam file shaped like the reference's ICL-NUIM / TUM files: Grey, RGB (empty images), Depth and the
    ground-truth trajectory, frames interleaved per time step (ground truth first, as the generators do)."""
    depth = np.ascontiguousarray(depth, np.uint16)
    n, h, w = depth.shape
    fx, fy, cx, cy = intrinsics_px
    intr = (fx / w, fy / h, cx / w, cy / h)
    sensors = []
    if gt_poses is not None:
        sensors.append(Sensor("GroundTruth", GROUND_TRUTH, description="GroundTruthSensor", rate=rate))
    if with_rgb_and_grey:
        sensors.append(Sensor("Grey", CAMERA, description="Grey", rate=rate, pixel_format=PIXEL_FORMATS["G_I_8"],
                              width=w, height=h, intrinsics=intr))
        sensors.append(Sensor("RGB", CAMERA, description="RGB", rate=rate, pixel_format=PIXEL_FORMATS["RGB_III_888"],
                              width=w, height=h, intrinsics=intr))
    sensors.append(Sensor("Depth", DEPTH, description="Depth", rate=rate, pixel_format=PIXEL_FORMATS["D_I_16"],
                          width=w, height=h, intrinsics=intr, disparity_params=(disparity_scale, 0.0)))
    index = {s.name: i for i, s in enumerate(sensors)}

    def frames():
        grey = bytes(w * h)
        rgb = bytes(3 * w * h)
        for i in range(n):
            t = float(timestamps[i])
            if gt_poses is not None:
                yield t, index["GroundTruth"], np.asarray(gt_poses[i], np.float32).T.tobytes()
            if with_rgb_and_grey:
                yield t, index["Grey"], grey
                yield t, index["RGB"], rgb
            yield t, index["Depth"], depth[i].tobytes()

    write_slam(path, sensors, frames())
    return sensors
