"""Importers of the two raw dataset layouts the reference's dataset generator turns into `.slam` files for a depth
plugin: TUM RGB-D folders and ICL-NUIM folders.  Each returns a DepthSequence (depth frames, time stamps, ground
truth) that `to_slam` writes as a `.slam` file or that can be fed to the pipeline directly.

TUM (reference framework/tools/dataset-tools/TUM.cpp:31-110, 264-336, include/TUM.h:42-120):
  depth.txt        "seconds.fraction filename" per line, '#' comments
  depth/*.png      16-bit greyscale, 5000 units per metre (kept as they are; the sensor's disparity parameters say
                   0.0002 m per unit)
  groundtruth.txt  "seconds.fraction tx ty tz qx qy qz qw"; the pose is the quaternion's rotation matrix and t
  intrinsics       by folder name: freiburg1 / freiburg2 / freiburg3, otherwise the ROS defaults

ICL-NUIM (reference framework/tools/dataset-tools/ICLNUIM.cpp:38-116, 130-211, 266-368):
  scene_00_NNNN.depth  640 x 480 floats as text: distance along the ray in metres.  Stored as uint16
                       value * 1000 (truncated), then divided by sqrt(((u - 319.5) / 481.2)^2 +
                       ((v - 239.5) / -480)^2 + 1) to get the distance along the optical axis
  scene_00_NNNN.txt    POV-Ray camera description: cam_pos, cam_dir, cam_up, cam_right -> pose rows
                       right, up, direction (each normalised) and the position
  groundtruth.txt      (ETH illumination variant) "frame tx ty tz qx qy qz qw" with depth in
                       scene_00_NNNN.depth.png
  25 frames per second, frame n at n / 25 s; intrinsics 481.2, -480 (or +480 with positive_focal), 319.5, 239.5
"""
import os
import re
import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

from . import slamfile

TUM_DEPTH_INTRINSICS = {                       # pixels at 640 x 480 (reference include/TUM.h:99-120)
    "freiburg1": (591.1, 590.1, 331.0, 234.0),
    "freiburg2": (580.8, 581.8, 308.8, 253.0),
    "freiburg3": (567.6, 570.2, 324.7, 250.1),
    "default": (525.0, 525.0, 319.5, 239.5),
}
ICL_INTRINSICS = (481.2, -480.0, 319.5, 239.5)


@dataclass
class DepthSequence:
    width: int
    height: int
    intrinsics_px: tuple                       # fx, fy, cx, cy in pixels
    disparity_scale: float                     # metres per raw unit
    rate: float
    depth: np.ndarray                          # uint16 [n, h, w]
    timestamps: list                           # (seconds, nanoseconds) per depth frame
    gt_poses: np.ndarray = field(default_factory=lambda: np.zeros((0, 4, 4), np.float32))
    gt_timestamps: list = field(default_factory=list)

    def seconds(self):
        return np.array([s + ns * 1e-9 for s, ns in self.timestamps])

    def gt_seconds(self):
        return np.array([s + ns * 1e-9 for s, ns in self.gt_timestamps])


# ------------------------------------------------------------------------------------------ PNG
def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    if pa <= pb and pa <= pc:
        return a
    return b if pb <= pc else c


def _decode_png_plain(data):
    """Minimal decoder (zlib + the five scanline filters) for non-interlaced 8/16-bit grey, RGB and RGBA files."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, header = 8, [], None
    while pos < len(data):
        (length,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if header is None:
        raise ValueError("PNG without header")
    w, h, depth, colour, _, _, interlace = header
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(colour)
    if channels is None or depth not in (8, 16) or interlace:
        raise ValueError("unsupported PNG variant (colour type %d, depth %d, interlace %d)" % (colour, depth, interlace))
    bpp = channels * depth // 8
    stride = w * bpp
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (stride + 1):
        raise ValueError("PNG data has an unexpected size")
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int64)
    for y in range(h):
        ftype = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int64)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 255
        elif ftype == 1:
            cur = line.reshape(-1, bpp).cumsum(axis=0).reshape(-1) & 255
        elif ftype in (3, 4):
            cur = line.tolist()
            up = prev.tolist()
            for i in range(stride):
                left = cur[i - bpp] if i >= bpp else 0
                if ftype == 3:
                    cur[i] = (cur[i] + ((left + up[i]) >> 1)) & 255
                else:
                    cur[i] = (cur[i] + _paeth(left, up[i], up[i - bpp] if i >= bpp else 0)) & 255
            cur = np.asarray(cur, np.int64)
        else:
            raise ValueError("bad PNG filter type %d" % ftype)
        out[y] = cur
        prev = cur
    if depth == 16:
        img = out.reshape(h, w, channels, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]                 # big endian samples
    else:
        img = out.reshape(h, w, channels)
    return img[..., 0] if channels == 1 else img


def decode_png(data):
    """uint8 / uint16 array [h, w] (grey) or [h, w, c] of a PNG file's bytes."""
    try:
        import io
        from PIL import Image
    except ImportError:
        return _decode_png_plain(data)
    img = Image.open(io.BytesIO(data))
    if img.mode in ("I;16", "I;16B", "I"):
        return np.asarray(img).astype(np.uint16)
    if img.mode in ("L", "RGB", "RGBA"):
        return np.asarray(img)
    return _decode_png_plain(data)


def encode_png_grey16(img):
    """A 16-bit greyscale PNG (filter 0) of a uint16 [h, w] array: what the TUM depth files are."""
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    lines = np.zeros((h, 1 + 2 * w), np.uint8)
    lines[:, 1:] = img.astype(">u2").view(np.uint8).reshape(h, 2 * w)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(lines.tobytes(), 6)) + chunk(b"IEND", b""))


# ------------------------------------------------------------------------------------------ shared pieces
_TS = re.compile(r"^([0-9]+)[.]([0-9]+)\s+(.*)$")
_COMMENT = re.compile(r"^\s*#.*$")


def _timestamp(sec, frac):
    return int(sec), int(frac) * 10 ** (9 - len(frac))


def quaternion_pose(tx, ty, tz, qx, qy, qz, qw):
    """Rotation matrix of (qw, qx, qy, qz) by the usual product formula in float32, without normalising
    (what Eigen's Quaternionf::toRotationMatrix does), and the translation."""
    f = np.float32
    x, y, z, w = f(qx), f(qy), f(qz), f(qw)
    tx2, ty2, tz2 = f(2) * x, f(2) * y, f(2) * z
    twx, twy, twz = tx2 * w, ty2 * w, tz2 * w
    txx, txy, txz = tx2 * x, ty2 * x, tz2 * x
    tyy, tyz, tzz = ty2 * y, tz2 * y, tz2 * z
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 1], m[0, 2] = f(1) - (tyy + tzz), txy - twz, txz + twy
    m[1, 0], m[1, 1], m[1, 2] = txy + twz, f(1) - (txx + tzz), tyz - twx
    m[2, 0], m[2, 1], m[2, 2] = txz - twy, tyz + twx, f(1) - (txx + tyy)
    m[:3, 3] = f(tx), f(ty), f(tz)
    return m


def _read_lines(path):
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line and not _COMMENT.match(line):
                yield line


# ------------------------------------------------------------------------------------------ TUM
def load_tum(dirname, frame_limit=0):
    for need in ("depth.txt", "depth", "groundtruth.txt"):
        if not os.path.exists(os.path.join(dirname, need)):
            raise FileNotFoundError("%s: %s is missing (not a TUM RGB-D folder)" % (dirname, need))
    key = next((k for k in ("freiburg1", "freiburg2", "freiburg3") if k in dirname), "default")
    depth, stamps = [], []
    for line in _read_lines(os.path.join(dirname, "depth.txt")):
        m = _TS.match(line)
        if not m:
            raise ValueError("depth.txt: unknown line %r" % line)
        with open(os.path.join(dirname, m.group(3)), "rb") as f:
            img = decode_png(f.read())
        if img.shape != (480, 640) or img.dtype != np.uint16:
            raise ValueError("%s: expected a 640x480 16-bit image" % m.group(3))
        depth.append(img)
        stamps.append(_timestamp(m.group(1), m.group(2)))
        if frame_limit and len(depth) >= frame_limit:
            break
    gt, gt_stamps = [], []
    for line in _read_lines(os.path.join(dirname, "groundtruth.txt")):
        parts = line.split()
        m = re.match(r"^([0-9]+)[.]([0-9]+)$", parts[0]) if len(parts) == 8 else None
        if not m:
            raise ValueError("groundtruth.txt: unknown line %r" % line)
        gt.append(quaternion_pose(*[float(v) for v in parts[1:]]))
        gt_stamps.append(_timestamp(m.group(1), m.group(2)))
    return DepthSequence(640, 480, TUM_DEPTH_INTRINSICS[key], 0.0002, 30.0, np.stack(depth), stamps,
                         np.stack(gt) if gt else np.zeros((0, 4, 4), np.float32), gt_stamps)


# ------------------------------------------------------------------------------------------ ICL-NUIM
def icl_planar_depth(ray_metres):
    """float32 [480, 640] distances along the ray -> uint16 millimetres along the optical axis, with the
    reference's two truncations (reference src/io/SLAMFrame.cpp:173-182 and ICLNUIM.cpp:96-116)."""
    d16 = (np.asarray(ray_metres, np.float32) * np.float32(1000)).astype(np.uint16)
    h, w = d16.shape
    uu = ((np.arange(w, dtype=np.uint32).astype(np.float32) - np.float32(319.5)) / np.float32(481.2)).astype(np.float64)
    vv = ((np.arange(h, dtype=np.uint32).astype(np.float32) - np.float32(239.5)) / np.float32(-480.0)).astype(np.float64)
    den = np.sqrt(uu[None, :] * uu[None, :] + vv[:, None] * vv[:, None] + 1)
    return (d16.astype(np.float64) / den).astype(np.uint16)


_KEY = re.compile(r"^([a-z_]+)")
_VEC = re.compile(r"([-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?), ([-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?), "
                  r"([-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?)")


def _normalised(v):
    v = np.asarray(v, np.float32)
    mag = np.float32(abs(np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
    return np.array([c / mag if c != 0 else c for c in v], np.float32)


class IclPoseReader:
    """Ground-truth pose of a scene_00_NNNN.txt file.  Stateful like the reference: with positive_focal the
    corrections are applied relative to the first pose read."""

    def __init__(self, positive_focal=False):
        self.positive_focal = positive_focal
        self.origin = None

    def pose(self, text):
        kv = {}
        for line in text.splitlines():
            k = _KEY.match(line)
            if not k:
                continue
            v = _VEC.search(line)
            if not v:
                if k.group(1) != "cam_angle":
                    raise ValueError("cannot read the camera line %r" % line)
                continue
            kv[k.group(1)] = np.array([float(v.group(i)) for i in (1, 2, 3)], np.float32)
        p = np.eye(4, dtype=np.float32)
        p[0, :3] = _normalised(kv["cam_right"])
        p[1, :3] = _normalised(kv["cam_up"])
        p[2, :3] = _normalised(kv["cam_dir"])
        p[:3, 3] = kv["cam_pos"]
        if self.positive_focal:
            p[0, 3] *= -1.0                                    # mirrored position
            if self.origin is None:
                self.origin = p.copy()
            o, oi = self.origin, np.linalg.inv(self.origin)
            p = oi @ p                                          # looking along the other z direction
            p[0, 2] *= -1.0; p[1, 2] *= -1.0; p[2, 0] *= -1.0; p[2, 1] *= -1.0
            p = o @ p
            p = oi @ p                                          # and upside down
            p[:2, :3] = -p[:2, :3]
            p = (o @ p).astype(np.float32)
        return p


def load_icl_nuim(dirname, positive_focal=False, frame_limit=0):
    eth = os.path.exists(os.path.join(dirname, "groundtruth.txt"))
    eth_lines = list(_read_lines(os.path.join(dirname, "groundtruth.txt"))) if eth else []
    poses = IclPoseReader(positive_focal)
    depth, stamps, gt = [], [], []
    n = 0
    while not frame_limit or n < frame_limit:
        base = os.path.join(dirname, "scene_00_%04d" % n)
        dpath = base + (".depth.png" if eth else ".depth")
        if not os.path.exists(dpath):
            break
        if eth:
            with open(dpath, "rb") as f:
                img = decode_png(f.read())
            if n >= len(eth_lines):
                break
            parts = eth_lines[n].split()
            if len(parts) != 8:
                raise ValueError("groundtruth.txt: unknown line %r" % eth_lines[n])
            gt.append(quaternion_pose(*[float(v) for v in parts[1:]]))
        else:
            if not os.path.exists(base + ".txt"):
                break
            with open(dpath) as f:
                values = np.array(f.read().split(), dtype=np.float32)
            if values.size != 640 * 480:
                raise ValueError("%s holds %d values, expected %d" % (dpath, values.size, 640 * 480))
            img = icl_planar_depth(values.reshape(480, 640))
            with open(base + ".txt") as f:
                gt.append(poses.pose(f.read()))
        if img.shape != (480, 640):
            raise ValueError("%s: expected a 640x480 image" % dpath)
        depth.append(img.astype(np.uint16))
        total_ns = int((1.0 / 25) * n * 1000000000)
        stamps.append((total_ns // 1000000000, total_ns % 1000000000))
        n += 1
    if not depth:
        raise FileNotFoundError("%s holds no scene_00_NNNN depth files" % dirname)
    intr = (ICL_INTRINSICS[0], 480.0 if positive_focal else -480.0, ICL_INTRINSICS[2], ICL_INTRINSICS[3])
    # The text files hold metres and are stored as millimetres: 0.001 m per unit.  (The reference writes 0.0002
    # into the sensor block of both layouts, include/ICLNUIM.h:35; pass disparity_scale to `to_slam` to copy that.)
    return DepthSequence(640, 480, intr, 0.0002 if eth else 0.001, 1.0, np.stack(depth), stamps, np.stack(gt), list(stamps))


# ------------------------------------------------------------------------------------------ .slam output
def to_slam(seq, path, disparity_scale=None):
    """Ground truth first, then the depth frames, each group in time order (the order the reference's TUM
    generator adds them in)."""
    fx, fy, cx, cy = seq.intrinsics_px
    sensors = [slamfile.Sensor("GroundTruth", slamfile.GROUND_TRUTH, description="GroundTruthSensor", rate=seq.rate),
               slamfile.Sensor("Depth", slamfile.DEPTH, description="Depth", rate=seq.rate,
                               pixel_format=slamfile.PIXEL_FORMATS["D_I_16"], width=seq.width, height=seq.height,
                               intrinsics=(fx / seq.width, fy / seq.height, cx / seq.width, cy / seq.height),
                               disparity_params=(seq.disparity_scale if disparity_scale is None else disparity_scale, 0.0))]

    def frames():
        for ts, pose in zip(seq.gt_timestamps, seq.gt_poses):
            yield ts, 0, np.asarray(pose, np.float32).T.tobytes()
        for ts, img in zip(seq.timestamps, seq.depth):
            yield ts, 1, img.tobytes()

    slamfile.write_slam(path, sensors, frames())
    return sensors
