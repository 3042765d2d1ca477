"""KFusion plugin parameters and the C-ABI configuration record.

Parameter names and defaults are those the reference's kfusion plugin prints in its log header
(reference icra2018_results/1080/violons_libkfusion-cpp_living_room_traj0_loop.log:20-31):
compute-size-ratio 1, integration-rate 2, tracking-rate 1, rendering-rate 4, icp-threshold 1e-05,
mu 0.1, volume-size 8,8,8, volume-direction 4,4,4, volume-resolution 256,256,256,
pyramid-level1..3 10/5/4.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Tuple

# Sensor intrinsics as the reference's dataset tools define them (in pixels at 640x480).
ICL_NUIM_INTRINSICS = (481.2, 480.0, 319.5, 239.5)   # reference framework/tools/dataset-tools/ICLNUIM.cpp:43-47
TUM_FR1_DEPTH_INTRINSICS = (591.1, 590.1, 331.0, 234.0)  # reference framework/tools/dataset-tools/include/TUM.h:99-101


class KfhConfig(C.Structure):
    """Mirror of struct kfh_config (include/kfusion_hip.h)."""
    _fields_ = [
        ("input_width", C.c_int32), ("input_height", C.c_int32),
        ("compute_size_ratio", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("depth_divisor", C.c_float),
        ("vol_res", C.c_int32 * 3), ("vol_dim", C.c_float * 3), ("init_pos", C.c_float * 3),
        ("mu", C.c_float), ("icp_threshold", C.c_float),
        ("pyramid", C.c_int32 * 3),
        ("integration_rate", C.c_int32), ("tracking_rate", C.c_int32),
        ("integrate_filtered", C.c_int32),
        ("slab_z0", C.c_int32), ("slab_z1", C.c_int32),
    ]


@dataclass
class KFusionConfig:
    """Host-side parameter set; field names follow the plugin's long option names."""
    input_size: Tuple[int, int] = (640, 480)                  # width, height of the depth sensor
    intrinsics: Tuple[float, float, float, float] = ICL_NUIM_INTRINSICS  # fx, fy, cx, cy in input pixels
    depth_divisor: float = 1000.0                              # raw units per metre (1/DisparityParams[0])
    compute_size_ratio: int = 1
    integration_rate: int = 2
    tracking_rate: int = 1
    rendering_rate: int = 4
    icp_threshold: float = 1e-5
    mu: float = 0.1
    volume_size: Tuple[float, float, float] = (8.0, 8.0, 8.0)
    volume_direction: Tuple[float, float, float] = (4.0, 4.0, 4.0)
    volume_resolution: Tuple[int, int, int] = (256, 256, 256)
    pyramid: Tuple[int, int, int] = (10, 5, 4)                # pyramid-level1 (finest) .. level3
    integrate_filtered: bool = False                           # fuse the raw metres depth (default) or the bilateral-filtered one
    slab: Tuple[int, int] = field(default=(0, 0))              # z range owned by this device; (0, 0) = whole volume

    def validate(self):
        w, h = self.input_size
        r = self.compute_size_ratio
        if r not in (1, 2, 4, 8):
            raise ValueError("compute-size-ratio must be 1, 2, 4 or 8")
        if w % (r * 4) or h % (r * 4):
            raise ValueError("input size must be divisible by 4*compute-size-ratio (3 pyramid levels)")
        if any(v <= 0 for v in self.volume_resolution) or any(v <= 0 for v in self.volume_size):
            raise ValueError("volume resolution and size must be positive")
        if self.integration_rate < 1 or self.tracking_rate < 1:
            raise ValueError("rates must be >= 1")
        if len(self.pyramid) != 3 or any(v < 0 for v in self.pyramid):
            raise ValueError("three non-negative pyramid iteration counts are required")
        z0, z1 = self.slab
        if (z0, z1) != (0, 0) and not (0 <= z0 < z1 <= self.volume_resolution[2]):
            raise ValueError("slab must satisfy 0 <= z0 < z1 <= volume_resolution[2]")

    @property
    def compute_size(self):
        return (self.input_size[0] // self.compute_size_ratio, self.input_size[1] // self.compute_size_ratio)

    @property
    def camera(self):
        """Intrinsics at compute size: the plugin divides (fx, fy, cx, cy) by compute-size-ratio."""
        r = float(self.compute_size_ratio)
        return tuple(v / r for v in self.intrinsics)

    def to_c(self, struct_type=KfhConfig):
        self.validate()
        c = struct_type()
        c.input_width, c.input_height = self.input_size
        c.compute_size_ratio = self.compute_size_ratio
        c.fx, c.fy, c.cx, c.cy = self.intrinsics
        c.depth_divisor = self.depth_divisor
        c.vol_res[:] = list(self.volume_resolution)
        c.vol_dim[:] = list(self.volume_size)
        c.init_pos[:] = list(self.volume_direction)
        c.mu = self.mu
        c.icp_threshold = self.icp_threshold
        c.pyramid[:] = list(self.pyramid)
        c.integration_rate = self.integration_rate
        c.tracking_rate = self.tracking_rate
        c.integrate_filtered = int(self.integrate_filtered)
        z0, z1 = self.slab
        if (z0, z1) == (0, 0):
            z1 = self.volume_resolution[2]
        c.slab_z0, c.slab_z1 = z0, z1
        return c
