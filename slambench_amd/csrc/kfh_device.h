// Device-side vector/matrix helpers shared by the gfx950 KinectFusion kernels.
//
// All code in csrc/ is compiled with -ffp-contract=off: every expression below is a sequence of
// IEEE-754 binary32 operations in the written order (no fused multiply-add), so index-critical
// results (pixel coordinates, voxel positions) match a plain C++ evaluation bit for bit.
#ifndef KFH_DEVICE_H_
#define KFH_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KFH_INVALID (-2.0f)
#define KFH_SHORT2FLOAT 0.00003051944088f
#define KFH_FLOAT2SHORT 32766.0f

namespace kfh {

struct f3 { float x, y, z; };
struct Mat4 { float m[16]; };   // row-major

__host__ __device__ inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__host__ __device__ inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__host__ __device__ inline f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__host__ __device__ inline f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__host__ __device__ inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__host__ __device__ inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
__host__ __device__ inline f3 normalize3(f3 v) {
    float inv_len = 1.0f / sqrtf(dot3(v, v));
    return v * inv_len;
}
// affine transform (rotation rows + translation column)
__host__ __device__ inline f3 xform(const Mat4 &M, f3 v) {
    return mk3(dot3(mk3(M.m[0], M.m[1], M.m[2]), v) + M.m[3],
               dot3(mk3(M.m[4], M.m[5], M.m[6]), v) + M.m[7],
               dot3(mk3(M.m[8], M.m[9], M.m[10]), v) + M.m[11]);
}
// linear part only
__host__ __device__ inline f3 rotate(const Mat4 &M, f3 v) {
    return mk3(dot3(mk3(M.m[0], M.m[1], M.m[2]), v), dot3(mk3(M.m[4], M.m[5], M.m[6]), v),
               dot3(mk3(M.m[8], M.m[9], M.m[10]), v));
}

__host__ __device__ inline Mat4 mat_mul(const Mat4 &A, const Mat4 &B) {
    Mat4 R;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = A.m[i * 4 + 0] * B.m[0 * 4 + j];
            s = s + A.m[i * 4 + 1] * B.m[1 * 4 + j];
            s = s + A.m[i * 4 + 2] * B.m[2 * 4 + j];
            s = s + A.m[i * 4 + 3] * B.m[3 * 4 + j];
            R.m[i * 4 + j] = s;
        }
    return R;
}

// 4x4 inverse by Gauss-Jordan elimination with partial pivoting, in float.
__host__ __device__ inline Mat4 mat_inverse(const Mat4 &A) {
    float a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = A.m[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0f : 0.0f;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabsf(a[r][c]) > fabsf(a[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 8; ++j) { float t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        float inv = 1.0f / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] = a[c][j] * inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            float f = a[r][c];
            if (f != 0.0f)
                for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[c][j];
        }
    }
    Mat4 R;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) R.m[i * 4 + j] = a[i][4 + j];
    return R;
}

__host__ __device__ inline Mat4 camera_matrix(float fx, float fy, float cx, float cy) {
    Mat4 K = {{fx, 0, cx, 0, 0, fy, cy, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    return K;
}
__host__ __device__ inline Mat4 inverse_camera_matrix(float fx, float fy, float cx, float cy) {
    Mat4 K = {{1.0f / fx, 0, -cx / fx, 0, 0, 1.0f / fy, -cy / fy, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    return K;
}

}  // namespace kfh
#endif
