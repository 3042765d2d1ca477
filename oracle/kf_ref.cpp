/*
 * kf_ref.cpp — CPU restatement of the SLAMBench `kfusion` C++ back-end.
 *
 * TEST INFRASTRUCTURE ONLY (see kf_ref.h).  PARITY UNPINNED: the kfusion sources are not
 * vendored in the reference tree (benchmarks/benchmarks.repos:3), so no function below can
 * cite a reference file:line for its arithmetic; each cites the upstream routine it restates
 * (pamela-project/kfusion, kfusion/src/cpp/kernels.cpp, kfusion/include/commons.h) and, where
 * one exists, the in-tree evidence that constrains it.
 *
 * Every float expression is written in the upstream evaluation order and the file is built
 * with -ffp-contract=off, so the results are plain IEEE-754 binary32 operations with no fused
 * multiply-add: the HIP kernels are built the same way and can be compared bit for bit
 * wherever no transcendental function is involved.
 */
#include "kf_ref.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr float KF_INVALID = -2.0f;
constexpr float SHORT2FLOAT = 0.00003051944088f;
constexpr float FLOAT2SHORT = 32766.0f;

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct M4 { f4 r[4]; };

inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
inline f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
inline f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
inline f3 normalize3(f3 v) {
    float inv_len = 1.0f / sqrtf(dot3(v, v));
    return v * inv_len;
}
inline f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }
/* commons.h operator*(Matrix4, float3): affine transform */
inline f3 xform(const M4 &M, f3 v) {
    return mk3(dot3(xyz(M.r[0]), v) + M.r[0].w,
               dot3(xyz(M.r[1]), v) + M.r[1].w,
               dot3(xyz(M.r[2]), v) + M.r[2].w);
}
/* commons.h rotate(Matrix4, float3): linear part only */
inline f3 rotate(const M4 &M, f3 v) {
    return mk3(dot3(xyz(M.r[0]), v), dot3(xyz(M.r[1]), v), dot3(xyz(M.r[2]), v));
}
inline M4 load_m4(const float *p) { M4 m; std::memcpy(&m, p, sizeof(M4)); return m; }
inline void store_m4(float *p, const M4 &m) { std::memcpy(p, &m, sizeof(M4)); }

inline M4 mat_mul(const M4 &A, const M4 &B) {
    M4 R;
    const float *a = &A.r[0].x, *b = &B.r[0].x;
    float *r = &R.r[0].x;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a[i * 4 + 0] * b[0 * 4 + j];
            s = s + a[i * 4 + 1] * b[1 * 4 + j];
            s = s + a[i * 4 + 2] * b[2 * 4 + j];
            s = s + a[i * 4 + 3] * b[3 * 4 + j];
            r[i * 4 + j] = s;
        }
    return R;
}

/* commons.h inverse(Matrix4): TooN::gaussian_elimination(A, I) in float, partial pivoting. */
inline M4 mat_inverse(const M4 &A) {
    float a[4][8];
    const float *src = &A.r[0].x;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = src[i * 4 + j];
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
    M4 R;
    float *dst = &R.r[0].x;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) dst[i * 4 + j] = a[i][4 + j];
    return R;
}

/* commons.h getCameraMatrix / getInverseCameraMatrix, k = (fx, fy, cx, cy) */
inline M4 camera_matrix(float fx, float fy, float cx, float cy) {
    M4 K;
    K.r[0] = f4{fx, 0, cx, 0};
    K.r[1] = f4{0, fy, cy, 0};
    K.r[2] = f4{0, 0, 1, 0};
    K.r[3] = f4{0, 0, 0, 1};
    return K;
}
inline M4 inverse_camera_matrix(float fx, float fy, float cx, float cy) {
    M4 K;
    K.r[0] = f4{1.0f / fx, 0, -cx / fx, 0};
    K.r[1] = f4{0, 1.0f / fy, -cy / fy, 0};
    K.r[2] = f4{0, 0, 1, 0};
    K.r[3] = f4{0, 0, 0, 1};
    return K;
}

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float clampf(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); }

/* Volume accessors: commons.h struct Volume — short2 voxels, x fastest. */
struct VolView {
    const int16_t *data;
    int sx, sy, sz;
    float dx, dy, dz;
    inline float vs(int x, int y, int z) const {
        return (float)data[2 * ((size_t)x + (size_t)y * sx + (size_t)z * sx * sy)];
    }
    /* Volume::interp — trilinear sample of the raw shorts, scaled once at the end */
    float interp(f3 pos) const {
        const f3 sp = mk3((pos.x * (float)sx / dx) - 0.5f, (pos.y * (float)sy / dy) - 0.5f,
                          (pos.z * (float)sz / dz) - 0.5f);
        const float flx = floorf(sp.x), fly = floorf(sp.y), flz = floorf(sp.z);
        const int bx = (int)flx, by = (int)fly, bz = (int)flz;
        const f3 f = mk3(sp.x - flx, sp.y - fly, sp.z - flz);
        const int lx = bx > 0 ? bx : 0, ly = by > 0 ? by : 0, lz = bz > 0 ? bz : 0;
        const int ux = (bx + 1 < sx - 1) ? bx + 1 : sx - 1;
        const int uy = (by + 1 < sy - 1) ? by + 1 : sy - 1;
        const int uz = (bz + 1 < sz - 1) ? bz + 1 : sz - 1;
        return (((vs(lx, ly, lz) * (1 - f.x) + vs(ux, ly, lz) * f.x) * (1 - f.y) +
                 (vs(lx, uy, lz) * (1 - f.x) + vs(ux, uy, lz) * f.x) * f.y) * (1 - f.z) +
                ((vs(lx, ly, uz) * (1 - f.x) + vs(ux, ly, uz) * f.x) * (1 - f.y) +
                 (vs(lx, uy, uz) * (1 - f.x) + vs(ux, uy, uz) * f.x) * f.y) * f.z) * SHORT2FLOAT;
    }
    /* Volume::grad — central differences blended trilinearly */
    f3 grad(f3 pos) const {
        const f3 sp = mk3((pos.x * (float)sx / dx) - 0.5f, (pos.y * (float)sy / dy) - 0.5f,
                          (pos.z * (float)sz / dz) - 0.5f);
        const float flx = floorf(sp.x), fly = floorf(sp.y), flz = floorf(sp.z);
        const int bx = (int)flx, by = (int)fly, bz = (int)flz;
        const f3 f = mk3(sp.x - flx, sp.y - fly, sp.z - flz);
        auto lo0 = [](int v) { return v > 0 ? v : 0; };
        const int llx = lo0(bx - 1), lly = lo0(by - 1), llz = lo0(bz - 1);
        const int lux = lo0(bx), luy = lo0(by), luz = lo0(bz);
        const int ulx = (bx + 1 < sx - 1) ? bx + 1 : sx - 1;
        const int uly = (by + 1 < sy - 1) ? by + 1 : sy - 1;
        const int ulz = (bz + 1 < sz - 1) ? bz + 1 : sz - 1;
        const int uux = (bx + 2 < sx - 1) ? bx + 2 : sx - 1;
        const int uuy = (by + 2 < sy - 1) ? by + 2 : sy - 1;
        const int uuz = (bz + 2 < sz - 1) ? bz + 2 : sz - 1;
        const int lx = lux, ly = luy, lz = luz;
        const int ux = ulx, uy = uly, uz = ulz;
        f3 g;
        g.x = (((vs(ulx, ly, lz) - vs(llx, ly, lz)) * (1 - f.x) +
                (vs(uux, ly, lz) - vs(lux, ly, lz)) * f.x) * (1 - f.y) +
               ((vs(ulx, uy, lz) - vs(llx, uy, lz)) * (1 - f.x) +
                (vs(uux, uy, lz) - vs(lux, uy, lz)) * f.x) * f.y) * (1 - f.z) +
              (((vs(ulx, ly, uz) - vs(llx, ly, uz)) * (1 - f.x) +
                (vs(uux, ly, uz) - vs(lux, ly, uz)) * f.x) * (1 - f.y) +
               ((vs(ulx, uy, uz) - vs(llx, uy, uz)) * (1 - f.x) +
                (vs(uux, uy, uz) - vs(lux, uy, uz)) * f.x) * f.y) * f.z;
        g.y = (((vs(lx, uly, lz) - vs(lx, lly, lz)) * (1 - f.x) +
                (vs(ux, uly, lz) - vs(ux, lly, lz)) * f.x) * (1 - f.y) +
               ((vs(lx, uuy, lz) - vs(lx, luy, lz)) * (1 - f.x) +
                (vs(ux, uuy, lz) - vs(ux, luy, lz)) * f.x) * f.y) * (1 - f.z) +
              (((vs(lx, uly, uz) - vs(lx, lly, uz)) * (1 - f.x) +
                (vs(ux, uly, uz) - vs(ux, lly, uz)) * f.x) * (1 - f.y) +
               ((vs(lx, uuy, uz) - vs(lx, luy, uz)) * (1 - f.x) +
                (vs(ux, uuy, uz) - vs(ux, luy, uz)) * f.x) * f.y) * f.z;
        g.z = (((vs(lx, ly, ulz) - vs(lx, ly, llz)) * (1 - f.x) +
                (vs(ux, ly, ulz) - vs(ux, ly, llz)) * f.x) * (1 - f.y) +
               ((vs(lx, uy, ulz) - vs(lx, uy, llz)) * (1 - f.x) +
                (vs(ux, uy, ulz) - vs(ux, uy, llz)) * f.x) * f.y) * (1 - f.z) +
              (((vs(lx, ly, uuz) - vs(lx, ly, luz)) * (1 - f.x) +
                (vs(ux, ly, uuz) - vs(ux, ly, luz)) * f.x) * (1 - f.y) +
               ((vs(lx, uy, uuz) - vs(lx, uy, luz)) * (1 - f.x) +
                (vs(ux, uy, uuz) - vs(ux, uy, luz)) * f.x) * f.y) * f.z;
        return g * mk3(dx / (float)sx, dy / (float)sy, dz / (float)sz) * (0.5f * SHORT2FLOAT);
    }
};

/* kernels.cpp raycast(): march one pixel's ray; returns (hit.xyz, t) or zeros */
inline f4 raycast_one(const VolView &vol, int px, int py, const M4 &view, float near_plane,
                      float far_plane, float step, float largestep) {
    const f3 origin = mk3(view.r[0].w, view.r[1].w, view.r[2].w);
    const f3 direction = rotate(view, mk3((float)px, (float)py, 1.f));

    const f3 invR = mk3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
    const f3 tbot = -1.0f * invR * origin;
    const f3 ttop = invR * (mk3(vol.dx, vol.dy, vol.dz) - origin);

    const f3 tmin = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    const f3 tmax = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));

    const float largest_tmin = fmaxf(fmaxf(tmin.x, tmin.y), fmaxf(tmin.x, tmin.z));
    const float smallest_tmax = fminf(fminf(tmax.x, tmax.y), fminf(tmax.x, tmax.z));

    const float tnear = fmaxf(largest_tmin, near_plane);
    const float tfar = fminf(smallest_tmax, far_plane);

    if (tnear < tfar) {
        float t = tnear;
        float stepsize = largestep;
        float f_t = vol.interp(origin + direction * t);
        float f_tt = 0;
        if (f_t > 0) {
            for (; t < tfar; t += stepsize) {
                f_tt = vol.interp(origin + direction * t);
                if (f_tt < 0) break;
                if (f_tt < 0.8f) stepsize = step;
                f_t = f_tt;
            }
            if (f_tt < 0) {
                t = t + stepsize * f_tt / (f_t - f_tt);
                const f3 p = origin + direction * t;
                return f4{p.x, p.y, p.z, t};
            }
        }
    }
    return f4{0, 0, 0, 0};
}

/* Symmetric 6x6 eigen-decomposition by cyclic Jacobi rotations, double precision.
 * For a symmetric matrix the singular values are |eigenvalues| and U = V up to sign, so
 * TooN's GR_SVD<6,6>::backsub(b, 1e6) equals  sum_i [ |l_i|*1e6 > max|l| ] (v_i.b / l_i) v_i . */
void jacobi_eig6(double A[6][6], double V[6][6], double w[6]) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 6; ++i) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < 6; ++j) off += A[i][j] * A[i][j];
        }
        if (off == 0.0 || off <= 1e-40 * diag) break;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 6; ++i) w[i] = A[i][i];
}

}  // namespace

extern "C" {

/* kernels.cpp mm2metersKernel: sub-sample by the integer size ratio, divide by 1000.0f.
 * The divisor is a parameter because the harness carries the unit in DepthSensor::DisparityParams
 * (reference framework/tools/dataset-tools/include/TUM.h:52-53, ICLNUIM.h:35). */
void kfr_mm2meters(float *out, int ow, int oh, const uint16_t *in, int iw, int ih, float divisor) {
    const int ratio = iw / ow;
    (void)ih;
#pragma omp parallel for
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x)
            out[x + ow * y] = (float)in[x * ratio + iw * y * ratio] / divisor;
}

/* kernels.cpp bilateralFilterKernel (radius 2, gaussian delta 4.0, e_d 0.1 in Kfusion) */
void kfr_bilateral(float *out, const float *in, int w, int h, int r, float delta, float e_d) {
    std::vector<float> gaussian(2 * r + 1);
    for (int i = 0; i < 2 * r + 1; ++i) {
        int x = i - r;
        gaussian[i] = expf(-(float)(x * x) / (2 * delta * delta));
    }
    const float e_d_squared_2 = e_d * e_d * 2;
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int pos = x + y * w;
            if (in[pos] == 0) { out[pos] = 0; continue; }
            float sum = 0.0f, t = 0.0f;
            const float center = in[pos];
            for (int i = -r; i <= r; ++i)
                for (int j = -r; j <= r; ++j) {
                    const int cx = clampi(x + i, 0, w - 1), cy = clampi(y + j, 0, h - 1);
                    const float cur = in[cx + cy * w];
                    if (cur > 0) {
                        const float mod = (cur - center) * (cur - center);
                        const float factor = gaussian[i + r] * gaussian[j + r] * expf(-mod / e_d_squared_2);
                        t += factor * cur;
                        sum += factor;
                    }
                }
            out[pos] = t / sum;
        }
}

/* kernels.cpp halfSampleRobustImageKernel (Kfusion calls it with e_d = 0.3, r = 1) */
void kfr_half_sample(float *out, const float *in, int iw, int ih, float e_d, int r) {
    const int ow = iw / 2, oh = ih / 2;
#pragma omp parallel for
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            const int cx = 2 * x, cy = 2 * y;
            float sum = 0.0f, t = 0.0f;
            const float center = in[cx + cy * iw];
            for (int i = -r + 1; i <= r; ++i)
                for (int j = -r + 1; j <= r; ++j) {
                    const int px = clampi(cx + j, 0, 2 * ow - 1), py = clampi(cy + i, 0, 2 * oh - 1);
                    const float cur = in[px + py * iw];
                    if (fabsf(cur - center) < e_d) { sum += 1.0f; t += cur; }
                }
            out[x + y * ow] = t / sum;
        }
}

/* kernels.cpp depth2vertexKernel */
void kfr_depth2vertex(float *vertex3, const float *depth, int w, int h, const float invK_[16]) {
    const M4 invK = load_m4(invK_);
    f3 *vertex = reinterpret_cast<f3 *>(vertex3);
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float d = depth[x + y * w];
            vertex[x + y * w] = d > 0 ? d * rotate(invK, mk3((float)x, (float)y, 1.f)) : mk3(0, 0, 0);
        }
}

/* kernels.cpp vertex2normalKernel */
void kfr_vertex2normal(float *normal3, const float *vertex3, int w, int h) {
    const f3 *in = reinterpret_cast<const f3 *>(vertex3);
    f3 *out = reinterpret_cast<f3 *>(normal3);
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const f3 left = in[(x - 1 > 0 ? x - 1 : 0) + y * w];
            const f3 right = in[(x + 1 < w - 1 ? x + 1 : w - 1) + y * w];
            const f3 up = in[x + (y - 1 > 0 ? y - 1 : 0) * w];
            const f3 down = in[x + (y + 1 < h - 1 ? y + 1 : h - 1) * w];
            if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) {
                out[x + y * w] = mk3(KF_INVALID, 0, 0);
                continue;
            }
            const f3 dxv = right - left;
            const f3 dyv = down - up;
            out[x + y * w] = normalize3(cross3(dyv, dxv));
        }
}

/* kernels.cpp trackKernel */
void kfr_track(kfr_track_data *out, int out_stride, const float *in_vertex3, const float *in_normal3,
               int iw, int ih, const float *ref_vertex3, const float *ref_normal3, int rw, int rh,
               const float Ttrack_[16], const float view_[16], float dist_threshold,
               float normal_threshold) {
    const M4 Ttrack = load_m4(Ttrack_), view = load_m4(view_);
    const f3 *inV = reinterpret_cast<const f3 *>(in_vertex3);
    const f3 *inN = reinterpret_cast<const f3 *>(in_normal3);
    const f3 *refV = reinterpret_cast<const f3 *>(ref_vertex3);
    const f3 *refN = reinterpret_cast<const f3 *>(ref_normal3);
#pragma omp parallel for
    for (int y = 0; y < ih; ++y)
        for (int x = 0; x < iw; ++x) {
            kfr_track_data &row = out[x + y * out_stride];
            if (inN[x + y * iw].x == KF_INVALID) { row.result = -1; continue; }
            const f3 projectedVertex = xform(Ttrack, inV[x + y * iw]);
            const f3 projectedPos = xform(view, projectedVertex);
            const float px = projectedPos.x / projectedPos.z + 0.5f;
            const float py = projectedPos.y / projectedPos.z + 0.5f;
            if (px < 0 || px > rw - 1 || py < 0 || py > rh - 1) { row.result = -2; continue; }
            const unsigned rx = (unsigned)px, ry = (unsigned)py;
            const f3 referenceNormal = refN[rx + ry * rw];
            if (referenceNormal.x == KF_INVALID) { row.result = -3; continue; }
            const f3 diff = refV[rx + ry * rw] - projectedVertex;
            const f3 projectedNormal = rotate(Ttrack, inN[x + y * iw]);
            if (length3(diff) > dist_threshold) { row.result = -4; continue; }
            if (dot3(projectedNormal, referenceNormal) < normal_threshold) { row.result = -5; continue; }
            row.result = 1;
            row.error = dot3(referenceNormal, diff);
            const f3 c = cross3(projectedVertex, referenceNormal);
            row.J[0] = referenceNormal.x; row.J[1] = referenceNormal.y; row.J[2] = referenceNormal.z;
            row.J[3] = c.x; row.J[4] = c.y; row.J[5] = c.z;
        }
}

/* kernels.cpp new_reduce + reduceKernel: 8 row-strided partials of 32 floats, then rows 1..7
 * added onto row 0.  Layout of a row: [0] sum e^2, [1..6] sum e*J, [7..27] upper triangle of
 * sum J J^T (row-major), [28] valid, [29] result==-4, [30] result==-5, [31] result>-4 (and <1). */
static int g_reduce_rows = 8;       /* 8 is the reference's layout; tests pick another summation order with fewer */
void kfr_set_reduce_rows(int rows) { g_reduce_rows = rows < 1 ? 1 : rows > 8 ? 8 : rows; }

void kfr_reduce(float out[8 * 32], const kfr_track_data *J, int j_stride, int w, int h) {
    const int rows = g_reduce_rows;
    for (int i = 0; i < 8 * 32; ++i) out[i] = 0.0f;
#pragma omp parallel for
    for (int b = 0; b < rows; ++b) {
        float s[32];
        for (int i = 0; i < 32; ++i) s[i] = 0.0f;
        for (int y = b; y < h; y += rows)
            for (int x = 0; x < w; ++x) {
                const kfr_track_data &row = J[x + y * j_stride];
                if (row.result < 1) {
                    s[29] += row.result == -4 ? 1 : 0;
                    s[30] += row.result == -5 ? 1 : 0;
                    s[31] += row.result > -4 ? 1 : 0;
                    continue;
                }
                s[0] += row.error * row.error;
                for (int i = 0; i < 6; ++i) s[1 + i] += row.error * row.J[i];
                int k = 7;
                for (int i = 0; i < 6; ++i)
                    for (int j = i; j < 6; ++j) s[k++] += row.J[i] * row.J[j];
                s[28] += 1;
            }
        for (int i = 0; i < 32; ++i) out[b * 32 + i] = s[i];
    }
    for (int b = 1; b < 8; ++b)
        for (int i = 0; i < 32; ++i) out[i] += out[b * 32 + i];
}

/* kernels.cpp solve(): b = vals[0..5], C from the upper triangle vals[6..26];
 * TooN GR_SVD<6,6>::backsub(b, 1e6) in double. */
static thread_local double g_cutoff_margin = 1e300;   /* of the last kfr_solve on this thread */
double kfr_last_cutoff_margin(void) { return g_cutoff_margin; }

void kfr_solve(double x[6], const float vals[27]) {
    double b[6], C[6][6], V[6][6], w[6];
    for (int i = 0; i < 6; ++i) b[i] = vals[i];
    int k = 6;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { C[i][j] = vals[k]; C[j][i] = vals[k]; ++k; }
    jacobi_eig6(C, V, w);
    double wmax = 0.0;
    for (int i = 0; i < 6; ++i) wmax = fmax(wmax, fabs(w[i]));
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
    /* diagnostic: how close the nearest keep/drop decision came to its limit, |ln(|w_i| * 1e6 / wmax)| */
    g_cutoff_margin = 1e300;
    for (int i = 0; i < 6; ++i)
        if (fabs(w[i]) > 0.0 && wmax > 0.0) g_cutoff_margin = fmin(g_cutoff_margin, fabs(log(fabs(w[i]) * 1e6 / wmax)));
    for (int i = 0; i < 6; ++i) {
        if (!(fabs(w[i]) * 1e6 > wmax)) continue;
        double p = 0.0;
        for (int j = 0; j < 6; ++j) p += V[j][i] * b[j];
        p /= w[i];
        for (int j = 0; j < 6; ++j) x[j] += p * V[j][i];
    }
}

/* TooN SE3<double>::exp(mu) with mu = (translation, rotation), then commons.h toMatrix4 (float) */
void kfr_se3_exp(float out[16], const double mu[6]) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double w0 = mu[3], w1 = mu[4], w2 = mu[5];
    const double theta_sq = w0 * w0 + w1 * w1 + w2 * w2;
    const double theta = sqrt(theta_sq);
    double A, B;
    const double cr[3] = {w1 * mu[2] - w2 * mu[1], w2 * mu[0] - w0 * mu[2], w0 * mu[1] - w1 * mu[0]};
    double t[3];
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
        for (int i = 0; i < 3; ++i) t[i] = mu[i] + 0.5 * cr[i];
    } else {
        double C;
        if (theta_sq < 1e-6) {
            C = one_6th * (1.0 - one_20th * theta_sq);
            A = 1.0 - theta_sq * C;
            B = 0.5 - 0.25 * one_6th * theta_sq;
        } else {
            const double inv_theta = 1.0 / theta;
            A = sin(theta) * inv_theta;
            B = (1 - cos(theta)) * (inv_theta * inv_theta);
            C = (1 - A) * (inv_theta * inv_theta);
        }
        const double wc[3] = {w1 * cr[2] - w2 * cr[1], w2 * cr[0] - w0 * cr[2], w0 * cr[1] - w1 * cr[0]};
        for (int i = 0; i < 3; ++i) t[i] = mu[i] + B * cr[i] + C * wc[i];
    }
    double R[3][3];
    {
        const double wx2 = w0 * w0, wy2 = w1 * w1, wz2 = w2 * w2;
        R[0][0] = 1.0 - B * (wy2 + wz2);
        R[1][1] = 1.0 - B * (wx2 + wz2);
        R[2][2] = 1.0 - B * (wx2 + wy2);
    }
    { const double a = A * w2, b = B * (w0 * w1); R[0][1] = b - a; R[1][0] = b + a; }
    { const double a = A * w1, b = B * (w0 * w2); R[0][2] = b + a; R[2][0] = b - a; }
    { const double a = A * w0, b = B * (w1 * w2); R[1][2] = b - a; R[2][1] = b + a; }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out[i * 4 + j] = (float)R[i][j];
        out[i * 4 + 3] = (float)t[i];
    }
    out[12] = 0; out[13] = 0; out[14] = 0; out[15] = 1;
}

/* kernels.cpp updatePoseKernel: pose = exp(x) * pose; converged when |x| < icp_threshold */
int kfr_update_pose(float pose[16], const float sums[32], float icp_threshold) {
    double x[6];
    kfr_solve(x, sums + 1);
    float delta[16];
    kfr_se3_exp(delta, x);
    const M4 np = mat_mul(load_m4(delta), load_m4(pose));
    store_m4(pose, np);
    double n2 = 0.0;
    for (int i = 0; i < 6; ++i) n2 += x[i] * x[i];
    return sqrt(n2) < icp_threshold ? 1 : 0;
}

/* kernels.cpp checkPoseKernel (track_threshold 0.15 in Kfusion) */
int kfr_check_pose(float pose[16], const float old_pose[16], const float sums[32], int w, int h,
                   float track_threshold) {
    if ((sqrtf(sums[0] / sums[28]) > 2e-2f) || (sums[28] / (float)(w * h) < track_threshold)) {
        std::memcpy(pose, old_pose, 16 * sizeof(float));
        return 0;
    }
    return 1;
}

/* kernels.cpp integrateKernel: per (x,y) column, z marched with incrementally accumulated
 * camera-space positions (pos += delta, cameraX += cameraDelta). */
void kfr_integrate(int16_t *vol, const int32_t res[3], const float dim[3], const float *depth,
                   int w, int h, const float invTrack_[16], const float K_[16], float mu,
                   float maxweight) {
    const M4 invTrack = load_m4(invTrack_), K = load_m4(K_);
    const int sx = res[0], sy = res[1], sz = res[2];
    const f3 delta = rotate(invTrack, mk3(0, 0, dim[2] / (float)sz));
    const f3 cameraDelta = rotate(K, delta);
#pragma omp parallel for
    for (int y = 0; y < sy; ++y)
        for (int x = 0; x < sx; ++x) {
            const f3 p0 = mk3(((float)x + 0.5f) * dim[0] / (float)sx,
                              ((float)y + 0.5f) * dim[1] / (float)sy,
                              (0.0f + 0.5f) * dim[2] / (float)sz);
            f3 pos = xform(invTrack, p0);
            f3 cameraX = xform(K, pos);
            for (int z = 0; z < sz; ++z, pos = pos + delta, cameraX = cameraX + cameraDelta) {
                if (pos.z < 0.0001f) continue;
                const float pxf = cameraX.x / cameraX.z + 0.5f;
                const float pyf = cameraX.y / cameraX.z + 0.5f;
                if (pxf < 0 || pxf > w - 1 || pyf < 0 || pyf > h - 1) continue;
                const unsigned px = (unsigned)pxf, py = (unsigned)pyf;
                const float d = depth[px + py * w];
                if (d == 0) continue;
                const float rx = pos.x / pos.z, ry = pos.y / pos.z;
                const float diff = (d - cameraX.z) * sqrtf(1 + rx * rx + ry * ry);
                if (diff > -mu) {
                    const float sdf = fminf(1.f, diff / mu);
                    int16_t *v = vol + 2 * ((size_t)x + (size_t)y * sx + (size_t)z * sx * sy);
                    float tsdf = (float)v[0] * SHORT2FLOAT;
                    float wgt = (float)v[1];
                    tsdf = clampf((wgt * tsdf + sdf) / (wgt + 1), -1.f, 1.f);
                    wgt = fminf(wgt + 1, maxweight);
                    v[0] = (int16_t)(tsdf * FLOAT2SHORT);
                    v[1] = (int16_t)wgt;
                }
            }
        }
}

/* kernels.cpp raycastKernel */
void kfr_raycast(float *vertex3, float *normal3, int w, int h, const int16_t *vol_,
                 const int32_t res[3], const float dim[3], const float view_[16], float near_plane,
                 float far_plane, float step, float largestep) {
    const M4 view = load_m4(view_);
    VolView vol{vol_, res[0], res[1], res[2], dim[0], dim[1], dim[2]};
    f3 *vertex = reinterpret_cast<f3 *>(vertex3);
    f3 *normal = reinterpret_cast<f3 *>(normal3);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const f4 hit = raycast_one(vol, x, y, view, near_plane, far_plane, step, largestep);
            if (hit.w > 0.0f) {
                vertex[x + y * w] = mk3(hit.x, hit.y, hit.z);
                const f3 g = vol.grad(mk3(hit.x, hit.y, hit.z));
                if (length3(g) == 0) normal[x + y * w] = mk3(KF_INVALID, 0, 0);
                else normal[x + y * w] = normalize3(g);
            } else {
                vertex[x + y * w] = mk3(0, 0, 0);
                normal[x + y * w] = mk3(KF_INVALID, 0, 0);
            }
        }
}

/* kernels.cpp gs2rgb: hue ramp used by renderDepthKernel (value 0.75, saturation 2/3, in double) */
static void gs2rgb(double h, uint8_t out[4]) {
    double r = 0, g = 0, b = 0;
    const double v = 0.75;
    if (v > 0) {
        const double m = 0.25, sv = 0.6667;
        h *= 6.0;
        const int sextant = (int)h;
        const double fract = h - sextant;
        const double vsf = v * sv * fract;
        const double mid1 = m + vsf, mid2 = v - vsf;
        switch (sextant) {
            case 0: r = v; g = mid1; b = m; break;
            case 1: r = mid2; g = v; b = m; break;
            case 2: r = m; g = v; b = mid1; break;
            case 3: r = m; g = mid2; b = v; break;
            case 4: r = mid1; g = m; b = v; break;
            case 5: r = v; g = m; b = mid2; break;
        }
    }
    out[0] = (uint8_t)(r * 255);
    out[1] = (uint8_t)(g * 255);
    out[2] = (uint8_t)(b * 255);
    out[3] = 0;
}

/* kernels.cpp renderDepthKernel */
void kfr_render_depth(uint8_t *rgba, const float *depth, int w, int h, float near_plane, float far_plane) {
    const float range_scale = 1 / (far_plane - near_plane);
    for (int i = 0; i < w * h; ++i) {
        uint8_t *o = rgba + 4 * i;
        if (depth[i] < near_plane) { o[0] = 255; o[1] = 255; o[2] = 255; o[3] = 0; }
        else if (depth[i] > far_plane) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; }
        else {
            const float d = (depth[i] - near_plane) * range_scale;
            gs2rgb(d, o);
        }
    }
}

/* kernels.cpp renderTrackKernel */
void kfr_render_track(uint8_t *rgba, const kfr_track_data *data, int w, int h) {
    for (int i = 0; i < w * h; ++i) {
        uint8_t *o = rgba + 4 * i;
        o[3] = 0;
        switch (data[i].result) {
            case 1: o[0] = 128; o[1] = 128; o[2] = 128; break;   /* ok */
            case -1: o[0] = 0; o[1] = 0; o[2] = 0; break;        /* no input */
            case -2: o[0] = 255; o[1] = 0; o[2] = 0; break;      /* not in image */
            case -3: o[0] = 0; o[1] = 255; o[2] = 0; break;      /* no correspondence */
            case -4: o[0] = 0; o[1] = 0; o[2] = 255; break;      /* too far away */
            case -5: o[0] = 255; o[1] = 255; o[2] = 0; break;    /* wrong normal */
            default: o[0] = 255; o[1] = 128; o[2] = 128; break;
        }
    }
}

/* kernels.cpp renderVolumeKernel: second raycast + Phong-style shading with a point light */
void kfr_render_volume(uint8_t *rgba, int w, int h, const int16_t *vol_, const int32_t res[3], const float dim[3],
                       const float view_[16], float near_plane, float far_plane, float step, float largestep,
                       const float light_[3], const float ambient_[3]) {
    const M4 view = load_m4(view_);
    VolView vol{vol_, res[0], res[1], res[2], dim[0], dim[1], dim[2]};
    const f3 light = mk3(light_[0], light_[1], light_[2]), ambient = mk3(ambient_[0], ambient_[1], ambient_[2]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t *o = rgba + 4 * (x + y * w);
            o[0] = o[1] = o[2] = o[3] = 0;
            const f4 hit = raycast_one(vol, x, y, view, near_plane, far_plane, step, largestep);
            if (hit.w > 0) {
                const f3 test = mk3(hit.x, hit.y, hit.z);
                const f3 surf_norm = vol.grad(test);
                if (length3(surf_norm) > 0) {
                    const f3 diff = normalize3(light - test);
                    const float dir = std::fmax(dot3(normalize3(surf_norm), diff), 0.f);
                    const float c[3] = {dir + ambient.x, dir + ambient.y, dir + ambient.z};
                    for (int k = 0; k < 3; ++k) o[k] = (uint8_t)(std::fmin(std::fmax(c[k], 0.f), 1.f) * 255);
                }
            }
        }
}

void kfr_camera_matrix(float K[16], float fx, float fy, float cx, float cy) {
    store_m4(K, camera_matrix(fx, fy, cx, cy));
}
void kfr_inverse_camera_matrix(float invK[16], float fx, float fy, float cx, float cy) {
    store_m4(invK, inverse_camera_matrix(fx, fy, cx, cy));
}
void kfr_mat_inverse(float out[16], const float in[16]) { store_m4(out, mat_inverse(load_m4(in))); }
void kfr_mat_mul(float out[16], const float a[16], const float b[16]) {
    store_m4(out, mat_mul(load_m4(a), load_m4(b)));
}

/* ------------------------------------------------------------------------------------------
 * Whole pipeline: Kfusion::preprocessing / tracking / integration / raycasting driven in the
 * order of kfusion_library.cpp's sb_process_once.  In-tree evidence for the schedule:
 * icra2018_results/1080/violons_libkfusion-cpp_living_room_traj0_loop.log:37-46 of the reference
 * (pose frozen for rows 1-4, raycast from row 4, integrate rows 1-5 then every second row).
 * ------------------------------------------------------------------------------------------ */
struct kfr_context {
    kfr_config cfg;
    int w, h;           /* compute size */
    float k[4];         /* intrinsics at compute size */
    int frame;
    M4 pose, old_pose, raycast_pose;
    std::vector<float> float_depth;
    std::vector<float> scaled_depth[3];
    std::vector<float> in_vertex[3], in_normal[3];
    std::vector<float> vertex, normal;
    std::vector<kfr_track_data> tracking_result;
    std::vector<int16_t> volume;
    float reduction[8 * 32];
    int tracked;
    int icp_iterations;      /* track + reduce passes of the last frame (all levels) */
    double cutoff_margin;    /* smallest kfr_last_cutoff_margin() of those passes */
};

kfr_context *kfr_create(const kfr_config *cfg) {
    kfr_context *c = new kfr_context();
    c->cfg = *cfg;
    const int r = cfg->compute_size_ratio;
    c->w = cfg->input_width / r;
    c->h = cfg->input_height / r;
    c->k[0] = cfg->fx / (float)r; c->k[1] = cfg->fy / (float)r;
    c->k[2] = cfg->cx / (float)r; c->k[3] = cfg->cy / (float)r;
    c->frame = 0;
    c->tracked = 0;
    c->icp_iterations = 0;
    c->cutoff_margin = 1e300;
    M4 p;
    p.r[0] = f4{1, 0, 0, cfg->init_pos[0]};
    p.r[1] = f4{0, 1, 0, cfg->init_pos[1]};
    p.r[2] = f4{0, 0, 1, cfg->init_pos[2]};
    p.r[3] = f4{0, 0, 0, 1};
    c->pose = c->old_pose = c->raycast_pose = p;
    const size_t n = (size_t)c->w * c->h;
    c->float_depth.assign(n, 0.f);
    for (int l = 0; l < 3; ++l) {
        const size_t nl = (size_t)(c->w >> l) * (c->h >> l);
        c->scaled_depth[l].assign(nl, 0.f);
        c->in_vertex[l].assign(3 * nl, 0.f);
        c->in_normal[l].assign(3 * nl, 0.f);
    }
    c->vertex.assign(3 * n, 0.f);
    c->normal.assign(3 * n, 0.f);
    c->tracking_result.assign(n, kfr_track_data{0, 0.f, {0, 0, 0, 0, 0, 0}});
    const size_t nv = (size_t)cfg->vol_res[0] * cfg->vol_res[1] * cfg->vol_res[2];
    c->volume.resize(2 * nv);
    for (size_t i = 0; i < nv; ++i) { c->volume[2 * i] = 32766; c->volume[2 * i + 1] = 0; }
    std::memset(c->reduction, 0, sizeof(c->reduction));
    return c;
}

void kfr_destroy(kfr_context *c) { delete c; }

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int kfr_process_frame(kfr_context *c, const uint16_t *depth, double *phase_seconds) {
    const kfr_config &cfg = c->cfg;
    const int w = c->w, h = c->h;
    const int frame = c->frame;
    int flags = 0;
    double t0 = now_s();

    /* Kfusion::preprocessing */
    kfr_mm2meters(c->float_depth.data(), w, h, depth, cfg.input_width, cfg.input_height, cfg.depth_divisor);
    kfr_bilateral(c->scaled_depth[0].data(), c->float_depth.data(), w, h, 2, 4.0f, 0.1f);
    double t1 = now_s();

    /* Kfusion::tracking */
    int tracked = 0;
    if (frame % cfg.tracking_rate == 0) {
        for (int l = 1; l < 3; ++l)
            kfr_half_sample(c->scaled_depth[l].data(), c->scaled_depth[l - 1].data(), w >> (l - 1),
                            h >> (l - 1), 0.1f * 3, 1);
        for (int l = 0; l < 3; ++l) {
            const float s = (float)(1 << l);
            float invK[16];
            kfr_inverse_camera_matrix(invK, c->k[0] / s, c->k[1] / s, c->k[2] / s, c->k[3] / s);
            kfr_depth2vertex(c->in_vertex[l].data(), c->scaled_depth[l].data(), w >> l, h >> l, invK);
            kfr_vertex2normal(c->in_normal[l].data(), c->in_vertex[l].data(), w >> l, h >> l);
        }
        c->old_pose = c->pose;
        c->icp_iterations = 0;
        c->cutoff_margin = 1e300;
        const M4 project_reference =
            mat_mul(camera_matrix(c->k[0], c->k[1], c->k[2], c->k[3]), mat_inverse(c->raycast_pose));
        for (int level = 2; level >= 0; --level) {
            const int lw = w >> level, lh = h >> level;
            for (int i = 0; i < cfg.pyramid[level]; ++i) {
                kfr_track(c->tracking_result.data(), w, c->in_vertex[level].data(),
                          c->in_normal[level].data(), lw, lh, c->vertex.data(), c->normal.data(), w, h,
                          &c->pose.r[0].x, &project_reference.r[0].x, 0.1f, 0.8f);
                kfr_reduce(c->reduction, c->tracking_result.data(), w, lw, lh);
                c->icp_iterations++;
                const int converged = kfr_update_pose(&c->pose.r[0].x, c->reduction, cfg.icp_threshold);
                c->cutoff_margin = fmin(c->cutoff_margin, kfr_last_cutoff_margin());
                if (converged) break;
            }
        }
        tracked = kfr_check_pose(&c->pose.r[0].x, &c->old_pose.r[0].x, c->reduction, w, h, 0.15f);
    }
    c->tracked = tracked;
    if (tracked) flags |= 1;
    double t2 = now_s();

    /* Kfusion::integration */
    if ((tracked && (frame % cfg.integration_rate) == 0) || frame <= 3) {
        const M4 inv_pose = mat_inverse(c->pose);
        const M4 K = camera_matrix(c->k[0], c->k[1], c->k[2], c->k[3]);
        const float *src = cfg.integrate_filtered ? c->scaled_depth[0].data() : c->float_depth.data();
        kfr_integrate(c->volume.data(), cfg.vol_res, cfg.vol_dim, src, w, h, &inv_pose.r[0].x, &K.r[0].x,
                      cfg.mu, 100.0f);
        flags |= 2;
    }
    double t3 = now_s();

    /* Kfusion::raycasting */
    if (frame > 2) {
        c->raycast_pose = c->pose;
        const M4 view = mat_mul(c->raycast_pose, inverse_camera_matrix(c->k[0], c->k[1], c->k[2], c->k[3]));
        const float mind = fminf(cfg.vol_dim[0], fminf(cfg.vol_dim[1], cfg.vol_dim[2]));
        int maxr = cfg.vol_res[0];
        if (cfg.vol_res[1] > maxr) maxr = cfg.vol_res[1];
        if (cfg.vol_res[2] > maxr) maxr = cfg.vol_res[2];
        const float step = mind / (float)maxr;
        kfr_raycast(c->vertex.data(), c->normal.data(), w, h, c->volume.data(), cfg.vol_res, cfg.vol_dim,
                    &view.r[0].x, 0.4f, 4.0f, step, 0.75f * cfg.mu);
        flags |= 4;
    }
    double t4 = now_s();
    if (phase_seconds) {
        phase_seconds[0] = t1 - t0; phase_seconds[1] = t2 - t1;
        phase_seconds[2] = t3 - t2; phase_seconds[3] = t4 - t3;
    }
    c->frame++;
    return flags;
}

void kfr_get_pose(const kfr_context *c, float pose[16]) { store_m4(pose, c->pose); }
void kfr_set_pose(kfr_context *c, const float pose[16]) { c->pose = load_m4(pose); }
int kfr_frame_index(const kfr_context *c) { return c->frame; }

/* Worker threads of the parallel loops (n <= 0: leave as is); returns the count in force. */
int kfr_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
const float *kfr_float_depth(const kfr_context *c) { return c->float_depth.data(); }
const float *kfr_scaled_depth(const kfr_context *c, int l) { return c->scaled_depth[l].data(); }
const float *kfr_input_vertex(const kfr_context *c, int l) { return c->in_vertex[l].data(); }
const float *kfr_input_normal(const kfr_context *c, int l) { return c->in_normal[l].data(); }
const float *kfr_model_vertex(const kfr_context *c) { return c->vertex.data(); }
const float *kfr_model_normal(const kfr_context *c) { return c->normal.data(); }
const int16_t *kfr_volume(const kfr_context *c) { return c->volume.data(); }
const float *kfr_reduction(const kfr_context *c) { return c->reduction; }
int kfr_icp_iterations(const kfr_context *c) { return c->icp_iterations; }
double kfr_cutoff_margin(const kfr_context *c) { return c->cutoff_margin; }
const kfr_track_data *kfr_tracking_result(const kfr_context *c) { return c->tracking_result.data(); }

}  /* extern "C" */
