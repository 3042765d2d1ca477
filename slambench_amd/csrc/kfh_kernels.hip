// gfx950 kernels of the KinectFusion hot path: depth conversion + bilateral filter, robust pyramid,
// vertex / normal maps, projective ICP residuals with the 32-float normal-equation reduction, the
// on-device 6x6 solve and pose update, TSDF integration and raycasting.
//
// Built with -ffp-contract=off (see kfh_device.h).  Wavefront width is 64 throughout.
#include "kfh_kernels.h"

#include <hip/hip_runtime.h>

namespace kfh {

// =====================================================================================================
// preprocess: mm2metersKernel fused with bilateralFilterKernel.
// A 32x8 output tile plus its radius-2 apron is converted once (u16 / divisor) into LDS; the raw metres
// value of the tile centre is written out and the 25-tap window is served from LDS.
// Algorithmic bytes: 2 B in + 4 B raw out + 4 B filtered out per pixel.
// =====================================================================================================
constexpr int PRE_TX = 32, PRE_TY = 8, PRE_R = 2;
constexpr int PRE_LW = PRE_TX + 2 * PRE_R;   // 36
constexpr int PRE_LH = PRE_TY + 2 * PRE_R;   // 12

struct BilateralConst {
    float g[5];            // gaussian[i] = expf(-(i-2)^2 / (2*delta*delta)), delta = 4
    float e_d_squared_2;   // 2 * e_d^2, e_d = 0.1
};

template <bool IN_U16>
__global__ __launch_bounds__(256) void k_preprocess(float *__restrict__ raw, float *__restrict__ filtered,
                                                    int ow, int oh, const void *__restrict__ in, int iw,
                                                    int ratio, float divisor, BilateralConst bc) {
    __shared__ float tile[PRE_LH][PRE_LW];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * PRE_TX, by = blockIdx.y * PRE_TY;
    for (int idx = tid; idx < PRE_LW * PRE_LH; idx += 256) {
        const int ly = idx / PRE_LW, lx = idx - ly * PRE_LW;
        const int gx = min(max(bx + lx - PRE_R, 0), ow - 1);
        const int gy = min(max(by + ly - PRE_R, 0), oh - 1);
        float v;
        if (IN_U16) {
            const uint16_t *in16 = static_cast<const uint16_t *>(in);
            v = (float)in16[gx * ratio + iw * (gy * ratio)] / divisor;
        } else {
            v = static_cast<const float *>(in)[gx + gy * ow];
        }
        tile[ly][lx] = v;
    }
    __syncthreads();
    const int tx = tid & (PRE_TX - 1), ty = tid >> 5;
    const int x = bx + tx, y = by + ty;
    if (x >= ow || y >= oh) return;
    const float center = tile[ty + PRE_R][tx + PRE_R];
    if (raw) raw[x + y * ow] = center;
    if (!filtered) return;
    if (center == 0) { filtered[x + y * ow] = 0; return; }
    float sum = 0.0f, t = 0.0f;
#pragma unroll
    for (int i = -PRE_R; i <= PRE_R; ++i) {
#pragma unroll
        for (int j = -PRE_R; j <= PRE_R; ++j) {
            const float cur = tile[ty + PRE_R + j][tx + PRE_R + i];
            if (cur > 0) {
                const float mod = (cur - center) * (cur - center);
                const float factor = bc.g[i + PRE_R] * bc.g[j + PRE_R] * expf(-mod / bc.e_d_squared_2);
                t += factor * cur;
                sum += factor;
            }
        }
    }
    filtered[x + y * ow] = t / sum;
}

static BilateralConst bilateral_const() {
    BilateralConst bc;
    const float delta = 4.0f, e_d = 0.1f;
    for (int i = 0; i < 5; ++i) {
        const int x = i - 2;
        bc.g[i] = expf(-(float)(x * x) / (2 * delta * delta));
    }
    bc.e_d_squared_2 = e_d * e_d * 2;
    return bc;
}

hipError_t launch_preprocess(float *raw, float *filtered, int ow, int oh, const void *in, bool in_is_u16,
                             int iw, int ratio, float divisor, hipStream_t s) {
    const dim3 grid((ow + PRE_TX - 1) / PRE_TX, (oh + PRE_TY - 1) / PRE_TY);
    const BilateralConst bc = bilateral_const();
    if (in_is_u16)
        hipLaunchKernelGGL(k_preprocess<true>, grid, dim3(256), 0, s, raw, filtered, ow, oh, in, iw, ratio,
                           divisor, bc);
    else
        hipLaunchKernelGGL(k_preprocess<false>, grid, dim3(256), 0, s, raw, filtered, ow, oh, in, iw, ratio,
                           divisor, bc);
    return hipGetLastError();
}

// =====================================================================================================
// halfSampleRobustImageKernel (r = 1): mean of the 2x2 block's pixels within e_d of its top-left pixel.
// Algorithmic bytes: 16 B in + 4 B out per output pixel.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_half_sample(float *__restrict__ out, const float *__restrict__ in,
                                                     int iw, int ow, int oh, float e_d) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const int cx = 2 * x, cy = 2 * y;
    const float2 top = *reinterpret_cast<const float2 *>(in + cx + cy * iw);
    const float2 bot = *reinterpret_cast<const float2 *>(in + cx + (cy + 1) * iw);
    const float center = top.x;
    float sum = 0.0f, t = 0.0f;
    // upstream order: i (row) outer, j (column) inner
    if (fabsf(top.x - center) < e_d) { sum += 1.0f; t += top.x; }
    if (fabsf(top.y - center) < e_d) { sum += 1.0f; t += top.y; }
    if (fabsf(bot.x - center) < e_d) { sum += 1.0f; t += bot.x; }
    if (fabsf(bot.y - center) < e_d) { sum += 1.0f; t += bot.y; }
    out[x + y * ow] = t / sum;
}

hipError_t launch_half_sample(float *out, const float *in, int iw, int ih, float e_d, hipStream_t s) {
    const int ow = iw / 2, oh = ih / 2;
    const dim3 grid((ow + 63) / 64, (oh + 3) / 4);
    hipLaunchKernelGGL(k_half_sample, grid, dim3(256), 0, s, out, in, iw, ow, oh, e_d);
    return hipGetLastError();
}

// =====================================================================================================
// depth2vertexKernel + vertex2normalKernel fused: the four neighbour vertices are recomputed from depth
// with the same expression the vertex map itself uses, so the normal is identical to the two-pass form.
// Algorithmic bytes: 4 B in + 32 B out per pixel (float4-strided maps).
// =====================================================================================================
__device__ inline f3 back_project(float d, int x, int y, const Mat4 &invK) {
    return d > 0 ? d * rotate(invK, mk3((float)x, (float)y, 1.f)) : mk3(0, 0, 0);
}

__global__ __launch_bounds__(256) void k_vertex_normal(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                       const float *__restrict__ depth, int w, int h,
                                                       Mat4 invK) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int xl = max(x - 1, 0), xr = min(x + 1, w - 1);
    const int yu = max(y - 1, 0), yd = min(y + 1, h - 1);
    const f3 v = back_project(depth[x + y * w], x, y, invK);
    const f3 left = back_project(depth[xl + y * w], xl, y, invK);
    const f3 right = back_project(depth[xr + y * w], xr, y, invK);
    const f3 up = back_project(depth[x + yu * w], x, yu, invK);
    const f3 down = back_project(depth[x + yd * w], x, yd, invK);
    vertex[x + y * w] = make_float4(v.x, v.y, v.z, 0.f);
    if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) {
        normal[x + y * w] = make_float4(KFH_INVALID, 0.f, 0.f, 0.f);
        return;
    }
    const f3 dxv = right - left;
    const f3 dyv = down - up;
    const f3 n = normalize3(cross3(dyv, dxv));
    normal[x + y * w] = make_float4(n.x, n.y, n.z, 0.f);
}

hipError_t launch_vertex_normal(float4 *vertex, float4 *normal, const float *depth, int w, int h, float fx,
                                float fy, float cx, float cy, hipStream_t s) {
    const dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_vertex_normal, grid, dim3(256), 0, s, vertex, normal, depth, w, h,
                       inverse_camera_matrix(fx, fy, cx, cy));
    return hipGetLastError();
}

// =====================================================================================================
// trackKernel + reduceKernel fused.  Each thread evaluates KFH_TRACK_PIX_PER_THREAD pixels and keeps the
// 32 sums in registers; a wavefront shuffle tree and a 4-wave LDS stage produce one 32-float partial per
// block.  Row layout: [0] sum e^2, [1..6] sum e*J, [7..27] upper triangle of sum J J^T, [28] valid,
// [29] result==-4, [30] result==-5, [31] result in {-1,-2,-3}.
// Algorithmic bytes: 64 B per pixel (in vertex+normal, gathered reference vertex+normal) + 128 B per block.
// =====================================================================================================
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ __launch_bounds__(KFH_TRACK_THREADS) void k_track_reduce(
    float *__restrict__ partials, kfh_track_data *__restrict__ track_out, int track_stride,
    const float4 *__restrict__ in_vertex, const float4 *__restrict__ in_normal, int iw, int ih,
    const float4 *__restrict__ ref_vertex, const float4 *__restrict__ ref_normal, int rw, int rh,
    const Mat4 *__restrict__ Ttrack_p, const Mat4 *__restrict__ view_p, float dist_threshold,
    float normal_threshold, const int *__restrict__ done_flag, int done_value) {
    if (done_flag && *done_flag == done_value) return;
    const Mat4 Ttrack = *Ttrack_p;
    const Mat4 view = *view_p;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0f;

    const int npix = iw * ih;
    const int base = blockIdx.x * (KFH_TRACK_THREADS * KFH_TRACK_PIX_PER_THREAD) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < KFH_TRACK_PIX_PER_THREAD; ++k) {
        const int p = base + k * KFH_TRACK_THREADS;
        if (p >= npix) break;
        const int y = p / iw, x = p - y * iw;
        int result;
        float error = 0.f;
        float J[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float4 nin4 = in_normal[p];
        if (nin4.x == KFH_INVALID) {
            result = -1;
        } else {
            const float4 vin4 = in_vertex[p];
            const f3 projectedVertex = xform(Ttrack, mk3(vin4.x, vin4.y, vin4.z));
            const f3 projectedPos = xform(view, projectedVertex);
            const float px = projectedPos.x / projectedPos.z + 0.5f;
            const float py = projectedPos.y / projectedPos.z + 0.5f;
            if (px < 0 || px > rw - 1 || py < 0 || py > rh - 1) {
                result = -2;
            } else {
                const unsigned rx = (unsigned)px, ry = (unsigned)py;
                const float4 rn4 = ref_normal[rx + ry * rw];
                if (rn4.x == KFH_INVALID) {
                    result = -3;
                } else {
                    const float4 rv4 = ref_vertex[rx + ry * rw];
                    const f3 referenceNormal = mk3(rn4.x, rn4.y, rn4.z);
                    const f3 diff = mk3(rv4.x, rv4.y, rv4.z) - projectedVertex;
                    const f3 projectedNormal = rotate(Ttrack, mk3(nin4.x, nin4.y, nin4.z));
                    if (length3(diff) > dist_threshold) {
                        result = -4;
                    } else if (dot3(projectedNormal, referenceNormal) < normal_threshold) {
                        result = -5;
                    } else {
                        result = 1;
                        error = dot3(referenceNormal, diff);
                        const f3 c = cross3(projectedVertex, referenceNormal);
                        J[0] = referenceNormal.x; J[1] = referenceNormal.y; J[2] = referenceNormal.z;
                        J[3] = c.x; J[4] = c.y; J[5] = c.z;
                    }
                }
            }
        }
        if (track_out) {
            kfh_track_data td;
            td.result = result;
            td.error = error;
            for (int i = 0; i < 6; ++i) td.J[i] = J[i];
            track_out[x + y * track_stride] = td;
        }
        if (result < 1) {
            acc[29] += result == -4 ? 1.f : 0.f;
            acc[30] += result == -5 ? 1.f : 0.f;
            acc[31] += result > -4 ? 1.f : 0.f;
        } else {
            acc[0] += error * error;
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[1 + i] += error * J[i];
            int q = 7;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) acc[q++] += J[i] * J[j];
            acc[28] += 1.f;
        }
    }

    __shared__ float wsum[KFH_TRACK_THREADS / 64][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) wsum[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = wsum[0][threadIdx.x];
#pragma unroll
        for (int wv = 1; wv < KFH_TRACK_THREADS / 64; ++wv) v += wsum[wv][threadIdx.x];
        partials[blockIdx.x * 32 + threadIdx.x] = v;
    }
}

hipError_t launch_track_reduce(float *partials, kfh_track_data *track_out, int track_stride,
                               const float4 *in_vertex, const float4 *in_normal, int iw, int ih,
                               const float4 *ref_vertex, const float4 *ref_normal, int rw, int rh,
                               const Mat4 *Ttrack, const Mat4 *view, float dist_threshold,
                               float normal_threshold, const int *done_flag, int done_value,
                               hipStream_t s) {
    const int nb = track_blocks(iw, ih);
    hipLaunchKernelGGL(k_track_reduce, dim3(nb), dim3(KFH_TRACK_THREADS), 0, s, partials, track_out,
                       track_stride, in_vertex, in_normal, iw, ih, ref_vertex, ref_normal, rw, rh, Ttrack,
                       view, dist_threshold, normal_threshold, done_flag, done_value);
    return hipGetLastError();
}

// Fixed-order sum of the per-block partials: 8 interleaved chains (blocks b, b+8, ...) added in order
// 0..7, so the result does not depend on scheduling.
__device__ inline void sum_partials_block(float *sums_lds /*[8][32]*/, const float *partials, int nblocks) {
    const int col = threadIdx.x & 31, chain = threadIdx.x >> 5;   // 256 threads = 8 chains x 32 columns
    float v = 0.f;
    for (int b = chain; b < nblocks; b += 8) v += partials[b * 32 + col];
    sums_lds[chain * 32 + col] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = sums_lds[col];
        for (int c = 1; c < 8; ++c) t += sums_lds[c * 32 + col];
        sums_lds[col] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_sum_partials(float *sums, const float *partials, int nblocks) {
    __shared__ float lds[8 * 32];
    sum_partials_block(lds, partials, nblocks);
    if (threadIdx.x < 32) sums[threadIdx.x] = lds[threadIdx.x];
}

hipError_t launch_sum_partials(float *sums, const float *partials, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, sums, partials, nblocks);
    return hipGetLastError();
}

// =====================================================================================================
// solve + updatePoseKernel + checkPoseKernel on the device (one lane, double precision).
// The reference solves the 6x6 normal equations with an SVD back-substitution that zeroes singular
// values below sigma_max / 1e6.  For a symmetric positive definite system whose condition number is
// provably below 1e6 that is the plain solution, obtained here by Cholesky; otherwise the eigen
// decomposition (cyclic Jacobi) applies the truncation rule literally.
// =====================================================================================================
__device__ void jacobi_solve6(const double C_in[6][6], const double b[6], double x[6]) {
    double A[6][6], V[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { A[i][j] = C_in[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
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
    double wmax = 0.0;
    for (int i = 0; i < 6; ++i) wmax = fmax(wmax, fabs(A[i][i]));
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
    for (int i = 0; i < 6; ++i) {
        const double w = A[i][i];
        if (!(fabs(w) * 1e6 > wmax)) continue;
        double p = 0.0;
        for (int j = 0; j < 6; ++j) p += V[j][i] * b[j];
        p /= w;
        for (int j = 0; j < 6; ++j) x[j] += p * V[j][i];
    }
}

// Returns true when the Cholesky solution is valid AND cond_2(C) < 1e6 is certain
// (cond_2 <= trace(C) * ||L^-1||_F^2).
__device__ bool cholesky_solve6(const double C[6][6], const double b[6], double x[6]) {
    double L[6][6];
    double trace = 0.0;
    for (int i = 0; i < 6; ++i) trace += C[i][i];
    for (int j = 0; j < 6; ++j) {
        double d = C[j][j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) return false;
        const double ljj = sqrt(d);
        L[j][j] = ljj;
        const double inv = 1.0 / ljj;
        for (int i = j + 1; i < 6; ++i) {
            double s = C[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * inv;
        }
    }
    // M = L^-1 (lower triangular), accumulate its squared Frobenius norm
    double M[6][6];
    double fro2 = 0.0;
    for (int j = 0; j < 6; ++j) {
        M[j][j] = 1.0 / L[j][j];
        fro2 += M[j][j] * M[j][j];
        for (int i = j + 1; i < 6; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s -= L[i][k] * M[k][j];
            M[i][j] = s / L[i][i];
            fro2 += M[i][j] * M[i][j];
        }
    }
    if (!(trace * fro2 < 1e6)) return false;
    // x = M^T (M b)
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s += M[i][k] * b[k];
        y[i] = s;
    }
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        for (int k = i; k < 6; ++k) s += M[k][i] * y[k];
        x[i] = s;
    }
    return true;
}

__device__ void solve_normal_equations(const float *vals /*27*/, double x[6]) {
    double b[6], C[6][6];
    for (int i = 0; i < 6; ++i) b[i] = (double)vals[i];
    int k = 6;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { C[i][j] = (double)vals[k]; C[j][i] = (double)vals[k]; ++k; }
    if (!cholesky_solve6(C, b, x)) jacobi_solve6(C, b, x);
}

// SE(3) exponential of a twist (translation first, rotation last), double in, float 4x4 out.
__device__ Mat4 se3_exp(const double mu[6]) {
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
    Mat4 out;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out.m[i * 4 + j] = (float)R[i][j];
        out.m[i * 4 + 3] = (float)t[i];
    }
    out.m[12] = 0; out.m[13] = 0; out.m[14] = 0; out.m[15] = 1;
    return out;
}

// pose <- exp(x) * pose ; returns |x| < threshold
__device__ bool update_pose_dev(Mat4 &pose, const float *sums, float icp_threshold, double x[6]) {
    solve_normal_equations(sums + 1, x);
    const Mat4 delta = se3_exp(x);
    pose = mat_mul(delta, pose);
    double n2 = 0.0;
    for (int i = 0; i < 6; ++i) n2 += x[i] * x[i];
    return sqrt(n2) < (double)icp_threshold;
}

// Matrices the later phases of the frame (and the next frame's tracking) read.
__device__ void derive_matrices(Ctrl *c, const float k[4], bool set_raycast_pose) {
    c->inv_pose = mat_inverse(c->pose);
    c->view = mat_mul(c->pose, inverse_camera_matrix(k[0], k[1], k[2], k[3]));
    if (set_raycast_pose) {
        c->raycast_pose = c->pose;
        c->project_ref = mat_mul(camera_matrix(k[0], k[1], k[2], k[3]), mat_inverse(c->raycast_pose));
    }
    c->old_pose = c->pose;
}

__device__ void finish_tracking(const SolveArgs &a, bool ran_tracking) {
    Ctrl *c = a.ctrl;
    int tracked = 0;
    if (ran_tracking) {
        const float *s = c->sums;
        // checkPoseKernel
        if ((sqrtf(s[0] / s[28]) > 2e-2f) || (s[28] / (float)(a.w * a.h) < a.track_threshold)) {
            c->pose = c->old_pose;
            tracked = 0;
        } else {
            tracked = 1;
        }
    }
    c->tracked = tracked;
    const int integrated = ((tracked && (a.frame % a.integration_rate) == 0) || a.frame <= 3) ? 1 : 0;
    c->integrate_gate = integrated;
    c->frame = a.frame;
    derive_matrices(c, a.k, a.will_raycast != 0);
    if (a.status_ring) {
        kfh_frame_status *st = a.status_ring + (a.frame % KFH_STATUS_RING);
        st->frame = a.frame;
        st->tracked = tracked;
        st->integrated = integrated;
        st->raycasted = a.will_raycast;
        for (int i = 0; i < 16; ++i) st->pose[i] = c->pose.m[i];
        for (int i = 0; i < 32; ++i) st->sums[i] = c->sums[i];
    }
}

__global__ __launch_bounds__(256) void k_icp_solve(SolveArgs a) {
    Ctrl *c = a.ctrl;
    if (c->level_done[a.level] == a.frame + 1) return;   // uniform: every thread reads the same word
    __shared__ float lds[8 * 32];
    sum_partials_block(lds, a.partials, a.nblocks);
    if (threadIdx.x == 0) {
        for (int i = 0; i < 32; ++i) c->sums[i] = lds[i];
        Mat4 pose = c->pose;
        double x[6];
        const bool converged = update_pose_dev(pose, lds, a.icp_threshold, x);
        c->pose = pose;
        for (int i = 0; i < 6; ++i) c->x[i] = x[i];
        const bool level_finished = converged || a.last_iter_of_level;
        if (level_finished) {
            c->level_done[a.level] = a.frame + 1;
            if (a.final_level) finish_tracking(a, true);
        }
    }
}

hipError_t launch_icp_solve(const SolveArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_icp_solve, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void k_track_skip(SolveArgs a) {
    if (threadIdx.x == 0) finish_tracking(a, false);
}

hipError_t launch_track_skip(const SolveArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_track_skip, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

__global__ void k_update_pose(float *pose16, double *x6, int *converged, const float *sums, float thr) {
    if (threadIdx.x != 0) return;
    Mat4 pose;
    for (int i = 0; i < 16; ++i) pose.m[i] = pose16[i];
    double x[6];
    const bool conv = update_pose_dev(pose, sums, thr, x);
    for (int i = 0; i < 16; ++i) pose16[i] = pose.m[i];
    for (int i = 0; i < 6; ++i) x6[i] = x[i];
    *converged = conv ? 1 : 0;
}

hipError_t launch_update_pose(float *pose16, double *x6, int *converged, const float *sums, float thr,
                              hipStream_t s) {
    hipLaunchKernelGGL(k_update_pose, dim3(1), dim3(64), 0, s, pose16, x6, converged, sums, thr);
    return hipGetLastError();
}

struct K4 { float k[4]; };
__global__ void k_pose_derive(Ctrl *c, K4 k, int set_raycast_pose) {
    if (threadIdx.x == 0) derive_matrices(c, k.k, set_raycast_pose != 0);
}

hipError_t launch_pose_derive(Ctrl *ctrl, const float k[4], int set_raycast_pose, hipStream_t s) {
    K4 kk;
    for (int i = 0; i < 4; ++i) kk.k[i] = k[i];
    hipLaunchKernelGGL(k_pose_derive, dim3(1), dim3(64), 0, s, ctrl, kk, set_raycast_pose);
    return hipGetLastError();
}

// =====================================================================================================
// integrateKernel: one thread per (x, y) column marches z with the incrementally accumulated camera-space
// position of the reference (pos += delta), so voxel -> pixel association is bit-identical.  Consecutive
// lanes hold consecutive x: every z step of a wavefront touches one 256-byte run of short2 voxels.
// A slab [z0, z1) replays the z < z0 additions without touching memory; `vol` starts at plane z0.
// Algorithmic bytes (convention): 8 B per voxel (4 read + 4 written) x voxels of the slab.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_integrate(short2 *__restrict__ vol, int sx, int sy, int sz,
                                                   float dimx, float dimy, float dimz, int z0, int z1,
                                                   const float *__restrict__ depth, int w, int h,
                                                   const Mat4 *__restrict__ inv_track_p, Mat4 K, float mu,
                                                   float maxweight, const int *__restrict__ gate) {
    if (gate && *gate == 0) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= sx || y >= sy) return;
    const Mat4 invTrack = *inv_track_p;
    const f3 delta = rotate(invTrack, mk3(0, 0, dimz / (float)sz));
    const f3 cameraDelta = rotate(K, delta);
    const f3 p0 = mk3(((float)x + 0.5f) * dimx / (float)sx, ((float)y + 0.5f) * dimy / (float)sy,
                      (0.0f + 0.5f) * dimz / (float)sz);
    f3 pos = xform(invTrack, p0);
    f3 cameraX = xform(K, pos);
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);
    const size_t plane = (size_t)sx * sy;
    short2 *col = vol + (size_t)x + (size_t)y * sx;
    for (int z = 0; z < z1; ++z, pos = pos + delta, cameraX = cameraX + cameraDelta) {
        if (z < z0) continue;
        if (pos.z < 0.0001f) continue;
        const float pxf = cameraX.x / cameraX.z + 0.5f;
        const float pyf = cameraX.y / cameraX.z + 0.5f;
        if (pxf < 0 || pxf > wmax || pyf < 0 || pyf > hmax) continue;
        const unsigned px = (unsigned)pxf, py = (unsigned)pyf;
        const float d = depth[px + py * w];
        if (d == 0) continue;
        const float rx = pos.x / pos.z, ry = pos.y / pos.z;
        const float diff = (d - cameraX.z) * sqrtf(1 + rx * rx + ry * ry);
        if (diff > -mu) {
            const float sdf = fminf(1.f, diff / mu);
            short2 *v = col + (size_t)(z - z0) * plane;
            const short2 raw = *v;
            float tsdf = (float)raw.x * KFH_SHORT2FLOAT;
            float wgt = (float)raw.y;
            tsdf = fmaxf(-1.f, fminf((wgt * tsdf + sdf) / (wgt + 1), 1.f));
            wgt = fminf(wgt + 1, maxweight);
            *v = make_short2((short)(tsdf * KFH_FLOAT2SHORT), (short)wgt);
        }
    }
}

hipError_t launch_integrate(short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                            const float *depth, int w, int h, const Mat4 *inv_track, Mat4 K, float mu,
                            float maxweight, const int *gate, hipStream_t s) {
    const dim3 grid((res[0] + 63) / 64, (res[1] + 3) / 4);
    hipLaunchKernelGGL(k_integrate, grid, dim3(256), 0, s, vol, res[0], res[1], res[2], dim[0], dim[1], dim[2],
                       z0, z1, depth, w, h, inv_track, K, mu, maxweight, gate);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_fill_volume(uint4 *vol4, size_t n4, short2 *vol, size_t nvox) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const unsigned word = 0x00007FFEu;   // short2(32766, 0), little endian
    for (size_t k = i; k < n4; k += stride) vol4[k] = make_uint4(word, word, word, word);
    if (i < (nvox & 3)) vol[(nvox & ~(size_t)3) + i] = make_short2(32766, 0);
}

hipError_t launch_fill_volume(short2 *vol, size_t nvox, hipStream_t s) {
    const size_t n4 = nvox / 4;
    hipLaunchKernelGGL(k_fill_volume, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint4 *>(vol), n4, vol, nvox);
    return hipGetLastError();
}

// =====================================================================================================
// raycastKernel: one thread per pixel; a wavefront covers an 8x8 pixel tile so that the trilinear gathers
// of neighbouring rays share cache lines and the rays diverge late.  Volume storage starts at plane zs0
// and holds planes [zs0, zs1); the march is clipped to z in [zc0, zc1] metres (the whole volume, or the
// owned slab in the sharded mode).
// =====================================================================================================
struct VolDev {
    const short2 *data;
    int sx, sy, sz;
    float dx, dy, dz;
    int zs0, zs1;   // stored planes
    __device__ inline float vs(int x, int y, int z) const {
        z = min(max(z, zs0), zs1 - 1) - zs0;
        return (float)data[(size_t)x + (size_t)y * sx + (size_t)z * sx * sy].x;
    }
    __device__ inline float interp(f3 pos) const {
        const f3 sp = mk3((pos.x * (float)sx / dx) - 0.5f, (pos.y * (float)sy / dy) - 0.5f,
                          (pos.z * (float)sz / dz) - 0.5f);
        const float flx = floorf(sp.x), fly = floorf(sp.y), flz = floorf(sp.z);
        const int bx = (int)flx, by = (int)fly, bz = (int)flz;
        const f3 f = mk3(sp.x - flx, sp.y - fly, sp.z - flz);
        const int lx = max(bx, 0), ly = max(by, 0), lz = max(bz, 0);
        const int ux = min(bx + 1, sx - 1), uy = min(by + 1, sy - 1), uz = min(bz + 1, sz - 1);
        return (((vs(lx, ly, lz) * (1 - f.x) + vs(ux, ly, lz) * f.x) * (1 - f.y) +
                 (vs(lx, uy, lz) * (1 - f.x) + vs(ux, uy, lz) * f.x) * f.y) * (1 - f.z) +
                ((vs(lx, ly, uz) * (1 - f.x) + vs(ux, ly, uz) * f.x) * (1 - f.y) +
                 (vs(lx, uy, uz) * (1 - f.x) + vs(ux, uy, uz) * f.x) * f.y) * f.z) * KFH_SHORT2FLOAT;
    }
    __device__ inline f3 grad(f3 pos) const {
        const f3 sp = mk3((pos.x * (float)sx / dx) - 0.5f, (pos.y * (float)sy / dy) - 0.5f,
                          (pos.z * (float)sz / dz) - 0.5f);
        const float flx = floorf(sp.x), fly = floorf(sp.y), flz = floorf(sp.z);
        const int bx = (int)flx, by = (int)fly, bz = (int)flz;
        const f3 f = mk3(sp.x - flx, sp.y - fly, sp.z - flz);
        const int llx = max(bx - 1, 0), lly = max(by - 1, 0), llz = max(bz - 1, 0);
        const int lux = max(bx, 0), luy = max(by, 0), luz = max(bz, 0);
        const int ulx = min(bx + 1, sx - 1), uly = min(by + 1, sy - 1), ulz = min(bz + 1, sz - 1);
        const int uux = min(bx + 2, sx - 1), uuy = min(by + 2, sy - 1), uuz = min(bz + 2, sz - 1);
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
        return g * mk3(dx / (float)sx, dy / (float)sy, dz / (float)sz) * (0.5f * KFH_SHORT2FLOAT);
    }
};

__global__ __launch_bounds__(256) void k_raycast(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                 float *__restrict__ ray_t, int w, int h, VolDev vol,
                                                 float zc0, float zc1, const Mat4 *__restrict__ view_p,
                                                 float near_plane, float far_plane, float step,
                                                 float largestep) {
    // 16x16 pixel block, each wavefront an 8x8 tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (x >= w || y >= h) return;
    const Mat4 view = *view_p;
    const f3 origin = mk3(view.m[3], view.m[7], view.m[11]);
    const f3 direction = rotate(view, mk3((float)x, (float)y, 1.f));

    const f3 invR = mk3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
    const f3 box_lo = mk3(0.f, 0.f, zc0);
    const f3 box_hi = mk3(vol.dx, vol.dy, zc1);
    const f3 tbot = invR * (box_lo - origin);
    const f3 ttop = invR * (box_hi - origin);
    const f3 tmin = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    const f3 tmax = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    const float largest_tmin = fmaxf(fmaxf(tmin.x, tmin.y), fmaxf(tmin.x, tmin.z));
    const float smallest_tmax = fminf(fminf(tmax.x, tmax.y), fminf(tmax.x, tmax.z));
    const float tnear = fmaxf(largest_tmin, near_plane);
    const float tfar = fminf(smallest_tmax, far_plane);

    float4 hit = make_float4(0.f, 0.f, 0.f, 0.f);
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
                hit = make_float4(p.x, p.y, p.z, t);
            }
        }
    }
    const int o = x + y * w;
    if (hit.w > 0.0f) {
        vertex[o] = make_float4(hit.x, hit.y, hit.z, 0.f);
        const f3 g = vol.grad(mk3(hit.x, hit.y, hit.z));
        if (length3(g) == 0) {
            normal[o] = make_float4(KFH_INVALID, 0.f, 0.f, 0.f);
        } else {
            const f3 n = normalize3(g);
            normal[o] = make_float4(n.x, n.y, n.z, 0.f);
        }
    } else {
        vertex[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        normal[o] = make_float4(KFH_INVALID, 0.f, 0.f, 0.f);
    }
    if (ray_t) ray_t[o] = hit.w;
}

hipError_t launch_raycast(float4 *vertex, float4 *normal, float *ray_t, int w, int h, const short2 *vol,
                          const int res[3], const float dim[3], int z0, int z1, const Mat4 *view,
                          float near_plane, float far_plane, float step, float largestep, hipStream_t s) {
    VolDev v;
    v.data = vol;
    v.sx = res[0]; v.sy = res[1]; v.sz = res[2];
    v.dx = dim[0]; v.dy = dim[1]; v.dz = dim[2];
    float zc0 = 0.f, zc1 = dim[2];
    if (z0 == 0 && z1 == res[2]) {
        v.zs0 = 0; v.zs1 = res[2];
    } else {
        // sharded: planes [z0-2, z1+2) are stored, the march is clipped to the owned z range
        v.zs0 = z0 - 2 > 0 ? z0 - 2 : 0;
        v.zs1 = z1 + 2 < res[2] ? z1 + 2 : res[2];
        zc0 = (float)z0 * (dim[2] / (float)res[2]);
        zc1 = (float)z1 * (dim[2] / (float)res[2]);
    }
    const dim3 grid((w + 15) / 16, (h + 15) / 16);
    hipLaunchKernelGGL(k_raycast, grid, dim3(256), 0, s, vertex, normal, ray_t, w, h, v, zc0, zc1, view,
                       near_plane, far_plane, step, largestep);
    return hipGetLastError();
}

// Per pixel, keep the slab candidate with the smallest positive hit distance.
__global__ __launch_bounds__(256) void k_composite(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                   float *__restrict__ ray_t, const float *__restrict__ t_parts,
                                                   const float4 *__restrict__ v_parts,
                                                   const float4 *__restrict__ n_parts, int n, int npix) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float best = 0.f;
    int arg = -1;
    for (int i = 0; i < n; ++i) {
        const float t = t_parts[(size_t)i * npix + p];
        if (t > 0.f && (arg < 0 || t < best)) { best = t; arg = i; }
    }
    if (arg >= 0) {
        vertex[p] = v_parts[(size_t)arg * npix + p];
        normal[p] = n_parts[(size_t)arg * npix + p];
    } else {
        vertex[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        normal[p] = make_float4(KFH_INVALID, 0.f, 0.f, 0.f);
    }
    if (ray_t) ray_t[p] = best;
}

hipError_t launch_composite(float4 *vertex, float4 *normal, float *ray_t, const float *t_parts,
                            const float4 *v_parts, const float4 *n_parts, int n, int npix, hipStream_t s) {
    hipLaunchKernelGGL(k_composite, dim3((npix + 255) / 256), dim3(256), 0, s, vertex, normal, ray_t, t_parts,
                       v_parts, n_parts, n, npix);
    return hipGetLastError();
}

}  // namespace kfh
