// gfx950 kernels of the KinectFusion hot path: depth conversion + bilateral filter, robust pyramid,
// vertex / normal maps, projective ICP residuals with the 32-float normal-equation reduction, the
// on-device 6x6 solve and pose update, TSDF integration and raycasting.
//
// Built with -ffp-contract=off (see kfh_device.h).  Wavefront width is 64 throughout.
#include "kfh_kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>

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

// halfSampleRobustImageKernel (r = 1) on a 2x2 block whose top-left pixel is `center`
__device__ inline float half_sample_2x2(float tl, float tr, float bl, float br, float e_d) {
    const float center = tl;
    float sum = 0.0f, t = 0.0f;
    // upstream order: i (row) outer, j (column) inner
    if (fabsf(tl - center) < e_d) { sum += 1.0f; t += tl; }
    if (fabsf(tr - center) < e_d) { sum += 1.0f; t += tr; }
    if (fabsf(bl - center) < e_d) { sum += 1.0f; t += bl; }
    if (fabsf(br - center) < e_d) { sum += 1.0f; t += br; }
    return t / sum;
}

// With PYRAMID the block also half-samples its filtered 32x8 tile twice (16x4, then 8x2 pixels: the 2x2 blocks
// of the robust half-sampling never straddle a tile), so the depth pyramid costs no further launches.
// One block of 256 threads produces the tile (tile_x, tile_y) of a grid that is tiles_x tiles wide.
template <bool IN_U16, bool PYRAMID>
__device__ inline void preprocess_tile(float *__restrict__ raw, float *__restrict__ filtered, int ow, int oh,
                                       const void *__restrict__ in, int iw, int ratio, float divisor,
                                       const BilateralConst &bc, float *__restrict__ level1,
                                       float *__restrict__ level2, float pyramid_e_d, float *__restrict__ tile_max,
                                       int tile_x, int tile_y, int tiles_x) {
    __shared__ float tile[PRE_LH][PRE_LW];
    __shared__ float f0[PRE_TY][PRE_TX];
    __shared__ float f1[PRE_TY / 2][PRE_TX / 2];
    __shared__ float wave_max[4];
    const int tid = threadIdx.x;
    const int bx = tile_x * PRE_TX, by = tile_y * PRE_TY;
    float vmax = 0.f;
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
        vmax = fmaxf(vmax, v);
    }
    if (tile_max) {
        // largest depth of the tile and its apron: lets the integration skip boxes that lie wholly behind the
        // surface seen through them (every raw or filtered value the tile can produce is at most this)
        for (int m = 32; m >= 1; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64));
        if ((tid & 63) == 0) wave_max[tid >> 6] = vmax;
    }
    __syncthreads();
    if (tile_max && tid == 0)
        tile_max[tile_y * tiles_x + tile_x] = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    const int tx = tid & (PRE_TX - 1), ty = tid >> 5;
    const int x = bx + tx, y = by + ty;
    const bool inside = x < ow && y < oh;
    const float center = tile[ty + PRE_R][tx + PRE_R];
    if (inside && raw) raw[x + y * ow] = center;
    float result = 0.f;
    if (filtered && center != 0) {
        float sum = 0.0f, t = 0.0f;
        // The reference evaluates expf(-mod / (2 * e_d^2)) in every tap.  Here: one reciprocal per pixel, folded
        // with log2(e), and the hardware's 2^x (1 ulp).  The argument moves by about two units in the last place,
        // so a weight differs by |argument| * 1e-7 relative: far inside the 1e-5 that the device's expf already
        // differs from the host's by, and the centre tap (weight 1) bounds the sum from below.
        const float inv_e = (1.0f / bc.e_d_squared_2) * 1.44269504088896341f;
#pragma unroll
        for (int i = -PRE_R; i <= PRE_R; ++i) {
#pragma unroll
            for (int j = -PRE_R; j <= PRE_R; ++j) {
                const float cur = tile[ty + PRE_R + j][tx + PRE_R + i];
                if (cur > 0) {
                    const float mod = (cur - center) * (cur - center);
                    const float factor = bc.g[i + PRE_R] * bc.g[j + PRE_R] * __builtin_amdgcn_exp2f(-mod * inv_e);
                    t += factor * cur;
                    sum += factor;
                }
            }
        }
        result = t / sum;
    }
    if (inside && filtered) filtered[x + y * ow] = result;
    if (!PYRAMID) return;
    // the image sizes are multiples of 4, so a tile is either fully inside the image or cut at a multiple of 4
    f0[ty][tx] = result;
    __syncthreads();
    const int ow1 = ow / 2, oh1 = oh / 2, ow2 = ow / 4, oh2 = oh / 4;
    if (tid < (PRE_TX / 2) * (PRE_TY / 2)) {
        const int hx = tid % (PRE_TX / 2), hy = tid / (PRE_TX / 2);
        const float v = half_sample_2x2(f0[2 * hy][2 * hx], f0[2 * hy][2 * hx + 1], f0[2 * hy + 1][2 * hx],
                                        f0[2 * hy + 1][2 * hx + 1], pyramid_e_d);
        f1[hy][hx] = v;
        const int gx = bx / 2 + hx, gy = by / 2 + hy;
        if (gx < ow1 && gy < oh1) level1[gx + gy * ow1] = v;
    }
    __syncthreads();
    if (tid < (PRE_TX / 4) * (PRE_TY / 4)) {
        const int qx = tid % (PRE_TX / 4), qy = tid / (PRE_TX / 4);
        const float v = half_sample_2x2(f1[2 * qy][2 * qx], f1[2 * qy][2 * qx + 1], f1[2 * qy + 1][2 * qx],
                                        f1[2 * qy + 1][2 * qx + 1], pyramid_e_d);
        const int gx = bx / 4 + qx, gy = by / 4 + qy;
        if (gx < ow2 && gy < oh2) level2[gx + gy * ow2] = v;
    }
}

template <bool IN_U16, bool PYRAMID>
__global__ __launch_bounds__(256) void k_preprocess(float *__restrict__ raw, float *__restrict__ filtered,
                                                    int ow, int oh, const void *__restrict__ in, int iw,
                                                    int ratio, float divisor, BilateralConst bc,
                                                    float *__restrict__ level1, float *__restrict__ level2,
                                                    float pyramid_e_d, float *__restrict__ tile_max) {
    preprocess_tile<IN_U16, PYRAMID>(raw, filtered, ow, oh, in, iw, ratio, divisor, bc, level1, level2, pyramid_e_d,
                                     tile_max, blockIdx.x, blockIdx.y, gridDim.x);
}

size_t depth_tile_count(int w, int h) { return (size_t)((w + PRE_TX - 1) / PRE_TX) * ((h + PRE_TY - 1) / PRE_TY); }

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
                             int iw, int ratio, float divisor, float *level1, float *level2, float pyramid_e_d,
                             float *tile_max, hipStream_t s) {
    const dim3 grid((ow + PRE_TX - 1) / PRE_TX, (oh + PRE_TY - 1) / PRE_TY);
    const BilateralConst bc = bilateral_const();
    const bool pyramid = level1 != nullptr && level2 != nullptr;
    if (pyramid && (!filtered || !in_is_u16 || ow % 4 != 0 || oh % 4 != 0)) return hipErrorInvalidValue;
    if (pyramid)
        hipLaunchKernelGGL((k_preprocess<true, true>), grid, dim3(256), 0, s, raw, filtered, ow, oh, in, iw, ratio,
                           divisor, bc, level1, level2, pyramid_e_d, tile_max);
    else if (in_is_u16)
        hipLaunchKernelGGL((k_preprocess<true, false>), grid, dim3(256), 0, s, raw, filtered, ow, oh, in, iw, ratio,
                           divisor, bc, level1, level2, pyramid_e_d, tile_max);
    else
        hipLaunchKernelGGL((k_preprocess<false, false>), grid, dim3(256), 0, s, raw, filtered, ow, oh, in, iw, ratio,
                           divisor, bc, level1, level2, pyramid_e_d, tile_max);
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
    out[x + y * ow] = half_sample_2x2(top.x, top.y, bot.x, bot.y, e_d);
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

// vertex2normalKernel for the pixel whose four clamped neighbours have the depths dl / dr / du / dd at the
// coordinates (xl, y), (xr, y), (x, yu), (x, yd).  Returns false (normal invalid) if a neighbour has no depth.
__device__ inline bool normal_from_depths(f3 &n, float dl, float dr, float du, float dd, int x, int y, int xl,
                                          int xr, int yu, int yd, const Mat4 &invK) {
    const f3 left = back_project(dl, xl, y, invK);
    const f3 right = back_project(dr, xr, y, invK);
    const f3 up = back_project(du, x, yu, invK);
    const f3 down = back_project(dd, x, yd, invK);
    if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) {
        n = mk3(KFH_INVALID, 0.f, 0.f);
        return false;
    }
    const f3 dxv = right - left;
    const f3 dyv = down - up;
    n = normalize3(cross3(dyv, dxv));
    return true;
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
    vertex[x + y * w] = make_float4(v.x, v.y, v.z, 0.f);
    f3 n;
    normal_from_depths(n, depth[xl + y * w], depth[xr + y * w], depth[x + yu * w], depth[x + yd * w], x, y, xl, xr,
                       yu, yd, invK);
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
// trackKernel + reduceKernel fused.  Each thread evaluates KFH_TRACK_PIX_PER_THREAD pixels; all input
// loads are issued before the projections and all reference gathers before the residuals, so one thread
// exposes two memory round trips, not two per pixel.  The 32 sums stay in registers; a DPP wavefront
// reduction and an LDS stage across the block's waves produce one 32-float partial per block.
// Row layout: [0] sum e^2, [1..6] sum e*J, [7..27] upper triangle of sum J J^T, [28] valid,
// [29] result==-4, [30] result==-5, [31] result in {-1,-2,-3}.
// Algorithmic bytes: 64 B per pixel (in vertex+normal, gathered reference vertex+normal) + 128 B per block.
// =====================================================================================================
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
    return v + __builtin_bit_cast(float, moved);
}

// Sum over the 64 lanes of a wavefront; the total is valid in lane 63.
__device__ inline float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Fixed-order sum of per-block partial rows (32 floats each) by one block of NT threads.  Eight threads
// cover one row with float4 loads, so NT/8 rows are fetched per pass and all passes are in flight together
// (load_partial_rows: a separate step, so a caller can request the rows long before it needs the sums).
// stage_partial_rows folds a thread's passes into its row slot in LDS; after a block barrier wavefront 0
// folds the NT/8 slots (fold_row_slots_wave0).  The result does not depend on scheduling.
template <int NT>
struct PartialRows {
    static constexpr int SLOTS = NT / 8;          // rows fetched per pass
    static constexpr int MAXPASS = 8;             // passes kept in flight
    float4 t[MAXPASS];
};

// Branch-free on purpose: every pass loads a valid row (the index is clamped; the buffer always holds at least
// one row) so that all passes are requested back to back; stage_partial_rows leaves out the rows past the end.
template <int NT>
__device__ inline void load_partial_rows(PartialRows<NT> &pr, const float *__restrict__ partials, int nrows,
                                         int r0 = 0) {
    const int slot = threadIdx.x >> 3, c4 = threadIdx.x & 7;
    const float4 *rows4 = reinterpret_cast<const float4 *>(partials);
    const int last = max(nrows - 1, 0);
#pragma unroll
    for (int u = 0; u < PartialRows<NT>::MAXPASS; ++u) {
        const int r = r0 + u * PartialRows<NT>::SLOTS + slot;
        pr.t[u] = rows4[min(r, last) * 8 + c4];
    }
}

// lds: NT * 4 floats, laid out [slot][32]
template <int NT>
__device__ inline void stage_partial_rows(float *lds, PartialRows<NT> &pr, const float *__restrict__ partials,
                                          int nrows) {
    constexpr int SLOTS = PartialRows<NT>::SLOTS, MAXPASS = PartialRows<NT>::MAXPASS;
    const int slot = threadIdx.x >> 3, c4 = threadIdx.x & 7;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = 0; r0 < nrows; r0 += SLOTS * MAXPASS) {
        if (r0 > 0) load_partial_rows<NT>(pr, partials, nrows, r0);
#pragma unroll
        for (int u = 0; u < MAXPASS; ++u) {
            const bool have = r0 + u * SLOTS + slot < nrows;
            acc.x += have ? pr.t[u].x : 0.f; acc.y += have ? pr.t[u].y : 0.f;
            acc.z += have ? pr.t[u].z : 0.f; acc.w += have ? pr.t[u].w : 0.f;
        }
    }
    reinterpret_cast<float4 *>(lds)[slot * 8 + c4] = acc;
}

// Called by the 64 lanes of wavefront 0 after the block barrier that follows stage_partial_rows.
// Lane l returns sum number (l & 31): each half of the wavefront folds half of the slots in slot order and
// the two halves are added (a + b == b + a, so both halves hold the same bits).
template <int NT>
__device__ inline float fold_row_slots_wave0(const float *lds) {
    constexpr int HALF = PartialRows<NT>::SLOTS / 2;
    static_assert(PartialRows<NT>::SLOTS % 2 == 0, "slot count must be even");
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    // four interleaved partial sums (slots k, k + 4, ...), combined pairwise: a quarter of the dependent additions
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < HALF; ++k) s4[k & 3] += lds[(half * HALF + k) * 32 + col];
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    return s + __shfl_xor(s, 32, 64);
}

__global__ __launch_bounds__(KFH_SUM_THREADS) void k_sum_partials(float *sums, const float *partials, int nrows) {
    __shared__ float lds[KFH_SUM_THREADS * 4];
    PartialRows<KFH_SUM_THREADS> pr;
    load_partial_rows<KFH_SUM_THREADS>(pr, partials, nrows);
    stage_partial_rows<KFH_SUM_THREADS>(lds, pr, partials, nrows);
    __syncthreads();
    if (threadIdx.x < 64) {
        const float s = fold_row_slots_wave0<KFH_SUM_THREADS>(lds);
        if (threadIdx.x < 32) sums[threadIdx.x] = s;
    }
}

hipError_t launch_sum_partials(float *sums, const float *partials, int nrows, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(KFH_SUM_THREADS), 0, s, sums, partials, nrows);
    return hipGetLastError();
}

// =====================================================================================================
// solve + updatePoseKernel + checkPoseKernel on the device, double precision, one wavefront.
// The reference solves the 6x6 normal equations with an SVD back-substitution that zeroes singular
// values below sigma_max / 1e6.  For a symmetric positive definite system whose condition number is
// provably below 1e6 that is the plain solution.  The wavefront holds the augmented system [C | b] and
// the identity one element per lane (lane = 8*row + column) and runs Gauss-Jordan elimination with
// cross-lane reads, which yields x = C^-1 b and C^-1 together; cond_2(C) <= ||C||_F * ||C^-1||_F
// certifies the result.  Otherwise lane 0 applies the truncation rule literally through a cyclic Jacobi
// eigen-decomposition (slow path: rank-deficient or nearly singular systems).
// =====================================================================================================
__device__ inline float readlane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// value of a double held by a lane whose index is the same for the whole wavefront
__device__ inline double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Slow path: cyclic Jacobi eigen-decomposition of the symmetric 6x6 matrix held one element per lane
// (lane = 8*row + column, rows/columns 6 and 7 are zero padding), then the truncated pseudo-inverse
// x = sum_i [ |w_i| * 1e6 > max|w| ] (v_i . b / w_i) v_i.  No per-lane arrays: every access to another
// element is a cross-lane read, so the routine needs no scratch memory.
__device__ void jacobi_solve6_wave(double a, const double b[6], double x[6]) {
    const int lane = threadIdx.x & 63;
    const int i = lane >> 3, j = lane & 7;
    const bool in_c = i < 6 && j < 6;
    double v = (i == j && i < 6) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = wave_sum_f64((in_c && i < j) ? a * a : 0.0);
        const double diag = wave_sum_f64((in_c && i == j) ? a * a : 0.0);
        if (off == 0.0 || off <= 1e-40 * diag) break;
#pragma unroll 1
        for (int p = 0; p < 5; ++p) {
#pragma unroll 1
            for (int q = p + 1; q < 6; ++q) {
                const double apq = __shfl(a, p * 8 + q, 64);
                if (apq == 0.0) continue;                              // wave-uniform
                const double app = __shfl(a, p * 8 + p, 64), aqq = __shfl(a, q * 8 + q, 64);
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                // columns p and q of A
                double aip = __shfl(a, i * 8 + p, 64), aiq = __shfl(a, i * 8 + q, 64);
                if (j == p) a = c * aip - s * aiq;
                else if (j == q) a = s * aip + c * aiq;
                // rows p and q of A
                const double apj = __shfl(a, p * 8 + j, 64), aqj = __shfl(a, q * 8 + j, 64);
                if (i == p) a = c * apj - s * aqj;
                else if (i == q) a = s * apj + c * aqj;
                // columns p and q of V
                const double vip = __shfl(v, i * 8 + p, 64), viq = __shfl(v, i * 8 + q, 64);
                if (j == p) v = c * vip - s * viq;
                else if (j == q) v = s * vip + c * viq;
            }
        }
    }
    // eigenvalue of column j, broadcast down the column
    const double w = __shfl(a, j * 8 + j, 64);
    double wmax = (in_c && i == j) ? fabs(a) : 0.0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, m, 64));
    // p_j = sum_i V[i][j] * b[i]  (sum over rows: lanes that differ in bits 3..5)
    double bi = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) bi = (i == m) ? b[m] : bi;
    double pj = in_c ? v * bi : 0.0;
    pj += __shfl_xor(pj, 8, 64);
    pj += __shfl_xor(pj, 16, 64);
    pj += __shfl_xor(pj, 32, 64);
    const bool keep = in_c && (fabs(w) * 1e6 > wmax);
    // x_i = sum_j keep_j * (p_j / w_j) * V[i][j]  (sum over columns: lanes that differ in bits 0..2)
    double xi = keep ? (pj / w) * v : 0.0;
    xi += __shfl_xor(xi, 1, 64);
    xi += __shfl_xor(xi, 2, 64);
    xi += __shfl_xor(xi, 4, 64);
#pragma unroll
    for (int m = 0; m < 6; ++m) x[m] = __shfl(xi, m * 8, 64);
}

// Lanes are laid out lane = 8*row + column.  Sum of v over the 8 lanes of a matrix row, returned in all of
// them (the same addition tree in every lane, so the lanes agree bit for bit).
template <int CTRL, int BANK_MASK>
__device__ inline double dpp_take_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double row_sum8_f64(double v) {
    v = v + dpp_take_f64<0xB1, 0xF>(0.0, v);                  // lane ^ 1
    v = v + dpp_take_f64<0x4E, 0xF>(0.0, v);                  // lane ^ 2
    double o = dpp_take_f64<0x104, 0x5>(0.0, v);              // lane ^ 4: banks 0, 2 read lane + 4
    o = dpp_take_f64<0x114, 0xA>(o, v);                       //           banks 1, 3 read lane - 4
    return v + o;
}

// y = M u for the symmetric 6x6 matrix held one element per lane; u and y are wave-uniform 6-vectors.
__device__ inline void matvec6_wave(double m_elem, const double u[6], double y[6]) {
    const int j = threadIdx.x & 7;
    double uj = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) uj = (j == k) ? u[k] : uj;
    const double yi = row_sum8_f64(m_elem * uj);
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = readlane_f64(yi, k * 8);
}

// Power iteration: dominant eigenpair of the symmetric positive definite 6x6 matrix held one element per
// lane.  Runs in rounds of 2 * PAIRS products until the residual |M u - theta u| / |theta u| drops below 1e-12
// (the eigenvalue is then good to 1e-24 relative, the eigenvector to 1e-12); gives up (returns false) after 96
// products, which happens when the two largest eigenvalues nearly coincide.
template <int PAIRS>
__device__ inline bool dominant_eigenpair_wave(double m_elem, double &theta, double u[6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = 1.0 + 0.0625 * k;      // generic start
#pragma unroll 1
    for (int round = 0; round < 48 / PAIRS; ++round) {
        double y[6];
#pragma unroll 1
        for (int it = 0; it < PAIRS; ++it) {
            matvec6_wave(m_elem, u, y);
            matvec6_wave(m_elem, y, u);
        }
        double big = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) big = fmax(big, fabs(u[k]));
        if (!(big > 0.0) || !(big < 1e300)) return false;
        const double scale = 1.0 / big;
        double uu = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { u[k] *= scale; uu += u[k] * u[k]; }
        matvec6_wave(m_elem, u, y);
        double uy = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) uy += u[k] * y[k];
        theta = uy / uu;
        double rr = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double r = y[k] - theta * u[k]; rr += r * r; }
        if (rr <= 1e-24 * theta * theta * uu) {
            const double inv_norm = 1.0 / sqrt(uu);
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] *= inv_norm;
            return true;
        }
    }
    return false;
}

// Middle path for systems just past the conditioning limit, where the truncation rule drops at most the
// smallest eigenvalue: lambda_max by power iteration on C, lambda_min and its eigenvector v by power
// iteration on C^-1 (already available from the elimination), then x = C^-1 b - (v.b / lambda_min) v.
// Returns false when it cannot decide safely (a second eigenvalue near the limit, slow convergence);
// the caller then runs the full eigen-decomposition.
__device__ inline bool truncate_smallest_wave(double c_elem, double inv_elem, const double b[6],
                                              const double x_full[6], double x[6], bool *truncated) {
    const int lane = threadIdx.x & 63;
    const int i = lane >> 3, j = lane & 7;
    const bool in_c = i < 6 && j < 6;
    double mu, v[6];
    // (rounds of 8 products: the iteration on C^-1 typically needs 6, and a round of 4 would fall just short)
    if (!dominant_eigenpair_wave<4>(in_c ? inv_elem : 0.0, mu, v)) return false;
    if (!(mu > 0.0)) return false;
    const double lam_min = 1.0 / mu;
    // lambda_max of C is only needed for the comparison lambda_min * 1e6 > lambda_max: any Rayleigh quotient bounds
    // it from below and the Frobenius norm from above, and four products from a generic start bring the quotient
    // within a fraction of a percent.  Only when lambda_min * 1e6 falls between the two bounds is the eigenvalue
    // itself computed (power iteration to convergence).
    const double fro_c = sqrt(wave_sum_f64(c_elem * c_elem));
    if (!(fro_c > 0.0) || !(fro_c < 1e150)) return false;
    double u[6], y[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = 1.0 + 0.0625 * k;      // generic start
    const double unit = 1.0 / fro_c;                          // keeps the iterates in range: |C / fro_c| <= 1
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        matvec6_wave(c_elem, u, y);
        matvec6_wave(c_elem, y, u);
#pragma unroll
        for (int k = 0; k < 6; ++k) u[k] *= unit * unit;
    }
    matvec6_wave(c_elem, u, y);
    double uu = 0.0, uy = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { uu += u[k] * u[k]; uy += u[k] * y[k]; }
    if (!(uu > 0.0)) return false;
    const double below = (uy / uu) * (1.0 - 1e-12), above = fro_c * (1.0 + 1e-12);
    double lam_upper = above;
    bool keep_min;                                            // the reference's rule for the smallest singular value
    if (lam_min * 1e6 > above) {
        keep_min = true;
    } else if (lam_min * 1e6 < below) {
        keep_min = false;
    } else {
        double lam_max, ubig[6];
        if (!dominant_eigenpair_wave<4>(c_elem, lam_max, ubig)) return false;
        if (!(lam_max > 0.0)) return false;
        keep_min = lam_min * 1e6 > lam_max;
        lam_upper = lam_max;
    }
    // second smallest eigenvalue of C: 1 / lambda_max(C^-1 - mu v v^T) >= 1 / ||C^-1 - mu v v^T||_F; it must clearly be
    // kept (an upper bound of lambda_max makes the test stricter, never laxer)
    double vi = 0.0, vj = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { vi = (i == k) ? v[k] : vi; vj = (j == k) ? v[k] : vj; }
    const double d = in_c ? inv_elem - mu * vi * vj : 0.0;
    const double fro2 = wave_sum_f64(d * d);
    if (!(lam_upper * lam_upper * fro2 < 0.96e12)) return false;      // second eigenvalue not clearly kept
    double vb = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) vb += v[k] * b[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = keep_min ? x_full[k] : x_full[k] - mu * vb * v[k];
    *truncated = !keep_min;
    return true;
}

// Called by all 64 lanes of one wavefront; vals = 27 floats {b[6], upper triangle of C}.  x is returned
// in every lane.  *used_jacobi (may be null) reports which path produced it.
__device__ void solve_normal_equations_wave(const float *vals, double x[6], int *used_jacobi) {
    const int lane = threadIdx.x & 63;
    const int i = lane >> 3, j = lane & 7;
    double a = 0.0;
    if (i < 6) {
        if (j < 6) {
            const int r = min(i, j), c = max(i, j);
            a = (double)vals[6 + r * 6 - (r * (r - 1)) / 2 + (c - r)];
        } else if (j == 6) {
            a = (double)vals[i];
        }
    }
    const bool in_c = i < 6 && j < 6;
    const double c_elem = in_c ? a : 0.0;
    double inv = (i == j && i < 6) ? 1.0 : 0.0;
    // the conditioning certificate only needs a few digits: float norms, DPP reductions
    const float fro_c2 = readlane63(wave_sum_to_lane63((float)c_elem * (float)c_elem));
    bool spd = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double p = readlane_f64(a, k * 8 + k);
        if (!(p > 0.0)) spd = false;                 // wave-uniform
        // reciprocal by hardware estimate + two Newton steps (the pivot is positive and finite here)
        double rinv = __builtin_amdgcn_rcp(p);
        rinv = fma(fma(-p, rinv, 1.0), rinv, rinv);
        rinv = fma(fma(-p, rinv, 1.0), rinv, rinv);
        const double rk_a = __shfl(a, k * 8 + j, 64) * rinv;
        const double rk_i = __shfl(inv, k * 8 + j, 64) * rinv;
        const double ck = __shfl(a, i * 8 + k, 64);
        if (i == k) {
            a = rk_a;
            inv = rk_i;
        } else {
            a = a - ck * rk_a;
            inv = inv - ck * rk_i;
        }
    }
    const float fro_i2 = readlane63(wave_sum_to_lane63(in_c ? (float)inv * (float)inv : 0.f));
    // (||C||_F ||C^-1||_F)^2 < (1e6)^2, with 1 % slack for the float norms
    const bool certified = spd && ((double)fro_c2 * (double)fro_i2 < 0.98e12);
    double xf[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) xf[m] = readlane_f64(a, m * 8 + 6);
    int path = 0;                                    // 0 plain solution, 1 eigen-decomposition, 2 smallest dropped
    if (certified) {
#pragma unroll
        for (int m = 0; m < 6; ++m) x[m] = xf[m];
    } else {
        double b[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) b[m] = (double)vals[m];
        bool truncated = false;
        // C^-1 from the elimination is accurate enough for the middle path while cond(C) < ~1e9
        if (spd && (double)fro_c2 * (double)fro_i2 < 1e18 &&
            truncate_smallest_wave(c_elem, inv, b, xf, x, &truncated)) {
            path = truncated ? 2 : 0;
        } else {
            jacobi_solve6_wave(c_elem, b, x);
            path = 1;
        }
    }
    if (used_jacobi) *used_jacobi = path;
}

// SE(3) exponential of a twist (translation first, rotation last), double in, float 4x4 out.
__device__ Mat4 se3_exp(const double mu[6]) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double w0 = mu[3], w1 = mu[4], w2 = mu[5];
    const double theta_sq = w0 * w0 + w1 * w1 + w2 * w2;
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
        } else if (theta_sq < 0.25) {
            // |theta| < 0.5 rad: the Taylor series of sin(t)/t, (1-cos t)/t^2 and (t-sin t)/t^3 in t^2
            // converge below 1e-15 in eight terms; the pose is rounded to float afterwards.  This replaces
            // a double-precision sin/cos/sqrt chain on the critical path of every ICP iteration.
            // (evaluated with fused multiply-adds: half the length of the dependent chain, and closer to the sum)
            const double t2 = theta_sq;
            A = fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, -1.0 / 1307674368000.0, 1.0 / 6227020800.0),
                -1.0 / 39916800), 1.0 / 362880), -1.0 / 5040), 1.0 / 120), -1.0 / 6), 1.0);
            B = fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, -1.0 / 20922789888000.0, 1.0 / 87178291200.0),
                -1.0 / 479001600), 1.0 / 3628800), -1.0 / 40320), 1.0 / 720), -1.0 / 24), 0.5);
            C = fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, -1.0 / 355687428096000.0, 1.0 / 1307674368000.0),
                -1.0 / 6227020800.0), 1.0 / 39916800), -1.0 / 362880), 1.0 / 5040), -1.0 / 120), 1.0 / 6);
        } else {
            const double theta = sqrt(theta_sq);
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

// delta * pose for a delta whose last row is (0, 0, 0, 1) (what se3_exp returns), entry by entry in mat_mul's order
// of operations.  When the pose's last row is (0, 0, 0, 1) as well, so is the product's: its entries are sums of
// 0 * p + ... + 1 * p33 over finite p, and the twelve other entries are all that is computed.
__device__ inline Mat4 pose_product(const Mat4 &D, const Mat4 &P) {
    const bool affine = P.m[12] == 0.f && P.m[13] == 0.f && P.m[14] == 0.f && P.m[15] == 1.f;     // wave-uniform
    if (!affine) return mat_mul(D, P);
    Mat4 R;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = D.m[i * 4 + 0] * P.m[0 * 4 + j];
            s = s + D.m[i * 4 + 1] * P.m[1 * 4 + j];
            s = s + D.m[i * 4 + 2] * P.m[2 * 4 + j];
            s = s + D.m[i * 4 + 3] * P.m[3 * 4 + j];
            R.m[i * 4 + j] = s;
        }
    R.m[12] = 0.f; R.m[13] = 0.f; R.m[14] = 0.f; R.m[15] = 1.f;
    return R;
}

// pose <- exp(x) * pose ; returns |x| < threshold.  Wave-cooperative: all 64 lanes call it and all get
// the same pose, x and result.
__device__ bool update_pose_wave(Mat4 &pose, const float *sums, float icp_threshold, double x[6], int *used_jacobi) {
    solve_normal_equations_wave(sums + 1, x, used_jacobi);
    const Mat4 delta = se3_exp(x);
    pose = pose_product(delta, pose);
    double n2 = 0.0;
    for (int i = 0; i < 6; ++i) n2 += x[i] * x[i];
    return n2 < (double)icp_threshold * (double)icp_threshold;   // |x| < threshold
}

// Matrices the later phases of the frame (and the next frame's tracking) read, and the ICP state the next
// frame's first iteration starts from.
__device__ void derive_matrices(Ctrl *c, const float k[4], bool set_raycast_pose) {
    c->inv_pose = mat_inverse(c->pose);
    c->view = mat_mul(c->pose, inverse_camera_matrix(k[0], k[1], k[2], k[3]));
    if (set_raycast_pose) {
        c->raycast_pose = c->pose;
        c->raycast_view = c->view;
        c->project_ref = mat_mul(camera_matrix(k[0], k[1], k[2], k[3]), mat_inverse(c->raycast_pose));
    }
    c->old_pose = c->pose;
    IcpState *st = &c->icp[0];
    st->pose = c->pose;
    for (int i = 0; i < 32; ++i) st->sums[i] = c->sums[i];
    st->pending_nblocks = 0;
    st->pending_level = 0;
    st->pending_last = 0;
    st->done_mask = 0;
    st->iterations = 0;
    st->slow_solves = 0;
}

__device__ void finish_tracking(const FinishArgs &a, bool ran_tracking, int iterations, int slow_solves) {
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
    c->ray_task_count = 0;      // the raycast of this frame appends its unfinished rays from here
    c->integrate_box_count[0] = c->integrate_box_count[1] = 0; // likewise the work list of this frame's integration
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
        st->icp_iterations = iterations;
        st->slow_solves = slow_solves & 0xFFFF;
        st->examined_solves = slow_solves >> 16;
        st->reserved_ = 0;
    }
}

// =====================================================================================================
// One ICP iteration = one launch.  Every block first finalises the PREVIOUS iteration (fixed-order sum of
// its partial rows, 6x6 solve, pose update, convergence test) and then evaluates this iteration's residuals
// with the new pose.  All blocks compute the same pose from the same data, so no block waits for another
// and an iteration costs one kernel boundary instead of two.  The state travels through two alternating
// IcpState records and two alternating partial-row buffers: launch `seq` reads record seq & 1 and row
// buffer seq & 1, and block 0 writes record (seq + 1) & 1.
//
// trackKernel + reduceKernel: all input loads are issued before the prologue and all reference gathers
// before the residuals; the residual is evaluated branch-free so the gathers stay unconditional.  The 32
// sums stay in registers; a DPP wavefront reduction and an LDS stage across the block's waves produce one
// 32-float partial row per block.  Row layout: [0] sum e^2, [1..6] sum e*J, [7..27] upper triangle of
// sum J J^T, [28] valid, [29] result==-4, [30] result==-5, [31] result in {-1,-2,-3}.
// Algorithmic bytes: 64 B per pixel (in vertex+normal, gathered reference vertex+normal) + 128 B per block.
// =====================================================================================================
// Division by a shared denominator, bit-identical to IEEE division while no intermediate leaves the
// normal range: the reciprocal estimate is refined once and each quotient takes two residual corrections.
__device__ inline float refined_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.f), r, r);
}
__device__ inline float div_shared(float n, float d, float r1) {
    const float q0 = n * r1;
    const float q1 = fmaf(fmaf(-d, q0, n), r1, q0);
    return fmaf(fmaf(-d, q1, n), r1, q1);
}

template <int PW>
struct IcpShared {
    float slots[64 * (1 + PW) * 4];            // row slots of the pending sum, [slot][32]
    float sums[32];                            // reduction of the last finalised iteration
    float part[PW][32];                        // this iteration's sums per pixel wavefront
    Mat4 pose;
    int done_mask, iterations, slow_solves;
};

// Wavefront 0: fold the staged rows, solve, update the pose; or carry the record over when nothing is pending.
template <int NT>
__device__ inline void icp_finalize_wave0(const IcpState *__restrict__ in, Mat4 pose, int pend, float icp_threshold,
                                          const float *slots, float *sums_lds, Mat4 *pose_lds, int *done_mask_lds,
                                          int *iterations_lds, int *slow_lds) {
    const int lane = threadIdx.x & 63;
    const int in_done_mask = in->done_mask, in_last = in->pending_last, in_level = in->pending_level;
    const int in_iterations = in->iterations, in_slow = in->slow_solves;
    if (pend > 0) {
        const float sum = fold_row_slots_wave0<NT>(slots);
        if (lane < 32) sums_lds[lane] = sum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double x[6];
        int slow;
        const bool converged = update_pose_wave(pose, sums_lds, icp_threshold, x, &slow);
        if (lane == 0) {
            const bool level_finished = converged || in_last != 0;
            *done_mask_lds = in_done_mask | (level_finished ? (1 << in_level) : 0);
            *iterations_lds = in_iterations + 1;
            // low half: full eigen-decompositions; high half: solves that had to examine the smallest singular value
            *slow_lds = in_slow + (slow == 1 ? 1 : 0) + (slow != 0 ? 0x10000 : 0);
        }
    } else {
        if (lane < 32) sums_lds[lane] = in->sums[lane];
        if (lane == 0) {
            *done_mask_lds = in_done_mask;
            *iterations_lds = in_iterations;
            *slow_lds = in_slow;
        }
    }
    if (lane == 0) *pose_lds = pose;
}

// one wavefront of block 0: lanes 0..31 store the sums, lane 0 the rest
template <int PW>
__device__ inline void icp_store_state(IcpState *out, const IcpShared<PW> &sh, int pending_nblocks, int level,
                                       int last_iter) {
    const int lane = threadIdx.x & 63;
    if (lane < 32) out->sums[lane] = sh.sums[lane];
    if (lane == 0) {
        out->pose = sh.pose;
        out->pending_nblocks = pending_nblocks;
        out->pending_level = level;
        out->pending_last = last_iter;
        out->done_mask = sh.done_mask;
        out->iterations = sh.iterations;
        out->slow_solves = sh.slow_solves;
    }
}

// Diagnostic time stamps (100 MHz real-time counter) of lane 0 of wavefront w, when a stamp buffer is given.
// Only in builds made with `make STAMPS=1`: the tests and branches cost the iteration kernel about 1 % even
// when no buffer is given.
#ifndef KFH_STAMPS_COMPILED
#define KFH_STAMPS_COMPILED 0
#endif
bool icp_stamps_compiled() { return KFH_STAMPS_COMPILED != 0; }
#define KFH_STAMP(slot, w)                                                                                   \
    do {                                                                                                     \
        if (KFH_STAMPS_COMPILED && a.stamps != nullptr && threadIdx.x == 64 * (w) && blockIdx.x < KFH_STAMP_BLOCKS &&               \
            a.seq < KFH_STAMP_LAUNCHES) {                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            a.stamps[((size_t)a.seq * KFH_STAMP_BLOCKS + blockIdx.x) * KFH_STAMP_SLOTS + (slot)] =           \
                __builtin_amdgcn_s_memrealtime();                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
    } while (0)

// v0 <- {v0 lanes 0..31, v1 lanes 0..31}; v1 <- {v0 lanes 32..63, v1 lanes 32..63}
__device__ inline void swap_halves(float &v0, float &v1) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v0), "+v"(v1));
}
// the same between the even and odd 16-lane rows of each half
__device__ inline void swap_rows(float &v0, float &v1) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v0), "+v"(v1));
}
template <int CTRL, int BANK_MASK>
__device__ inline float dpp_take(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, v), CTRL, 0xF, BANK_MASK, false));
}

// Sum each of 32 per-lane values over the 64 lanes of a wavefront.  Instead of 32 full reductions the
// values are halved at every step: partners (lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2) each keep one half of the
// values and add the partner's copy of that half; after five steps a lane holds one value, and a last
// add with lane ^ 1 completes it.  Returns the total of value number (lane >> 1).  Fixed order, 31 exchanges.
__device__ inline float wave_reduce32(float (&v)[32]) {
    const int lane = threadIdx.x & 63;
    // lane ^ 32: lower half keeps v[0..15], upper half v[16..31]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        swap_halves(v[i], v[16 + i]);
        v[i] = v[i] + v[16 + i];
    }
    // lane ^ 16: even rows keep v[0..7], odd rows v[8..15]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        swap_rows(v[i], v[8 + i]);
        v[i] = v[i] + v[8 + i];
    }
    // lane ^ 8 (rotate the 16-lane row by 8)
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b3 ? v[4 + i] : v[i], give = b3 ? v[i] : v[4 + i];
        v[i] = keep + dpp_take<0x128, 0xF>(0.f, give);                 // row_ror:8
    }
    // lane ^ 4: banks 0 and 2 read from lane + 4, banks 1 and 3 from lane - 4
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? v[2 + i] : v[i], give = b2 ? v[i] : v[2 + i];
        float got = dpp_take<0x104, 0x5>(0.f, give);                   // row_shl:4 into banks 0, 2
        got = dpp_take<0x114, 0xA>(got, give);                         // row_shr:4 into banks 1, 3
        v[i] = keep + got;
    }
    // lane ^ 2
    {
        const float keep = b1 ? v[1] : v[0], give = b1 ? v[0] : v[1];
        v[0] = keep + dpp_take<0x4E, 0xF>(0.f, give);                  // quad_perm [2,3,0,1]
    }
    // lane ^ 1: both lanes of a pair hold the same value index
    return v[0] + dpp_take<0xB1, 0xF>(0.f, v[0]);                      // quad_perm [1,0,3,2]
}

// PACKED: the pipeline's form.  Input normals are recomputed from the depth pyramid (five depths per pixel
// instead of a 16-byte normal), the model map is the packed {t, normal} image written by the raycast and
// the model vertex is rebuilt from the ray through the reference pixel.  !PACKED: caller-provided maps.
//
// Timeline of a block: all loads that do not depend on the pose (input depths, the previous launch's
// partial rows) are requested at entry.  Wavefront 0 then finalises the previous iteration; meanwhile the
// pixel wavefronts project their pixels with the pose of BEFORE that update and gather the model map there
// (the update moves few projections across a pixel boundary), and compute the input normals.  With the new
// pose only pixels whose reference index changed are gathered again.
// The leading scalar parameters repeat fields of `a`: the dispatcher preloads the first kernel-argument dwords
// into scalar registers, so the loads that start the critical path (partial rows, control block, depth) can
// be issued without first waiting for the argument block to arrive from memory.
template <bool PACKED, int PW, int PPT>
__global__ __launch_bounds__(64 * (1 + PW)) void k_icp_iteration(const float *__restrict__ pre_partials_in,
                                                                Ctrl *__restrict__ pre_ctrl,
                                                                const float *__restrict__ pre_in_depth,
                                                                unsigned long long *pre_stamps,
                                                                int pre_pending_rows, int pre_seq, int pre_iw,
                                                                int pre_ih, IcpIterArgs a) {
    constexpr int NT = 64 * (1 + PW);
    a.partials_in = pre_partials_in;
    a.ctrl = pre_ctrl;
    a.in_depth = pre_in_depth;
    a.stamps = pre_stamps;
    a.pending_rows = pre_pending_rows;
    a.seq = pre_seq;
    a.iw = pre_iw;
    a.ih = pre_ih;
    __shared__ IcpShared<PW> sh;
    KFH_STAMP(0, 0);
    if (KFH_STAMPS_COMPILED && a.stamps != nullptr && threadIdx.x == 0 && blockIdx.x < KFH_STAMP_BLOCKS && a.seq < KFH_STAMP_LAUNCHES)
        a.stamps[((size_t)a.seq * KFH_STAMP_BLOCKS + blockIdx.x) * KFH_STAMP_SLOTS + 12] = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pw = wave - 1;                          // pixel wavefront index; wavefront 0 has no pixels
    const IcpState *in = &a.ctrl->icp[a.seq & 1];
    IcpState *out = &a.ctrl->icp[(a.seq + 1) & 1];

    PartialRows<NT> pr;
    load_partial_rows<NT>(pr, a.partials_in, a.pending_rows);

    const int tile_h = PW * PPT;
    const int tiles_x = (a.iw + KFH_TRACK_TILE_W - 1) / KFH_TRACK_TILE_W;
    const int tile = (int)blockIdx.x + a.block0;      // block0 != 0: this device evaluates a share of the image's tiles
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int x_raw = tile_x * KFH_TRACK_TILE_W + lane;
    const int y0 = tile_y * tile_h + max(pw, 0) * PPT;
    const int x = min(x_raw, a.iw - 1);               // clamped: every load stays inside the image
    const int xl = max(x - 1, 0), xr = min(x + 1, a.iw - 1);

    float dcol[PPT + 2], dlft[PPT], drgt[PPT];        // PACKED: rows y0-1 .. y0+PPT of the column, left, right
    float4 nin4[PPT];                                 // !PACKED
    int yy[PPT];
    bool live[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        live[k] = x_raw < a.iw && y0 + k < a.ih;
        yy[k] = min(y0 + k, a.ih - 1);
    }
    if (pw >= 0) {
        if (PACKED) {
#pragma unroll
            for (int k = 0; k < PPT + 2; ++k) dcol[k] = a.in_depth[x + min(max(y0 - 1 + k, 0), a.ih - 1) * a.iw];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                dlft[k] = a.in_depth[xl + yy[k] * a.iw];
                drgt[k] = a.in_depth[xr + yy[k] * a.iw];
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                dcol[k + 1] = a.in_depth[x + yy[k] * a.iw];
                nin4[k] = a.in_normal[x + yy[k] * a.iw];
            }
        }
    }

    // everything the block reads from the control block is requested here, ahead of the first barrier
    const int pend = in->pending_nblocks;             // uniform
    const int in_done_mask = in->done_mask;
    const Mat4 pose_old = in->pose;
    const Mat4 view = a.ctrl->project_ref;
    Mat4 rview;
    if (PACKED) rview = a.ctrl->raycast_view;
    if (pend == 0 && ((in_done_mask >> a.level) & 1)) {
        // the level finished in an earlier launch: carry the record over
        if (blockIdx.x == 0 && wave == 0) {
            if (lane < 32) out->sums[lane] = in->sums[lane];
            if (lane == 0) {
                out->pose = pose_old;
                out->pending_nblocks = 0;
                out->pending_level = a.level;
                out->pending_last = 0;
                out->done_mask = in_done_mask;
                out->iterations = in->iterations;
                out->slow_solves = in->slow_solves;
            }
        }
        return;
    }
    if (pend > 0) stage_partial_rows<NT>(sh.slots, pr, a.partials_in, pend);
    __syncthreads();
    KFH_STAMP(1, 0);

    const Mat4 invK = {{a.ik[0], 0, a.ik[1], 0, 0, a.ik[2], a.ik[3], 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    f3 vin[PPT], nin[PPT];
    float4 rsa[PPT], rsb[PPT];                        // gathered model map (PACKED: rsa = {t, normal})
    unsigned sidx[PPT];
    if (wave == 0) {
        icp_finalize_wave0<NT>(in, pose_old, pend, a.icp_threshold, sh.slots, sh.sums, &sh.pose,
                                              &sh.done_mask, &sh.iterations, &sh.slow_solves);
    } else {
        // Branch-free on purpose: the PPT pixels are independent dependency chains (divisions, square
        // roots) that the scheduler can only interleave inside one basic block.
        if (PACKED) {
            f3 vcol[PPT + 2];                         // vertices of the column, rows y0-1 .. y0+PPT (clamped)
#pragma unroll
            for (int k = 0; k < PPT + 2; ++k)
                vcol[k] = back_project(dcol[k], x, min(max(y0 - 1 + k, 0), a.ih - 1), invK);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                // vertex2normalKernel: the vertices above and below are the column's own neighbours
                vin[k] = vcol[k + 1];
                const f3 left = back_project(dlft[k], xl, yy[k], invK);
                const f3 right = back_project(drgt[k], xr, yy[k], invK);
                const f3 up = vcol[k], down = vcol[k + 2];
                const bool hole = left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0;
                const f3 n = normalize3(cross3(down - up, right - left));
                nin[k] = hole ? mk3(KFH_INVALID, 0.f, 0.f) : n;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                vin[k] = back_project(dcol[k + 1], x, yy[k], invK);
                nin[k] = mk3(nin4[k].x, nin4[k].y, nin4[k].z);
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            // where the pixel projected with the previous pose (approximate arithmetic is enough here);
            // gathering there brings the model map lines into the caches before the new pose is known
            const f3 pp = xform(view, xform(pose_old, vin[k]));
            const float r = __builtin_amdgcn_rcpf(pp.z);
            const float sx = fminf(fmaxf(pp.x * r + 0.5f, 0.f), (float)(a.rw - 1));
            const float sy = fminf(fmaxf(pp.y * r + 0.5f, 0.f), (float)(a.rh - 1));
            sidx[k] = (unsigned)sx + (unsigned)sy * (unsigned)a.rw;
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (PACKED) {
                rsa[k] = a.ref_tn[sidx[k]];
            } else {
                rsa[k] = a.ref_normal[sidx[k]];
                rsb[k] = a.ref_vertex[sidx[k]];
            }
        }
    }
    KFH_STAMP(2, 0);
    KFH_STAMP(7, 1);
    KFH_STAMP(8, 4);
    __syncthreads();
    KFH_STAMP(3, 1);

    if (sh.done_mask & (1 << a.level)) {             // the pending iteration finished this level
        if (blockIdx.x == 0 && wave == 0) icp_store_state(out, sh, 0, a.level, 0);
        return;
    }

    if (wave != 0) {
        Mat4 Ttrack;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            Ttrack.m[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sh.pose.m[i])));
        f3 projectedVertex[PPT], projectedPos[PPT];
        bool img_ok[PPT];
        float px[PPT], py[PPT], rxf[PPT], ryf[PPT];
        bool all_fast = true;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            projectedVertex[k] = xform(Ttrack, vin[k]);
            projectedPos[k] = xform(view, projectedVertex[k]);
            const f3 p = projectedPos[k];
            // x / z and y / z share one reciprocal; bit-identical to IEEE division while no intermediate leaves
            // the normal range, which the range test guarantees
            const float az = fabsf(p.z), ax = fabsf(p.x), ay = fabsf(p.y);
            const bool fast = az > 1e-12f && az < 1e12f && ax < 1e12f && ay < 1e12f && ax > 1e-12f * az &&
                              ay > 1e-12f * az;
            all_fast = all_fast && fast;
            const float r1 = refined_rcp(p.z);
            px[k] = div_shared(p.x, p.z, r1) + 0.5f;
            py[k] = div_shared(p.y, p.z, r1) + 0.5f;
        }
        if (!all_fast) {                              // rare: operands near the ends of the float range
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                px[k] = projectedPos[k].x / projectedPos[k].z + 0.5f;
                py[k] = projectedPos[k].y / projectedPos[k].z + 0.5f;
            }
        }
        unsigned ridx[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            img_ok[k] = !(px[k] < 0 || px[k] > a.rw - 1 || py[k] < 0 || py[k] > a.rh - 1);
            const unsigned rx = img_ok[k] ? (unsigned)px[k] : 0u, ry = img_ok[k] ? (unsigned)py[k] : 0u;
            rxf[k] = (float)rx;
            ryf[k] = (float)ry;
            ridx[k] = rx + ry * (unsigned)a.rw;
        }
        // the speculative gathers have done their work (their lines are cached); retire them
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            asm volatile("" ::"v"(rsa[k].x), "v"(rsa[k].y), "v"(rsa[k].z), "v"(rsa[k].w));
            if (!PACKED) asm volatile("" ::"v"(rsb[k].x), "v"(rsb[k].y), "v"(rsb[k].z), "v"(rsb[k].w));
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (PACKED) {
                rsa[k] = a.ref_tn[ridx[k]];
            } else {
                rsa[k] = a.ref_normal[ridx[k]];
                rsb[k] = a.ref_vertex[ridx[k]];
            }
        }
        KFH_STAMP(4, 1);
        float acc[32];
        int code[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            f3 referenceNormal, refVertex;
            if (PACKED) {
                // the raycast stored vertex = origin + direction * t with direction = rotate(view, (x, y, 1)):
                // the same expression on the same matrix gives the same bits
                const f3 rorigin = mk3(rview.m[3], rview.m[7], rview.m[11]);
                const f3 rdir = rotate(rview, mk3(rxf[k], ryf[k], 1.f));
                const float t = rsa[k].x;
                const f3 p = rorigin + rdir * t;
                refVertex = t > 0.0f ? p : mk3(0.f, 0.f, 0.f);
                referenceNormal = mk3(rsa[k].y, rsa[k].z, rsa[k].w);
            } else {
                refVertex = mk3(rsb[k].x, rsb[k].y, rsb[k].z);
                referenceNormal = mk3(rsa[k].x, rsa[k].y, rsa[k].z);
            }
            const f3 diff = refVertex - projectedVertex[k];
            const f3 projectedNormal = rotate(Ttrack, nin[k]);
            const bool dist_bad = dot3(diff, diff) > a.dist2_max;      // length(diff) > dist_threshold
            const bool normal_bad = dot3(projectedNormal, referenceNormal) < a.normal_threshold;
            const int result = nin[k].x == KFH_INVALID ? -1
                               : !img_ok[k]            ? -2
                               : referenceNormal.x == KFH_INVALID ? -3
                               : dist_bad              ? -4
                               : normal_bad            ? -5
                                                       : 1;
            code[k] = result;
            const bool valid = result == 1;
            const f3 c = cross3(projectedVertex[k], referenceNormal);
            const float error = valid ? dot3(referenceNormal, diff) : 0.f;
            float J[6];
            J[0] = valid ? referenceNormal.x : 0.f; J[1] = valid ? referenceNormal.y : 0.f;
            J[2] = valid ? referenceNormal.z : 0.f;
            J[3] = valid ? c.x : 0.f; J[4] = valid ? c.y : 0.f; J[5] = valid ? c.z : 0.f;
            if (!PACKED && a.track_out && live[k]) {
                kfh_track_data td;
                td.result = result;
                td.error = error;
                for (int i = 0; i < 6; ++i) td.J[i] = J[i];
                a.track_out[x + yy[k] * a.track_stride] = td;
            }
            // Rows and columns past the image edge repeat the edge pixel: they add zeros.  The products go into
            // the sums by fused multiply-add: one rounding instead of the reference's two, a difference of half a
            // unit in the last place of a term, far below what the different order of summation already makes.
            const bool counted = live[k] && valid;
            const float e = counted ? error : 0.f;
            float Jl[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) Jl[i] = counted ? J[i] : 0.f;
            // the first pixel starts the sums (a product is the fused multiply-add onto zero), the others add
            const bool first = k == 0;
            acc[0] = first ? e * e : fmaf(e, e, acc[0]);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[1 + i] = first ? e * Jl[i] : fmaf(e, Jl[i], acc[1 + i]);
            int q = 7;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) {
                    acc[q] = first ? Jl[i] * Jl[j] : fmaf(Jl[i], Jl[j], acc[q]);
                    ++q;
                }
            const float c28 = (live[k] && valid) ? 1.f : 0.f, c29 = (live[k] && result == -4) ? 1.f : 0.f;
            const float c30 = (live[k] && result == -5) ? 1.f : 0.f;
            const float c31 = (live[k] && result < 1 && result > -4) ? 1.f : 0.f;
            acc[28] = first ? c28 : acc[28] + c28;
            acc[29] = first ? c29 : acc[29] + c29;
            acc[30] = first ? c30 : acc[30] + c30;
            acc[31] = first ? c31 : acc[31] + c31;
        }
        KFH_STAMP(5, 1);
        KFH_STAMP(9, 4);
        if (a.result_out != nullptr) {               // only when the tracking render was asked for
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if (live[k]) a.result_out[x + yy[k] * a.result_stride] = (signed char)code[k];
        }
        const float total = wave_reduce32(acc);
        if ((lane & 1) == 0) sh.part[pw][lane >> 1] = total;
        KFH_STAMP(10, 4);
    }
    __syncthreads();
    if (wave == 0) {
        if (lane < 32) {
            float t = sh.part[0][lane];
#pragma unroll
            for (int w = 1; w < PW; ++w) t += sh.part[w][lane];
            a.partials_out[blockIdx.x * 32 + lane] = t;
        }
        // rows_reduced: a launch between this one and the next folds the rows (and those of the other devices)
        // into a single row
        if (blockIdx.x == 0) icp_store_state(out, sh, a.rows_reduced ? 1 : (int)gridDim.x, a.level, a.last_iter);
    }
    KFH_STAMP(6, 0);
    if (KFH_STAMPS_COMPILED && a.stamps != nullptr && threadIdx.x == 0 && blockIdx.x < KFH_STAMP_BLOCKS && a.seq < KFH_STAMP_LAUNCHES)
        a.stamps[((size_t)a.seq * KFH_STAMP_BLOCKS + blockIdx.x) * KFH_STAMP_SLOTS + 13] = __builtin_amdgcn_s_memtime();
}

// largest float d2 with sqrtf(d2) <= t  (sqrtf is monotonic and correctly rounded on host and device)
static float max_arg_with_sqrt_le(float t) {
    if (!(t >= 0.f)) return -1.f;
    float c = t * t;
    while (sqrtf(c) > t) c = nextafterf(c, 0.f);
    while (sqrtf(nextafterf(c, INFINITY)) <= t) c = nextafterf(c, INFINITY);
    return c;
}

// The instantiated block shapes: every (pixel wavefronts, rows per thread) pair track_shape can return.
#define KFH_ICP_SHAPE(PW_, PPT_)                                                                              \
    if (a.pw == PW_ && a.ppt == PPT_) {                                                                      \
        hipLaunchKernelGGL((k_icp_iteration<PACKED, PW_, PPT_>), dim3(nb), dim3(64 * (1 + PW_)), 0, s,      \
                           a.partials_in, a.ctrl, a.in_depth, a.stamps, a.pending_rows, a.seq, a.iw, a.ih, a); \
        return true;                                                                                         \
    }
template <bool PACKED>
static bool launch_icp_shape(const IcpIterArgs &a, int nb, hipStream_t s) {
    KFH_ICP_SHAPE(3, 1) KFH_ICP_SHAPE(4, 1) KFH_ICP_SHAPE(5, 1) KFH_ICP_SHAPE(6, 1) KFH_ICP_SHAPE(8, 1)
    KFH_ICP_SHAPE(4, 2) KFH_ICP_SHAPE(8, 2)
    KFH_ICP_SHAPE(4, 3) KFH_ICP_SHAPE(7, 3) KFH_ICP_SHAPE(8, 3)
    KFH_ICP_SHAPE(4, 5) KFH_ICP_SHAPE(8, 5) KFH_ICP_SHAPE(4, 8) KFH_ICP_SHAPE(8, 8)
    return false;
}
#undef KFH_ICP_SHAPE

hipError_t launch_icp_iteration(const IcpIterArgs &a_in, hipStream_t s) {
    IcpIterArgs a = a_in;
    const bool packed = a.ref_vertex == nullptr;
    if (packed && (a.in_normal != nullptr || a.ref_tn == nullptr || a.track_out != nullptr)) return hipErrorInvalidValue;
    if (!packed && (a.in_normal == nullptr || a.ref_normal == nullptr)) return hipErrorInvalidValue;
    if (a.pw == 0 || a.ppt == 0) track_shape(a.iw, a.ih, a.max_blocks > 0 ? a.max_blocks : 256, &a.pw, &a.ppt);
    a.dist2_max = max_arg_with_sqrt_le(a.dist_threshold);
    {
        const Mat4 ik = inverse_camera_matrix(a.k[0], a.k[1], a.k[2], a.k[3]);
        a.ik[0] = ik.m[0]; a.ik[1] = ik.m[2]; a.ik[2] = ik.m[5]; a.ik[3] = ik.m[6];
    }
    const int all = track_blocks(a.iw, a.ih, a.pw, a.ppt);
    if (a.block0 < 0 || a.block_count < 0 || a.block0 + a.block_count > all) return hipErrorInvalidValue;
    const int nb = a.block_count > 0 ? a.block_count : all - a.block0;      // block_count == 0: all tiles from block0 on
    if (nb <= 0) return hipErrorInvalidValue;
    bool ok = false;
    ok = packed ? launch_icp_shape<true>(a, nb, s) : launch_icp_shape<false>(a, nb, s);
    if (!ok) return hipErrorInvalidValue;
    return hipGetLastError();
}

// End of tracking by one wavefront whose lanes all hold the same values: checkPoseKernel, the matrices derived
// from the accepted pose, the control block, the first ICP record of the next frame and the status record.  The
// new contents are assembled in LDS as dword images of the structures and copied out by all lanes (coalesced);
// one lane storing some 270 dwords one after the other took several microseconds.
constexpr int KFH_CTRL_HEAD_DWORDS = (int)(offsetof(Ctrl, icp) + sizeof(IcpState)) / 4;   // everything up to icp[0]
constexpr int KFH_STATUS_DWORDS = (int)sizeof(kfh_frame_status) / 4;
#define KFH_CTRL_DW(field) ((int)(offsetof(Ctrl, field) / 4))
#define KFH_STATUS_DW(field) ((int)(offsetof(kfh_frame_status, field) / 4))

__device__ inline void put_mat(unsigned *img, int at, const Mat4 &m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) img[at + i] = __builtin_bit_cast(unsigned, m.m[i]);
}

__device__ inline void finish_tracking_wave(const FinishArgs &a, Mat4 pose, const Mat4 &old_pose, const float *sums_lds,
                                            int iterations, int slow_solves, unsigned *ctrl_img, unsigned *status_img) {
    const int lane = threadIdx.x & 63;
    // checkPoseKernel
    const float s0 = sums_lds[0], s28 = sums_lds[28];
    const bool rejected = (sqrtf(s0 / s28) > 2e-2f) || (s28 / (float)(a.w * a.h) < a.track_threshold);
    const int tracked = rejected ? 0 : 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) pose.m[i] = rejected ? old_pose.m[i] : pose.m[i];
    const int integrated = ((tracked && (a.frame % a.integration_rate) == 0) || a.frame <= 3) ? 1 : 0;
    const Mat4 inv = mat_inverse(pose);
    const Mat4 view = mat_mul(pose, inverse_camera_matrix(a.k[0], a.k[1], a.k[2], a.k[3]));
    const bool rc = a.will_raycast != 0;
    if (lane == 0) {
        put_mat(ctrl_img, KFH_CTRL_DW(pose), pose);
        put_mat(ctrl_img, KFH_CTRL_DW(old_pose), pose);
        put_mat(ctrl_img, KFH_CTRL_DW(inv_pose), inv);
        put_mat(ctrl_img, KFH_CTRL_DW(view), view);
        if (rc) {
            // the model maps are re-rendered from this pose: K * inverse(raycast_pose) = K * inv
            put_mat(ctrl_img, KFH_CTRL_DW(raycast_pose), pose);
            put_mat(ctrl_img, KFH_CTRL_DW(raycast_view), view);
            put_mat(ctrl_img, KFH_CTRL_DW(project_ref), mat_mul(camera_matrix(a.k[0], a.k[1], a.k[2], a.k[3]), inv));
        }
        ctrl_img[KFH_CTRL_DW(tracked)] = (unsigned)tracked;
        ctrl_img[KFH_CTRL_DW(integrate_gate)] = (unsigned)integrated;
        ctrl_img[KFH_CTRL_DW(frame)] = (unsigned)a.frame;
        ctrl_img[KFH_CTRL_DW(ray_task_count)] = 0u;        // this frame's raycast appends its unfinished rays from here
        ctrl_img[KFH_CTRL_DW(integrate_box_count)] = 0u;   // likewise the work list of this frame's integration
        ctrl_img[KFH_CTRL_DW(integrate_box_count) + 1] = 0u;
        for (int i = 0; i < 2; ++i) ctrl_img[KFH_CTRL_DW(pad_) + i] = 0u;
        // the record the next frame's first iteration starts from
        const int st = KFH_CTRL_DW(icp);
        put_mat(ctrl_img, st + (int)(offsetof(IcpState, pose) / 4), pose);
        for (int i = (int)(offsetof(IcpState, pending_nblocks) / 4); i < (int)(sizeof(IcpState) / 4); ++i) ctrl_img[st + i] = 0u;
        status_img[KFH_STATUS_DW(frame)] = (unsigned)a.frame;
        status_img[KFH_STATUS_DW(tracked)] = (unsigned)tracked;
        status_img[KFH_STATUS_DW(integrated)] = (unsigned)integrated;
        status_img[KFH_STATUS_DW(raycasted)] = (unsigned)a.will_raycast;
        put_mat(status_img, KFH_STATUS_DW(pose), pose);
        status_img[KFH_STATUS_DW(icp_iterations)] = (unsigned)iterations;
        status_img[KFH_STATUS_DW(slow_solves)] = (unsigned)slow_solves & 0xFFFFu;
        status_img[KFH_STATUS_DW(examined_solves)] = (unsigned)slow_solves >> 16;
        status_img[KFH_STATUS_DW(reserved_)] = 0u;
    }
    if (lane < 32) {
        const unsigned v = __builtin_bit_cast(unsigned, sums_lds[lane]);
        ctrl_img[KFH_CTRL_DW(sums) + lane] = v;
        ctrl_img[KFH_CTRL_DW(icp) + (int)(offsetof(IcpState, sums) / 4) + lane] = v;
        status_img[KFH_STATUS_DW(sums) + lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned *c = reinterpret_cast<unsigned *>(a.ctrl);
    for (int d = lane; d < KFH_CTRL_HEAD_DWORDS; d += 64) {
        const bool raycast_field = (d >= KFH_CTRL_DW(raycast_pose) && d < KFH_CTRL_DW(inv_pose)) ||
                                   (d >= KFH_CTRL_DW(raycast_view) && d < KFH_CTRL_DW(sums));
        const bool running_total = d >= KFH_CTRL_DW(integrate_boxes_total) && d < KFH_CTRL_DW(icp);
        if ((rc || !raycast_field) && !running_total) c[d] = ctrl_img[d];
    }
    if (a.status_ring) {
        unsigned *st = reinterpret_cast<unsigned *>(a.status_ring + (a.frame % KFH_STATUS_RING));
        for (int d = lane; d < KFH_STATUS_DWORDS; d += 64) st[d] = status_img[d];
    }
    if (a.mail) {
        // the record in host memory: pose and flag first, the launch number behind a release so that a reader who
        // sees the number also sees the record
        if (lane < 16) a.mail[lane] = status_img[KFH_STATUS_DW(pose) + lane];
        if (lane == 16) a.mail[16] = (unsigned)tracked;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(a.mail + 31, a.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Last launch of tracking: finalise the last iteration, then checkPoseKernel and the derived matrices.  The
// leading scalar parameters repeat fields of `a` so that they arrive preloaded in registers.
__global__ __launch_bounds__(KFH_SUM_THREADS) void k_icp_finish(const float *__restrict__ pre_partials_in,
                                                                Ctrl *__restrict__ pre_ctrl, int pre_pending_rows,
                                                                int pre_seq, FinishArgs a) {
    __shared__ float slots[KFH_SUM_THREADS * 4];
    __shared__ float sums[32];
    __shared__ Mat4 pose;
    __shared__ int done_mask, iterations, slow_solves;
    __shared__ unsigned ctrl_img[KFH_CTRL_HEAD_DWORDS], status_img[KFH_STATUS_DWORDS];
    Ctrl *c = pre_ctrl;
    const IcpState *in = &c->icp[pre_seq & 1];
    PartialRows<KFH_SUM_THREADS> pr;
    load_partial_rows<KFH_SUM_THREADS>(pr, pre_partials_in, pre_pending_rows);
    const int pend = in->pending_nblocks;
    const Mat4 pose_in = in->pose;
    const Mat4 old_pose = c->old_pose;
    if (pend > 0) stage_partial_rows<KFH_SUM_THREADS>(slots, pr, pre_partials_in, pend);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    icp_finalize_wave0<KFH_SUM_THREADS>(in, pose_in, pend, a.icp_threshold, slots, sums, &pose, &done_mask, &iterations,
                                        &slow_solves);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    finish_tracking_wave(a, pose, old_pose, sums, iterations, slow_solves, ctrl_img, status_img);
}

hipError_t launch_icp_finish(const FinishArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_icp_finish, dim3(1), dim3(KFH_SUM_THREADS), 0, s, a.partials_in, a.ctrl, a.pending_rows, a.seq, a);
    return hipGetLastError();
}

__global__ void k_track_skip(FinishArgs a) {
    if (threadIdx.x != 0) return;
    finish_tracking(a, false, 0, 0);
    if (a.mail) {
        for (int i = 0; i < 16; ++i) a.mail[i] = __builtin_bit_cast(unsigned, a.ctrl->pose.m[i]);
        a.mail[16] = 0u;
        __hip_atomic_store(a.mail + 31, a.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_track_skip(const FinishArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_track_skip, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

// trackKernel + reduceKernel on caller-provided matrices (parity tests): one iteration launch whose state
// record carries Ttrack, then the fixed-order sum of its partial rows.
__global__ void k_icp_seed(Ctrl *c, Mat4 Ttrack, Mat4 view) {
    if (threadIdx.x != 0) return;
    c->project_ref = view;
    IcpState *st = &c->icp[0];
    st->pose = Ttrack;
    for (int i = 0; i < 32; ++i) st->sums[i] = 0.f;
    st->pending_nblocks = 0;
    st->pending_level = 0;
    st->pending_last = 0;
    st->done_mask = 0;
    st->iterations = 0;
    st->slow_solves = 0;
}

hipError_t launch_track_reduce(Ctrl *scratch_ctrl, float *partials, float *sums, kfh_track_data *track_out,
                               int track_stride, const float *in_depth, const float k[4], const float4 *in_normal,
                               int iw, int ih,
                               const float4 *ref_vertex, const float4 *ref_normal, int rw, int rh,
                               const Mat4 &Ttrack, const Mat4 &view, float dist_threshold, float normal_threshold,
                               hipStream_t s) {
    hipLaunchKernelGGL(k_icp_seed, dim3(1), dim3(64), 0, s, scratch_ctrl, Ttrack, view);
    IcpIterArgs a;
    a.ctrl = scratch_ctrl;
    a.partials_in = partials;
    a.partials_out = partials;
    a.pending_rows = 0;
    a.ref_tn = nullptr;
    track_shape(iw, ih, 256, &a.pw, &a.ppt);
    a.max_blocks = 256;
    a.seq = 0;
    a.level = 0;
    a.last_iter = 1;
    a.in_depth = in_depth; a.in_normal = in_normal; a.iw = iw; a.ih = ih;
    for (int i = 0; i < 4; ++i) a.k[i] = k[i];
    a.ref_vertex = ref_vertex; a.ref_normal = ref_normal; a.rw = rw; a.rh = rh;
    a.dist_threshold = dist_threshold; a.normal_threshold = normal_threshold;
    a.icp_threshold = 0.f;
    a.track_out = track_out; a.track_stride = track_stride;
    a.result_out = nullptr; a.result_stride = 0;
    a.stamps = nullptr;
    a.block0 = 0; a.block_count = 0; a.rows_reduced = 0;
    hipError_t e = launch_icp_iteration(a, s);
    if (e != hipSuccess) return e;
    return launch_sum_partials(sums, partials, track_blocks(iw, ih, a.pw, a.ppt), s);
}

__global__ void k_update_pose(float *pose16, double *x6, int *converged, const float *sums, float thr) {
    Mat4 pose;
    for (int i = 0; i < 16; ++i) pose.m[i] = pose16[i];
    double x[6];
    int used_jacobi;
    const bool conv = update_pose_wave(pose, sums, thr, x, &used_jacobi);
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 16; ++i) pose16[i] = pose.m[i];
    for (int i = 0; i < 6; ++i) x6[i] = x[i];
    converged[0] = conv ? 1 : 0;
    converged[1] = used_jacobi;
}

hipError_t launch_update_pose(float *pose16, double *x6, int *converged, const float *sums, float thr,
                              hipStream_t s) {
    hipLaunchKernelGGL(k_update_pose, dim3(1), dim3(64), 0, s, pose16, x6, converged, sums, thr);
    return hipGetLastError();
}

// Diagnostic: n dependent pose updates by one wavefront (each update perturbs the sums of the next), to time
// the solve + exponential + pose product chain without launch effects.
__global__ __launch_bounds__(64) void k_update_pose_chain(float *pose16, const float *sums, float thr, int n) {
    __shared__ float s[32];
    Mat4 pose;
    for (int i = 0; i < 16; ++i) pose.m[i] = pose16[i];
    if (threadIdx.x < 32) s[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
    double x[6];
    int path = 0;
    for (int it = 0; it < n; ++it) {
        update_pose_wave(pose, s, thr, x, &path);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (threadIdx.x == 1) s[1] = s[1] * (1.0f + 1e-7f * (float)(x[0] > 0 ? 1 : -1));
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < 16; ++i) pose16[i] = pose.m[i];
}

hipError_t launch_update_pose_chain(float *pose16, const float *sums, float thr, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_update_pose_chain, dim3(1), dim3(64), 0, s, pose16, sums, thr, n);
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
#ifndef KFH_INTEGRATE_UNROLL
#define KFH_INTEGRATE_UNROLL 8
#endif
#ifndef KFH_INTEGRATE_MIN_WAVES
#define KFH_INTEGRATE_MIN_WAVES 4
#endif
constexpr int INTEGRATE_UNROLL = KFH_INTEGRATE_UNROLL;     // z steps a lane keeps in flight
// smallest raw tsdf for which a trilinear sample is certainly >= 0.8:  26300 * 0.0000305194 = 0.8027, and
// the interpolation of values >= m stays above m * (1 - 1e-6)
constexpr int RAY_COARSE_SAFE = 26300;


// Correctly rounded float division with the denominator's reciprocal shared between several numerators.
// This is the instruction sequence of the compiler's own IEEE division (reciprocal estimate, one Newton
// step, two quotient corrections) without the exponent pre-scaling, which only matters for operands near
// the ends of the float range; the integrate kernel divides metres and pixels by a depth >= 1e-4.
// Footprint of a box in x and y (256 columns, one per lane of a workgroup).  With 128 x 2 a wavefront's 8-byte accesses
// in the two-column march cover 512 contiguous bytes of one row instead of two 256-byte pieces in two rows; the bare
// access pattern is 10 % faster that way, the march itself is not (dense 512^3: 219 against 220 us), and the longer
// boxes are culled worse (the benchmark frame takes 4.76 M voxels instead of 3.03 M, 1.2 % fewer frames/s): 64 x 4.
#ifndef KFH_BOX_X
#define KFH_BOX_X 64
#endif
constexpr int BOX_X = KFH_BOX_X, BOX_Y = 256 / BOX_X;
static_assert(BOX_X == 64 || BOX_X == 128, "a box is 64 x 4 or 128 x 2 columns");

struct IntegrateArgs {
    short2 *vol;
    int sx, sy, sz;
    float dimx, dimy, dimz;
    int z0, z1;                 // stored / owned planes
    const float *depth;
    int w, h;
    const Mat4 *inv_track;
    Mat4 K;
    float mu, maxweight;
    const int *gate;            // device flag: 0 = this frame does not integrate
    int zchunk;
    int gx, gy, gz;             // boxes of BOX_X x BOX_Y x zchunk voxels
    uint2 *boxes;               // work items: up to 32 boxes of one (bx, by) column that follow each other in z and may
                                //   touch the camera frustum.  x = bx | by << 8 | first bz << 20 | inside-the-image << 28
                                //   (every voxel of the item is in front of the camera and projects into the image),
                                //   y = mask of its boxes
    unsigned *box_count;        // two counters: items appended from the front of the list (the long ones: they start
                                //   first, so that the launch does not end with one of them) and from its back
    unsigned list_capacity;     // entries of `boxes`
    int min_group;              // boxes per work item: a power of two between min_group and 32, chosen per launch so
    int target_items;           //   that about target_items items come out (from the previous launch's box count)
    unsigned char *bricks;      // per 8^3 brick: 1 while no voxel of the brick (plus one voxel past it) went
    int bgx, bgy;               //   below RAY_COARSE_SAFE since the last reset; may be null
    const float *tile_max;      // largest depth per preprocessing tile (PRE_TX x PRE_TY pixels, apron included); may be null
    int tiles_x, tiles_y;
    unsigned long long *totals; // running totals {boxes kept, launches with an open gate, boxes kept up to the previous
                                //   launch, boxes kept by the previous launch}; may be null
    unsigned long long *dbg;
    int vec2;                   // items inside the image are marched two columns per lane (even sx, volume 8-byte aligned,
                                //   0 <= maxweight <= 32767)
};

// Conservative frustum culling of the box of BOX_X x BOX_Y x (zc1 - zc0) voxels at (bx * BOX_X, by * BOX_Y, zc0), grown by
// two voxels per side.  A voxel is only ever touched if it passes the half-space tests of the march; if all
// eight corners of the grown box fail the same half-space, no voxel of the box can pass it (the two voxels
// of slack are far more than the drift of the incrementally accumulated positions).
// Returns 0 when no voxel of the box can be touched, 2 when every voxel of the box certainly lies in front of the
// camera and projects inside the image (with two pixels to spare: the march may then leave out its own tests),
// 1 otherwise.
__device__ inline int box_class(const IntegrateArgs &a, const Mat4 &invTrack, int bx, int by, int zc0,
                                int zc1, const float *tile_max_lds, const float *coarse_max_lds,
                                bool depth_test) {
    const float wmax = (float)(a.w - 1), hmax = (float)(a.h - 1);
    const float vx = a.dimx / (float)a.sx, vy = a.dimy / (float)a.sy, vz = a.dimz / (float)a.sz;
    const float bx0 = ((float)(bx * BOX_X - 2) + 0.5f) * vx;
    const float bx1 = ((float)(min(bx * BOX_X + BOX_X - 1, a.sx - 1) + 2) + 0.5f) * vx;
    const float by0 = ((float)(by * BOX_Y - 2) + 0.5f) * vy;
    const float by1 = ((float)(min(by * BOX_Y + BOX_Y - 1, a.sy - 1) + 2) + 0.5f) * vy;
    const float bz0 = ((float)(zc0 - 2) + 0.5f) * vz;
    const float bz1 = ((float)(zc1 + 1) + 0.5f) * vz;
    const f3 c0 = xform(invTrack, mk3(bx0, by0, bz0));
    const f3 ex = rotate(invTrack, mk3(bx1 - bx0, 0.f, 0.f));
    const f3 ey = rotate(invTrack, mk3(0.f, by1 - by0, 0.f));
    const f3 ez = rotate(invTrack, mk3(0.f, 0.f, bz1 - bz0));
    bool behind = true, left = true, right = true, up = true, down = true;
    bool all_front = true;
    float zmin = 3.0e38f, pxmin = 3.0e38f, pxmax = -3.0e38f, pymin = 3.0e38f, pymax = -3.0e38f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f3 P = c0;
        if (q & 1) P = P + ex;
        if (q & 2) P = P + ey;
        if (q & 4) P = P + ez;
        const float cxp = a.K.m[0] * P.x + a.K.m[2] * P.z;
        const float cyp = a.K.m[5] * P.y + a.K.m[6] * P.z;
        behind = behind && (P.z < 0.0001f);
        left = left && (cxp < -1.5f * P.z);
        right = right && (cxp > (wmax + 1.5f) * P.z);
        up = up && (cyp < -1.5f * P.z);
        down = down && (cyp > (hmax + 1.5f) * P.z);
        all_front = all_front && (P.z > 0.01f);
        zmin = fminf(zmin, P.z);
        const float px = cxp / P.z, py = cyp / P.z;
        pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px);
        pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
    }
    if (behind || left || right || up || down) return 0;
    // Inside the image: with every corner of the grown box in front of the camera, every voxel of the box projects
    // into the bounding rectangle of the corners' projections (the projection of a convex set in front of the
    // camera is the convex hull of the projected corners).  The march adds 0.5 and accepts [0, w - 1] x [0, h - 1];
    // two pixels are kept in hand (the accumulated positions differ from exact ones by far less than a pixel).
    // The accumulated positions drift from the exact ones by a rounding residual that repeats from plane to plane, so
    // the margin grows with the number of planes (2 pixels at 512 planes were ample; + 1 pixel per 512 planes).  Such
    // an item also lies more than mu in front of the camera: a pixel without depth (0) then reads as "behind the
    // surface" (0 - z <= -mu) with no test of its own.
    const float margin = 2.f + (float)a.z1 * (1.f / 512.f);
    const bool interior = all_front && zmin > a.mu + 0.01f && pxmin >= margin && pxmax <= wmax - 1.f - margin &&
                          pymin >= margin && pymax <= hmax - 1.f - margin;
    // Behind the surface: a voxel is left untouched when the depth seen through its pixel is at least mu in
    // front of it.  With every corner of the grown box in front of the camera the box projects inside the
    // bounding rectangle of its corners; if even the largest depth of the tiles under that rectangle (4 pixels
    // of slack) ends mu before the nearest corner, no voxel of the box can be updated.
    if (depth_test && all_front) {
        const int x0 = max((int)fmaxf(floorf(pxmin), -8.f) - 4, 0), x1 = min((int)fminf(ceilf(pxmax), wmax + 8.f) + 4, a.w - 1);
        const int y0 = max((int)fmaxf(floorf(pymin), -8.f) - 4, 0), y1 = min((int)fminf(ceilf(pymax), hmax + 8.f) + 4, a.h - 1);
        const int tx0 = x0 / PRE_TX, tx1 = x1 / PRE_TX, ty0 = y0 / PRE_TY, ty1 = y1 / PRE_TY;
        // One thread reads the maxima under its rectangle from the block's LDS copies: tile by tile for a small
        // rectangle, otherwise from the 4 x 4-tile cells that cover it.  A single loop over the cells, four
        // reads in flight: the lanes of a wavefront have rectangles of different shapes, and nested loops would
        // run for the largest width times the largest height among them.  A box near the camera covers much
        // of the image and is rarely behind everything seen there: it is not tested.
        if (x0 <= x1 && y0 <= y1) {
            float maxd = 3.0e38f;
            const bool fine_cells = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= 16;
            const int ox = fine_cells ? tx0 : tx0 >> 2, oy = fine_cells ? ty0 : ty0 >> 2;
            const int nx = (fine_cells ? tx1 : tx1 >> 2) - ox + 1, ny = (fine_cells ? ty1 : ty1 >> 2) - oy + 1;
            const int pitch = fine_cells ? a.tiles_x : (a.tiles_x + 3) >> 2;
            const float *cells = fine_cells ? tile_max_lds : coarse_max_lds;
            const int n = nx * ny;
            if (n <= 24) {
                maxd = 0.f;
                const float inv_nx = 1.0f / (float)nx;
                for (int i = 0; i < n; i += 4) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = min(i + k, n - 1);
                        const int jy = (int)(((float)j + 0.5f) * inv_nx), jx = j - jy * nx;
                        v[k] = cells[(oy + jy) * pitch + ox + jx];
                    }
                    maxd = fmaxf(fmaxf(maxd, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                }
            }
            if (zmin - maxd > a.mu + 1e-3f) return 0;         // includes "no valid depth at all" for boxes past mu
        }
    }
    return interior ? 2 : 1;
}

// First launch: one thread per box decides whether the box can touch the frustum and appends the survivors
// to the list (typically a few percent of the volume).  The work launch then only runs on those, instead
// of dispatching a workgroup per box of the whole volume.
constexpr int KFH_PLAN_TILES_PER_THREAD = 8;
constexpr int KFH_PLAN_MAX_TILES = 256 * KFH_PLAN_TILES_PER_THREAD;     // depth tiles a plan block holds in LDS (640 x 480: 1200)
// cells of 4 x 4 tiles a plan block holds; narrow, tall images have more cells per tile (160 x 3272 pixels: 5 x 409
// tiles make 2 x 103 cells), so the depth test is only taken when the cells fit
constexpr int KFH_PLAN_COARSE_CELLS = KFH_PLAN_MAX_TILES / 16 + 64;
// Boxes per work item for this launch.  A work item replays its column's additions from z = 0 up to its first box
// and then carries them from box to box, so longer items replay less; but fewer, longer items leave the device
// short of wavefronts while few boxes survive.  The previous integration's box count (frames resemble each other)
// decides: the smallest power of two that brings the expected number of items down to target_items.
__device__ inline int integrate_group_size(const IntegrateArgs &a) {
    const unsigned last = a.totals ? (unsigned)a.totals[3] : 0u;     // only the work launch writes it: stable here
    // No earlier launch to go by (a context's first integration, or a caller without running totals): the longest
    // runs.  Guessing "dense" for a sparse frame costs that one launch some wavefronts; guessing "sparse" for a dense
    // volume makes every box replay its column (three times the launch's time).
    if (last == 0u) return 32;
    // at least half of the volume was visited: the longest runs leave nothing to replay and still make more items
    // than the device has wavefront slots to fill
    if (last >= (unsigned)(a.gx * a.gy) * (unsigned)a.gz / 2u) return 32;
    int g = a.min_group;
    while (g < 32 && last / (unsigned)g > (unsigned)a.target_items) g <<= 1;
    return g;
}

__global__ __launch_bounds__(256) void k_integrate_plan(IntegrateArgs a) {
    __shared__ float tile_max_lds[KFH_PLAN_MAX_TILES];
    __shared__ float coarse_max_lds[KFH_PLAN_COARSE_CELLS];            // maxima of 4 x 4 tiles
    // everything the block reads from memory is requested at once, the gate included
    const int ntiles = a.tiles_x * a.tiles_y;
    const bool depth_test = a.tile_max != nullptr && ntiles <= KFH_PLAN_MAX_TILES && a.tiles_x <= 64 &&
                            ((a.tiles_x + 3) >> 2) * ((a.tiles_y + 3) >> 2) <= KFH_PLAN_COARSE_CELLS;
    float tm[KFH_PLAN_TILES_PER_THREAD];
#pragma unroll
    for (int k = 0; k < KFH_PLAN_TILES_PER_THREAD; ++k)
        tm[k] = depth_test ? a.tile_max[min((int)threadIdx.x + 256 * k, ntiles - 1)] : 0.f;
    const Mat4 inv_track = *a.inv_track;
    const int gate = a.gate ? *a.gate : 1;
    const int G = integrate_group_size(a);
    if (gate == 0) return;
    if (a.totals && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.totals + 1, 1ull);
    if (depth_test) {
#pragma unroll
        for (int k = 0; k < KFH_PLAN_TILES_PER_THREAD; ++k) tile_max_lds[threadIdx.x + 256 * k] = tm[k];
        __syncthreads();
        // four threads per cell, one tile row each, combined inside the quad
        const int cw = (a.tiles_x + 3) >> 2, ch = (a.tiles_y + 3) >> 2;
        for (int i = threadIdx.x; i < ((cw * ch * 4 + 255) & ~255); i += 256) {
            const int cell = min(i >> 2, cw * ch - 1), r = i & 3;
            const int cy = cell / cw, cx = cell - cy * cw;
            const int ty = min(cy * 4 + r, a.tiles_y - 1);
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmaxf(m, tile_max_lds[ty * a.tiles_x + min(cx * 4 + q, a.tiles_x - 1)]);
            m = fmaxf(m, __shfl_xor(m, 1, 64));
            m = fmaxf(m, __shfl_xor(m, 2, 64));
            if (r == 0 && (i >> 2) < cw * ch) coarse_max_lds[cell] = m;
        }
        __syncthreads();
    }
    // G neighbouring lanes hold the boxes of one work item (same x and y, consecutive in z); the first of them
    // appends the item with the mask of its surviving boxes
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int groups = (a.gz + G - 1) / G;
    const int nthreads = a.gx * a.gy * groups * G;
    bool keep = false, edge = false;
    int bx = 0, by = 0, g = 0;
    const int r = id & (G - 1);
    if (id < nthreads) {
        // x fastest, then y, then z: the lanes of a wavefront test boxes of similar depth and footprint, and
        // neighbouring items of the list touch neighbouring rows of the volume
        const int q = id / G;
        bx = q % a.gx;
        by = (q / a.gx) % a.gy;
        g = q / (a.gx * a.gy);
        const int bz = g * G + r;
        const int zc0 = max(a.z0, bz * a.zchunk), zc1 = min(a.z1, (bz + 1) * a.zchunk);
        const int cls = (bz < a.gz && zc0 < zc1)
                            ? box_class(a, inv_track, bx, by, zc0, zc1, tile_max_lds, coarse_max_lds, depth_test) : 0;
        keep = cls != 0;
        edge = cls == 1;
    }
    // List positions are handed out per workgroup: the wavefronts count their items in LDS and one lane adds the
    // block's totals to the two list counters with ONE 64-bit atomic (and to the running box total with another).
    // Every wavefront adding to the same words itself cost a dense volume's plan launch 18 us, most of it the
    // 1536 atomics queueing on three addresses.
    __shared__ unsigned blk_heavy, blk_light, blk_boxes, base_heavy, base_light;
    if (threadIdx.x == 0) { blk_heavy = 0u; blk_light = 0u; blk_boxes = 0u; }
    __syncthreads();
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63;
    unsigned gm = 0u, ge = 0u, off_h = 0u, off_t = 0u;
    unsigned long long heavy_mask = 0ull, light_mask = 0ull;
    bool leader = false, heavy = false, paired = false;
    if (m) {
        gm = (unsigned)((m >> (lane & ~(G - 1))) & ((1ull << G) - 1ull));
        // an item whose surviving boxes are all of the "inside the image" kind says so (bit 28)
        ge = (unsigned)((__ballot(edge) >> (lane & ~(G - 1))) & ((1ull << G) - 1ull));
        leader = r == 0 && gm != 0u;
        // Two items side by side in x (an even box column and the next one, whose groups sit G lanes apart in this
        // wavefront) with the same surviving boxes, all of the inside kind, trade halves (bit 29 in both): the even
        // one takes the first planes of both footprints, the odd one the rest, so that a wavefront marches whole
        // 512-byte rows of the volume.
        if (BOX_X == 64 && G <= 32 && a.vec2) {
            const bool odd = (bx & 1) != 0;
            const int peer = (odd ? lane - G : lane + G) & 63;
            const unsigned gm_peer = __shfl(gm, peer, 64), ge_peer = __shfl(ge, peer, 64);
            const bool peer_here = odd ? lane >= G : (lane + G < 64 && bx + 1 < a.gx);
            paired = peer_here && ge == 0u && ge_peer == 0u && gm_peer == gm && __popc(gm) >= 2;
        }
        const unsigned long long lm = __ballot(leader);
        const int first = __ffsll((long long)lm) - 1;
        // An item's cost grows with the planes it marches and, more slowly, with the planes it replays first.  Items in
        // the upper half of what an item of this launch can cost go to the front of the list, the others to its back;
        // the work launch takes the front first.
        // (a marched plane costs about sixteen replayed ones; single-box items all cost the same: they keep their order)
        const int cost = __popc(gm) * a.zchunk * 16 + g * G * a.zchunk;
        heavy = leader && (G == 1 || 2 * cost >= G * a.zchunk * 16 + a.gz * a.zchunk);
        heavy_mask = __ballot(heavy);
        light_mask = lm & ~heavy_mask;
        if (lane == first) {
            if (heavy_mask) off_h = atomicAdd(&blk_heavy, (unsigned)__popcll(heavy_mask));
            if (light_mask) off_t = atomicAdd(&blk_light, (unsigned)__popcll(light_mask));
            atomicAdd(&blk_boxes, (unsigned)__popcll(m));
        }
        off_h = __builtin_amdgcn_readlane(off_h, first);
        off_t = __builtin_amdgcn_readlane(off_t, first);
    }
    __syncthreads();
    if (threadIdx.x == 0 && blk_boxes != 0u) {
        const unsigned long long old =
            atomicAdd(reinterpret_cast<unsigned long long *>(a.box_count), ((unsigned long long)blk_light << 32) | blk_heavy);
        if (a.totals) atomicAdd(a.totals, (unsigned long long)blk_boxes);
        base_heavy = (unsigned)old;
        base_light = (unsigned)(old >> 32);
    }
    __syncthreads();
    if (leader) {
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned at = heavy ? base_heavy + off_h + __popcll(heavy_mask & below)
                                  : a.list_capacity - 1u - (base_light + off_t + __popcll(light_mask & below));
        a.boxes[at] =
            make_uint2((unsigned)bx | ((unsigned)by << 8) | ((unsigned)(g * G) << 20) | (ge == 0u ? 1u << 28 : 0u) |
                       (paired ? 1u << 29 : 0u), gm);
    }
}

// -----------------------------------------------------------------------------------------------------
// The march of a work item that lies wholly inside the image (and more than mu in front of the camera), with TWO
// x-adjacent columns per lane: voxels are read and written 8 bytes per lane (a dword per lane does not reach 60 % of
// the memory system's rate in this traversal whatever the arithmetic; 8 bytes do), and everything the two columns
// evaluate alike runs as packed f32 instructions.  The 256 columns of a box then need 128 lanes, so a block's
// wavefronts 0-1 march the first half of the item's planes and wavefronts 2-3 the second half.
//
// The planes are taken V2U at a time: a batch's voxels are requested first (their addresses need no arithmetic), then
// its positions are advanced and projected and its depth pixels requested, the batch is updated in registers and
// stored - loads and stores each in one burst.
//
// Per voxel the results are the reference's: the positions are its running sums (one IEEE addition per component and
// plane), the pixel comes from the correctly rounded quotients (shared-reciprocal division, here on packed pairs),
// and the update either takes the free-space shortcut on the raw word (value 32766 stays, weight + 1 capped) or the
// reference's expression sequence.
// -----------------------------------------------------------------------------------------------------
// cache policy bits of the voxel loads and stores of the two-column march (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef KFH_VOX_LOAD_AUX
#define KFH_VOX_LOAD_AUX 0
#endif
#ifndef KFH_VOX_STORE_AUX
#define KFH_VOX_STORE_AUX 0
#endif
// segments a wavefront pair cuts its planes into (the first one rotates from launch to launch); dense 512^3 work
// launch with 1 / 2 / 4 / 8 segments: 237 / 221 / 223 / 232 us (more segments replay more additions)
#ifndef KFH_INTEGRATE_SEGMENTS
#define KFH_INTEGRATE_SEGMENTS 2
#endif
#ifndef KFH_INTEGRATE_V2U
#define KFH_INTEGRATE_V2U 8
#endif
constexpr int V2U = KFH_INTEGRATE_V2U;      // planes per batch of the two-column march
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ inline v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline v2f splat2(float v) { return v2f{v, v}; }
// the two quotients n.x / d and n.y / d, each bit-identical to IEEE division (div_shared on a packed pair)
__device__ inline v2f div_shared2(v2f n, float d, float r1) {
    const v2f dd = splat2(-d), rr = splat2(r1);
    const v2f q0 = n * rr;
    const v2f q1 = pk_fma(pk_fma(dd, q0, n), rr, q0);
    return pk_fma(pk_fma(dd, q1, n), rr, q1);
}

struct Columns2 {            // running sums of the columns x (A) and x + 1 (B)
    v2f pxyA, pxyB;          // volume-to-camera position, x and y
    v2f cxyA, cxyB;          // K * position, x and y
    v2f pz, cz;              // z of both: {A, B}
};

// The reference's update of one voxel outside the free-space shortcut: `dz` = depth - camera z, (px, py, pz) the
// voxel in camera space.  Returns the raw word to store (the old one when the voxel stays untouched).
__device__ inline unsigned integrate_update_general(unsigned rawword, float dz, float px, float py, float pz, float mu,
                                                    float maxweight) {
    if (dz <= -mu) return rawword;
    float sdf = 1.f;
    if (dz < mu) {
        const float r1 = refined_rcp(pz);
        const float rx = div_shared(px, pz, r1), ry = div_shared(py, pz, r1);
        const float diff = dz * sqrtf(1 + rx * rx + ry * ry);
        if (!(diff > -mu)) return rawword;
        sdf = fminf(1.f, diff / mu);
    }
    const short rx16 = (short)(rawword & 0xFFFFu), ry16 = (short)(rawword >> 16);
    float tsdf = (float)rx16 * KFH_SHORT2FLOAT;
    float wgt = (float)ry16;
    tsdf = fmaxf(-1.f, fminf((wgt * tsdf + sdf) / (wgt + 1), 1.f));
    wgt = fminf(wgt + 1, maxweight);
    return ((unsigned)(unsigned short)(short)(tsdf * KFH_FLOAT2SHORT)) | ((unsigned)(unsigned short)(short)wgt << 16);
}

__device__ inline void integrate_clear_marks(const IntegrateArgs &a, int x, int y, int z) {
    const int ex = ((x & 7) == 0 && x > 0) ? 1 : 0, ey = ((y & 7) == 0 && y > 0) ? 1 : 0;
    const int ez = ((z & 7) == 0 && z > 0) ? 1 : 0;
    for (int iz = 0; iz <= ez; ++iz)
        for (int iy = 0; iy <= ey; ++iy)
            for (int ix = 0; ix <= ex; ++ix)
                a.bricks[((x >> 3) - ix) + a.bgx * (((y >> 3) - iy) + a.bgy * ((z >> 3) - iz))] = 0;
}

// One wavefront's share of an inside-the-image item: columns (x, x + 1) of row y, planes of the item's boxes clipped
// to [zlo, zhi).  Returns the number of voxels it stored (diagnostics).
__device__ inline unsigned integrate_item_inside2(const IntegrateArgs &a, const Mat4 &invTrack, const Mat4 &K, f3 delta,
                                                  f3 cameraDelta, int x, int y, int first_bz, unsigned item_mask,
                                                  int zlo, int zhi, int segments, int first_segment) {
    const int sx = a.sx, sy = a.sy, sz = a.sz;
    const float mu = a.mu;
    const bool in_vol = x < sx && y < sy;                  // sx is even: x + 1 < sx as well
    const unsigned col = ((unsigned)min(x, sx - 2) + (unsigned)min(y, sy - 1) * (unsigned)sx) * (unsigned)sizeof(short2);
    const unsigned plane_bytes = (unsigned)sx * (unsigned)sy * (unsigned)sizeof(short2);     // at most 2^26
    // depth pixels through a buffer descriptor: an index past the image reads 0 (no depth), whatever the lane holds
    const __amdgpu_buffer_rsrc_t depth_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.depth), 0, a.w * a.h * (int)sizeof(float), 0x00020000);
    const unsigned row_bytes = (unsigned)a.w * 4u;
    const unsigned capword = ((unsigned)(int)a.maxweight << 16) | 32766u;   // checked by the host: 0 <= maxweight <= 32767

    auto start_sums = [&](Columns2 &c) {
        const float py = ((float)y + 0.5f) * a.dimy / (float)sy, pz = (0.0f + 0.5f) * a.dimz / (float)sz;
        const f3 posA = xform(invTrack, mk3(((float)x + 0.5f) * a.dimx / (float)sx, py, pz));
        const f3 posB = xform(invTrack, mk3(((float)(x + 1) + 0.5f) * a.dimx / (float)sx, py, pz));
        const f3 camA = xform(K, posA), camB = xform(K, posB);
        c.pxyA = v2f{posA.x, posA.y}; c.pxyB = v2f{posB.x, posB.y};
        c.cxyA = v2f{camA.x, camA.y}; c.cxyB = v2f{camB.x, camB.y};
        c.pz = v2f{posA.z, posB.z}; c.cz = v2f{camA.z, camB.z};
    };
    const v2f dpxy = {delta.x, delta.y}, dcxy = {cameraDelta.x, cameraDelta.y};
    const v2f dpz = splat2(delta.z), dcz = splat2(cameraDelta.z);
    auto advance = [&](Columns2 &c) {
        c.pxyA += dpxy; c.pxyB += dpxy;
        c.cxyA += dcxy; c.cxyB += dcxy;
        c.pz += dpz; c.cz += dcz;
    };

    Columns2 c;
    start_sums(c);
    int zacc = 0;                                  // the sums hold the state of plane zacc
    unsigned written = 0;

    // A batch of V2U planes: slot u holds the voxels, the depth pixels and the camera z of one plane.
    uint2 raw[V2U], out[V2U];
    v2f dpix[V2U], camz[V2U];
    v2f sA, sB, sZ;          // position sums at the batch's first plane (only the band near a surface needs positions:
                             //   they are re-derived from here)
    int ze = 0;              // end of the run being marched

    // Voxels go through a buffer descriptor that starts at the batch's first plane: a plane is a scalar offset and the
    // column a 32-bit lane offset, so no lane computes a 64-bit address.
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t vol_rsrc;
    auto open_batch = [&](int zi) {
        vol_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char *>(a.vol) + (size_t)(zi - a.z0) * plane_bytes, 0, (int)(V2U * plane_bytes), 0x00020000);
    };
    // request the voxels of the u-th plane of the batch at zi (a plane past the end of the run is never stored: its
    // load re-reads the run's last plane)
    auto request = [&](int u, int zi) {
        const v2u v = __builtin_amdgcn_raw_buffer_load_b64(vol_rsrc, col, (unsigned)min(u, ze - 1 - zi) * plane_bytes, KFH_VOX_LOAD_AUX);
        raw[u] = make_uint2(v.x, v.y);
    };
    // project the plane the sums stand at into slot u, request its depth pixels, advance the sums
    auto project = [&](int u) {
        camz[u] = c.cz;
        const v2f r0 = {__builtin_amdgcn_rcpf(c.cz.x), __builtin_amdgcn_rcpf(c.cz.y)};
        const v2f r1 = pk_fma(pk_fma(-c.cz, r0, splat2(1.f)), r0, r0);          // refined_rcp of both
        const v2f pfA = div_shared2(c.cxyA, c.cz.x, r1.x) + splat2(0.5f);
        const v2f pfB = div_shared2(c.cxyB, c.cz.y, r1.y) + splat2(0.5f);
        // (24-bit multiply: the row is below 2^16 and a row's bytes below 2^24, checked by the host)
        const unsigned offA = ((unsigned)pfA.x << 2) + __umul24((unsigned)pfA.y, row_bytes);
        const unsigned offB = ((unsigned)pfB.x << 2) + __umul24((unsigned)pfB.y, row_bytes);
        dpix[u] = v2f{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(depth_rsrc, offA, 0, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(depth_rsrc, offB, 0, 0))};
        advance(c);
    };
    // update plane z from slot u (the plane is the u-th of its batch); the new words go to out[u]
    auto update = [&](int u, int z) {
        const v2f dz = dpix[u] - camz[u];
        const unsigned rawA = raw[u].x, rawB = raw[u].y;
        // free space seen as free space again: 32766 * KFH_SHORT2FLOAT is exactly 1.0f, so the reference's
        // (w * 1 + 1) / (w + 1) is 1 for every weight w >= 0 and only the weight moves
        const bool freeA = dz.x >= mu, freeB = dz.y >= mu;
        const bool plainA = (rawA & 0x8000FFFFu) == 32766u, plainB = (rawB & 0x8000FFFFu) == 32766u;
        unsigned outA = (freeA & plainA) ? min(rawA + 0x10000u, capword) : rawA;
        unsigned outB = (freeB & plainB) ? min(rawB + 0x10000u, capword) : rawB;
        // everything else that the reference touches: the band around the surface (-mu < dz < mu), and free space
        // over a voxel that has seen a surface.  (dz <= -mu: behind the surface, untouched; a pixel without depth
        // has dz = -camz < -mu, as the plan launch only calls an item inside when it lies beyond mu.)
        const bool slowA = (__builtin_fabsf(dz.x) < mu) | (freeA & !plainA);
        const bool slowB = (__builtin_fabsf(dz.y) < mu) | (freeB & !plainB);
        if (slowA | slowB) {
            v2f pxyA = sA, pxyB = sB, pz = sZ;
#pragma unroll
            for (int k = 0; k < u; ++k) { pxyA += dpxy; pxyB += dpxy; pz += dpz; }
            if (slowA) outA = integrate_update_general(rawA, dz.x, pxyA.x, pxyA.y, pz.x, mu, a.maxweight);
            if (slowB) outB = integrate_update_general(rawB, dz.y, pxyB.x, pxyB.y, pz.y, mu, a.maxweight);
            if (a.bricks && in_vol && z < ze) {
                if ((short)rawA >= RAY_COARSE_SAFE && (short)outA < RAY_COARSE_SAFE) integrate_clear_marks(a, x, y, z);
                if ((short)rawB >= RAY_COARSE_SAFE && (short)outB < RAY_COARSE_SAFE) integrate_clear_marks(a, x + 1, y, z);
            }
        }
        out[u] = make_uint2(outA, outB);
    };
    auto store = [&](int u, int z) {
        // free space seen often enough stays at (1, maxweight): nothing to store then
        if (in_vol & (z < ze) & ((out[u].x != raw[u].x) | (out[u].y != raw[u].y))) {
            __builtin_amdgcn_raw_buffer_store_b64(v2u{out[u].x, out[u].y}, vol_rsrc, col, (unsigned)u * plane_bytes, KFH_VOX_STORE_AUX);
            if (KFH_STAMPS_COMPILED) written += (out[u].x != raw[u].x) + (out[u].y != raw[u].y);
        }
    };

    // The planes [zlo, zhi) are taken as `segments` equal segments (cut at box boundaries), starting with segment
    // `first_segment` and going round: a launch that starts with the segment the launch before it ended with finds
    // those planes in the memory-side cache, which holds about half of a 512^3 volume.  A segment that does not
    // follow its predecessor in z reaches its sums by replaying the additions (from plane 0 when it lies below).
    const int seg_planes = ((zhi - zlo + segments - 1) / segments + a.zchunk - 1) / a.zchunk * a.zchunk;
    for (int pass = 0; pass < segments; ++pass) {
    const int seg = (first_segment + pass) % segments;
    const int clo = zlo + seg * seg_planes, chi = min(zhi, clo + seg_planes);
    if (clo >= chi) continue;
    unsigned mask = item_mask;
    while (mask) {
        // a run of boxes that follow each other without a gap
        const int k0 = __ffs(mask) - 1;
        const unsigned above = ~(mask >> k0);
        const int len = above ? __ffs(above) - 1 : 32 - k0;
        mask &= ~((len >= 32 ? ~0u : ((1u << len) - 1u)) << k0);
        const int zs = max(max(clo, a.z0), (first_bz + k0) * a.zchunk);
        ze = min(min(chi, a.z1), (first_bz + k0 + len) * a.zchunk);
        if (zs >= ze) continue;
        if (zacc > zs) {              // the previous run left the sums past this run's start
            start_sums(c);
            zacc = 0;
        }
#pragma unroll 4
        for (int z = zacc; z < zs; ++z) advance(c);
        // Per batch of V2U planes the memory operations go out in bursts (stores and loads interleaved with the
        // arithmetic plane by plane move the same bytes more slowly): the voxel loads back to back, then the
        // projections with the depth loads, the updates in registers, and the stores back to back.
        for (int zi = zs; zi < ze; zi += V2U) {
            open_batch(zi);
#pragma unroll
            for (int u = 0; u < V2U; ++u) request(u, zi);
            __builtin_amdgcn_sched_barrier(0);
            sA = c.pxyA; sB = c.pxyB; sZ = c.pz;
#pragma unroll
            for (int u = 0; u < V2U; ++u) project(u);
#pragma unroll
            for (int u = 0; u < V2U; ++u) update(u, zi + u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < V2U; ++u) store(u, zi + u);
            __builtin_amdgcn_sched_barrier(0);
        }
        zacc = zs + (ze - zs + V2U - 1) / V2U * V2U;
    }
    }
    return written;
}

// Second launch: a workgroup takes boxes from the list; one thread per (x, y) column of the box.
__global__ __launch_bounds__(256, KFH_INTEGRATE_MIN_WAVES) void k_integrate(IntegrateArgs a) {
    if (!KFH_STAMPS_COMPILED) a.dbg = nullptr;      // per-wavefront statistics only in diagnostic builds
    const unsigned nfront = a.box_count[0], nbox = nfront + a.box_count[1];
    if (a.totals && blockIdx.x == 0 && threadIdx.x == 0) {
        // the plan launch has ended: what it added to the running total is this integration's box count, which
        // the next plan launch sizes its work items by
        const unsigned long long t = a.totals[0];
        if (t != a.totals[2]) {
            a.totals[3] = t - a.totals[2];
            a.totals[2] = t;
        }
    }
    if (nbox == 0) return;
    // launches with an open gate so far (the plan launch has counted this one)
    const unsigned launch_number = a.totals ? __builtin_amdgcn_readfirstlane((unsigned)a.totals[1]) : 0u;
    // diagnostics (dbg != null): per box and wavefront {start, end} of the 100 MHz counter, candidates, written
    const Mat4 invTrack = *a.inv_track;
    const int sx = a.sx, sy = a.sy, sz = a.sz, w = a.w;
    const float dimx = a.dimx, dimy = a.dimy, dimz = a.dimz, mu = a.mu, maxweight = a.maxweight;
    const Mat4 K = a.K;
    const float wmax = (float)(a.w - 1), hmax = (float)(a.h - 1);
    const f3 delta = rotate(invTrack, mk3(0, 0, dimz / (float)sz));
    const f3 cameraDelta = rotate(K, delta);
    const size_t plane = (size_t)sx * sy;
    const float *__restrict__ depth = a.depth;
    for (unsigned bi = blockIdx.x; bi < nbox; bi += gridDim.x) {
        unsigned long long dbg_t0 = 0;
        unsigned dbg_cand = 0, dbg_written = 0;
        if (a.dbg) dbg_t0 = __builtin_amdgcn_s_memrealtime();
        const uint2 item = a.boxes[bi < nfront ? bi : a.list_capacity - 1u - (bi - nfront)];
        const int bx = item.x & 0xFF, by = (item.x >> 8) & 0xFFF, first_bz = (item.x >> 20) & 0xFF;
        unsigned group_mask = item.y;
        const bool item_inside = (item.x >> 28) & 1u;
        // (an item of a single box stays with the one-column march below: with two columns per lane every wavefront
        // replays twice the additions up to the box, which a single box does not pay back - 39.3 against 36.7 us per
        // integration of the benchmark frame, whose items are single boxes)
        if (a.vec2 && item_inside && __popc(group_mask) >= 2) {
            // two columns per lane: wavefronts 0-1 take the rows of the item's first planes, wavefronts 2-3 those of
            // the rest (split between two boxes; a single box is split in the middle)
            const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
            const unsigned imask = __builtin_amdgcn_readfirstlane(group_mask);
            const int ifirst = __builtin_amdgcn_readfirstlane(first_bz);
            const int nboxes = __popc(imask);
            int zsplit;
            if (nboxes >= 2) {
                unsigned m = imask;
                for (int i = 0; i < (nboxes + 1) / 2; ++i) m &= m - 1u;
                zsplit = (ifirst + __ffs(m) - 1) * a.zchunk;
            } else {
                const int bz = ifirst + __ffs(imask) - 1;
                const int zc0 = max(a.z0, bz * a.zchunk), zc1 = min(a.z1, (bz + 1) * a.zchunk);
                zsplit = zc0 + ((zc1 - zc0 + 1) / 2 + V2U - 1) / V2U * V2U;
            }
            // (an item that trades halves with its neighbour in x: each wavefront takes one row of both footprints,
            // 128 columns, over the first planes (even box column) or the rest (odd one))
            const bool wide = __builtin_amdgcn_readfirstlane((item.x >> 29) & 1u) != 0u;
            const bool second = wide ? (bx & 1) != 0 : wv >= 2;
            // each of the two takes its planes in segments, starting with another one from launch to launch
            const int span_lo = max(a.z0, (ifirst + __ffs(imask) - 1) * a.zchunk);
            const int span_hi = min(a.z1, (ifirst + 32 - __clz(imask)) * a.zchunk);
            const int zlo = second ? max(zsplit, span_lo) : span_lo;
            const int zhi = second ? span_hi : min(zsplit, span_hi);
            // (short items are not cut up: a segment is at least two boxes)
            const int segments = min(KFH_INTEGRATE_SEGMENTS, max(1, (zhi - zlo) / (2 * a.zchunk)));
            const int first_segment = (segments - (int)(launch_number % (unsigned)segments)) % segments;
            const unsigned stored =
                integrate_item_inside2(a, invTrack, K, delta, cameraDelta,
                                       wide ? (bx & ~1) * BOX_X + lane * 2 : bx * BOX_X + (lane % (BOX_X / 2)) * 2,
                                       wide ? by * BOX_Y + wv : by * BOX_Y + (wv & 1) * (BOX_Y / 2) + lane / (BOX_X / 2),
                                       ifirst, imask, zlo, zhi, segments, first_segment);
            if (a.dbg) {
                unsigned long long tot = stored;
                for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, 64);
                if (lane == 0) {
                    unsigned long long *d = a.dbg + ((size_t)bi * 4 + wv) * 4;
                    d[0] = dbg_t0;
                    d[1] = __builtin_amdgcn_s_memrealtime();
                    d[2] = 0;
                    d[3] = tot;
                }
            }
            continue;
        }
        const int x = bx * BOX_X + ((int)threadIdx.x & (BOX_X - 1));
        const int y = by * BOX_Y + (int)threadIdx.x / BOX_X;
        const bool in_vol = x < sx && y < sy;
        const f3 p0 = mk3(((float)x + 0.5f) * dimx / (float)sx, ((float)y + 0.5f) * dimy / (float)sy,
                          (0.0f + 0.5f) * dimz / (float)sz);
        // a voxel's address is a plane address (the same for the whole block: scalar registers) plus the column's
        // 32-bit offset inside a plane
        // (as a byte offset: a plane of at most 4096 x 4096 voxels takes 2^26 bytes)
        const unsigned col = ((unsigned)min(x, sx - 1) + (unsigned)min(y, sy - 1) * (unsigned)sx) * (unsigned)sizeof(short2);
        // The reference accumulates the camera-space position of a column step by step from z = 0.  The same
        // additions in the same order are replayed up to each box, two components per packed add; the boxes of a
        // group follow each other in z, so the replay is paid once per group and carried from box to box.
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f acc0, acc1, acc2;
        {
            const f3 pos0 = xform(invTrack, p0);
            const f3 cam0 = xform(K, pos0);
            acc0 = v2f{pos0.x, pos0.y}; acc1 = v2f{pos0.z, cam0.x}; acc2 = v2f{cam0.y, cam0.z};
        }
        const v2f d0 = {delta.x, delta.y}, d1 = {delta.z, cameraDelta.x}, d2 = {cameraDelta.y, cameraDelta.z};
        int zacc = 0;                    // the accumulators hold the state of step zacc
        while (group_mask) {
            const int bz = first_bz + __ffs(group_mask) - 1;
            group_mask &= group_mask - 1u;
            const int zc0 = max(a.z0, bz * a.zchunk);
            const int zc1 = min(a.z1, (bz + 1) * a.zchunk);
#pragma unroll 8
            for (int z = zacc; z < zc0; ++z) {
                acc0 += d0;
                acc1 += d1;
                acc2 += d2;
            }
            zacc = zc0;
            // A box that is not a whole number of batches (the first or last box of a z-slab) leaves the accumulators
            // past its end: the next box then replays from z = 0 (rare, and cheaper than carrying copies in registers)
            const bool whole_batches = ((zc1 - zc0) & (INTEGRATE_UNROLL - 1)) == 0;
            // INTEGRATE_UNROLL z steps are advanced arithmetically (the same packed additions as the replay), then
            // their depth pixels and voxels are fetched together (the voxel read is issued as soon as the pixel test
            // passes, before the depth value is known) and finally updated: one memory round trip per batch of
            // steps instead of two per voxel.
            auto march = [&](auto inside_tag) {
            constexpr bool INSIDE = decltype(inside_tag)::value;
            for (int zb = zc0; zb < zc1; zb += INTEGRATE_UNROLL) {
                v2f posxy[INTEGRATE_UNROLL];
                float posz[INTEGRATE_UNROLL], camz[INTEGRATE_UNROLL];
                unsigned pidx[INTEGRATE_UNROLL];
                bool cand[INTEGRATE_UNROLL];
#pragma unroll
                for (int u = 0; u < INTEGRATE_UNROLL; ++u) {
                    const int z = zb + u;
                    const float pz = acc1.x, cx = acc1.y, cy = acc2.x, cz = acc2.y;
                    posxy[u] = acc0;
                    posz[u] = pz;
                    camz[u] = cz;
                    // INSIDE: the plan launch has shown that every voxel of the item lies in front of the camera and
                    // projects inside the image, so the reference's tests all pass and are left out
                    bool c = in_vol && z < zc1 && (INSIDE || !(pz < 0.0001f));
                    // Cheap conservative rejection (one pixel of slack, far above the rounding of the exact test):
                    // a voxel that fails it cannot pass the exact in-image test below.
                    if (!INSIDE && c && (cx < -1.5f * cz || cx > (wmax + 1.5f) * cz || cy < -1.5f * cz || cy > (hmax + 1.5f) * cz))
                        c = false;
                    unsigned pi = 0;
                    if (c) {
                        const float r1 = refined_rcp(cz);
                        const float pxf = div_shared(cx, cz, r1) + 0.5f;
                        const float pyf = div_shared(cy, cz, r1) + 0.5f;
                        if (!INSIDE && (pxf < 0 || pxf > wmax || pyf < 0 || pyf > hmax)) c = false;
                        else pi = (unsigned)pxf + (unsigned)pyf * (unsigned)w;
                    }
                    cand[u] = c;
                    pidx[u] = pi;
                    acc0 += d0;
                    acc1 += d1;
                    acc2 += d2;
                }
                float dk[INTEGRATE_UNROLL];
                short2 rawk[INTEGRATE_UNROLL];
                short2 *const zplane = a.vol + (size_t)(zb - a.z0) * plane;
                {
                    const short2 *pz = zplane;
#pragma unroll
                    for (int u = 0; u < INTEGRATE_UNROLL; ++u) {
                        dk[u] = 0.f;
                        rawk[u] = make_short2(0, 0);
                        if (cand[u]) {
                            dk[u] = depth[pidx[u]];
                            rawk[u] = *reinterpret_cast<const short2 *>(reinterpret_cast<const char *>(pz) + col);
                        }
                        pz += plane;
                        if (a.dbg) dbg_cand += __popcll(__ballot(cand[u]));
                    }
                }
                short2 *pz_out = zplane;
#pragma unroll
                for (int u = 0; u < INTEGRATE_UNROLL; ++u, pz_out += plane) {
                    if (!cand[u] || dk[u] == 0) continue;
                    // diff = (d - z) * s with s = sqrt(1 + rx^2 + ry^2) >= 1, so
                    //   d - z <= -mu  implies  diff <= -mu        (voxel behind the surface: untouched)
                    //   d - z >=  mu  implies  min(1, diff / mu) == 1   (free space)
                    // and only the band in between needs s.
                    const float dz = dk[u] - camz[u];
                    if (dz <= -mu) continue;
                    float sdf = 1.f;
                    if (dz < mu) {
                        const float r1 = refined_rcp(posz[u]);
                        const float rx = div_shared(posxy[u].x, posz[u], r1), ry = div_shared(posxy[u].y, posz[u], r1);
                        const float diff = dz * sqrtf(1 + rx * rx + ry * ry);
                        if (!(diff > -mu)) continue;
                        sdf = fminf(1.f, diff / mu);
                    }
                    {
                        short2 out;
                        if (sdf == 1.f && rawk[u].x == 32766 && rawk[u].y >= 0) {
                            // Free space seen as free space again.  32766 * KFH_SHORT2FLOAT is exactly 1.0f, so the
                            // reference's (w * 1 + 1) / (w + 1) is (w + 1) / (w + 1) = 1 for every weight w >= 0
                            // (all of w, w + 1 are exact integers in float): the value stays 32766 without any
                            // arithmetic, only the weight moves.
                            static_assert(32766.0f * KFH_SHORT2FLOAT == 1.0f, "the free-space shortcut needs 32766 -> 1.0f");
                            out = make_short2((short)32766, (short)fminf((float)rawk[u].y + 1, maxweight));
                        } else {
                            float tsdf = (float)rawk[u].x * KFH_SHORT2FLOAT;
                            float wgt = (float)rawk[u].y;
                            tsdf = fmaxf(-1.f, fminf((wgt * tsdf + sdf) / (wgt + 1), 1.f));
                            wgt = fminf(wgt + 1, maxweight);
                            out = make_short2((short)(tsdf * KFH_FLOAT2SHORT), (short)wgt);
                        }
                        // free space seen often enough stays at (1, maxweight): nothing to store then
                        if (out.x != rawk[u].x || out.y != rawk[u].y) {
                            *reinterpret_cast<short2 *>(reinterpret_cast<char *>(pz_out) + col) = out;
                            if (a.dbg) ++dbg_written;
                        }
                        // The voxel drops below the level that lets a coarse raycast step pass unseen: every brick
                        // whose samples can read it (its own, and the previous one along each axis where the voxel
                        // is the first of its brick) loses its "nothing here" mark.  Marks are only ever cleared.
                        if (a.bricks && rawk[u].x >= RAY_COARSE_SAFE && out.x < RAY_COARSE_SAFE) {
                            const int z = zb + u;
                            const int ex = ((x & 7) == 0 && x > 0) ? 1 : 0, ey = ((y & 7) == 0 && y > 0) ? 1 : 0;
                            const int ez = ((z & 7) == 0 && z > 0) ? 1 : 0;
                            for (int iz = 0; iz <= ez; ++iz)
                                for (int iy = 0; iy <= ey; ++iy)
                                    for (int ix = 0; ix <= ex; ++ix)
                                        a.bricks[((x >> 3) - ix) + a.bgx * (((y >> 3) - iy) + a.bgy * ((z >> 3) - iz))] = 0;
                        }
                    }
                }
            }
            };
            if (item_inside) march(std::true_type{}); else march(std::false_type{});
            // the march has advanced the accumulators itself: after whole batches they hold the state of step zc1
            if (whole_batches) {
                zacc = zc1;
            } else {
                const f3 pos0 = xform(invTrack, p0);
                const f3 cam0 = xform(K, pos0);
                acc0 = v2f{pos0.x, pos0.y}; acc1 = v2f{pos0.z, cam0.x}; acc2 = v2f{cam0.y, cam0.z};
                zacc = 0;
            }
            }   // boxes of the item
        if (a.dbg) {
            unsigned long long tot = dbg_written;
            for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, 64);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long *d = a.dbg + ((size_t)bi * 4 + (threadIdx.x >> 6)) * 4;
                d[0] = dbg_t0;
                d[1] = __builtin_amdgcn_s_memrealtime();
                d[2] = dbg_cand;
                d[3] = tot;
            }
        }
    }
}

static unsigned long long *g_integrate_debug = nullptr;   // diagnostics buffer, [box][4 wavefronts][4]
void set_integrate_debug(unsigned long long *d_buffer) { g_integrate_debug = d_buffer; }

// z extent of a box (a multiple of INTEGRATE_UNROLL); KFH_INTEGRATE_ZCHUNK overrides it for experiments
static int integrate_zchunk(int z1) {
    static const int forced = std::getenv("KFH_INTEGRATE_ZCHUNK") ? std::atoi(std::getenv("KFH_INTEGRATE_ZCHUNK")) : 0;
    if (forced >= INTEGRATE_UNROLL && forced % INTEGRATE_UNROLL == 0) return forced;
    // measured per integration at 512^3: 16 -> 35 us, 32 -> 40 us, 64 -> 54 us; at 1024^3: 16 -> 121 us, 32 -> 102 us,
    // 64 -> 107 us; at 2048^3 (items of 256 planes): 64 -> 428 us, 32 -> 399 us (finer boxes are culled better)
    return z1 <= 512 ? 16 : 32;
}

// Smallest number of boxes per work item (a power of two up to 32; the plan launch raises it when the previous
// integration kept many boxes).  Up to 1024 planes single boxes are best while few survive; beyond that the replay
// of a column's additions (up to thousands of steps) outweighs the loss of wavefronts.  KFH_INTEGRATE_GROUP
// overrides it, KFH_INTEGRATE_TARGET_ITEMS the number of work items the plan launch aims for.
static int integrate_min_group(int z1) {
    static const int forced = std::getenv("KFH_INTEGRATE_GROUP") ? std::atoi(std::getenv("KFH_INTEGRATE_GROUP")) : 0;
    if (forced >= 1 && forced <= 32 && (forced & (forced - 1)) == 0) return forced;
    return z1 <= 1024 ? 1 : 8;
}
static int integrate_target_items() {
    static const int forced = std::getenv("KFH_INTEGRATE_TARGET_ITEMS") ? std::atoi(std::getenv("KFH_INTEGRATE_TARGET_ITEMS")) : 0;
    return forced > 0 ? forced : 4096;
}

int integrate_box_voxels(int z1) { return BOX_X * BOX_Y * integrate_zchunk(z1); }

size_t integrate_box_list_bytes(const int res[3], int z1) {
    const int zchunk = integrate_zchunk(z1);
    // an item holds at least min_group boxes' worth of z range
    const int g = integrate_min_group(z1);
    return (size_t)((res[0] + BOX_X - 1) / BOX_X) * ((res[1] + BOX_Y - 1) / BOX_Y) * (((z1 + zchunk - 1) / zchunk + g - 1) / g) * sizeof(uint2);
}

hipError_t launch_integrate(short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                            const float *depth, int w, int h, const Mat4 *inv_track, Mat4 K, float mu,
                            float maxweight, const int *gate, void *boxes, unsigned *box_count,
                            unsigned char *bricks, const float *tile_max, hipStream_t s,
                            unsigned long long *totals) {
    // Short z chunks: boxes that miss the camera frustum never reach the work launch, the others replay the
    // column prefix (a few packed adds per step) and then march only zchunk steps, so no wavefront runs long.
    IntegrateArgs a;
    a.vol = vol;
    a.sx = res[0]; a.sy = res[1]; a.sz = res[2];
    a.dimx = dim[0]; a.dimy = dim[1]; a.dimz = dim[2];
    a.z0 = z0; a.z1 = z1;
    a.depth = depth; a.w = w; a.h = h;
    a.inv_track = inv_track; a.K = K; a.mu = mu; a.maxweight = maxweight; a.gate = gate;
    a.zchunk = integrate_zchunk(z1);
    a.gx = (res[0] + BOX_X - 1) / BOX_X; a.gy = (res[1] + BOX_Y - 1) / BOX_Y; a.gz = (z1 + a.zchunk - 1) / a.zchunk;
    a.boxes = static_cast<uint2 *>(boxes); a.box_count = box_count;
    a.list_capacity = (unsigned)(integrate_box_list_bytes(res, z1) / sizeof(uint2));
    a.min_group = integrate_min_group(z1); a.target_items = integrate_target_items();
    a.bricks = bricks; a.bgx = (res[0] + 7) / 8; a.bgy = (res[1] + 7) / 8;
    a.tile_max = tile_max; a.tiles_x = (w + PRE_TX - 1) / PRE_TX; a.tiles_y = (h + PRE_TY - 1) / PRE_TY;
    a.totals = totals;
    a.dbg = g_integrate_debug;
    static const bool vec2_enabled = !(std::getenv("KFH_INTEGRATE_VEC2") && std::atoi(std::getenv("KFH_INTEGRATE_VEC2")) == 0);
    a.vec2 = vec2_enabled && (res[0] & 1) == 0 && (reinterpret_cast<uintptr_t>(vol) & 7u) == 0 && maxweight >= 0.f &&
             maxweight <= 32767.f && (long long)w * h * 4 < (1ll << 31) && w < (1 << 22) && h < (1 << 16);
    if (a.gx > 256 || a.gy > 4096 || a.gz > 256) return hipErrorInvalidValue;   // packed item
    // the plan launch decides its group size itself (1 to 32 lanes per work item): threads for the widest layout
    const int plan_threads = a.gx * a.gy * ((a.gz + 31) / 32) * 32;
    const int nitems = a.gx * a.gy * ((a.gz + a.min_group - 1) / a.min_group);
    static const int work_blocks = std::getenv("KFH_INTEGRATE_BLOCKS") ? std::atoi(std::getenv("KFH_INTEGRATE_BLOCKS")) : 2048;
    const dim3 plan_grid((plan_threads + 255) / 256), work_grid(nitems < work_blocks ? nitems : work_blocks);
    hipLaunchKernelGGL(k_integrate_plan, plan_grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_integrate, work_grid, dim3(256), 0, s, a);
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
constexpr int RAY_BATCH = 4;        // steps one lane keeps in flight for its own ray
constexpr int RAY_MAX_ROUNDS = 10;  // first pass: at most 40 samples per ray (the median ray needs about 32); 14 rounds for
                                    //   volumes beyond 1024 voxels per side, whose fine step is shorter
constexpr int RAY_MAX_HEAVY = 6;    // first pass: at most this many rounds that fetch voxels (frames/s with 4 / 5 / 6 / 7 / 8 /
                                    //   no limit: 5356 / 5610 / 5690 / 5682 / 5671 / 5640)
constexpr int RAY_FEW = 8;          // first pass ends for a wavefront with at most this many live rays
constexpr int RAY_TAIL_BLOCKS = 1024;

// value of `v` in lane `src`; src is the same for the whole wavefront
__device__ inline float bcast_lane(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// Two x-adjacent voxels fetched with one 8-byte load (the volume is only 4-byte aligned at odd x).
// The empty asm makes the whole 64-bit value live, so the load is not narrowed into two 2-byte loads.
typedef unsigned long long u64_align4 __attribute__((aligned(4)));
template <bool POW2, int OFFW>
struct VolDev {
    const short2 *data;
    int sx, sy, sz;
    float dx, dy, dz;
    int zs0, zs1;        // stored planes
    // size / dim; used when POW2 (sizes and dims are powers of two), where pos * size / dim == pos * (size / dim)
    // bit for bit because both scalings are exact
    float scx, scy, scz;
    // per 8^3 brick: non-zero while all its voxels (plus one past it per axis) are >= RAY_COARSE_SAFE; may be null
    const unsigned char *bricks;
    int bgx, bgy;
    // POW2: log2 of sx, of sx * sy, of bgx and of bgx * bgy (index arithmetic by shifts: a 32-bit integer
    // multiply issues at a quarter of the rate of a shift)
    int lsx, lsxy, lbgx, lbgxy;
    // OFFW == 0: the stored planes take less than 4 GiB, so byte offsets fit 32 bits and loads use base + 32-bit
    // offset addressing instead of a 64-bit address per lane.  OFFW == 1: voxel numbers still fit 32 bits (up to
    // 16 GiB; 1024^3).  OFFW == 2: voxel numbers are 64 bits wide (2048^3 on one device is 34 GB).
    static constexpr bool OFF32 = OFFW == 0;
    typedef typename std::conditional<OFFW == 2, unsigned long long, unsigned>::type Off;

    // index of the brick whose mark covers every tap of a sample at pos: the taps lie in [c, c + 1] per axis,
    // c = max(floor(scaled position), 0)
    __device__ inline unsigned brick_of(f3 pos) const {
        const f3 sp = to_voxel(pos);
        const int cx = min(max((int)floorf(sp.x), 0), sx - 1) >> 3;
        const int cy = min(max((int)floorf(sp.y), 0), sy - 1) >> 3;
        const int cz = min(max((int)floorf(sp.z), 0), sz - 1) >> 3;
        if (POW2) return (unsigned)cx + ((unsigned)cy << lbgx) + ((unsigned)cz << lbgxy);
        return (unsigned)cx + (unsigned)bgx * ((unsigned)cy + (unsigned)bgy * (unsigned)cz);
    }

    __device__ inline f3 to_voxel(f3 pos) const {
        if (POW2) return mk3(pos.x * scx - 0.5f, pos.y * scy - 0.5f, pos.z * scz - 0.5f);
        return mk3((pos.x * (float)sx / dx) - 0.5f, (pos.y * (float)sy / dy) - 0.5f,
                   (pos.z * (float)sz / dz) - 0.5f);
    }
    __device__ inline unsigned row(int y) const { return POW2 ? (unsigned)y << lsx : (unsigned)y * (unsigned)sx; }
    __device__ inline Off plane(int z) const {
        z = min(max(z, zs0), zs1 - 1) - zs0;
        return POW2 ? (Off)z << lsxy : (Off)z * (Off)(sx * sy);
    }
    // address of voxel number o
    __device__ inline const short2 *at(Off o) const {
        if (OFF32) return reinterpret_cast<const short2 *>(reinterpret_cast<const char *>(data) + (o << 2));
        return data + o;
    }
    __device__ inline float vs(int x, int y, int z) const { return (float)at((unsigned)x + row(y) + plane(z))->x; }
    // Trilinear sample of the raw tsdf shorts in three steps, so that a caller can put the fetches of
    // several samples in flight before it consumes any of them.  lx and ux are either equal or adjacent,
    // so each of the four (y, z) rows needs one 8-byte fetch at x0 = min(lx, sx - 2).
    struct Tap {
        Off o00, o10, o01, o11;        // voxel offsets of the four rows at x0
        f3 f;                          // interpolation fractions
        bool l_second, u_second;       // lower / upper x tap is the second voxel of the pair
    };
    struct TapData { unsigned long long r00, r10, r01, r11; };

    __device__ inline Tap prep(f3 pos) const {
        const f3 sp = to_voxel(pos);
        const float flx = floorf(sp.x), fly = floorf(sp.y), flz = floorf(sp.z);
        const int bx = (int)flx, by = (int)fly, bz = (int)flz;
        Tap t;
        t.f = mk3(sp.x - flx, sp.y - fly, sp.z - flz);
        // Inside the volume these are the reference's max(base, 0) / min(base + 1, size - 1).  The march also
        // fetches (and discards) a sample up to one coarse step past the far end of its ray, which can lie more
        // than a voxel outside the volume: rows are clamped on both sides so that such a fetch stays inside the
        // allocation (x0 and plane() already are).
        const int lx = max(bx, 0), ly = min(max(by, 0), sy - 1), lz = max(bz, 0);
        const int ux = min(bx + 1, sx - 1), uy = max(min(by + 1, sy - 1), 0), uz = min(bz + 1, sz - 1);
        const int x0 = min(lx, sx - 2);
        t.l_second = lx != x0;
        t.u_second = ux != x0;
        const unsigned rl = row(ly), ru = row(uy);
        const Off pl = plane(lz), pu = plane(uz);
        t.o00 = (unsigned)x0 + rl + pl;
        t.o10 = (unsigned)x0 + ru + pl;
        t.o01 = (unsigned)x0 + rl + pu;
        t.o11 = (unsigned)x0 + ru + pu;
        return t;
    }
    __device__ inline TapData fetch(const Tap &t) const {
        TapData d;
        d.r00 = *reinterpret_cast<const u64_align4 *>(at(t.o00));
        d.r10 = *reinterpret_cast<const u64_align4 *>(at(t.o10));
        d.r01 = *reinterpret_cast<const u64_align4 *>(at(t.o01));
        d.r11 = *reinterpret_cast<const u64_align4 *>(at(t.o11));
        return d;
    }
    __device__ static inline float tap_first(unsigned long long r) { return (float)(short)(r & 0xFFFFull); }
    __device__ static inline float tap_second(unsigned long long r) { return (float)(short)((r >> 32) & 0xFFFFull); }
    __device__ inline float combine(const Tap &t, TapData d) const {
        // the empty asm keeps each 64-bit value whole, so the loads are not narrowed to two 2-byte loads
        asm("" : "+v"(d.r00), "+v"(d.r10), "+v"(d.r01), "+v"(d.r11));
        const f3 f = t.f;
        const float v000 = t.l_second ? tap_second(d.r00) : tap_first(d.r00);
        const float v100 = t.u_second ? tap_second(d.r00) : tap_first(d.r00);
        const float v010 = t.l_second ? tap_second(d.r10) : tap_first(d.r10);
        const float v110 = t.u_second ? tap_second(d.r10) : tap_first(d.r10);
        const float v001 = t.l_second ? tap_second(d.r01) : tap_first(d.r01);
        const float v101 = t.u_second ? tap_second(d.r01) : tap_first(d.r01);
        const float v011 = t.l_second ? tap_second(d.r11) : tap_first(d.r11);
        const float v111 = t.u_second ? tap_second(d.r11) : tap_first(d.r11);
        return (((v000 * (1 - f.x) + v100 * f.x) * (1 - f.y) + (v010 * (1 - f.x) + v110 * f.x) * f.y) * (1 - f.z) +
                ((v001 * (1 - f.x) + v101 * f.x) * (1 - f.y) + (v011 * (1 - f.x) + v111 * f.x) * f.y) * f.z) *
               KFH_SHORT2FLOAT;
    }
    __device__ inline float interp(f3 pos) const {
        const Tap t = prep(pos);
        return combine(t, fetch(t));
    }
    // Smallest of the eight raw tsdf values of a fetched sample (the low 16 bits of each voxel).  A march
    // only needs the interpolated value when this minimum is small: with all eight values >= m > 0 the
    // interpolation is a sum of non-negative products and stays >= m (1 - 1e-6).
    __device__ static inline int min_tsdf(TapData d) {
        asm("" : "+v"(d.r00), "+v"(d.r10), "+v"(d.r01), "+v"(d.r11));
        typedef short s2 __attribute__((ext_vector_type(2)));
        const unsigned a = __builtin_amdgcn_perm((unsigned)(d.r00 >> 32), (unsigned)d.r00, 0x05040100u);
        const unsigned b = __builtin_amdgcn_perm((unsigned)(d.r10 >> 32), (unsigned)d.r10, 0x05040100u);
        const unsigned c = __builtin_amdgcn_perm((unsigned)(d.r01 >> 32), (unsigned)d.r01, 0x05040100u);
        const unsigned e = __builtin_amdgcn_perm((unsigned)(d.r11 >> 32), (unsigned)d.r11, 0x05040100u);
        const s2 m = __builtin_elementwise_min(
            __builtin_elementwise_min(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)),
            __builtin_elementwise_min(__builtin_bit_cast(s2, c), __builtin_bit_cast(s2, e)));
        return min((int)m.x, (int)m.y);
    }
    __device__ inline f3 grad(f3 pos) const {
        const f3 sp = to_voxel(pos);
        const float flx = floorf(sp.x), fly = floorf(sp.y), flz = floorf(sp.z);
        const int bx = (int)flx, by = (int)fly, bz = (int)flz;
        const f3 f = mk3(sp.x - flx, sp.y - fly, sp.z - flz);
        // the reference's clamps (one side each); the other side only acts on positions outside the volume, which
        // a hit never is, and keeps every fetch inside the allocation whatever the caller passes
        const int llx = min(max(bx - 1, 0), sx - 1), lly = min(max(by - 1, 0), sy - 1), llz = max(bz - 1, 0);
        const int lux = min(max(bx, 0), sx - 1), luy = min(max(by, 0), sy - 1), luz = max(bz, 0);
        const int ulx = max(min(bx + 1, sx - 1), 0), uly = max(min(by + 1, sy - 1), 0), ulz = min(bz + 1, sz - 1);
        const int uux = max(min(bx + 2, sx - 1), 0), uuy = max(min(by + 2, sy - 1), 0), uuz = min(bz + 2, sz - 1);
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

// Ray through pixel (x, y) and its parameter range inside the (slab of the) volume and [near, far].
template <bool POW2, int OFFW>
__device__ inline void ray_setup(const VolDev<POW2, OFFW> &vol, const Mat4 &view, int x, int y, float zc0, float zc1,
                                 float near_plane, float far_plane, f3 &direction, float &tnear, float &tfar) {
    const f3 origin = mk3(view.m[3], view.m[7], view.m[11]);
    direction = rotate(view, mk3((float)x, (float)y, 1.f));
    const f3 invR = mk3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
    const f3 box_lo = mk3(0.f, 0.f, zc0);
    const f3 box_hi = mk3(vol.dx, vol.dy, zc1);
    const f3 tbot = invR * (box_lo - origin);
    const f3 ttop = invR * (box_hi - origin);
    const f3 tmin = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    const f3 tmax = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    const float largest_tmin = fmaxf(fmaxf(tmin.x, tmin.y), fmaxf(tmin.x, tmin.z));
    const float smallest_tmax = fminf(fminf(tmax.x, tmax.y), fminf(tmax.x, tmax.z));
    tnear = fmaxf(largest_tmin, near_plane);
    tfar = fminf(smallest_tmax, far_plane);
}

// End of a ray: `hit` says the march stopped at a negative sample f_tt at parameter t, reached with step
// `stepsize` from the sample at t_prev.  Interpolates the zero crossing, evaluates the normal and writes
// the pixel of every output map.
template <bool POW2, int OFFW>
__device__ inline void finish_ray(const VolDev<POW2, OFFW> &vol, f3 origin, f3 direction, bool hit, float t, float t_prev,
                                  float stepsize, float f_tt, int o, float4 *__restrict__ vertex,
                                  float4 *__restrict__ normal, float *__restrict__ ray_t,
                                  float4 *__restrict__ ref_tn) {
    float4 hitv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hit) {
        const float f_t = vol.interp(origin + direction * t_prev);     // value of the last sample before the crossing
        const float th = t + stepsize * f_tt / (f_t - f_tt);
        const f3 p = origin + direction * th;
        hitv = make_float4(p.x, p.y, p.z, th);
    }
    float4 nrm = make_float4(KFH_INVALID, 0.f, 0.f, 0.f);
    float4 vtx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hitv.w > 0.0f) {
        vtx = make_float4(hitv.x, hitv.y, hitv.z, 0.f);
        const f3 g = vol.grad(mk3(hitv.x, hitv.y, hitv.z));
        if (length3(g) != 0) {
            const f3 n = normalize3(g);
            nrm = make_float4(n.x, n.y, n.z, 0.f);
        }
    }
    // the float4 maps are for callers; the pipeline's own raycast leaves them out (null) and has them built from
    // the packed map when somebody asks (k_model_maps)
    if (vertex) vertex[o] = vtx;
    if (normal) normal[o] = nrm;
    if (ray_t) ray_t[o] = hitv.w > 0.0f ? hitv.w : 0.f;
    // packed model map for tracking: hit distance + normal (the vertex is origin + direction * t)
    if (ref_tn) ref_tn[o] = make_float4(hitv.w > 0.0f ? hitv.w : 0.f, nrm.x, nrm.y, nrm.z);
}

// A ray the first pass left unfinished: next sample parameter, parameter of the last consumed sample, mode.
struct RayTask {
    int pixel;
    float t, t_prev;
    int fine;
};

// The reference loop per ray is
//   for (; t < tfar; t += stepsize) { f_tt = interp(t); if (f_tt < 0) break; if (f_tt < 0.8) stepsize = step; f_t = f_tt; }
// The same samples and decisions are taken here, but a sample is only interpolated when the smallest of its
// eight raw values allows an event (below RAY_COARSE_SAFE with the coarse step, negative with the fine
// step); f_t is evaluated once, at the end, from the remembered parameter of the previous sample.
//
// First pass: a wavefront is an 8x8 pixel tile and every lane marches its own ray, RAY_BATCH consecutive
// samples fetched together (their positions only depend on the current step size; a step-size change
// discards the rest of the batch).  A wavefront stops after RAY_MAX_ROUNDS batches, or when few of its
// rays are still alive, and appends those rays to the task list of the second pass.
template <bool POW2, int OFFW>
__global__ __launch_bounds__(256, 5) void k_raycast_march(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                       float *__restrict__ ray_t, float4 *__restrict__ ref_tn,
                                                       int w, int h, VolDev<POW2, OFFW> vol, float zc0, float zc1,
                                                       const Mat4 *__restrict__ view_p, float near_plane,
                                                       float far_plane, float step, float largestep,
                                                       RayTask *__restrict__ tasks, unsigned *task_count,
                                                       int max_rounds, int few, int max_heavy, unsigned long long *dbg) {
    // 16x16 pixel block, each wavefront an 8x8 tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // diagnostics (dbg != null): per wavefront {start, end} of the 100 MHz counter, batches, rays handed over
    if (!KFH_STAMPS_COMPILED) dbg = nullptr;        // per-wavefront statistics only in diagnostic builds
    unsigned long long dbg_t0 = 0;
    if (dbg) dbg_t0 = __builtin_amdgcn_s_memrealtime();
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = x < w && y < h;
    const Mat4 view = *view_p;
    const f3 origin = mk3(view.m[3], view.m[7], view.m[11]);
    f3 direction;
    float tnear, tfar;
    ray_setup(vol, view, x, y, zc0, zc1, near_plane, far_plane, direction, tnear, tfar);

    float t = tnear, t_prev = tnear, f_tt = 0.f;
    bool fine = false, hit = false, marching = false;
    if (inside && tnear < tfar) {
        // already inside the surface: nothing to render
        const f3 p0 = origin + direction * t;
        if (vol.bricks && vol.bricks[vol.brick_of(p0)]) {
            marching = true;
        } else {
            const auto tap = vol.prep(p0);
            const auto raw = vol.fetch(tap);
            marching = VolDev<POW2, OFFW>::min_tsdf(raw) > 0 ? true : vol.combine(tap, raw) > 0;
        }
    }

    int rounds = 0, heavy_rounds = 0;
    unsigned dbg_samples = 0, dbg_skipped = 0, dbg_heavy_rounds = 0;
    unsigned long long dbg_heavy_ticks = 0, dbg_loop_t0 = 0, dbg_loop_t1 = 0;
    if (dbg) dbg_loop_t0 = __builtin_amdgcn_s_memrealtime();
    while (true) {
        const unsigned long long alive = __ballot(marching);
        if (alive == 0 || (int)__popcll(alive) <= few || rounds == max_rounds || heavy_rounds >= max_heavy) break;
        ++rounds;
        bool dbg_round_heavy = false;
        unsigned long long dbg_round_t0 = 0;
        if (dbg) dbg_round_t0 = __builtin_amdgcn_s_memrealtime();
        if (marching) {
            const float stepsize = fine ? step : largestep;
            float tk[RAY_BATCH], fk[RAY_BATCH];
            f3 pos[RAY_BATCH];
            tk[0] = t;
#pragma unroll
            for (int k = 1; k < RAY_BATCH; ++k) tk[k] = tk[k - 1] + stepsize;
            typename VolDev<POW2, OFFW>::TapData raw[RAY_BATCH];
#pragma unroll
            for (int k = 0; k < RAY_BATCH; ++k) {
                const float ts = tk[k] < tfar ? tk[k] : t;      // never sample past the far bound
                pos[k] = origin + direction * ts;
            }
            // With the coarse step a sample in a brick marked "all voxels >= RAY_COARSE_SAFE" cannot be an event:
            // one byte per sample (the marks of a 512^3 volume fit the caches) instead of four 8-byte gathers.
            bool skip[RAY_BATCH];
#pragma unroll
            for (int k = 0; k < RAY_BATCH; ++k) skip[k] = false;
            if (vol.bricks && !fine) {
                unsigned char mark[RAY_BATCH];
#pragma unroll
                for (int k = 0; k < RAY_BATCH; ++k) mark[k] = vol.bricks[vol.brick_of(pos[k])];
#pragma unroll
                for (int k = 0; k < RAY_BATCH; ++k) skip[k] = mark[k] != 0;
            }
#pragma unroll
            for (int k = 0; k < RAY_BATCH; ++k) fk[k] = 1.0f;       // stands for "no event possible"
            if (!(skip[0] && skip[1] && skip[2] && skip[3])) {
                static_assert(RAY_BATCH == 4, "the skip test above spells out four samples");
#pragma unroll
                for (int k = 0; k < RAY_BATCH; ++k) {
                    raw[k].r00 = raw[k].r10 = raw[k].r01 = raw[k].r11 = 0ull;
                    if (!skip[k]) raw[k] = vol.fetch(vol.prep(pos[k]));
                }
                bool need[RAY_BATCH], any_need = false;
#pragma unroll
                for (int k = 0; k < RAY_BATCH; ++k) {
                    const int mn = VolDev<POW2, OFFW>::min_tsdf(raw[k]);
                    need[k] = !skip[k] && (fine ? mn < 0 : mn < RAY_COARSE_SAFE);
                    any_need = any_need || need[k];
                }
                if (any_need) {
#pragma unroll
                    for (int k = 0; k < RAY_BATCH; ++k) {
                        if (need[k]) {
                            f3 p = pos[k];
                            asm volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z));   // recompute the fractions here
                            fk[k] = vol.combine(vol.prep(p), raw[k]);
                        }
                    }
                }
            }
            if (dbg) {
#pragma unroll
                for (int k = 0; k < RAY_BATCH; ++k) {
                    dbg_samples += 1;
                    dbg_skipped += skip[k] ? 1 : 0;
                }
            }
            dbg_round_heavy = !(skip[0] && skip[1] && skip[2] && skip[3]);
            // The samples are consumed in order; written with selects instead of branches (a lane that has met
            // its event stops being `active`), so the four steps cost a few compares each.
            bool active = true;
#pragma unroll
            for (int k = 0; k < RAY_BATCH; ++k) {
                const bool in = tk[k] < tfar;
                const bool neg = active && in && fk[k] < 0;                  // zero crossing
                const bool cont = active && in && !(fk[k] < 0);
                const bool ends = active && !in;                             // left the volume: no hit
                const bool sw = cont && !fine && fk[k] < 0.8f;               // switch to the fine step
                t = neg ? tk[k] : (sw ? tk[k] + step : t);
                f_tt = neg ? fk[k] : f_tt;
                hit = hit || neg;
                t_prev = cont ? tk[k] : t_prev;
                fine = fine || sw;
                marching = marching && !(ends || neg);
                active = cont && !sw;
            }
            if (active) t = tk[RAY_BATCH - 1] + stepsize;
        }
        if (__ballot(dbg_round_heavy)) ++heavy_rounds;
        if (dbg && __ballot(dbg_round_heavy)) {
            dbg_heavy_rounds += 1;
            dbg_heavy_ticks += __builtin_amdgcn_s_memrealtime() - dbg_round_t0;
        }
    }
    if (dbg) dbg_loop_t1 = __builtin_amdgcn_s_memrealtime();

    // rays still marching go to the second pass
    const unsigned long long left = __ballot(marching);
    if (left) {
        const int first = __ffsll((long long)left) - 1;
        unsigned base = 0;
        if (lane == first) base = atomicAdd(task_count, (unsigned)__popcll(left));
        base = __builtin_amdgcn_readlane(base, first);
        if (marching) {
            RayTask e;
            e.pixel = x + y * w;
            e.t = t;
            e.t_prev = t_prev;
            e.fine = fine ? 1 : 0;
            tasks[base + __popcll(left & ((1ull << lane) - 1ull))] = e;
        }
    }
    if (inside && !marching)
        finish_ray(vol, origin, direction, hit, t, t_prev, fine ? step : largestep, f_tt, x + y * w, vertex, normal,
                   ray_t, ref_tn);
    if (dbg && lane == 0) {
        unsigned long long *d = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
        d[0] = dbg_t0;
        d[1] = __builtin_amdgcn_s_memrealtime();
        // rounds | rounds with voxel fetches << 8 | their ticks << 16;  rays left | ticks before the loop << 8 |
        // ticks after it << 20
        d[2] = (unsigned long long)rounds | ((unsigned long long)dbg_heavy_rounds << 8) | ((dbg_heavy_ticks & 0xFFFFull) << 16);
        d[3] = (unsigned long long)__popcll(left) | (((dbg_loop_t0 - dbg_t0) & 0xFFFull) << 8) |
               (((d[1] - dbg_loop_t1) & 0xFFFull) << 20);
    }
    if (dbg) {      // lane sums of samples taken / skipped by brick marks, packed into the record's upper halves
        unsigned long long a = dbg_samples, b = dbg_skipped;
        for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        if (lane == 0) {
            unsigned long long *d = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
            d[2] |= a << 32;
            d[3] |= b << 32;
        }
    }
}

// Second pass: RAY_TAIL_LANES lanes share one unfinished ray; lane k of the group evaluates step k of the
// next RAY_TAIL_LANES steps (parameters by repeated addition, exactly as the sequential loop produces
// them), and the first lane in step order that meets a loop exit or a step-size change decides.  Long
// rays (grazing a surface with the fine step) thus take a sixteenth of the dependent round trips, and
// the task list spreads them over the whole device instead of the few tiles they come from.
template <bool POW2, int OFFW, int RAY_TAIL_LANES>
__device__ inline void raycast_tail_block(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                          float *__restrict__ ray_t, float4 *__restrict__ ref_tn, int w, int h,
                                          const VolDev<POW2, OFFW> &vol, float zc0, float zc1,
                                          const Mat4 *__restrict__ view_p, float near_plane, float far_plane,
                                          float step, float largestep, const RayTask *__restrict__ tasks,
                                          const unsigned *__restrict__ task_count, unsigned block, unsigned blocks) {
    constexpr int GROUPS = 64 / RAY_TAIL_LANES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / RAY_TAIL_LANES, sub = lane % RAY_TAIL_LANES;
    const int grp_lane0 = grp * RAY_TAIL_LANES;
    const unsigned n = *task_count;
    const unsigned stride = blocks * 4 * GROUPS;
    const Mat4 view = *view_p;
    const f3 origin = mk3(view.m[3], view.m[7], view.m[11]);
    for (unsigned base = (block * 4 + wave) * GROUPS; base < n; base += stride) {      // wave-uniform
        const unsigned gi = base + grp;
        const bool have = gi < n;
        RayTask e;
        e.pixel = 0; e.t = 0.f; e.t_prev = 0.f; e.fine = 1;
        if (have) e = tasks[gi];
        const int py = e.pixel / w, px = e.pixel - py * w;
        f3 direction;
        float tnear, tfar;
        ray_setup(vol, view, px, py, zc0, zc1, near_plane, far_plane, direction, tnear, tfar);
        float bt = e.t, t_prev = e.t_prev, f_hit = 0.f;
        bool fine = e.fine != 0, go = have, hit = false;
        while (__ballot(go)) {                                                // wave-uniform
            const float stepsize = fine ? step : largestep;
            float tk = bt;                                                    // bt advanced `sub` times
#pragma unroll
            for (int i = 0; i < RAY_TAIL_LANES - 1; ++i) tk = sub > i ? tk + stepsize : tk;
            const bool valid = tk < tfar;
            const auto tap = vol.prep(origin + direction * (valid ? tk : bt));
            const auto raw = vol.fetch(tap);
            const int mn = VolDev<POW2, OFFW>::min_tsdf(raw);
            const bool need = go && valid && (fine ? mn < 0 : mn < RAY_COARSE_SAFE);
            float f = 1.0f;
            if (need) f = vol.combine(tap, raw);
            constexpr unsigned long long GMASK = RAY_TAIL_LANES == 64 ? ~0ull : ((1ull << (RAY_TAIL_LANES & 63)) - 1ull);
            const unsigned long long m_end = (__ballot(go && !valid) >> grp_lane0) & GMASK;
            const unsigned long long m_neg = (__ballot(go && valid && f < 0.f) >> grp_lane0) & GMASK;
            const unsigned long long m_near =
                (__ballot(go && valid && !fine && !(f < 0.f) && f < 0.8f) >> grp_lane0) & GMASK;
            const unsigned long long events = m_end | m_neg | m_near;
            const int first = events ? __ffsll((long long)events) - 1 : RAY_TAIL_LANES - 1;
            // all cross-lane reads are taken by every lane, outside divergent control flow
            const float t_first = __shfl(tk, grp_lane0 + first, 64);
            const float f_first = __shfl(f, grp_lane0 + first, 64);
            const float t_before = __shfl(tk, grp_lane0 + max(first - 1, 0), 64);
            if (go) {
                if (events == 0) {                          // all steps of the group consumed, none decisive
                    t_prev = t_first;                       // first == last lane here
                    bt = t_first + stepsize;
                } else if ((m_end >> first) & 1ull) {       // ran past the far bound: no hit
                    go = false;
                } else if ((m_neg >> first) & 1ull) {       // zero crossing
                    hit = true;
                    bt = t_first;
                    f_hit = f_first;
                    if (first > 0) t_prev = t_before;
                    go = false;
                } else {                                    // first sample below 0.8: switch to the fine step
                    fine = true;
                    t_prev = t_first;
                    bt = t_first + step;
                }
            }
        }
        if (have && sub == 0)
            finish_ray(vol, origin, direction, hit, bt, t_prev, fine ? step : largestep, f_hit, e.pixel, vertex, normal,
                       ray_t, ref_tn);
    }
}

template <bool POW2, int OFFW, int RAY_TAIL_LANES>
__global__ __launch_bounds__(256) void k_raycast_tail(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                      float *__restrict__ ray_t, float4 *__restrict__ ref_tn, int w,
                                                      int h, VolDev<POW2, OFFW> vol, float zc0, float zc1,
                                                      const Mat4 *__restrict__ view_p, float near_plane,
                                                      float far_plane, float step, float largestep,
                                                      const RayTask *__restrict__ tasks,
                                                      const unsigned *__restrict__ task_count) {
    raycast_tail_block<POW2, OFFW, RAY_TAIL_LANES>(vertex, normal, ray_t, ref_tn, w, h, vol, zc0, zc1, view_p, near_plane,
                                                   far_plane, step, largestep, tasks, task_count, blockIdx.x, gridDim.x);
}

// The second pass of one frame's raycast and the preprocessing of the next frame in one launch: neither depends
// on the other, each leaves most of the device idle on its own, and a launch boundary is saved.  The blocks of
// the second pass come first (its longest rays decide when the launch ends).
struct PreprocessArgs {
    float *raw, *filtered;
    int ow, oh;
    const void *in;
    int iw, ratio;
    float divisor;
    BilateralConst bc;
    float *level1, *level2;
    float pyramid_e_d;
    float *tile_max;
    int tiles_x, tiles_y;
};

template <bool POW2, int OFFW>
__global__ __launch_bounds__(256) void k_raycast_tail_preprocess(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                                 float *__restrict__ ray_t, float4 *__restrict__ ref_tn,
                                                                 int w, int h, VolDev<POW2, OFFW> vol, float zc0,
                                                                 float zc1, const Mat4 *__restrict__ view_p,
                                                                 float near_plane, float far_plane, float step,
                                                                 float largestep, const RayTask *__restrict__ tasks,
                                                                 const unsigned *__restrict__ task_count,
                                                                 unsigned tail_blocks, PreprocessArgs pre) {
    if (blockIdx.x < tail_blocks) {
        raycast_tail_block<POW2, OFFW, 32>(vertex, normal, ray_t, ref_tn, w, h, vol, zc0, zc1, view_p, near_plane,
                                           far_plane, step, largestep, tasks, task_count, blockIdx.x, tail_blocks);
    } else {
        const int t = (int)(blockIdx.x - tail_blocks);
        preprocess_tile<true, true>(pre.raw, pre.filtered, pre.ow, pre.oh, pre.in, pre.iw, pre.ratio, pre.divisor, pre.bc,
                                    pre.level1, pre.level2, pre.pyramid_e_d, pre.tile_max, t % pre.tiles_x,
                                    t / pre.tiles_x, pre.tiles_x);
    }
}

static unsigned long long *g_raycast_debug = nullptr;   // diagnostics buffer, [blocks][4 waves][4]
void set_raycast_debug(unsigned long long *d_buffer) { g_raycast_debug = d_buffer; }

template <bool POW2, int OFFW>
static hipError_t launch_raycast_t(float4 *vertex, float4 *normal, float *ray_t, float4 *ref_tn, int w, int h,
                                   const short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                                   const Mat4 *view, float near_plane, float far_plane, float step, float largestep,
                                   void *tasks, unsigned *task_count, const unsigned char *bricks, int part,
                                   const PreprocessLaunch *pre, hipStream_t s) {
    VolDev<POW2, OFFW> v;
    v.data = vol;
    v.bricks = bricks;
    v.bgx = (res[0] + 7) / 8;
    v.bgy = (res[1] + 7) / 8;
    auto ilog2 = [](int n) { int l = 0; while ((1 << l) < n) ++l; return l; };
    v.lsx = ilog2(res[0]);
    v.lsxy = v.lsx + ilog2(res[1]);
    v.lbgx = ilog2(v.bgx);
    v.lbgxy = v.lbgx + ilog2(v.bgy);
    v.sx = res[0]; v.sy = res[1]; v.sz = res[2];
    v.dx = dim[0]; v.dy = dim[1]; v.dz = dim[2];
    v.scx = (float)res[0] / dim[0]; v.scy = (float)res[1] / dim[1]; v.scz = (float)res[2] / dim[2];
    float zc0 = 0.f, zc1 = dim[2];
    if (z0 == 0 && z1 == res[2]) {
        v.zs0 = 0; v.zs1 = res[2];
    } else {
        // sharded: planes [z0-2, z1+2) are stored, the march is clipped to the owned z range
        v.zs0 = z0 - 2 > 0 ? z0 - 2 : 0;
        v.zs1 = z1 + 2 < res[2] ? z1 + 2 : res[2];
        // one voxel past the owned range (the halo holds it), so that a zero crossing between the last
        // sample of this slab and the first of the next is still seen here
        zc0 = (float)z0 * (dim[2] / (float)res[2]);
        zc1 = (float)(z1 < res[2] ? z1 + 1 : z1) * (dim[2] / (float)res[2]);
    }
    const dim3 grid((w + 15) / 16, (h + 15) / 16);
    // tuning knobs for experiments (defaults: RAY_MAX_ROUNDS, 32 lanes per unfinished ray)
    // measured, frames/s: 1024^3 with 10 / 14 / 20 / 28 rounds 4302 / 4228 / 4100 / 3892; 2048^3 with 10 / 14 / 20 / 28
    // rounds 2071 / 2302 / 2259 / 2196
    static const int forced_rounds = std::getenv("KFH_RAY_ROUNDS") ? std::atoi(std::getenv("KFH_RAY_ROUNDS")) : 0;
    const int res_max = res[0] > res[1] ? (res[0] > res[2] ? res[0] : res[2]) : (res[1] > res[2] ? res[1] : res[2]);
    const int max_rounds = forced_rounds > 0 ? forced_rounds : (res_max > 1024 ? 14 : RAY_MAX_ROUNDS);
    static const int few = std::getenv("KFH_RAY_FEW") ? std::atoi(std::getenv("KFH_RAY_FEW")) : RAY_FEW;
    // a wavefront whose rays keep fetching voxels (they run along a surface) hands them over after RAY_MAX_HEAVY such
    // rounds: the launch ends with its slowest wavefront, and the second pass spreads those rays over the device
    // (2048^3, frames/s with 6 / 8 / 10 / no limit: 2327 / 2385 / 2376 / 2367)
    static const int forced_heavy = std::getenv("KFH_RAY_HEAVY") ? std::atoi(std::getenv("KFH_RAY_HEAVY")) : 0;
    // (without brick marks every round fetches voxels: the limit then would end the first pass for all rays)
    const int max_heavy = forced_heavy > 0 ? forced_heavy
                          : bricks == nullptr ? 1 << 20 : (res_max > 1024 ? RAY_MAX_HEAVY + 2 : RAY_MAX_HEAVY);
    static const int tail_lanes = std::getenv("KFH_RAY_TAIL_LANES") ? std::atoi(std::getenv("KFH_RAY_TAIL_LANES")) : 32;
    if (part != RAYCAST_TAIL)
        hipLaunchKernelGGL((k_raycast_march<POW2, OFFW>), grid, dim3(256), 0, s, vertex, normal, ray_t, ref_tn, w, h, v, zc0, zc1,
                           view, near_plane, far_plane, step, largestep, static_cast<RayTask *>(tasks), task_count,
                           max_rounds, few, max_heavy, g_raycast_debug);
    if (part == RAYCAST_MARCH) return hipGetLastError();
    const RayTask *ct = static_cast<const RayTask *>(tasks);
    if (pre != nullptr && tail_lanes == 32) {
        // the second pass together with the preprocessing of the next frame
        PreprocessArgs pa;
        pa.raw = pre->raw; pa.filtered = pre->filtered; pa.ow = pre->ow; pa.oh = pre->oh; pa.in = pre->in;
        pa.iw = pre->iw; pa.ratio = pre->ratio; pa.divisor = pre->divisor; pa.bc = bilateral_const();
        pa.level1 = pre->level1; pa.level2 = pre->level2; pa.pyramid_e_d = pre->pyramid_e_d; pa.tile_max = pre->tile_max;
        pa.tiles_x = (pre->ow + PRE_TX - 1) / PRE_TX; pa.tiles_y = (pre->oh + PRE_TY - 1) / PRE_TY;
        const unsigned tail_blocks = RAY_TAIL_BLOCKS;
        hipLaunchKernelGGL((k_raycast_tail_preprocess<POW2, OFFW>), dim3(tail_blocks + pa.tiles_x * pa.tiles_y), dim3(256), 0,
                           s, vertex, normal, ray_t, ref_tn, w, h, v, zc0, zc1, view, near_plane, far_plane, step, largestep,
                           ct, task_count, tail_blocks, pa);
        return hipGetLastError();
    }
    if (tail_lanes == 64)
        hipLaunchKernelGGL((k_raycast_tail<POW2, OFFW, 64>), dim3(RAY_TAIL_BLOCKS * 2), dim3(256), 0, s, vertex, normal, ray_t,
                           ref_tn, w, h, v, zc0, zc1, view, near_plane, far_plane, step, largestep, ct, task_count);
    else if (tail_lanes == 32)
        hipLaunchKernelGGL((k_raycast_tail<POW2, OFFW, 32>), dim3(RAY_TAIL_BLOCKS), dim3(256), 0, s, vertex, normal, ray_t,
                           ref_tn, w, h, v, zc0, zc1, view, near_plane, far_plane, step, largestep, ct, task_count);
    else
        hipLaunchKernelGGL((k_raycast_tail<POW2, OFFW, 16>), dim3(RAY_TAIL_BLOCKS), dim3(256), 0, s, vertex, normal, ray_t,
                           ref_tn, w, h, v, zc0, zc1, view, near_plane, far_plane, step, largestep, ct, task_count);
    if (pre != nullptr)      // no fused variant for this lane count: the preprocessing follows on its own
        return launch_preprocess(pre->raw, pre->filtered, pre->ow, pre->oh, pre->in, true, pre->iw, pre->ratio, pre->divisor,
                                 pre->level1, pre->level2, pre->pyramid_e_d, pre->tile_max, s);
    return hipGetLastError();
}

size_t raycast_task_bytes(int w, int h) { return (size_t)w * h * sizeof(RayTask); }

hipError_t launch_raycast(float4 *vertex, float4 *normal, float *ray_t, float4 *ref_tn, int w, int h,
                          const short2 *vol, const int res[3], const float dim[3], int z0, int z1, const Mat4 *view,
                          float near_plane, float far_plane, float step, float largestep, void *tasks,
                          unsigned *task_count, const unsigned char *bricks, hipStream_t s) {
    return launch_raycast_part(vertex, normal, ray_t, ref_tn, w, h, vol, res, dim, z0, z1, view, near_plane, far_plane,
                               step, largestep, tasks, task_count, bricks, RAYCAST_BOTH, nullptr, s);
}

hipError_t launch_raycast_part(float4 *vertex, float4 *normal, float *ray_t, float4 *ref_tn, int w, int h,
                               const short2 *vol, const int res[3], const float dim[3], int z0, int z1,
                               const Mat4 *view, float near_plane, float far_plane, float step, float largestep,
                               void *tasks, unsigned *task_count, const unsigned char *bricks, int part,
                               const PreprocessLaunch *pre, hipStream_t s) {
    // (brick marks, when given, are indexed by the whole volume's brick grid; for a slab only bricks whose planes and
    // the plane above them are stored may carry a mark)
    if (res[0] < 2 || res[1] < 2 || res[2] < 2) return hipErrorInvalidValue;   // a fetch takes two x-adjacent voxels
    if (pre != nullptr && (part != RAYCAST_TAIL || !pre->filtered || !pre->level1 || !pre->level2 || pre->ow % 4 != 0 ||
                           pre->oh % 4 != 0))
        return hipErrorInvalidValue;
    bool pow2 = true;
    for (int i = 0; i < 3; ++i) {
        int e = 0;
        const bool dim_pow2 = std::frexp(dim[i], &e) == 0.5f;
        const bool res_pow2 = (res[i] & (res[i] - 1)) == 0;
        if (!dim_pow2 || !res_pow2 || res[i] < 2) pow2 = false;
    }
    const int stored = (z0 == 0 && z1 == res[2]) ? res[2] : (z1 + 2 < res[2] ? z1 + 2 : res[2]) - (z0 - 2 > 0 ? z0 - 2 : 0);
    const double stored_voxels = (double)res[0] * res[1] * stored;
    if ((double)res[0] * res[1] > 2147483647.0) return hipErrorInvalidValue;      // a plane is indexed with 32 bits
    const int offw = stored_voxels * sizeof(short2) <= 4294967296.0 - 16.0 ? 0 : stored_voxels <= 4294967296.0 - 4.0 ? 1 : 2;
#define KFH_RAYCAST_ARGS vertex, normal, ray_t, ref_tn, w, h, vol, res, dim, z0, z1, view, near_plane, far_plane, step, \
                         largestep, tasks, task_count, bricks, part, pre, s
    if (pow2)
        return offw == 0 ? launch_raycast_t<true, 0>(KFH_RAYCAST_ARGS)
             : offw == 1 ? launch_raycast_t<true, 1>(KFH_RAYCAST_ARGS) : launch_raycast_t<true, 2>(KFH_RAYCAST_ARGS);
    return offw == 0 ? launch_raycast_t<false, 0>(KFH_RAYCAST_ARGS)
         : offw == 1 ? launch_raycast_t<false, 1>(KFH_RAYCAST_ARGS) : launch_raycast_t<false, 2>(KFH_RAYCAST_ARGS);
#undef KFH_RAYCAST_ARGS
}

// Per pixel, keep the slab candidate with the smallest positive hit distance.  Candidates and result are packed
// maps {hit distance (0 = miss), normal}: the vertex of a candidate is origin + direction * t of the pixel's ray,
// the same for every slab, so the distance and the normal are all that has to travel between devices.
__global__ __launch_bounds__(256) void k_composite(float4 *__restrict__ ref_tn, const float4 *__restrict__ tn_parts,
                                                   int n, int npix) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float4 best = make_float4(0.f, KFH_INVALID, 0.f, 0.f);
    bool have = false;
    for (int i = 0; i < n; ++i) {
        const float4 c = tn_parts[(size_t)i * npix + p];
        if (c.x > 0.f && (!have || c.x < best.x)) { best = c; have = true; }
    }
    ref_tn[p] = best;
}

hipError_t launch_composite(float4 *ref_tn, const float4 *tn_parts, int n, int npix, hipStream_t s) {
    hipLaunchKernelGGL(k_composite, dim3((npix + 255) / 256), dim3(256), 0, s, ref_tn, tn_parts, n, npix);
    return hipGetLastError();
}

// The same selection with every slab's candidates read where they lie: part[i] is the packed segment map of slab i,
// in this device's memory or in a peer's (read over the link, 16 bytes per pixel and slab; nothing is copied first).
// Candidates are taken in the order of the array, like the rows of the gathered array above: the same winners.
struct CompositeParts { const float4 *part[KFH_GROUP_MAX]; };
__global__ __launch_bounds__(256) void k_composite_parts(float4 *__restrict__ ref_tn, CompositeParts parts, int n, int npix) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float4 cand[KFH_GROUP_MAX];
#pragma unroll
    for (int i = 0; i < KFH_GROUP_MAX; ++i)                  // all requests leave before the first is looked at
        cand[i] = i < n ? parts.part[i][p] : make_float4(0.f, KFH_INVALID, 0.f, 0.f);
    float4 best = make_float4(0.f, KFH_INVALID, 0.f, 0.f);
    bool have = false;
#pragma unroll
    for (int i = 0; i < KFH_GROUP_MAX; ++i)
        if (i < n && cand[i].x > 0.f && (!have || cand[i].x < best.x)) { best = cand[i]; have = true; }
    ref_tn[p] = best;
}

hipError_t launch_composite_parts(float4 *ref_tn, const float4 *const *parts, int n, int npix, hipStream_t s) {
    if (n < 1 || n > KFH_GROUP_MAX) return hipErrorInvalidValue;
    CompositeParts cp;
    for (int i = 0; i < KFH_GROUP_MAX; ++i) cp.part[i] = i < n ? parts[i] : nullptr;
    hipLaunchKernelGGL(k_composite_parts, dim3((npix + 255) / 256), dim3(256), 0, s, ref_tn, cp, n, npix);
    return hipGetLastError();
}

// The same selection for the pixels [first, first + count) only, the result stored into EVERY member's map: with the
// image split among the members each one reads N candidates for 1/N of the pixels and writes its share N times, so a
// member moves about two maps' worth of bytes over the links however many members there are (k_composite_parts reads
// N - 1 whole maps from its peers).
struct CompositeOuts { float4 *out[KFH_GROUP_MAX]; };
__global__ __launch_bounds__(256) void k_composite_share(CompositeOuts outs, CompositeParts parts, int n, int first,
                                                         int count) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    const int p = first + q;
    float4 cand[KFH_GROUP_MAX];
#pragma unroll
    for (int i = 0; i < KFH_GROUP_MAX; ++i)
        cand[i] = i < n ? parts.part[i][p] : make_float4(0.f, KFH_INVALID, 0.f, 0.f);
    float4 best = make_float4(0.f, KFH_INVALID, 0.f, 0.f);
    bool have = false;
#pragma unroll
    for (int i = 0; i < KFH_GROUP_MAX; ++i)
        if (i < n && cand[i].x > 0.f && (!have || cand[i].x < best.x)) { best = cand[i]; have = true; }
#pragma unroll
    for (int i = 0; i < KFH_GROUP_MAX; ++i)
        if (i < n) outs.out[i][p] = best;
}

hipError_t launch_composite_share(float4 *const *outs, const float4 *const *parts, int n, int first, int count,
                                  hipStream_t s) {
    if (n < 1 || n > KFH_GROUP_MAX || first < 0 || count < 0) return hipErrorInvalidValue;
    if (count == 0) return hipSuccess;
    CompositeParts cp;
    CompositeOuts co;
    for (int i = 0; i < KFH_GROUP_MAX; ++i) {
        cp.part[i] = i < n ? parts[i] : nullptr;
        co.out[i] = i < n ? outs[i] : nullptr;
    }
    hipLaunchKernelGGL(k_composite_share, dim3((count + 255) / 256), dim3(256), 0, s, co, cp, n, first, count);
    return hipGetLastError();
}

// Vertex, normal and hit-distance maps of the last raycast rebuilt from the packed map {t, normal}: the raycast
// stored vertex = origin + direction * t with direction = rotate(view, (x, y, 1)); the same expressions on the same
// matrix give the same bits, so a caller sees exactly what the raycast would have written.
__global__ __launch_bounds__(256) void k_model_maps(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                    float *__restrict__ ray_t, const float4 *__restrict__ ref_tn,
                                                    const Mat4 *__restrict__ view_p, int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const Mat4 view = *view_p;
    const f3 origin = mk3(view.m[3], view.m[7], view.m[11]);
    const f3 direction = rotate(view, mk3((float)x, (float)y, 1.f));
    const int o = x + y * w;
    const float4 tn = ref_tn[o];
    const f3 p = origin + direction * tn.x;
    vertex[o] = tn.x > 0.0f ? make_float4(p.x, p.y, p.z, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    normal[o] = make_float4(tn.y, tn.z, tn.w, 0.f);
    ray_t[o] = tn.x;
}

hipError_t launch_model_maps(float4 *vertex, float4 *normal, float *ray_t, const float4 *ref_tn, const Mat4 *view,
                             int w, int h, hipStream_t s) {
    hipLaunchKernelGGL(k_model_maps, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, vertex, normal, ray_t, ref_tn,
                       view, w, h);
    return hipGetLastError();
}

// Host-to-device hand-over of a depth frame by the compute queue itself: the frame sits in pinned host memory,
// every lane pulls 16 bytes over the bus.  A copy-engine transfer between two kernels of the same stream costs
// two queue hand-overs, which is more than the transfer of a 0.6 MB frame takes.
typedef unsigned int stage_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stage_copy(stage_u32x4 *__restrict__ dst, const stage_u32x4 *__restrict__ src,
                                                    size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = __builtin_nontemporal_load(src + i);
}

hipError_t launch_stage_copy(void *dst, const void *pinned_src, size_t bytes, hipStream_t s) {
    if (bytes % 16 != 0) return hipErrorInvalidValue;
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, static_cast<stage_u32x4 *>(dst),
                       static_cast<const stage_u32x4 *>(pinned_src), n16);
    return hipGetLastError();
}

// ---- GUI images -------------------------------------------------------------------------------------
// gs2rgb of the reference renderer: hue ramp with value 0.75 and saturation 0.6667, evaluated in double
__device__ inline uchar4 hue_ramp(double h) {
    const double v = 0.75, m = 0.25, sv = 0.6667;
    h *= 6.0;
    const int sextant = (int)h;
    const double fract = h - sextant;
    const double vsf = v * sv * fract;
    const double mid1 = m + vsf, mid2 = v - vsf;
    double r = 0, g = 0, b = 0;
    switch (sextant) {
        case 0: r = v; g = mid1; b = m; break;
        case 1: r = mid2; g = v; b = m; break;
        case 2: r = m; g = v; b = mid1; break;
        case 3: r = m; g = mid2; b = v; break;
        case 4: r = mid1; g = m; b = v; break;
        case 5: r = v; g = m; b = mid2; break;
        default: break;                               // h == 1 exactly: black, as in the reference
    }
    return make_uchar4((unsigned char)(r * 255), (unsigned char)(g * 255), (unsigned char)(b * 255), 0);
}

// renderDepthKernel
__global__ __launch_bounds__(256) void k_render_depth(uchar4 *__restrict__ out, const float *__restrict__ depth,
                                                      int npix, float near_plane, float far_plane) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float range_scale = 1 / (far_plane - near_plane);
    const float d = depth[p];
    uchar4 c;
    if (d < near_plane) c = make_uchar4(255, 255, 255, 0);
    else if (d > far_plane) c = make_uchar4(0, 0, 0, 0);
    else c = hue_ramp((d - near_plane) * range_scale);
    out[p] = c;
}

// renderTrackKernel
__global__ __launch_bounds__(256) void k_render_track(uchar4 *__restrict__ out, const kfh_track_data *__restrict__ data,
                                                      const signed char *__restrict__ codes, int stride, int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int result = data ? data[x + (size_t)y * stride].result : (int)codes[x + (size_t)y * stride];
    uchar4 c;
    switch (result) {
        case 1: c = make_uchar4(128, 128, 128, 0); break;    // ok
        case -1: c = make_uchar4(0, 0, 0, 0); break;         // no input
        case -2: c = make_uchar4(255, 0, 0, 0); break;       // not in image
        case -3: c = make_uchar4(0, 255, 0, 0); break;       // no correspondence
        case -4: c = make_uchar4(0, 0, 255, 0); break;       // too far away
        case -5: c = make_uchar4(255, 255, 0, 0); break;     // wrong normal
        default: c = make_uchar4(255, 128, 128, 0); break;
    }
    out[x + (size_t)y * w] = c;
}

// Shading half of renderVolumeKernel.  The raycast launch before it stored the hit point and the normalised
// TSDF gradient, and an invalid normal wherever the ray missed or the gradient vanished.
__global__ __launch_bounds__(256) void k_render_shade(uchar4 *__restrict__ out, const float4 *__restrict__ vertex,
                                                      const float4 *__restrict__ normal, int npix, f3 light,
                                                      f3 ambient) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float4 n4 = normal[p], v4 = vertex[p];
    uchar4 c = make_uchar4(0, 0, 0, 0);
    if (n4.x != KFH_INVALID) {
        const f3 diff = normalize3(light - mk3(v4.x, v4.y, v4.z));
        const float dir = fmaxf(dot3(mk3(n4.x, n4.y, n4.z), diff), 0.f);
        const float r = fminf(fmaxf(dir + ambient.x, 0.f), 1.f) * 255;
        const float g = fminf(fmaxf(dir + ambient.y, 0.f), 1.f) * 255;
        const float b = fminf(fmaxf(dir + ambient.z, 0.f), 1.f) * 255;
        c = make_uchar4((unsigned char)r, (unsigned char)g, (unsigned char)b, 0);
    }
    out[p] = c;
}

__global__ void k_view_from_pose(Mat4 *out, Mat4 pose, float fx, float fy, float cx, float cy) {
    if (threadIdx.x == 0) *out = mat_mul(pose, inverse_camera_matrix(fx, fy, cx, cy));
}

hipError_t launch_render_depth(uchar4 *out, const float *depth, int npix, float near_plane, float far_plane,
                               hipStream_t s) {
    hipLaunchKernelGGL(k_render_depth, dim3((npix + 255) / 256), dim3(256), 0, s, out, depth, npix, near_plane, far_plane);
    return hipGetLastError();
}

hipError_t launch_render_track(uchar4 *out, const kfh_track_data *data, const signed char *codes, int stride, int w,
                               int h, hipStream_t s) {
    if ((data == nullptr) == (codes == nullptr) || stride < w) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_render_track, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, out, data, codes, stride, w, h);
    return hipGetLastError();
}

hipError_t launch_render_shade(uchar4 *out, const float4 *vertex, const float4 *normal, int npix,
                               const float light[3], const float ambient[3], hipStream_t s) {
    hipLaunchKernelGGL(k_render_shade, dim3((npix + 255) / 256), dim3(256), 0, s, out, vertex, normal, npix,
                       mk3(light[0], light[1], light[2]), mk3(ambient[0], ambient[1], ambient[2]));
    return hipGetLastError();
}

__global__ void k_store_mat4(Mat4 *out, Mat4 m, unsigned *zeroed) {
    if (threadIdx.x < 16) out->m[threadIdx.x] = m.m[threadIdx.x];
    if ((threadIdx.x == 16 || threadIdx.x == 17) && zeroed) zeroed[threadIdx.x - 16] = 0u;     // the two list counters
}

hipError_t launch_store_mat4(Mat4 *d_out, const Mat4 &m, unsigned *d_zeroed, hipStream_t s) {
    hipLaunchKernelGGL(k_store_mat4, dim3(1), dim3(64), 0, s, d_out, m, d_zeroed);
    return hipGetLastError();
}

hipError_t launch_view_from_pose(Mat4 *d_view, const Mat4 &pose, const float k[4], hipStream_t s) {
    hipLaunchKernelGGL(k_view_from_pose, dim3(1), dim3(64), 0, s, d_view, pose, k[0], k[1], k[2], k[3]);
    return hipGetLastError();
}

}  // namespace kfh
