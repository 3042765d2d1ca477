// Memory-side ceiling of the dense integration's traversal: every voxel of a V^3 short2 volume is read, changed and
// written in place by lanes that each march one (or VEC x-adjacent) column(s) along z, U planes in flight, with
// ALU dummy multiply-adds per voxel standing in for the projection arithmetic.  The figure to compare with is the
// linear read-modify-write sweep of the same bytes (mode "linear").
//
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rmw_skeleton.hip -o tools/micro/rmw_skeleton && tools/micro/rmw_skeleton [V]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VEC> struct Vox;
template <> struct Vox<1> { typedef unsigned T; };
template <> struct Vox<2> { typedef uint2 T; };
template <> struct Vox<4> { typedef uint4 T; };

__device__ inline unsigned bump(unsigned v, float f) { return v + 0x10000u + (f == 12345.f ? 1u : 0u); }
__device__ inline uint2 bump(uint2 v, float f) { return make_uint2(bump(v.x, f), bump(v.y, f)); }
__device__ inline uint4 bump(uint4 v, float f) { return make_uint4(bump(v.x, f), bump(v.y, f), bump(v.z, f), bump(v.w, f)); }

// block = 4 wavefronts arranged WX along x and 4 / WX along y; a lane owns VEC x-adjacent columns
template <int VEC, int WX, int U, int ALU>
__global__ __launch_bounds__(256) void k_columns(unsigned *vol, int V, int zrun) {
    typedef typename Vox<VEC>::T T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wx = wave % WX, wy = wave / WX;
    const int tile_w = 64 * VEC * WX, tile_h = 4 / WX;
    const int tiles_x = V / tile_w, tiles_y = V / tile_h;
    const int runs = V / zrun;
    for (int item = blockIdx.x; item < tiles_x * tiles_y * runs; item += gridDim.x) {
        const int tx = item % tiles_x, ty = (item / tiles_x) % tiles_y, r = item / (tiles_x * tiles_y);
        const int x = tx * tile_w + wx * 64 * VEC + lane * VEC, y = ty * tile_h + wy;
        const size_t plane = (size_t)V * V;
        unsigned *p = vol + (size_t)r * zrun * plane + (size_t)y * V + x;
        float f = (float)x * 1e-3f, g = (float)y * 1e-3f;
        for (int z = 0; z < zrun; z += U) {
            T v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const T *>(p + (size_t)u * plane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < ALU; ++k) f = __builtin_fmaf(f, 1.0000001f, g);
                *reinterpret_cast<T *>(p + (size_t)u * plane) = bump(v[u], f);
            }
            p += (size_t)U * plane;
        }
    }
}

// Every block owns a column tile for the whole launch and takes its planes in SEG segments whose order is reversed
// when `parity` is odd: a launch then starts with the planes the launch before it touched last, which is what a
// memory-side cache of half the volume's size still holds.
template <int VEC, int U, int SEG>
__global__ __launch_bounds__(256) void k_segments(unsigned *vol, int V, int parity) {
    typedef typename Vox<VEC>::T T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile_w = 64 * VEC, tiles_x = V / tile_w;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * tile_w + lane * VEC, y = ty * 4 + wave;
    const size_t plane = (size_t)V * V;
    const int zseg = V / SEG;
    for (int s = 0; s < SEG; ++s) {
        const int seg = (parity & 1) ? SEG - 1 - s : s;
        unsigned *p = vol + (size_t)seg * zseg * plane + (size_t)y * V + x;
        for (int z = 0; z < zseg; z += U) {
            T v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const T *>(p + (size_t)u * plane);
#pragma unroll
            for (int u = 0; u < U; ++u) *reinterpret_cast<T *>(p + (size_t)u * plane) = bump(v[u], 0.f);
            p += (size_t)U * plane;
        }
    }
}

// The mapping of the integrate kernel's two-column march: a 64 x 4 tile per block, a lane holds two x-adjacent columns,
// a wavefront two rows of 32 lanes (two 256-byte pieces per instruction); wavefronts 0-1 take the first half of the
// tile's planes, wavefronts 2-3 the second.  GATHER adds two dependent-free 4-byte loads per lane and plane from a small
// table (the depth pixels).
template <int U, int GATHER>
__global__ __launch_bounds__(256) void k_box64x4(unsigned *vol, int V, const float *table, int parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = V / 64;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 64 + (lane & 31) * 2, y = ty * 4 + (wave & 1) * 2 + (lane >> 5);
    const size_t plane = (size_t)V * V;
    const int half = (wave >> 1) ^ (parity & 1);
    unsigned *p = vol + (size_t)half * (V / 2) * plane + (size_t)y * V + x;
    float acc = 0.f;
    for (int z = 0; z < V / 2; z += U) {
        uint2 v[U];
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const uint2 *>(p + (size_t)u * plane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            g[u] = 0.f;
            if (GATHER) g[u] = table[(y * 640 + x + (z + u)) & 0x3FFFF] + table[(y * 640 + x + 1 + (z + u)) & 0x3FFFF];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc += g[u];
            *reinterpret_cast<uint2 *>(p + (size_t)u * plane) = bump(v[u], acc);
        }
        p += (size_t)U * plane;
    }
}

// The integrate kernel's rolling window on 128 x 2 tiles: wavefront w marches row (w & 1) of the tile over half
// (w >> 1) of the planes, two columns per lane (512 contiguous bytes per instruction); while plane z is stored, the
// loads of planes z + 1 .. z + 8 are in flight.  ALU dependent multiply-adds per plane stand in for the arithmetic.
template <int ALU, int GATHER>
__global__ __launch_bounds__(256) void k_roll(unsigned *vol, int V, const float *table, int parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = V / 128;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 128 + lane * 2, y = ty * 2 + (wave & 1);
    const size_t plane = (size_t)V * V;
    const int half = (wave >> 1) ^ (parity & 1);
    unsigned *p = vol + (size_t)half * (V / 2) * plane + (size_t)y * V + x;
    const int n = V / 2;
    uint2 v[8];
    float g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        v[u] = *reinterpret_cast<const uint2 *>(p + (size_t)u * plane);
        g[u] = GATHER ? table[(y * 640 + x + u) & 0x3FFFF] + table[(y * 640 + x + 1 + u) & 0x3FFFF] : 0.f;
    }
    float f = (float)x * 1e-3f;
    for (int z = 0; z < n; z += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < ALU; ++k) f = __builtin_fmaf(f, 1.0000001f, g[u]);
            *reinterpret_cast<uint2 *>(p + (size_t)(z + u) * plane) = bump(v[u], f);
            const int zn = min(z + 8 + u, n - 1);
            v[u] = *reinterpret_cast<const uint2 *>(p + (size_t)zn * plane);
            if (GATHER) g[u] = table[(y * 640 + x + zn) & 0x3FFFF] + table[(y * 640 + x + 1 + zn) & 0x3FFFF];
        }
    }
}

// Two register sets taking turns, memory operations in bursts: per batch of 8 planes the (stand-in) depth gathers of
// batch b, then the 8 stores of batch b - 1 back to back, then the 8 loads of batch b + 1 back to back into the
// registers the stores have just read, then the arithmetic of batch b.  Mapping as k_roll (HALVES = 1) or one
// wavefront per row marching all planes (HALVES = 0, 128-thread blocks).
template <int ALU, int GATHER, int HALVES>
__global__ __launch_bounds__(256) void k_pipe(unsigned *vol, int V, const float *table, int parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = V / 128;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 128 + lane * 2, y = ty * 2 + (wave & 1);
    const size_t plane = (size_t)V * V;
    const int half = HALVES ? (wave >> 1) ^ (parity & 1) : 0;
    const int n = HALVES ? V / 2 : V;
    unsigned *p = vol + (size_t)half * (V / 2) * plane + (size_t)y * V + x;
    uint2 s0[8], s1[8];
    float f = (float)x * 1e-3f;
    auto loads = [&](uint2 (&s)[8], int z) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = *reinterpret_cast<const uint2 *>(p + (size_t)min(z + u, n - 1) * plane);
    };
    auto stores = [&](uint2 (&s)[8], int z) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<uint2 *>(p + (size_t)(z + u) * plane) = s[u];
    };
    auto work = [&](uint2 (&s)[8], int z) {
        float g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = GATHER ? table[(y * 640 + x + z + u) & 0x3FFFF] + table[(y * 640 + x + 1 + z + u) & 0x3FFFF] : 0.f;
        return g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7];
    };
    auto alu = [&](uint2 (&s)[8], float g) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < ALU; ++k) f = __builtin_fmaf(f, 1.0000001f, g);
            s[u] = bump(s[u], f);
        }
    };
    loads(s0, 0);
    loads(s1, 8);
    {
        const float g = work(s0, 0);
        alu(s0, g);
    }
    for (int z = 8; z < n; z += 16) {
        // batch at z in s1; s0 holds the finished batch z - 8
        float g = work(s1, z);
        __builtin_amdgcn_sched_barrier(0);
        stores(s0, z - 8);
        loads(s0, z + 8);
        __builtin_amdgcn_sched_barrier(0);
        alu(s1, g);
        if (z + 8 >= n) { stores(s1, z); break; }
        g = work(s0, z + 8);
        __builtin_amdgcn_sched_barrier(0);
        stores(s1, z);
        loads(s1, z + 16);
        __builtin_amdgcn_sched_barrier(0);
        alu(s0, g);
        if (z + 16 >= n) { stores(s0, z + 8); break; }
    }
}

// Plain loop (U loads, U stores) on tiles of 128 x ROWS columns, two columns per lane, one row per wavefront, blocks of
// ROWS wavefronts (HALVES = 0) or of 2 * ROWS wavefronts of which the second set marches the upper half of the planes.
template <int ROWS, int HALVES, int U>
__global__ void k_tile128(unsigned *vol, int V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = V / 128;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 128 + lane * 2, y = ty * ROWS + (wave % ROWS);
    const size_t plane = (size_t)V * V;
    const int half = HALVES ? wave / ROWS : 0;
    const int n = HALVES ? V / 2 : V;
    unsigned *p = vol + (size_t)half * (V / 2) * plane + (size_t)y * V + x;
    for (int z = 0; z < n; z += U) {
        uint2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const uint2 *>(p + (size_t)(z + u) * plane);
#pragma unroll
        for (int u = 0; u < U; ++u) *reinterpret_cast<uint2 *>(p + (size_t)(z + u) * plane) = bump(v[u], 0.f);
    }
}

__global__ __launch_bounds__(256) void k_linear(uint4 *vol, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = vol[i];
        vol[i] = bump(v, 0.f);
    }
}

template <typename F> static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int VEC, int WX, int U, int ALU> static void run(unsigned *vol, int V, int zrun, int blocks, const char *name) {
    const float us = time_us([&] { hipLaunchKernelGGL((k_columns<VEC, WX, U, ALU>), dim3(blocks), dim3(256), 0, 0, vol, V, zrun); }, 15);
    printf("%-34s zrun %4d blocks %5d : %8.1f us  %6.2f TB/s  frac %.3f\n", name, zrun, blocks, us,
           8.0 * V * V * (double)V / us * 1e-6, 8.0 * V * V * (double)V / us * 1e-6 / 8.0);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 512;
    unsigned *vol;
    const size_t n = (size_t)V * V * V;
    CK(hipMalloc(&vol, n * 4));
    CK(hipMemset(vol, 0, n * 4));
    for (int blocks : {2048, 8192, 65536}) {
        const float us = time_us([&] { hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, (uint4 *)vol, n / 4); }, 15);
        printf("linear dwordx4, %5d blocks                         : %8.1f us  %6.2f TB/s  frac %.3f\n", blocks, us,
               8.0 * n / us * 1e-6, 8.0 * n / us * 1e-6 / 8.0);
    }
    for (int alt = 0; alt < 2; ++alt) {
        // alt = 0: the same plane order in every launch; alt = 1: the order reversed from launch to launch
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("segments VEC2 U8 SEG2, %s:", alt ? "order alternating" : "order fixed");
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_segments<2, 8, 2>), dim3((V / 128) * (V / 4)), dim3(256), 0, 0, vol, V, alt ? i : 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %.1f", ms * 1e3f);
        }
        printf(" us\n");
        printf("segments VEC2 U8 SEG4, %s:", alt ? "order alternating" : "order fixed");
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_segments<2, 8, 4>), dim3((V / 128) * (V / 4)), dim3(256), 0, 0, vol, V, alt ? i : 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %.1f", ms * 1e3f);
        }
        printf(" us\n");
    }
    {
        float *table;
        CK(hipMalloc(&table, 0x40000 * 4));
        CK(hipMemset(table, 0, 0x40000 * 4));
        const int blocks = (V / 64) * (V / 4);
        float us = time_us([&] { hipLaunchKernelGGL((k_box64x4<8, 0>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("box 64x4, 2 columns per lane, U8           : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_box64x4<8, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("box 64x4, 2 columns per lane, U8, gathers  : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_box64x4<4, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("box 64x4, 2 columns per lane, U4, gathers  : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_box64x4<16, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("box 64x4, 2 columns per lane, U16, gathers : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        CK(hipFree(table));
    }
    {
        float *table;
        CK(hipMalloc(&table, 0x40000 * 4));
        CK(hipMemset(table, 0, 0x40000 * 4));
        const int blocks = (V / 128) * (V / 2);
        float us = time_us([&] { hipLaunchKernelGGL((k_roll<0, 0>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("rolling window 128x2 halves                : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_roll<0, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("rolling window 128x2 halves, gathers       : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_roll<32, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("rolling window 128x2 halves, gathers, ALU32: %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_roll<64, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("rolling window 128x2 halves, gathers, ALU64: %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<0, 0, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("bursts, two sets, 128x2 halves              : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<32, 1, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("bursts, two sets, 128x2 halves, gathers ALU32: %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<64, 1, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, 0); }, 15);
        printf("bursts, two sets, 128x2 halves, gathers ALU64: %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<0, 0, 0>), dim3(blocks), dim3(128), 0, 0, vol, V, table, 0); }, 15);
        printf("bursts, two sets, 128x2 whole columns       : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<32, 1, 0>), dim3(blocks), dim3(128), 0, 0, vol, V, table, 0); }, 15);
        printf("bursts, two sets, whole columns, gathers ALU32: %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<64, 1, 0>), dim3(blocks), dim3(128), 0, 0, vol, V, table, 0); }, 15);
        printf("bursts, two sets, whole columns, gathers ALU64: %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        int par = 0;
        us = time_us([&] { hipLaunchKernelGGL((k_pipe<32, 1, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, par++); }, 15);
        printf("bursts, halves swapped every launch, ALU32   : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        par = 0;
        us = time_us([&] { hipLaunchKernelGGL((k_roll<32, 1>), dim3(blocks), dim3(256), 0, 0, vol, V, table, par++); }, 15);
        printf("same, ALU32, halves swapped every launch   : %8.1f us  frac %.3f\n", us, 8.0 * n / us * 1e-6 / 8.0);
        CK(hipFree(table));
    }
    {
        float us;
#define TILE(ROWS, HALVES, U) \
        us = time_us([&] { hipLaunchKernelGGL((k_tile128<ROWS, HALVES, U>), dim3((V / 128) * (V / ROWS)), dim3(64 * ROWS * (HALVES ? 2 : 1)), 0, 0, vol, V); }, 15); \
        printf("tile 128 x %d, %s, U%d, plain loop : %8.1f us  frac %.3f\n", ROWS, HALVES ? "halves in one block" : "whole columns", U, us, 8.0 * n / us * 1e-6 / 8.0);
        TILE(1, 0, 8) TILE(2, 0, 8) TILE(4, 0, 8) TILE(8, 0, 8) TILE(1, 1, 8) TILE(2, 1, 8) TILE(4, 1, 8) TILE(2, 0, 4) TILE(2, 1, 4) TILE(4, 0, 4)
    }
#define RUN(VEC, WX, U, ALU, zrun, blocks) run<VEC, WX, U, ALU>(vol, V, zrun, blocks, "VEC" #VEC " WX" #WX " U" #U " ALU" #ALU)
    // the shape of the integrate kernel as it stands: 64 x 4 tile, a dword per lane, 8 planes in flight, whole columns
    RUN(1, 1, 8, 0, V, 2048);
    RUN(1, 1, 8, 0, V, 1024);
    RUN(1, 1, 16, 0, V, 1024);
    RUN(1, 1, 4, 0, V, 1024);
    RUN(1, 1, 8, 0, 128, 2048);
    RUN(1, 1, 8, 0, 128, 4096);
    RUN(1, 1, 8, 0, 32, 4096);
    // a row of 256 voxels per block (1 KiB contiguous per plane)
    RUN(1, 4, 8, 0, V, 1024);
    RUN(1, 4, 8, 0, 128, 4096);
    RUN(1, 2, 8, 0, V, 1024);
    // two and four columns per lane
    RUN(2, 1, 8, 0, V, 512);
    RUN(2, 1, 8, 0, 128, 2048);
    RUN(2, 1, 4, 0, 128, 2048);
    RUN(2, 2, 8, 0, 128, 2048);
    RUN(2, 4, 8, 0, 128, 2048);
    RUN(2, 4, 4, 0, 128, 2048);
    RUN(4, 1, 4, 0, 128, 1024);
    RUN(4, 1, 8, 0, 128, 1024);
    RUN(4, 2, 4, 0, 128, 1024);
    // how much arithmetic per voxel the traversal hides
    RUN(1, 1, 8, 16, V, 1024);
    RUN(1, 1, 8, 32, V, 1024);
    RUN(1, 1, 8, 48, V, 1024);
    RUN(1, 1, 8, 64, V, 1024);
    RUN(2, 1, 8, 16, 128, 2048);
    RUN(2, 1, 8, 32, 128, 2048);
    RUN(2, 1, 8, 48, 128, 2048);
    CK(hipFree(vol));
    return 0;
}
