// Micro-benchmark: what a grid-wide exchange costs inside one persistent kernel, as a tracking loop would need it.
// Every block writes a 128-byte row, arrives at a counter (release, agent scope), waits until all blocks of the
// round have arrived (bounded spin), then reads all rows (after an acquire) and sums them.  Printed: time per
// round, for a given grid.  All blocks must be resident together: grid <= compute units.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(576) void k_rounds(float *rows, unsigned *counter, int rounds, float *out, int *timeout_flag) {
    __shared__ float sums[32];
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    const int nb = gridDim.x;
    float carry = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float *my = rows + ((size_t)(r & 1) * nb + blockIdx.x) * 32;
        if (threadIdx.x < 32) my[threadIdx.x] = (float)(blockIdx.x + r) + carry * 1e-9f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)nb * (unsigned)(r + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) { *timeout_flag = 1; gave_up = 1; break; }      // never hang the device
            }
        }
        __syncthreads();
        if (gave_up) break;                                  // block-uniform: every thread reads the same flag
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        {
            // all rows requested at once: eight threads per row, 72 rows per pass (as the tracking kernel does)
            __shared__ float4 slots[72 * 8];
            const int slot = threadIdx.x >> 3, c4 = threadIdx.x & 7;
            const float4 *base = reinterpret_cast<const float4 *>(rows + (size_t)(r & 1) * nb * 32);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = slot; b < nb; b += 72) {
                const float4 v = base[b * 8 + c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            slots[slot * 8 + c4] = acc;
            __syncthreads();
            if (threadIdx.x < 32) {
                float s = 0.f;
                for (int k = 0; k < 72; ++k) s += reinterpret_cast<const float *>(slots)[k * 32 + threadIdx.x];
                sums[threadIdx.x] = s;
            }
        }
        __syncthreads();
        carry = sums[0];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = carry;
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? std::atoi(argv[1]) : 200, rounds = argc > 2 ? std::atoi(argv[2]) : 1000;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (grid > cus) { std::printf("grid %d exceeds the %d compute units\n", grid, cus); return 1; }
    float *rows, *out; unsigned *counter; int *flag;
    CK(hipMalloc(&rows, (size_t)2 * grid * 32 * sizeof(float)));
    CK(hipMalloc(&out, sizeof(float)));
    CK(hipMalloc(&counter, sizeof(unsigned)));
    CK(hipMalloc(&flag, sizeof(int)));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, sizeof(unsigned)));
        CK(hipMemset(flag, 0, sizeof(int)));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_rounds, dim3(grid), dim3(576), 0, 0, rows, counter, rounds, out, flag);
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        float h = 0.f; int hf = 0;
        CK(hipMemcpy(&h, out, sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hf, flag, sizeof(int), hipMemcpyDeviceToHost));
        // expected last sum: sum over b of (b + rounds - 1) (+ a tiny carry term)
        const double want = (double)grid * (grid - 1) / 2 + (double)grid * (rounds - 1);
        std::printf("grid %d: %.3f us per round (%d rounds), timeout %d, sum %.1f (expected about %.1f)\n", grid,
                    us / rounds, rounds, hf, h, want);
    }
    return 0;
}
