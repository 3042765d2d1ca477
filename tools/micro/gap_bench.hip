// Micro-benchmark: time per dependent kernel for a chain of small kernels, launched (a) one by one on a stream,
// (b) as one captured graph.  Each kernel reads the value its predecessor wrote (a real dependency).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_step(float *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[(i + 1) % n] + 1.0f;
}

int main(int argc, char **argv) {
    const int chain = argc > 1 ? std::atoi(argv[1]) : 25, reps = 200, n = 200 * 576;
    float *d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run_stream = [&]() { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_step, dim3(200), dim3(576), 0, s, d, n); };
    for (int w = 0; w < 20; ++w) run_stream();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run_stream();
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("stream launches: %.3f us per kernel (chain %d)\n", us / (reps * chain), chain);

    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    run_stream();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("graph replay:    %.3f us per kernel\n", us / (reps * chain));
    return 0;
}
