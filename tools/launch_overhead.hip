// Micro-benchmark: cost of a dependent kernel boundary (same stream), eager vs hipGraph, tiny vs 256-workgroup kernels.
// Build:  hipcc -O3 --offload-arch=gfx950 tools/launch_overhead.hip -o /tmp/launch_overhead   (links the system ROCm runtime)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void touch(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
    int n = 1000;
    float *p;
    CK(hipMalloc(&p, 1 << 20));
    CK(hipMemset(p, 0, 1 << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int variant = 0; variant < 2; ++variant) {
        int blocks = variant == 0 ? 1 : 256;
        // eager
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, p, blocks * 256);
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, p, blocks * 256);
        CK(hipStreamSynchronize(st));
        double eager = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n;
        // graph of 100 kernels, replayed 10x
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, p, blocks * 256);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double graph = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 1000;
        printf("blocks=%d  eager %.2f us/kernel   graph %.2f us/kernel\n", blocks, eager, graph);
    }
    return 0;
}
