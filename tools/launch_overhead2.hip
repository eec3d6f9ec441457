// What makes a small dependent kernel expensive?  graph of 100 dependent launches, 256 workgroups each; vary block size, dynamic LDS,
// a cold dependent load, barriers.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(float *p, const float *cold, int stride) {
    extern __shared__ float lds[];
    float v = 1.0f;
    if (MODE & 1) v = cold[(size_t)blockIdx.x * stride + threadIdx.x];       // one cold global load
    if (MODE & 2) { lds[threadIdx.x] = v; __syncthreads(); v += lds[(threadIdx.x + 64) % blockDim.x]; __syncthreads(); }
    if (threadIdx.x < 64) p[blockIdx.x * 64 + threadIdx.x] = v;
}

template <int MODE>
double run(float *p, const float *cold, int threads, size_t lds, hipStream_t st, int blocks = 256) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipGraph_t g; hipGraphExec_t ge;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, st, p, cold, 4096);
    CK(hipStreamSynchronize(st));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, st, p, cold + (size_t)(i % 50) * 4096 * 256, 4096);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 1000;
}

int main() {
    float *p, *cold;
    CK(hipMalloc(&p, 1 << 22));
    CK(hipMalloc(&cold, (size_t)56 * 4096 * 256 * 4));  // 50 offsets of 256 blocks + up to 1024 blocks of 4096 floats each
    CK(hipMemset(cold, 0, (size_t)56 * 4096 * 256 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("store only        256thr lds0    %.2f us\n", run<0>(p, cold, 256, 0, st));
    printf("store only        256thr lds16K  %.2f us\n", run<0>(p, cold, 256, 16 << 10, st));
    printf("store only        256thr lds64K  %.2f us\n", run<0>(p, cold, 256, 64 << 10, st));
    printf("store only        256thr lds128K %.2f us\n", run<0>(p, cold, 256, 128 << 10, st));
    printf("store only       1024thr lds0    %.2f us\n", run<0>(p, cold, 1024, 0, st));
    printf("store only       1024thr lds80K  %.2f us\n", run<0>(p, cold, 1024, 80 << 10, st));
    printf("cold load+store   256thr lds0    %.2f us\n", run<1>(p, cold, 256, 0, st));
    printf("cold load+store   256thr lds16K  %.2f us\n", run<1>(p, cold, 256, 16 << 10, st));
    printf("load+2 barriers   256thr lds16K  %.2f us\n", run<3>(p, cold, 256, 16 << 10, st));
    printf("load+2 barriers  1024thr lds80K  %.2f us\n", run<3>(p, cold, 1024, 80 << 10, st));
    printf("load+2 barriers    64thr lds16K  %.2f us (1024 blocks)\n", run<3>(p, cold, 64, 16 << 10, st, 1024));
    printf("cold load+store   256thr lds0    %.2f us (64 blocks)\n", run<1>(p, cold, 256, 0, st, 64));
    printf("cold load+store   256thr lds0    %.2f us (1024 blocks)\n", run<1>(p, cold, 256, 0, st, 1024));
    return 0;
}
