// Which bf16 MFMA shape does this chip run faster under load?  Bare loops on random operands held in registers, one wave per SIMD, the same
// 64 x 64 output tile per wave (MI355X_MICROARCH.md 'DVFS give-back' item 7: the clock the chip holds depends on the shape).
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_shape.hip -o tools/experiments/mfma_shape && tools/experiments/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int LDSR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const uint4 *in, float *out, int iters) {
    __shared__ uint4 lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = in[(blockIdx.x * 2048 + i) & 65535];
    __syncthreads();
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 65535]; b[i] = in[(tid * 8 + 4 + i) & 65535]; }
    float acc_out = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {   // two k-steps of 16: 8 MFMAs = 64 x 64 x 32
                if constexpr (LDSR) {
                    for (int i = 0; i < 2; ++i) { a[2 * ks + i] = lds[(tid + 256 * (2 * ks + i) + it) & 2047]; b[2 * ks + i] = lds[(tid + 256 * (4 + 2 * ks + i) + it) & 2047]; }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        c[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * ks + i]), __builtin_bit_cast(bf16x8, b[2 * ks + j]), c[2 * i + j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc_out += c[i][e];
    } else {
        f32x4 c[16] = {};
        for (int it = 0; it < iters; ++it) {   // one k-step of 32: 16 MFMAs = 64 x 64 x 32
            if constexpr (LDSR) {
                for (int i = 0; i < 4; ++i) { a[i] = lds[(tid + 256 * i + it) & 2047]; b[i] = lds[(tid + 256 * (4 + i) + it) & 2047]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c[4 * i + j], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc_out += c[i][e];
    }
    out[blockIdx.x * 256 + tid] = acc_out;
}

int main() {
    const int n = 65536;
    std::vector<uint16_t> h(n * 8);
    srand(1);
    for (auto &x : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; uint32_t u; memcpy(&u, &f, 4); x = u >> 16; }
    uint4 *in; float *out;
    hipMalloc(&in, n * 16); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 4;
    auto run = [&](auto kern, const char *name) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, iters);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 4.0 * grid * 4 * (double)iters * 64 * 64 * 32 * 2;
            printf("%-34s %8.2f ms  %7.1f TF/s\n", name, ms / 4, flops / (ms * 1e-3) / 1e12);
        }
    };
    run(k<32, 0>, "32x32x16 operands in registers");
    run(k<16, 0>, "16x16x32 operands in registers");
    run(k<32, 1>, "32x32x16 operands from LDS");
    run(k<16, 1>, "16x16x32 operands from LDS");
    run(k<32, 0>, "32x32x16 operands in registers");
    return 0;
}
