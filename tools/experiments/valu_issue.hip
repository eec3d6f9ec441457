// Issue cost of single VALU instructions for ONE wave on a SIMD (cycles per instruction, 64 x 64 independent instructions back to back):
//   hipcc -O3 --offload-arch=gfx950 -o tools/experiments/valu_issue.bin tools/experiments/valu_issue.hip   (the .bin is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
template <int KIND>
__global__ void k(long long *out, float *sink, float seed) {
    float a[8]; f2 b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = f2{seed + i, seed - i}; }
    const float m = 1.0000001f; const f2 m2 = {1.0000001f, 0.9999999f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int i = u & 7;
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(m2));
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 3) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(a[(i + 1) & 7])); a[i] = __uint_as_float(r); }
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(m2));
            if (KIND == 5) asm volatile("s_nop 0");
            if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
    if (s == 12345.f) sink[0] = s;
}
template <int KIND> void run(const char *n, long long *d, float *sink) {
    k<KIND><<<1, 64>>>(d, sink, 1.f); k<KIND><<<1, 64>>>(d, sink, 1.f);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-22s %.2f cycles / instruction\n", n, (double)h / 4096);
}
int main() {
    long long *d; float *sink; (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 4);
    run<0>("v_mul_f32", d, sink); run<6>("v_fma_f32", d, sink); run<1>("v_pk_mul_f32", d, sink); run<4>("v_pk_fma_f32", d, sink);
    run<2>("v_exp_f32", d, sink); run<3>("v_cvt_pk_bf16_f32", d, sink); run<5>("s_nop 0", d, sink);
    return 0;
}
