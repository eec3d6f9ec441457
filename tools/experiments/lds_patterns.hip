// LDS access patterns of attn_bwd1p.hip timed in isolation: one wave, 64 x 32 back-to-back instructions of one pattern, cycles per instruction.
//   hipcc -O3 --offload-arch=gfx950 -o tools/experiments/lds_patterns.bin tools/experiments/lds_patterns.hip (the .bin is git-ignored; gpurun -- tools/experiments/lds_patterns.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 *lds_s4;
__device__ int tl_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int KIND>
__global__ void k(long long *out, int *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    int addr = 0;
    if (KIND == 0) addr = tl_off(lr, 1) + 8 * lh;                       // b64 write, old dS image
    if (KIND == 1) addr = tl_off(lr, 2 + lh);                           // b128 write, TL row pattern
    if (KIND == 2) addr = lane * 16;                                    // b128 write, linear
    if (KIND == 3) addr = (lr * 36 + 4 * lh) * 4 + 32;                  // b128 write, flush (pitch 36 floats)
    if (KIND == 4) addr = ((2 * wave + lh) * 36 + lr) * 4;              // b32 read, old sum rows
    if (KIND == 5) addr = ((wave + 8 * lh) * 36 + lr) * 4;              // b32 read, new sum rows
    if (KIND == 6) addr = tl_off(4 * lh + (i16 >> 2), 2 * g1 + ((i16 & 3) >> 1)) + 8 * (i16 & 1);   // transposing read
    if (KIND == 7) addr = tl_off(lr, 2 + lh);                           // b128 read, row fragments
    if (KIND == 8) addr = (8 + 4 * lh) * 4;                             // b128 read, broadcast of the statistics
    if (KIND == 9) addr = (lr * 40 + 4 * lh) * 4 + 32;                  // b128 write, pitch 40 floats
    if (KIND == 10) addr = lr * 64 + ((2 + lh) ^ (lr & 3)) * 16;        // b128 write, swizzle by row & 3
    if (KIND == 11) addr = lr * 80 + (2 + lh) * 16;                     // b128 write, pitch 80 bytes
    addr += wave * 8192;
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (KIND == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(u32x2{v.x, v.y}) : "memory");
            else if (KIND == 1 || KIND == 2 || KIND == 3 || KIND == 9 || KIND == 10 || KIND == 11) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
            else if (KIND == 4 || KIND == 5) { unsigned r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc.x += r; }
            else if (KIND == 6) { u32x2 r; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc.x += r.x; }
            else { u32x4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc.x += r.x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[wave] = t1 - t0;
    if (acc.x == 0x12345) sink[0] = acc.x;
}
template <int KIND>
void run(const char *name, int waves, long long *d, int *sink) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<KIND><<<1, 64 * waves, 65536>>>(d, sink);
    k<KIND><<<1, 64 * waves, 65536>>>(d, sink);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s waves %d: %.2f cycles / instruction (wave 0)\n", name, waves, (double)h[0] / (64 * 32));
}
int main() {
    long long *d; int *sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 4);
    for (int waves = 1; waves <= 4; waves += 3) {
        run<0>("b64 write, dS image 8-byte pieces", waves, d, sink);
        run<1>("b128 write, TL row pattern", waves, d, sink);
        run<2>("b128 write, linear", waves, d, sink);
        run<3>("b128 write, flush pitch 36", waves, d, sink);
        run<9>("b128 write, pitch 40 floats", waves, d, sink);
        run<10>("b128 write, swizzle row & 3", waves, d, sink);
        run<11>("b128 write, pitch 80 bytes", waves, d, sink);
        run<4>("b32 read, sum rows 2w + lh", waves, d, sink);
        run<5>("b32 read, sum rows w + 8 lh", waves, d, sink);
        run<6>("transposing read", waves, d, sink);
        run<7>("b128 read, row fragments", waves, d, sink);
        run<8>("b128 read, statistics broadcast", waves, d, sink);
    }
    return 0;
}
