// fp32 no-return atomic adds in the one-pass attention backward's pattern, alone: every workgroup of 256 threads walks the [4096 q][H][32] fp32 rows of
// its (sequence, head) in 64-row tiles, lanes along d (a wave instruction = two 128-byte row segments); the 8 (or 16) key-block workgroups of a
// (sequence, head) add to the same rows.  hipcc -O3 --offload-arch=gfx950 -o tools/experiments/atomic_rate.bin tools/experiments/atomic_rate.hip (the .bin is git-ignored; gpurun -- tools/experiments/atomic_rate.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float *buf, int H, int nblk, int rows, int swz) {
    int vid = blockIdx.x;
    if (swz) { const int per = gridDim.x >> 3; vid = (vid & 7) * per + (vid >> 3); }
    const int kb = vid % nblk, h = (vid / nblk) % H, b = vid / (nblk * H);
    (void)kb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int ldq = H * 32;
    float *base = buf + ((size_t)b * rows * H + h) * 32;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, ((rows - 1) * ldq + 32) * 4, 0x00020000);
    const unsigned voff = ((wave + 8 * lh) * ldq + lr) * 4;
    for (int t = 0; t < rows / 64; ++t)
#pragma unroll
        for (int n = 0; n < 8; ++n)
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, r, voff, ((t * 64 + (n >> 2) * 32 + 4 * (n & 1) + 16 * ((n >> 1) & 1)) * ldq) * 4, 0);
}
int main() {
    const int B = 32, H = 16, rows = 4096;
    float *buf;
    const size_t bytes = (size_t)B * rows * H * 32 * 4;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    for (int nblk = 8; nblk <= 16; nblk *= 2)
        for (int swz = 0; swz < 2; ++swz) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<nblk * H * B, 256>>>(buf, H, nblk, rows, swz);
            hipEventRecord(e0);
            for (int i = 0; i < 3; ++i) k<<<nblk * H * B, 256>>>(buf, H, nblk, rows, swz);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            printf("key blocks %2d, XCD swizzle %d: %.3f ms, %.2f TB/s of added floats (%.2f GB)\n", nblk, swz, ms, bytes * nblk / ms / 1e9, bytes * nblk / 1e9);
        }
    float v; hipMemcpy(&v, buf, 4, hipMemcpyDeviceToHost);
    printf("check: element 0 = %.0f (expected %d)\n", v, (8 + 8 + 16 + 16) * 4);
    return 0;
}
