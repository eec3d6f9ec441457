// Does a buffer LDS-DMA (buffer_load_dwordx4 ... offen lds) write ZEROS into LDS for lanes whose offset lies beyond the resource's
// num_records?  (Wanted: ragged last K-tile of the weight-gradient GEMMs without a remainder launch.)  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t *src, int valid_bytes, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, valid_bytes, 0x00020000);
    const uint32_t voff = threadIdx.x * 16;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t dst = (uint32_t)(uintptr_t)(lds_ptr)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(dst), "v"(voff), "s"(rs) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t *src, *out;
    hipMalloc(&src, 4096);
    hipMalloc(&out, 1024);
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    const int valid = 40 * 16 + 8;   // lanes 0..39 fully inside, lane 40 half inside, lanes 41..63 outside
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, valid, out);
    std::vector<uint32_t> o(256);
    hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    for (int lane : {0, 39, 40, 41, 63}) printf("lane %2d: %08x %08x %08x %08x\n", lane, o[lane * 4], o[lane * 4 + 1], o[lane * 4 + 2], o[lane * 4 + 3]);
    return 0;
}
