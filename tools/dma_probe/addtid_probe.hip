// (word3: DATA_FORMAT bits 15-18 are the high bits of the stride when ADD_TID_ENABLE is set -- they must be 0: with the default 0x27000 the probe faulted)
// Does a buffer resource with ADD_TID_ENABLE (word3 bit 23, stride 16) make `buffer_load_dwordx4 off, s[rsrc], soffset lds` copy
// lane l's 16 bytes from base + soffset + 16 l -- an LDS-DMA piece that reads NO vector register?  (round 6, profiles/r06_linear_diag.md)
//   hipcc --offload-arch=gfx950 -O3 tools/dma_probe/addtid_probe.hip -o tools/dma_probe/addtid_probe && tools/dma_probe/addtid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned short* __restrict__ g, unsigned short* out, int soff_bytes, int n_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* lds = (unsigned short*)smem;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, (short)16, n_bytes, 0x00007000 | (1 << 23));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + w * 512), 16, 0, soff_bytes + w * 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 512; i += blockDim.x) out[i] = lds[i];
}
int main() {
    const int n = 1 << 16;
    std::vector<unsigned short> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i * 7 + 3);
    unsigned short *g, *o;
    hipMalloc(&g, n * 2); hipMalloc(&o, 4 * 512 * 2);
    hipMemcpy(g, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemset(o, 0xff, 4 * 512 * 2);
    const int soff = 4096;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, g, o, soff, n * 2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    std::vector<unsigned short> r(4 * 512);
    hipMemcpy(r.data(), o, 4 * 512 * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4 * 512; ++i) if (r[i] != h[soff / 2 + i]) { if (bad < 8) printf("  [%d] got %u want %u\n", i, r[i], h[soff / 2 + i]); ++bad; }
    printf("add_tid buffer_load ... lds: %d of %d words wrong\n", bad, 4 * 512);
    return bad != 0;
}
