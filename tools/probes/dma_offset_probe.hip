// Probe: does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global
// source?  (It decides whether four 1-KiB pieces can share one M0.)   hipcc --offload-arch=gfx950 -O2 -o probe dma_offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* g, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];  // 4 KiB
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned voff = threadIdx.x * 16;
    const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_waitcnt vmcnt(0)"
                 :: "s"(m0v), "v"(voff), "s"(g) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *g, *o;
    hipMalloc(&g, 4096 * 4);
    hipMalloc(&o, 1024 * 4);
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, o);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    printf("lds[0]=%u lds[1]=%u lds[255]=%u | lds[256]=%u lds[257]=%u lds[511]=%u\n", r[0], r[1], r[255], r[256], r[257], r[511]);
    printf("%s\n", r[256] == 256 ? "offset applies to BOTH global and LDS address"
                   : (r[0] == 256 ? "offset applies to the GLOBAL address only" : "unexpected"));
    return 0;
}
