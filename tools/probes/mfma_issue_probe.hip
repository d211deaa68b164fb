// Probe: cycles per v_mfma_f32_32x32x16_bf16 for the instruction patterns of the row-chain kernel's blocks.
// One 256-thread workgroup (one wave per SIMD); s_memtime around 512 repetitions of an 8-MFMA block.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF "v_mfma_f32_32x32x16_bf16 "
#define REP 512

template <int V> __global__ __launch_bounds__(256) void probe(long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (threadIdx.x & 63) * 16;
    bf16x8 f[8], n[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) f[i][j] = (__bf16)1.0f, b[i][j] = (__bf16)0.5f, n[i][j] = (__bf16)0.f;
    f32x16 c[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if constexpr (V == 0) {  // dependent chain, accumulator in AGPRs, nothing else
            asm volatile(MF "%0, %1, %9, %0\n" MF "%0, %2, %10, %0\n" MF "%0, %3, %11, %0\n" MF "%0, %4, %12, %0\n"
                         MF "%0, %5, %13, %0\n" MF "%0, %6, %14, %0\n" MF "%0, %7, %15, %0\n" MF "%0, %8, %16, %0\n"
                         : "+a"(c[0]) : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]),
                           "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
        } else if constexpr (V == 1) {  // dependent chain, accumulator in VGPRs
            asm volatile(MF "%0, %1, %9, %0\n" MF "%0, %2, %10, %0\n" MF "%0, %3, %11, %0\n" MF "%0, %4, %12, %0\n"
                         MF "%0, %5, %13, %0\n" MF "%0, %6, %14, %0\n" MF "%0, %7, %15, %0\n" MF "%0, %8, %16, %0\n"
                         : "+v"(c[0]) : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]),
                           "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
        } else if constexpr (V == 2) {  // 8 independent accumulators (AGPRs), one B operand
            asm volatile(MF "%0, %8, %16, %0\n" MF "%1, %9, %16, %1\n" MF "%2, %10, %16, %2\n" MF "%3, %11, %16, %3\n"
                         MF "%4, %12, %16, %4\n" MF "%5, %13, %16, %5\n" MF "%6, %14, %16, %6\n" MF "%7, %15, %16, %7\n"
                         : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7])
                         : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]), "v"(b[0]));
        } else if constexpr (V == 3) {  // V0 + one ds_read_b128 after each MFMA (into other registers), no waits
            asm volatile(MF "%0, %1, %9, %0\n ds_read_b128 %17, %25\n" MF "%0, %2, %10, %0\n ds_read_b128 %18, %25 offset:1024\n"
                         MF "%0, %3, %11, %0\n ds_read_b128 %19, %25 offset:2048\n" MF "%0, %4, %12, %0\n ds_read_b128 %20, %25 offset:3072\n"
                         MF "%0, %5, %13, %0\n ds_read_b128 %21, %25 offset:4096\n" MF "%0, %6, %14, %0\n ds_read_b128 %22, %25 offset:5120\n"
                         MF "%0, %7, %15, %0\n ds_read_b128 %23, %25 offset:6144\n" MF "%0, %8, %16, %0\n ds_read_b128 %24, %25 offset:7168\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+a"(c[0]) : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]),
                           "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]),
                           "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(n[4]), "v"(n[5]), "v"(n[6]), "v"(n[7]), "v"(ra));
        } else if constexpr (V == 4) {  // V0 with an s_barrier per block
            asm volatile("s_barrier\n" MF "%0, %1, %9, %0\n" MF "%0, %2, %10, %0\n" MF "%0, %3, %11, %0\n" MF "%0, %4, %12, %0\n"
                         MF "%0, %5, %13, %0\n" MF "%0, %6, %14, %0\n" MF "%0, %7, %15, %0\n" MF "%0, %8, %16, %0\n"
                         : "+a"(c[0]) : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]),
                           "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
        }
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[V] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + (float)n[i][0];
    sink[threadIdx.x] = s;
}
int main() {
    long long* o; float* sink;
    (void)hipMalloc(&o, 64); (void)hipMalloc(&sink, 1024);
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(256), 0, 0, o, sink);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(256), 0, 0, o, sink);
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(256), 0, 0, o, sink);
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(256), 0, 0, o, sink);
    hipLaunchKernelGGL(probe<4>, dim3(1), dim3(256), 0, 0, o, sink);
    long long h[8];
    (void)hipMemcpy(h, o, 64, hipMemcpyDeviceToHost);
    const char* names[] = {"chain, acc in AGPR", "chain, acc in VGPR", "8 independent accs", "chain + ds_read_b128 per MFMA", "chain + barrier per 8"};
    for (int v = 0; v < 5; ++v) printf("%-32s %7.1f s_memtime ticks per MFMA\n", names[v], (double)h[v] / (REP * 8));
    return 0;
}
