// Probe: wall-clock FLOP/s of v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 on RANDOM operands (the clock the chip
// holds under load can depend on the MFMA shape: MI355X_MICROARCH.md, DVFS give-back item 7).  One wave per SIMD, 256 workgroups
// of 4 waves, operands in registers; equal FLOPs per iteration (one 32x32x16 = two 16x16x32 ... x2: see below).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe tools/probes/mfma_shape_probe.hip && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HC(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: 4 independent 32x32 accumulators, 32x32x16 (32768 MAC-pairs each).  MODE 1: 16 independent 16x16 accumulators, 16x16x32
// (8192 each): 4 x 32768 = 16 x 8192 products per iteration in both.
template <int MODE>
__global__ __launch_bounds__(256) void shape_kernel(const unsigned* seed, float* sink, int iters) {
    bf16x8 a[4], b[4];
    unsigned s = seed[threadIdx.x + 256 * (blockIdx.x & 15)];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            a[i][j] = (__bf16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
            s = s * 1664525u + 1013904223u;
            b[i][j] = (__bf16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
        }
    float r = 0.f;
    if (MODE == 0) {
        f32x16 c[4] = {};
        for (int it = 0; it < iters; ++it)  // (static operand indices: a dynamic one would put the operands in scratch)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + 1) & 3], c[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][15];
    } else {
        f32x4 c[16] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) r += c[i][0] + c[i][3];
    }
    if (r == 12345.678f) sink[0] = r;
}

int main() {
    unsigned* seed; float* sink;
    HC(hipMalloc(&seed, 4096 * 4)); HC(hipMalloc(&sink, 64));
    unsigned h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 12345u + 7919u * i;
    HC(hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const int iters = 100000;  // 4 x 32 cycles per iteration -> ~6 ms per launch
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            for (int w = 0; w < 40; ++w) {  // ~0.5 s of back-to-back launches before the timed ones
                if (mode == 0) hipLaunchKernelGGL(shape_kernel<0>, dim3(256), dim3(256), 0, 0, seed, sink, iters);
                else hipLaunchKernelGGL(shape_kernel<1>, dim3(256), dim3(256), 0, 0, seed, sink, iters);
            }
            HC(hipEventRecord(e0, 0));
            for (int w = 0; w < 20; ++w) {
                if (mode == 0) hipLaunchKernelGGL(shape_kernel<0>, dim3(256), dim3(256), 0, 0, seed, sink, iters);
                else hipLaunchKernelGGL(shape_kernel<1>, dim3(256), dim3(256), 0, 0, seed, sink, iters);
            }
            HC(hipEventRecord(e1, 0)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 20.0 * 256 * 4 * (double)iters * 4 * 32768 * 2;
            fflush(stdout); printf("%s: %.2f ms per launch, %.0f TFLOP/s\n", mode == 0 ? "32x32x16" : "16x16x32", ms / 20, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
