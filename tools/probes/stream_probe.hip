// Probe: L2 -> LDS weight-stream rate per CU for the row-chain kernel's access pattern and variations of it.
// 63 workgroups x 256 threads each stream the same 2.5 MiB buffer (160 units of 16 KiB) through an 8-slot LDS ring.
//   V0: chain protocol (each wave a contiguous 4-KiB quarter of every unit, vmcnt(20) + s_barrier per unit)
//   V1: as V0 without the barrier            V2: pieces interleaved across waves (piece 4 j + wave)
//   V3: V0 with vmcnt(28) (8 units primed)   V4: V0, every workgroup starting at a different unit (rotation)
//   hipcc --offload-arch=gfx950 -O2 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define UNITS 160
template <int V> __global__ __launch_bounds__(256) void probe(const uint4* w, long long* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int rot = V == 4 ? (blockIdx.x * 20) % UNITS : 0;
    const int pre = V == 3 ? 8 : 7;
    auto issue = [&](int u) {
        int su = u + rot;
        if (su >= UNITS) su -= UNITS;
        if (su >= UNITS) su -= UNITS;
        const uint4* sb = w + (long long)su * 1024;
        for (int j = 0; j < 4; ++j) {
            const int piece = V == 2 ? 4 * j + wave : 4 * wave + j;
            const unsigned voff = piece * 1024 + lane * 16;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (u & 7) * 16384 + piece * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sb) : "memory");
        }
    };
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int u = 0; u < pre; ++u) issue(u);
    for (int u = 0; u < UNITS; ++u) {
        if (V == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        if (V != 1) asm volatile("s_barrier" ::: "memory");
        issue(u + pre < UNITS ? u + pre : UNITS - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[V] = t1 - t0;
}
int main() {
    uint4* w; long long* o;
    (void)hipMalloc(&w, (size_t)UNITS * 16384 + (1 << 20));
    (void)hipMemset(w, 1, (size_t)UNITS * 16384 + (1 << 20));
    (void)hipMalloc(&o, 64);
    const int lds = 8 * 16384;
#define RUN(V)                                                                                         \
    (void)hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<V>, dim3(63), dim3(256), lds, 0, w, o);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    long long h[8];
    (void)hipMemcpy(h, o, 64, hipMemcpyDeviceToHost);
    const char* names[] = {"chain protocol", "no barrier", "pieces interleaved across waves", "8 units primed", "per-workgroup rotation"};
    for (int v = 0; v < 5; ++v)
        printf("%-34s %8lld ticks  = %6.1f ticks/unit  (%.1f B/tick/CU)\n", names[v], h[v], (double)h[v] / UNITS, 16384.0 * UNITS / h[v]);
    return 0;
}
