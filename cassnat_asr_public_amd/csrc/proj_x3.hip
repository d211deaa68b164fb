// d_model-deep projections of the split-bf16 ("bf16x3") engine, gfx950, K = 256:
//     C[M][N] = A[M][256] . W[N][256]^T + bias            (split-bf16 output: Q / K / V, cross-attention queries and memories)
//     x[M][N] = resid + scale * (A . W^T + bias)           (fp32 output: the attention output projection onto the residual)
// i.e. the nn.Linear layers of MultiHeadedAttention (src/models/modules/attention.py:57-66) around the attention kernel.
// Every product is three MFMAs on (hi, lo) bf16 pairs (common.h split_t), fp32 accumulation - the same arithmetic as the tiled
// GEMM (gemm.hip, Frag<split_t>), which these shapes ran on before: K = 256 is 8 of its slabs, a barrier each, on 64 x 64
// tiles (17 us per output projection of 8000 rows, 44 TFLOP/s).  Here, as in genmax.hip:
//   * a workgroup owns 32 MT rows; their activations go to LDS once, as MFMA fragments (a hi and a lo plane), behind the
//     kernel's only barrier;
//   * every wave owns the output column tiles w, w + 4, ... and streams their pre-tiled 1-KiB weight fragments (hi, lo per
//     k-step; pack_proj_x3) from L2 into four rotating register sets, three groups of two k-steps in flight;
//   * fp32 output: activations are the first MFMA operand, a lane owns one output column: bias, residual (requested when
//     the tile starts) and 128-byte row segments per store;  split output: weights first, a lane owns 16 columns of one
//     row, the half-waves trade groups (v_permlane32_swap) and every store is 16 bytes of hi or lo halves.
#include <cstdlib>
#include <cstring>

#include "proj_x3_phase.h"

struct ProjX3Params {
    const unsigned char* A;  // [M] rows of 256 split-bf16 elements
    long long lda_bytes;
    PxPhase ph;              // weights, bias, output, residual, M, N (proj_x3_phase.h)
};

// (32-row workgroups stay under 128 registers: four of them share a CU, and what a CU reads from L2 is latency x bytes in
// flight - 12 KiB per wave here)
template <int MT, bool SPLIT_OUT>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 2) void proj_x3_kernel(ProjX3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int m0 = blockIdx.x * 32 * MT;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.ph.N >> 5;
    // blockIdx.y: the chunk of column tiles this workgroup computes (tile_lo .. tile_hi - 1, dealt to its waves round robin)
    const int tiles_per_chunk = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int tile_lo = blockIdx.y * tiles_per_chunk, tile_hi = tile_lo + tiles_per_chunk < ntiles ? tile_lo + tiles_per_chunk : ntiles;
    px_stage_rows<MT>(p.A, p.lda_bytes, m0, p.ph.M, smem, tid);  // activations -> LDS, behind the kernel's only barrier
    __syncthreads();
    px_tiles<MT, SPLIT_OUT>(p.ph, smem, smem + 32768, m0, tile_lo, tile_hi, wave_u, lane);
}

template <int MT, bool SPLIT_OUT> static int launch_proj_x3_variant(const ProjX3Params& p, hipStream_t s) {
    constexpr int lds = MT * 32768;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)proj_x3_kernel<MT, SPLIT_OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_once.mark(attr_dev);
    }
    // column chunks (blockIdx.y): the fewest that put a workgroup on every CU, among those that deal every wave the same number
    // of tiles (each chunk stages the row tile again: 32 MT KiB beside 32 KiB per column tile)
    const int row_tiles = cn_ceil_div(p.ph.M, 32 * MT), ntiles = p.ph.N / 32;
    int chunks = 1;
    for (int c = 1; c <= ntiles / 4; ++c) {
        if (ntiles % (4 * c) != 0) continue;
        chunks = c;
        if (row_tiles * c >= 256) break;
    }
    hipLaunchKernelGGL((proj_x3_kernel<MT, SPLIT_OUT>), dim3(row_tiles, chunks), dim3(256), lds, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

bool proj_x3_applies(int N, int K) { return K == 256 && N >= 32 && N % 32 == 0 && N <= 1024; }

int launch_proj_x3(const ProjX3Args& a, hipStream_t s) {
    if (!proj_x3_applies(a.N, 256) || !a.bias || !a.wp || a.lda % 32 != 0 || (!a.c_f32 && a.ldc % 32 != 0) || (a.resid && !a.c_f32)) {
        cn_set_error("proj_x3: needs K == 256, N a multiple of 32 (<= 1024), 32-element row strides, a residual only with fp32 output");
        return -1;
    }
    if (a.M <= 0) return 0;
    if (a.c_f32 && (long long)a.M * (a.ldc > a.ldr ? a.ldc : a.ldr) * 4 >= (1ll << 31)) {
        cn_set_error("proj_x3: fp32 output beyond 2 GiB");
        return -1;
    }
    ProjX3Params p;
    p.A = reinterpret_cast<const unsigned char*>(a.A);
    p.lda_bytes = (long long)a.lda * 4;
    p.ph.wp = reinterpret_cast<const unsigned char*>(a.wp);
    p.ph.bias = a.bias;
    p.ph.C = a.C;
    p.ph.ldc = a.ldc;
    p.ph.resid = a.resid;
    p.ph.ldr = a.ldr;
    p.ph.resid_scale = a.resid_scale;
    p.ph.M = a.M;
    p.ph.N = a.N;
    // 64-row workgroups from 4096 rows on: what bounds these launches is the chip's L2 -> CU traffic (about 9 TB/s for this
    // pattern, whatever the tiling: the 64 x 64 GEMM tiles moved the same bytes in the same time), and a wider row tile halves
    // the weight stream per row; below that the launch needs the workgroups more
    const bool wide = a.M >= 4096;
    if (a.c_f32) return wide ? launch_proj_x3_variant<2, false>(p, s) : launch_proj_x3_variant<1, false>(p, s);
    return wide ? launch_proj_x3_variant<2, true>(p, s) : launch_proj_x3_variant<1, true>(p, s);
}

static inline uint16_t px_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// byte offset of element (n, k)'s hi half in the packed stream (the lo half sits 1024 bytes further):
// frag(tile, ks, plane, lane)[j] = W[32 tile + (lane & 31)][16 ks + 8 (lane >> 5) + j]
size_t proj_x3_off(int n, int k) {
    const int tile = n >> 5, ks = k >> 4, lane = ((k >> 3) & 1) * 32 + (n & 31), j = k & 7;
    return ((((size_t)tile * 16 + ks) * 2) * 64 + lane) * 16 + j * 2;
}

// W [N][256] fp32 -> the packed stream (N * 1024 bytes): hi = bf16(w), lo = bf16(w - hi)
void pack_proj_x3(const float* w, int N, unsigned char* out) {
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < 256; ++k) {
            const float f = w[(size_t)n * 256 + k];
            const uint16_t hi = px_bf16_bits(f);
            const uint32_t hb = (uint32_t)hi << 16;
            float hf;
            memcpy(&hf, &hb, 4);
            const uint16_t lo = px_bf16_bits(f - hf);
            const size_t o = proj_x3_off(n, k);
            memcpy(out + o, &hi, 2);
            memcpy(out + o + 1024, &lo, 2);
        }
}
