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

#include "kernels.h"

struct ProjX3Params {
    const unsigned char* A;  // [M] rows of 256 split-bf16 elements
    long long lda_bytes;
    const unsigned char* wp; // [N/32][16 k-steps][hi, lo][64 lanes][16 B]
    const float* bias;       // [N]
    void* C;                 // fp32 [M][ldc] or split-bf16 rows of ldc elements
    int ldc;
    const float* resid;      // fp32 [M][ldr] or null (may alias C)
    int ldr;
    float resid_scale;
    int M, N;
};

// (32-row workgroups stay under 128 registers: four of them share a CU, and what a CU reads from L2 is latency x bytes in
// flight - 12 KiB per wave here)
template <int MT, bool SPLIT_OUT>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 2) void proj_x3_kernel(ProjX3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntiles = p.N >> 5;
    // blockIdx.y: the chunk of column tiles this workgroup computes (tile_lo .. tile_hi - 1, dealt to its waves round robin)
    const int tiles_per_chunk = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int tile_lo = blockIdx.y * tiles_per_chunk, tile_hi = tile_lo + tiles_per_chunk < ntiles ? tile_lo + tiles_per_chunk : ntiles;
    unsigned char* xs = smem;  // [MT][16 k-steps][hi, lo][64 lanes][16 B]

    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.wp), 0, ntiles * 32768, 0x00020000);
    const int lane_off = lane * 16;
#define PX_WFRAG(tile, ks, pl) \
    __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off, (((tile) * 16 + (ks)) * 2 + (pl)) * 1024, 0))
    // group g (0..7) of a column tile = k-steps 2g, 2g + 1 (a: hi, b: lo of the first; c, d of the second), in set g & 3
    bf16x8 w0a, w0b, w0c, w0d, w1a, w1b, w1c, w1d, w2a, w2b, w2c, w2d, w3a, w3b, w3c, w3d;
#define PX_LDW(S_, tile, g)                                                                            \
    w##S_##a = PX_WFRAG(tile, 2 * (g), 0); w##S_##b = PX_WFRAG(tile, 2 * (g), 1);                      \
    w##S_##c = PX_WFRAG(tile, 2 * (g) + 1, 0); w##S_##d = PX_WFRAG(tile, 2 * (g) + 1, 1);
    const int t_first = tile_lo + wave_u < tile_hi ? tile_lo + wave_u : tile_lo;  // (a wave without a tile still takes part in the staging below)
    PX_LDW(0, t_first, 0) PX_LDW(1, t_first, 1) PX_LDW(2, t_first, 2)

    // activations -> LDS.  A split-bf16 row is 8 groups of 32 elements, 64 B of hi halves then 64 B of lo halves: 16-byte
    // chunk ch = 8 q + 4 plane + sub holds k = 32 q + 8 sub .. + 7, i.e. k-step 2 q + (sub >> 1), lane half sub & 1
    // (all requests first, then the LDS writes: written as one loop, every chunk waits for its own round trip - 8 MT in series)
    {
        uint4 stage[BM / 4];
#pragma unroll
        for (int i = 0; i < BM / 4; ++i) {
            const int c = tid + 256 * i, r = c >> 6, ch = c & 63;
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            stage[i] = ld16(p.A + (long long)m * p.lda_bytes + 16 * ch);
        }
#pragma unroll
        for (int i = 0; i < BM / 4; ++i) {
            const int c = tid + 256 * i, r = c >> 6, ch = c & 63;
            const int ks = 2 * (ch >> 3) + ((ch & 3) >> 1), pl = (ch >> 2) & 1;
            st16(xs + ((((r >> 5) * 16 + ks) * 2 + pl) * 64 + (ch & 1) * 32 + (r & 31)) * 16, stage[i]);
        }
    }
    __syncthreads();
    if (tile_lo + wave_u >= tile_hi) return;

    const unsigned char* xfrag = xs + lane * 16;
    bf16x8 x0[MT][4], x1[MT][4];
#define PX_LDX(X_, g)                                                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int j = 0; j < 4; ++j)    \
        X_[mt][j] = *reinterpret_cast<const bf16x8*>(xfrag + ((mt * 32 + 4 * (g) + j) * 64) * 16);
#define PX_MFMA(w_, x_, c_)                                                                            \
    if constexpr (SPLIT_OUT) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_, x_, c_, 0, 0, 0);        \
    else c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x_, w_, c_, 0, 0, 0);
    // group g: 6 MT MFMAs (the small cross terms first) on set WS / XS; requests group g + 3 (set WN) and reads group g + 1's
    // activations (XN) in the MFMA gaps
#define PX_GROUP(g, WS, XS, WN, XN, NT, NG)                                                            \
    PX_LDX(XN, ((g) + 1) & 7) PX_LDW(WN, NT, NG)                                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
        PX_MFMA(w##WS##a, XS[mt][1], acc[mt]) PX_MFMA(w##WS##b, XS[mt][0], acc[mt])                    \
        PX_MFMA(w##WS##c, XS[mt][3], acc[mt]) PX_MFMA(w##WS##d, XS[mt][2], acc[mt])                    \
        PX_MFMA(w##WS##a, XS[mt][0], acc[mt]) PX_MFMA(w##WS##c, XS[mt][2], acc[mt])                    \
    }                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < MT; ++r_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
        }                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                              \
    }                                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x8, 2 * MT, 0);                                              \
    __builtin_amdgcn_sched_barrier(0);

    // fp32 output and residual through buffer descriptors that end with the matrix: a request is lane offset (row m0 + 4 half,
    // column l31) + a scalar per accumulator register and tile, added into the vector offset (the part of the address the range
    // check is certain to see); rows past M fall outside: loads give 0, stores are dropped
    const auto crsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, SPLIT_OUT ? 0 : (int)((((long long)p.M - 1) * p.ldc + p.N) * 4), 0x00020000);
    const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid), 0,
                                                         p.resid ? (int)((((long long)p.M - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
    const int c_voff = ((m0 + 4 * half) * p.ldc + l31) * 4, r_voff = ((m0 + 4 * half) * p.ldr + l31) * 4;
    (void)crsrc; (void)rrsrc; (void)c_voff; (void)r_voff;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    PX_LDX(x0, 0)
    for (int cur = tile_lo + wave_u; cur < tile_hi; cur += 4) {
        const int nxt = cur + 4 < tile_hi ? cur + 4 : cur;  // after the last tile: three groups requested again, unused
        // what the tile's epilogue needs from memory is requested now
        float bv[SPLIT_OUT ? 16 : 1];
        float rres[SPLIT_OUT ? 1 : MT][16];
        // (the row strides are re-read per tile behind an empty asm: otherwise the 32 MT per-register offsets below are hoisted
        // out of the loop into as many VGPRs, and the kernel spills)
        int ldr_s = p.ldr, ldc_s = p.ldc;
        asm volatile("" : "+s"(ldr_s), "+s"(ldc_s));
        if constexpr (SPLIT_OUT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + 32 * cur + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[4 * g + e] = b4[e];
            }
        } else {
            bv[0] = p.bias[32 * cur + l31];
            if (p.resid) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rres[mt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rrsrc, r_voff + ((32 * mt + 8 * (r >> 2) + (r & 3)) * ldr_s * 4 + 128 * cur), 0, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        PX_GROUP(0, 0, x0, 3, x1, cur, 3)
        PX_GROUP(1, 1, x1, 0, x0, cur, 4)
        PX_GROUP(2, 2, x0, 1, x1, cur, 5)
        PX_GROUP(3, 3, x1, 2, x0, cur, 6)
        PX_GROUP(4, 0, x0, 3, x1, cur, 7)
        PX_GROUP(5, 1, x1, 0, x0, nxt, 0)
        PX_GROUP(6, 2, x0, 1, x1, nxt, 1)
        PX_GROUP(7, 3, x1, 2, x0, nxt, 2)
        if constexpr (SPLIT_OUT) {
            // lane: row m0 + 32 mt + l31, columns 32 cur + 8 g + 4 half + e.  Groups 2 gp and 2 gp + 1 swap across the half-waves:
            // a lane then owns columns 16 gp + 8 half .. + 7 (the swap needs all 64 lanes: only the stores are guarded)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + 32 * mt + l31;
                unsigned char* orow = reinterpret_cast<unsigned char*>(p.C) + (long long)m * p.ldc * 4 + cur * 128 + 16 * half;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v0[4], v1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = acc[mt][8 * gp + e] + bv[8 * gp + e];
                        v1[e] = acc[mt][8 * gp + 4 + e] + bv[8 * gp + 4 + e];
                    }
                    bf16x4 h0, l0, h1, l1;
                    cn_split4(v0, h0, l0);
                    cn_split4(v1, h1, l1);
                    const uint2 ha = __builtin_bit_cast(uint2, h0), hb = __builtin_bit_cast(uint2, h1);
                    const uint2 la = __builtin_bit_cast(uint2, l0), lb = __builtin_bit_cast(uint2, l1);
                    const auto hs0 = __builtin_amdgcn_permlane32_swap(ha.x, hb.x, false, false);
                    const auto hs1 = __builtin_amdgcn_permlane32_swap(ha.y, hb.y, false, false);
                    const auto ls0 = __builtin_amdgcn_permlane32_swap(la.x, lb.x, false, false);
                    const auto ls1 = __builtin_amdgcn_permlane32_swap(la.y, lb.y, false, false);
                    if (m < p.M) {
                        *reinterpret_cast<u32x4*>(orow + 32 * gp) = u32x4{hs0[0], hs1[0], hs0[1], hs1[1]};
                        *reinterpret_cast<u32x4*>(orow + 32 * gp + 64) = u32x4{ls0[0], ls1[0], ls0[1], ls1[1]};
                    }
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mt][r] + bv[0];
                    if (p.resid) v = rres[mt][r] + p.resid_scale * v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crsrc,
                                                          c_voff + ((32 * mt + 8 * (r >> 2) + (r & 3)) * ldc_s * 4 + 128 * cur), 0, 0);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    }
#undef PX_GROUP
#undef PX_MFMA
#undef PX_LDX
#undef PX_LDW
#undef PX_WFRAG
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
    const int row_tiles = cn_ceil_div(p.M, 32 * MT), ntiles = p.N / 32;
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
    p.wp = reinterpret_cast<const unsigned char*>(a.wp);
    p.bias = a.bias;
    p.C = a.C;
    p.ldc = a.ldc;
    p.resid = a.resid;
    p.ldr = a.ldr;
    p.resid_scale = a.resid_scale;
    p.M = a.M;
    p.N = a.N;
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
