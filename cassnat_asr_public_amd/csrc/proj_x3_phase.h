// The two phases of a K = 256 split-bf16 projection over one row tile of a workgroup, as device functions: proj_x3.hip's own
// kernel is stage + barrier + tiles; the row-chain form of the split-bf16 engine (fused_x3.hip: attention output projection in
// front of the feed-forward sublayer, the next attention's Q|K|V behind it) runs the same code on row tiles that never leave LDS.
#pragma once
#include "kernels.h"

struct PxPhase {
    const unsigned char* wp;  // pack_proj_x3 stream: [N/32][16 k-steps][hi, lo][64 lanes][16 B]
    const float* bias;        // [N]
    void* C;                  // fp32 [M][ldc] or split-bf16 rows of ldc elements
    int ldc;
    const float* resid;       // fp32 [M][ldr] or null (may alias C)
    int ldr;
    float resid_scale;
    int M, N;
};

// Rows m0 .. m0 + 32 MT - 1 of A (split-bf16, 256 elements) -> LDS as MFMA fragments: per 32-row tile mt a block of 32 KiB at
// xs + mt * 32768, [16 k-steps][hi, lo][64 lanes][16 B].  A split-bf16 row is 8 groups of 32 elements, 64 B of hi halves then 64 B
// of lo halves: 16-byte chunk ch = 8 q + 4 plane + sub holds k = 32 q + 8 sub .. + 7, i.e. k-step 2 q + (sub >> 1), lane half sub & 1.
// (All requests first, then the LDS writes: written as one loop, every chunk waits for its own round trip - 8 MT in series.)
// The caller puts a barrier between this and px_tiles.
template <int MT>
__device__ __forceinline__ void px_stage_rows(const unsigned char* A, long long lda_bytes, int m0, int M, unsigned char* xs, int tid) {
    constexpr int BM = 32 * MT;
    uint4 stage[BM / 4];
#pragma unroll
    for (int i = 0; i < BM / 4; ++i) {
        const int c = tid + 256 * i, r = c >> 6, ch = c & 63;
        int m = m0 + r;
        if (m >= M) m = M - 1;
        stage[i] = ld16(A + (long long)m * lda_bytes + 16 * ch);
    }
#pragma unroll
    for (int i = 0; i < BM / 4; ++i) {
        const int c = tid + 256 * i, r = c >> 6, ch = c & 63;
        const int ks = 2 * (ch >> 3) + ((ch & 3) >> 1), pl = (ch >> 2) & 1;
        st16(xs + ((((r >> 5) * 16 + ks) * 2 + pl) * 64 + (ch & 1) * 32 + (r & 31)) * 16, stage[i]);
    }
}

// byte offset, inside a 32-row tile's 32-KiB fragment block, of the 8 bytes that hold elements k .. k + 3 (k % 4 == 0) of row r
// (hi plane; the lo plane 1024 bytes further): what a LayerNorm that keeps its output in LDS writes
__device__ __forceinline__ int px_frag_off(int r, int k) {
    return (((k >> 4) * 2) * 64 + ((k >> 3) & 1) * 32 + (r & 31)) * 16 + (k & 7) * 2;
}

// Column tiles cur = tile_lo + wave, + 4, ... < tile_hi of C for the workgroup's 32 MT rows, whose fragments sit at xs0 (tile 0) and
// xs1 (tile 1): every wave streams its tiles' pre-tiled 1-KiB weight fragments (hi, lo per k-step) from L2 into four rotating register
// sets, three groups of two k-steps in flight.  fp32 output: activations are the first MFMA operand, a lane owns one output column:
// bias, residual (requested when the tile starts) and 128-byte row segments per store; split output: weights first, a lane owns 16
// columns of one row, the half-waves trade groups (v_permlane32_swap) and every store is 16 bytes of hi or lo halves.
// DEEP (a caller with the whole register file to itself: the row-chain kernel, one 512-register wave per SIMD): eight register sets,
// seven groups in flight - a column tile is exactly eight groups, so group g lives in set g and its block requests group g - 1 of
// the wave's NEXT tile (g = 0: this tile's group 7).  With three groups in flight and nothing else on the CU to overlap with, the
// loop ran at the L2 round trip / 3 per group (measured: 60k cycles per workgroup for 24.6k of matrix pipe).
template <int MT, bool SPLIT_OUT, bool DEEP = false>
__device__ __forceinline__ void px_tiles(const PxPhase& p, const unsigned char* xs0, const unsigned char* xs1, int m0, int tile_lo,
                                         int tile_hi, int wave_u, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    const int ntiles = p.N >> 5;
    if (tile_lo + wave_u >= tile_hi) return;
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.wp), 0, ntiles * 32768, 0x00020000);
    const int lane_off = lane * 16;
#define PX_WFRAG(tile, ks, pl) \
    __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off, (((tile) * 16 + (ks)) * 2 + (pl)) * 1024, 0))
    // group g (0..7) of a column tile = k-steps 2g, 2g + 1 (a: hi, b: lo of the first; c, d of the second), in set g & 3
    bf16x8 w0a, w0b, w0c, w0d, w1a, w1b, w1c, w1d, w2a, w2b, w2c, w2d, w3a, w3b, w3c, w3d;
    bf16x8 w4a, w4b, w4c, w4d, w5a, w5b, w5c, w5d, w6a, w6b, w6c, w6d, w7a, w7b, w7c, w7d;  // (DEEP only)
#define PX_LDW(S_, tile, g)                                                                            \
    w##S_##a = PX_WFRAG(tile, 2 * (g), 0); w##S_##b = PX_WFRAG(tile, 2 * (g), 1);                      \
    w##S_##c = PX_WFRAG(tile, 2 * (g) + 1, 0); w##S_##d = PX_WFRAG(tile, 2 * (g) + 1, 1);
    const int t_first = tile_lo + wave_u;
    PX_LDW(0, t_first, 0) PX_LDW(1, t_first, 1) PX_LDW(2, t_first, 2)
    if constexpr (DEEP) {
        PX_LDW(3, t_first, 3) PX_LDW(4, t_first, 4) PX_LDW(5, t_first, 5) PX_LDW(6, t_first, 6)
    }

    const unsigned char* xfrag[2] = {xs0 + lane * 16, (MT > 1 ? xs1 : xs0) + lane * 16};
    bf16x8 x0[MT][4], x1[MT][4];
#define PX_LDX(X_, g)                                                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int j = 0; j < 4; ++j)    \
        X_[mt][j] = *reinterpret_cast<const bf16x8*>(xfrag[mt] + ((4 * (g) + j) * 64) * 16);
#define PX_MFMA(w_, x_, c_)                                                                            \
    if constexpr (SPLIT_OUT) c_ = CN_MFMA16(w_, x_, c_, 0, 0, 0);        \
    else c_ = CN_MFMA16(x_, w_, c_, 0, 0, 0);
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
        const int nxt = cur + 4 < tile_hi ? cur + 4 : cur;  // after the last tile: three (DEEP: seven) groups requested again, unused
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
        if constexpr (DEEP) {
            PX_GROUP(0, 0, x0, 7, x1, cur, 7)
            PX_GROUP(1, 1, x1, 0, x0, nxt, 0)
            PX_GROUP(2, 2, x0, 1, x1, nxt, 1)
            PX_GROUP(3, 3, x1, 2, x0, nxt, 2)
            PX_GROUP(4, 4, x0, 3, x1, nxt, 3)
            PX_GROUP(5, 5, x1, 4, x0, nxt, 4)
            PX_GROUP(6, 6, x0, 5, x1, nxt, 5)
            PX_GROUP(7, 7, x1, 6, x0, nxt, 6)
        } else {
            PX_GROUP(0, 0, x0, 3, x1, cur, 3)
            PX_GROUP(1, 1, x1, 0, x0, cur, 4)
            PX_GROUP(2, 2, x0, 1, x1, cur, 5)
            PX_GROUP(3, 3, x1, 2, x0, cur, 6)
            PX_GROUP(4, 0, x0, 3, x1, cur, 7)
            PX_GROUP(5, 1, x1, 0, x0, nxt, 0)
            PX_GROUP(6, 2, x0, 1, x1, nxt, 1)
            PX_GROUP(7, 3, x1, 2, x0, nxt, 2)
        }
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
