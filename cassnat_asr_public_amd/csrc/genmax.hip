// Fused generator for gfx950 (bf16 MFMA, d_model = 256):  per row  argmax_v and max_v of  log_softmax(W x + b)
// Replaces Generator.forward + argmax / topk(1) of the reference (src/models/cassnat.py:110-113, 378, 611) for the greedy
// path: the (M, V) logits tensor (160 MB at M = 8000, V = 5000) never exists.  (Beam search and the capture mode, which
// need full log-probability rows, keep the GEMM + row kernel.)
//
// A workgroup owns 32*MT rows: their activations sit in LDS as MFMA B fragments (read two per MFMA, one group ahead); every
// wave owns a quarter of the vocabulary and streams ITS pre-tiled 1-KiB weight fragments straight from L2 into four
// rotating register sets (buffer loads: descriptor + scalar tile offset + lane offset; three groups of four fragments in
// flight) - the stream is read once and by one wave, so an LDS hop buys nothing (the first version had a private LDS-DMA
// ring per wave: two MFMAs per request at ~70 issue cycles a request, reads -> wait -> MFMAs in series: 3.9k cycles per
// vocabulary tile of 32 MFMAs).  The product is computed swapped (logits^T: vocabulary on accumulator rows, the row index on
// the lane) so the running max / arg-max / sum-of-exponentials update is 16 in-lane values per tile; halves and waves are
// merged once at the end.  The kernel stays under 256 registers per lane: two workgroups share a CU, one's update
// arithmetic runs beside the other's MFMAs.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

struct GenmaxParams {
    const bf16* h;      // [M][256] activations (bf16)
    const uint4* wp;    // [4][VTW][16][64] fragments (pack_genmax_w)
    const float* bp;    // [4][VTW*32] biases, -inf for padded vocabulary rows
    int* arg;           // [M]
    float* maxlp;       // [M]
    int M, V, vtw;      // vtw = vocabulary tiles per wave (even)
    // GATHER variant (language-model scoring): instead of the arg-max, the log-probability of a given target per row:
    // row m = b * tgt_U + u reads tgt[b * tgt_ld + u] and writes tgt_lp[b * tgt_ld + u]
    const int* tgt;
    float* tgt_lp;
    int tgt_U, tgt_ld;
};

constexpr int GM_MAX_VTW = 48;  // vocabulary tiles per wave (V <= 6144)
template <int MT> constexpr int gm_lds_bytes() {
    constexpr int main_ = MT * 16384 + 4 * GM_MAX_VTW * 32 * 4, merge_ = 4 * 32 * MT * 16;
    return main_ > merge_ ? main_ : merge_;
}

// ---- merge the two lane halves, then the NW_ waves (ties: the lower vocabulary index wins, as torch.argmax).  The first
// barrier: every wave is done with the activation fragments and the bias table, LDS is reused for the cross-wave merge;
// GATHER: exactly one lane half of one wave saw the target
#define GM_EPILOGUE(NW_) \
_Pragma("unroll") \
    for (int mt = 0; mt < MT; ++mt) { \
        const float om = __shfl_xor(m_run[mt], 32), os = __shfl_xor(s_run[mt], 32); \
        const int oi = __shfl_xor(i_run[mt], 32); \
        const float nm = fmaxf(m_run[mt], om); \
        s_run[mt] = s_run[mt] * __expf(m_run[mt] - nm) + os * __expf(om - nm); \
        if (om > m_run[mt] || (om == m_run[mt] && oi < i_run[mt])) i_run[mt] = oi; \
        m_run[mt] = nm; \
        if constexpr (GATHER) tv[mt] = fmaxf(tv[mt], __shfl_xor(tv[mt], 32)); \
    } \
    __syncthreads(); \
    if (half == 0) { \
_Pragma("unroll") \
        for (int mt = 0; mt < MT; ++mt) { \
            float* d = merge + ((wave * BM) + 32 * mt + l31) * 4; \
            d[0] = m_run[mt]; \
            d[1] = s_run[mt]; \
            d[2] = __int_as_float(i_run[mt]); \
            d[3] = tv[mt]; \
        } \
    } \
    __syncthreads(); \
    if (tid < BM && m0 + tid < p.M) { \
        float bm = merge[tid * 4 + 0], bs = merge[tid * 4 + 1]; \
        int bi = __float_as_int(merge[tid * 4 + 2]); \
        float bt = merge[tid * 4 + 3]; \
_Pragma("unroll") \
        for (int wv = 1; wv < NW_; ++wv) { \
            const float* d = merge + (wv * BM + tid) * 4; \
            const float om = d[0], os = d[1]; \
            bt = fmaxf(bt, d[3]); \
            const int oi = __float_as_int(d[2]); \
            const float nm = fmaxf(bm, om); \
            bs = bs * __expf(bm - nm) + os * __expf(om - nm); \
            if (om > bm || (om == bm && oi < bi)) bi = oi; \
            bm = nm; \
        } \
        if constexpr (GATHER) { \
            const int m = m0 + tid; \
            p.tgt_lp[(long long)(m / p.tgt_U) * p.tgt_ld + (m % p.tgt_U)] = (bt - bm) - logf(bs); \
        } else { \
            p.arg[m0 + tid] = bi; \
            if constexpr (LSE) p.maxlp[m0 + tid] = -logf(bs); \
        } \
    }

// LSE = false: the arg-max alone (no exponentials, no log-sum-exp): what the CTC alignment of the greedy path needs
template <int MT, bool GATHER, bool LSE = true>
__global__ __launch_bounds__(256, 2) void genmax_kernel(GenmaxParams p) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned char* xs = smem;                                         // [MT][16 k-steps][64 lanes][16 B]: B fragments of the rows
    float* bias_s = reinterpret_cast<float*>(smem + MT * 16384);      // [4][vtw * 32]
    float* merge = reinterpret_cast<float*>(smem);                    // epilogue: [4 waves][BM][4] (aliases the above)

    // the wave's stream as a buffer resource: a request is descriptor + scalar offset of the fragment + lane offset
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.wp)) + (size_t)wave_u * p.vtw * 16384, 0, p.vtw * 16384, 0x00020000);
    const int lane_off = lane * 16;
#define GM_WFRAG(tile, ks) \
    __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off, ((tile) * 16 + (ks)) * 1024, 0))
    // group g (0..3) of a vocabulary tile = the fragments of k-steps 4g..4g+3, in register set g
    bf16x8 w0a, w0b, w0c, w0d, w1a, w1b, w1c, w1d, w2a, w2b, w2c, w2d, w3a, w3b, w3c, w3d;
#define GM_LDW(S_, tile, g)                                                                            \
    w##S_##a = GM_WFRAG(tile, 4 * (g) + 0); w##S_##b = GM_WFRAG(tile, 4 * (g) + 1);                    \
    w##S_##c = GM_WFRAG(tile, 4 * (g) + 2); w##S_##d = GM_WFRAG(tile, 4 * (g) + 3);
    // (GM_ROTATE: every workgroup walks its vocabulary tiles in a rotation of its own - ties go to the lower index whatever
    // the order.  -6 % for the launch alone, nothing on the benchmark, and the log-sum-exp's accumulation order then depends
    // on the workgroup a row lands in: off, so that a batch gives the same scores in a merged pass as alone.)
#ifdef GM_ROTATE
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)p.vtw);
#else
    const int rot = 0;
#endif
#define GM_TT(t) ((t) + rot >= p.vtw ? (t) + rot - p.vtw : (t) + rot)
    {
        const int t0 = GM_TT(0);
        GM_LDW(0, t0, 0) GM_LDW(1, t0, 1) GM_LDW(2, t0, 2)
    }
#ifdef GM_EXP_NO_LOAD  // timing experiments (wrong results): the loop without its weight requests / without the update arithmetic
#define GM_LDW_LOOP(S_, tile, g)
#else
#define GM_LDW_LOOP(S_, tile, g) GM_LDW(S_, tile, g)
#endif

    // activations -> LDS: 16-byte chunk c = 2 ks + half of row r at fragment (mt = r >> 5, ks), lane slot 32 half + (r & 31)
    // (all requests - rows and bias table, 16 bytes each - before the first LDS write: written as copy loops, every chunk
    // waited for its own round trip, 4 MT + 10 of them in series)
    {
        constexpr int NB = (4 * GM_MAX_VTW * 32 / 4 + 255) / 256;
        const int nb16 = 4 * p.vtw * 32 / 4;
        uint4 stage[BM / 8], bst[NB];
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int c = tid + 256 * i, r = c >> 5, ch = c & 31;
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            stage[i] = ld16(reinterpret_cast<const unsigned char*>(p.h + (long long)m * 256) + 16 * ch);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + 256 * i < nb16) bst[i] = ld16(reinterpret_cast<const unsigned char*>(p.bp) + 16 * (tid + 256 * i));
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int c = tid + 256 * i, r = c >> 5, ch = c & 31;
            st16(xs + ((((r >> 5) * 16 + (ch >> 1)) * 64) + (ch & 1) * 32 + (r & 31)) * 16, stage[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + 256 * i < nb16) st16(reinterpret_cast<unsigned char*>(bias_s) + 16 * (tid + 256 * i), bst[i]);
    }
    __syncthreads();

    float m_run[MT], s_run[MT];
    int i_run[MT];
    int tg[MT];      // GATHER: this lane's rows' target labels and their logits once seen
    float tv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        m_run[mt] = CN_NEG_FILL;
        s_run[mt] = 0.f;
        i_run[mt] = 0;
        tg[mt] = -1;
        tv[mt] = -INFINITY;
        if constexpr (GATHER) {
            int m = m0 + 32 * mt + l31;
            if (m >= p.M) m = p.M - 1;
            tg[mt] = p.tgt[(long long)(m / p.tgt_U) * p.tgt_ld + (m % p.tgt_U)];
        }
    }

    // activation fragments of a group: xq[mt][j] = k-step 4g + j of M-tile mt; two sets, group g + 1's reads ride beside group g's MFMAs
    const unsigned char* xfrag = xs + lane * 16;
    bf16x8 x0[MT][4], x1[MT][4];
#define GM_LDX(X_, g)                                                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int j = 0; j < 4; ++j)    \
        X_[mt][j] = *reinterpret_cast<const bf16x8*>(xfrag + ((mt * 16 + 4 * (g) + j) * 64) * 16);
    // group g: 4 MT MFMAs on set WS / XS; requests group g + 3 (set WN: of this tile, or 0..2 of the next) and reads group g + 1's
    // activations (XN) in the MFMA gaps
#define GM_GROUP(g, WS, XS, WN, XN, NT, NG)                                                            \
    GM_LDX(XN, ((g) + 1) & 3) GM_LDW_LOOP(WN, NT, NG)                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
        acc[mt] = CN_MFMA16(w##WS##a, XS[mt][0], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##b, XS[mt][1], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##c, XS[mt][2], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##d, XS[mt][3], acc[mt], 0, 0, 0);      \
    }                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < MT; ++r_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
        }                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                              \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
    // fold one finished vocabulary tile into the running statistics
#define GM_FOLD(tile)                                                                                  \
    {                                                                                                  \
        const int vbase = 32 * (wave_u * p.vtw + (tile)) + 4 * half;                                   \
        const float* bt_ = bias_s + 32 * (wave_u * p.vtw + (tile)) + 4 * half;                         \
        float bv[16];                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                \
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt_ + 8 * g);                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[4 * g + e] = b4[e];                       \
        }                                                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                            \
            float tmax = -INFINITY;                                                                    \
            int tidx = 0;                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                           \
                const float v = acc[mt][r] + bv[r];                                                    \
                acc[mt][r] = v;                                                                        \
                if (v > tmax) { tmax = v; tidx = vbase + (r & 3) + 8 * (r >> 2); }                     \
                if constexpr (GATHER) { if (vbase + (r & 3) + 8 * (r >> 2) == tg[mt]) tv[mt] = v; }    \
            }                                                                                          \
            if (tmax > m_run[mt] || (tmax == m_run[mt] && tidx < i_run[mt])) {                         \
                if constexpr (LSE) s_run[mt] *= __expf(m_run[mt] - tmax);                              \
                m_run[mt] = tmax;                                                                      \
                i_run[mt] = tidx;                                                                      \
            }                                                                                          \
            if constexpr (LSE) {                                                                       \
                float ps = 0.f;                                                                        \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) ps += __expf(acc[mt][r] - m_run[mt]);   \
                s_run[mt] += ps;                                                                       \
            }                                                                                          \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;                           \
        }                                                                                              \
    }

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    GM_LDX(x0, 0)
    for (int t = 0; t < p.vtw; ++t) {
        const int cur = GM_TT(t);
        const int nxt = GM_TT(t + 1 < p.vtw ? t + 1 : t);  // after the last tile: three groups requested again, unused
        GM_GROUP(0, 0, x0, 3, x1, cur, 3)
        GM_GROUP(1, 1, x1, 0, x0, nxt, 0)
        GM_GROUP(2, 2, x0, 1, x1, nxt, 1)
        GM_GROUP(3, 3, x1, 2, x0, nxt, 2)
#ifndef GM_EXP_NO_FOLD
        GM_FOLD(cur)
#else  // (one use of every accumulator keeps the products alive)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            m_run[mt] = fmaxf(m_run[mt], acc[mt][0]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        }
#endif
    }
#undef GM_GROUP
#undef GM_LDX
#undef GM_LDW
#undef GM_LDW_LOOP
#undef GM_WFRAG

    GM_EPILOGUE(4)
}

// ---- the same tail in the split-bf16 precision (CN_PREC_X3): operands are hi + lo bf16 pairs, every product three MFMAs
// (W_hi x_lo + W_lo x_hi + W_hi x_hi, fp32 accumulation), so the arg-max and the log-probabilities meet the fp32 gate.
// Eight waves share a workgroup's rows (an eighth of the vocabulary each: twice the bytes per row in LDS and per weight in
// the stream, so twice the waves per staged row); a group is two k-steps = four 1-KiB weight fragments (hi, lo, hi, lo) and
// 6 MT MFMAs; register sets, look-ahead, fold and merge are those of the bf16 kernel.
constexpr int GM3_WAVES = 8;
constexpr int GM3_MAX_VTW = 24;  // vocabulary tiles per wave (V <= 6144)
template <int MT> constexpr int gm3_lds_bytes() { return MT * 32768 + GM3_WAVES * GM3_MAX_VTW * 32 * 4; }

template <int MT, bool GATHER, bool LSE = true>
__global__ __launch_bounds__(512, 1) void genmax_x3_kernel(GenmaxParams p) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned char* xs = smem;                                         // [MT][16 k-steps][hi, lo][64 lanes][16 B]
    float* bias_s = reinterpret_cast<float*>(smem + MT * 32768);      // [8][vtw * 32]
    float* merge = reinterpret_cast<float*>(smem);                    // epilogue: [8 waves][BM][4] (aliases the above)

    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.wp)) + (size_t)wave_u * p.vtw * 32768, 0, p.vtw * 32768, 0x00020000);
    const int lane_off = lane * 16;
#define GM3_WFRAG(tile, ks, pl) \
    __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off, (((tile) * 16 + (ks)) * 2 + (pl)) * 1024, 0))
    // group g (0..7) of a vocabulary tile = k-steps 2g, 2g + 1 (a: hi, b: lo of the first; c, d of the second), in set g & 3
    bf16x8 w0a, w0b, w0c, w0d, w1a, w1b, w1c, w1d, w2a, w2b, w2c, w2d, w3a, w3b, w3c, w3d;
#define GM3_LDW(S_, tile, g)                                                                           \
    w##S_##a = GM3_WFRAG(tile, 2 * (g), 0); w##S_##b = GM3_WFRAG(tile, 2 * (g), 1);                    \
    w##S_##c = GM3_WFRAG(tile, 2 * (g) + 1, 0); w##S_##d = GM3_WFRAG(tile, 2 * (g) + 1, 1);
    GM3_LDW(0, 0, 0) GM3_LDW(1, 0, 1) GM3_LDW(2, 0, 2)

    // activations -> LDS.  A split-bf16 row is 8 groups of 32 elements, 64 B of hi halves then 64 B of lo halves: 16-byte
    // chunk ch = 8 q + 4 plane + sub holds k = 32 q + 8 sub .. + 7, i.e. k-step 2 q + (sub >> 1), lane half sub & 1
    {
        constexpr int NB = (GM3_WAVES * GM3_MAX_VTW * 32 / 4 + 511) / 512;
        const int nb16 = GM3_WAVES * p.vtw * 32 / 4;
        uint4 stage[BM / 8], bst[NB];
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int c = tid + 512 * i, r = c >> 6, ch = c & 63;
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            stage[i] = ld16(reinterpret_cast<const unsigned char*>(p.h) + (long long)m * 1024 + 16 * ch);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + 512 * i < nb16) bst[i] = ld16(reinterpret_cast<const unsigned char*>(p.bp) + 16 * (tid + 512 * i));
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int c = tid + 512 * i, r = c >> 6, ch = c & 63;
            const int ks = 2 * (ch >> 3) + ((ch & 3) >> 1), pl = (ch >> 2) & 1;
            st16(xs + ((((r >> 5) * 16 + ks) * 2 + pl) * 64 + (ch & 1) * 32 + (r & 31)) * 16, stage[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + 512 * i < nb16) st16(reinterpret_cast<unsigned char*>(bias_s) + 16 * (tid + 512 * i), bst[i]);
    }
    __syncthreads();

    float m_run[MT], s_run[MT];
    int i_run[MT];
    int tg[MT];
    float tv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        m_run[mt] = CN_NEG_FILL;
        s_run[mt] = 0.f;
        i_run[mt] = 0;
        tg[mt] = -1;
        tv[mt] = -INFINITY;
        if constexpr (GATHER) {
            int m = m0 + 32 * mt + l31;
            if (m >= p.M) m = p.M - 1;
            tg[mt] = p.tgt[(long long)(m / p.tgt_U) * p.tgt_ld + (m % p.tgt_U)];
        }
    }

    // activation fragments of a group: x[mt][0..3] = hi, lo of k-step 2g, hi, lo of k-step 2g + 1
    const unsigned char* xfrag = xs + lane * 16;
    bf16x8 x0[MT][4], x1[MT][4];
#define GM3_LDX(X_, g)                                                                                 \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int j = 0; j < 4; ++j)    \
        X_[mt][j] = *reinterpret_cast<const bf16x8*>(xfrag + ((mt * 32 + 4 * (g) + j) * 64) * 16);
    // group g: 6 MT MFMAs (the small cross terms first) on set WS / XS; requests group g + 3 (set WN) and reads group g + 1's
    // activations (XN) in the MFMA gaps
#define GM3_GROUP(g, WS, XS, WN, XN, NT, NG)                                                           \
    GM3_LDX(XN, ((g) + 1) & 7) GM3_LDW(WN, NT, NG)                                                     \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
        acc[mt] = CN_MFMA16(w##WS##a, XS[mt][1], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##b, XS[mt][0], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##c, XS[mt][3], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##d, XS[mt][2], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##a, XS[mt][0], acc[mt], 0, 0, 0);      \
        acc[mt] = CN_MFMA16(w##WS##c, XS[mt][2], acc[mt], 0, 0, 0);      \
    }                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < MT; ++r_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
        }                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                              \
    }                                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x8, 2 * MT, 0);                                              \
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    GM3_LDX(x0, 0)
    for (int t = 0; t < p.vtw; ++t) {
        const int cur = t;
        const int nxt = t + 1 < p.vtw ? t + 1 : t;  // after the last tile: three groups requested again, unused
        GM3_GROUP(0, 0, x0, 3, x1, cur, 3)
        GM3_GROUP(1, 1, x1, 0, x0, cur, 4)
        GM3_GROUP(2, 2, x0, 1, x1, cur, 5)
        GM3_GROUP(3, 3, x1, 2, x0, cur, 6)
        GM3_GROUP(4, 0, x0, 3, x1, cur, 7)
        GM3_GROUP(5, 1, x1, 0, x0, nxt, 0)
        GM3_GROUP(6, 2, x0, 1, x1, nxt, 1)
        GM3_GROUP(7, 3, x1, 2, x0, nxt, 2)
        GM_FOLD(cur)
    }
#undef GM3_GROUP
#undef GM3_LDX
#undef GM3_LDW
#undef GM3_WFRAG
    GM_EPILOGUE(GM3_WAVES)
}
#undef GM_FOLD
#undef GM_TT
#undef GM_EPILOGUE

template <int MT, bool GATHER, bool LSE> static int launch_genmax_x3_variant(const GenmaxParams& p, hipStream_t s) {
    const int main_ = MT * 32768 + GM3_WAVES * p.vtw * 32 * 4, merge_ = GM3_WAVES * 32 * MT * 16;
    const int lds = main_ > merge_ ? main_ : merge_;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)genmax_x3_kernel<MT, GATHER, LSE>, hipFuncAttributeMaxDynamicSharedMemorySize, gm3_lds_bytes<MT>()));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL((genmax_x3_kernel<MT, GATHER, LSE>), dim3(cn_ceil_div(p.M, 32 * MT)), dim3(512), lds, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}


template <int MT, bool GATHER, bool LSE> static int launch_genmax_variant(const GenmaxParams& p, hipStream_t s) {
    // (sized for this vocabulary, not for the largest one: at V = 5000 three workgroups fit a CU's LDS)
    const int main_ = MT * 16384 + 4 * p.vtw * 32 * 4, merge_ = 4 * 32 * MT * 16;
    const int lds = main_ > merge_ ? main_ : merge_;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)genmax_kernel<MT, GATHER, LSE>, hipFuncAttributeMaxDynamicSharedMemorySize, gm_lds_bytes<MT>()));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL((genmax_kernel<MT, GATHER, LSE>), dim3(cn_ceil_div(p.M, 32 * MT)), dim3(256), lds, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int genmax_vtw(int V) {
    int vtw = cn_ceil_div(cn_ceil_div(V, 32), 4);
    return vtw + (vtw & 1);
}
int genmax_x3_vtw(int V) { return cn_ceil_div(cn_ceil_div(V, 32), GM3_WAVES); }
bool genmax_applies(int prec, int d, int V) {
    if (d != 256 || V < 1) return false;
    if (prec == CN_PREC_BF16) return genmax_vtw(V) <= GM_MAX_VTW;
    if (prec == CN_PREC_X3) return genmax_x3_vtw(V) <= GM3_MAX_VTW;
    return false;
}

int launch_genmax(const GenmaxArgs& a, hipStream_t s) {
    const int vtw = a.x3 ? genmax_x3_vtw(a.V) : genmax_vtw(a.V);
    if (!genmax_applies(a.x3 ? CN_PREC_X3 : CN_PREC_BF16, a.d, a.V)) {
        cn_set_error("genmax: needs d_model == 256 and V <= 6144");
        return -1;
    }
    if (a.M <= 0) return 0;
    GenmaxParams p;
    p.h = reinterpret_cast<const bf16*>(a.h);  // (split-bf16 rows when a.x3: 1024 bytes each)
    p.wp = reinterpret_cast<const uint4*>(a.wp);
    p.bp = a.bp;
    p.arg = a.arg;
    p.maxlp = a.maxlp;
    p.M = a.M;
    p.V = a.V;
    p.vtw = vtw;
    p.tgt = a.tgt;
    p.tgt_lp = a.tgt_lp;
    p.tgt_U = a.tgt_U;
    p.tgt_ld = a.tgt_ld;
    // split-bf16: one workgroup (8 waves) per CU; 32 rows each while that fills the chip in one round, 64 beyond (half the
    // weight stream per row)
    const bool wide = cn_ceil_div(a.M, 32) > 256;
    if (a.tgt) {
        if (!a.tgt_lp || a.tgt_U < 1 || a.tgt_ld < a.tgt_U || a.M % a.tgt_U != 0) {
            cn_set_error("genmax: the target gather needs rows = B x U, an output buffer and ld >= U");
            return -1;
        }
        if (a.x3) return wide ? launch_genmax_x3_variant<2, true, true>(p, s) : launch_genmax_x3_variant<1, true, true>(p, s);
        return a.M > 32 ? launch_genmax_variant<2, true, true>(p, s) : launch_genmax_variant<1, true, true>(p, s);
    }
    if (a.x3) {
        if (!a.maxlp) return wide ? launch_genmax_x3_variant<2, false, false>(p, s) : launch_genmax_x3_variant<1, false, false>(p, s);
        return wide ? launch_genmax_x3_variant<2, false, true>(p, s) : launch_genmax_x3_variant<1, false, true>(p, s);
    }
    if (!a.maxlp)  // arg-max only
        return a.M > 32 ? launch_genmax_variant<2, false, false>(p, s) : launch_genmax_variant<1, false, false>(p, s);
    return a.M > 32 ? launch_genmax_variant<2, false, true>(p, s) : launch_genmax_variant<1, false, true>(p, s);
}

static inline uint16_t gm_bf16_bits(float f) { return cn_host_op16(f); }  // (the engine's 16-bit operand: common.h)

// W [V][256] fp32 -> [4][vtw][16][64][8] bf16: frag(w, t, ks, lane)[j] = W[32(w*vtw + t) + (lane&31)][16ks + 8(lane>>5) + j]
// (zero rows past V);  b [V] -> [4*vtw*32] fp32 with -inf past V.
void pack_genmax(const float* w, const float* b, int V, uint16_t* wout, float* bout) {
    const int vtw = genmax_vtw(V);
    for (int wv = 0; wv < 4; ++wv)
        for (int t = 0; t < vtw; ++t) {
            for (int ks = 0; ks < 16; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int v = 32 * (wv * vtw + t) + (lane & 31);
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * ks + 8 * (lane >> 5) + j;
                        wout[((((size_t)wv * vtw + t) * 16 + ks) * 64 + lane) * 8 + j] =
                            v < V ? gm_bf16_bits(w[(size_t)v * 256 + k]) : 0;
                    }
                }
            for (int i = 0; i < 32; ++i) {
                const int v = 32 * (wv * vtw + t) + i;
                bout[((size_t)wv * vtw + t) * 32 + i] = v < V ? b[v] : -INFINITY;
            }
        }
}

// W [V][256] fp32 -> [8][vtw][16][hi, lo][64][8] bf16 (hi = bf16(w), lo = bf16(w - hi)), same lane map;  b -> [8*vtw*32] fp32
void pack_genmax_x3(const float* w, const float* b, int V, uint16_t* wout, float* bout) {
    const int vtw = genmax_x3_vtw(V);
    for (int wv = 0; wv < GM3_WAVES; ++wv)
        for (int t = 0; t < vtw; ++t) {
            for (int ks = 0; ks < 16; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int v = 32 * (wv * vtw + t) + (lane & 31);
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * ks + 8 * (lane >> 5) + j;
                        uint16_t hi = 0, lo = 0;
                        if (v < V) {
                            const float f = w[(size_t)v * 256 + k];
                            hi = gm_bf16_bits(f);
                            const uint32_t hb = (uint32_t)hi << 16;
                            float hf;
                            memcpy(&hf, &hb, 4);
                            lo = gm_bf16_bits(f - hf);
                        }
                        const size_t o = (((((size_t)wv * vtw + t) * 16 + ks) * 2) * 64 + lane) * 8 + j;
                        wout[o] = hi;
                        wout[o + 512] = lo;
                    }
                }
            for (int i = 0; i < 32; ++i) {
                const int v = 32 * (wv * vtw + t) + i;
                bout[((size_t)wv * vtw + t) * 32 + i] = v < V ? b[v] : -INFINITY;
            }
        }
}
