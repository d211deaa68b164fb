// First subsampling convolution: Conv2d(1, C, 3, stride 2, pad 1) + ReLU on padded (B,T,F) fbank frames
// (reference: src/models/modules/embedding.py:103,112-114).  HBM-bound: 9 MACs per output element, so
// the kernel is shaped by its store: each lane produces 8 consecutive channels of one (b,t1,f1) position
// and writes them with one 16-byte (bf16) / two 16-byte (f32) stores into the channels-last image
// (B,T1,F1,C) that the implicit-GEMM second convolution reads; a wave covers two full 512-B/1-KiB rows.
#include <cstdlib>

#include "kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// PLANES (T = bf16): the split-bf16 image as two bf16 images, the hi halves at `out` and the lo halves `lo_off` elements
// further, both in the bordered layout - what the LDS-DMA conv2 kernel reads in its split form (conv2.hip, X3)
// MIXP (T = bf16 as a 2-byte cell): the three planes of conv2's MIX form (conv2.hip) - half-precision hi values at `out`, then
// (`lo_off` = the image's cell count) the bytes l = e4m3((v - hi) out_scale) at 2 lo_off and q = e4m3(v out_scale2) at 3 lo_off
template <typename T, bool PLANES = false, bool MIXP = false>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                    const float* __restrict__ bias, T* __restrict__ out, int B, int Tn,
                                                    int F, int T1, int F1, int C, int halo, long long lo_off = 0,
                                                    const UttMeta* __restrict__ utt_meta = nullptr, float out_scale = 1.f,
                                                    float out_scale2 = 1.f) {
    // A thread keeps its 8 channels for the whole kernel, so the 72 tap weights + 8 biases live in registers (as pairs: the
    // 72 multiply-adds of an output position are 36 v_pk_fma_f32).  The three input rows of an output row are staged once in
    // LDS with their zero padding (index f + 1, f = -1 .. F), so a position's nine taps are nine unconditional LDS reads - the
    // 32 lanes of a position read the same address - instead of nine predicated global loads, each in a branch of its own.
    extern __shared__ float xs_all[];  // [2][3][F + 2]: double-buffered over the workgroup's rows (one barrier per row)
    const int FW = F + 2;
    const int cg = C >> 3;
    // halo = 1: the image is written as [B][T1 + 2][F1 + 2][C] with a border of zeros (the padding of the second
    // convolution, conv2.hip): the grid then covers the border cells too.
    // Work decomposition: a workgroup takes whole image rows (b, t1) in a grid-stride loop (one 32-bit division per
    // row, not four 64-bit ones per output); inside a row its 256 threads are (256 / cg) positions x cg channel groups.
    // halo = 2: the same bordered layout, but the border cells hold their zeros already (same shape as the previous image
    // in this buffer): they are skipped instead of rewritten
    const bool skip_border = halo == 2;
    halo = halo ? 1 : 0;
    const int T1p = T1 + 2 * halo, F1p = F1 + 2 * halo;
    const int c0 = (int)(threadIdx.x % cg) << 3;
    const int p0 = threadIdx.x / cg, pstep = 256 / cg;
    f32x2 w[9][4], bz[4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[tap][j] = f32x2{w9c[tap * C + c0 + 2 * j], w9c[tap * C + c0 + 2 * j + 1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) bz[j] = f32x2{bias[c0 + 2 * j], bias[c0 + 2 * j + 1]};
    const int rows = B * T1p;
    int buf = 0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x, buf ^= 1) {
        const int b = row / T1p, t1 = row - b * T1p - halo;
        T* orow = out + (long long)row * F1p * C + (__is_same(T, split_t) ? 0 : c0);
        // merged pass: the utterance's own batch is `tl` frames long - later frames are zero padding, later image rows zeros
        const int tl = utt_meta ? utt_meta[b].frames : Tn;
        const int t1_own = utt_meta ? (tl - 1) / 2 + 1 : T1;
        const bool row_in = t1 >= 0 && t1 < t1_own;
        const bool past_own = t1 >= t1_own && t1 < T1;  // an interior row of the merged image that the own batch does not have
        float* xs = xs_all + buf * 3 * FW;
        if (row_in) {
            for (int i = threadIdx.x; i < 3 * FW; i += 256) {
                const int kh = i / FW, f = i - kh * FW - 1, t = 2 * t1 - 1 + kh;
                xs[i] = (t >= 0 && t < tl && f >= 0 && f < F) ? x[((long long)b * Tn + t) * F + f] : 0.f;
            }
        }
        __syncthreads();  // (a buffer is rewritten two rows later: every thread has passed the barrier in between)
        for (int fp = p0; fp < F1p; fp += pstep) {
            const int f1 = fp - halo;
            T* dst = orow + (long long)fp * C;
            const bool cell_in = row_in && f1 >= 0 && f1 < F1;
            if (!cell_in) {  // border cell (bordered images only), or a row past the utterance's own batch (written as zeros)
                if (skip_border && !past_own) continue;
                if constexpr (sizeof(T) == 1) {  // e4m3fn image: 8 channels = 8 bytes
                    *reinterpret_cast<uint2*>(dst) = make_uint2(0, 0);
                } else if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
                    if constexpr (PLANES) *reinterpret_cast<uint4*>(dst + lo_off) = make_uint4(0, 0, 0, 0);
                    if constexpr (MIXP) {
                        unsigned char* b8 = reinterpret_cast<unsigned char*>(out) + 2 * lo_off + (dst - out);
                        *reinterpret_cast<uint2*>(b8) = make_uint2(0, 0);
                        *reinterpret_cast<uint2*>(b8 + lo_off) = make_uint2(0, 0);
                    }
                } else if constexpr (!__is_same(T, split_t)) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(dst + 4) = make_uint4(0, 0, 0, 0);
                } else {  // split-bf16 (no bordered form: only rows past the utterance's own batch come here): hi and lo halves
                    unsigned char* db = reinterpret_cast<unsigned char*>(dst) + cn_split_off((size_t)c0);
                    *reinterpret_cast<uint4*>(db) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(db + 64) = make_uint4(0, 0, 0, 0);
                }
                continue;
            }
            f32x2 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = bz[j];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float v = xs[kh * FW + 2 * f1 + kw];  // f = 2 f1 - 1 + kw at index f + 1
                    const f32x2 vv = {v, v};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_elementwise_fma(vv, w[kh * 3 + kw][j], acc[j]);
                }
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] = fmaxf(acc[j][0], 0.f);
                o[2 * j + 1] = fmaxf(acc[j][1], 0.f);
            }
            if constexpr (sizeof(T) == 1) {  // e4m3fn at `out_scale`, saturating (the values are >= 0 already)
                unsigned w0 = 0, w1 = 0;
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(o[0] * out_scale, CN_FP8_MAX), fminf(o[1] * out_scale, CN_FP8_MAX), w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(o[2] * out_scale, CN_FP8_MAX), fminf(o[3] * out_scale, CN_FP8_MAX), w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(o[4] * out_scale, CN_FP8_MAX), fminf(o[5] * out_scale, CN_FP8_MAX), w1, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(o[6] * out_scale, CN_FP8_MAX), fminf(o[7] * out_scale, CN_FP8_MAX), w1, true);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_nontemporal_store(u32x2{w0, w1}, reinterpret_cast<u32x2*>(dst));
            } else if constexpr (__is_same(T, split_t)) {  // (no bordered image in this precision: halo == 0)
                bf16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    hi[j] = (bf16)o[j];
                    lo[j] = (bf16)(o[j] - (float)hi[j]);
                }
                unsigned char* db = reinterpret_cast<unsigned char*>(dst) + cn_split_off((size_t)c0);
                __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8*>(db));
                __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(db + 64));
            } else if constexpr (MIXP) {
                typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                h16x8 hv;
                float lo[8], qv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    hv[j] = (_Float16)o[j];
                    lo[j] = __builtin_amdgcn_fmed3f((o[j] - (float)hv[j]) * out_scale, -CN_FP8_MAX, CN_FP8_MAX);
                    qv[j] = fminf(o[j] * out_scale2, CN_FP8_MAX);
                }
                __builtin_nontemporal_store(hv, reinterpret_cast<h16x8*>(dst));
                unsigned l0 = 0, l1 = 0, q0 = 0, q1 = 0;
                l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l0, false);
                l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l0, true);
                l1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4], lo[5], l1, false);
                l1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[6], lo[7], l1, true);
                q0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], q0, false);
                q0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], q0, true);
                q1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4], qv[5], q1, false);
                q1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[6], qv[7], q1, true);
                unsigned char* b8 = reinterpret_cast<unsigned char*>(out) + 2 * lo_off + (dst - out);
                __builtin_nontemporal_store(u32x2{l0, l1}, reinterpret_cast<u32x2*>(b8));
                __builtin_nontemporal_store(u32x2{q0, q1}, reinterpret_cast<u32x2*>(b8 + lo_off));
            } else if constexpr (sizeof(T) == 2) {
                bf16x8 ob;
#pragma unroll
                for (int j = 0; j < 8; ++j) ob[j] = (bf16)o[j];
                // (streamed out past L2: the image is 345 MB per batch, written once and read once by conv2, whose weight slabs
                // then stay resident - conv1 -4 %, conv2 -5 %, the benchmark +1.2 % A/B)
                __builtin_nontemporal_store(ob, reinterpret_cast<bf16x8*>(dst));
                if constexpr (PLANES) {
                    bf16x8 lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) lo[j] = (bf16)(o[j] - (float)ob[j]);
                    __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(dst + lo_off));
                }
            } else {
                f32x4 o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0[j] = o[j];
                    o1[j] = o[4 + j];
                }
                *reinterpret_cast<f32x4*>(dst) = o0;
                *reinterpret_cast<f32x4*>(dst + 4) = o1;
            }
        }
    }
}

int launch_conv1(int prec, const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1,
                 int F1, int C, int halo, hipStream_t s, const UttMeta* utt_meta) {
    if (C % 8 != 0 || (256 % (C / 8)) != 0) {
        cn_set_error("conv1: channel count must be a multiple of 8 with C/8 dividing 256");
        return -1;
    }
    long long blocks = (long long)B * (T1 + 2 * (halo ? 1 : 0));  // one image row per workgroup pass
    if (blocks > 256 * 8) blocks = 256 * 8;             // 8 workgroups per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)2 * 3 * (F + 2) * sizeof(float);  // the rows being read and the next ones
    if (prec == CN_PREC_X3 && (halo || C % 32 != 0)) {
        cn_set_error("conv1: the split-bf16 image has no halo form and needs C % 32 == 0");
        return -1;
    }
    if (prec == CN_PREC_F32)
        hipLaunchKernelGGL(conv1_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, s, x, w9c, bias, (float*)out, B, T,
                           F, T1, F1, C, halo, 0ll, utt_meta);
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(conv1_kernel<split_t>, dim3((unsigned)blocks), dim3(256), lds, s, x, w9c, bias, (split_t*)out, B, T,
                           F, T1, F1, C, halo, 0ll, utt_meta);
    else
        hipLaunchKernelGGL(conv1_kernel<bf16>, dim3((unsigned)blocks), dim3(256), lds, s, x, w9c, bias, (bf16*)out, B, T,
                           F, T1, F1, C, halo, 0ll, utt_meta);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// the split-bf16 engine's image for the LDS-DMA conv2 kernel: two bordered bf16 planes ([B][T1 + 2][F1 + 2][C] each; hi at
// `out`, lo right behind it).  halo: 1 = write the border zeros, 2 = they are there already (launch_conv1)
int launch_conv1_planes(const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1, int F1,
                        int C, int halo, hipStream_t s, const UttMeta* utt_meta) {
    if (C % 8 != 0 || (256 % (C / 8)) != 0 || (halo != 1 && halo != 2)) {
        cn_set_error("conv1 (planes): channel count must be a multiple of 8 with C/8 dividing 256; the image is always bordered");
        return -1;
    }
    long long blocks = (long long)B * (T1 + 2);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)2 * 3 * (F + 2) * sizeof(float);
    const long long plane = (long long)B * (T1 + 2) * (F1 + 2) * C;
    hipLaunchKernelGGL((conv1_kernel<bf16, true>), dim3((unsigned)blocks), dim3(256), lds, s, x, w9c, bias, (bf16*)out, B, T, F, T1,
                       F1, C, halo, plane, utt_meta);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// the three planes of conv2's MIX form (conv2.hip): half-precision hi values, then l = e4m3((v - hi) scale_l) bytes, then
// q = e4m3(v scale_q) bytes, each bordered ([B][T1 + 2][F1 + 2][C] cells); `out` holds 4 bytes per cell
int launch_conv1_mixplanes(const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1, int F1, int C,
                           int halo, float scale_l, float scale_q, hipStream_t s, const UttMeta* utt_meta) {
    if (C % 8 != 0 || (256 % (C / 8)) != 0 || (halo != 1 && halo != 2)) {
        cn_set_error("conv1 (mix planes): channel count must be a multiple of 8 with C/8 dividing 256; the image is always bordered");
        return -1;
    }
    long long blocks = (long long)B * (T1 + 2);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)2 * 3 * (F + 2) * sizeof(float);
    const long long cells = (long long)B * (T1 + 2) * (F1 + 2) * C;
    hipLaunchKernelGGL((conv1_kernel<bf16, false, true>), dim3((unsigned)blocks), dim3(256), lds, s, x, w9c, bias, (bf16*)out, B, T, F, T1,
                       F1, C, halo, cells, utt_meta, scale_l, scale_q);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- e4m3 image on the matrix cores -------------------------------------------------------------------------------
// The VALU kernel above is shaped by its store; with one byte per element the store halves and what is left is its arithmetic
// (36 packed FMAs, 8 maxima, conversions per 8 channels: 0.70 ms per ten-batch pass - more than the bf16 image's 0.64, which is
// HBM-bound).  Here a position's 256 channels are one column of eight 32x32x16 bf16 MFMAs: out^T[channel][position] =
// W[channel][k] . X[k][position] with k = nine taps, then a constant 1 that carries the bias (the remaining six k are zero);
// the image scale is folded into W and the bias (a power of two: exact).  A wave takes 32 consecutive cells of an utterance's
// bordered image; its rows of channels are permuted so that a lane ends up with 16 consecutive channels per tile (one
// 16-byte piece), the tile goes through a wave-private LDS pad and leaves as 1-KiB contiguous stores (4 cells x 256 B).
// Inputs and weights go in as split bf16 (hi + lo halves, three MFMAs per tile: lo.hi + hi.lo + hi.hi - the matrix pipe has the
// time) so that the sums carry 16 significant bits in front of the e4m3 rounding: with plain bf16 operands some 6 % of the
// cells would land one code away from the fp32 convolution's (tests/test_gpu_kernels.py::test_conv_frontend_fp8 allows 0.2 %).
constexpr int C1M_ROWB = 272;                      // bytes of one cell's row in the pad (256 + 16: bank spread)
constexpr int C1M_PAD = 32 * C1M_ROWB;             // per wave
constexpr int C1M_PRE = 6;                         // staged floats per thread: 256 x 6 >= C1M_MAXROWS x (F + 2), i.e. F <= 126
constexpr int C1M_MAXROWS = 12;                    // input rows a block of 128 cells can need (F1 + 2 >= 32: five image rows)

// OUT8 = false: the same kernel writing the bf16 engine's bordered image (512 bytes per cell, scale 1): a cell's row crosses the pad
// in two halves of four tiles (128 channels = 256 bytes each), so the pad and the store pattern are the e4m3 form's.
template <bool OUT8>
__global__ __launch_bounds__(256) void conv1_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                         const float* __restrict__ bias, unsigned char* __restrict__ out,
                                                         int B, int Tn, int F, int T1, int F1, float scale,
                                                         const UttMeta* __restrict__ utt_meta) {
    constexpr int CELLB = OUT8 ? 256 : 512;  // bytes of a cell's 256 channels in the image
    extern __shared__ __attribute__((aligned(16))) unsigned char c1m_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int FW = F + 2, T1p = T1 + 2, F1p = F1 + 2, P = T1p * F1p;
    float* xs = reinterpret_cast<float*>(c1m_smem);                       // [C1M_MAXROWS][FW]
    unsigned char* pad = c1m_smem + ((C1M_MAXROWS * FW * 4 + 15) & ~15) + wave * C1M_PAD;
    // weight fragments (A operands), kept for the whole kernel: tile nt, row r -> channel 32 nt + 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3)
    bf16x8 wf[8], wl[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int ch = 32 * nt + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * half + j;
            const float v = (k < 9 ? w9c[k * 256 + ch] : (k == 9 ? bias[ch] : 0.f)) * scale;
            wf[nt][j] = (bf16)v;
            wl[nt][j] = (bf16)(v - (float)wf[nt][j]);
        }
    }
    // this lane's tap offsets inside the staged rows: tap = 8 half + j -> (tap / 3) rows down, tap % 3 to the right; -1 = none
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int tap = 8 * half + j;
        toff[j] = tap < 9 ? (tap / 3) * FW + tap % 3 : -1;
    }
    const int nblk = (P + 127) / 128;
    // a block's input rows travel in registers while the block before it is computed (the workgroup is resident twice per CU: with
    // the rows requested at the top of a block, the round trip to L2 / HBM was most of the block's time)
    float pre[C1M_PRE];
    auto request_rows = [&](int blk) {
        const int b = blk / nblk, p0 = (blk - b * nblk) * 128;
        const int plast = p0 + 127 < P ? p0 + 127 : P - 1;
        const int r_lo = p0 / F1p, r_hi = plast / F1p;
        const int t_base = 2 * (r_lo - 1) - 1, nrows = 2 * (r_hi - r_lo) + 3;
        const int tl = utt_meta ? utt_meta[b].frames : Tn;
#pragma unroll
        for (int q = 0; q < C1M_PRE; ++q) {
            const int i = tid + 256 * q;
            const int lr = i / FW, f = i - lr * FW - 1, t = t_base + lr;
            pre[q] = (i < nrows * FW && t >= 0 && t < tl && f >= 0 && f < F) ? x[((long long)b * Tn + t) * F + f] : 0.f;
        }
    };
    if ((int)blockIdx.x < B * nblk) request_rows(blockIdx.x);
    // the wave's 32 cells waiting in the pad: where they go and how many of them exist (0: nothing waits)
    long long pend_off = 0;
    int pend_cells = 0;
    auto flush = [&](int grp) {  // (LDS operations of one wave complete in order: these reads see the writes before them, later writes come after)
        if (pend_cells > 0) {
            typedef unsigned c1m_u32x4 __attribute__((ext_vector_type(4)));
            unsigned char* ob = out + pend_off + 256 * grp;
            c1m_u32x4 v[8];  // all eight pieces first, then the stores (a read, its wait and its store in turn: eight LDS round trips)
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx)
                v[sidx] = *reinterpret_cast<const c1m_u32x4*>(pad + (4 * sidx + (lane >> 4)) * C1M_ROWB + 16 * (lane & 15));
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) {
                const int cell = 4 * sidx + (lane >> 4), chunk = lane & 15;
                if (cell < pend_cells)
                    __builtin_nontemporal_store(v[sidx], reinterpret_cast<c1m_u32x4*>(ob + cell * CELLB + 16 * chunk));
            }
        }
    };
    for (int blk = blockIdx.x; blk < B * nblk; blk += gridDim.x) {
        const int b = blk / nblk, p0 = (blk - b * nblk) * 128;
        const int plast = p0 + 127 < P ? p0 + 127 : P - 1;
        const int r_lo = p0 / F1p, r_hi = plast / F1p;      // bordered image rows of the block
        const int t_base = 2 * (r_lo - 1) - 1;               // first input row that can be needed
        const int nrows = 2 * (r_hi - r_lo) + 3;
        const int tl = utt_meta ? utt_meta[b].frames : Tn;   // merged pass: the utterance's own batch is `tl` frames long
        const int t1_own = utt_meta ? (tl - 1) / 2 + 1 : T1;
        __syncthreads();  // the previous block's readers are done with xs
#pragma unroll
        for (int q = 0; q < C1M_PRE; ++q)
            if (tid + 256 * q < nrows * FW) xs[tid + 256 * q] = pre[q];
        __syncthreads();
        if (blk + (int)gridDim.x < B * nblk) request_rows(blk + gridDim.x);
        // ---- this wave's 32 cells
        const int p = p0 + 32 * wave + l31;
        const int r = p / F1p, t1 = r - 1, f1 = p - r * F1p - 1;
        const bool cell_in = p < P && t1 >= 0 && t1 < t1_own && f1 >= 0 && f1 < F1;
        const int base = (2 * t1 - 1 - t_base) * FW + 2 * f1;  // input row 2 t1 - 1 + kh at staged row .. + kh, column f = 2 f1 - 1 + kw at index f + 1
        bf16x8 xf, xl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (cell_in && toff[j] >= 0) v = xs[base + toff[j]];
            if (half == 1 && j == 1) v = cell_in ? 1.f : 0.f;  // k = 9: the bias rides on a constant one (a border cell stays 0)
            xf[j] = (bf16)v;
            xl[j] = (bf16)(v - (float)xf[j]);
        }
        // The pad's content leaves one phase late: the last group of block k is stored at the top of block k + 1, in front of its
        // MFMAs.  hipcc waits for a block's row requests with vmcnt(0) (loads and stores share the counter), i.e. for every store
        // issued before that point as well; issued here the stores have a block's arithmetic to land before that wait.
        // (Measured alone on the bf16 image of ten batches, 3.45 GB: 0.63-0.71 ms by box; without its stores 0.38, without its row
        // requests 0.72, a plain fill of the buffer 0.50 - arithmetic and stores overlap only in part; `tools/conv1_time.py`.)
        flush(OUT8 ? 0 : 1);
        const bool have = p0 + 32 * wave < P;  // (wave-uniform: the wave has at least one cell)
        if (have) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
                acc = CN_MFMA16(wl[nt], xf, acc, 0, 0, 0);
                acc = CN_MFMA16(wf[nt], xl, acc, 0, 0, 0);
                acc = CN_MFMA16(wf[nt], xf, acc, 0, 0, 0);
                // registers 0..15 of lane half h = channels 32 nt + 16 h + (0..15) of cell l31
                if constexpr (OUT8) {  // ReLU + saturation, e4m3, one 16-byte piece
                    unsigned w[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float v0 = __builtin_amdgcn_fmed3f(acc[4 * g + 0], 0.f, CN_FP8_MAX), v1 = __builtin_amdgcn_fmed3f(acc[4 * g + 1], 0.f, CN_FP8_MAX);
                        const float v2 = __builtin_amdgcn_fmed3f(acc[4 * g + 2], 0.f, CN_FP8_MAX), v3 = __builtin_amdgcn_fmed3f(acc[4 * g + 3], 0.f, CN_FP8_MAX);
                        unsigned d = 0;
                        d = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, d, false);
                        w[g] = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, d, true);
                    }
                    *reinterpret_cast<uint4*>(pad + l31 * C1M_ROWB + 32 * nt + 16 * half) = make_uint4(w[0], w[1], w[2], w[3]);
                } else {  // ReLU, bf16: two 16-byte pieces at 64 (nt & 3) + 32 half of the pad row
                    bf16x8 o0, o1;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        o0[q] = (bf16)fmaxf(acc[q], 0.f);
                        o1[q] = (bf16)fmaxf(acc[8 + q], 0.f);
                    }
                    unsigned char* pr = pad + l31 * C1M_ROWB + 64 * (nt & 3) + 32 * half;
                    *reinterpret_cast<bf16x8*>(pr) = o0;
                    *reinterpret_cast<bf16x8*>(pr + 16) = o1;
                    if (nt == 3) {  // the first half of the cells' rows leaves in the middle of the block
                        pend_off = ((long long)b * P + p0 + 32 * wave) * CELLB;
                        pend_cells = P - (p0 + 32 * wave);
                        flush(0);
                    }
                }
            }
        }
        pend_off = ((long long)b * P + p0 + 32 * wave) * CELLB;
        pend_cells = have ? P - (p0 + 32 * wave) : 0;
    }
    flush(OUT8 ? 0 : 1);
}

static inline bool conv1_mfma_applies(int C, int F1) {  // (F1 + 2 >= 32: a block of 128 cells spans at most 5 image rows = 11 input rows)
    return C == 256 && F1 + 2 >= 32 && F1 <= 63;        // (weight fragments and pad rows are laid out for 256 channels; F <= 126: C1M_PRE)
}
static inline int launch_conv1_mfma(bool out8, const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1,
                                    int F1, float scale, hipStream_t s, const UttMeta* utt_meta) {
    const long long nb = (long long)B * (((long long)(T1 + 2) * (F1 + 2) + 127) / 128);
    const unsigned grid = (unsigned)(nb < 256 * 4 ? (nb < 1 ? 1 : nb) : 256 * 4);  // 4 workgroups per CU (38 KiB of LDS each), grid-stride
    const size_t lds = (((size_t)C1M_MAXROWS * (F + 2) * 4 + 15) & ~(size_t)15) + 4 * (size_t)C1M_PAD;
    if (out8)
        hipLaunchKernelGGL(conv1_mfma_kernel<true>, dim3(grid), dim3(256), lds, s, x, w9c, bias, (unsigned char*)out, B, T, F, T1, F1, scale, utt_meta);
    else
        hipLaunchKernelGGL(conv1_mfma_kernel<false>, dim3(grid), dim3(256), lds, s, x, w9c, bias, (unsigned char*)out, B, T, F, T1, F1, 1.f, utt_meta);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// the bf16 engine's bordered image on the matrix cores (conv2's LDS-DMA kernel reads it): every cell of the border is written
int launch_conv1_bordered_bf16(const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1, int F1, int C,
                               hipStream_t s, const UttMeta* utt_meta) {
    if (!conv1_mfma_applies(C, F1)) {
        cn_set_error("conv1 (matrix-core form): 256 channels and at least 30 feature columns after the stride");
        return -1;
    }
    return launch_conv1_mfma(false, x, w9c, bias, out, B, T, F, T1, F1, 1.f, s, utt_meta);
}
// (A/B on the benchmark, `profiles/r05g_conv1_mfma_ab.txt`: the bf16 image is bound by its 3.4 GB of writes either way - 0.077 ms per
// batch against the VALU kernel's 0.080, end to end a tie - so the engine keeps the VALU kernel, whose sums are the fp32 FMA chain's;
// CASSNAT_CONV1_MFMA=1 selects this form - an experiment switch: the capture path keeps the VALU image, so with it the tests that hold
// the product path against captures to the last bit, and the bf16 beam-search report gate tuned on the VALU image's near-ties, fail:
// 3 of 359)
bool conv1_bordered_bf16_applies(int C, int F1) {
    static const bool on = cn_exp_env("CASSNAT_CONV1_MFMA") != nullptr;
    return on && conv1_mfma_applies(C, F1);
}

// the fp8 engine's image for conv2's e4m3 form (launch_conv2_f8): bordered, one byte per element at `scale` (a power of two)
int launch_conv1_f8(const float* x, const float* w9c, const float* bias, void* out8, int B, int T, int F, int T1, int F1, int C,
                    int halo, float scale, hipStream_t s, const UttMeta* utt_meta) {
    if (C % 8 != 0 || (256 % (C / 8)) != 0 || (halo != 1 && halo != 2)) {
        cn_set_error("conv1 (e4m3): channel count must be a multiple of 8 with C/8 dividing 256; the image is always bordered");
        return -1;
    }
    static const bool no_mfma = cn_exp_env("CASSNAT_CONV1_F8_VALU") != nullptr;
    if (conv1_mfma_applies(C, F1) && !no_mfma) return launch_conv1_mfma(true, x, w9c, bias, out8, B, T, F, T1, F1, scale, s, utt_meta);
    long long blocks = (long long)B * (T1 + 2);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)2 * 3 * (F + 2) * sizeof(float);
    hipLaunchKernelGGL((conv1_kernel<fp8_t>), dim3((unsigned)blocks), dim3(256), lds, s, x, w9c, bias, (fp8_t*)out8, B, T, F, T1, F1,
                       C, halo, 0ll, utt_meta, scale);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
