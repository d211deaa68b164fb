// Fused position-wise feed-forward sublayer of the split-bf16 ("bf16x3") engine, gfx950, d_model = 256:
//     x <- x + W2 . relu(W1 . LN(x) + b1) + b2          [and optionally  xn_next <- LN_next(x), split-bf16]
// Same sublayer and the same structure as fused.hip (SublayerConnection(LayerNorm -> PositionwiseFeedForward): src/models/
// modules/utils.py:23-32, positionff.py:15-16, norm.py:15-18), in the parity-grade precision: every operand is a (bf16 hi,
// bf16 lo) pair and every product three MFMAs (lo.hi + hi.lo + hi.hi, fp32 accumulation; common.h split_t).  The generic
// path spends two GEMM launches and a 2 x M x d_ff x 4-byte round trip of the hidden activations per layer on this; here
// they never leave registers:
//   * LN(x) of the workgroup's 64 rows is written once to LDS as B-operand fragments, a hi plane and a lo plane;
//   * every wave owns d_ff / 4 hidden units and streams ITS weight fragments (hi and lo, pre-tiled at pack time in
//     consumption order: 64 KiB per 32 hidden units) straight from global memory (L2) into four rotating register sets,
//     three groups of four 1-KiB fragments in flight under the MFMAs of the current one; the stream is read once and by
//     one wave, so there is no LDS hop and no barrier in the main loop;
//   * X^T[f][m] = W1 . xn^T is computed swapped, so its accumulator (after bias + ReLU and the hi / lo split) is directly the
//     B operand of out^T[n][m] += W2[n][f] X[f][m];
//   * the four waves' out^T partials are summed through LDS, fused with b2, the residual and the next LayerNorm.
// Per streamed byte this precision does 1.5x the MFMAs of the bf16 kernel, which moves the kernel from L2-stream-bound
// towards MFMA-bound: 64 rows x 12.6 MFLOP per row x 3 = 2.4 GFLOP of MFMA work per 4 MiB of stream.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "proj_x3_phase.h"

struct FfnX3Params {
    float* x;             // [M][256] residual stream, in place
    const float* ln_a;
    const float* ln_b;
    const uint4* wst;     // [dff/32][64][64] fragments: per hidden tile 8 W1 groups then 8 W2 groups of (hi, lo, hi, lo)
    const float* b1;      // [dff]
    const float* b2;      // [256]
    const float* nln_a;   // next LayerNorm (may be null)
    const float* nln_b;
    unsigned char* xn_out;  // [M][256] split-bf16, written when nln_a != null
    int M, dff;
    float eps;
    int rotate;
    // MIXF (template): the sublayer's two products in the mixed arithmetic (below); the four E8M0 scale bytes of the weights'
    // e4m3 parts {W1 q, W1 l, W2 q, W2 l} sit behind the stream (pack_ffn_x3)
    const int* mixq;
    // row-chain form (template PRO / TAIL; proj_x3_phase.h): the attention's output projection onto the residual stream in front
    // of the sublayer - x <- x + Wo . ctx + bo on the workgroup's rows, ctx split-bf16 [M][ldctx] - and the next attention's
    // projection (Q|K|V, split-bf16 output) of LN_next(x) behind it, on rows that never leave LDS
    const unsigned char* ctx;
    long long ldctx_bytes;
    PxPhase pro, tail;
#ifdef FX_STAMPS
    unsigned long long* stamps;  // [workgroup][wave][8]: kernel entry, main loop entry, main loop exit, kernel exit (s_memtime);
                                 // then the main loop's cycles by phase: W1 blocks, bias / ReLU (first half), W2 blocks
#endif
};
#ifdef FX_STAMPS  // measurement build only (tools/ffn_x3_stamps.py): where a workgroup's time goes
#define FX_STAMP(i) if (lane == 0) p.stamps[(blockIdx.x * 4 + wave) * 8 + (i)] = __builtin_readcyclecounter();
#define FX_PHASE(i) { const unsigned long long now_ = __builtin_readcyclecounter(); phase_[i] += now_ - last_; last_ = now_; }
#else
#define FX_STAMP(i)
#define FX_PHASE(i)
#endif

constexpr int FX_D = 256;
constexpr int FX_MT = 2;                          // 32-row M-tiles per workgroup
constexpr int FX_PLANE = FX_MT * 16384;           // bytes of one plane (hi or lo) of the xn fragments
constexpr int FX_MAX_DFF = 2048;
constexpr int FX_P_STRIDE = FX_D + 4;             // floats per partial row in LDS
constexpr int FX_LDS_MAIN = 2 * FX_PLANE + FX_MAX_DFF * 4;
constexpr int FX_LDS_PART = 4 * 32 * FX_P_STRIDE * 4;
constexpr int FX_LDS = FX_LDS_MAIN > FX_LDS_PART ? FX_LDS_MAIN : FX_LDS_PART;
static_assert(FX_LDS <= 160 * 1024, "LDS budget");
// TAIL form: the partial rows at a stride of exactly 256 floats with an XOR swizzle of the 16-byte chunks instead of the padding
// (131072 bytes), which leaves the CU's last 32 KiB for the first M-tile's LN_next fragments; the second M-tile's go to offset 0
// once the partials are dead
constexpr int FX_PART_TAIL = 4 * 32 * FX_D * 4;
constexpr int FX_LDS_TAIL = FX_PART_TAIL + 32768;
static_assert(FX_LDS_TAIL == 160 * 1024, "LDS budget (row-chain form)");

#define FX_MFMA(a, b, c) c = CN_MFMA16(a, b, c, 0, 0, 0)
// The kernel holds 18 accumulator tiles (16 of out^T, 2 of the hidden tile) = 288 registers.  The compiler selects every MFMA
// builtin with an AGPR destination (there are 256) and, left alone, parks the hidden tile there too and shuttles output tiles
// between the register files every iteration (96-160 v_accvgpr moves per hidden tile, plus the reads the VALU needs to see
// the hidden tile at all).  The W1 MFMAs are therefore written out with the hidden tile in ordinary VGPRs; the 16 output
// tiles then fill the AGPRs exactly and never move.  Written-out instructions are opaque to the hazard recogniser and to
// sched_group_barrier: the W1 blocks are ordered by sched_barrier fences instead, and the one read-after-MFMA the
// compiler cannot see (bias / ReLU on the finished tile) gets its own s_nop.
#define FX_MFMA_V(a, b, c) asm(CN_MFMA16_ASM "%0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FX_MFMA_V0(a, b, c) asm(CN_MFMA16_ASM "%0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b))

// MIXF: the two products of the sublayer in the arithmetic of conv2.hip's MIX form instead of three bf16 MFMAs per product - a value
// v is h = half(v) plus two e4m3 bytes at fixed power-of-two scales, l = e4m3((v - h) 2^11 S) and q = e4m3(v S), and
//     a b = h_a h_b + l_a q_b + q_a l_b :
// per 32 k one v_mfma_f32_32x32x16_f16 per k-step of 16 and ONE v_mfma_scale_f32_32x32x64_f8f6f4 whose K = 64 is the two cross
// terms side by side (k-block 0: q of the weights against l of the activations, k-block 1: l against q; the per-lane scale
// operands differ by 2^11 between the blocks, the products' scales agree) - 2 MFMA units where the split form spends 3.  The
// stream, the register sets, the LDS planes and the request / read pattern are the split form's: a weight group is
// [hi(k0)][e4m3 bytes 0-15][hi(k1)][e4m3 bytes 16-31] per lane, the second LDS plane holds the activations' e4m3 fragments.
// The hidden tile's e4m3 operand needs every lane's 32 hidden units: the half-waves trade their 16 (v_permlane32_swap).
template <bool PRO, bool TAIL, bool MIXF = false>
__global__ __launch_bounds__(256) void ffn_x3_kernel(FfnX3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xn_s = smem;                                          // [plane][mt][16 k-steps][64 lanes][16 B]
    float* b1_s = reinterpret_cast<float*>(smem + 2 * FX_PLANE);         // [dff]
    float* part = reinterpret_cast<float*>(smem);                        // [4][32][260] fp32, epilogue only (aliases everything)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * 32 * FX_MT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    FX_STAMP(0)
    if constexpr (PRO) {
        // x <- x + Wo . ctx + bo on this workgroup's 64 rows (wave w: output column tiles w and w + 4), written through to global
        // memory; the barrier's vmcnt(0) makes the rows visible to the waves of this workgroup that read them below (one L1)
        px_stage_rows<FX_MT>(p.ctx, p.ldctx_bytes, m0, p.M, smem, tid);
        __syncthreads();
        px_tiles<FX_MT, false, true>(p.pro, smem, smem + 32768, m0, 0, FX_D / 32, wave_u, lane);
        __syncthreads();
    }

    const int tiles_per_wave = p.dff / 32 / 4;
    const int ft0 = wave_u * tiles_per_wave;
    // every workgroup / wave walks its hidden tiles in a rotation of its own (the sum over tiles is order-free; the
    // workgroups then do not pull the same L2 lines at the same moment)
    const int rot = p.rotate ? (blockIdx.x * 7 + wave_u * 3) % tiles_per_wave : 0;
#define FX_TT(t) (((t) + rot) % tiles_per_wave)
    // the wave's stream as a buffer resource: a request is descriptor + scalar offset of the group + lane offset + immediate,
    // no vector arithmetic per request (flat addressing spent two VALU adds on each group)
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.wst)) + (size_t)ft0 * 64 * 1024, 0, tiles_per_wave * 64 * 1024, 0x00020000);
    const int lane_off = lane * 16;
#define FX_WFRAG(tile, g, j_)                                                                          \
    __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off + (j_) * 1024, ((tile) * 64 + 4 * (g)) * 1024, 0))

    // Weight register sets: group g (0..15 of a hidden tile: 8 of W1, 8 of W2; four 1-KiB fragments = hi, lo, hi, lo) lives
    // in set g & 7.  The stream is private to the wave and read once, so it goes global -> VGPR with no LDS hop (an LDS-DMA
    // ring was tried first: its requests cost 60-190 issue cycles apiece beside MFMAs, 4 per 12 MFMAs, and the kernel ran at
    // 0.37 us per block whatever the ring depth, 4 or 6 groups).  Block g computes on set g & 7 and, in its MFMA gaps,
    // requests group g + 7 into the set block g - 1 has just released: seven groups (28 KiB per wave) in flight.
    bf16x8 w0a, w0b, w0c, w0d, w1a, w1b, w1c, w1d, w2a, w2b, w2c, w2d, w3a, w3b, w3c, w3d;
    bf16x8 w4a, w4b, w4c, w4d, w5a, w5b, w5c, w5d, w6a, w6b, w6c, w6d, w7a, w7b, w7c, w7d;
#define FX_LDW(S_, tile, g)                                                                            \
    {                                                                                                  \
        w##S_##a = FX_WFRAG(tile, g, 0); w##S_##b = FX_WFRAG(tile, g, 1); w##S_##c = FX_WFRAG(tile, g, 2); w##S_##d = FX_WFRAG(tile, g, 3); \
    }
    // the workgroup's rows (wave w: rows w, w + 4, ...), all requested before anything else: they are needed first
    f32x4 xrow[32 * FX_MT / 4];
#pragma unroll
    for (int i = 0; i < 32 * FX_MT / 4; ++i) {
        int m = m0 + wave + 4 * i;
        if (m >= p.M) m = p.M - 1;
        xrow[i] = *reinterpret_cast<const f32x4*>(p.x + (long long)m * FX_D + 4 * lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    // prologue: the first four groups go in flight before the LayerNorm below, three more after it (the rows' registers
    // are free by then)
    {
        const int t0 = FX_TT(0);
        FX_LDW(0, t0, 0) FX_LDW(1, t0, 1) FX_LDW(2, t0, 2) FX_LDW(3, t0, 3)
    }
    // b1 -> LDS: eight floats per thread, requested now, stored after the LayerNorm (nothing waits for them in between)
    f32x4 b1lo = {0.f, 0.f, 0.f, 0.f}, b1hi = {0.f, 0.f, 0.f, 0.f};
    if (8 * tid < p.dff) {
        b1lo = *reinterpret_cast<const f32x4*>(p.b1 + 8 * tid);
        b1hi = *reinterpret_cast<const f32x4*>(p.b1 + 8 * tid + 4);
    }

    // ---- LayerNorm of the workgroup's rows -> hi / lo bf16 fragments in LDS
    {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_a + 4 * lane);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * lane);
        // element k = 4 lane + j of row r: fragment (mt = r >> 5, ks = k >> 4), lane slot 32 ((k >> 3) & 1) + (r & 31), byte 2 (k & 7)
        const int ks = lane >> 2, kh = (lane >> 1) & 1, kb = (lane & 1) * 8;
        // four rows at a time, their reductions side by side
#pragma unroll
        for (int i0 = 0; i0 < 32 * FX_MT / 4; i0 += 4) {
            float mean[4], ss[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) mean[q] = (xrow[i0 + q][0] + xrow[i0 + q][1]) + (xrow[i0 + q][2] + xrow[i0 + q][3]);
            wave_sum_n(mean);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mean[q] /= (float)FX_D;
                ss[q] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) ss[q] = fmaf(xrow[i0 + q][j] - mean[q], xrow[i0 + q][j] - mean[q], ss[q]);
            }
            wave_sum_n(ss);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = wave + 4 * (i0 + q);
                const float inv = 1.f / (sqrtf(ss[q] / (float)(FX_D - 1)) + p.eps);  // one division per row, not per element
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = g[j] * (xrow[i0 + q][j] - mean[q]) * inv + bb[j];
                unsigned char* dst = xn_s + (((r >> 5) * 16 + ks) * 64 + kh * 32 + (r & 31)) * 16 + kb;
                if constexpr (MIXF) {
                    typedef _Float16 fxh4 __attribute__((ext_vector_type(4)));
                    fxh4 hh;
                    float lo[4], qv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hh[j] = (_Float16)o[j];
                        lo[j] = __builtin_amdgcn_fmed3f((o[j] - (float)hh[j]) * 2048.f, -CN_FP8_MAX, CN_FP8_MAX);
                        qv[j] = __builtin_amdgcn_fmed3f(o[j], -CN_FP8_MAX, CN_FP8_MAX);
                    }
                    *reinterpret_cast<fxh4*>(dst) = hh;
                    unsigned l8 = 0, q8 = 0;
                    l8 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l8, false);
                    l8 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l8, true);
                    q8 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], q8, false);
                    q8 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], q8, true);
                    // k = 4 lane + j: 32-k block lane >> 3, byte 4 (lane & 7) + j of the lane's 32 = sub-fragment (lane & 7) >> 2,
                    // byte 4 (lane & 3); k-block 0 (lane slots 0..31) takes the l bytes, k-block 1 the q bytes
                    unsigned char* f8 = xn_s + FX_PLANE + ((r >> 5) * 16 + 2 * (lane >> 3) + ((lane & 7) >> 2)) * 1024 + 4 * (lane & 3);
                    *reinterpret_cast<unsigned*>(f8 + (r & 31) * 16) = l8;
                    *reinterpret_cast<unsigned*>(f8 + (32 + (r & 31)) * 16) = q8;
                } else {
                    bf16x4 hi, lo;
                    cn_split4(o, hi, lo);
                    *reinterpret_cast<bf16x4*>(dst) = hi;
                    *reinterpret_cast<bf16x4*>(dst + FX_PLANE) = lo;
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const int t0 = FX_TT(0);
        FX_LDW(4, t0, 4) FX_LDW(5, t0, 5) FX_LDW(6, t0, 6)
    }
    if (8 * tid < p.dff) {
        *reinterpret_cast<f32x4*>(b1_s + 8 * tid) = b1lo;
        *reinterpret_cast<f32x4*>(b1_s + 8 * tid + 4) = b1hi;
    }
    __syncthreads();

    f32x16 acc[FX_MT][8];
#pragma unroll
    for (int mt = 0; mt < FX_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // Activation register sets x0 / x1 (eight fragments each: hi / lo x two k-steps x two M-tiles): W1 block a uses set a & 1;
    // the LDS reads of block a + 1 ride between block a's MFMAs.
    bf16x8 x0a, x0b, x0c, x0d, x0e, x0f, x0g, x0h, x1a, x1b, x1c, x1d, x1e, x1f, x1g, x1h;
    const unsigned char* xfrag = xn_s + lane * 16;
#define FX_XFRAG(off) (*reinterpret_cast<const bf16x8*>(xfrag + (off)))
#define FX_LDX(S_, AI_)                                                                                  \
    x##S_##a = FX_XFRAG((2 * (AI_) + 0) * 1024);             x##S_##b = FX_XFRAG(FX_PLANE + (2 * (AI_) + 0) * 1024);          \
    x##S_##c = FX_XFRAG((2 * (AI_) + 1) * 1024);             x##S_##d = FX_XFRAG(FX_PLANE + (2 * (AI_) + 1) * 1024);          \
    x##S_##e = FX_XFRAG(16384 + (2 * (AI_) + 0) * 1024);     x##S_##f = FX_XFRAG(FX_PLANE + 16384 + (2 * (AI_) + 0) * 1024);  \
    x##S_##g = FX_XFRAG(16384 + (2 * (AI_) + 1) * 1024);     x##S_##h = FX_XFRAG(FX_PLANE + 16384 + (2 * (AI_) + 1) * 1024);
    // the hi and lo fragment of k-step 2 a + q, M-tile at byte offset mo
#if defined(FX_STAMPS) && defined(FX_EXP_NO_XREAD)  // timing experiment: the W1 blocks without their activation reads
#define FX_LDX2(S_, F0_, F1_, AI_, q, mo)
#else
#define FX_LDX2(S_, F0_, F1_, AI_, q, mo)                                                              \
    x##S_##F0_ = FX_XFRAG((mo) + (2 * (AI_) + (q)) * 1024); x##S_##F1_ = FX_XFRAG(FX_PLANE + (mo) + (2 * (AI_) + (q)) * 1024);
#endif
    // issue order inside a W2 block: three MFMAs, one weight request
#define FX_PIN_B()                                                                                     \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);    \
    }
    // A W1 block on register sets (W_, X_): fragments a..d = W1 hi(k0), lo(k0), hi(k1), lo(k1); X a..h = M-tile 0 hi(k0),
    // lo(k0), hi(k1), lo(k1), M-tile 1 hi(k0), lo(k0), hi(k1), lo(k1).  Per accumulator the order is lo.hi, hi.lo, hi.hi (k0),
    // then the same for k1; the two accumulators alternate so that no MFMA waits for the one just issued.  Four fenced
    // quarters of three MFMAs, each with one weight request (Q0..Q3: group a + 3, fragment by fragment) and two of the
    // next block's activation reads (R0..R3).  M0, M1: the first two MFMAs (FX_MFMA_V0 in block 0: they start the tile).
#define FX_FENCE() __builtin_amdgcn_sched_barrier(0);
#define FX_MFMA_A(W_, X_, M0, M1, Q0, Q1, Q2, Q3, R0, R1, R2, R3)                                      \
    M0(W_##b, X_##a, xh[0]); M1(W_##b, X_##e, xh[1]); FX_MFMA_V(W_##a, X_##b, xh[0]); Q0 R0 FX_FENCE()  \
    FX_MFMA_V(W_##a, X_##f, xh[1]); FX_MFMA_V(W_##a, X_##a, xh[0]); FX_MFMA_V(W_##a, X_##e, xh[1]); Q1 R1 FX_FENCE() \
    FX_MFMA_V(W_##d, X_##c, xh[0]); FX_MFMA_V(W_##d, X_##g, xh[1]); FX_MFMA_V(W_##c, X_##d, xh[0]); Q2 R2 FX_FENCE() \
    FX_MFMA_V(W_##c, X_##h, xh[1]); FX_MFMA_V(W_##c, X_##c, xh[0]); FX_MFMA_V(W_##c, X_##g, xh[1]); Q3 R3 FX_FENCE()
    // MFMAs of W2 block b: s = b >> 2, output tiles nt0 = 2 (b & 3), nt0 + 1; fragments a..d = W2 hi(nt0), lo(nt0), hi(nt0+1),
    // lo(nt0+1); four accumulators in rotation, each in the order lo.hi, hi.lo, hi.hi
#define FX_MFMA_B(BI_, W_)                                                                             \
    FX_MFMA(W_##b, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3)]);                                       \
    FX_MFMA(W_##d, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3) + 1]);                                   \
    FX_MFMA(W_##b, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3)]);                                       \
    FX_MFMA(W_##d, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3) + 1]);                                   \
    FX_MFMA(W_##a, pbl[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3)]);                                       \
    FX_MFMA(W_##c, pbl[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3) + 1]);                                   \
    FX_MFMA(W_##a, pbl[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3)]);                                       \
    FX_MFMA(W_##c, pbl[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3) + 1]);                                   \
    FX_MFMA(W_##a, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3)]);                                       \
    FX_MFMA(W_##c, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3) + 1]);                                   \
    FX_MFMA(W_##a, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3)]);                                       \
    FX_MFMA(W_##c, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3) + 1]);
    // W1 block a (weights in set WS, activations in XS): request group a + 3 into set WN, read block a + 1's activations into XN
#if defined(FX_STAMPS) && defined(FX_EXP_NO_WLOAD)  // timing experiment: the main loop without its weight requests
#define FX_LDW1(S_, F_, tile, g, j_)
#define FX_LDW_LOOP(S_, tile, g)
#else
#define FX_LDW_LOOP(S_, tile, g) FX_LDW(S_, tile, g)
#define FX_LDW1(S_, F_, tile, g, j_) w##S_##F_ = FX_WFRAG(tile, g, j_);
#endif
#define FX_BLOCK_A(AI_, WS, XS, WN, XN, M0, M1)                                                          \
    FX_MFMA_A(w##WS, x##XS, M0, M1,                                                                    \
              FX_LDW1(WN, a, cur, (AI_) + 7, 0), FX_LDW1(WN, b, cur, (AI_) + 7, 1), FX_LDW1(WN, c, cur, (AI_) + 7, 2), FX_LDW1(WN, d, cur, (AI_) + 7, 3), \
              FX_LDX2(XN, a, b, (AI_) + 1, 0, 0), FX_LDX2(XN, c, d, (AI_) + 1, 1, 0), FX_LDX2(XN, e, f, (AI_) + 1, 0, 16384), FX_LDX2(XN, g, h, (AI_) + 1, 1, 16384))
#define FX_BLOCK_A7(WS, XS, WN)                                                                        \
    FX_MFMA_A(w##WS, x##XS, FX_MFMA_V, FX_MFMA_V,                                                      \
              FX_LDW1(WN, a, cur, 14, 0), FX_LDW1(WN, b, cur, 14, 1), FX_LDW1(WN, c, cur, 14, 2), FX_LDW1(WN, d, cur, 14, 3), , , , )
    // W2 block b (group 8 + b): request group 15 + b (the last of this tile, or 0..6 of the next) into set WN
#define FX_BLOCK_B(b, WS, WN, TILE, G)                                                                 \
    FX_LDW_LOOP(WN, TILE, G) FX_MFMA_B(b, w##WS) FX_PIN_B()                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // bias + ReLU on hidden units f = 32 tile + 8 g + 4 half + e, e = 0..3 (accumulator registers 4 g + e), split into the
    // hi / lo B operands of the W2 blocks: g = 0, 1 feed blocks 0..3 (s = 0), g = 2, 3 blocks 4..7
#define FX_RELU_PACK(tile, g)                                                                          \
    {                                                                                                  \
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1_s + 32 * (ft0 + (tile)) + 4 * half + 8 * (g)); \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int mt = 0; mt < FX_MT; ++mt) { \
            const float h_ = fmaxf(xh[mt][4 * (g) + e] + bv[e], 0.f);                                  \
            const bf16 hh_ = (bf16)h_;                                                                 \
            pbh[mt][(g) >> 1][4 * ((g) & 1) + e] = hh_;                                                \
            pbl[mt][(g) >> 1][4 * ((g) & 1) + e] = (bf16)(h_ - (float)hh_);                            \
        }                                                                                              \
    }
    // a W2 block that also carries one quarter of the bias / ReLU / split work (needed from block 4 on) between its MFMAs
#define FX_PIN_BV()                                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                              \
    }
#define FX_BLOCK_BV(b, WS, WN, TILE, G, RG)                                                            \
    FX_LDW_LOOP(WN, TILE, G) FX_RELU_PACK(cur, RG) FX_MFMA_B(b, w##WS) FX_PIN_BV()                          \
    __builtin_amdgcn_sched_barrier(0);
    // ---- MIXF forms of the blocks.  Register sets as above with another meaning of the fragments: weights a = hi(k0), b = e4m3
    // bytes 0-15, c = hi(k1), d = e4m3 bytes 16-31 (W2 blocks: a = hi(s = 0), c = hi(s = 1) of output tile b); activations
    // a / e = hi(k0) of M-tile 0 / 1, c / g = hi(k1), (b, d) / (f, h) = the e4m3 fragment
    typedef int fxi4 __attribute__((ext_vector_type(4)));
    typedef int fxi8 __attribute__((ext_vector_type(8)));
    typedef _Float16 fxh8 __attribute__((ext_vector_type(8)));
#define FXM_J(lo_, hi_) __builtin_shufflevector(__builtin_bit_cast(fxi4, lo_), __builtin_bit_cast(fxi4, hi_), 0, 1, 2, 3, 4, 5, 6, 7)
#define FXM_H(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fxh8, a), __builtin_bit_cast(fxh8, b), c, 0, 0, 0)
#define FXM_H_V(a, b, c) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FXM_H_V0(a, b, c) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b))
#define FXM_8_V(a8, b8, c, sa, sb) asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a8), "v"(b8), "v"(sa), "v"(sb))
#define FXM_8(a8, b8, c, sa, sb) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 0, 0, 0, sa, 0, sb)
#define FXM_MFMA_A(W_, X_, M0, M1, Q0, Q1, Q2, Q3, R0, R1, R2, R3)                                     \
    M0(W_##a, X_##a, xh[0]); M1(W_##a, X_##e, xh[1]); Q0 R0 FX_FENCE()                                 \
    FXM_H_V(W_##c, X_##c, xh[0]); FXM_H_V(W_##c, X_##g, xh[1]); Q1 R1 FX_FENCE()                       \
    { const fxi8 a8_ = FXM_J(W_##b, W_##d); const fxi8 b8_ = FXM_J(X_##b, X_##d); FXM_8_V(a8_, b8_, xh[0], sa1, sb1); } Q2 R2 FX_FENCE() \
    { const fxi8 a8_ = FXM_J(W_##b, W_##d); const fxi8 b8_ = FXM_J(X_##f, X_##h); FXM_8_V(a8_, b8_, xh[1], sa1, sb1); } Q3 R3 FX_FENCE()
#define FXM_BLOCK_A(AI_, WS, XS, WN, XN, M0, M1)                                                         \
    FXM_MFMA_A(w##WS, x##XS, M0, M1,                                                                   \
               FX_LDW1(WN, a, cur, (AI_) + 7, 0), FX_LDW1(WN, b, cur, (AI_) + 7, 1), FX_LDW1(WN, c, cur, (AI_) + 7, 2), FX_LDW1(WN, d, cur, (AI_) + 7, 3), \
               FX_LDX2(XN, a, b, (AI_) + 1, 0, 0), FX_LDX2(XN, c, d, (AI_) + 1, 1, 0), FX_LDX2(XN, e, f, (AI_) + 1, 0, 16384), FX_LDX2(XN, g, h, (AI_) + 1, 1, 16384))
#define FXM_BLOCK_A7(WS, XS, WN)                                                                       \
    FXM_MFMA_A(w##WS, x##XS, FXM_H_V, FXM_H_V,                                                         \
               FX_LDW1(WN, a, cur, 14, 0), FX_LDW1(WN, b, cur, 14, 1), FX_LDW1(WN, c, cur, 14, 2), FX_LDW1(WN, d, cur, 14, 3), , , , )
    // W2 block b = output tile b: two half-precision k-steps and the e4m3 K = 64 on the two M-tiles' accumulators in turn
#define FXM_MFMA_B(BI_, W_)                                                                            \
    FXM_H(W_##a, pbh[0][0], acc[0][BI_]); FXM_H(W_##a, pbh[1][0], acc[1][BI_]);                        \
    FXM_H(W_##c, pbh[0][1], acc[0][BI_]); FXM_H(W_##c, pbh[1][1], acc[1][BI_]);                        \
    { const fxi8 a8_ = FXM_J(W_##b, W_##d); FXM_8(a8_, pb8[0], acc[0][BI_], sa2, sb2); FXM_8(a8_, pb8[1], acc[1][BI_], sa2, sb2); }
#define FXM_PIN_B()                                                                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);    \
    }
#define FXM_BLOCK_B(b, WS, WN, TILE, G)                                                                \
    FX_LDW_LOOP(WN, TILE, G) FXM_MFMA_B(b, w##WS) FXM_PIN_B()                                          \
    __builtin_amdgcn_sched_barrier(0);
    // bias + ReLU on the hidden tile, then its three operand forms: half-precision fragments (as the split form's hi halves) and
    // the e4m3 bytes - a lane has 16 of its column's 32 hidden units (accumulator register r: unit (r & 3) + 8 (r >> 2) + 4 half);
    // L = its l bytes, Q = its q bytes (dword g = registers 4 g .. 4 g + 3); v_permlane32_swap(L, Q) leaves the lower half-wave
    // with (L own, L partner) = the l bytes of all 32 units in the order [half 0's 16 | half 1's 16] - k-block 0 - and the upper
    // one with (Q partner, Q own) = the q bytes in the same order - k-block 1
#define FXM_RELU_PACK(tile)                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < FX_MT; ++mt) {                                              \
        unsigned L_[4], Q_[4];                                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b1_s + 32 * (ft0 + (tile)) + 4 * half + 8 * g); \
            float lo_[4], qv_[4];                                                                      \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
                const float h_ = fmaxf(xh[mt][4 * g + e] + bv[e], 0.f);                                \
                const _Float16 hh_ = (_Float16)h_;                                                     \
                ph16[mt][g >> 1][4 * (g & 1) + e] = hh_;                                               \
                lo_[e] = __builtin_amdgcn_fmed3f((h_ - (float)hh_) * 2048.f, -CN_FP8_MAX, CN_FP8_MAX); \
                qv_[e] = fminf(h_, CN_FP8_MAX);                                                        \
            }                                                                                          \
            unsigned l8_ = 0, q8_ = 0;                                                                 \
            l8_ = __builtin_amdgcn_cvt_pk_fp8_f32(lo_[0], lo_[1], l8_, false);                         \
            l8_ = __builtin_amdgcn_cvt_pk_fp8_f32(lo_[2], lo_[3], l8_, true);                          \
            q8_ = __builtin_amdgcn_cvt_pk_fp8_f32(qv_[0], qv_[1], q8_, false);                         \
            q8_ = __builtin_amdgcn_cvt_pk_fp8_f32(qv_[2], qv_[3], q8_, true);                          \
            const auto sw_ = __builtin_amdgcn_permlane32_swap(l8_, q8_, false, false);                 \
            L_[g] = sw_[0];                                                                            \
            Q_[g] = sw_[1];                                                                            \
        }                                                                                              \
        pb8[mt] = fxi8{(int)L_[0], (int)L_[1], (int)L_[2], (int)L_[3], (int)Q_[0], (int)Q_[1], (int)Q_[2], (int)Q_[3]}; \
        pbh[mt][0] = __builtin_bit_cast(bf16x8, ph16[mt][0]);                                          \
        pbh[mt][1] = __builtin_bit_cast(bf16x8, ph16[mt][1]);                                          \
    }
    f32x16 xh[FX_MT];
    bf16x8 pbh[FX_MT][2], pbl[FX_MT][2];
    FX_LDX(0, 0)
    FX_STAMP(1)
#ifdef FX_STAMPS
    unsigned long long phase_[3] = {0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    if constexpr (MIXF) {
        fxh8 ph16[FX_MT][2];
        fxi8 pb8[FX_MT];
        // per-lane E8M0 scale bytes (a lane's 32 k are one scale block): weights {q, l} by k-block, activations {l at 2^11, q at 1}
        const int sa1 = half ? p.mixq[1] : p.mixq[0], sa2 = half ? p.mixq[3] : p.mixq[2];
        const int sb1 = half ? 127 : 127 - 11, sb2 = sb1;
        for (int t = 0; t < tiles_per_wave; ++t) {
            const int cur = FX_TT(t);
            const int nxt = FX_TT(t + 1 < tiles_per_wave ? t + 1 : t);
            FXM_BLOCK_A(0, 0, 0, 7, 1, FXM_H_V0, FXM_H_V0)
            FXM_BLOCK_A(1, 1, 1, 0, 0, FXM_H_V, FXM_H_V)
            FXM_BLOCK_A(2, 2, 0, 1, 1, FXM_H_V, FXM_H_V)
            FXM_BLOCK_A(3, 3, 1, 2, 0, FXM_H_V, FXM_H_V)
            FXM_BLOCK_A(4, 4, 0, 3, 1, FXM_H_V, FXM_H_V)
            FXM_BLOCK_A(5, 5, 1, 4, 0, FXM_H_V, FXM_H_V)
            FXM_BLOCK_A(6, 6, 0, 5, 1, FXM_H_V, FXM_H_V)
            FXM_BLOCK_A7(7, 1, 6)
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(xh[0]), "+v"(xh[1]));  // MFMA results (16 passes) -> VALU
            FX_PHASE(0)
            FXM_RELU_PACK(cur)
            __builtin_amdgcn_sched_barrier(0);
            FX_PHASE(1)
            FXM_BLOCK_B(0, 0, 7, cur, 15)
            FXM_BLOCK_B(1, 1, 0, nxt, 0)
            FXM_BLOCK_B(2, 2, 1, nxt, 1)
            FXM_BLOCK_B(3, 3, 2, nxt, 2)
            FXM_BLOCK_B(4, 4, 3, nxt, 3)
            FXM_BLOCK_B(5, 5, 4, nxt, 4)
            FXM_BLOCK_B(6, 6, 5, nxt, 5)
            FX_LDX(0, 0)
            FXM_BLOCK_B(7, 7, 6, nxt, 6)
            FX_PHASE(2)
        }
    } else
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int cur = FX_TT(t);
        const int nxt = FX_TT(t + 1 < tiles_per_wave ? t + 1 : t);  // after the last tile: seven groups requested again, unused
        FX_BLOCK_A(0, 0, 0, 7, 1, FX_MFMA_V0, FX_MFMA_V0)
        FX_BLOCK_A(1, 1, 1, 0, 0, FX_MFMA_V, FX_MFMA_V)
        FX_BLOCK_A(2, 2, 0, 1, 1, FX_MFMA_V, FX_MFMA_V)
        FX_BLOCK_A(3, 3, 1, 2, 0, FX_MFMA_V, FX_MFMA_V)
        FX_BLOCK_A(4, 4, 0, 3, 1, FX_MFMA_V, FX_MFMA_V)
        FX_BLOCK_A(5, 5, 1, 4, 0, FX_MFMA_V, FX_MFMA_V)
        FX_BLOCK_A(6, 6, 0, 5, 1, FX_MFMA_V, FX_MFMA_V)
        FX_BLOCK_A7(7, 1, 6)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(xh[0]), "+v"(xh[1]));  // MFMA results -> VALU
        FX_PHASE(0)
        FX_RELU_PACK(cur, 0)
        FX_RELU_PACK(cur, 1)
        __builtin_amdgcn_sched_barrier(0);
        FX_PHASE(1)
        FX_BLOCK_BV(0, 0, 7, cur, 15, 2)
        FX_BLOCK_BV(1, 1, 0, nxt, 0, 3)
        FX_BLOCK_B(2, 2, 1, nxt, 1)
        FX_BLOCK_B(3, 3, 2, nxt, 2)
        FX_BLOCK_B(4, 4, 3, nxt, 3)
        FX_BLOCK_B(5, 5, 4, nxt, 4)
        FX_BLOCK_B(6, 6, 5, nxt, 5)
        FX_LDX(0, 0)
        FX_BLOCK_B(7, 7, 6, nxt, 6)
        FX_PHASE(2)
    }
#ifdef FX_STAMPS
    if (lane == 0)
        for (int i = 0; i < 3; ++i) p.stamps[(blockIdx.x * 4 + wave) * 8 + 4 + i] = phase_[i];
#endif
    FX_STAMP(2)
#undef FXM_BLOCK_A
#undef FXM_BLOCK_A7
#undef FXM_BLOCK_B
#undef FXM_MFMA_A
#undef FXM_MFMA_B
#undef FXM_RELU_PACK
#undef FX_BLOCK_A
#undef FX_BLOCK_A7
#undef FX_BLOCK_B
#undef FX_BLOCK_BV
#undef FX_MFMA_A
#undef FX_MFMA_B
#undef FX_RELU_PACK
#undef FX_LDW
#undef FX_LDX
#undef FX_TT

    // ---- cross-wave reduction of the out^T partials (one 32-row M-tile at a time), + b2 + residual, next LayerNorm
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(p.b2 + 4 * lane);
    f32x4 ng, nb;
    if (p.nln_a) {
        ng = *reinterpret_cast<const f32x4*>(p.nln_a + 4 * lane);
        nb = *reinterpret_cast<const f32x4*>(p.nln_b + 4 * lane);
    }
#pragma unroll
    for (int mt = 0; mt < FX_MT; ++mt) {
        // the residual rows of this round (wave w: rows w, w + 4, ... of the M-tile), requested before the partials go
        // through LDS; everything below is branch-free so that the eight rows' chains interleave, only the stores are guarded
        f32x4 xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m = m0 + 32 * mt + wave + 4 * i;
            if (m >= p.M) m = p.M - 1;
            xv[i] = *reinterpret_cast<const f32x4*>(p.x + (long long)m * FX_D + 4 * lane);
        }
        __syncthreads();  // xn fragments (mt == 0) or the previous round's partials are no longer read
        constexpr int PSTR = TAIL ? FX_D : FX_P_STRIDE;
        float* mine = part + (wave * 32 + l31) * PSTR;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[mt][nt][4 * g + e];
                const int ch = 8 * nt + 2 * g + half;  // 16-byte chunk of the row (TAIL: swizzled by the row's low bits)
                *reinterpret_cast<f32x4*>(mine + 4 * (TAIL ? ch ^ (l31 & 7) : ch)) = o;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + 4 * i;
            const int m = m0 + 32 * mt + r;
            f32x4 v = xv[i];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(part + (w * 32 + r) * PSTR + 4 * (TAIL ? lane ^ (r & 7) : lane));
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += q[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += b2v[j];
            if (m < p.M) *reinterpret_cast<f32x4*>(p.x + (long long)m * FX_D + 4 * lane) = v;  // wave-uniform
            xv[i] = v;
        }
        // TAIL: LN_next(x) stays in LDS as the projection's fragments - M-tile 0 in the 32 KiB behind the partials, M-tile 1 at
        // offset 0, once every wave has read the last round's partials
        unsigned char* nfrag = smem + (mt == 0 ? FX_PART_TAIL : 0);
        if constexpr (TAIL) {
            if (mt == FX_MT - 1) __syncthreads();
        }
        if (p.nln_a) {
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 4) {
                float mean[4], ss[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) mean[q] = (xv[i0 + q][0] + xv[i0 + q][1]) + (xv[i0 + q][2] + xv[i0 + q][3]);
                wave_sum_n(mean);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mean[q] /= (float)FX_D;
                    ss[q] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ss[q] = fmaf(xv[i0 + q][j] - mean[q], xv[i0 + q][j] - mean[q], ss[q]);
                }
                wave_sum_n(ss);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + 32 * mt + wave + 4 * (i0 + q);
                    const float inv = 1.f / (sqrtf(ss[q] / (float)(FX_D - 1)) + p.eps);
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = ng[j] * (xv[i0 + q][j] - mean[q]) * inv + nb[j];
                    bf16x4 hi, lo;
                    cn_split4(o, hi, lo);
                    if constexpr (TAIL) {  // (rows past M hold the last row's values: their projections are never stored)
                        unsigned char* ob = nfrag + px_frag_off(wave + 4 * (i0 + q), 4 * lane);
                        *reinterpret_cast<bf16x4*>(ob) = hi;
                        *reinterpret_cast<bf16x4*>(ob + 1024) = lo;
                    } else if (m < p.M) {
                        unsigned char* ob = p.xn_out + (long long)m * FX_D * 4 + cn_split_off((size_t)(4 * lane));
                        *reinterpret_cast<bf16x4*>(ob) = hi;
                        *reinterpret_cast<bf16x4*>(ob + 64) = lo;
                    }
                }
            }
        }
    }
    if constexpr (TAIL) {
        // the next attention's projection of LN_next(x): wave w computes column tiles w, w + 4, ... (split-bf16 rows to global)
        __syncthreads();
        px_tiles<FX_MT, true, true>(p.tail, smem + FX_PART_TAIL, smem, m0, 0, p.tail.N >> 5, wave_u, lane);
    }
    FX_STAMP(3)
}

#ifdef FX_STAMPS
static unsigned long long* g_fx_stamps = nullptr;
// copies the stamps of the last launch (n workgroups x 4 waves x 8) to the host
extern "C" int cn_debug_ffn_x3_stamps(unsigned long long* out, int n_wg) {
    if (!g_fx_stamps || n_wg > 4096) return -1;
    return hipMemcpy(out, g_fx_stamps, (size_t)n_wg * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

bool ffn_x3_applies(int d, int dff) { return d == FX_D && dff % 128 == 0 && dff >= 128 && dff <= 2048 && !cn_exp_env("CASSNAT_NO_FFN_X3"); }

int launch_ffn_x3(const FfnX3Args& a, hipStream_t s) {
    if (!ffn_x3_applies(a.d, a.dff)) {
        cn_set_error("ffn_x3: needs d_model == 256 and d_ff % 128 == 0, d_ff <= 2048");
        return -1;
    }
    if (a.M <= 0) return 0;
    FfnX3Params p;
    p.x = a.x;
    p.ln_a = a.ln_a;
    p.ln_b = a.ln_b;
    p.wst = reinterpret_cast<const uint4*>(a.wst);
    p.b1 = a.b1;
    p.b2 = a.b2;
    p.nln_a = a.nln_a;
    p.nln_b = a.nln_b;
    p.xn_out = reinterpret_cast<unsigned char*>(a.xn_out);
    p.M = a.M;
    p.dff = a.dff;
    p.eps = a.eps;
    // (off by default: the rotation made a row's accumulation order depend on its workgroup and measured no gain)
    static const int rotate = cn_exp_env("CASSNAT_FFN_X3_ROTATE") ? atoi(cn_exp_env("CASSNAT_FFN_X3_ROTATE")) : 0;
    p.rotate = rotate;
#ifdef FX_STAMPS
    static unsigned long long* stamps_dev = nullptr;
    if (!stamps_dev) CN_HIP_CHECK(hipMalloc(&stamps_dev, 4096 * 32 * sizeof(unsigned long long)));
    if (cn_ceil_div(p.M, 32 * FX_MT) > 4096) return -1;
    p.stamps = stamps_dev;
    g_fx_stamps = stamps_dev;
#endif
    const bool pro = a.ctx != nullptr, tail = a.tail_p != nullptr;
    if (pro && (!a.wo_p || !a.bo || a.ldctx % 32 != 0)) {
        cn_set_error("ffn_x3: the output-projection form needs its packed weights, its bias and a 32-element ctx row stride");
        return -1;
    }
    if (tail && (!a.nln_a || !a.tail_b || !a.tail_out || a.tail_n < 128 || a.tail_n % 128 != 0 || a.tail_n > 1024 || a.ld_tail % 32 != 0)) {
        cn_set_error("ffn_x3: the tail-projection form needs the next LayerNorm, a bias, an output and 128 <= tail_n <= 1024, a multiple of 128");
        return -1;
    }
    if ((long long)a.M * FX_D * 4 >= (1ll << 31)) {
        cn_set_error("ffn_x3: residual stream beyond 2 GiB");
        return -1;
    }
    p.ctx = reinterpret_cast<const unsigned char*>(a.ctx);
    p.ldctx_bytes = (long long)a.ldctx * 4;
    p.pro = PxPhase{reinterpret_cast<const unsigned char*>(a.wo_p), a.bo, a.x, FX_D, a.x, FX_D, 1.f, a.M, FX_D};
    p.tail = PxPhase{reinterpret_cast<const unsigned char*>(a.tail_p), a.tail_b, a.tail_out, a.ld_tail, nullptr, 0, 1.f, a.M, a.tail_n};
    const dim3 grid(cn_ceil_div(p.M, 32 * FX_MT));
    p.mixq = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(a.wst) + (size_t)(a.dff / 32) * 64 * 1024);
#define FX_LAUNCH(PRO_, TAIL_, MIX_, LDS_)                                                                                 \
    {                                                                                                                      \
        static CnAttrOnce attr_once;                                                                                       \
        int attr_dev;                                                                                                      \
        if (attr_once.need(&attr_dev)) {                                                                                   \
            CN_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_x3_kernel<PRO_, TAIL_, MIX_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_)); \
            attr_once.mark(attr_dev);                                                                                      \
        }                                                                                                                  \
        hipLaunchKernelGGL((ffn_x3_kernel<PRO_, TAIL_, MIX_>), grid, dim3(256), LDS_, s, p);                                \
    }
    if (a.mix) {
        if (pro && tail) FX_LAUNCH(true, true, true, FX_LDS_TAIL)
        else if (pro) FX_LAUNCH(true, false, true, FX_LDS)
        else if (tail) FX_LAUNCH(false, true, true, FX_LDS_TAIL)
        else FX_LAUNCH(false, false, true, FX_LDS)
    } else {
        if (pro && tail) FX_LAUNCH(true, true, false, FX_LDS_TAIL)
        else if (pro) FX_LAUNCH(true, false, false, FX_LDS)
        else if (tail) FX_LAUNCH(false, true, false, FX_LDS_TAIL)
        else FX_LAUNCH(false, false, false, FX_LDS)
    }
#undef FX_LAUNCH
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- host-side packing: nn.Linear weights -> the per-tile fragment stream (hi and lo halves) ------------------------
static inline uint16_t fx_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline void fx_split(float v, uint16_t& hi, uint16_t& lo) {
    hi = fx_bf16_bits(v);
    uint32_t hb = (uint32_t)hi << 16;
    float hf;
    memcpy(&hf, &hb, 4);
    lo = fx_bf16_bits(v - hf);
}

// (+ 64 bytes behind the tiles: the mixed form's E8M0 scale bytes of the weights' e4m3 parts, four ints)
size_t ffn_x3_stream_bytes(int dff) { return (size_t)(dff / 32) * 64 * 1024 + 64; }
bool ffn_mix_applies() { return !cn_exp_env("CASSNAT_NO_FFN_MIX"); }

// the MIXF stream (kernel comment): per hidden tile ft, group a (0..7, W1 k-block a): [hi(k0 = 2a)][e4m3 bytes 0-15][hi(k1)][e4m3 bytes
// 16-31] with hi(ks)[lane][j] = half(W1[32ft + (lane&31)][16ks + 8(lane>>5) + j]) and the lane's 32 e4m3 bytes p = k - 32a of row
// 32ft + (lane&31): k-block (lane>>5) 0 = q = e4m3(w S_q), 1 = l = e4m3((w - half(w)) S_l); group 8 + b (output tile b of W2):
// [hi(s = 0)][e4m3 0-15][hi(s = 1)][e4m3 16-31] with hi(s)[lane][j] = half(W2[32b + (lane&31)][32ft + 16s + 8(j>>2) + 4(lane>>5) + (j&3)])
// and e4m3 byte p of row 32b + (lane&31): hidden unit 32ft + (p&3) + 8((p&15)>>2) + 4(p>>4) - the order in which the kernel's
// half-wave exchange lines the hidden tile up
static void pack_ffn_mix(const float* w1, const float* w2, int dff, unsigned char* out) {
    auto h16 = [](float v) { const _Float16 h = (_Float16)v; uint16_t b; memcpy(&b, &h, 2); return b; };
    auto hval = [](float v) { return (float)(_Float16)v; };
    float m1 = 0.f, m2 = 0.f;
    for (size_t i = 0; i < (size_t)dff * FX_D; ++i) { m1 = std::max(m1, std::fabs(w1[i])); m2 = std::max(m2, std::fabs(w2[i])); }
    const int lg1 = m1 > 0.f ? (int)std::floor(std::log2(448.f / m1)) : 0, lg2 = m2 > 0.f ? (int)std::floor(std::log2(448.f / m2)) : 0;
    const float s1q = std::ldexp(1.f, lg1), s1l = std::ldexp(1.f, lg1 + 11), s2q = std::ldexp(1.f, lg2), s2l = std::ldexp(1.f, lg2 + 11);
    for (int ft = 0; ft < dff / 32; ++ft) {
        unsigned char* tile = out + (size_t)ft * 64 * 1024;
        for (int a = 0; a < 8; ++a)
            for (int lane = 0; lane < 64; ++lane) {
                const float* row = w1 + (size_t)(32 * ft + (lane & 31)) * FX_D;
                for (int q = 0; q < 2; ++q)
                    for (int j = 0; j < 8; ++j) {
                        const uint16_t b = h16(row[16 * (2 * a + q) + 8 * (lane >> 5) + j]);
                        memcpy(tile + ((size_t)(4 * a + 2 * q) * 64 + lane) * 16 + 2 * j, &b, 2);
                    }
                for (int pb = 0; pb < 32; ++pb) {
                    const float v = row[32 * a + pb];
                    const unsigned char e = (lane >> 5) ? cn_f32_to_e4m3_host((v - hval(v)) * s1l) : cn_f32_to_e4m3_host(v * s1q);
                    tile[((size_t)(4 * a + 1 + 2 * (pb >> 4)) * 64 + lane) * 16 + (pb & 15)] = e;
                }
            }
        for (int b = 0; b < 8; ++b)
            for (int lane = 0; lane < 64; ++lane) {
                const float* row = w2 + (size_t)(32 * b + (lane & 31)) * dff + 32 * ft;
                for (int sx = 0; sx < 2; ++sx)
                    for (int j = 0; j < 8; ++j) {
                        const uint16_t hb = h16(row[16 * sx + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)]);
                        memcpy(tile + ((size_t)(32 + 4 * b + 2 * sx) * 64 + lane) * 16 + 2 * j, &hb, 2);
                    }
                for (int pb = 0; pb < 32; ++pb) {
                    const float v = row[(pb & 3) + 8 * ((pb & 15) >> 2) + 4 * (pb >> 4)];
                    const unsigned char e = (lane >> 5) ? cn_f32_to_e4m3_host((v - hval(v)) * s2l) : cn_f32_to_e4m3_host(v * s2q);
                    tile[((size_t)(32 + 4 * b + 1 + 2 * (pb >> 4)) * 64 + lane) * 16 + (pb & 15)] = e;
                }
            }
    }
    const int q[4] = {127 - lg1, 127 - (lg1 + 11), 127 - lg2, 127 - (lg2 + 11)};
    memcpy(out + (size_t)(dff / 32) * 64 * 1024, q, 16);
}

// out: [dff/32][64 fragments][64 lanes][8] bf16.  Per hidden tile ft:
//   fragments 4a + {0,1,2,3}      (a = 0..7): W1 hi(ks = 2a), lo(2a), hi(2a+1), lo(2a+1)      frag(ks)[lane][j] = W1[32ft + (lane&31)][16ks + 8(lane>>5) + j]
//   fragments 32 + 4b + {0,1,2,3} (b = 0..7): s = b>>2, nt0 = 2(b&3): W2 hi(s,nt0), lo(s,nt0), hi(s,nt0+1), lo(s,nt0+1)
//                                             frag(s,nt)[lane][j] = W2[32nt + (lane&31)][32ft + 16s + 8(j>>2) + 4(lane>>5) + (j&3)]
void pack_ffn_x3(const float* w1, const float* w2, int dff, uint16_t* out, bool mix) {
    if (mix) {
        pack_ffn_mix(w1, w2, dff, reinterpret_cast<unsigned char*>(out));
        return;
    }
    memset(reinterpret_cast<unsigned char*>(out) + (size_t)(dff / 32) * 64 * 1024, 0, 64);
    for (int ft = 0; ft < dff / 32; ++ft) {
        uint16_t* tile = out + (size_t)ft * 64 * 64 * 8;
        for (int a = 0; a < 8; ++a)
            for (int q = 0; q < 2; ++q) {
                const int ks = 2 * a + q;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        uint16_t hi, lo;
                        fx_split(w1[(size_t)(32 * ft + (lane & 31)) * FX_D + 16 * ks + 8 * (lane >> 5) + j], hi, lo);
                        tile[((size_t)(4 * a + 2 * q + 0) * 64 + lane) * 8 + j] = hi;
                        tile[((size_t)(4 * a + 2 * q + 1) * 64 + lane) * 8 + j] = lo;
                    }
            }
        for (int b = 0; b < 8; ++b)
            for (int q = 0; q < 2; ++q) {
                const int s = b >> 2, nt = 2 * (b & 3) + q;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int f = 32 * ft + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                        uint16_t hi, lo;
                        fx_split(w2[(size_t)(32 * nt + (lane & 31)) * dff + f], hi, lo);
                        tile[((size_t)(32 + 4 * b + 2 * q + 0) * 64 + lane) * 8 + j] = hi;
                        tile[((size_t)(32 + 4 * b + 2 * q + 1) * 64 + lane) * 8 + j] = lo;
                    }
            }
    }
}
