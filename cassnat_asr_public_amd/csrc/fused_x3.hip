// Fused position-wise feed-forward sublayer of the split-bf16 ("bf16x3") engine, gfx950, d_model = 256:
//     x <- x + W2 . relu(W1 . LN(x) + b1) + b2          [and optionally  xn_next <- LN_next(x), split-bf16]
// Same sublayer and the same structure as fused.hip (SublayerConnection(LayerNorm -> PositionwiseFeedForward): src/models/
// modules/utils.py:23-32, positionff.py:15-16, norm.py:15-18), in the parity-grade precision: every operand is a (bf16 hi,
// bf16 lo) pair and every product three MFMAs (lo.hi + hi.lo + hi.hi, fp32 accumulation; common.h split_t).  The generic
// path spends two GEMM launches and a 2 x M x d_ff x 4-byte round trip of the hidden activations per layer on this; here
// they never leave registers:
//   * LN(x) of the workgroup's 64 rows is written once to LDS as B-operand fragments, a hi plane and a lo plane;
//   * every wave owns d_ff / 4 hidden units and streams ITS weight fragments (hi and lo, pre-tiled at pack time in
//     consumption order: 64 KiB per 32 hidden units) by LDS-DMA into a private 16-slot ring - four groups of four 1-KiB
//     fragments; a group's slots are re-requested as soon as its fragments are in registers, every wait is a counted
//     vmcnt on the wave's own queue, no barrier in the main loop;
//   * X^T[f][m] = W1 . xn^T is computed swapped, so its accumulator (after bias + ReLU and the hi / lo split) is directly the
//     B operand of out^T[n][m] += W2[n][f] X[f][m];
//   * the four waves' out^T partials are summed through LDS, fused with b2, the residual and the next LayerNorm.
// Per streamed byte this precision does 1.5x the MFMAs of the bf16 kernel, which moves the kernel from L2-stream-bound
// towards MFMA-bound: 64 rows x 12.6 MFLOP per row x 3 = 2.4 GFLOP of MFMA work per 4 MiB of stream.
#include <cstdlib>
#include <cstring>

#include "kernels.h"

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
};

constexpr int FX_D = 256;
constexpr int FX_MT = 2;                          // 32-row M-tiles per workgroup
#define FX_PLANE_B 32768                          /* literal for the asm offsets: FX_MT * 16384 */
constexpr int FX_PLANE = FX_PLANE_B;              // bytes of one plane (hi or lo) of the xn fragments
static_assert(FX_PLANE == FX_MT * 16384, "plane size");
constexpr int FX_RING = 16 * 1024;                // per wave: 4 groups x 4 fragments x 1 KiB
constexpr int FX_P_STRIDE = FX_D + 4;             // floats per partial row in LDS
constexpr int FX_LDS_MAIN = 2 * FX_PLANE + 4 * FX_RING;
constexpr int FX_LDS_PART = 4 * 32 * FX_P_STRIDE * 4;
constexpr int FX_LDS = FX_LDS_MAIN > FX_LDS_PART ? FX_LDS_MAIN : FX_LDS_PART;
static_assert(FX_LDS <= 160 * 1024, "LDS budget");

#define FX_STR2(x) #x
#define FX_STR(x) FX_STR2(x)
#define FX_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__global__ __launch_bounds__(256) void ffn_x3_kernel(FfnX3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xn_s = smem;                    // [plane][mt][16 k-steps][64 lanes][16 B]
    float* part = reinterpret_cast<float*>(smem);  // [4][32][260] fp32, epilogue only (aliases everything)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * 32 * FX_MT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned char* ring = smem + 2 * FX_PLANE + wave_u * FX_RING;

    const int tiles_per_wave = p.dff / 32 / 4;
    const int ft0 = wave_u * tiles_per_wave;
    // every workgroup / wave walks its hidden tiles in a rotation of its own (the sum over tiles is order-free; the
    // workgroups then do not pull the same L2 lines at the same moment)
    const int rot = p.rotate ? (blockIdx.x * 7 + wave_u * 3) % tiles_per_wave : 0;
#define FX_TT(t) (((t) + rot) % tiles_per_wave)
    const uint4* wst = p.wst + (long long)ft0 * 64 * 64 + lane;
#define FX_DMA(src, slot)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(ring + (slot) * 1024), 16, 0, 0)
    // group g (0..15) of tile `tile` into ring section g & 3
#define FX_FILL(tile, g) { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                             \
        FX_DMA(wst + ((long long)(tile) * 64 + 4 * (g) + j_) * 64, 4 * ((g) & 3) + j_); }
    // one fragment of it (the refills ride in the MFMA gaps of a block, one per three MFMAs)
#define FX_FILL1(tile, g, j_) FX_DMA(wst + ((long long)(tile) * 64 + 4 * (g) + (j_)) * 64, 4 * ((g) & 3) + (j_));
#define FX_NOFILL1(tile, g, j_)
#define FX_PIN_3M1V()                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);

    // biases of the wave's hidden tiles in PROCESSING order (as fused.hip): register j, lane 32 p + i holds b1 of hidden
    // unit i of the tile processed at position 2 j + p
    float bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pos = 2 * j + half;
        bq[j] = pos < tiles_per_wave ? p.b1[32 * (ft0 + FX_TT(pos)) + l31] : 0.f;
    }
    // prologue: the first four groups go in flight before the LayerNorm below
    {
        const int t0 = FX_TT(0);
        FX_FILL(t0, 0) FX_FILL(t0, 1) FX_FILL(t0, 2) FX_FILL(t0, 3)
    }

    // ---- LayerNorm of the workgroup's rows -> hi / lo bf16 fragments in LDS (wave w: rows w, w+4, ...)
    {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_a + 4 * lane);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * lane);
        // element k = 4 lane + j of row r: fragment (mt = r >> 5, ks = k >> 4), lane slot 32 ((k >> 3) & 1) + (r & 31), byte 2 (k & 7)
        const int ks = lane >> 2, kh = (lane >> 1) & 1, kb = (lane & 1) * 8;
#pragma unroll
        for (int i = 0; i < 32 * FX_MT / 4; ++i) {
            const int r = wave + 4 * i;
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (long long)m * FX_D + 4 * lane);
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)FX_D;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) ss = fmaf(v[j] - mean, v[j] - mean, ss);
            const float denom = sqrtf(wave_sum(ss) / (float)(FX_D - 1)) + p.eps;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = g[j] * (v[j] - mean) / denom + bb[j];
            bf16x4 hi, lo;
            cn_split4(o, hi, lo);
            unsigned char* dst = xn_s + (((r >> 5) * 16 + ks) * 64 + kh * 32 + (r & 31)) * 16 + kb;
            *reinterpret_cast<bf16x4*>(dst) = hi;
            *reinterpret_cast<bf16x4*>(dst + FX_PLANE) = lo;
        }
    }
    __syncthreads();

    // LDS byte addresses for the inline-asm reads (hipcc drains vmcnt in front of every LDS access it can see while an
    // LDS-DMA is outstanding: all main-loop reads are asm with their own counted waits)
    const unsigned xfrag_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(xn_s + lane * 16);
    const unsigned slot_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(ring + lane * 16);

    f32x16 acc[FX_MT][8];
#pragma unroll
    for (int mt = 0; mt < FX_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // Software pipeline, one block = one group of four weight fragments (g = 0..15 per hidden tile: 8 of W1, 8 of W2):
    //   block g:  [counted vmcnt: group g + 1 has landed] -> issue the LDS reads of block g + 1 into the OTHER register set
    //             -> the twelve MFMAs of block g, with the four DMAs that re-request group g's slots (for group g + 4) in
    //                their gaps, one per three MFMAs -> lgkmcnt(0) for block g + 1's operands.
    // The LDS read burst of a block (12 KiB per wave for a W1 block) and its latency thus run under the previous block's
    // MFMAs instead of in front of its own (one wave per SIMD: nothing else would cover them).  DMAs younger than group
    // g + 1's at the wait: groups g + 2, g + 3 = 8 (4, 0 at the end of the last tile).
    // Register sets: weights W0 / W1 (four fragments each), activations X0 / X1 (eight each: hi / lo x two k-steps x two
    // M-tiles); block g uses set g & 1.
    bf16x8 w0a, w0b, w0c, w0d, w1a, w1b, w1c, w1d;
    bf16x8 x0a, x0b, x0c, x0d, x0e, x0f, x0g, x0h, x1a, x1b, x1c, x1d, x1e, x1f, x1g, x1h;
    // reads of W1 block a (k-steps 2a, 2a + 1): weights from ring section a & 3, activations hi/lo for both M-tiles
#define FX_READ_A(AI_, WAITN, W_, X_)                                                                           \
    asm volatile("s_waitcnt vmcnt(" FX_STR(WAITN) ")\n\t"                                                     \
                 "ds_read_b128 %0, %12 offset:" FX_STR((4 * ((AI_) & 3) + 0) * 1024) "\n\t"                     \
                 "ds_read_b128 %1, %12 offset:" FX_STR((4 * ((AI_) & 3) + 1) * 1024) "\n\t"                     \
                 "ds_read_b128 %2, %12 offset:" FX_STR((4 * ((AI_) & 3) + 2) * 1024) "\n\t"                     \
                 "ds_read_b128 %3, %12 offset:" FX_STR((4 * ((AI_) & 3) + 3) * 1024) "\n\t"                     \
                 "ds_read_b128 %4, %13 offset:" FX_STR((2 * (AI_) + 0) * 1024) "\n\t"                           \
                 "ds_read_b128 %5, %13 offset:" FX_STR(FX_PLANE_B + (2 * (AI_) + 0) * 1024) "\n\t"              \
                 "ds_read_b128 %6, %13 offset:" FX_STR((2 * (AI_) + 1) * 1024) "\n\t"                           \
                 "ds_read_b128 %7, %13 offset:" FX_STR(FX_PLANE_B + (2 * (AI_) + 1) * 1024) "\n\t"              \
                 "ds_read_b128 %8, %13 offset:" FX_STR(16384 + (2 * (AI_) + 0) * 1024) "\n\t"                   \
                 "ds_read_b128 %9, %13 offset:" FX_STR(FX_PLANE_B + 16384 + (2 * (AI_) + 0) * 1024) "\n\t"      \
                 "ds_read_b128 %10, %13 offset:" FX_STR(16384 + (2 * (AI_) + 1) * 1024) "\n\t"                  \
                 "ds_read_b128 %11, %13 offset:" FX_STR(FX_PLANE_B + 16384 + (2 * (AI_) + 1) * 1024)              \
                 : "=&v"(W_##a), "=&v"(W_##b), "=&v"(W_##c), "=&v"(W_##d), "=&v"(X_##a), "=&v"(X_##b), "=&v"(X_##c), \
                   "=&v"(X_##d), "=&v"(X_##e), "=&v"(X_##f), "=&v"(X_##g), "=&v"(X_##h)                       \
                 : "v"(slot_a), "v"(xfrag_a)                                                                  \
                 : "memory");                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // reads of W2 block b: four weight fragments from ring section b & 3
#define FX_READ_B(BI_, WAITN, W_)                                                                               \
    asm volatile("s_waitcnt vmcnt(" FX_STR(WAITN) ")\n\t"                                                     \
                 "ds_read_b128 %0, %4 offset:" FX_STR((4 * ((BI_) & 3) + 0) * 1024) "\n\t"                      \
                 "ds_read_b128 %1, %4 offset:" FX_STR((4 * ((BI_) & 3) + 1) * 1024) "\n\t"                      \
                 "ds_read_b128 %2, %4 offset:" FX_STR((4 * ((BI_) & 3) + 2) * 1024) "\n\t"                      \
                 "ds_read_b128 %3, %4 offset:" FX_STR((4 * ((BI_) & 3) + 3) * 1024)                               \
                 : "=&v"(W_##a), "=&v"(W_##b), "=&v"(W_##c), "=&v"(W_##d)                                     \
                 : "v"(slot_a)                                                                                \
                 : "memory");                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define FX_NOREAD()
#define FX_WAIT_A(W_, X_)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W_##a), "+v"(W_##b), "+v"(W_##c), "+v"(W_##d), "+v"(X_##a), "+v"(X_##b), \
                 "+v"(X_##c), "+v"(X_##d), "+v"(X_##e), "+v"(X_##f), "+v"(X_##g), "+v"(X_##h) :: "memory");   \
    __builtin_amdgcn_sched_barrier(0);
#define FX_WAIT_B(W_)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W_##a), "+v"(W_##b), "+v"(W_##c), "+v"(W_##d) :: "memory");    \
    __builtin_amdgcn_sched_barrier(0);
    // MFMAs of a W1 block on register set (W_, X_): fragments a..d = W1 hi(k0), lo(k0), hi(k1), lo(k1); X a..h = M-tile 0
    // hi(k0), lo(k0), hi(k1), lo(k1), M-tile 1 hi(k0), lo(k0), hi(k1), lo(k1)
#define FX_MFMA_A(W_, X_, REFILL)                                                                             \
    FX_MFMA(W_##b, X_##a, xh[0]); FX_MFMA(W_##a, X_##b, xh[0]); FX_MFMA(W_##a, X_##a, xh[0]); REFILL(0)       \
    FX_MFMA(W_##b, X_##e, xh[1]); FX_MFMA(W_##a, X_##f, xh[1]); FX_MFMA(W_##a, X_##e, xh[1]); REFILL(1)       \
    FX_MFMA(W_##d, X_##c, xh[0]); FX_MFMA(W_##c, X_##d, xh[0]); FX_MFMA(W_##c, X_##c, xh[0]); REFILL(2)       \
    FX_MFMA(W_##d, X_##g, xh[1]); FX_MFMA(W_##c, X_##h, xh[1]); FX_MFMA(W_##c, X_##g, xh[1]); REFILL(3)       \
    FX_PIN_3M1V() FX_PIN_3M1V() FX_PIN_3M1V() FX_PIN_3M1V()                                                   \
    __builtin_amdgcn_sched_barrier(0);
    // MFMAs of W2 block b: s = b >> 2, output tiles nt0 = 2 (b & 3), nt0 + 1; fragments a..d = W2 hi(nt0), lo(nt0), hi(nt0+1), lo(nt0+1)
#define FX_MFMA_B(BI_, W_, REFILL)                                                                              \
    FX_MFMA(W_##b, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3)]);                                                  \
    FX_MFMA(W_##a, pbl[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3)]);                                                  \
    FX_MFMA(W_##a, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3)]); REFILL(0)                                        \
    FX_MFMA(W_##d, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3) + 1]);                                              \
    FX_MFMA(W_##c, pbl[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3) + 1]);                                              \
    FX_MFMA(W_##c, pbh[0][(BI_) >> 2], acc[0][2 * ((BI_) & 3) + 1]); REFILL(1)                                    \
    FX_MFMA(W_##b, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3)]);                                                  \
    FX_MFMA(W_##a, pbl[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3)]);                                                  \
    FX_MFMA(W_##a, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3)]); REFILL(2)                                        \
    FX_MFMA(W_##d, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3) + 1]);                                              \
    FX_MFMA(W_##c, pbl[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3) + 1]);                                              \
    FX_MFMA(W_##c, pbh[1][(BI_) >> 2], acc[1][2 * ((BI_) & 3) + 1]); REFILL(3)                                    \
    FX_PIN_3M1V() FX_PIN_3M1V() FX_PIN_3M1V() FX_PIN_3M1V()                                                   \
    __builtin_amdgcn_sched_barrier(0);
    // bias + ReLU on hidden unit f = 32 tile + acc_row(r, lane), split into the hi / lo B operands of the W2 blocks
#define FX_RELU_PACK(pos)                                                                                     \
    {                                                                                                         \
        const int src0 = 32 * ((pos) & 1) + 4 * half;                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int e = 0; e < 4; ++e) {         \
            const float bv = __shfl(bq[0], src0 + 8 * g + e);                                                 \
            _Pragma("unroll") for (int mt = 0; mt < FX_MT; ++mt) {                                            \
                const float h_ = fmaxf(xh[mt][4 * g + e] + bv, 0.f);                                          \
                const bf16 hh_ = (bf16)h_;                                                                    \
                pbh[mt][g >> 1][4 * (g & 1) + e] = hh_;                                                       \
                pbl[mt][g >> 1][4 * (g & 1) + e] = (bf16)(h_ - (float)hh_);                                   \
            }                                                                                                 \
        }                                                                                                     \
        if ((pos) & 1) { _Pragma("unroll") for (int j = 0; j < 7; ++j) bq[j] = bq[j + 1]; }                   \
    }
#define FX_ZERO_XH()                                                                                          \
    _Pragma("unroll") for (int mt = 0; mt < FX_MT; ++mt) _Pragma("unroll") for (int r = 0; r < 16; ++r) xh[mt][r] = 0.f;
    // the sixteen blocks of a tile; RF(g): refill macro of block g (group g + 4 of this tile, or g - 12 of the next);
    // NEXT_A0: the reads of the next tile's block 0 (nothing after the last tile); VA / VB / VC: vmcnt of the reads issued in
    // blocks 12 / 13 / 14 (8 8 8 in a middle tile; 8 4 0 in the last one, where nothing is re-requested any more)
#define FX_TILE(RF12, RF13, RF14, RF15, VB, VC, NEXT_A0, NEXT_WAIT)                                           \
    FX_ZERO_XH()                                                                                              \
    FX_READ_A(1, 8, w1, x1) FX_MFMA_A(w0, x0, R4_) FX_WAIT_A(w1, x1)                                          \
    FX_READ_A(2, 8, w0, x0) FX_MFMA_A(w1, x1, R5_) FX_WAIT_A(w0, x0)                                          \
    FX_READ_A(3, 8, w1, x1) FX_MFMA_A(w0, x0, R6_) FX_WAIT_A(w1, x1)                                          \
    FX_READ_A(4, 8, w0, x0) FX_MFMA_A(w1, x1, R7_) FX_WAIT_A(w0, x0)                                          \
    FX_READ_A(5, 8, w1, x1) FX_MFMA_A(w0, x0, R8_) FX_WAIT_A(w1, x1)                                          \
    FX_READ_A(6, 8, w0, x0) FX_MFMA_A(w1, x1, R9_) FX_WAIT_A(w0, x0)                                          \
    FX_READ_A(7, 8, w1, x1) FX_MFMA_A(w0, x0, R10_) FX_WAIT_A(w1, x1)                                         \
    FX_READ_B(0, 8, w0) FX_MFMA_A(w1, x1, R11_) FX_WAIT_B(w0)                                                 \
    FX_RELU_PACK(t)                                                                                           \
    FX_READ_B(1, 8, w1) FX_MFMA_B(0, w0, R12_) FX_WAIT_B(w1)                                                  \
    FX_READ_B(2, 8, w0) FX_MFMA_B(1, w1, R13_) FX_WAIT_B(w0)                                                  \
    FX_READ_B(3, 8, w1) FX_MFMA_B(2, w0, R14_) FX_WAIT_B(w1)                                                  \
    FX_READ_B(4, 8, w0) FX_MFMA_B(3, w1, R15_) FX_WAIT_B(w0)                                                  \
    FX_READ_B(5, 8, w1) FX_MFMA_B(4, w0, RF12) FX_WAIT_B(w1)                                                  \
    FX_READ_B(6, VB, w0) FX_MFMA_B(5, w1, RF13) FX_WAIT_B(w0)                                                 \
    FX_READ_B(7, VC, w1) FX_MFMA_B(6, w0, RF14) FX_WAIT_B(w1)                                                 \
    NEXT_A0 FX_MFMA_B(7, w1, RF15) NEXT_WAIT
#define R4_(j) FX_FILL1(cur, 4, j)
#define R5_(j) FX_FILL1(cur, 5, j)
#define R6_(j) FX_FILL1(cur, 6, j)
#define R7_(j) FX_FILL1(cur, 7, j)
#define R8_(j) FX_FILL1(cur, 8, j)
#define R9_(j) FX_FILL1(cur, 9, j)
#define R10_(j) FX_FILL1(cur, 10, j)
#define R11_(j) FX_FILL1(cur, 11, j)
#define R12_(j) FX_FILL1(cur, 12, j)
#define R13_(j) FX_FILL1(cur, 13, j)
#define R14_(j) FX_FILL1(cur, 14, j)
#define R15_(j) FX_FILL1(cur, 15, j)
#define RN0_(j) FX_FILL1(nxt, 0, j)
#define RN1_(j) FX_FILL1(nxt, 1, j)
#define RN2_(j) FX_FILL1(nxt, 2, j)
#define RN3_(j) FX_FILL1(nxt, 3, j)
#define RNONE_(j)
    f32x16 xh[FX_MT];
    bf16x8 pbh[FX_MT][2], pbl[FX_MT][2];
    // the first block's operands: groups 1..3 of the prologue are younger
    FX_READ_A(0, 12, w0, x0) FX_WAIT_A(w0, x0)
    int t = 0;
    for (; t + 1 < tiles_per_wave; ++t) {
        const int cur = FX_TT(t), nxt = FX_TT(t + 1);
        FX_TILE(RN0_, RN1_, RN2_, RN3_, 8, 8, FX_READ_A(0, 8, w0, x0), FX_WAIT_A(w0, x0))
    }
    {
        const int cur = FX_TT(t);
        FX_TILE(RNONE_, RNONE_, RNONE_, RNONE_, 4, 0, FX_NOREAD(), FX_NOREAD())
    }
#undef FX_TILE
#undef FX_READ_A
#undef FX_READ_B
#undef FX_WAIT_A
#undef FX_WAIT_B
#undef FX_MFMA_A
#undef FX_MFMA_B
#undef FX_RELU_PACK
#undef FX_ZERO_XH
#undef FX_FILL
#undef FX_DMA
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
        __syncthreads();  // rings / xn fragments (mt == 0) or the previous round's partials are no longer read
        float* mine = part + (wave * 32 + l31) * FX_P_STRIDE;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[mt][nt][4 * g + e];
                *reinterpret_cast<f32x4*>(mine + 32 * nt + 8 * g + 4 * half) = o;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + 4 * i;
            const int m = m0 + 32 * mt + r;
            if (m >= p.M) continue;  // wave-uniform
            float* xr = p.x + (long long)m * FX_D + 4 * lane;
            f32x4 v = *reinterpret_cast<const f32x4*>(xr);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(part + (w * 32 + r) * FX_P_STRIDE + 4 * lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += q[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += b2v[j];
            *reinterpret_cast<f32x4*>(xr) = v;
            if (p.nln_a) {
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)FX_D;
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) ss = fmaf(v[j] - mean, v[j] - mean, ss);
                const float denom = sqrtf(wave_sum(ss) / (float)(FX_D - 1)) + p.eps;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = ng[j] * (v[j] - mean) / denom + nb[j];
                bf16x4 hi, lo;
                cn_split4(o, hi, lo);
                unsigned char* ob = p.xn_out + (long long)m * FX_D * 4 + cn_split_off((size_t)(4 * lane));
                *reinterpret_cast<bf16x4*>(ob) = hi;
                *reinterpret_cast<bf16x4*>(ob + 64) = lo;
            }
        }
    }
}

bool ffn_x3_applies(int d, int dff) { return d == FX_D && dff % 128 == 0 && dff >= 128 && dff <= 2048 && !getenv("CASSNAT_NO_FFN_X3"); }

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
    static const int rotate = getenv("CASSNAT_FFN_X3_ROTATE") ? atoi(getenv("CASSNAT_FFN_X3_ROTATE")) : 1;
    p.rotate = rotate;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL(ffn_x3_kernel, dim3(cn_ceil_div(p.M, 32 * FX_MT)), dim3(256), FX_LDS, s, p);
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

size_t ffn_x3_stream_bytes(int dff) { return (size_t)(dff / 32) * 64 * 1024; }

// out: [dff/32][64 fragments][64 lanes][8] bf16.  Per hidden tile ft:
//   fragments 4a + {0,1,2,3}      (a = 0..7): W1 hi(ks = 2a), lo(2a), hi(2a+1), lo(2a+1)      frag(ks)[lane][j] = W1[32ft + (lane&31)][16ks + 8(lane>>5) + j]
//   fragments 32 + 4b + {0,1,2,3} (b = 0..7): s = b>>2, nt0 = 2(b&3): W2 hi(s,nt0), lo(s,nt0), hi(s,nt0+1), lo(s,nt0+1)
//                                             frag(s,nt)[lane][j] = W2[32nt + (lane&31)][32ft + 16s + 8(j>>2) + 4(lane>>5) + (j&3)]
void pack_ffn_x3(const float* w1, const float* w2, int dff, uint16_t* out) {
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
