// Second subsampling convolution (Conv2d(256, 256, 3, stride 2, padding 1) + ReLU; reference:
// src/models/modules/embedding.py:104-108) as an implicit GEMM for gfx950, bf16:
//     out[m][co] = relu(bias[co] + sum_{tap, ci} in[b][2 t2 - 1 + kh][2 f2 - 1 + kw][ci] . W[co][tap][ci]),
//     m = (b, t2, f2),  M = B T2 F2 = 160 000 rows at config 2,  N = 256,  K = 9 x 256 = 2304  (188.7 GFLOP).
//
// This is the largest single product of the path (34 % of its FLOPs).  The generic GEMM (gemm.hip: 128x128 tiles,
// operands staged global -> registers -> LDS by the compute waves) reaches 0.6 PFLOP/s on it; what it pays for is
// exactly the staging (16 ds_write_b128 and their address math per wave and K step) and two-byte output stores.
//
//   * Tile 256 rows x all 256 output channels per workgroup, 4 waves (2 x 2, a 128 x 128 sub-tile = 16 accumulators =
//     256 AGPRs each; two waves per SIMD would leave 128 + 128 registers, which does not hold 128 accumulator
//     registers plus two fragment sets).  Every K sub-step is one hand-scheduled block of 16 MFMAs with the next
//     sub-step's 8 fragment reads and a later step's DMA pieces issued in its MFMA gaps.
//   * Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), 64 KiB per K step (64 channels of one
//     tap), double buffered; the XOR swizzle that makes the fragment reads conflict-free is applied to the per-lane
//     SOURCE address (the LDS image of a DMA piece is lane-linear).  conv1 writes its image with a one-cell zero
//     halo, so the padding taps are ordinary reads and a step's source offset is one scalar.  One s_barrier per K step: it is placed inside the step's last block, where every wave already
//     holds its last fragments, so the next step's first fragments are prefetched behind it.
//   * The products are computed transposed (D[co][m]: weights as the A operand) so that a lane ends up with four
//     consecutive channels of one row; the tile goes through LDS once and leaves as 8-KiB contiguous row runs.
//   * All LDS reads of the main loop are issued from inline asm (hipcc drains vmcnt before LDS accesses it can see
//     while an LDS-DMA is in flight).
#include <cstdlib>

#include "kernels.h"

struct Conv2Params {
    const unsigned char* A;      // conv1 output with a zero halo [B][T1 + 2][F1 + 2][256] bf16
    const unsigned char* W;      // [256][9 * 256] bf16, k = (kh * 3 + kw) * 256 + ci
    // X3 (template; split-bf16 engine): A and W are the planes of the hi halves, these the planes of the lo halves (same layouts)
    const unsigned char* A_lo;
    const unsigned char* W_lo;
    // F8 (template; the fp8 engine of BASELINE config 5): A and W hold e4m3fn bytes (same layouts, 256 / 9 * 256 bytes per row),
    // q8 = the two E8M0 scale bytes of the products (device: they travel in the weight blob): 127 - log2(scale of W), 127 - log2(scale of A)
    const int* q8;
    // MIX (template): A / W are the half-precision hi planes; the e4m3 planes of the cross terms - l = e4m3((v - hi) S_l),
    // q = e4m3(v S_q), images bordered like A, weights [256][9 * 256] bytes - and q8 = the four E8M0 bytes {W_q, A_l, W_l, A_q}
    const unsigned char* A8l;
    const unsigned char* A8q;
    const unsigned char* W8q;
    const unsigned char* W8l;
    // F8 convolution: out8_scale > 0 = the output rows are e4m3fn bytes at that scale ([M][256] bytes: the A operand of linear_out's
    // e4m3 form) instead of bf16
    float out8_scale;
    const float* bias;
    bf16* out;                   // [M][256] (X3: split-bf16 rows, 1 KiB each)
    int M, T1, F1, T2, F2, ntiles;
    // LINEAR mode (template): a plain [M][K] x [256][K]^T product with the embedding epilogue of linear_out
    // (embedding.py:118-119): out_f32[m][n] = (acc + bias[n]) * scale + pe[m % pe_period][n] (pe may be null)
    long long lda_bytes;  // row stride of A
    int ksteps;           // K / 64
    float* out_f32;
    const float* pe;
    int pe_period;
    float scale;
};

constexpr int C2_BM = 256, C2_N = 256, C2_C = 256, C2_ROWB = 128;
constexpr int C2_SLAB = C2_BM * C2_ROWB;             // 32 KiB: 256 rows (of A, or of W) x 128 bytes (64 channels)
constexpr int C2_WBASE = 3 * C2_SLAB;                // LDS: three A stages, then two W stages = the CU's whole 160 KiB
constexpr int C2_OSTRIDE = 528;                      // epilogue image: 512-byte rows + 16 (LDS bank spread)
constexpr int C2_LDS = 5 * C2_SLAB;
static_assert(C2_BM * C2_OSTRIDE <= C2_LDS && C2_BM == C2_N, "epilogue image fits; A and W slabs are the same size");
static_assert(C2_LDS <= 160 * 1024, "LDS budget");
constexpr int C2_KSTEPS = 9 * (C2_C / 64);
constexpr int C2_LIN8_KSTEPS = 40;  // K steps (of 128 bytes) of the e4m3 linear_out form, unrolled: K = 5120

#define C2_STR2(x) #x
#define C2_STR(x) C2_STR2(x)
#define C2_MF CN_MFMA16_ASM
#define C2_DMA(src, dst)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),          \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// One 16-wide k sub-step of a wave's 128 x 128 sub-tile: D[nt][mt] += Wc[nt] (rows = channels) x Ac[mt] (columns =
// output rows), 16 MFMAs; behind the first eight go the eight fragment reads of the NEXT sub-step (An from address an_:
// + 4096 mt; Wn from wn_: + 4096 nt), behind the last eight up to eight LDS-DMA pieces of a later K step (DMA = the
// instruction text; destination M0 = st_ + 4096 i, source = scalar base sb_ + per-piece lane offset vo[i]).
#ifdef C2_EXP_NO_READ  // timing experiments (wrong results): the blocks without their fragment reads / their slab requests
#define C2_RDA(i) ""
#define C2_RDW(i) ""
#else
#define C2_RDA(i) "ds_read_b128 %[na" #i "], %[an] offset:" C2_STR(i * 4096) "\n\t"
#define C2_RDW(i) "ds_read_b128 %[nw" #i "], %[wn] offset:" C2_STR(i * 4096) "\n\t"
#endif
#ifdef C2_EXP_NO_DMA
#define C2_DMA_I(i) ""
#else
#define C2_DMA_I(i) "s_add_u32 m0, %[st], " C2_STR(i * 4096) "\n\tglobal_load_lds_dwordx4 %[vo" #i "], %[sb]\n\t"
#endif
#define C2_NODMA(i) ""
#define C2_M(MF, nt, mt) MF "%[c" #nt #mt "], %[w" #nt "], %[a" #mt "], %[c" #nt #mt "]\n\t"
#define C2_BLOCK_X(MF, PRE, Ac, Wc, An, Wn, DMA, vo)                                                          \
    asm volatile(PRE "s_waitcnt lgkmcnt(0)\n\t"                                                                \
                 C2_M(MF, 0, 0) C2_RDA(0) C2_M(MF, 0, 1) C2_RDA(1) C2_M(MF, 0, 2) C2_RDA(2) C2_M(MF, 0, 3) C2_RDA(3)           \
                 C2_M(MF, 1, 0) C2_RDW(0) C2_M(MF, 1, 1) C2_RDW(1) C2_M(MF, 1, 2) C2_RDW(2) C2_M(MF, 1, 3) C2_RDW(3)           \
                 C2_M(MF, 2, 0) DMA(0) C2_M(MF, 2, 1) DMA(1) C2_M(MF, 2, 2) DMA(2) C2_M(MF, 2, 3) DMA(3)                       \
                 C2_M(MF, 3, 0) DMA(4) C2_M(MF, 3, 1) DMA(5) C2_M(MF, 3, 2) DMA(6) C2_M(MF, 3, 3) DMA(7)                       \
                 : [c00] "+a"(acc[0]), [c01] "+a"(acc[1]), [c02] "+a"(acc[2]), [c03] "+a"(acc[3]), [c10] "+a"(acc[4]), \
                   [c11] "+a"(acc[5]), [c12] "+a"(acc[6]), [c13] "+a"(acc[7]), [c20] "+a"(acc[8]), [c21] "+a"(acc[9]), \
                   [c22] "+a"(acc[10]), [c23] "+a"(acc[11]), [c30] "+a"(acc[12]), [c31] "+a"(acc[13]),          \
                   [c32] "+a"(acc[14]), [c33] "+a"(acc[15]), [na0] "=&v"(An[0]), [na1] "=&v"(An[1]),            \
                   [na2] "=&v"(An[2]), [na3] "=&v"(An[3]), [nw0] "=&v"(Wn[0]), [nw1] "=&v"(Wn[1]),              \
                   [nw2] "=&v"(Wn[2]), [nw3] "=&v"(Wn[3])                                                      \
                 : [a0] "v"(Ac[0]), [a1] "v"(Ac[1]), [a2] "v"(Ac[2]), [a3] "v"(Ac[3]), [w0] "v"(Wc[0]),        \
                   [w1] "v"(Wc[1]), [w2] "v"(Wc[2]), [w3] "v"(Wc[3]), [an] "v"(an_), [wn] "v"(wn_),            \
                   [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]),   \
                   [vo5] "v"(vo[5]), [vo6] "v"(vo[6]), [vo7] "v"(vo[7]), [st] "s"(st_), [sb] "s"(sb_)          \
                 : "memory", "scc")
#define C2_BLOCK(PRE, Ac, Wc, An, Wn, DMA, vo) C2_BLOCK_X(C2_MF, PRE, Ac, Wc, An, Wn, DMA, vo)
// the same block on half-precision operands whatever the library's own 16-bit type (the MIX form's hi x hi products)
#define C2_BLOCK_H(PRE, Ac, Wc, An, Wn, DMA, vo) C2_BLOCK_X("v_mfma_f32_32x32x16_f16 ", PRE, Ac, Wc, An, Wn, DMA, vo)
#define C2_BLK(PRE, Ac, Wc, An, Wn, DMA, vo)                      \
    do {                                                          \
        if constexpr (MIX) {                                      \
            C2_BLOCK_H(PRE, Ac, Wc, An, Wn, DMA, vo);             \
        } else {                                                  \
            C2_BLOCK(PRE, Ac, Wc, An, Wn, DMA, vo);               \
        }                                                         \
    } while (0)

// F8 form of a block: one 64-wide k sub-step = 16 x v_mfma_scale_f32_32x32x64_f8f6f4 (64 cycles each: the block takes as long as two
// bf16 blocks and covers four times the contraction).  An operand is the pair of 16-byte fragments the bf16 kernel reads for
// sub-steps 2 s and 2 s + 1 (same LDS image, same read instructions, same swizzle): lane half h then holds bytes of chunks
// 4 s + h and 4 s + 2 + h of the 128-byte slab row - for A and W alike, so every channel of the 64 meets its weight.  Sixteen
// fragment reads of the NEXT sub-step go behind the first eight MFMAs, up to eight DMA pieces behind the last eight.
#define C2_MF8 "v_mfma_scale_f32_32x32x64_f8f6f4 "
#define C2_M8(nt, mt) C2_MF8 "%[c" #nt #mt "], %[w" #nt "], %[a" #mt "], %[c" #nt #mt "], %[qa], %[qb] op_sel_hi:[0,0,0]\n\t"
#define C2_RDA8(i) "ds_read_b128 %[na" #i "], %[an0] offset:" C2_STR(i * 4096) "\n\tds_read_b128 %[nb" #i "], %[an1] offset:" C2_STR(i * 4096) "\n\t"
#define C2_RDW8(i) "ds_read_b128 %[nw" #i "], %[wn0] offset:" C2_STR(i * 4096) "\n\tds_read_b128 %[nv" #i "], %[wn1] offset:" C2_STR(i * 4096) "\n\t"
// W slab pieces are 32 weight rows apart: one lane offset, stepped by the block itself (eight more offset registers were
// eight too many for the e4m3 form: hipcc kept them in scratch and drained the request queue to reload them, every block)
#define C2_DMA_W8(i) "s_add_u32 m0, %[st], " C2_STR(i * 4096) "\n\tglobal_load_lds_dwordx4 %[tv], %[sb]\n\tv_add_u32 %[tv], %[wstep], %[tv]\n\t"
#define C2_JOIN(lo, hi) __builtin_shufflevector(__builtin_bit_cast(c2_v4i, lo), __builtin_bit_cast(c2_v4i, hi), 0, 1, 2, 3, 4, 5, 6, 7)
#define C2_BLOCK8(PRE, A0c, W0c, A1c, W1c, A0n, W0n, A1n, W1n, DMA, vo)                                        \
    asm volatile(PRE "s_waitcnt lgkmcnt(0)\n\t"                                                                \
                 C2_M8(0, 0) C2_RDA8(0) C2_M8(0, 1) C2_RDA8(1) C2_M8(0, 2) C2_RDA8(2) C2_M8(0, 3) C2_RDA8(3)   \
                 C2_M8(1, 0) C2_RDW8(0) C2_M8(1, 1) C2_RDW8(1) C2_M8(1, 2) C2_RDW8(2) C2_M8(1, 3) C2_RDW8(3)   \
                 C2_M8(2, 0) DMA(0) C2_M8(2, 1) DMA(1) C2_M8(2, 2) DMA(2) C2_M8(2, 3) DMA(3)                   \
                 C2_M8(3, 0) DMA(4) C2_M8(3, 1) DMA(5) C2_M8(3, 2) DMA(6) C2_M8(3, 3) DMA(7)                   \
                 : [c00] "+a"(acc[0]), [c01] "+a"(acc[1]), [c02] "+a"(acc[2]), [c03] "+a"(acc[3]), [c10] "+a"(acc[4]), \
                   [c11] "+a"(acc[5]), [c12] "+a"(acc[6]), [c13] "+a"(acc[7]), [c20] "+a"(acc[8]), [c21] "+a"(acc[9]), \
                   [c22] "+a"(acc[10]), [c23] "+a"(acc[11]), [c30] "+a"(acc[12]), [c31] "+a"(acc[13]),          \
                   [c32] "+a"(acc[14]), [c33] "+a"(acc[15]), [na0] "=&v"(A0n[0]), [na1] "=&v"(A0n[1]),          \
                   [na2] "=&v"(A0n[2]), [na3] "=&v"(A0n[3]), [nb0] "=&v"(A1n[0]), [nb1] "=&v"(A1n[1]),          \
                   [nb2] "=&v"(A1n[2]), [nb3] "=&v"(A1n[3]), [nw0] "=&v"(W0n[0]), [nw1] "=&v"(W0n[1]),          \
                   [nw2] "=&v"(W0n[2]), [nw3] "=&v"(W0n[3]), [nv0] "=&v"(W1n[0]), [nv1] "=&v"(W1n[1]),          \
                   [nv2] "=&v"(W1n[2]), [nv3] "=&v"(W1n[3]), [tv] "+v"(tv_)                                    \
                 : [a0] "v"(C2_JOIN(A0c[0], A1c[0])), [a1] "v"(C2_JOIN(A0c[1], A1c[1])), [a2] "v"(C2_JOIN(A0c[2], A1c[2])), \
                   [a3] "v"(C2_JOIN(A0c[3], A1c[3])), [w0] "v"(C2_JOIN(W0c[0], W1c[0])), [w1] "v"(C2_JOIN(W0c[1], W1c[1])), \
                   [w2] "v"(C2_JOIN(W0c[2], W1c[2])), [w3] "v"(C2_JOIN(W0c[3], W1c[3])), [an0] "v"(an0_), [an1] "v"(an1_), \
                   [wn0] "v"(wn0_), [wn1] "v"(wn1_), [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), \
                   [vo4] "v"(vo[4]), [vo5] "v"(vo[5]), [vo6] "v"(vo[6]), [vo7] "v"(vo[7]), [st] "s"(st_), [sb] "s"(sb_), \
                   [qa] "v"(qa_), [qb] "v"(qb_), [wstep] "s"(wstep_)                                           \
                 : "memory", "scc")
typedef int c2_v4i __attribute__((ext_vector_type(4)));

// LINEAR = false: the 3x3 / stride-2 convolution; true: a plain K-major GEMM with N = 256 (same tiles, stream and blocks)
// X3 (convolution only): the split-bf16 product.  conv1 wrote the image as two bf16 planes (hi, lo) and the weights come as two
// matrices; every K step of the bf16 loop becomes three - (A_hi, W_lo), (A_lo, W_hi), (A_hi, W_hi): the small terms first, the hi
// slab of A asked for twice in a row (the second time from L2) - on the same stages, blocks and accumulators: the kernel is the
// bf16 kernel with 108 K steps and another epilogue (bias + ReLU + hi / lo split, split-bf16 rows).
// F8 (convolution only): e4m3 operands on the K = 64 block-scaled MFMA at twice the bf16 rate (C2_BLOCK8); slabs, stages and
// requests as in the bf16 kernel, a slab row now being 128 channels: 18 K steps of two blocks each
// MIX (convolution only; the split-bf16 engine's input planes in another arithmetic): a value v is kept as h = half(v) plus
// two e4m3 bytes at fixed power-of-two scales, l = e4m3((v - h) S_l) and q = e4m3(v S_q), and a product is
//     a b = h_a h_b + l_a q_b + q_a l_b            (the dropped l_a l_b is 2^-24 relative, the 4-bit significands of the cross
// terms' factors cost 2^-16 each): 36 K steps of the 16-bit loop on v_mfma_f32_32x32x16_f16, then 2 x 18 e4m3 K steps - (A_l, W_q),
// (A_q, W_l) - at twice the rate, on the same stages and accumulators: 2 MFMA units per product where X3 spends 3.  Output: the
// split-bf16 rows of the X3 form (linear_out of that engine reads them).
template <bool LINEAR, bool X3 = false, bool F8 = false, bool MIX = false>
__global__ __launch_bounds__(256) void conv2_kernel(Conv2Params p) {
    static_assert(!(LINEAR && X3) && !(F8 && X3), "the split form exists for the convolution only; e4m3: convolution and linear_out");
    static_assert(!MIX || (!LINEAR && !X3 && !F8), "MIX is a form of the convolution of its own");
    constexpr int ES = F8 ? 1 : 2;  // bytes per image / weight element (MIX: of the 16-bit phase; its e4m3 phase recomputes for 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wnn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware tile order (as gemm.hip): neighbouring M tiles share conv halo rows; keep them on one XCD's L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, rm = nwg & 7;
    const int tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (blockIdx.x >> 3);
    const int m0 = tile * C2_BM;

    // ---- DMA sources of this lane: pieces j = wave + 4 i (i = 0..7) of the A slab and of the W slab; a piece is 8 rows
    // x 128 bytes, lane = 8 (row in piece) + chunk position; the lane fetches chunk (position ^ swizzle(row)).
    const int r8 = lane >> 3, cp = lane & 7;
    // (row >> 1) & 7 of row = 8 (wave + 4 i) + r8 does not depend on i: one swizzled chunk offset per lane
    const unsigned sw = (unsigned)((cp ^ ((4 * wave + (r8 >> 1)) & 7)) << 4);
    // The input image carries a one-cell zero halo ([B][T1+2][F1+2][256], written by conv1), so no tap is ever out of
    // range: the source of a row is (its fixed byte offset, a VGPR) + (a per-step, wave-uniform offset, the scalar base
    // of the DMA): the address arithmetic of a K step is two scalar adds.
    // Lane offsets are 32-bit and RELATIVE to the tile's first row (a wave-uniform 64-bit base, part of the scalar base of
    // every request): the rows of a tile are consecutive (b, t2, f2) positions, so they span a few image rows however large
    // the image (a merged engine pass of many batches exceeds 4 GiB; with absolute 32-bit offsets the reads wrapped silently)
    const int F1p = p.F1 + 2;
    auto a_off = [&](int mm, int es) -> long long {
        if constexpr (LINEAR) return (long long)mm * p.lda_bytes;
        const int f2 = mm % p.F2, bt = mm / p.F2;
        const int t2 = bt % p.T2, b = bt / p.T2;
        return (((long long)b * (p.T1 + 2) + 2 * t2) * F1p + 2 * f2) * (C2_C * es);
    };
    long long tile_off = a_off(m0, ES);  // (m0 < M: the grid has ceil(M / 256) workgroups)
    unsigned pa[8], pw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * (wave + 4 * i) + r8;
        const int m = m0 + row;
        const int mc = m < p.M ? m : p.M - 1;  // rows past M compute on the last row's data and are never stored
        pa[i] = (unsigned)(a_off(mc, ES) - tile_off) + sw;
        if constexpr (LINEAR)
            pw[i] = (unsigned)(row * (p.ksteps * 128)) + sw;
        else
            pw[i] = (unsigned)(row * (9 * C2_C * ES)) + sw;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned m0_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    // source bases / destinations of K step kt (channel block outermost, tap innermost), stage kt & 1
    const int KSTEPS = LINEAR ? p.ksteps : (X3 ? 3 * C2_KSTEPS : (F8 ? C2_KSTEPS / 2 : C2_KSTEPS));
    auto a_base = [&](int kt) -> const unsigned char* {
        if constexpr (LINEAR) return p.A + tile_off + (long long)kt * 128;
        const unsigned char* plane = p.A + tile_off;
        if constexpr (X3) {
            const int k = kt / 3;
            if (kt - 3 * k == 1) plane = p.A_lo + tile_off;
            kt = k;
        }
        const int cb = kt / 9, tap = kt - 9 * cb;
        const int kh = tap / 3, kw = tap - 3 * kh;
        return plane + (long long)((kh * F1p + kw) * (C2_C * ES) + cb * 128);
    };
    auto w_base = [&](int kt) -> const unsigned char* {
        if constexpr (LINEAR) return p.W + (long long)kt * 128;
        const unsigned char* plane = p.W;
        if constexpr (X3) {
            const int k = kt / 3;
            if (kt - 3 * k == 0) plane = p.W_lo;
            kt = k;
        }
        const int cb = kt / 9, tap = kt - 9 * cb;
        return plane + (long long)(tap * C2_C * ES + cb * 128);
    };
    // K step kt lives in A stage kt % 3 and W stage kt & 1
    auto a_dst = [&](int k3) -> unsigned { return m0_wave + (unsigned)(k3 * C2_SLAB); };
    auto w_dst = [&](int kt) -> unsigned { return m0_wave + (unsigned)(C2_WBASE + (kt & 1) * C2_SLAB); };
#define C2_ISSUE8(dst, vo, sbase)                                                                             \
    {                                                                                                         \
        const unsigned d_ = (dst);                                                                            \
        const unsigned char* s_ = (sbase);                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                      \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(d_ + i_ * 4096), "v"(vo[i_]), \
                         "s"(s_) : "memory");                                                                 \
    }

    // ---- fragment read addresses: row r of a slab, 16-byte chunk c lives at r * 128 + ((c ^ ((r >> 1) & 7)) << 4); the
    // swizzle of this lane's rows is the same for every 32-row tile, and chunk 2 ks + half of k sub-step ks sits at
    // (address of sub-step 0) ^ (ks << 5)
    const unsigned off0 = (((unsigned)half) ^ (unsigned)((l31 >> 1) & 7)) << 4;
    const unsigned a_rd0 = lds0 + (unsigned)((wm * 128 + l31) * C2_ROWB) + off0;
    const unsigned w_rd0 = lds0 + (unsigned)(C2_WBASE + (wnn * 128 + l31) * C2_ROWB) + off0;
#define C2_RD(base, ks, stage_off) (((base) ^ (unsigned)((ks) << 5)) + (stage_off))

    f32x16 acc[16];  // acc[4 nt + mt]: channels wnn * 128 + 32 nt + (accumulator rows), output rows wm * 128 + 32 mt + (lanes)
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 Ax[4], Wx[4], Ay[4], Wy[4];  // fragment sets X / Y alternate between sub-steps (4 per step: a step starts on X)

    const int KL = KSTEPS - 1;
#ifdef C2_EXP_NO_VMWAIT  // timing experiment: the K loop without its wait for the next slabs (wrong results)
#define C2_PRE3 "s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"
#else
#define C2_PRE3 "s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)\n\ts_barrier\n\t"
#endif
#define C2_PRE3_ALL "s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\ts_barrier\n\t"
    if constexpr (!F8) {  // ---- the 16-bit K steps (MIX: the hi x hi products on half-precision operands)
    C2_ISSUE8(a_dst(0), pa, a_base(0))
    C2_ISSUE8(w_dst(0), pw, w_base(0))
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    C2_ISSUE8(a_dst(1), pa, a_base(min(1, KL)))
    {  // fragments of step 0, sub-step 0
        const unsigned an_ = a_rd0, wn_ = w_rd0;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\t"
                     "ds_read_b128 %3, %8 offset:12288\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:4096\n\t"
                     "ds_read_b128 %6, %9 offset:8192\n\tds_read_b128 %7, %9 offset:12288"
                     : "=&v"(Ax[0]), "=&v"(Ax[1]), "=&v"(Ax[2]), "=&v"(Ax[3]), "=&v"(Wx[0]), "=&v"(Wx[1]), "=&v"(Wx[2]),
                       "=&v"(Wx[3])
                     : "v"(an_), "v"(wn_)
                     : "memory");
    }
    // K step kt reads A stage kt % 3 and W stage kt & 1.  Block 0 requests the W slab of step kt + 1, block 1 the A slab of step
    // kt + 2 - both into stages the barrier of step kt - 1 has freed, W first: block 3 waits for this wave's last fragments
    // (lgkmcnt) and for its share of step kt + 1 with vmcnt(8), i.e. everything but the A slab just requested, then the barrier
    // publishes step kt + 1.  The A slab comes from HBM: with two stages it had one K step (2048 cycles of MFMA) to arrive and
    // the loop waited for it (without the wait: -10 %); now it has a step and a half.  W comes from L2.
    // (LINEAR: branch-free, past the last K step the requests repeat the last slabs into stages nobody reads any more; the
    // convolution branches instead - in the branch-free form hipcc spills the request offsets to scratch inside the loop)
    int k3 = 0;  // kt % 3
    for (int kt = 0; kt < KSTEPS; ++kt) {
        const int k3n = k3 == 2 ? 0 : k3 + 1, k3p = k3 == 0 ? 2 : k3 - 1;
        const unsigned so_a = (unsigned)(k3 * C2_SLAB), so_w = (unsigned)((kt & 1) * C2_SLAB);
        const unsigned sn_a = (unsigned)(k3n * C2_SLAB), sn_w = (unsigned)(((kt + 1) & 1) * C2_SLAB);
        {
            const unsigned an_ = C2_RD(a_rd0, 1, so_a), wn_ = C2_RD(w_rd0, 1, so_w);
            const unsigned* vo = pw;
            if (LINEAR || kt + 1 < KSTEPS) {
                const unsigned st_ = w_dst(kt + 1);
                const unsigned char* sb_ = w_base(LINEAR ? min(kt + 1, KL) : kt + 1);
                C2_BLK("", Ax, Wx, Ay, Wy, C2_DMA_I, vo);
            } else {
                const unsigned st_ = 0;
                const unsigned char* sb_ = p.W;
                C2_BLK("", Ax, Wx, Ay, Wy, C2_NODMA, vo);
            }
        }
        {
            const unsigned an_ = C2_RD(a_rd0, 2, so_a), wn_ = C2_RD(w_rd0, 2, so_w);
            const unsigned* vo = pa;
            if (LINEAR || kt + 2 < KSTEPS) {
                const unsigned st_ = a_dst(k3p);  // (kt + 2) % 3
                const unsigned char* sb_ = a_base(LINEAR ? min(kt + 2, KL) : kt + 2);
                C2_BLK("", Ay, Wy, Ax, Wx, C2_DMA_I, vo);
            } else {
                const unsigned st_ = 0;
                const unsigned char* sb_ = p.W;
                C2_BLK("", Ay, Wy, Ax, Wx, C2_NODMA, vo);
            }
        }
        {
            const unsigned an_ = C2_RD(a_rd0, 3, so_a), wn_ = C2_RD(w_rd0, 3, so_w);
            const unsigned st_ = 0;
            const unsigned char* sb_ = p.W;
            const unsigned* vo = pw;
            C2_BLK("", Ax, Wx, Ay, Wy, C2_NODMA, vo);
        }
        {
            const unsigned an_ = a_rd0 + sn_a, wn_ = w_rd0 + sn_w;
            const unsigned st_ = 0;
            const unsigned char* sb_ = p.W;
            const unsigned* vo = pw;
            if (LINEAR || kt + 2 < KSTEPS) {  // the A slab of step kt + 2 is this wave's eight youngest requests
                C2_BLK(C2_PRE3, Ay, Wy, Ax, Wx, C2_NODMA, vo);
            } else {
                C2_BLK(C2_PRE3_ALL, Ay, Wy, Ax, Wx, C2_NODMA, vo);
            }
        }
        k3 = k3n;
    }
    // MFMA results -> vector reads below.  Both fragment sets are operands: the last block requested fragments nobody uses - dead
    // values to the compiler, whose registers it handed to the epilogue's address arithmetic while the reads were still on their
    // way (tools/pending_reg_check.py; harmless only as long as the reads land within the block's remaining MFMAs)
    asm volatile("s_nop 13\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(Ax[0]), "+v"(Ax[1]), "+v"(Ax[2]), "+v"(Ax[3]), "+v"(Wx[0]), "+v"(Wx[1]), "+v"(Wx[2]), "+v"(Wx[3]),
                   "+v"(Ay[0]), "+v"(Ay[1]), "+v"(Ay[2]), "+v"(Ay[3]), "+v"(Wy[0]), "+v"(Wy[1]), "+v"(Wy[2]), "+v"(Wy[3])
                 :: "memory");
    }
    if constexpr (F8 || MIX) {
        // ---- e4m3 loop: two blocks per K step.  Block 0 (sub-step 0 of stage kt; reads sub-step 1) requests the A slab of step
        // kt + 2; block 1 starts with the step's barrier (own fragments landed; everything but that A slab landed: vmcnt(8)), reads
        // sub-step 0 of step kt + 1 behind it and requests the W slab of step kt + 2 into the W stage the barrier has just freed.
        // So an A slab (HBM) has a step and a half to arrive, a W slab (L2) one step - as in the bf16 loop.
        bf16x8 Az[4], Wz[4], Au[4], Wu[4];
        constexpr int ES8 = 1;
        if constexpr (MIX) {  // the lane offsets once more, for one-byte elements
            tile_off = a_off(m0, ES8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * (wave + 4 * i) + r8;
                const int m = m0 + row;
                const int mc = m < p.M ? m : p.M - 1;
                pa[i] = (unsigned)(a_off(mc, ES8) - tile_off) + sw;
                pw[i] = (unsigned)(row * (9 * C2_C * ES8)) + sw;
            }
        }
        const unsigned wstep_ = LINEAR ? 32u * (unsigned)(p.ksteps * 128) : 32u * (9 * C2_C * ES8);  // byte distance of consecutive W pieces (32 weight rows)
        const unsigned pw0 = pw[0];
        unsigned tv_ = pw0;
        // K steps of this phase (128 channels each).  MIX: ONE run of 2 x 18 steps - (A_l, W_q) with q8[0..1], then (A_q, W_l) with
        // q8[2..3] - on the same stages: the second half's first slabs are requested by the first half's last steps
        // (everything unrolled: across a back edge hipcc joins the fragment sets with moves of registers whose reads are in flight)
        constexpr int KH8 = C2_KSTEPS / 2;
        constexpr int KS8 = LINEAR ? C2_LIN8_KSTEPS : (MIX ? 2 * KH8 : KH8);
        const int KL8 = KS8 - 1;
        const int q8a0 = p.q8[0], q8b0 = p.q8[1], q8a1 = MIX ? p.q8[2] : 0, q8b1 = MIX ? p.q8[3] : 0;
        const unsigned char* const A8_0 = MIX ? p.A8l + tile_off : p.A + tile_off;
        const unsigned char* const A8_1 = MIX ? p.A8q + tile_off : p.A + tile_off;
        const unsigned char* const W8_0 = MIX ? p.W8q : p.W;
        const unsigned char* const W8_1 = MIX ? p.W8l : p.W;
        auto a_base8 = [&](int kt) -> const unsigned char* {
            if constexpr (LINEAR) return A8_0 + (long long)kt * 128;
            const unsigned char* pl = (MIX && kt >= KH8) ? A8_1 : A8_0;
            if (MIX && kt >= KH8) kt -= KH8;
            const int cb = kt / 9, tap = kt - 9 * cb;
            const int kh = tap / 3, kw = tap - 3 * kh;
            return pl + (long long)((kh * F1p + kw) * (C2_C * ES8) + cb * 128);
        };
        auto w_base8 = [&](int kt) -> const unsigned char* {
            if constexpr (LINEAR) return W8_0 + (long long)kt * 128;
            const unsigned char* pl = (MIX && kt >= KH8) ? W8_1 : W8_0;
            if (MIX && kt >= KH8) kt -= KH8;
            const int cb = kt / 9, tap = kt - 9 * cb;
            return pl + (long long)(tap * C2_C * ES8 + cb * 128);
        };
        {
        C2_ISSUE8(a_dst(0), pa, a_base8(0))
        C2_ISSUE8(w_dst(0), pw, w_base8(0))
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        C2_ISSUE8(a_dst(1), pa, a_base8(min(1, KL8)))
        C2_ISSUE8(w_dst(1), pw, w_base8(min(1, KL8)))
        {  // fragments of step 0, sub-step 0 (chunks of the bf16 sub-steps 0 and 1), landed before anything else touches them
            const unsigned an0_ = a_rd0, an1_ = C2_RD(a_rd0, 1, 0), wn0_ = w_rd0, wn1_ = C2_RD(w_rd0, 1, 0);
            asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:4096\n\tds_read_b128 %2, %16 offset:8192\n\t"
                         "ds_read_b128 %3, %16 offset:12288\n\tds_read_b128 %4, %17\n\tds_read_b128 %5, %17 offset:4096\n\t"
                         "ds_read_b128 %6, %17 offset:8192\n\tds_read_b128 %7, %17 offset:12288\n\t"
                         "ds_read_b128 %8, %18\n\tds_read_b128 %9, %18 offset:4096\n\tds_read_b128 %10, %18 offset:8192\n\t"
                         "ds_read_b128 %11, %18 offset:12288\n\tds_read_b128 %12, %19\n\tds_read_b128 %13, %19 offset:4096\n\t"
                         "ds_read_b128 %14, %19 offset:8192\n\tds_read_b128 %15, %19 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(Ax[0]), "=&v"(Ax[1]), "=&v"(Ax[2]), "=&v"(Ax[3]), "=&v"(Wx[0]), "=&v"(Wx[1]), "=&v"(Wx[2]),
                           "=&v"(Wx[3]), "=&v"(Ay[0]), "=&v"(Ay[1]), "=&v"(Ay[2]), "=&v"(Ay[3]), "=&v"(Wy[0]), "=&v"(Wy[1]),
                           "=&v"(Wy[2]), "=&v"(Wy[3])
                         : "v"(an0_), "v"(wn0_), "v"(an1_), "v"(wn1_)
                         : "memory");
        }
        // (fully unrolled - 18 steps of two blocks: with a back edge hipcc gave the 256 accumulator registers another
        // assignment at the loop's end than at its head and moved half of them through scratch, every iteration)
        // (LINEAR: linear_out of the 20 x 256 = 5120-wide embedding input: 40 steps - the launcher checks)
#pragma unroll
        for (int kt = 0; kt < KS8; ++kt) {
            const int qa_ = (MIX && kt >= KH8) ? q8a1 : q8a0, qb_ = (MIX && kt >= KH8) ? q8b1 : q8b0;
            const int k3 = kt % 3;
            const int k3n = k3 == 2 ? 0 : k3 + 1, k3p = k3 == 0 ? 2 : k3 - 1;
            const unsigned so_a = (unsigned)(k3 * C2_SLAB), so_w = (unsigned)((kt & 1) * C2_SLAB);
            const unsigned sn_a = (unsigned)(k3n * C2_SLAB), sn_w = (unsigned)(((kt + 1) & 1) * C2_SLAB);
            // (branch-free: past the last K step the requests repeat the last slabs into stages nobody reads any more - with the
            // two forms of a block behind a branch, as in the bf16 loop, hipcc reshuffled all fragment sets through scratch at every
            // join; the wait after the loop covers the redundant requests)
            {
                const unsigned an0_ = C2_RD(a_rd0, 2, so_a), an1_ = C2_RD(a_rd0, 3, so_a);
                const unsigned wn0_ = C2_RD(w_rd0, 2, so_w), wn1_ = C2_RD(w_rd0, 3, so_w);
                const unsigned* vo = pa;
                const unsigned st_ = a_dst(k3p);  // (kt + 2) % 3
                const unsigned char* sb_ = a_base8(min(kt + 2, KL8));
                C2_BLOCK8("", Ax, Wx, Ay, Wy, Az, Wz, Au, Wu, C2_DMA_I, vo);
            }
            {
                const unsigned an0_ = a_rd0 + sn_a, an1_ = C2_RD(a_rd0, 1, sn_a);
                const unsigned wn0_ = w_rd0 + sn_w, wn1_ = C2_RD(w_rd0, 1, sn_w);
                const unsigned* vo = pa;  // (not used by this block's requests)
                const unsigned st_ = w_dst(kt + 2);
                const unsigned char* sb_ = w_base8(min(kt + 2, KL8));
                tv_ = pw0;
                C2_BLOCK8(C2_PRE3, Az, Wz, Au, Wu, Ax, Wx, Ay, Wy, C2_DMA_W8, vo);
            }
        }
        // a 16-pass MFMA's result is readable 19 wait states later; the last block's (unused) fragments land here
        asm volatile("s_nop 15\n\ts_nop 4\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)"
                     : "+v"(Ax[0]), "+v"(Ax[1]), "+v"(Ax[2]), "+v"(Ax[3]), "+v"(Wx[0]), "+v"(Wx[1]), "+v"(Wx[2]), "+v"(Wx[3]),
                       "+v"(Ay[0]), "+v"(Ay[1]), "+v"(Ay[2]), "+v"(Ay[3]), "+v"(Wy[0]), "+v"(Wy[1]), "+v"(Wy[2]), "+v"(Wy[3])
                     :: "memory");
        asm volatile("" : "+v"(Az[0]), "+v"(Az[1]), "+v"(Az[2]), "+v"(Az[3]), "+v"(Wz[0]), "+v"(Wz[1]), "+v"(Wz[2]), "+v"(Wz[3]),
                     "+v"(Au[0]), "+v"(Au[1]), "+v"(Au[2]), "+v"(Au[3]), "+v"(Wu[0]), "+v"(Wu[1]), "+v"(Wu[2]), "+v"(Wu[3]) :: "memory");
        }
    }
    if constexpr (LINEAR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant requests of the last steps (they write this workgroup's LDS)

    if constexpr (LINEAR) {
        // ---- linear_out epilogue: (acc + bias) * scale + PE row, fp32, four consecutive channels of one row per store
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int mrow = m0 + wm * 128 + 32 * mt + l31;
            if (mrow >= p.M) continue;
            float* orow = p.out_f32 + (long long)mrow * C2_N + wnn * 128 + 4 * half;
            const float* prow = p.pe ? p.pe + (long long)(mrow % p.pe_period) * C2_N + wnn * 128 + 4 * half : nullptr;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + wnn * 128 + 32 * nt + 8 * g + 4 * half);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (acc[4 * nt + mt][4 * g + e] + bv[e]) * p.scale;
                    if (prow) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(prow + 32 * nt + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += pv[e];
                    }
                    *reinterpret_cast<f32x4*>(orow + 32 * nt + 8 * g) = o;
                }
            asm volatile("" ::: "memory");  // one row tile in flight at a time (register pressure)
        }
        return;
    }
    if constexpr (X3 || MIX) {
        // ---- split epilogue: + bias, ReLU, hi / lo split; a split-bf16 row is 1 KiB (per 32 channels 64 B of hi halves, then
        // 64 B of lo halves), so the tile goes through LDS in two halves of 128 rows (the waves wm = 0, then wm = 1) and leaves
        // as contiguous 1-KiB rows
        constexpr int OST = 1024 + 16;
        static_assert(128 * OST <= C2_LDS, "half a split tile fits");
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if (wm == pass) {
                unsigned char* orow = smem + l31 * OST + (wnn * 4) * 128 + 8 * half;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 bv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + wnn * 128 + 32 * nt + 8 * g + 4 * half);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = fmaxf(acc[4 * nt + mt][4 * g + e] + bv[g][e], 0.f);
                            bf16x4 hi, lo;
                            cn_split4(o, hi, lo);
                            unsigned char* d = orow + mt * 32 * OST + nt * 128 + 16 * g;
                            *reinterpret_cast<bf16x4*>(d) = hi;
                            *reinterpret_cast<bf16x4*>(d + 64) = lo;
                        }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int row = 4 * it + (tid >> 6), ch = tid & 63;
                const int m = m0 + 128 * pass + row;
                if (m < p.M)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (long long)m * 1024 + ch * 16) =
                        *reinterpret_cast<const uint4*>(smem + row * OST + ch * 16);
            }
        }
        return;
    }
    if constexpr (F8) {
        if (p.out8_scale > 0.f) {
            // ---- e4m3 epilogue: + bias, ReLU (and saturation) in one v_med3, x scale, four channels = one dword into the row image
            // (272-byte rows), out as contiguous 256-byte rows
            constexpr int OST8 = 256 + 16;
            __syncthreads();
            unsigned char* orow = smem + (wm * 128 + l31) * OST8 + (wnn * 128 + 4 * half);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 bv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + wnn * 128 + 32 * nt + 8 * g + 4 * half);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = __builtin_amdgcn_fmed3f((acc[4 * nt + mt][4 * g + e] + bv[g][e]) * p.out8_scale, 0.f, CN_FP8_MAX);
                        unsigned d = 0;
                        d = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], d, false);
                        d = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], d, true);
                        *reinterpret_cast<unsigned*>(orow + mt * 32 * OST8 + 32 * nt + 8 * g) = d;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = 16 * it + (tid >> 4), ch = tid & 15;
                const int m = m0 + row;
                if (m < p.M)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (long long)m * 256 + ch * 16) =
                        *reinterpret_cast<const uint4*>(smem + row * OST8 + ch * 16);
            }
            return;
        }
    }
    // ---- epilogue: + bias, ReLU, bf16, through LDS (row image of the tile), out as contiguous 512-byte rows
    __syncthreads();
    {
        unsigned char* orow = smem + (wm * 128 + l31) * C2_OSTRIDE + (wnn * 128 + 4 * half) * 2;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + wnn * 128 + 32 * nt + 8 * g + 4 * half);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)fmaxf(acc[4 * nt + mt][4 * g + e] + bv[g][e], 0.f);
                    *reinterpret_cast<bf16x4*>(orow + mt * 32 * C2_OSTRIDE + (32 * nt + 8 * g) * 2) = o;
                }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int row = 8 * it + (tid >> 5), ch = tid & 31;
        const int m = m0 + row;
        if (m < p.M)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (long long)m * 512 + ch * 16) =
                *reinterpret_cast<const uint4*>(smem + row * C2_OSTRIDE + ch * 16);
    }
}

// bf16, 256 -> 256 channels only; everything else stays on the generic implicit GEMM (gemm.hip)
bool conv2_dma_applies(int prec, int C, int N) { return prec == CN_PREC_BF16 && C == C2_C && N == C2_N && !cn_exp_env("CASSNAT_NO_CONV2_DMA"); }

// The lane offsets of a tile are relative to its first row: what has to fit 32 bits is the span of 256 consecutive output
// positions in the image - (256 / F2 + 2) output rows of two image rows each, plus a batch boundary's two halo rows
static bool conv2_tile_span_too_large(int T1, int F1, int F2) {
    const long long span = ((256ll / (F2 > 0 ? F2 : 1) + 3) * 2 + 4) * (F1 + 2) * C2_C * 2;
    if (span >= (1ll << 32)) {
        cn_set_error("conv2: one 256-row tile spans more than 4 GiB of the image (feature dimension too wide for the LDS-DMA kernel)");
        return true;
    }
    (void)T1;
    return false;
}

// `in`: conv1 output WITH the zero halo, [B][T1 + 2][F1 + 2][256] bf16
int launch_conv2_dma(const void* in, const void* w, const float* bias, void* out, int B, int T1, int F1, int T2, int F2,
                     hipStream_t s) {
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)conv2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS));
        attr_once.mark(attr_dev);
    }
    Conv2Params p;
    p.A = (const unsigned char*)in;
    p.W = (const unsigned char*)w;
    p.bias = bias;
    p.out = (bf16*)out;
    p.M = B * T2 * F2;
    p.T1 = T1;
    p.F1 = F1;
    p.T2 = T2;
    p.F2 = F2;
    p.ntiles = cn_ceil_div(p.M, C2_BM);
    if (p.M <= 0) return 0;
    if (conv2_tile_span_too_large(T1, F1, F2)) return -1;
    hipLaunchKernelGGL(conv2_kernel<false>, dim3(p.ntiles), dim3(256), C2_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// the fp8 engine's convolution (BASELINE config 5): `in8` = conv1's e4m3fn image WITH the zero halo ([B][T1 + 2][F1 + 2][256] bytes),
// `w8` = [256][9 * 256] e4m3fn (k = (kh * 3 + kw) * 256 + ci), `q8_dev` = {127 - log2(weight scale), 127 - log2(image scale)}; out bf16
bool conv2_f8_applies(int C, int N) { return C == C2_C && N == C2_N && !cn_exp_env("CASSNAT_NO_CONV2_F8"); }

int launch_conv2_f8(const void* in8, const void* w8, const int* q8_dev, const float* bias, void* out, int B, int T1, int F1, int T2,
                    int F2, hipStream_t s, float out8_scale) {
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)conv2_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS));
        attr_once.mark(attr_dev);
    }
    if (conv2_tile_span_too_large(T1, F1, F2)) return -1;
    Conv2Params p = {};
    p.A = (const unsigned char*)in8;
    p.W = (const unsigned char*)w8;
    p.q8 = q8_dev;
    p.out8_scale = out8_scale;
    p.bias = bias;
    p.out = (bf16*)out;
    p.M = B * T2 * F2;
    p.T1 = T1;
    p.F1 = F1;
    p.T2 = T2;
    p.F2 = F2;
    p.ntiles = cn_ceil_div(p.M, C2_BM);
    if (p.M <= 0) return 0;
    hipLaunchKernelGGL((conv2_kernel<false, false, true>), dim3(p.ntiles), dim3(256), C2_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// linear_out of the fp8 engine on the same tile kernel: A [M][5120] e4m3fn (conv2's e4m3 output rows viewed per frame), W [256][5120]
// e4m3fn, q8_dev = {127 - log2(weight scale), 127 - log2(activation scale)}; epilogue as the bf16 form: (acc + bias) * scale + PE, fp32
bool linear256_f8_applies(int N, int K) { return N == C2_N && K == C2_LIN8_KSTEPS * 128 && !cn_exp_env("CASSNAT_NO_LINEAR_F8"); }

int launch_linear256_f8(const void* A8, const void* W8, const int* q8_dev, const float* bias, float* out, int M, int K, float scale,
                        const float* pe, int pe_period, hipStream_t s) {
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)conv2_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS));
        attr_once.mark(attr_dev);
    }
    if (M <= 0) return 0;
    if (!linear256_f8_applies(C2_N, K)) {
        cn_set_error("linear256 (e4m3): K must be 5120");
        return -1;
    }
    Conv2Params p = {};
    p.A = (const unsigned char*)A8;
    p.W = (const unsigned char*)W8;
    p.q8 = q8_dev;
    p.bias = bias;
    p.M = M;
    p.T1 = p.F1 = p.T2 = p.F2 = 1;
    p.ntiles = cn_ceil_div(M, C2_BM);
    p.lda_bytes = (long long)K;
    p.ksteps = K / 128;
    p.out_f32 = out;
    p.pe = pe;
    p.pe_period = pe_period > 0 ? pe_period : 1;
    p.scale = scale;
    hipLaunchKernelGGL((conv2_kernel<true, false, true>), dim3(p.ntiles), dim3(256), C2_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// the split-bf16 engine's convolution on the same kernel: `in_hi` / `in_lo` are the haloed bf16 planes conv1 wrote
// ([B][T1 + 2][F1 + 2][256] each), `w_hi` / `w_lo` the [256][9 * 256] matrices of the weights' halves; out: split-bf16 rows
bool conv2_x3_applies(int prec, int C, int N) { return prec == CN_PREC_X3 && C == C2_C && N == C2_N && !cn_exp_env("CASSNAT_NO_CONV2_X3"); }

int launch_conv2_x3(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias, void* out, int B,
                    int T1, int F1, int T2, int F2, hipStream_t s) {
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)conv2_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS));
        attr_once.mark(attr_dev);
    }
    if (conv2_tile_span_too_large(T1, F1, F2)) return -1;
    Conv2Params p = {};
    p.A = (const unsigned char*)in_hi;
    p.A_lo = (const unsigned char*)in_lo;
    p.W = (const unsigned char*)w_hi;
    p.W_lo = (const unsigned char*)w_lo;
    p.bias = bias;
    p.out = (bf16*)out;
    p.M = B * T2 * F2;
    p.T1 = T1;
    p.F1 = F1;
    p.T2 = T2;
    p.F2 = F2;
    p.ntiles = cn_ceil_div(p.M, C2_BM);
    if (p.M <= 0) return 0;
    hipLaunchKernelGGL((conv2_kernel<false, true>), dim3(p.ntiles), dim3(256), C2_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// the split-bf16 engine's convolution in the MIX arithmetic (kernel comment): `img` = conv1's three bordered planes - half-precision
// hi values ([B][T1 + 2][F1 + 2][256] x 2 bytes), then the l bytes, then the q bytes -, `w_hi` [256][9 * 256] half, `w8q` / `w8l` the
// e4m3 matrices, `q8_dev` = {127 - log2 S(W_q), 127 - log2 S(A_l), 127 - log2 S(W_l), 127 - log2 S(A_q)}; out: split-bf16 rows
bool conv2_mix_applies(int prec, int C, int N) { return prec == CN_PREC_X3 && C == C2_C && N == C2_N && !cn_exp_env("CASSNAT_NO_CONV2_MIX"); }

int launch_conv2_mix(const void* img, const void* w_hi, const void* w8q, const void* w8l, const int* q8_dev, const float* bias, void* out,
                     int B, int T1, int F1, int T2, int F2, hipStream_t s) {
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)conv2_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS));
        attr_once.mark(attr_dev);
    }
    if (conv2_tile_span_too_large(T1, F1, F2)) return -1;
    const size_t cells = (size_t)B * (T1 + 2) * (F1 + 2) * C2_C;
    Conv2Params p = {};
    p.A = (const unsigned char*)img;
    p.A8l = p.A + 2 * cells;
    p.A8q = p.A + 3 * cells;
    p.W = (const unsigned char*)w_hi;
    p.W8q = (const unsigned char*)w8q;
    p.W8l = (const unsigned char*)w8l;
    p.q8 = q8_dev;
    p.bias = bias;
    p.out = (bf16*)out;
    p.M = B * T2 * F2;
    p.T1 = T1;
    p.F1 = F1;
    p.T2 = T2;
    p.F2 = F2;
    p.ntiles = cn_ceil_div(p.M, C2_BM);
    if (p.M <= 0) return 0;
    hipLaunchKernelGGL((conv2_kernel<false, false, false, true>), dim3(p.ntiles), dim3(256), C2_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// linear_out (+ sqrt(d) scale + positional rows) on the same tile kernel: A [M][K] bf16 (row stride lda elements), W [256][K]
// bf16, K % 64 == 0, A and W below 4 GiB.  A launch occupies ceil(M / 256) CUs (32 for the encoder input of config 2) for about
// as long as the generic GEMM occupies all of them.
bool linear256_dma_applies(int prec, int N, int K) { return prec == CN_PREC_BF16 && N == C2_N && K % 64 == 0 && K >= 128 && !cn_exp_env("CASSNAT_NO_LINEAR_DMA"); }

int launch_linear256_dma(const void* A, int lda, const void* W, const float* bias, float* out, int M, int K, float scale,
                         const float* pe, int pe_period, hipStream_t s) {
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)conv2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS));
        attr_once.mark(attr_dev);
    }
    if (M <= 0) return 0;
    if (256ll * lda * 2 >= (1ll << 32)) {
        cn_set_error("linear256: a 256-row tile of A exceeds the 32-bit lane offsets of the LDS-DMA kernel");
        return -1;
    }
    Conv2Params p = {};
    p.A = (const unsigned char*)A;
    p.W = (const unsigned char*)W;
    p.bias = bias;
    p.M = M;
    p.T1 = p.F1 = p.T2 = p.F2 = 1;
    p.ntiles = cn_ceil_div(M, C2_BM);
    p.lda_bytes = (long long)lda * 2;
    p.ksteps = K / 64;
    p.out_f32 = out;
    p.pe = pe;
    p.pe_period = pe_period > 0 ? pe_period : 1;
    p.scale = scale;
    hipLaunchKernelGGL(conv2_kernel<true>, dim3(p.ntiles), dim3(256), C2_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
