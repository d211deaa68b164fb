// Row-chain kernel for gfx950 (bf16 MFMA, d_model = 256): everything of a transformer layer that is not the
// attention itself, for a block of 128 rows, in ONE launch:
//
//     x  <- x + Wo . ctx + bo                      output projection of the attention that just ran + residual
//     x  <- x + W2 . relu(W1 . LN1(x) + b1) + b2   position-wise feed-forward sublayer + residual
//     y  <- LNn(x)                                 the NEXT sublayer's pre-norm (or the stack's final norm)
//     out<- Wt . y + bt                            the NEXT attention's fused Q|K|V (or Q) projection
//
// i.e. SublayerConnection / MultiHeadedAttention.linears[3] / PositionwiseFeedForward / LayerNorm /
// MultiHeadedAttention.linears[0..2] of the reference (src/models/modules/utils.py:23-32, attention.py:44-66,
// positionff.py:15-16, norm.py:15-18) - six launches and four round trips of the activations in the unfused path.
//
// Why this shape.  At B=32 a layer has M = 8000 rows.  The per-layer weights (2.6 MB bf16) stream from L2 into a CU at
// ~60 GB/s whatever the instruction (LDS-DMA or register loads), so a workgroup that owns few rows is bound by that
// stream, not by MFMA: the 64-row fused FFN kernel (fused.hip) keeps its CU's matrix pipe ~20 % busy.  Here a
// workgroup owns 128 rows - each of its four waves a private 32-row block - so every streamed fragment feeds four
// MFMAs and the pipe is ~80 % busy; a layer then occupies only 63 CUs, which is what lets the four decode pipelines
// of bench.py run side by side instead of queueing for the whole chip.
//
//   * Activations never leave registers.  Every product is computed transposed, Y^T[n][m] = sum_k W[n][k] X^T[k][m],
//     with the weight fragment as the A operand and the wave's 32 rows on the lane axis; the 32x32 fp32 accumulator
//     tile (n on registers, m on lanes) is, after a bf16 pack, exactly a B operand of the next product, provided the
//     next weight is packed in the k-order that implies ("acc order" below).  LayerNorm, bias, ReLU and the residual
//     are register arithmetic on that layout (one cross-half-wave shuffle per LayerNorm statistic).
//   * Weights are packed at load time into 16-KiB units (16 fragments of 64 lanes x 16 B) in consumption order and
//     stream through an 8-slot LDS ring by LDS-DMA (global_load_lds_dwordx4); each wave issues a quarter of every
//     unit, waits for its own quarter with a counted vmcnt, and ONE s_barrier per unit publishes the slot.  A slot is
//     refilled as soon as the barrier after its last read has passed, so 5-6 units (~90 KiB) stay in flight per CU;
//     (the outputs were first written with the nt policy, to keep them from evicting the weight stream from L2: the stores then
//     took so long to retire that the tail phase ran 40 % longer - plain stores now, see CH_TAIL_STORE.)
//   * All LDS reads are issued from inline asm: hipcc drains vmcnt to 0 in front of any LDS access it can see while
//     an LDS-DMA is outstanding.  Per-channel vectors (biases, LayerNorm gains) ride the same DMA into a 20-KiB table.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels.h"

constexpr int CH_D = 256;
constexpr int CH_UNIT_BYTES = 16384;
constexpr int CH_RING = 8;
constexpr int CH_TAB_FLOATS = 5120;
constexpr int CH_TAB_BYTES = CH_TAB_FLOATS * 4;  // 20 pieces of 1 KiB, 5 per wave
constexpr int CH_LDS = CH_TAB_BYTES + CH_RING * CH_UNIT_BYTES;
static_assert(CH_LDS <= 160 * 1024, "LDS budget");
// table layout (floats)
constexpr int CT_BO = 0, CT_LN1A = 256, CT_LN1B = 512, CT_B2 = 768, CT_NLNA = 1024, CT_NLNB = 1280, CT_BT = 1536,
              CT_B1 = 3072;  // tail biases: up to 1536; b1: up to 2048 -> 5120 = the whole table
constexpr int CH_MAX_TAIL = 48, CH_MAX_FFN_TILES = 64;

__device__ long long ch_stamps[16];  // phase timestamps of workgroup 0 / wave 0 (investigation aid: CASSNAT_CHAIN_STAMPS)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

struct ChainParams {
    float* x;          // [M][256] fp32 residual stream, updated in place
    const bf16* ctx;   // [M][ldctx] bf16 (null: no output projection)
    int ldctx;
    const uint4* wstream;  // packed units, consumption order
    const float* tab;      // CH_TAB_FLOATS floats
    bf16* out;             // tail projection [M][ldo] bf16
    int ldo;
    bf16* ln_out;          // LNn(x) itself [M][ld_ln] bf16, or null (both may be asked for: the last encoder layer writes enc_h
    int ld_ln;             // and the cross-attention K|V of every decoder-side layer)
    int M, ffn_tiles, tail_tiles, has_next;
    float eps;
    int stamps;
    int stamp_block;  // which workgroup writes the stamps (CASSNAT_CHAIN_STAMP_BLOCK: a later round's phases are not the first round's)
    int x_in_blk, x_out_blk, store_x;
    int out_blk;  // tail projection in the blocked layout (cn_blk16_off): every store instruction writes 1 KiB contiguous
    int ctx_blk;  // ctx in the attention kernel's blocked output layout: per 32-row block [16 k-steps][64 lanes][16 B]
    // F8 form (template): E8M0 scale bytes of the feed-forward products (v_mfma_scale_f32_32x32x64_f8f6f4: the product is
    // multiplied by 2^(qa - 127) 2^(qb - 127)) - W1 . LN1(x) and W2 . hidden
    const int* f8_q;  // [4]: qa1, qb1, qa2, qb2
};

#define CH_STR2(x) #x
#define CH_STR(x) CH_STR2(x)
#define CH_DMA(src, dst)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),          \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// eight fragments of a unit: OFFK = 0 (first half) or 8 (second half)
#define CH_READ8(F, addr, OFFK)                                                                       \
    asm volatile("ds_read_b128 %0, %8 offset:" CH_STR(((OFFK) + 0) * 1024) "\n\t"                     \
                 "ds_read_b128 %1, %8 offset:" CH_STR(((OFFK) + 1) * 1024) "\n\t"                     \
                 "ds_read_b128 %2, %8 offset:" CH_STR(((OFFK) + 2) * 1024) "\n\t"                     \
                 "ds_read_b128 %3, %8 offset:" CH_STR(((OFFK) + 3) * 1024) "\n\t"                     \
                 "ds_read_b128 %4, %8 offset:" CH_STR(((OFFK) + 4) * 1024) "\n\t"                     \
                 "ds_read_b128 %5, %8 offset:" CH_STR(((OFFK) + 5) * 1024) "\n\t"                     \
                 "ds_read_b128 %6, %8 offset:" CH_STR(((OFFK) + 6) * 1024) "\n\t"                     \
                 "ds_read_b128 %7, %8 offset:" CH_STR(((OFFK) + 7) * 1024)                            \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]),      \
                   "=&v"(F[6]), "=&v"(F[7])                                                           \
                 : "v"(addr)                                                                          \
                 : "memory")
#define CH_WAIT8(F)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                               \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]) \
                 :: "memory")
// four consecutive-channel quads of a per-channel table for this lane: channels c0 + 8g + 4*half + (0..3), g = 0..3.
// `dep` is not used by the instructions: it orders the read after the arithmetic that produced it, which keeps the
// compiler from hoisting every table read of a phase above the phase's arithmetic (and spilling what it read).
__device__ __forceinline__ void ch_tab4_nowait(unsigned addr, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:32\n\t"
                 "ds_read_b128 %2, %4 offset:64\n\t"
                 "ds_read_b128 %3, %4 offset:96"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(addr)
                 : "memory");
}
// (the same with `dep`: the first reads of a LayerNorm name its last statistic, so that they - and, volatile statements
// keeping their order, every later table read - stay behind the statistics sweep.  Hoisted above it, two tiles' worth of
// PENDING destination registers sat in the middle of a sweep that wants 128 temporaries, and the compiler moved them out
// of the way before their data had arrived: tools/pending_reg_check.py finds such accesses in the assembly.)
template <typename D>
__device__ __forceinline__ void ch_tab4_nowait_after(unsigned addr, const D& dep, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:32\n\t"
                 "ds_read_b128 %2, %4 offset:64\n\t"
                 "ds_read_b128 %3, %4 offset:96"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(addr), "v"(dep)
                 : "memory");
}
template <typename D>
__device__ __forceinline__ void ch_tab4(unsigned addr, const D& dep, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:32\n\t"
                 "ds_read_b128 %2, %4 offset:64\n\t"
                 "ds_read_b128 %3, %4 offset:96\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(addr), "v"(dep)
                 : "memory");
}

#define CH_SB0_ __builtin_amdgcn_sched_barrier(0)
// wait until at most N of this wave's LDS reads are outstanding; the listed values are (re-)defined here, so arithmetic on
// them cannot move above the wait
#define CH_LGKM_WAIT4(N, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")

// acc (+ per-channel add from the table) -> LayerNorm over the 256 channels of each row -> bf16 B operands.
// Row m lives in lanes m and m+32 (128 channels each): statistics need one exchange across the half-waves.
// The per-channel vectors of a 32-channel tile (gain, offset, optionally a bias added to acc afterwards: ADD) are read one
// tile ahead of their use (round 2 read them tile by tile with a full wait each: sixteen exposed LDS latencies per LayerNorm).
template <bool ADD>
__device__ __forceinline__ void ch_layernorm_pack(f32x16 (&acc)[8], unsigned tab_lane, int off_a, int off_b, int off_add,
                                                  float eps, bf16x8 (&bop)[16]) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[nt][r];
    s += __shfl_xor(s, 32);
    const float mean = s / (float)CH_D;
    float ss = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss = fmaf(acc[nt][r] - mean, acc[nt][r] - mean, ss);
    ss += __shfl_xor(ss, 32);
    const float inv = 1.0f / (sqrtf(ss / (float)(CH_D - 1)) + eps);  // the bf16 rounding below dwarfs x/d vs x*(1/d)
    f32x4 ga[2][4], be[2][4], ad[2][4];
    CH_SB0_;
    ch_tab4_nowait_after(tab_lane + off_a * 4, inv, ga[0][0], ga[0][1], ga[0][2], ga[0][3]);
    ch_tab4_nowait(tab_lane + off_b * 4, be[0][0], be[0][1], be[0][2], be[0][3]);
    if constexpr (ADD) ch_tab4_nowait(tab_lane + off_add * 4, ad[0][0], ad[0][1], ad[0][2], ad[0][3]);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int c = nt & 1, n = c ^ 1;
        if (nt + 1 < 8) {  // the next tile's vectors go out before this tile's are waited for
            ch_tab4_nowait(tab_lane + (off_a + 32 * (nt + 1)) * 4, ga[n][0], ga[n][1], ga[n][2], ga[n][3]);
            ch_tab4_nowait(tab_lane + (off_b + 32 * (nt + 1)) * 4, be[n][0], be[n][1], be[n][2], be[n][3]);
            if constexpr (ADD) ch_tab4_nowait(tab_lane + (off_add + 32 * (nt + 1)) * 4, ad[n][0], ad[n][1], ad[n][2], ad[n][3]);
            if constexpr (ADD) {
                CH_LGKM_WAIT4(12, ga[c][0], ga[c][1], ga[c][2], ga[c][3]);
                CH_LGKM_WAIT4(12, be[c][0], be[c][1], be[c][2], be[c][3]);
                CH_LGKM_WAIT4(12, ad[c][0], ad[c][1], ad[c][2], ad[c][3]);
            } else {
                CH_LGKM_WAIT4(8, ga[c][0], ga[c][1], ga[c][2], ga[c][3]);
                CH_LGKM_WAIT4(8, be[c][0], be[c][1], be[c][2], be[c][3]);
            }
        } else {
            CH_LGKM_WAIT4(0, ga[c][0], ga[c][1], ga[c][2], ga[c][3]);
            CH_LGKM_WAIT4(0, be[c][0], be[c][1], be[c][2], be[c][3]);
            if constexpr (ADD) CH_LGKM_WAIT4(0, ad[c][0], ad[c][1], ad[c][2], ad[c][3]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                bop[2 * nt + (r >> 3)][r & 7] = (bf16)fmaf(ga[c][g][e] * (acc[nt][r] - mean), inv, be[c][g][e]);
                if constexpr (ADD) acc[nt][r] += ad[c][g][e];
            }
        // materialise this tile's operands here: without the pin the compiler defers half of the arithmetic to the
        // operands' first use and spills the table values it still needs (scratch reloads drain the DMA queue)
        asm volatile("" : "+v"(bop[2 * nt]), "+v"(bop[2 * nt + 1]));
        if constexpr (ADD) asm volatile("" : "+a"(acc[nt]));
    }
}

__device__ __forceinline__ void ch_add_channel(f32x16 (&acc)[8], unsigned tab_lane, int off) {
    f32x4 b[2][4];
    ch_tab4_nowait(tab_lane + off * 4, b[0][0], b[0][1], b[0][2], b[0][3]);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int c = nt & 1, n = c ^ 1;
        if (nt + 1 < 8) {
            ch_tab4_nowait(tab_lane + (off + 32 * (nt + 1)) * 4, b[n][0], b[n][1], b[n][2], b[n][3]);
            CH_LGKM_WAIT4(4, b[c][0], b[c][1], b[c][2], b[c][3]);
        } else {
            CH_LGKM_WAIT4(0, b[c][0], b[c][1], b[c][2], b[c][3]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][4 * g + e] += b[c][g][e];
        asm volatile("" : "+a"(acc[nt]));
    }
}

// The wave's 32 rows of x (32 pieces of 16 bytes into AGPRs) and, with CTX, of ctx (16 pieces into VGPRs) in ONE asm statement that
// ends with the wait for them.  (They used to be one statement per load, waited for further down: whatever the compiler does to
// such a register between the two - under register pressure it copied whole tiles to other registers to reuse the
// destination - happens before the data has arrived, which it cannot know.)  XS / CS: byte distance of consecutive pieces (32 in
// a row-major matrix, 1024 in the blocked layouts); xb[nt] / cb[q]: address of piece 4 nt / 4 q.
template <int XS, int CS, bool CTX>
__device__ __forceinline__ void ch_load_rows(f32x4 (&xr)[32], bf16x8 (&bop)[16], const float* const (&xb)[8], const bf16* const (&cb)[4]) {
    if constexpr (CTX) {
        asm volatile("global_load_dwordx4 %[x0], %[xb0], off\n\t"
                     "global_load_dwordx4 %[x1], %[xb0], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x2], %[xb0], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x3], %[xb0], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x4], %[xb1], off\n\t"
                     "global_load_dwordx4 %[x5], %[xb1], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x6], %[xb1], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x7], %[xb1], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x8], %[xb2], off\n\t"
                     "global_load_dwordx4 %[x9], %[xb2], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x10], %[xb2], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x11], %[xb2], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x12], %[xb3], off\n\t"
                     "global_load_dwordx4 %[x13], %[xb3], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x14], %[xb3], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x15], %[xb3], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x16], %[xb4], off\n\t"
                     "global_load_dwordx4 %[x17], %[xb4], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x18], %[xb4], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x19], %[xb4], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x20], %[xb5], off\n\t"
                     "global_load_dwordx4 %[x21], %[xb5], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x22], %[xb5], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x23], %[xb5], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x24], %[xb6], off\n\t"
                     "global_load_dwordx4 %[x25], %[xb6], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x26], %[xb6], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x27], %[xb6], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x28], %[xb7], off\n\t"
                     "global_load_dwordx4 %[x29], %[xb7], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x30], %[xb7], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x31], %[xb7], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[c0], %[cb0], off\n\t"
                     "global_load_dwordx4 %[c1], %[cb0], off offset:%[co1]\n\t"
                     "global_load_dwordx4 %[c2], %[cb0], off offset:%[co2]\n\t"
                     "global_load_dwordx4 %[c3], %[cb0], off offset:%[co3]\n\t"
                     "global_load_dwordx4 %[c4], %[cb1], off\n\t"
                     "global_load_dwordx4 %[c5], %[cb1], off offset:%[co1]\n\t"
                     "global_load_dwordx4 %[c6], %[cb1], off offset:%[co2]\n\t"
                     "global_load_dwordx4 %[c7], %[cb1], off offset:%[co3]\n\t"
                     "global_load_dwordx4 %[c8], %[cb2], off\n\t"
                     "global_load_dwordx4 %[c9], %[cb2], off offset:%[co1]\n\t"
                     "global_load_dwordx4 %[c10], %[cb2], off offset:%[co2]\n\t"
                     "global_load_dwordx4 %[c11], %[cb2], off offset:%[co3]\n\t"
                     "global_load_dwordx4 %[c12], %[cb3], off\n\t"
                     "global_load_dwordx4 %[c13], %[cb3], off offset:%[co1]\n\t"
                     "global_load_dwordx4 %[c14], %[cb3], off offset:%[co2]\n\t"
                     "global_load_dwordx4 %[c15], %[cb3], off offset:%[co3]\n\t"
                     "s_waitcnt vmcnt(0)"
                     : [x0] "=&a"(xr[0]), [x1] "=&a"(xr[1]), [x2] "=&a"(xr[2]), [x3] "=&a"(xr[3]), [x4] "=&a"(xr[4]), [x5] "=&a"(xr[5]), [x6] "=&a"(xr[6]), [x7] "=&a"(xr[7]), [x8] "=&a"(xr[8]), [x9] "=&a"(xr[9]), [x10] "=&a"(xr[10]), [x11] "=&a"(xr[11]), [x12] "=&a"(xr[12]), [x13] "=&a"(xr[13]), [x14] "=&a"(xr[14]), [x15] "=&a"(xr[15]), [x16] "=&a"(xr[16]), [x17] "=&a"(xr[17]), [x18] "=&a"(xr[18]), [x19] "=&a"(xr[19]), [x20] "=&a"(xr[20]), [x21] "=&a"(xr[21]), [x22] "=&a"(xr[22]), [x23] "=&a"(xr[23]), [x24] "=&a"(xr[24]), [x25] "=&a"(xr[25]), [x26] "=&a"(xr[26]), [x27] "=&a"(xr[27]), [x28] "=&a"(xr[28]), [x29] "=&a"(xr[29]), [x30] "=&a"(xr[30]), [x31] "=&a"(xr[31]), [c0] "=&v"(bop[0]), [c1] "=&v"(bop[1]), [c2] "=&v"(bop[2]), [c3] "=&v"(bop[3]), [c4] "=&v"(bop[4]), [c5] "=&v"(bop[5]), [c6] "=&v"(bop[6]), [c7] "=&v"(bop[7]), [c8] "=&v"(bop[8]), [c9] "=&v"(bop[9]), [c10] "=&v"(bop[10]), [c11] "=&v"(bop[11]), [c12] "=&v"(bop[12]), [c13] "=&v"(bop[13]), [c14] "=&v"(bop[14]), [c15] "=&v"(bop[15])
                     : [xb0] "v"(xb[0]), [xb1] "v"(xb[1]), [xb2] "v"(xb[2]), [xb3] "v"(xb[3]), [xb4] "v"(xb[4]), [xb5] "v"(xb[5]), [xb6] "v"(xb[6]), [xb7] "v"(xb[7]), [cb0] "v"(cb[0]), [cb1] "v"(cb[1]), [cb2] "v"(cb[2]), [cb3] "v"(cb[3])
                       , [xo1] "n"(XS), [xo2] "n"(2 * XS), [xo3] "n"(3 * XS), [co1] "n"(CS), [co2] "n"(2 * CS), [co3] "n"(3 * CS)
                     : "memory");
    } else {
        asm volatile("global_load_dwordx4 %[x0], %[xb0], off\n\t"
                     "global_load_dwordx4 %[x1], %[xb0], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x2], %[xb0], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x3], %[xb0], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x4], %[xb1], off\n\t"
                     "global_load_dwordx4 %[x5], %[xb1], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x6], %[xb1], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x7], %[xb1], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x8], %[xb2], off\n\t"
                     "global_load_dwordx4 %[x9], %[xb2], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x10], %[xb2], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x11], %[xb2], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x12], %[xb3], off\n\t"
                     "global_load_dwordx4 %[x13], %[xb3], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x14], %[xb3], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x15], %[xb3], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x16], %[xb4], off\n\t"
                     "global_load_dwordx4 %[x17], %[xb4], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x18], %[xb4], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x19], %[xb4], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x20], %[xb5], off\n\t"
                     "global_load_dwordx4 %[x21], %[xb5], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x22], %[xb5], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x23], %[xb5], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x24], %[xb6], off\n\t"
                     "global_load_dwordx4 %[x25], %[xb6], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x26], %[xb6], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x27], %[xb6], off offset:%[xo3]\n\t"
                     "global_load_dwordx4 %[x28], %[xb7], off\n\t"
                     "global_load_dwordx4 %[x29], %[xb7], off offset:%[xo1]\n\t"
                     "global_load_dwordx4 %[x30], %[xb7], off offset:%[xo2]\n\t"
                     "global_load_dwordx4 %[x31], %[xb7], off offset:%[xo3]\n\t"
                     "s_waitcnt vmcnt(0)"
                     : [x0] "=&a"(xr[0]), [x1] "=&a"(xr[1]), [x2] "=&a"(xr[2]), [x3] "=&a"(xr[3]), [x4] "=&a"(xr[4]), [x5] "=&a"(xr[5]), [x6] "=&a"(xr[6]), [x7] "=&a"(xr[7]), [x8] "=&a"(xr[8]), [x9] "=&a"(xr[9]), [x10] "=&a"(xr[10]), [x11] "=&a"(xr[11]), [x12] "=&a"(xr[12]), [x13] "=&a"(xr[13]), [x14] "=&a"(xr[14]), [x15] "=&a"(xr[15]), [x16] "=&a"(xr[16]), [x17] "=&a"(xr[17]), [x18] "=&a"(xr[18]), [x19] "=&a"(xr[19]), [x20] "=&a"(xr[20]), [x21] "=&a"(xr[21]), [x22] "=&a"(xr[22]), [x23] "=&a"(xr[23]), [x24] "=&a"(xr[24]), [x25] "=&a"(xr[25]), [x26] "=&a"(xr[26]), [x27] "=&a"(xr[27]), [x28] "=&a"(xr[28]), [x29] "=&a"(xr[29]), [x30] "=&a"(xr[30]), [x31] "=&a"(xr[31])
                     : [xb0] "v"(xb[0]), [xb1] "v"(xb[1]), [xb2] "v"(xb[2]), [xb3] "v"(xb[3]), [xb4] "v"(xb[4]), [xb5] "v"(xb[5]), [xb6] "v"(xb[6]), [xb7] "v"(xb[7])
                       , [xo1] "n"(XS), [xo2] "n"(2 * XS), [xo3] "n"(3 * XS)
                     : "memory");
    }
}

// The F8 form of LayerNorm 1: the normalised row leaves as e4m3 bytes in the B-operand order of the K = 64 MFMA - of the 64
// contraction indices of step kk a lane half h holds 32: byte 16 (nt & 1) + r of b8[nt] pairs (nt = 2 kk, 2 kk + 1) is channel
// 32 nt + 8 (r >> 2) + 4 h + (r & 3), r = 0..15, i.e. accumulator register r of tile nt.  The activation scale (x16, a power of
// two) is folded into the gain and offset vectors of the table at pack time; values saturate at +-448 (v_med3_f32) as in the
// unfused fp8 path (rowops.hip: layernorm_kernel<fp8_t>).  b2 is added to the accumulators in the same sweep.
__device__ __forceinline__ void ch_layernorm_pack8(f32x16 (&acc)[8], unsigned tab_lane, int off_a, int off_b, int off_add,
                                                   float eps, v4i (&b8)[8]) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[nt][r];
    s += __shfl_xor(s, 32);
    const float mean = s / (float)CH_D;
    float ss = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss = fmaf(acc[nt][r] - mean, acc[nt][r] - mean, ss);
    ss += __shfl_xor(ss, 32);
    const float inv = 1.0f / (sqrtf(ss / (float)(CH_D - 1)) + eps);
    f32x4 ga[2][4], be[2][4], ad[2][4];
    CH_SB0_;
    ch_tab4_nowait_after(tab_lane + off_a * 4, inv, ga[0][0], ga[0][1], ga[0][2], ga[0][3]);
    ch_tab4_nowait(tab_lane + off_b * 4, be[0][0], be[0][1], be[0][2], be[0][3]);
    ch_tab4_nowait(tab_lane + off_add * 4, ad[0][0], ad[0][1], ad[0][2], ad[0][3]);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int c = nt & 1, n = c ^ 1;
        if (nt + 1 < 8) {
            ch_tab4_nowait(tab_lane + (off_a + 32 * (nt + 1)) * 4, ga[n][0], ga[n][1], ga[n][2], ga[n][3]);
            ch_tab4_nowait(tab_lane + (off_b + 32 * (nt + 1)) * 4, be[n][0], be[n][1], be[n][2], be[n][3]);
            ch_tab4_nowait(tab_lane + (off_add + 32 * (nt + 1)) * 4, ad[n][0], ad[n][1], ad[n][2], ad[n][3]);
            CH_LGKM_WAIT4(12, ga[c][0], ga[c][1], ga[c][2], ga[c][3]);
            CH_LGKM_WAIT4(12, be[c][0], be[c][1], be[c][2], be[c][3]);
            CH_LGKM_WAIT4(12, ad[c][0], ad[c][1], ad[c][2], ad[c][3]);
        } else {
            CH_LGKM_WAIT4(0, ga[c][0], ga[c][1], ga[c][2], ga[c][3]);
            CH_LGKM_WAIT4(0, be[c][0], be[c][1], be[c][2], be[c][3]);
            CH_LGKM_WAIT4(0, ad[c][0], ad[c][1], ad[c][2], ad[c][3]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __builtin_amdgcn_fmed3f(fmaf(ga[c][g][e] * (acc[nt][4 * g + e] - mean), inv, be[c][g][e]), -448.f, 448.f);
                acc[nt][4 * g + e] += ad[c][g][e];
            }
            // (the first conversion's "old" operand is whatever the register held: both halves are written)
            b8[nt][g] = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], b8[nt][g], false);
            b8[nt][g] = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], b8[nt][g], true);
        }
        asm volatile("" : "+v"(b8[nt]));
        asm volatile("" : "+a"(acc[nt]));
    }
}

// "every LDS read of this wave has landed", with both fragment sets as operands: the last block of a stage requests fragments
// nobody uses (the next unit's first half) - dead values to the compiler, whose registers it may hand to something else while
// the reads are still on their way unless they are kept alive up to a wait
#define CH_LAND_FRAGMENTS(EXTRA)                                                                                \
    asm volatile(EXTRA "s_waitcnt lgkmcnt(0)"                                                                  \
                 : "+v"(Fa[0]), "+v"(Fa[1]), "+v"(Fa[2]), "+v"(Fa[3]), "+v"(Fa[4]), "+v"(Fa[5]), "+v"(Fa[6]), "+v"(Fa[7]),     \
                   "+v"(Fb[0]), "+v"(Fb[1]), "+v"(Fb[2]), "+v"(Fb[3]), "+v"(Fb[4]), "+v"(Fb[5]), "+v"(Fb[6]), "+v"(Fb[7])      \
                 :: "memory")

// ---- hand-scheduled half-unit blocks ------------------------------------------------------------------------
// One wave per SIMD issues at most one instruction every ~4 cycles and a 32x32x16 MFMA occupies the matrix pipe for
// 32: everything that is not an MFMA has to be issued in the gaps BETWEEN MFMAs or it runs with the pipe idle (a
// first version that left address arithmetic, waits and branches between compiler-scheduled MFMA clusters spent
// 60 cycles per MFMA).  So:
//   * the stream is consumed in GROUPS of 8 units = one trip round the ring: slot, LDS offsets, DMA destinations and
//     source offsets of every block are compile-time constants, and the only per-group work is two base pointers;
//   * a block = 8 MFMAs on the half-unit already in registers; the 8 ds_reads of the NEXT half-unit go two per gap
//     behind MFMAs 0-3 (so their latency is covered by MFMAs 4-7), the wave's four refill DMAs behind MFMAs 4-7;
//   * second-half blocks of EVEN positions start with: own quarters of units u+1 and u+2 landed (counted vmcnt) ->
//     s_barrier (publishes both; every wave has consumed unit u-1) -> M0 <- destination of this wave's quarter of unit
//     u+7's slot; those of odd positions only set M0 (CH_PRE_BE / CH_PRE_BO).
// The compiler pads its own MFMA -> vector-read sequences with wait states (s_nop 11 for this MFMA) but cannot see
// into asm: a block whose accumulators the compiler may touch next ends with them itself (CH_DRAIN).
#define CH_MF CN_MFMA16_ASM
#define CH_DRAIN "s_nop 13\n\t"
#define CH_NODRAIN ""
#ifdef CH_EXP_NO_READ  // timing experiment: the blocks without their fragment reads (stale registers: wrong results)
#define CH_RD2(n0, n1, K) ""
#else
#define CH_RD2(n0, n1, K) "ds_read_b128 %[" #n0 "], %[ra] offset:" CH_STR((K) * 1024) "\n\t"                   \
                          "ds_read_b128 %[" #n1 "], %[ra] offset:" CH_STR(((K) + 1) * 1024) "\n\t"
#endif
#define CH_PRE_A "s_waitcnt lgkmcnt(0)\n\t"
// first W2 block of a hidden tile: the eight fragment reads are older than the four bias-table reads of the NEXT tile that were
// issued behind the ReLU section (LDS returns in order) - waiting for those as well exposed their whole latency once per tile
// (FFN loop 95.9k -> 91.4k cycles)
#define CH_PRE_A4 "s_waitcnt lgkmcnt(4)\n\t"
// One barrier per TWO units (a barrier costs the matrix pipe ~80 cycles): the second-half block of an EVEN position waits for
// this wave's quarters of the next two units (vmcnt(16): four units of requests may stay outstanding) and its barrier publishes
// both; an odd position has neither.  Slot reuse needs no more: the requests of positions k and k + 1 go into the slots of units
// whose last reads lie before position k's barrier.
#define CH_PRE_BO(M0OFF) "s_add_u32 m0, %[m0w], " CH_STR(M0OFF) "\n\ts_waitcnt lgkmcnt(0)\n\t"
#ifdef CH_EXP_NO_BARRIER  // timing experiment, only together with CH_EXP_NO_DMA and CH_EXP_NO_READ (nothing left to order)
#define CH_PRE_BE(M0OFF) CH_PRE_BO(M0OFF)
#else
#define CH_PRE_BE(M0OFF) "s_waitcnt vmcnt(16)\n\ts_barrier\n\ts_add_u32 m0, %[m0w], " CH_STR(M0OFF) "\n\ts_waitcnt lgkmcnt(0)\n\t"
#endif
#define CH_PRE_B_0 CH_PRE_BE
#define CH_PRE_B_1 CH_PRE_BO
#define CH_PRE_B_2 CH_PRE_BE
#define CH_PRE_B_3 CH_PRE_BO
#define CH_PRE_B_4 CH_PRE_BE
#define CH_PRE_B_5 CH_PRE_BO
#define CH_PRE_B_6 CH_PRE_BE
#define CH_PRE_B_7 CH_PRE_BO
#ifdef CH_EXP_NO_DMA  // timing experiment (tools/chain_stamps.py): the blocks without their refill requests (stale ring: wrong results)
#define CH_DMA0(SOFF) "v_add_u32 %[tv], " CH_STR(SOFF) ", %[vo]\n\t"
#define CH_DMA1 ""
#define CH_DMA2 ""
#define CH_DMA3 ""
#else
#define CH_DMA0(SOFF) "v_add_u32 %[tv], " CH_STR(SOFF) ", %[vo]\n\tglobal_load_lds_dwordx4 %[tv], %[sb]\n\t"
#define CH_DMA1 "global_load_lds_dwordx4 %[tv], %[sb] offset:1024\n\t"
#define CH_DMA2 "global_load_lds_dwordx4 %[tv], %[sb] offset:2048\n\t"
#define CH_DMA3 "global_load_lds_dwordx4 %[tv], %[sb] offset:3072\n\t"
#endif
// single accumulator (CIO/CC: "+"/"=&" and "v"/"a"; C0 = "0" starts the accumulation, "%[c]" continues it), eight B
// operands B[BO..BO+7]; reads the next 8 fragments from RA at RK KiB
#define CH_BLK1(CIO, CC, C0, acc, Fc, B, BO, Fn, RA, RK, PRE, D0, D1, D2, D3, SB, POST)                        \
    asm volatile(PRE                                                                                           \
                 CH_MF "%[c], %[f0], %[b0], " C0 "\n\t" CH_RD2(n0, n1, (RK) + 0)                               \
                 CH_MF "%[c], %[f1], %[b1], %[c]\n\t" CH_RD2(n2, n3, (RK) + 2)                                 \
                 CH_MF "%[c], %[f2], %[b2], %[c]\n\t" CH_RD2(n4, n5, (RK) + 4)                                 \
                 CH_MF "%[c], %[f3], %[b3], %[c]\n\t" CH_RD2(n6, n7, (RK) + 6)                                 \
                 CH_MF "%[c], %[f4], %[b4], %[c]\n\t" D0                                                       \
                 CH_MF "%[c], %[f5], %[b5], %[c]\n\t" D1                                                       \
                 CH_MF "%[c], %[f6], %[b6], %[c]\n\t" D2                                                       \
                 CH_MF "%[c], %[f7], %[b7], %[c]\n\t" D3 POST                                                  \
                 : [c] CIO CC(acc), [n0] "=&v"(Fn[0]), [n1] "=&v"(Fn[1]), [n2] "=&v"(Fn[2]), [n3] "=&v"(Fn[3]), \
                   [n4] "=&v"(Fn[4]), [n5] "=&v"(Fn[5]), [n6] "=&v"(Fn[6]), [n7] "=&v"(Fn[7]), [tv] "=&v"(tv)  \
                 : [f0] "v"(Fc[0]), [f1] "v"(Fc[1]), [f2] "v"(Fc[2]), [f3] "v"(Fc[3]), [f4] "v"(Fc[4]),        \
                   [f5] "v"(Fc[5]), [f6] "v"(Fc[6]), [f7] "v"(Fc[7]), [b0] "v"(B[(BO) + 0]), [b1] "v"(B[(BO) + 1]), \
                   [b2] "v"(B[(BO) + 2]), [b3] "v"(B[(BO) + 3]), [b4] "v"(B[(BO) + 4]), [b5] "v"(B[(BO) + 5]), \
                   [b6] "v"(B[(BO) + 6]), [b7] "v"(B[(BO) + 7]), [ra] "v"(RA), [m0w] "s"(m0_wave), [vo] "v"(voff), \
                   [sb] "s"(SB)                                                                                \
                 : "memory", "scc")
// single accumulator acc[K] of eight AGPR accumulators that all stay tied in place (no copies around the block)
#define CH_BLK1A(K, acc, Fc, B, BO, Fn, RA, RK, PRE, D0, D1, D2, D3, SB, POST)                                 \
    asm volatile(PRE                                                                                           \
                 CH_MF "%[c" #K "], %[f0], %[b0], %[c" #K "]\n\t" CH_RD2(n0, n1, (RK) + 0)                       \
                 CH_MF "%[c" #K "], %[f1], %[b1], %[c" #K "]\n\t" CH_RD2(n2, n3, (RK) + 2)                       \
                 CH_MF "%[c" #K "], %[f2], %[b2], %[c" #K "]\n\t" CH_RD2(n4, n5, (RK) + 4)                       \
                 CH_MF "%[c" #K "], %[f3], %[b3], %[c" #K "]\n\t" CH_RD2(n6, n7, (RK) + 6)                       \
                 CH_MF "%[c" #K "], %[f4], %[b4], %[c" #K "]\n\t" D0                                           \
                 CH_MF "%[c" #K "], %[f5], %[b5], %[c" #K "]\n\t" D1                                           \
                 CH_MF "%[c" #K "], %[f6], %[b6], %[c" #K "]\n\t" D2                                           \
                 CH_MF "%[c" #K "], %[f7], %[b7], %[c" #K "]\n\t" D3 POST                                      \
                 : [c0] "+a"(acc[0]), [c1] "+a"(acc[1]), [c2] "+a"(acc[2]), [c3] "+a"(acc[3]), [c4] "+a"(acc[4]), \
                   [c5] "+a"(acc[5]), [c6] "+a"(acc[6]), [c7] "+a"(acc[7]), [n0] "=&v"(Fn[0]), [n1] "=&v"(Fn[1]), \
                   [n2] "=&v"(Fn[2]), [n3] "=&v"(Fn[3]), [n4] "=&v"(Fn[4]), [n5] "=&v"(Fn[5]), [n6] "=&v"(Fn[6]), \
                   [n7] "=&v"(Fn[7]), [tv] "=&v"(tv)                                                           \
                 : [f0] "v"(Fc[0]), [f1] "v"(Fc[1]), [f2] "v"(Fc[2]), [f3] "v"(Fc[3]), [f4] "v"(Fc[4]),        \
                   [f5] "v"(Fc[5]), [f6] "v"(Fc[6]), [f7] "v"(Fc[7]), [b0] "v"(B[(BO) + 0]), [b1] "v"(B[(BO) + 1]), \
                   [b2] "v"(B[(BO) + 2]), [b3] "v"(B[(BO) + 3]), [b4] "v"(B[(BO) + 4]), [b5] "v"(B[(BO) + 5]), \
                   [b6] "v"(B[(BO) + 6]), [b7] "v"(B[(BO) + 7]), [ra] "v"(RA), [m0w] "s"(m0_wave), [vo] "v"(voff), \
                   [sb] "s"(SB)                                                                                \
                 : "memory", "scc")
// eight accumulators acc[0..7] (AGPRs), one B operand
#define CH_BLK2(acc, Fc, pbv, Fn, RA, RK, PRE, D0, D1, D2, D3, SB, POST)                                       \
    asm volatile(PRE                                                                                           \
                 CH_MF "%[c0], %[f0], %[pb], %[c0]\n\t" CH_RD2(n0, n1, (RK) + 0)                               \
                 CH_MF "%[c1], %[f1], %[pb], %[c1]\n\t" CH_RD2(n2, n3, (RK) + 2)                               \
                 CH_MF "%[c2], %[f2], %[pb], %[c2]\n\t" CH_RD2(n4, n5, (RK) + 4)                               \
                 CH_MF "%[c3], %[f3], %[pb], %[c3]\n\t" CH_RD2(n6, n7, (RK) + 6)                               \
                 CH_MF "%[c4], %[f4], %[pb], %[c4]\n\t" D0                                                     \
                 CH_MF "%[c5], %[f5], %[pb], %[c5]\n\t" D1                                                     \
                 CH_MF "%[c6], %[f6], %[pb], %[c6]\n\t" D2                                                     \
                 CH_MF "%[c7], %[f7], %[pb], %[c7]\n\t" D3 POST                                                \
                 : [c0] "+a"(acc[0]), [c1] "+a"(acc[1]), [c2] "+a"(acc[2]), [c3] "+a"(acc[3]), [c4] "+a"(acc[4]), \
                   [c5] "+a"(acc[5]), [c6] "+a"(acc[6]), [c7] "+a"(acc[7]), [n0] "=&v"(Fn[0]), [n1] "=&v"(Fn[1]), \
                   [n2] "=&v"(Fn[2]), [n3] "=&v"(Fn[3]), [n4] "=&v"(Fn[4]), [n5] "=&v"(Fn[5]), [n6] "=&v"(Fn[6]), \
                   [n7] "=&v"(Fn[7]), [tv] "=&v"(tv)                                                           \
                 : [f0] "v"(Fc[0]), [f1] "v"(Fc[1]), [f2] "v"(Fc[2]), [f3] "v"(Fc[3]), [f4] "v"(Fc[4]),        \
                   [f5] "v"(Fc[5]), [f6] "v"(Fc[6]), [f7] "v"(Fc[7]), [pb] "v"(pbv), [ra] "v"(RA),             \
                   [m0w] "s"(m0_wave), [vo] "v"(voff), [sb] "s"(SB)                                            \
                 : "memory", "scc")
// One MFMA per asm statement (single accumulator X in VGPRs), optionally with the two fragment reads that go into its gap: the
// first-half block of a W1 position is written as eight of these with pieces of C++ (the previous hidden tile's ReLU + bf16 pack)
// between them - sched_barrier(0) on both sides of every piece pins the instruction order, so the VALU work of tile t rides in the
// MFMA gaps of tile t + 1 instead of sitting, with the matrix pipe idle, between the two products of tile t (320 cycles per 1024
// of MFMA in round 2's stamps).
#define CH_M1R(PRE, X, F, B, N0, N1, RA, K)                                                                    \
    asm volatile(PRE CH_MF "%[c], %[f], %[b], %[c]\n\t" CH_RD2(n0, n1, K)                                      \
                 : [c] "+v"(X), [n0] "=&v"(N0), [n1] "=&v"(N1)                                                 \
                 : [f] "v"(F), [b] "v"(B), [ra] "v"(RA)                                                        \
                 : "memory")
#define CH_M1(X, F, B) asm volatile(CH_MF "%[c], %[f], %[b], %[c]" : [c] "+v"(X) : [f] "v"(F), [b] "v"(B) : "memory")
#define CH_SB0 __builtin_amdgcn_sched_barrier(0)

// ---- F8 form of the feed-forward blocks: v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3 operands (K = 64 per instruction, 64 cycles:
// twice the bf16 instruction's products per cycle - profiles/r03s_mfma_f8_probe.txt).  A 16-KiB unit is then 64 hidden units of
// W1 (or 64 contraction indices of W2) = 8 MFMAs, so a unit takes the matrix pipe as long as a bf16 unit does, with the same
// sixteen fragment reads and the same refill requests: the ring, the barriers and the counted waits carry over unchanged.  An
// operand is 8 VGPRs = two 16-byte fragment pieces (lane half h holds bytes 32 h .. 32 h + 31 of the 64 contraction indices;
// which index a byte stands for is free as long as both operands agree - pack_chain uses the order the accumulator layout
// implies).  The pieces are read into separate 4-register values and joined by the compiler (no copies: it allocates the pair
// as one 8-register tuple).  The scale operands are per-tensor powers of two (E8M0 bytes, the same on every lane).
#define CH8_MF "v_mfma_scale_f32_32x32x64_f8f6f4 "
#define CH8_SC ", %[qa], %[qb] op_sel_hi:[0,0,0]\n\t"
#define CH8_DRAIN "s_nop 15\n\ts_nop 4\n\t"  // a 16-pass MFMA's result is readable by a vector instruction 19 wait states later
#define CH8_JOIN(lo, hi) __builtin_shufflevector(__builtin_bit_cast(v4i, lo), __builtin_bit_cast(v4i, hi), 0, 1, 2, 3, 4, 5, 6, 7)
#define CH8_A(F, t) CH8_JOIN(F[2 * (t)], F[2 * (t) + 1])
// one MFMA (accumulator X in VGPRs) and, in its gap, four fragment reads of the next half-unit (pieces K .. K + 3 at RA)
#define CH8_M1R(PRE, X, A, B, N0, N1, N2, N3, RA, K, QA, QB)                                                   \
    asm volatile(PRE CH8_MF "%[c], %[a], %[b], %[c]" CH8_SC CH_RD2(n0, n1, K) CH_RD2(n2, n3, (K) + 2)          \
                 : [c] "+v"(X), [n0] "=&v"(N0), [n1] "=&v"(N1), [n2] "=&v"(N2), [n3] "=&v"(N3)                 \
                 : [a] "v"(A), [b] "v"(B), [ra] "v"(RA), [qa] "v"(QA), [qb] "v"(QB), [m0w] "s"(m0_wave)         \
                 : "memory", "scc")
#define CH8_M1(X, A, B, QA, QB)                                                                                \
    asm volatile(CH8_MF "%[c], %[a], %[b], %[c]" CH8_SC : [c] "+v"(X) : [a] "v"(A), [b] "v"(B), [qa] "v"(QA), [qb] "v"(QB) : "memory")
// one MFMA and two refill requests (M0 = their destination is set again in every statement: nothing tells the compiler that
// it is live between two of them)
#define CH8_M1D(M0OFF, X, A, B, D0, D1, SB, QA, QB)                                                            \
    asm volatile("s_add_u32 m0, %[m0w], " CH_STR(M0OFF) "\n\t" CH8_MF "%[c], %[a], %[b], %[c]" CH8_SC D0 D1    \
                 : [c] "+v"(X), [tv] "+v"(tv)                                                                  \
                 : [a] "v"(A), [b] "v"(B), [qa] "v"(QA), [qb] "v"(QB), [vo] "v"(voff), [sb] "s"(SB),            \
                   [m0w] "s"(m0_wave)                                                                          \
                 : "memory", "scc")
// four of the eight AGPR accumulators (operand names C0..C3), one B operand (the activated hidden pair); eight fragment reads
// behind the first two MFMAs, four refill requests behind the last two
#define CH8_BLK2(acc, C0, C1, C2, C3, A0, A1, A2, A3, HB, Fn, RA, RK, PRE, D0, D1, D2, D3, SB, POST, QA, QB)    \
    asm volatile(PRE                                                                                           \
                 CH8_MF "%[" #C0 "], %[a0], %[hb], %[" #C0 "]" CH8_SC CH_RD2(n0, n1, (RK) + 0) CH_RD2(n2, n3, (RK) + 2) \
                 CH8_MF "%[" #C1 "], %[a1], %[hb], %[" #C1 "]" CH8_SC CH_RD2(n4, n5, (RK) + 4) CH_RD2(n6, n7, (RK) + 6) \
                 CH8_MF "%[" #C2 "], %[a2], %[hb], %[" #C2 "]" CH8_SC D0 D1                                    \
                 CH8_MF "%[" #C3 "], %[a3], %[hb], %[" #C3 "]" CH8_SC D2 D3 POST                               \
                 : [c0] "+a"(acc[0]), [c1] "+a"(acc[1]), [c2] "+a"(acc[2]), [c3] "+a"(acc[3]), [c4] "+a"(acc[4]), \
                   [c5] "+a"(acc[5]), [c6] "+a"(acc[6]), [c7] "+a"(acc[7]), [n0] "=&v"(Fn[0]), [n1] "=&v"(Fn[1]), \
                   [n2] "=&v"(Fn[2]), [n3] "=&v"(Fn[3]), [n4] "=&v"(Fn[4]), [n5] "=&v"(Fn[5]), [n6] "=&v"(Fn[6]), \
                   [n7] "=&v"(Fn[7]), [tv] "=&v"(tv)                                                           \
                 : [a0] "v"(A0), [a1] "v"(A1), [a2] "v"(A2), [a3] "v"(A3), [hb] "v"(HB), [ra] "v"(RA),         \
                   [m0w] "s"(m0_wave), [vo] "v"(voff), [sb] "s"(SB), [qa] "v"(QA), [qb] "v"(QB)                \
                 : "memory", "scc")

// Position k of a group: its second half lives in slot k (first-half blocks read it), the first half of the next unit
// in slot k+1; while unit k is consumed, unit k+7 of the stream is requested into slot k-1 - i.e. unit 7 of THIS group
// for k = 0 and unit k-1 of the NEXT group otherwise.  Slots 0-3 are addressed from ra0, 4-7 from ra1 = ra0 + 64 KiB.
//   X(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB)
#define CH_POSITIONS(X)                                                                                        \
    X(0, ra0, 8, ra0, 16, 0x1C000, 0x1C000, sb_cur)  X(1, ra0, 24, ra0, 32, 0x0, 0x0, sb_next)                 \
    X(2, ra0, 40, ra0, 48, 0x4000, 0x4000, sb_next)  X(3, ra0, 56, ra1, 0, 0x8000, 0x8000, sb_next)            \
    X(4, ra1, 8, ra1, 16, 0xC000, 0xC000, sb_next)   X(5, ra1, 24, ra1, 32, 0x10000, 0x10000, sb_next)         \
    X(6, ra1, 40, ra1, 48, 0x14000, 0x14000, sb_next) X(7, ra1, 56, ra0, 0, 0x18000, 0x18000, sb_next)

// SWISH: the feed-forward activation is x * sigmoid(x) (the conformer's macaron halves) instead of ReLU
// F8: the two feed-forward products run on e4m3 operands (BASELINE config 5), everything else as in the bf16 form
template <bool SWISH, bool F8>
__global__ __launch_bounds__(256) void chain_kernel(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int m = blockIdx.x * 128 + 32 * wave + l31;
    const int mc = m < p.M ? m : p.M - 1;
    const bool live = m < p.M;
#define CH_STAMP(i)                                                                              \
    if (p.stamps && (int)blockIdx.x == p.stamp_block && tid == 0) ch_stamps[i] = (long long)__builtin_amdgcn_s_memtime();
    CH_STAMP(0)
    if (p.stamps && (int)blockIdx.x == p.stamp_block && tid == 0) ch_stamps[10] = (long long)__builtin_amdgcn_s_memrealtime();
    // groups of 8 units in consumption order: [output projection] [FFN: ffn_tiles / 4 groups, walked in a rotation of
    // its own by every workgroup - the sum over tiles is order-free, and the workgroups of a launch then do not pull the
    // same L2 lines at the same moment] [tail projection: tail_tiles / 8 groups]
    // (F8: a unit is 64 hidden units - half as many units)
    const int G_OUT = p.ctx ? 1 : 0, G_FFN = F8 ? p.ffn_tiles >> 3 : p.ffn_tiles >> 2, G_TAIL = p.tail_tiles >> 3;
    const int NG = G_OUT + G_FFN + G_TAIL;
    // (Every workgroup used to walk the FFN groups in a rotation of its own, to keep the workgroups from pulling the same L2
    // lines at the same moment.  It bought nothing measurable (A/B on one box), and it made a row's fp32 accumulation order
    // depend on which workgroup the row lands in - so a batch decoded in a merged engine pass differed in its scores' last
    // digits from the same batch decoded alone.  CH_ROTATE keeps the old walk for experiments.)
#ifdef CH_ROTATE
    const int rotg = G_FFN ? (int)((blockIdx.x * 5u) % (unsigned)G_FFN) : 0;
#else
    const int rotg = 0;
#endif
    unsigned char* ring = smem + CH_TAB_BYTES;
    // stream group consumed at position g (past the end: the last group again - dummy refills keep the DMA queue depth,
    // and with it every vmcnt of the loop, constant; their slots are never read)
    auto ffn_group = [&](int g) -> int { return g + rotg >= G_FFN ? g + rotg - G_FFN : g + rotg; };
    auto group_base = [&](int g) -> const uint4* {
        if (g > NG - 1) g = NG - 1;
        if (g >= G_OUT && g < G_OUT + G_FFN) g = G_OUT + ffn_group(g - G_OUT);
        return p.wstream + (long long)g * 8 * 1024;
    };

    // ---- prologue: the table and the first seven units of the stream are requested first, then the wave's 32 rows of x
    // (accumulator layout) and ctx (B operands) are loaded and waited for (ch_load_rows): the requests travel meanwhile
    f32x4 xr[32];
    bf16x8 bop[16];
    const int rb = blockIdx.x * 4 + wave, nrb = (p.M + 31) >> 5;  // this wave's 32-row block; blocks that hold rows
    {
        const uint4* ts = reinterpret_cast<const uint4*>(p.tab) + 5 * wave * 64 + lane;
#pragma unroll
        for (int j = 0; j < 5; ++j) CH_DMA(ts + j * 64, smem + (5 * wave + j) * 1024);
    }
    if (NG > 0) {
        const uint4* g0 = group_base(0);
        for (int u = 0; u < CH_RING - 1; ++u) {  // units 0..6 of the first group; unit 7 is requested by position 0
            const uint4* src = g0 + ((long long)u * 16 + 4 * wave) * 64 + lane;
            unsigned char* dst = ring + u * CH_UNIT_BYTES + 4 * wave * 1024;
#pragma unroll
            for (int j = 0; j < 4; ++j) CH_DMA(src + j * 64, dst + j * 1024);
        }
    }
    {
        // piece i = 4 nt + g holds channels 32 nt + 8 g + 4 half + (0..3): row-major it sits at float 8 i of the row,
        // blocked at float 256 i + 4 lane of the row block
#ifdef CH_EXP_HOT_X  // timing experiment: every workgroup reads row block (wave) of x - L2-hot after the first round (wrong results):
                     // what a launch costs when its x rows do not have to come from HBM at the start of a workgroup
        const float* xp = p.x + (long long)wave * 8192 + 4 * lane;
#else
        const float* xp = p.x_in_blk ? p.x + (long long)(rb < nrb ? rb : nrb - 1) * 8192 + 4 * lane
                                     : p.x + (long long)mc * CH_D + 4 * half;
#endif
        const int step = p.x_in_blk ? 256 : 8;
        const float* xb[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) xb[nt] = xp + step * 4 * nt;
        // ctx row-major: the lane's row, 16-byte pieces 32 bytes apart; blocked (the attention kernel's o_blocked form): k-step ks
        // of this wave's row block is 1 KiB contiguous, [lane][16 B] - the lane order of the B operand itself
        const bf16* cp = !p.ctx ? nullptr
                         : p.ctx_blk ? p.ctx + ((long long)(rb < nrb ? rb : nrb - 1) * 16 * 64 + lane) * 8
                                     : p.ctx + (long long)mc * p.ldctx + 8 * half;
        const int cstep = p.ctx_blk ? 512 : 16;
        const bf16* cb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cb[q] = cp + cstep * 4 * q;
        if (!p.ctx) {
            if (p.x_in_blk) ch_load_rows<1024, 32, false>(xr, bop, xb, cb);
            else ch_load_rows<32, 32, false>(xr, bop, xb, cb);
        } else if (p.x_in_blk) {
            if (p.ctx_blk) ch_load_rows<1024, 1024, true>(xr, bop, xb, cb);
            else ch_load_rows<1024, 32, true>(xr, bop, xb, cb);
        } else {
            if (p.ctx_blk) ch_load_rows<32, 1024, true>(xr, bop, xb, cb);
            else ch_load_rows<32, 32, true>(xr, bop, xb, cb);
        }
    }

    f32x16 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][4 * g + e] = xr[4 * nt + g][e];
    CH_STAMP(1)

    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned ra0 = ring_lds + lane * 16, ra1 = ra0 + 4 * CH_UNIT_BYTES;
    const unsigned tab_lane = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + 16 * half);
    const unsigned voff = (unsigned)(lane * 16 + wave * 4096);  // this lane's byte offset inside a unit (piece 4 wave + j: + 1024 j)
    const unsigned m0_wave = __builtin_amdgcn_readfirstlane(ring_lds + wave * 4096);
    unsigned tv = 0;  // scratch VGPR of the blocks (source offset of a refill)

    bf16x8 Fa[8], Fb[8];
    // unit 0 (and the table): every wave's quarter has landed (ch_load_rows ended with vmcnt(0)) -> barrier
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    CH_STAMP(2)
    int gpos = 0;  // group position in consumption order

    // ---- S1: output projection (natural k order; B operands = ctx), accumulating on top of the residual
    if (p.ctx) {
        const uint4* sb_cur = group_base(gpos);
        const uint4* sb_next = group_base(gpos + 1);
#define CH_S1(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB)                                                     \
        CH_BLK1A(k, acc, Fa, bop, 0, Fb, RA_A, RK_A, CH_PRE_A, "", "", "", "", SB, CH_NODRAIN);                \
        CH_BLK1A(k, acc, Fb, bop, 8, Fa, RA_B, RK_B, CH_PRE_B_##k(M0OFF), CH_DMA0(SOFF), CH_DMA1, CH_DMA2, CH_DMA3, SB, CH_NODRAIN);
        // Every stage reads the first half of its first unit itself, right in front of its loop, and waits for it: a fragment
        // set requested by the previous stage's last block would be live (and PENDING, which hipcc cannot see: it spilled such
        // registers right behind the reads) across the compiler-scheduled LayerNorm code in between
        CH_READ8(Fa, ra0, 0);
        CH_WAIT8(Fa);
        asm volatile("s_nop 1" ::: "memory");  // the compiler's copies into the accumulators -> first MFMA
        CH_POSITIONS(CH_S1)
#undef CH_S1
        // (the last block's fragment reads - the next unit's first half - are pending: they land here, before compiler-scheduled
        // code may spill or move their registers)
        CH_LAND_FRAGMENTS("");
        asm volatile(CH_DRAIN : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]),
                     "+a"(acc[6]), "+a"(acc[7]) :: "memory");
        ++gpos;
        ch_add_channel(acc, tab_lane, CT_BO);
    }
    CH_STAMP(3)

    // ---- S3: feed-forward sublayer.  Unit order (pack_chain): W1(0), then W1(t + 1), W2(t) for t = 0 .. n - 2, then W2(n - 1) -
    // software-pipelined by one hidden tile: tile t's bias + ReLU + bf16 pack (VALU) is issued in the MFMA gaps of W1(t + 1), whose
    // accumulator is the other one of two (xh0 / xh1), so W2(t) starts with its operand ready and no MFMA -> VALU drain.  In a
    // group of 8 units the odd positions are W1 products (tiles 4 g + 1 .. 4 g + 4), the even ones W2 (tiles 4 g - 1 .. 4 g + 2);
    // the first group opens with W1(0), the last one ends with W2(n - 1) in place of a W1.
    if constexpr (F8) {
      if (p.ffn_tiles) {
        // ---- S3, F8 form.  LayerNorm 1 -> e4m3 B operands (b2 added to x in the same sweep).  Unit order (pack_chain): W1(0),
        // then W1(p + 1), W2(p) for pairs p = 0 .. n - 2 of 32-wide hidden tiles, then W2(n - 1); a W1 position is eight MFMAs
        // (two tiles x four 64-channel steps) into two accumulators, a W2 position eight (one per 32 output channels) on the
        // pair's 64 activated values, which a lane holds as exactly one B operand: registers of tile 2 p, then of tile 2 p + 1.
        // Pair p is activated (ReLU and saturation in one v_med3, e4m3 pack) in the MFMA gaps of W1(p + 1), one dword per gap.
        v4i b8[8] = {};
        ch_layernorm_pack8(acc, tab_lane, CT_LN1A, CT_LN1B, CT_B2, p.eps, b8);
        CH_STAMP(4)
        const int NP = p.ffn_tiles >> 1;
        f32x16 xa0, xb0, xa1, xb1;
        f32x4 b1v[8];
        v8i hid8 = {};
        const int qa1 = p.f8_q[0], qb1 = p.f8_q[1], qa2 = p.f8_q[2], qb2 = p.f8_q[3];
        auto tabp = [&](int pp) -> unsigned { return tab_lane + (unsigned)((CT_B1 + 64 * (pp < NP ? pp : NP - 1)) * 4); };
#define CH8_TAB8(addr)                                                                                         \
        ch_tab4_nowait(addr, b1v[0], b1v[1], b1v[2], b1v[3]);                                                  \
        ch_tab4_nowait((addr) + 128, b1v[4], b1v[5], b1v[6], b1v[7]);
#define CH8_B(kk) CH8_JOIN(b8[2 * (kk)], b8[2 * (kk) + 1])
        CH8_TAB8(tabp(0))
        CH_READ8(Fa, ra0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]), "+v"(b1v[4]), "+v"(b1v[5]),
                     "+v"(b1v[6]), "+v"(b1v[7]), "+v"(Fa[0]), "+v"(Fa[1]), "+v"(Fa[2]), "+v"(Fa[3]), "+v"(Fa[4]), "+v"(Fa[5]),
                     "+v"(Fa[6]), "+v"(Fa[7]) :: "memory");
        // dword i of the activated pair: values 4 (i & 3) .. + 3 of tile XA (i < 4) or XB
#define CH8_ACT_PIECE(XA, XB, i)                                                                               \
        {                                                                                                      \
            const float v0_ = __builtin_amdgcn_fmed3f(((i) < 4 ? XA : XB)[4 * ((i) & 3) + 0], 0.f, 448.f);     \
            const float v1_ = __builtin_amdgcn_fmed3f(((i) < 4 ? XA : XB)[4 * ((i) & 3) + 1], 0.f, 448.f);     \
            const float v2_ = __builtin_amdgcn_fmed3f(((i) < 4 ? XA : XB)[4 * ((i) & 3) + 2], 0.f, 448.f);     \
            const float v3_ = __builtin_amdgcn_fmed3f(((i) < 4 ? XA : XB)[4 * ((i) & 3) + 3], 0.f, 448.f);     \
            hid8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(v0_, v1_, hid8[i], false);                               \
            hid8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(v2_, v3_, hid8[i], true);                                \
            asm volatile("" : "+v"(hid8));                                                                     \
        }
#define CH8_ACT_GAP(HAS_OLD, XA, XB, i)                                                                        \
        CH_SB0;                                                                                                \
        if constexpr (HAS_OLD) CH8_ACT_PIECE(XA, XB, i)                                                        \
        CH_SB0;
#define CH8_BIAS16(lo) __builtin_shufflevector(__builtin_shufflevector(b1v[lo], b1v[(lo) + 1], 0, 1, 2, 3, 4, 5, 6, 7),   \
                                               __builtin_shufflevector(b1v[(lo) + 2], b1v[(lo) + 3], 0, 1, 2, 3, 4, 5, 6, 7), \
                                               0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
        // W1 position: (XNA, XNB) = the pair's two tiles . xn on top of their biases (the table holds b1 at the hidden scale);
        // the previous pair (XOA, XOB) is activated into hid8 in the gaps; then the next W1 pair's bias reads go out
#define CH8_W1N(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB, XNA, XNB, XOA, XOB, HAS_OLD, PNEXT)                 \
        XNA = CH8_BIAS16(0);                                                                                   \
        XNB = CH8_BIAS16(4);                                                                                   \
        CH8_M1R(CH_PRE_A, XNA, CH8_A(Fa, 0), CH8_B(0), Fb[0], Fb[1], Fb[2], Fb[3], RA_A, (RK_A) + 0, qa1, qb1); \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 0)                                                                      \
        CH8_M1R("", XNA, CH8_A(Fa, 1), CH8_B(1), Fb[4], Fb[5], Fb[6], Fb[7], RA_A, (RK_A) + 4, qa1, qb1);      \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 1)                                                                      \
        CH8_M1(XNA, CH8_A(Fa, 2), CH8_B(2), qa1, qb1);                                                         \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 2)                                                                      \
        CH8_M1(XNA, CH8_A(Fa, 3), CH8_B(3), qa1, qb1);                                                         \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 3)                                                                      \
        CH8_M1R(CH_PRE_B_##k(M0OFF), XNB, CH8_A(Fb, 0), CH8_B(0), Fa[0], Fa[1], Fa[2], Fa[3], RA_B, (RK_B) + 0, qa1, qb1); \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 4)                                                                      \
        CH8_M1R("", XNB, CH8_A(Fb, 1), CH8_B(1), Fa[4], Fa[5], Fa[6], Fa[7], RA_B, (RK_B) + 4, qa1, qb1);      \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 5)                                                                      \
        CH8_M1D(M0OFF, XNB, CH8_A(Fb, 2), CH8_B(2), CH_DMA0(SOFF), CH_DMA1, SB, qa1, qb1);                            \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 6)                                                                      \
        CH8_M1D(M0OFF, XNB, CH8_A(Fb, 3), CH8_B(3), CH_DMA2, CH_DMA3, SB, qa1, qb1);                                  \
        CH8_ACT_GAP(HAS_OLD, XOA, XOB, 7)                                                                      \
        CH8_TAB8(tabp(PNEXT))
        // W2 position: acc[nt] += W2 unit (pair, nt) . hid8; PREA = "eight younger bias reads may still be out" right after a W1
#define CH8_PRE_A8 "s_waitcnt lgkmcnt(8)\n\t"
#define CH8_W2N(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB, PREA, POST)                                        \
        CH8_BLK2(acc, c0, c1, c2, c3, CH8_A(Fa, 0), CH8_A(Fa, 1), CH8_A(Fa, 2), CH8_A(Fa, 3), hid8, Fb, RA_A, RK_A, PREA, "", "", \
                 "", "", SB, CH_NODRAIN, qa2, qb2);                                                            \
        CH8_BLK2(acc, c4, c5, c6, c7, CH8_A(Fb, 0), CH8_A(Fb, 1), CH8_A(Fb, 2), CH8_A(Fb, 3), hid8, Fa, RA_B, RK_B,              \
                 CH_PRE_B_##k(M0OFF), CH_DMA0(SOFF), CH_DMA1, CH_DMA2, CH_DMA3, SB, POST, qa2, qb2);
#define CH8_FFN_GROUP(FIRST, LAST)                                                                             \
        {                                                                                                      \
            const uint4* sb_cur = group_base(gpos);                                                            \
            const uint4* sb_next = group_base(gpos + 1);                                                       \
            const int t0 = 4 * g;                                                                              \
            if constexpr (FIRST) {                                                                             \
                CH8_W1N(0, ra0, 8, ra0, 16, 0x1C000, 0x1C000, sb_cur, xa0, xb0, xa1, xb1, false, 1)            \
                /* two W1 positions in a row: the bias read is consumed by the very next block */             \
                asm volatile(CH8_DRAIN "s_waitcnt lgkmcnt(0)" : "+v"(xa0), "+v"(xb0), "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), \
                             "+v"(b1v[3]), "+v"(b1v[4]), "+v"(b1v[5]), "+v"(b1v[6]), "+v"(b1v[7]), "+v"(Fa[0]), "+v"(Fa[1]),     \
                             "+v"(Fa[2]), "+v"(Fa[3]), "+v"(Fa[4]), "+v"(Fa[5]), "+v"(Fa[6]), "+v"(Fa[7]) :: "memory");          \
            } else {                                                                                           \
                CH8_W2N(0, ra0, 8, ra0, 16, 0x1C000, 0x1C000, sb_cur, CH8_PRE_A8, CH_NODRAIN)                  \
            }                                                                                                  \
            CH8_W1N(1, ra0, 24, ra0, 32, 0x0, 0x0, sb_next, xa1, xb1, xa0, xb0, true, t0 + 2)                  \
            CH8_W2N(2, ra0, 40, ra0, 48, 0x4000, 0x4000, sb_next, CH8_PRE_A8, CH_NODRAIN)                      \
            CH8_W1N(3, ra0, 56, ra1, 0, 0x8000, 0x8000, sb_next, xa0, xb0, xa1, xb1, true, t0 + 3)             \
            CH8_W2N(4, ra1, 8, ra1, 16, 0xC000, 0xC000, sb_next, CH8_PRE_A8, CH_NODRAIN)                       \
            CH8_W1N(5, ra1, 24, ra1, 32, 0x10000, 0x10000, sb_next, xa1, xb1, xa0, xb0, true, t0 + 4)          \
            CH8_W2N(6, ra1, 40, ra1, 48, 0x14000, 0x14000, sb_next, CH8_PRE_A8, CH_NODRAIN)                    \
            if constexpr (!(LAST)) {                                                                           \
                CH8_W1N(7, ra1, 56, ra0, 0, 0x18000, 0x18000, sb_next, xa0, xb0, xa1, xb1, true, t0 + 5)       \
            } else {                                                                                           \
                /* the last pair has no W1 position behind it to hide in (see the bf16 form) */               \
                asm volatile("" : "+v"(xa1), "+v"(xb1));                                                       \
                CH_SB0;                                                                                        \
                CH8_ACT_PIECE(xa1, xb1, 0) CH8_ACT_PIECE(xa1, xb1, 1) CH8_ACT_PIECE(xa1, xb1, 2) CH8_ACT_PIECE(xa1, xb1, 3) \
                CH8_ACT_PIECE(xa1, xb1, 4) CH8_ACT_PIECE(xa1, xb1, 5) CH8_ACT_PIECE(xa1, xb1, 6) CH8_ACT_PIECE(xa1, xb1, 7) \
                CH_SB0;                                                                                        \
                CH8_W2N(7, ra1, 56, ra0, 0, 0x18000, 0x18000, sb_next, CH_PRE_A, CH8_DRAIN "s_waitcnt lgkmcnt(0)\n\t") \
            }                                                                                                  \
            ++g;                                                                                               \
            ++gpos;                                                                                            \
        }
        int g = 0;
        if (G_FFN == 1) {
            CH8_FFN_GROUP(true, true)
        } else {
            CH8_FFN_GROUP(true, false)
            while (g < G_FFN - 1) CH8_FFN_GROUP(false, false)
            CH8_FFN_GROUP(false, true)
        }
#undef CH8_FFN_GROUP
#undef CH8_W1N
#undef CH8_W2N
#undef CH8_ACT_GAP
#undef CH8_ACT_PIECE
        CH_LAND_FRAGMENTS("");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]), "+v"(b1v[4]), "+v"(b1v[5]),
                     "+v"(b1v[6]), "+v"(b1v[7]) :: "memory");
      }
    } else
    if (p.ffn_tiles) {
        // LayerNorm 1 -> B operands; b2 is added to x in the same sweep (once: the W2 products accumulate on top of x + b2)
        ch_layernorm_pack<true>(acc, tab_lane, CT_LN1A, CT_LN1B, CT_B2, p.eps, bop);
        CH_STAMP(4)
        const int NTL = p.ffn_tiles;
        f32x16 xh0, xh1;
        f32x4 b1v[4];
        bf16x8 pb[2];
        auto tabb = [&](int t) -> unsigned { return tab_lane + (unsigned)((CT_B1 + 32 * (t < NTL ? t : NTL - 1)) * 4); };
        ch_tab4_nowait(tabb(0), b1v[0], b1v[1], b1v[2], b1v[3]);
        CH_READ8(Fa, ra0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]), "+v"(Fa[0]), "+v"(Fa[1]), "+v"(Fa[2]),
                     "+v"(Fa[3]), "+v"(Fa[4]), "+v"(Fa[5]), "+v"(Fa[6]), "+v"(Fa[7]) :: "memory");
        // bias + activation + pack of values 2 i, 2 i + 1 of the finished tile XO (ReLU as an integer max on the fp32 bits)
#define CH_ACT_PIECE(XO, i)                                                                                    \
        {                                                                                                      \
            if constexpr (SWISH) {                                                                             \
                const float v0_ = XO[2 * (i)], v1_ = XO[2 * (i) + 1];                                          \
                pb[(i) >> 2][2 * ((i) & 3)] = (bf16)(v0_ * (1.f / (1.f + __expf(-v0_))));                      \
                pb[(i) >> 2][2 * ((i) & 3) + 1] = (bf16)(v1_ * (1.f / (1.f + __expf(-v1_))));                  \
            } else {                                                                                           \
                const int b0_ = __float_as_int(XO[2 * (i)]), b1_ = __float_as_int(XO[2 * (i) + 1]);            \
                pb[(i) >> 2][2 * ((i) & 3)] = (bf16)__int_as_float(b0_ > 0 ? b0_ : 0);                         \
                pb[(i) >> 2][2 * ((i) & 3) + 1] = (bf16)__int_as_float(b1_ > 0 ? b1_ : 0);                     \
            }                                                                                                  \
            asm volatile("" : "+v"(pb[(i) >> 2]));  /* the packed pair is produced HERE, not where W2 first reads it */ \
        }
#define CH_ACT_GAP(HAS_OLD, XO, i)                                                                             \
        CH_SB0;                                                                                                \
        if constexpr (HAS_OLD) CH_ACT_PIECE(XO, i)                                                             \
        CH_SB0;
        // W1 position: XN = W1 tile . xn on top of the tile's bias (the 16 table values ARE the accumulator's initial value),
        // the previous tile XO activated and packed into pb in the first block's gaps; then the next W1 tile's bias read goes
        // out (younger than the fragment reads of the W2 block that follows, which waits with lgkmcnt(4))
#define CH_W1N(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB, XN, XO, HAS_OLD, TNEXT)                             \
        XN = __builtin_shufflevector(__builtin_shufflevector(b1v[0], b1v[1], 0, 1, 2, 3, 4, 5, 6, 7),          \
                                     __builtin_shufflevector(b1v[2], b1v[3], 0, 1, 2, 3, 4, 5, 6, 7),          \
                                     0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);                   \
        CH_M1R(CH_PRE_A, XN, Fa[0], bop[0], Fb[0], Fb[1], RA_A, (RK_A) + 0);                                   \
        CH_ACT_GAP(HAS_OLD, XO, 0)                                                                             \
        CH_M1R("", XN, Fa[1], bop[1], Fb[2], Fb[3], RA_A, (RK_A) + 2);                                         \
        CH_ACT_GAP(HAS_OLD, XO, 1)                                                                             \
        CH_M1R("", XN, Fa[2], bop[2], Fb[4], Fb[5], RA_A, (RK_A) + 4);                                         \
        CH_ACT_GAP(HAS_OLD, XO, 2)                                                                             \
        CH_M1R("", XN, Fa[3], bop[3], Fb[6], Fb[7], RA_A, (RK_A) + 6);                                         \
        CH_ACT_GAP(HAS_OLD, XO, 3)                                                                             \
        CH_M1(XN, Fa[4], bop[4]);                                                                              \
        CH_ACT_GAP(HAS_OLD, XO, 4)                                                                             \
        CH_M1(XN, Fa[5], bop[5]);                                                                              \
        CH_ACT_GAP(HAS_OLD, XO, 5)                                                                             \
        CH_M1(XN, Fa[6], bop[6]);                                                                              \
        CH_ACT_GAP(HAS_OLD, XO, 6)                                                                             \
        CH_M1(XN, Fa[7], bop[7]);                                                                              \
        CH_ACT_GAP(HAS_OLD, XO, 7)                                                                             \
        CH_BLK1("+", "v", "%[c]", XN, Fb, bop, 8, Fa, RA_B, RK_B, CH_PRE_B_##k(M0OFF), CH_DMA0(SOFF), CH_DMA1, CH_DMA2, \
                CH_DMA3, SB, CH_NODRAIN);                                                                      \
        ch_tab4_nowait(tabb(TNEXT), b1v[0], b1v[1], b1v[2], b1v[3]);
        // W2 position: acc[nt] += W2 tile (s, nt) . act(xh); PREA = CH_PRE_A4 right after a W1 position (its bias read is younger
        // than the eight fragment reads), CH_PRE_A otherwise
#define CH_W2N(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB, PREA, POST)                                        \
        CH_BLK2(acc, Fa, pb[0], Fb, RA_A, RK_A, PREA, "", "", "", "", SB, CH_NODRAIN);                         \
        CH_BLK2(acc, Fb, pb[1], Fa, RA_B, RK_B, CH_PRE_B_##k(M0OFF), CH_DMA0(SOFF), CH_DMA1, CH_DMA2, CH_DMA3, SB, POST);
        // One group of 8 units.  FIRST / LAST are compile-time: the three forms are separate straight-line regions (a branch inside
        // the loop body made hipcc shuffle all 128 accumulator registers at its join, every iteration)
#define CH_FFN_GROUP(FIRST, LAST)                                                                              \
        {                                                                                                      \
            const uint4* sb_cur = group_base(gpos);                                                            \
            const uint4* sb_next = group_base(gpos + 1);                                                       \
            const int t0 = 4 * g;                                                                              \
            if constexpr (FIRST) {                                                                             \
                CH_W1N(0, ra0, 8, ra0, 16, 0x1C000, 0x1C000, sb_cur, xh0, xh1, false, 1)                       \
                /* two W1 positions in a row: this bias read is consumed by the very next block - wait for it here; and the */ \
                /* activation pieces of the next position read xh0 one MFMA after its last product was issued (everywhere   */ \
                /* else a whole W2 position lies in between): the MFMA -> VALU wait states the compiler cannot see           */ \
                asm volatile("s_nop 13\n\ts_nop 13\n\ts_waitcnt lgkmcnt(0)" : "+v"(xh0), "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]), "+v"(Fa[0]), \
                             "+v"(Fa[1]), "+v"(Fa[2]), "+v"(Fa[3]), "+v"(Fa[4]), "+v"(Fa[5]), "+v"(Fa[6]), "+v"(Fa[7]) :: "memory"); \
            } else {                                                                                           \
                CH_W2N(0, ra0, 8, ra0, 16, 0x1C000, 0x1C000, sb_cur, CH_PRE_A4, CH_NODRAIN)                    \
            }                                                                                                  \
            CH_W1N(1, ra0, 24, ra0, 32, 0x0, 0x0, sb_next, xh1, xh0, true, t0 + 2)                             \
            CH_W2N(2, ra0, 40, ra0, 48, 0x4000, 0x4000, sb_next, CH_PRE_A4, CH_NODRAIN)                        \
            CH_W1N(3, ra0, 56, ra1, 0, 0x8000, 0x8000, sb_next, xh0, xh1, true, t0 + 3)                        \
            CH_W2N(4, ra1, 8, ra1, 16, 0xC000, 0xC000, sb_next, CH_PRE_A4, CH_NODRAIN)                         \
            CH_W1N(5, ra1, 24, ra1, 32, 0x10000, 0x10000, sb_next, xh1, xh0, true, t0 + 4)                     \
            CH_W2N(6, ra1, 40, ra1, 48, 0x14000, 0x14000, sb_next, CH_PRE_A4, CH_NODRAIN)                      \
            if constexpr (!(LAST)) {                                                                           \
                CH_W1N(7, ra1, 56, ra0, 0, 0x18000, 0x18000, sb_next, xh0, xh1, true, t0 + 5)                  \
            } else {                                                                                           \
                /* the last hidden tile has no W1 product behind it to hide in: activation + pack on its own, then its W2.   */ \
                /* (xh1 is re-defined HERE: without it the compiler is free to hoist this arithmetic above the W2 position  */ \
                /* before it, right behind the MFMAs that produce xh1 - whose wait states it cannot see)                      */ \
                asm volatile("" : "+v"(xh1));                                                                  \
                CH_SB0;                                                                                        \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) CH_ACT_PIECE(xh1, i)                             \
                CH_SB0;                                                                                        \
                CH_W2N(7, ra1, 56, ra0, 0, 0x18000, 0x18000, sb_next, CH_PRE_A, CH_DRAIN "s_waitcnt lgkmcnt(0)\n\t") \
            }                                                                                                  \
            ++g;                                                                                               \
            ++gpos;                                                                                            \
        }
        int g = 0;
        if (G_FFN == 1) {
            CH_FFN_GROUP(true, true)
        } else {
            CH_FFN_GROUP(true, false)
            while (g < G_FFN - 1) CH_FFN_GROUP(false, false)
            CH_FFN_GROUP(false, true)
        }
#undef CH_FFN_GROUP
#undef CH_W1N
#undef CH_W2N
#undef CH_ACT_GAP
#undef CH_ACT_PIECE
        // the last (unused) bias read and the last block's (unused) fragment reads
        CH_LAND_FRAGMENTS("");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]) :: "memory");
    }
    CH_STAMP(5)

    // ---- S4: the residual stream goes back to memory (row m: 16-byte pieces at channels 32 nt + 8 g + 4 half)
    if (p.store_x && (p.x_out_blk ? rb < nrb : live)) {  // (blocked: whole tiles, rows past M land in the buffer's padding)
        float* xp = p.x_out_blk ? p.x + (long long)rb * 8192 + 4 * lane : p.x + (long long)m * CH_D + 4 * half;
        const int step = p.x_out_blk ? 256 : 8;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[nt][4 * g + e];
                *reinterpret_cast<f32x4*>(xp + step * (4 * nt + g)) = o;
            }
    }
    CH_STAMP(6)
    if (p.has_next) {
        ch_layernorm_pack<false>(acc, tab_lane, CT_NLNA, CT_NLNB, 0, p.eps, bop);
        CH_STAMP(7)
        if (p.ln_out) {
            // y = LNn(x) itself, row-major bf16.  bop[2 nt + s][j] is channel 32 nt + 16 s + 8 (j >> 2) + 4 half + (j & 3)
            if (live) {
                bf16* op = p.ln_out + (long long)m * p.ld_ln + 4 * half;
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = bop[k][4 * q + e];
                        *reinterpret_cast<bf16x4*>(op + 16 * k + 8 * q) = o;
                    }
            }
        }
        // ---- S5: tail projection (acc-order k; B operands = LNn(x)), eight 32-column tiles per group.  Software-pipelined like
        // the feed-forward loop: tile tt accumulates in one of two accumulators (q0 / q1) while the previous tile's epilogue -
        // bf16 pack, the half-wave exchange, two 16-byte stores - is issued in the MFMA gaps of its first block (from the third
        // gap on: the MFMA -> VALU wait states of the previous tile's last product have passed by then), so no drain and no
        // stretch of VALU + store issue with the matrix pipe idle between tiles.
        if (G_TAIL > 0) {
            f32x16 q0, q1;
            f32x4 btv[4];
            bf16x4 o_[4];
            u32x4 w_[2];
            const int NTT = 8 * G_TAIL;
            auto tabt = [&](int tt) -> unsigned { return tab_lane + (unsigned)((CT_BT + 32 * (tt < NTT ? tt : NTT - 1)) * 4); };
            bf16* op16 = p.out + (long long)m * p.ldo + 8 * half;  // (row-major) tile tt: + 32 tt; after the half-wave exchange
            // blocked: tile (row block rb, column tile tt) is 2 KiB = [16-column half gp][lane][16 B]
            unsigned char* ob16 = reinterpret_cast<unsigned char*>(p.out) + (long long)rb * (p.ldo >> 5) * 2048 + lane * 16;
            // one store form for both layouts (no branches between the MFMAs): base + tile stride + half stride
            unsigned char* st_base = p.out_blk ? ob16 : reinterpret_cast<unsigned char*>(op16);
            const int st_tile = p.out_blk ? 2048 : 64, st_half = p.out_blk ? 1024 : 32;
            const bool st_on = (p.out_blk ? rb < nrb : live) && p.stamps != 2;  // (blocked: whole tiles, rows past M land in the padding)
            ch_tab4_nowait(tabt(0), btv[0], btv[1], btv[2], btv[3]);
            CH_READ8(Fa, ra0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(btv[0]), "+v"(btv[1]), "+v"(btv[2]), "+v"(btv[3]), "+v"(Fa[0]), "+v"(Fa[1]),
                         "+v"(Fa[2]), "+v"(Fa[3]), "+v"(Fa[4]), "+v"(Fa[5]), "+v"(Fa[6]), "+v"(Fa[7]) :: "memory");
// (plain stores: with the nt policy the tail ran 40 % longer - 38k against 27k cycles for 24 positions at 63 workgroups, 79k
// against 51k at 256, `tools/chain_stamps.py`; a tail without its stores takes 21k - and the whole benchmark 2.6 % longer)
#define CH_TAIL_STORE(v, ptr) *(ptr) = (v)
            // epilogue of tile TT held in QO, in four pieces
#define CH_EPI_CVT(QO, g0)                                                                                     \
            {                                                                                                  \
                _Pragma("unroll") for (int g_ = (g0); g_ < (g0) + 2; ++g_)                                     \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) o_[g_][e] = (bf16)QO[4 * g_ + e];             \
                asm volatile("" : "+v"(o_[g0]), "+v"(o_[(g0) + 1]));                                            \
            }
            // a lane holds channels 8 g + 4 half + (0..3) of its row: the two half-waves trade their odd / even groups
            // (v_permlane32_swap: upper half of the first operand <-> lower half of the second), after which a lane owns 8
            // consecutive channels of groups (half, half + 2): two 16-byte stores, not four 8-byte ones
#define CH_EPI_OUT(TT, gp)                                                                                     \
            {                                                                                                  \
                const uint2 lo_ = __builtin_bit_cast(uint2, o_[2 * (gp)]), hi_ = __builtin_bit_cast(uint2, o_[2 * (gp) + 1]); \
                const auto s0_ = __builtin_amdgcn_permlane32_swap(lo_.x, hi_.x, false, false);                 \
                const auto s1_ = __builtin_amdgcn_permlane32_swap(lo_.y, hi_.y, false, false);                 \
                w_[gp] = u32x4{s0_[0], s1_[0], s0_[1], s1_[1]};                                                \
                if (st_on) CH_TAIL_STORE(w_[gp], reinterpret_cast<u32x4*>(st_base + (long long)(TT) * st_tile + (gp) * st_half)); \
                else asm volatile("" :: "v"(w_[gp]));                                                          \
            }
#define CH_EPI_GAP(HAS_OLD, CODE)                                                                              \
            CH_SB0;                                                                                            \
            if constexpr (HAS_OLD) CODE                                                                        \
            CH_SB0;
            // position k of group g = tile tt = 8 g + k into QN; QO holds tile tt - 1
#define CH_S5N(k, RA_A, RK_A, RA_B, RK_B, M0OFF, SOFF, SB, QN, QO, HAS_OLD)                                    \
            QN = __builtin_shufflevector(__builtin_shufflevector(btv[0], btv[1], 0, 1, 2, 3, 4, 5, 6, 7),      \
                                         __builtin_shufflevector(btv[2], btv[3], 0, 1, 2, 3, 4, 5, 6, 7),      \
                                         0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);  /* bias = initial value */ \
            CH_M1R(CH_PRE_A, QN, Fa[0], bop[0], Fb[0], Fb[1], RA_A, (RK_A) + 0);                               \
            CH_SB0;                                                                                            \
            ch_tab4_nowait(tabt(8 * g + (k) + 1), btv[0], btv[1], btv[2], btv[3]);  /* the next tile's bias: landed by this position's second block */ \
            CH_SB0;                                                                                            \
            CH_M1R("", QN, Fa[1], bop[1], Fb[2], Fb[3], RA_A, (RK_A) + 2);                                     \
            CH_SB0;                                                                                            \
            CH_M1R("", QN, Fa[2], bop[2], Fb[4], Fb[5], RA_A, (RK_A) + 4);                                     \
            CH_EPI_GAP(HAS_OLD, CH_EPI_CVT(QO, 0))                                                             \
            CH_M1R("", QN, Fa[3], bop[3], Fb[6], Fb[7], RA_A, (RK_A) + 6);                                     \
            CH_EPI_GAP(HAS_OLD, CH_EPI_CVT(QO, 2))                                                             \
            CH_M1(QN, Fa[4], bop[4]);                                                                          \
            CH_EPI_GAP(HAS_OLD, CH_EPI_OUT(8 * g + (k) - 1, 0))                                                \
            CH_M1(QN, Fa[5], bop[5]);                                                                          \
            CH_EPI_GAP(HAS_OLD, CH_EPI_OUT(8 * g + (k) - 1, 1))                                                \
            CH_M1(QN, Fa[6], bop[6]);                                                                          \
            CH_SB0;                                                                                            \
            CH_M1(QN, Fa[7], bop[7]);                                                                          \
            CH_SB0;                                                                                            \
            CH_BLK1("+", "v", "%[c]", QN, Fb, bop, 8, Fa, RA_B, RK_B, CH_PRE_B_##k(M0OFF), CH_DMA0(SOFF), CH_DMA1, CH_DMA2, \
                    CH_DMA3, SB, CH_NODRAIN);
#define CH_TAIL_GROUP(FIRST, LAST)                                                                             \
            {                                                                                                  \
                const uint4* sb_cur = group_base(gpos);                                                        \
                const uint4* sb_next = group_base(gpos + 1);                                                   \
                CH_S5N(0, ra0, 8, ra0, 16, 0x1C000, 0x1C000, sb_cur, q0, q1, !(FIRST))                         \
                CH_S5N(1, ra0, 24, ra0, 32, 0x0, 0x0, sb_next, q1, q0, true)                                   \
                CH_S5N(2, ra0, 40, ra0, 48, 0x4000, 0x4000, sb_next, q0, q1, true)                             \
                CH_S5N(3, ra0, 56, ra1, 0, 0x8000, 0x8000, sb_next, q1, q0, true)                              \
                CH_S5N(4, ra1, 8, ra1, 16, 0xC000, 0xC000, sb_next, q0, q1, true)                              \
                CH_S5N(5, ra1, 24, ra1, 32, 0x10000, 0x10000, sb_next, q1, q0, true)                           \
                CH_S5N(6, ra1, 40, ra1, 48, 0x14000, 0x14000, sb_next, q0, q1, true)                           \
                CH_S5N(7, ra1, 56, ra0, 0, 0x18000, 0x18000, sb_next, q1, q0, true)                            \
                if constexpr (LAST) {                                                                          \
                    /* the last tile: drain (its products were just issued), then its epilogue on its own */   \
                    CH_LAND_FRAGMENTS("");                                                                     \
                    asm volatile(CH_DRAIN : "+v"(q1));                                                         \
                    CH_SB0;                                                                                    \
                    CH_EPI_CVT(q1, 0) CH_EPI_CVT(q1, 2) CH_EPI_OUT(8 * g + 7, 0) CH_EPI_OUT(8 * g + 7, 1)      \
                    CH_SB0;                                                                                    \
                }                                                                                              \
                ++g;                                                                                           \
                ++gpos;                                                                                        \
            }
            int g = 0;
            if (G_TAIL == 1) {
                CH_TAIL_GROUP(true, true)
            } else {
                CH_TAIL_GROUP(true, false)
                while (g < G_TAIL - 1) CH_TAIL_GROUP(false, false)
                CH_TAIL_GROUP(false, true)
            }
#undef CH_TAIL_GROUP
#undef CH_S5N
#undef CH_EPI_GAP
#undef CH_EPI_OUT
#undef CH_EPI_CVT
            CH_LAND_FRAGMENTS("");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(btv[0]), "+v"(btv[1]), "+v"(btv[2]), "+v"(btv[3]) :: "memory");
        }
    }
    CH_STAMP(8)
    // the dummy refills of the last units may still be writing this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    CH_STAMP(9)
    if (p.stamps && (int)blockIdx.x == p.stamp_block && tid == 0) ch_stamps[11] = (long long)__builtin_amdgcn_s_memrealtime();
}

int chain_print_stamps() {
    long long h[16];
    CN_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ch_stamps), sizeof(h)));
    static const char* names[] = {"prologue issue", "x/ctx landed + acc", "first unit visible", "S1 out-proj", "LN1 + b2",
                                  "S3 ffn loop", "S4 x store", "LNn", "S5 tail", "drain"};
    for (int i = 1; i < 10; ++i) fprintf(stderr, "[chain stamps] %-20s %8lld ticks\n", names[i], h[i] - h[i - 1]);
    fprintf(stderr, "[chain stamps] total %lld ticks (s_memtime)\n", h[9] - h[0]);
    // s_memrealtime runs at a constant 100 MHz: the ratio is the shader clock the workgroup saw
    fprintf(stderr, "[chain stamps] wall %.2f us (s_memrealtime) -> %.0f MHz\n", (h[11] - h[10]) / 100.0,
            (double)(h[9] - h[0]) / ((h[11] - h[10]) / 100.0));
    return 0;
}

int launch_chain(const ChainArgs& a, hipStream_t s) {
    if (a.f8 && (a.swish || a.dff <= 0 || a.dff % 256 != 0 || !a.f8_q)) {
        cn_set_error("chain: the e4m3 feed-forward form needs d_ff % 256 == 0 and the ReLU activation");
        return -1;
    }
    if (a.d != CH_D || a.dff < 0 || a.dff % 128 != 0 || a.dff / 32 > CH_MAX_FFN_TILES || a.tail_n < 0 || a.tail_n % 256 != 0 ||
        a.tail_n / 32 > CH_MAX_TAIL || (a.tail_n > 0 && (!a.has_next || !a.out)) || (a.has_next && !a.out && !a.ln_out)) {
        cn_set_error("chain: needs d_model == 256, d_ff % 128 == 0 <= 2048, tail width % 256 == 0 <= 1536");
        return -1;
    }
    if (a.M <= 0) return 0;
    ChainParams p;
    p.x = a.x;
    p.ctx = reinterpret_cast<const bf16*>(a.ctx);
    p.ldctx = a.ldctx;
    p.wstream = reinterpret_cast<const uint4*>(a.wstream);
    p.tab = a.tab;
    p.out = reinterpret_cast<bf16*>(a.out);
    p.ldo = a.ldo;
    // (no tail: `out` is where LNn(x) goes, as before)
    p.ln_out = reinterpret_cast<bf16*>(a.ln_out ? a.ln_out : (a.tail_n == 0 ? a.out : nullptr));
    p.ld_ln = a.ln_out ? a.ld_ln : a.ldo;
    p.M = a.M;
    p.ffn_tiles = a.dff / 32;
    p.tail_tiles = a.tail_n / 32;
    p.has_next = a.has_next;
    p.eps = a.eps;
    static const int stamps = cn_exp_env("CASSNAT_CHAIN_STAMPS") ? atoi(cn_exp_env("CASSNAT_CHAIN_STAMPS")) : 0;  // (thread-safe init)
    p.stamps = stamps;
    static const int stamp_block = cn_exp_env("CASSNAT_CHAIN_STAMP_BLOCK") ? atoi(cn_exp_env("CASSNAT_CHAIN_STAMP_BLOCK")) : 0;
    p.stamp_block = stamp_block;
    p.x_in_blk = a.x_in_blocked;
    p.x_out_blk = a.x_out_blocked;
    p.store_x = a.store_x && (a.ctx || a.dff);
    p.out_blk = a.out_blocked && a.tail_n > 0;
    p.ctx_blk = a.ctx && a.ctx_blocked;
    p.f8_q = a.f8_q;
    if (p.ctx_blk && a.ldctx != CH_D) {
        cn_set_error("chain: a blocked ctx has 256 columns");
        return -1;
    }
    if (p.out_blk && a.ldo % 32 != 0) {
        cn_set_error("chain: a blocked tail output needs ldo % 32 == 0");
        return -1;
    }
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)chain_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)chain_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)chain_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        attr_once.mark(attr_dev);
    }
    if (a.f8)
        hipLaunchKernelGGL((chain_kernel<false, true>), dim3(cn_ceil_div(p.M, 128)), dim3(256), CH_LDS, s, p);
    else if (a.swish)
        hipLaunchKernelGGL((chain_kernel<true, false>), dim3(cn_ceil_div(p.M, 128)), dim3(256), CH_LDS, s, p);
    else
        hipLaunchKernelGGL((chain_kernel<false, false>), dim3(cn_ceil_div(p.M, 128)), dim3(256), CH_LDS, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- host-side packing --------------------------------------------------------------------------------------
static inline uint16_t ch_bf16_bits(float f) { return cn_host_op16(f); }  // (the engine's 16-bit operand: common.h)

size_t chain_stream_units(int has_outproj, int dff, int tail_n, int f8) {
    return (has_outproj ? 8 : 0) + (f8 ? 1 : 2) * (dff / 32) + tail_n / 32;
}

// One unit = rows 32 rt .. +31 of a [N][ld] weight against 256 contraction indices starting at column k0:
//   natural order : frag ks, lane, j -> W[32 rt + (lane & 31)][k0 + 16 ks + 8 (lane >> 5) + j]
//   acc order     : frag 2 nt + s   -> W[32 rt + (lane & 31)][k0 + 32 nt + 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)]
//   (W2 units use acc order with k0 = 32 t and 32 contraction indices per (s) pair: frag 8 s + nt, rows 32 nt ..)
static void ch_pack_rows(const float* w, int ld, int rt, bool acc_order, uint16_t* out) {
    for (int f = 0; f < 16; ++f)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int k = acc_order ? 32 * (f >> 1) + 16 * (f & 1) + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)
                                        : 16 * f + 8 * (lane >> 5) + j;
                out[((size_t)f * 64 + lane) * 8 + j] = ch_bf16_bits(w[(size_t)(32 * rt + (lane & 31)) * ld + k]);
            }
}
static void ch_pack_w2(const float* w2, int dff, int t, uint16_t* out) {
    for (int s = 0; s < 2; ++s)
        for (int nt = 0; nt < 8; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    out[((size_t)(8 * s + nt) * 64 + lane) * 8 + j] = ch_bf16_bits(
                        w2[(size_t)(32 * nt + (lane & 31)) * dff + 32 * t + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)]);
}
// float -> OCP e4m3fn byte, round to nearest even, saturating at +-448 (what v_cvt_pk_fp8_f32 produces on gfx950)
unsigned char cn_f32_to_e4m3_host(float f) {
    const unsigned char sign = std::signbit(f) ? 0x80 : 0;
    float a = std::fabs(f);
    if (a != a) return 0x7f;
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0.0009765625f) return sign;  // below half of the smallest subnormal (2^-9): zero (the tie goes to even = 0)
    int e;
    (void)std::frexp(a, &e);
    int E = e - 1;  // a = 1.xxx * 2^E
    if (E < -6) {   // subnormal: multiples of 2^-9
        const int r = (int)std::nearbyint(a * 512.f);
        return sign | (unsigned char)(r >= 8 ? 0x08 : r);
    }
    int r = (int)std::nearbyint((a / std::ldexp(1.f, E) - 1.f) * 8.f);
    if (r == 8) {
        r = 0;
        ++E;
    }
    const int bits = ((E + 7) << 3) | r;
    return sign | (unsigned char)(bits > 0x7e ? 0x7e : bits);
}

// F8 units (e4m3 bytes at `scale`).  An A operand of the K = 64 MFMA is 32 bytes per lane = pieces 2 t, 2 t + 1 of the unit
// (piece = [lane][16 B]); byte e = 16 (piece & 1) + j.  Row of the accumulator layout behind byte position (half, r = e & 15):
//   acc_k(r, half) = 8 (r >> 2) + 4 half + (r & 3)
//   W1 unit of pair p : operand t = 4 s + kk -> W1[64 p + 32 s + (lane & 31)][64 kk + 32 (e >> 4) + acc_k(e & 15, lane >> 5)]
//   W2 unit of pair p : operand t = nt       -> W2[32 nt + (lane & 31)][64 p + 32 (e >> 4) + acc_k(e & 15, lane >> 5)]
static void ch_pack_w1_f8(const float* w1, int p, float scale, unsigned char* out) {
    for (int t = 0; t < 8; ++t)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 32; ++e) {
                const int s = t >> 2, kk = t & 3, r = e & 15;
                const int k = 64 * kk + 32 * (e >> 4) + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                out[((size_t)(2 * t + (e >> 4)) * 64 + lane) * 16 + r] =
                    cn_f32_to_e4m3_host(w1[(size_t)(64 * p + 32 * s + (lane & 31)) * CH_D + k] * scale);
            }
}
static void ch_pack_w2_f8(const float* w2, int dff, int p, float scale, unsigned char* out) {
    for (int nt = 0; nt < 8; ++nt)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 32; ++e) {
                const int r = e & 15;
                const int k = 64 * p + 32 * (e >> 4) + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                out[((size_t)(2 * nt + (e >> 4)) * 64 + lane) * 16 + r] =
                    cn_f32_to_e4m3_host(w2[(size_t)(32 * nt + (lane & 31)) * dff + k] * scale);
            }
}
// largest power of two that keeps max |w| * scale <= 448 (as the unfused fp8 path chooses it: model.hip, quant8)
static float ch_f8_scale(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    return mx > 0.f ? std::ldexp(1.f, (int)std::floor(std::log2(448.f / mx))) : 1.f;
}

void pack_chain(const ChainWeights& w, uint16_t* stream, float* tab, int* f8_q) {
    const size_t unit = CH_UNIT_BYTES / 2;
    size_t u = 0;
    const bool f8 = f8_q != nullptr;
    if (w.wo)
        for (int rt = 0; rt < 8; ++rt) ch_pack_rows(w.wo, CH_D, rt, false, stream + unit * u++);
    // software-pipelined by one hidden tile (chain_kernel, S3): W1(0), then W1(t + 1), W2(t) for t = 0 .. n - 2, then W2(n - 1)
    // (F8: the same with pairs of tiles, one unit each)
    const int nt_ffn = f8 ? w.dff / 64 : w.dff / 32;
    float ln_mul = 1.f, hid_mul = 1.f;
    if (f8 && nt_ffn > 0) {
        const float s1 = ch_f8_scale(w.w1, (size_t)w.dff * CH_D), s2 = ch_f8_scale(w.w2, (size_t)w.dff * CH_D);
        ln_mul = CHAIN_F8_S_LN;
        hid_mul = CHAIN_F8_S_HID;
        // W1 product: e4m3(W1 s1) . e4m3(LN1(x) S_LN), wanted at the hidden scale: x S_HID / (s1 S_LN); W2: x 1 / (s2 S_HID)
        f8_q[0] = 127 - (int)std::lround(std::log2(s1));
        f8_q[1] = 127 + (int)std::lround(std::log2(hid_mul)) - (int)std::lround(std::log2(ln_mul));
        f8_q[2] = 127 - (int)std::lround(std::log2(s2));
        f8_q[3] = 127 - (int)std::lround(std::log2(hid_mul));
        unsigned char* bytes = reinterpret_cast<unsigned char*>(stream);
        ch_pack_w1_f8(w.w1, 0, s1, bytes + CH_UNIT_BYTES * u++);
        for (int p = 0; p + 1 < nt_ffn; ++p) {
            ch_pack_w1_f8(w.w1, p + 1, s1, bytes + CH_UNIT_BYTES * u++);
            ch_pack_w2_f8(w.w2, w.dff, p, s2, bytes + CH_UNIT_BYTES * u++);
        }
        ch_pack_w2_f8(w.w2, w.dff, nt_ffn - 1, s2, bytes + CH_UNIT_BYTES * u++);
    } else {
        if (nt_ffn > 0) ch_pack_rows(w.w1, CH_D, 0, true, stream + unit * u++);
        for (int t = 0; t + 1 < nt_ffn; ++t) {
            ch_pack_rows(w.w1, CH_D, t + 1, true, stream + unit * u++);
            ch_pack_w2(w.w2, w.dff, t, stream + unit * u++);
        }
        if (nt_ffn > 0) ch_pack_w2(w.w2, w.dff, nt_ffn - 1, stream + unit * u++);
    }
    for (int jt = 0; jt < w.tail_n / 32; ++jt) ch_pack_rows(w.wt, CH_D, jt, true, stream + unit * u++);
    memset(tab, 0, sizeof(float) * CH_TAB_FLOATS);
    auto put = [&](int off, const float* src, int n, float mul) {
        if (src)
            for (int i = 0; i < n; ++i) tab[off + i] = src[i] * mul;  // (mul is 1 or a power of two: exact)
    };
    put(CT_BO, w.bo, CH_D, 1.f);
    put(CT_LN1A, w.ln1_a, CH_D, ln_mul);
    put(CT_LN1B, w.ln1_b, CH_D, ln_mul);
    put(CT_B2, w.b2, CH_D, 1.f);
    put(CT_NLNA, w.nln_a, CH_D, 1.f);
    put(CT_NLNB, w.nln_b, CH_D, 1.f);
    put(CT_BT, w.bt, w.tail_n, 1.f);
    put(CT_B1, w.b1, w.dff, hid_mul);
}
