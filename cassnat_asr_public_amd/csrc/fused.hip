// Fused position-wise feed-forward sublayer for gfx950 (bf16 MFMA, d_model = 256):
//     x <- x + W2 . relu(W1 . LN(x) + b1) + b2          [and optionally  xn_next <- LN_next(x)]
// Replaces SublayerConnection(LayerNorm -> PositionwiseFeedForward) of the reference
// (src/models/modules/utils.py:23-32, positionff.py:15-16, norm.py:15-18) - four launches (LN, w_1+ReLU,
// w_2+residual, next LN) and a 2 x M x d_ff round trip of the hidden activations through HBM.
//
// Shape of the problem at B=32: M = 8000 rows, d = 256, d_ff = 2048.  With only 32 rows per workgroup (250
// workgroups for 256 CUs) a weight fragment is used by exactly one MFMA per workgroup, so staging weights in
// LDS buys no reuse.  Instead:
//   * weights are pre-tiled at pack time into MFMA fragment order; every wave streams ITS slice of d_ff as
//     fully coalesced 1-KiB pieces (64 lanes x 16 B) by LDS-DMA (global_load_lds_dwordx4) into a private
//     32-slot ring - one slot per fragment of a hidden tile, refilled for the next tile right after its
//     MFMA - so ~31 KiB per wave stay in flight at no VGPR cost, every wait is a counted vmcnt(28) on the
//     wave's own queue, and there is no barrier in the main loop;
//   * the first product is computed swapped, X[f][m] = sum_k W1[f][k] xn[m][k], so its 32x32 accumulator
//     (f on rows, m on lanes) is, after bias+ReLU and a bf16 pack, directly the B operand of the second
//     product out^T[n][m] += sum_f W2[n][f] X[f][m]; W2 is packed in the k-order that operand implies.
//     The hidden activations never leave registers.
//   * each wave owns d_ff/4 of the hidden units and a full 256x32 fp32 partial of out^T; the four partials
//     are summed through LDS once per workgroup, fused with bias, residual add and the next LayerNorm.
// The kernel is bound by the per-CU L2 bandwidth of the weight stream (2 MB per workgroup), not by MFMA.
#include <cstdlib>
#include <cstring>

#include "kernels.h"

struct FfnParams {
    float* x;  // [M][256] residual stream, updated in place
    const float* ln_a;
    const float* ln_b;
    const bf16x8* w1p;  // [dff/32][16][64]   A fragments of W1 (rows f, natural k order)
    const float* b1;    // [dff]
    const bf16x8* w2p;  // [dff/32][2][8][64] A fragments of W2 (rows n, k = f in accumulator-operand order)
    const float* b2;    // [256]
    const float* nln_a;  // next LayerNorm (may be null)
    const float* nln_b;
    bf16* xn_out;  // [M][256] bf16, written when nln_a != null
    int M, dff;
    float eps;
    // d_ff split (few rows, e.g. an autoregressive decode step): gridDim.y = nslice workgroups share a row tile, each takes
    // dff / nslice hidden units and writes its W2 partial products to partial[slice][M][256] (no bias, no residual, x is
    // not touched); ffn_reduce_kernel adds them up in slice order.  nslice == 1: the whole sublayer in place.
    int nslice;
    float* partial;
};

constexpr int FF_D = 256;
constexpr int FF_P_STRIDE = FF_D + 4;  // floats per partial row in LDS
constexpr int FF_MAX_DFF = 2048;
constexpr int FF_RING_BYTES = 32 * 1024;  // per wave: 32 fragments of 1 KiB
// LDS: [MT x 16 KiB xn fragments][4 x 32 KiB rings]; the 4 x 32 x 260 fp32 partials of the epilogue alias all of it
template <int MT> constexpr int ff_lds_bytes() { return MT * 16384 + 4 * FF_RING_BYTES; }
static_assert(4 * 32 * FF_P_STRIDE * 4 <= ff_lds_bytes<1>(), "partials must fit");
static_assert(ff_lds_bytes<2>() <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void ln_row_to(const f32x4 v, float mean, float denom, const f32x4 g, const f32x4 bb,
                                          float* o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] * (v[j] - mean) / denom + bb[j];
}

// MT = 32-row M-tiles per workgroup (1 or 2).  With MT = 2 every streamed weight fragment feeds two MFMAs, which halves
// the L2->LDS bytes per row - the resource this kernel is bound by.
// DBG: timing experiments only (1 = no DMA refills, 2 = no MFMAs, 3 = DMA stream only; results are wrong); 0 in production
template <int MT, int DBG>
__global__ __launch_bounds__(256) void ffn_fused_kernel(FfnParams p) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xn_s = smem;                     // [MT][16 k-steps][64 lanes][16 B]: B fragments of xn^T, ready to read
    float* part = reinterpret_cast<float*>(smem);   // [4][32][260] fp32, epilogue only (aliases everything)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned char* ring = smem + MT * 16384 + wave_u * FF_RING_BYTES;

    const int tiles_per_wave = p.dff / 32 / 4 / p.nslice;
    const int ft0 = ((int)blockIdx.y * 4 + wave_u) * tiles_per_wave;
    // Every workgroup walks its hidden tiles in a different rotation: the sum over tiles is order-free, and the
    // workgroups no longer pull the same L2 lines at the same moment (L2 channel hot-spotting).
    const int rot = (blockIdx.x * 7 + wave_u * 3) % tiles_per_wave;
#define FF_TT(t) (((t) + rot) % tiles_per_wave)
    // fragment stream of this wave: tile t, fragment i (0-15: W1 k-steps, 16-31: W2 (s, nt)) -> 1 KiB piece
    const uint4* w1 = reinterpret_cast<const uint4*>(p.w1p) + (long long)ft0 * 16 * 64 + lane;
    const uint4* w2 = reinterpret_cast<const uint4*>(p.w2p) + (long long)ft0 * 16 * 64 + lane;
#define FF_DMA(src, slot)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(ring + (slot) * 1024), 16, 0, 0)
    // The wave's biases (32 per hidden tile, <= 512) live in 8 VGPRs per lane, in PROCESSING order: register j, lane
    // 32*p + i holds b1 of hidden unit i of the tile processed at position 2j+p.  A lane fetches the 16 it needs
    // per tile with ds_bpermute (no LDS memory, no global load inside the DMA-pipelined loop), and the queue is
    // shifted down every second tile so the register index stays static.
    float bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pos = 2 * j + half;
        bq[j] = pos < tiles_per_wave ? p.b1[32 * (ft0 + FF_TT(pos)) + l31] : 0.f;
    }
    // prologue: the whole first tile goes in flight before the LayerNorm below
#pragma unroll
    for (int i = 0; i < 16; ++i) FF_DMA(w1 + ((long long)FF_TT(0) * 16 + i) * 64, i);
#pragma unroll
    for (int i = 0; i < 16; ++i) FF_DMA(w2 + ((long long)FF_TT(0) * 16 + i) * 64, 16 + i);

    // ---- LayerNorm of the workgroup's rows -> bf16 fragments in LDS (wave w: rows w, w+4, ...)
    {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_a + 4 * lane);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * lane);
        // element k = 4*lane + j of row r lands in fragment (mt = r>>5, ks = k>>4), lane slot 32*((k>>3)&1) + (r&31), byte 2*(k&7)
        const int ks = lane >> 2, kh = (lane >> 1) & 1, kb = (lane & 1) * 8;
#pragma unroll
        for (int i = 0; i < BM / 4; ++i) {
            const int r = wave + 4 * i;
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (long long)m * FF_D + 4 * lane);
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)FF_D;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) ss = fmaf(v[j] - mean, v[j] - mean, ss);
            const float denom = sqrtf(wave_sum(ss) / (float)(FF_D - 1)) + p.eps;
            float o[4];
            ln_row_to(v, mean, denom, g, bb, o);
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) ob[j] = (bf16)o[j];
            *reinterpret_cast<bf16x4*>(xn_s + (((r >> 5) * 16 + ks) * 64 + kh * 32 + (r & 31)) * 16 + kb) = ob;
        }
    }
    __syncthreads();

    // LDS byte addresses for the inline-asm reads below.  hipcc inserts a full vmcnt(0) in front of every ordinary
    // LDS access while an LDS-DMA is outstanding (it cannot tell the slots apart), so all main-loop LDS reads are
    // issued from asm statements that carry their own counted vmcnt / lgkmcnt(0) (cdna_hip_programming.md 5.7, form i).
    const unsigned xfrag_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(xn_s + lane * 16);
    const unsigned slot_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(ring + lane * 16);

    f32x16 acc[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // One hidden tile = 8 groups of 4 fragments (0-3: W1, 4-7: W2).  Each group: [counted vmcnt] -> issue its LDS reads ->
    // issue the DMA refills of the PREVIOUS group's slots (their data is already in registers; the refills overlap the
    // LDS-read latency and keep the DMA queue full) -> lgkmcnt(0) -> MFMAs.  With refills trailing by one group, the DMAs
    // younger than group g's are always 6 groups = 24 in steady state; the last tile issues none, so its counts fall
    // 24, 24, 20, 16, 12, 8, 4, 0 (the first wait still sees the refill of the previous tile's group 7... which the last
    // tile's group 0 block does issue).  Every wait is a literal vmcnt on the wave's own queue.
#define FF_STR2(x) #x
#define FF_STR(x) FF_STR2(x)
#define FF_MFMA(a, b, c) c = CN_MFMA16(a, b, c, 0, 0, 0)
#define FF_REFILL_W1(g) { _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
        FF_DMA(w1 + ((long long)tnext * 16 + 4 * (g) + j) * 64, 4 * (g) + j); }
#define FF_REFILL_W2(g) { _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
        FF_DMA(w2 + ((long long)tnext * 16 + 4 * (g) + j) * 64, 16 + 4 * (g) + j); }
    // REFILL: statement(s) issued between the read issue and the wait (the previous group's refill, or nothing)
#define FF_PHASE_A(g, WAITN, REFILL)                                                                          \
    {                                                                                                         \
        bf16x8 wf0, wf1, wf2, wf3, xq0, xq1, xq2, xq3, yq0, yq1, yq2, yq3;                                    \
        if constexpr (DBG == 3) { asm volatile("s_waitcnt vmcnt(" FF_STR(WAITN) ")" ::: "memory"); }          \
        else if constexpr (MT == 1)                                                                           \
        asm volatile("s_waitcnt vmcnt(" FF_STR(WAITN) ")\n\t"                                                 \
                     "ds_read_b128 %0, %8 offset:" FF_STR((4 * (g) + 0) * 1024) "\n\t"                        \
                     "ds_read_b128 %1, %8 offset:" FF_STR((4 * (g) + 1) * 1024) "\n\t"                        \
                     "ds_read_b128 %2, %8 offset:" FF_STR((4 * (g) + 2) * 1024) "\n\t"                        \
                     "ds_read_b128 %3, %8 offset:" FF_STR((4 * (g) + 3) * 1024) "\n\t"                        \
                     "ds_read_b128 %4, %9 offset:" FF_STR((4 * (g) + 0) * 1024) "\n\t"                        \
                     "ds_read_b128 %5, %9 offset:" FF_STR((4 * (g) + 1) * 1024) "\n\t"                        \
                     "ds_read_b128 %6, %9 offset:" FF_STR((4 * (g) + 2) * 1024) "\n\t"                        \
                     "ds_read_b128 %7, %9 offset:" FF_STR((4 * (g) + 3) * 1024)                                \
                     : "=&v"(wf0), "=&v"(wf1), "=&v"(wf2), "=&v"(wf3), "=&v"(xq0), "=&v"(xq1), "=&v"(xq2),    \
                       "=&v"(xq3)                                                                             \
                     : "v"(slot_a), "v"(xfrag_a)                                                              \
                     : "memory");                                                                             \
        else                                                                                                  \
        asm volatile("s_waitcnt vmcnt(" FF_STR(WAITN) ")\n\t"                                                 \
                     "ds_read_b128 %0, %12 offset:" FF_STR((4 * (g) + 0) * 1024) "\n\t"                       \
                     "ds_read_b128 %1, %12 offset:" FF_STR((4 * (g) + 1) * 1024) "\n\t"                       \
                     "ds_read_b128 %2, %12 offset:" FF_STR((4 * (g) + 2) * 1024) "\n\t"                       \
                     "ds_read_b128 %3, %12 offset:" FF_STR((4 * (g) + 3) * 1024) "\n\t"                       \
                     "ds_read_b128 %4, %13 offset:" FF_STR((4 * (g) + 0) * 1024) "\n\t"                       \
                     "ds_read_b128 %5, %13 offset:" FF_STR((4 * (g) + 1) * 1024) "\n\t"                       \
                     "ds_read_b128 %6, %13 offset:" FF_STR((4 * (g) + 2) * 1024) "\n\t"                       \
                     "ds_read_b128 %7, %13 offset:" FF_STR((4 * (g) + 3) * 1024) "\n\t"                       \
                     "ds_read_b128 %8, %13 offset:" FF_STR((16 + 4 * (g) + 0) * 1024) "\n\t"                  \
                     "ds_read_b128 %9, %13 offset:" FF_STR((16 + 4 * (g) + 1) * 1024) "\n\t"                  \
                     "ds_read_b128 %10, %13 offset:" FF_STR((16 + 4 * (g) + 2) * 1024) "\n\t"                 \
                     "ds_read_b128 %11, %13 offset:" FF_STR((16 + 4 * (g) + 3) * 1024)                         \
                     : "=&v"(wf0), "=&v"(wf1), "=&v"(wf2), "=&v"(wf3), "=&v"(xq0), "=&v"(xq1), "=&v"(xq2),    \
                       "=&v"(xq3), "=&v"(yq0), "=&v"(yq1), "=&v"(yq2), "=&v"(yq3)                             \
                     : "v"(slot_a), "v"(xfrag_a)                                                              \
                     : "memory");                                                                             \
        if constexpr (DBG != 1) { REFILL }                                                                    \
        if constexpr (DBG != 3) {                                                                             \
            if constexpr (MT == 1)                                                                            \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf0), "+v"(wf1), "+v"(wf2), "+v"(wf3), "+v"(xq0),  \
                             "+v"(xq1), "+v"(xq2), "+v"(xq3) :: "memory");                                    \
            else                                                                                              \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf0), "+v"(wf1), "+v"(wf2), "+v"(wf3), "+v"(xq0),  \
                             "+v"(xq1), "+v"(xq2), "+v"(xq3), "+v"(yq0), "+v"(yq1), "+v"(yq2), "+v"(yq3)      \
                             :: "memory");                                                                    \
        }                                                                                                     \
        if constexpr (DBG == 0 || DBG == 1) {                                                                 \
            FF_MFMA(wf0, xq0, xh[0]); FF_MFMA(wf1, xq1, xh[0]); FF_MFMA(wf2, xq2, xh[0]); FF_MFMA(wf3, xq3, xh[0]); \
            if constexpr (MT == 2) {                                                                          \
                FF_MFMA(wf0, yq0, xh[MT - 1]); FF_MFMA(wf1, yq1, xh[MT - 1]);                                 \
                FF_MFMA(wf2, yq2, xh[MT - 1]); FF_MFMA(wf3, yq3, xh[MT - 1]);                                 \
            }                                                                                                 \
        } else if constexpr (DBG == 2) {                                                                      \
            xh[0][0] += (float)wf0[0] + (float)wf1[0] + (float)wf2[0] + (float)wf3[0] + (float)xq0[0];        \
        }                                                                                                     \
    }
    // phase B group g: W2 fragments (s = g>>1, nt = 4(g&1)..+3) in slots 16+4g..
#define FF_PHASE_B(g, WAITN, REFILL)                                                                          \
    {                                                                                                         \
        bf16x8 wf0, wf1, wf2, wf3;                                                                            \
        if constexpr (DBG == 3) { asm volatile("s_waitcnt vmcnt(" FF_STR(WAITN) ")" ::: "memory"); } else     \
        asm volatile("s_waitcnt vmcnt(" FF_STR(WAITN) ")\n\t"                                                 \
                     "ds_read_b128 %0, %4 offset:" FF_STR((16 + 4 * (g) + 0) * 1024) "\n\t"                   \
                     "ds_read_b128 %1, %4 offset:" FF_STR((16 + 4 * (g) + 1) * 1024) "\n\t"                   \
                     "ds_read_b128 %2, %4 offset:" FF_STR((16 + 4 * (g) + 2) * 1024) "\n\t"                   \
                     "ds_read_b128 %3, %4 offset:" FF_STR((16 + 4 * (g) + 3) * 1024)                           \
                     : "=&v"(wf0), "=&v"(wf1), "=&v"(wf2), "=&v"(wf3)                                         \
                     : "v"(slot_a)                                                                            \
                     : "memory");                                                                             \
        if constexpr (DBG != 1) { REFILL }                                                                    \
        if constexpr (DBG != 3)                                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf0), "+v"(wf1), "+v"(wf2), "+v"(wf3) :: "memory");    \
        if constexpr (DBG == 0 || DBG == 1) {                                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                               \
                FF_MFMA(wf0, pb[mt][(g) >> 1], acc[mt][4 * ((g) & 1) + 0]);                                   \
                FF_MFMA(wf1, pb[mt][(g) >> 1], acc[mt][4 * ((g) & 1) + 1]);                                   \
                FF_MFMA(wf2, pb[mt][(g) >> 1], acc[mt][4 * ((g) & 1) + 2]);                                   \
                FF_MFMA(wf3, pb[mt][(g) >> 1], acc[mt][4 * ((g) & 1) + 3]);                                   \
            }                                                                                                 \
        } else if constexpr (DBG == 2) {                                                                      \
            acc[0][0][0] += (float)wf0[0] + (float)wf1[0] + (float)wf2[0] + (float)wf3[0];                    \
        }                                                                                                     \
    }
    // bias + ReLU on hidden unit f = 32*tile + acc_row(r, lane), then pack as the B operand of phase B
#define FF_RELU_PACK(pos)                                                                                     \
    bf16x8 pb[MT][2];                                                                                         \
    {                                                                                                         \
        const int src0 = 32 * ((pos) & 1) + 4 * half;                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int e = 0; e < 4; ++e) {         \
            const float bv = __shfl(bq[0], src0 + 8 * g + e);                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                 \
                pb[mt][g >> 1][4 * (g & 1) + e] = (bf16)fmaxf(xh[mt][4 * g + e] + bv, 0.f);                   \
        }                                                                                                     \
        if ((pos) & 1) { _Pragma("unroll") for (int j = 0; j < 7; ++j) bq[j] = bq[j + 1]; }                   \
    }
#define FF_ZERO_XH()                                                                                          \
    f32x16 xh[MT];                                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int r = 0; r < 16; ++r) xh[mt][r] = 0.f;

    // Refill schedule: group h of the NEXT tile is re-issued in the block of group h+1 of this tile; the refill of this
    // tile's last W2 group (slot group 7) is issued in group 0's block of the next tile (tnext there = that tile).
    int t = 0;
    int tnext = FF_TT(0);  // tile whose slot group 7 is still to be refilled at the start of the loop: none for t == 0
    for (; t + 1 < tiles_per_wave; ++t) {
        FF_ZERO_XH()
        if (t == 0) {
            FF_PHASE_A(0, 28, )
        } else {
            FF_PHASE_A(0, 24, FF_REFILL_W2(3))
        }
        tnext = FF_TT(t + 1);
        FF_PHASE_A(1, 24, FF_REFILL_W1(0)) FF_PHASE_A(2, 24, FF_REFILL_W1(1)) FF_PHASE_A(3, 24, FF_REFILL_W1(2))
        FF_RELU_PACK(t)
        FF_PHASE_B(0, 24, FF_REFILL_W1(3)) FF_PHASE_B(1, 24, FF_REFILL_W2(0)) FF_PHASE_B(2, 24, FF_REFILL_W2(1))
        FF_PHASE_B(3, 24, FF_REFILL_W2(2))
    }
    {
        FF_ZERO_XH()
        if (t == 0) {
            FF_PHASE_A(0, 28, )
        } else {
            FF_PHASE_A(0, 24, FF_REFILL_W2(3))
        }
        FF_PHASE_A(1, 24, ) FF_PHASE_A(2, 20, ) FF_PHASE_A(3, 16, )
        FF_RELU_PACK(t)
        FF_PHASE_B(0, 12, ) FF_PHASE_B(1, 8, ) FF_PHASE_B(2, 4, ) FF_PHASE_B(3, 0, )
    }
#undef FF_PHASE_A
#undef FF_REFILL_W1
#undef FF_REFILL_W2
#undef FF_PHASE_B
#undef FF_RELU_PACK
#undef FF_ZERO_XH
#undef FF_MFMA
#undef FF_TT
#undef FF_DMA

    // ---- cross-wave reduction of out^T partials (one 32-row M-tile at a time), + b2 + residual, next LayerNorm
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(p.b2 + 4 * lane);
    f32x4 ng, nb;
    if (p.nln_a) {
        ng = *reinterpret_cast<const f32x4*>(p.nln_a + 4 * lane);
        nb = *reinterpret_cast<const f32x4*>(p.nln_b + 4 * lane);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __syncthreads();  // rings / xn fragments (mt == 0) or the previous round's partials are no longer read
        float* mine = part + (wave * 32 + l31) * FF_P_STRIDE;
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
            if (p.nslice > 1) {  // this slice's share of W2 h only, waves added in a fixed order
                f32x4 v = *reinterpret_cast<const f32x4*>(part + r * FF_P_STRIDE + 4 * lane);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(part + (w * 32 + r) * FF_P_STRIDE + 4 * lane);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += q[j];
                }
                *reinterpret_cast<f32x4*>(p.partial + ((long long)blockIdx.y * p.M + m) * FF_D + 4 * lane) = v;
                continue;
            }
            float* xr = p.x + (long long)m * FF_D + 4 * lane;
            f32x4 v = *reinterpret_cast<const f32x4*>(xr);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(part + (w * 32 + r) * FF_P_STRIDE + 4 * lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += q[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += b2v[j];
            *reinterpret_cast<f32x4*>(xr) = v;
            if (p.nln_a) {
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)FF_D;
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) ss = fmaf(v[j] - mean, v[j] - mean, ss);
                const float denom = sqrtf(wave_sum(ss) / (float)(FF_D - 1)) + p.eps;
                float o[4];
                ln_row_to(v, mean, denom, ng, nb, o);
                bf16x4 ob;
#pragma unroll
                for (int j = 0; j < 4; ++j) ob[j] = (bf16)o[j];
                *reinterpret_cast<bf16x4*>(p.xn_out + (long long)m * FF_D + 4 * lane) = ob;
            }
        }
    }
}

template <int MT, int DBG> static int launch_ffn_variant(const FfnParams& p, hipStream_t s) {
    constexpr int lds = ff_lds_bytes<MT>();
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_fused_kernel<MT, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL((ffn_fused_kernel<MT, DBG>), dim3(cn_ceil_div(p.M, 32 * MT), p.nslice), dim3(256), lds, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_ffn_fused(const FfnFusedArgs& a, hipStream_t s) {
    if (a.d != FF_D || a.dff <= 0 || a.dff % 128 != 0 || a.dff > FF_MAX_DFF) {
        cn_set_error("ffn_fused: needs d_model == 256 and d_ff % 128 == 0, d_ff <= 2048");
        return -1;
    }
    if (a.M <= 0) return 0;
    FfnParams p;
    p.x = a.x;
    p.ln_a = a.ln_a;
    p.ln_b = a.ln_b;
    p.w1p = reinterpret_cast<const bf16x8*>(a.w1p);
    p.b1 = a.b1;
    p.w2p = reinterpret_cast<const bf16x8*>(a.w2p);
    p.b2 = a.b2;
    p.nln_a = a.nln_a;
    p.nln_b = a.nln_b;
    p.xn_out = reinterpret_cast<bf16*>(a.xn_out);
    p.M = a.M;
    p.dff = a.dff;
    p.eps = a.eps;
    p.nslice = a.nslice > 1 ? a.nslice : 1;
    p.partial = a.partial;
    if (p.nslice > 1 && (a.dff % (128 * p.nslice) != 0 || !a.partial)) {
        cn_set_error("ffn_fused: a d_ff split needs d_ff % (128 * slices) == 0 and a partial-sum buffer");
        return -1;
    }
    static const int dbg = cn_exp_env("CASSNAT_FFN_DEBUG") ? atoi(cn_exp_env("CASSNAT_FFN_DEBUG")) : 0;  // (magic statics: thread-safe)
    static const int force_mt = cn_exp_env("CASSNAT_FFN_MT") ? atoi(cn_exp_env("CASSNAT_FFN_MT")) : 0;
    const int mt = force_mt ? force_mt : (a.M > 32 ? 2 : 1);
    if (mt == 2) {
        if (dbg == 3) return launch_ffn_variant<2, 3>(p, s);
        return launch_ffn_variant<2, 0>(p, s);
    }
    if (dbg == 1) return launch_ffn_variant<1, 1>(p, s);
    if (dbg == 2) return launch_ffn_variant<1, 2>(p, s);
    if (dbg == 3) return launch_ffn_variant<1, 3>(p, s);
    return launch_ffn_variant<1, 0>(p, s);
}

// ---- second half of the d_ff split: x[m] += b2 + sum over slices of partial[slice][m] (slice order: deterministic), then
// optionally the next LayerNorm of the row in bf16 (the norm the following sublayer would otherwise launch for).
__global__ __launch_bounds__(256) void ffn_reduce_kernel(float* __restrict__ x, const float* __restrict__ partial, int nslice,
                                                         const float* __restrict__ b2, const float* __restrict__ nln_a,
                                                         const float* __restrict__ nln_b, bf16* __restrict__ xn_out, int M,
                                                         float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float* xr = x + (long long)m * FF_D + 4 * lane;
    f32x4 v = *reinterpret_cast<const f32x4*>(xr);
    f32x4 acc = *reinterpret_cast<const f32x4*>(partial + (long long)m * FF_D + 4 * lane);
    for (int sl = 1; sl < nslice; ++sl) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(partial + ((long long)sl * M + m) * FF_D + 4 * lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += q[j];
    }
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2 + 4 * lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (v[j] + acc[j]) + b2v[j];
    *reinterpret_cast<f32x4*>(xr) = v;
    if (nln_a) {
        const f32x4 ng = *reinterpret_cast<const f32x4*>(nln_a + 4 * lane);
        const f32x4 nb = *reinterpret_cast<const f32x4*>(nln_b + 4 * lane);
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)FF_D;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ss = fmaf(v[j] - mean, v[j] - mean, ss);
        const float denom = sqrtf(wave_sum(ss) / (float)(FF_D - 1)) + eps;
        float o[4];
        ln_row_to(v, mean, denom, ng, nb, o);
        bf16x4 ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[j] = (bf16)o[j];
        *reinterpret_cast<bf16x4*>(xn_out + (long long)m * FF_D + 4 * lane) = ob;
    }
}

int launch_ffn_reduce(float* x, const float* partial, int nslice, const float* b2, const float* nln_a, const float* nln_b,
                      void* xn_out, int M, float eps, hipStream_t s) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(ffn_reduce_kernel, dim3(cn_ceil_div(M, 4)), dim3(256), 0, s, x, partial, nslice, b2, nln_a, nln_b,
                       reinterpret_cast<bf16*>(xn_out), M, eps);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- host-side packing of nn.Linear weights into the fragment streams above -----------------------------
static inline uint16_t bf16_bits(float f) { return cn_host_op16(f); }  // (the engine's 16-bit operand: common.h)

// W1 [dff][256] fp32 -> [dff/32][16][64][8] bf16 :  frag(ft, ks, lane)[j] = W1[32ft + (lane&31)][16ks + 8(lane>>5) + j]
void pack_ffn_w1(const float* w1, int dff, uint16_t* out) {
    for (int ft = 0; ft < dff / 32; ++ft)
        for (int ks = 0; ks < 16; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    out[(((size_t)ft * 16 + ks) * 64 + lane) * 8 + j] =
                        bf16_bits(w1[(size_t)(32 * ft + (lane & 31)) * FF_D + 16 * ks + 8 * (lane >> 5) + j]);
}

// W2 [256][dff] fp32 -> [dff/32][2][8][64][8] bf16 :
//   frag(ft, s, nt, lane)[j] = W2[32nt + (lane&31)][32ft + 16s + 8(j>>2) + 4(lane>>5) + (j&3)]
// (element j of lane half h of an accumulator used as a 32x32x16 B operand is accumulator row 16s + 8(j>>2) + 4h + (j&3))
void pack_ffn_w2(const float* w2, int dff, uint16_t* out) {
    for (int ft = 0; ft < dff / 32; ++ft)
        for (int s = 0; s < 2; ++s)
            for (int nt = 0; nt < 8; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int f = 32 * ft + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                        out[((((size_t)ft * 2 + s) * 8 + nt) * 64 + lane) * 8 + j] =
                            bf16_bits(w2[(size_t)(32 * nt + (lane & 31)) * dff + f]);
                    }
}
