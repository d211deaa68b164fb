// Fused generator for gfx950 (bf16 MFMA, d_model = 256):  per row  argmax_v and max_v of  log_softmax(W x + b)
// Replaces Generator.forward + argmax / topk(1) of the reference (src/models/cassnat.py:110-113, 378, 611) for the greedy
// path: the (M, V) logits tensor (160 MB at M = 8000, V = 5000) never exists.  (Beam search and the capture mode, which
// need full log-probability rows, keep the GEMM + row kernel.)
//
// Same streaming machinery as the fused FFN (fused.hip): a workgroup owns 32*MT rows; its activations sit in registers
// as MFMA B fragments; every wave streams ITS quarter of the vocabulary as pre-tiled 1-KiB weight fragments by LDS-DMA
// into a private 32-slot ring (2 vocabulary tiles of 32 rows x 16 k-steps) with counted vmcnt waits.  The product is
// computed swapped (logits^T: vocabulary on accumulator rows, the row index on the lane) so the running max / arg-max /
// sum-of-exponentials update is 16 in-lane values per tile; halves and waves are merged once at the end.
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

constexpr int GM_RING_BYTES = 32 * 1024;

// LSE = false: the arg-max alone (no exponentials, no log-sum-exp): what the CTC alignment of the greedy path needs
template <int MT, bool GATHER, bool LSE = true>
__global__ __launch_bounds__(256) void genmax_kernel(GenmaxParams p) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned char* ring = smem + wave_u * GM_RING_BYTES;
    float* merge = reinterpret_cast<float*>(smem);  // epilogue: [4 waves][BM][4] (aliases the rings)

    const int nrt = p.vtw / 2;  // ring tiles (2 vocabulary tiles each) per wave
    const uint4* w = p.wp + (long long)wave_u * p.vtw * 16 * 64 + lane;
#define GM_DMA(src, slot)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(ring + (slot) * 1024), 16, 0, 0)
    // biases of this wave in 2*ceil(vtw/2)... registers: register j, lane 32*p+i = bias of row i of vocabulary tile 2j+p
    constexpr int NBQ = 24;  // up to 48 vocabulary tiles per wave (V <= 6144)
    float bq[NBQ];
#pragma unroll
    for (int j = 0; j < NBQ; ++j) {
        const int pos = 2 * j + half;
        bq[j] = pos < p.vtw ? p.bp[(long long)wave_u * p.vtw * 32 + pos * 32 + l31] : 0.f;
    }
    // activations as B fragments: lane holds h[m = l31 (+32 mt)][16 ks + 8 half + j]
    bf16x8 ef[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m0 + 32 * mt + l31;
        if (m >= p.M) m = p.M - 1;
        const unsigned char* row = reinterpret_cast<const unsigned char*>(p.h + (long long)m * 256) + 16 * half;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) ef[mt][ks] = __builtin_bit_cast(bf16x8, ld16(row + 32 * ks));
    }
    // keep the plain loads above out of the DMA-pipelined region (hipcc would drain the DMA queue for them)
#pragma unroll
    for (int j = 0; j < NBQ; ++j) asm volatile("" : "+v"(bq[j]));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(ef[mt][ks]));
#pragma unroll
    for (int i = 0; i < 32; ++i) GM_DMA(w + (long long)i * 64, i);

    const unsigned slot_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(ring + lane * 16);
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

#define GM_STR2(x) #x
#define GM_STR(x) GM_STR2(x)
    // group g (0-3: first vocabulary tile of the ring tile, 4-7: second): fragments of k-steps 4(g&3)..+3
#define GM_GROUP(g, WAITN, REFILL)                                                                            \
    {                                                                                                         \
        bf16x8 wf0, wf1, wf2, wf3;                                                                            \
        asm volatile("s_waitcnt vmcnt(" GM_STR(WAITN) ")\n\t"                                                 \
                     "ds_read_b128 %0, %4 offset:" GM_STR((4 * (g) + 0) * 1024) "\n\t"                        \
                     "ds_read_b128 %1, %4 offset:" GM_STR((4 * (g) + 1) * 1024) "\n\t"                        \
                     "ds_read_b128 %2, %4 offset:" GM_STR((4 * (g) + 2) * 1024) "\n\t"                        \
                     "ds_read_b128 %3, %4 offset:" GM_STR((4 * (g) + 3) * 1024)                                \
                     : "=&v"(wf0), "=&v"(wf1), "=&v"(wf2), "=&v"(wf3)                                         \
                     : "v"(slot_a)                                                                            \
                     : "memory");                                                                             \
        REFILL                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf0), "+v"(wf1), "+v"(wf2), "+v"(wf3) :: "memory");        \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                   \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, ef[mt][4 * ((g) & 3) + 0], acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, ef[mt][4 * ((g) & 3) + 1], acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2, ef[mt][4 * ((g) & 3) + 2], acc[mt], 0, 0, 0); \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf3, ef[mt][4 * ((g) & 3) + 3], acc[mt], 0, 0, 0); \
        }                                                                                                     \
    }
#define GM_REFILL(g) { _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
        GM_DMA(w + ((long long)rnext * 32 + 4 * (g) + j) * 64, 4 * (g) + j); }
#define GM_ZERO()                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    // fold one finished vocabulary tile (position pos within this wave's slice) into the running statistics
#define GM_FOLD(pos)                                                                                          \
    {                                                                                                         \
        const int src0 = 32 * ((pos) & 1) + 4 * half;                                                         \
        const int vbase = 32 * (wave_u * p.vtw + (pos)) + 4 * half;                                           \
        float bv[16];                                                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int e = 0; e < 4; ++e)           \
            bv[4 * g + e] = __shfl(bq[0], src0 + 8 * g + e);                                                  \
        if ((pos) & 1) { _Pragma("unroll") for (int j = 0; j < NBQ - 1; ++j) bq[j] = bq[j + 1]; }             \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                   \
            float tmax = -INFINITY;                                                                           \
            int tidx = 0;                                                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
                const float v = acc[mt][r] + bv[r];                                                           \
                acc[mt][r] = v;                                                                               \
                if (v > tmax) { tmax = v; tidx = vbase + (r & 3) + 8 * (r >> 2); }                            \
                if constexpr (GATHER) { if (vbase + (r & 3) + 8 * (r >> 2) == tg[mt]) tv[mt] = v; }           \
            }                                                                                                 \
            if (tmax > m_run[mt]) {                                                                           \
                if constexpr (LSE) s_run[mt] *= __expf(m_run[mt] - tmax);                                     \
                m_run[mt] = tmax;                                                                             \
                i_run[mt] = tidx;                                                                             \
            }                                                                                                 \
            if constexpr (LSE) {                                                                              \
                float ps = 0.f;                                                                               \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) ps += __expf(acc[mt][r] - m_run[mt]);          \
                s_run[mt] += ps;                                                                              \
            }                                                                                                 \
        }                                                                                                     \
    }

    f32x16 acc[MT];
    int rt = 0, rnext = 0;
    for (; rt + 1 < nrt; ++rt) {
        GM_ZERO()
        if (rt == 0) {
            GM_GROUP(0, 28, )
        } else {
            GM_GROUP(0, 24, GM_REFILL(7))
        }
        rnext = rt + 1;
        GM_GROUP(1, 24, GM_REFILL(0)) GM_GROUP(2, 24, GM_REFILL(1)) GM_GROUP(3, 24, GM_REFILL(2))
        GM_FOLD(2 * rt)
        GM_ZERO()
        GM_GROUP(4, 24, GM_REFILL(3)) GM_GROUP(5, 24, GM_REFILL(4)) GM_GROUP(6, 24, GM_REFILL(5))
        GM_GROUP(7, 24, GM_REFILL(6))
        GM_FOLD(2 * rt + 1)
    }
    {
        GM_ZERO()
        if (rt == 0) {
            GM_GROUP(0, 28, )
        } else {
            GM_GROUP(0, 24, GM_REFILL(7))
        }
        GM_GROUP(1, 24, ) GM_GROUP(2, 20, ) GM_GROUP(3, 16, )
        GM_FOLD(2 * rt)
        GM_ZERO()
        GM_GROUP(4, 12, ) GM_GROUP(5, 8, ) GM_GROUP(6, 4, ) GM_GROUP(7, 0, )
        GM_FOLD(2 * rt + 1)
    }
#undef GM_GROUP
#undef GM_REFILL
#undef GM_ZERO
#undef GM_FOLD
#undef GM_DMA

    // ---- merge the two lane halves, then the four waves (ties: the lower vocabulary index wins, as torch.argmax)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float om = __shfl_xor(m_run[mt], 32), os = __shfl_xor(s_run[mt], 32);
        const int oi = __shfl_xor(i_run[mt], 32);
        const float nm = fmaxf(m_run[mt], om);
        s_run[mt] = s_run[mt] * __expf(m_run[mt] - nm) + os * __expf(om - nm);
        if (om > m_run[mt] || (om == m_run[mt] && oi < i_run[mt])) i_run[mt] = oi;
        m_run[mt] = nm;
        if constexpr (GATHER) tv[mt] = fmaxf(tv[mt], __shfl_xor(tv[mt], 32));  // exactly one lane half of one wave saw it
    }
    __syncthreads();  // all rings idle (last wait was vmcnt(0)); reuse LDS for the cross-wave merge
    if (half == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float* d = merge + ((wave * BM) + 32 * mt + l31) * 4;
            d[0] = m_run[mt];
            d[1] = s_run[mt];
            d[2] = __int_as_float(i_run[mt]);
            d[3] = tv[mt];
        }
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
        float bm = merge[tid * 4 + 0], bs = merge[tid * 4 + 1];
        int bi = __float_as_int(merge[tid * 4 + 2]);
        float bt = merge[tid * 4 + 3];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) {
            const float* d = merge + (wv * BM + tid) * 4;
            const float om = d[0], os = d[1];
            bt = fmaxf(bt, d[3]);
            const int oi = __float_as_int(d[2]);
            const float nm = fmaxf(bm, om);
            bs = bs * __expf(bm - nm) + os * __expf(om - nm);
            if (om > bm || (om == bm && oi < bi)) bi = oi;
            bm = nm;
        }
        if constexpr (GATHER) {
            const int m = m0 + tid;
            p.tgt_lp[(long long)(m / p.tgt_U) * p.tgt_ld + (m % p.tgt_U)] = (bt - bm) - logf(bs);
        } else {
            p.arg[m0 + tid] = bi;
            if constexpr (LSE) p.maxlp[m0 + tid] = -logf(bs);
        }
    }
}

template <int MT, bool GATHER, bool LSE> static int launch_genmax_variant(const GenmaxParams& p, hipStream_t s) {
    constexpr int lds = 4 * GM_RING_BYTES;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)genmax_kernel<MT, GATHER, LSE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
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

int launch_genmax(const GenmaxArgs& a, hipStream_t s) {
    const int vtw = genmax_vtw(a.V);
    if (a.d != 256 || a.V < 1 || vtw > 48) {
        cn_set_error("genmax: needs d_model == 256 and V <= 6144");
        return -1;
    }
    if (a.M <= 0) return 0;
    GenmaxParams p;
    p.h = reinterpret_cast<const bf16*>(a.h);
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
    if (a.tgt) {
        if (!a.tgt_lp || a.tgt_U < 1 || a.tgt_ld < a.tgt_U || a.M % a.tgt_U != 0) {
            cn_set_error("genmax: the target gather needs rows = B x U, an output buffer and ld >= U");
            return -1;
        }
        return a.M > 32 ? launch_genmax_variant<2, true, true>(p, s) : launch_genmax_variant<1, true, true>(p, s);
    }
    if (!a.maxlp)  // arg-max only
        return a.M > 32 ? launch_genmax_variant<2, false, false>(p, s) : launch_genmax_variant<1, false, false>(p, s);
    return a.M > 32 ? launch_genmax_variant<2, false, true>(p, s) : launch_genmax_variant<1, false, true>(p, s);
}

static inline uint16_t gm_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

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
