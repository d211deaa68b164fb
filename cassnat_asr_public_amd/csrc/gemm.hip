// Tiled MFMA GEMM for gfx950:  C[M][N] = epi(A[M][K] . W[N][K]^T + bias).
//
// Replaces every nn.Linear on the hot path (reference: src/models/modules/attention.py:57-66,
// positionff.py:15-16, embedding.py:118, cassnat.py:113) and, through the implicit-GEMM A loader,
// the second subsampling convolution (embedding.py:104, Conv2d(d,d,3,2,1)).
//
// Structure: 256 threads = 4 waves in a 2x2 grid; each wave owns (BM/2)x(BN/2) of the tile as
// 32x32 MFMA accumulators.  A and W slabs of 128 bytes of K per row (64 bf16 / 32 f32) are staged
// global -> registers -> LDS (double buffered: the loads of slab k+1 are issued before the MFMAs of
// slab k and written after them), 16 bytes per lane.  LDS rows are 128 B; chunk c of row r lives at
// chunk c ^ ((r>>1)&7), which makes the ds_read_b128 fragment reads and the ds_write_b128 staging
// writes bank-conflict free (MI355X LDS: 64 banks x 4 B, b128 reads serviced in 16-lane groups).
#include "kernels.h"

struct GemmParams {
    const unsigned char* A;
    const unsigned char* W;
    const float* bias;
    void* C;
    const float* resid;
    const float* pe;
    long long lda_bytes;
    int ldc, ldr;
    int M, N, K;
    int epi;
    int pe_period;
    float scale, resid_scale;
    const float* w_inv_scale;
    float acc_scale, c_scale;  // fp8 operands: product -> real units; fp8 output: real units -> the consumer's scale
    int ntn;  // tiles along N
    int cT1, cF1, cC, cT2, cF2;
};

// FULLK: K is exactly 4 slabs (256 bf16 / 128 f32): all slabs are loaded up front and staged behind ONE barrier
// instead of four load -> barrier round trips (the K=256 projections are latency-, not throughput-bound).
template <typename T, typename TC, int BM, int BN, bool CONV, bool FULLK = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    constexpr int ROWB = 128;
    constexpr int BK = ROWB / (int)sizeof(T);
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int BUF = (BM + BN) * ROWB;
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so block b and
    // b+8 share an L2.  Give every XCD a contiguous run of tiles (bijective for any grid size): the N-tiles of one
    // M-tile, which read the same A rows, and M-neighbours, which share conv halo rows, then meet in one L2.
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, rm = nwg & 7;
    const int tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (blockIdx.x >> 3);
    const int tile_m = tile / p.ntn, tile_n = tile % p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int st_off = lrow * ROWB + ((lchunk ^ ((lrow >> 1) & 7)) << 4);  // + 32*i rows keeps the swizzle

    // ---- per-thread global sources
    const unsigned char* a_src[A_IT];
    const unsigned char* w_src[B_IT];
    int cv_t[A_IT], cv_f[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.M;
        if (m >= p.M) m = p.M - 1;
        if (CONV) {
            const int f2 = m % p.cF2;
            const int bt = m / p.cF2;
            const int t2 = bt % p.cT2, b = bt / p.cT2;
            cv_t[i] = 2 * t2 - 1;
            cv_f[i] = 2 * f2 - 1;
            a_src[i] = p.A + ((long long)b * p.cT1 * p.cF1) * p.cC * (long long)sizeof(T) + lchunk * 16;
        } else {
            a_src[i] = p.A + (long long)m * p.lda_bytes + lchunk * 16;
        }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int n = n0 + lrow + 32 * i;
        if (n >= p.N) n = p.N - 1;
        w_src[i] = p.W + (long long)n * p.K * (long long)sizeof(T) + lchunk * 16;
    }

    uint4 a_reg[A_IT], w_reg[B_IT];
    // Implicit-GEMM K order: channel block outermost, the 9 taps innermost.  An input element is used by up to 2.25
    // taps of one tile; with the taps adjacent in time those re-reads are ~32 KB of traffic apart (L2 hits) instead of
    // a whole channel sweep apart.  The weight slab for step kt is k-offset tap*C + c0 of the [Cout][(kh,kw,Cin)] matrix.
    auto load_slab = [&](int kt) {
        if (CONV) {
            const int cb = kt / 9, tap = kt - 9 * cb, c0 = cb * BK;
            const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int t1 = cv_t[i] + kh, f1 = cv_f[i] + kw;
                const bool ok = a_ok[i] && t1 >= 0 && t1 < p.cT1 && f1 >= 0 && f1 < p.cF1;
                a_reg[i] = ok ? ld16(a_src[i] + ((long long)(t1 * p.cF1 + f1) * p.cC + c0) * (long long)sizeof(T))
                              : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) a_reg[i] = ld16(a_src[i] + (long long)kt * ROWB);
        }
        if (CONV) {
            const int cb = kt / 9, tap = kt - 9 * cb;
            const long long koff = ((long long)tap * p.cC + (long long)cb * BK) * (long long)sizeof(T);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) w_reg[i] = ld16(w_src[i] + koff);
        } else {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) w_reg[i] = ld16(w_src[i] + (long long)kt * ROWB);
        }
    };
    auto store_slab = [&](int buf) {
        unsigned char* a_dst = smem + buf * BUF + st_off;
        unsigned char* w_dst = a_dst + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) st16(a_dst + i * 32 * ROWB, a_reg[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) st16(w_dst + i * 32 * ROWB, w_reg[i]);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (row part fixed per thread, chunk part per k-step)
    const int a_row0 = wm * WM + l31, w_row0 = wn * WN + l31;

    auto compute_slab = [&](int buf) {
        const unsigned char* a_base = smem + buf * BUF;
        const unsigned char* w_base = a_base + BM * ROWB;
        if constexpr (__is_same(T, split_t)) {
            // a 128-byte slab row = one 32-element group: chunks 0-3 the hi halves of k = 8 c .. 8 c + 7, chunks 4-7 the lo halves
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = 2 * ks + half;
                split_frag af[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int r = a_row0 + 32 * i, sw = (r >> 1) & 7;
                    af[i].hi = as_frag<bf16>(ld16(a_base + r * ROWB + ((ch ^ sw) << 4)));
                    af[i].lo = as_frag<bf16>(ld16(a_base + r * ROWB + (((ch + 4) ^ sw) << 4)));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int r = w_row0 + 32 * j, sw = (r >> 1) & 7;
                    wf[j].hi = as_frag<bf16>(ld16(w_base + r * ROWB + ((ch ^ sw) << 4)));
                    wf[j].lo = as_frag<bf16>(ld16(w_base + r * ROWB + (((ch + 4) ^ sw) << 4)));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_frag(af[i], wf[j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
                const int chunk = 2 * cs + half;
                frag_t af[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int r = a_row0 + 32 * i;
                    af[i] = as_frag<T>(ld16(a_base + r * ROWB + ((chunk ^ ((r >> 1) & 7)) << 4)));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int r = w_row0 + 32 * j;
                    wf[j] = as_frag<T>(ld16(w_base + r * ROWB + ((chunk ^ ((r >> 1) & 7)) << 4)));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_frag(af[i], wf[j], acc[i][j]);
            }
        }
    };

    const int nk = p.K / BK;
    if constexpr (FULLK) {
        uint4 a_all[4][A_IT], w_all[4][B_IT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) a_all[kt][i] = ld16(a_src[i] + (long long)kt * ROWB);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) w_all[kt][i] = ld16(w_src[i] + (long long)kt * ROWB);
        }
        // keep all 4*(A_IT+B_IT) loads in flight together: without this hipcc re-uses one register quad and
        // serialises load -> wait -> ds_write sixteen times to minimise VGPRs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            unsigned char* a_dst = smem + kt * BUF + st_off;
            unsigned char* w_dst = a_dst + BM * ROWB;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) st16(a_dst + i * 32 * ROWB, a_all[kt][i]);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) st16(w_dst + i * 32 * ROWB, w_all[kt][i]);
        }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) compute_slab(kt);
    } else {
        load_slab(0);
        store_slab(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) load_slab(kt + 1);
            compute_slab(cur);
            if (kt + 1 < nk) store_slab(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: lane owns column n, 16 rows per 32x32 accumulator
    TC* C = reinterpret_cast<TC*>(p.C);
    float deq = 1.f;
    if constexpr (sizeof(T) == 1) deq = p.acc_scale * (p.w_inv_scale ? *p.w_inv_scale : 1.f);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + 32 * j + l31;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + 32 * i + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if constexpr (sizeof(T) == 1) v *= deq;
                v += bias;
                if (p.epi & CN_EPI_RELU) v = fmaxf(v, 0.f);
                if (p.epi & CN_EPI_SWISH) v = v * (1.f / (1.f + __expf(-v)));
                if (p.epi & CN_EPI_EMBED) v = v * p.scale + (p.pe ? p.pe[(long long)(m % p.pe_period) * p.N + n] : 0.f);
                if (p.epi & CN_EPI_RESID) v = p.resid[(long long)m * p.ldr + n] + p.resid_scale * v;
                if constexpr (sizeof(TC) == 1) v *= p.c_scale;
                if constexpr (__is_same(TC, split_t)) {
                    unsigned char* cb = reinterpret_cast<unsigned char*>(p.C) + (long long)m * p.ldc * 4 + cn_split_off((size_t)n);
                    const bf16 hi = (bf16)v;
                    *reinterpret_cast<bf16*>(cb) = hi;
                    *reinterpret_cast<bf16*>(cb + 64) = (bf16)(v - (float)hi);
                } else {
                    C[(long long)m * p.ldc + n] = from_f32<TC>(v);
                }
            }
        }
    }
}

template <typename T, typename TC, int BM, int BN, bool CONV, bool FULLK = false>
static int run_gemm(const GemmArgs& a, hipStream_t s) {
    constexpr int BK = 128 / (int)sizeof(T);
    GemmParams p;
    p.A = (const unsigned char*)a.A;
    p.W = (const unsigned char*)a.W;
    p.bias = a.bias;
    p.C = a.C;
    p.resid = a.resid;
    p.pe = a.pe;
    p.lda_bytes = (long long)a.lda * sizeof(T);
    p.ldc = a.ldc;
    p.ldr = a.ldr;
    p.M = a.M;
    p.N = a.N;
    p.K = a.K;
    p.epi = a.epi;
    p.pe_period = a.pe_period > 0 ? a.pe_period : 1;
    p.scale = a.scale;
    p.resid_scale = a.resid_scale;
    p.acc_scale = a.acc_scale;
    p.w_inv_scale = a.w_inv_scale;
    p.c_scale = a.c_scale;
    p.ntn = cn_ceil_div(a.N, BN);
    p.cT1 = a.cT1;
    p.cF1 = a.cF1;
    p.cC = a.cC;
    p.cT2 = a.cT2;
    p.cF2 = a.cF2;
    const int ntm = cn_ceil_div(a.M, BM);
    const size_t lds = (FULLK ? 4 : 2) * (size_t)(BM + BN) * 128;
    auto kern = gemm_kernel<T, TC, BM, BN, CONV, FULLK>;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.mark(attr_dev);
    }
    (void)BK;
    hipLaunchKernelGGL(kern, dim3(ntm * p.ntn), dim3(256), lds, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T> static int dispatch_gemm(const GemmArgs& a, hipStream_t s) {
    constexpr int BK = 128 / (int)sizeof(T);
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || a.K % BK != 0) {
        cn_set_error("gemm: K=" + std::to_string(a.K) + " must be a positive multiple of " + std::to_string(BK));
        return -1;
    }
    if (!a.conv && ((a.lda * sizeof(T)) % 16 != 0)) {
        cn_set_error("gemm: lda must keep rows 16-byte aligned");
        return -1;
    }
    if (a.conv) {
        if (a.cC % BK != 0 || a.K != 9 * a.cC || a.M != a.cB * a.cT2 * a.cF2) {
            cn_set_error("gemm(conv): inconsistent implicit-GEMM shape");
            return -1;
        }
        return run_gemm<T, T, 128, 128, true>(a, s);
    }
    // Tile choice: 128x128 when that still yields >= 2 workgroups per CU, else 64x64 to fill 256 CUs.
    const long long big = (long long)cn_ceil_div(a.M, 128) * cn_ceil_div(a.N, 128);
    const bool use_big = big >= 512;
    const bool fullk = a.K == 4 * BK;
    if (a.c_f32) {
        if (use_big) return run_gemm<T, float, 128, 128, false>(a, s);
        return fullk ? run_gemm<T, float, 64, 64, false, true>(a, s) : run_gemm<T, float, 64, 64, false>(a, s);
    }
    if (use_big) return run_gemm<T, T, 128, 128, false>(a, s);
    return fullk ? run_gemm<T, T, 64, 64, false, true>(a, s) : run_gemm<T, T, 64, 64, false>(a, s);
}

// fp8 operands (the encoder products of config 5): bf16 / fp32 / fp8 output, no implicit-GEMM form
static int dispatch_gemm_fp8(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.conv || a.K <= 0 || a.K % 128 != 0 || a.lda % 16 != 0) {
        cn_set_error("gemm(fp8): K must be a positive multiple of 128 and rows 16-byte aligned; no convolution form");
        return -1;
    }
    const bool use_big = (long long)cn_ceil_div(a.M, 128) * cn_ceil_div(a.N, 128) >= 512;
    if (a.c_f32) return use_big ? run_gemm<fp8_t, float, 128, 128, false>(a, s) : run_gemm<fp8_t, float, 64, 64, false>(a, s);
    if (a.c_fp8) return use_big ? run_gemm<fp8_t, fp8_t, 128, 128, false>(a, s) : run_gemm<fp8_t, fp8_t, 64, 64, false>(a, s);
    return use_big ? run_gemm<fp8_t, bf16, 128, 128, false>(a, s) : run_gemm<fp8_t, bf16, 64, 64, false>(a, s);
}

int launch_gemm(int prec, const GemmArgs& a, hipStream_t s) {
    if (a.ab_fp8) return dispatch_gemm_fp8(a, s);
    if (a.conv && a.conv_halo) {
        if (!(a.epi == CN_EPI_RELU && a.ldc == a.N && conv2_dma_applies(prec, a.cC, a.N))) {
            cn_set_error("gemm: a haloed conv input is only understood by the bf16 256-channel conv2 kernel");
            return -1;
        }
        return launch_conv2_dma(a.A, a.W, a.bias, a.C, a.cB, a.cT1, a.cF1, a.cT2, a.cF2, s);
    }
    // K-deep products onto all 256 output columns with the embedding epilogue (linear_out): the LDS-DMA tile kernel
    if (!a.conv && a.epi == CN_EPI_EMBED && a.c_f32 && a.ldc == a.N && a.K >= 1024 && linear256_dma_applies(prec, a.N, a.K))
        return launch_linear256_dma(a.A, a.lda, a.W, a.bias, (float*)a.C, a.M, a.K, a.scale, a.pe, a.pe_period, s);
    if (prec == CN_PREC_X3) {
        if (a.lda % 32 != 0 || (!a.c_f32 && a.ldc % 32 != 0)) {
            cn_set_error("gemm: split-bf16 operands need row strides that are multiples of 32 elements");
            return -1;
        }
        return dispatch_gemm<split_t>(a, s);
    }
    return prec == CN_PREC_F32 ? dispatch_gemm<float>(a, s) : dispatch_gemm<bf16>(a, s);
}
