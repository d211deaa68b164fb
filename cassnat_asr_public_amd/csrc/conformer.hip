// Element-wise / small kernels of the conformer convolution module (reference: src/models/modules/conformer_related.py:15-44):
//   pointwise conv (a GEMM, gemm.hip) -> GLU -> depthwise conv over time -> GroupNorm(1, C) over the (C x T) image of each
//   utterance (padded frames included, as the reference does) -> Swish -> pointwise conv (GEMM).
// All HBM/latency-bound row work; the GroupNorm statistics are accumulated in double (they are a sum over up to
// T' x C = 64 000 values per utterance and feed every element of the sublayer).
#include <cstdlib>

#include "kernels.h"

template <typename T>
__global__ void glu_kernel(const T* __restrict__ in, T* __restrict__ out, long long M, int d) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * d) return;
    const long long m = i / d;
    const int c = (int)(i - m * d);
    const float a = cn_ld_elem<T>(in, m, 2 * d, c), g = cn_ld_elem<T>(in, m, 2 * d, d + c);
    cn_st_elem<T>(out, m, d, c, a * (1.f / (1.f + __expf(-g))));  // F.glu(dim=channels): first half * sigmoid(second half)
}

// y[b][t][c] = bias[c] + sum_k w[c][k] x[b][t + k - pad][c], zero outside [0, L)   (nn.Conv1d(groups = C), stride 1)
template <typename T>
__global__ void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                              float* __restrict__ y, int B, int L, int d, int k) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * L * d) return;
    const int c = (int)(i % d);
    const long long bt = i / d;
    const int t = (int)(bt % L);
    const long long b = bt / L;
    const int pad = (k - 1) / 2;
    float acc = bias[c];
    for (int j = 0; j < k; ++j) {
        const int tt = t + j - pad;
        if (tt >= 0 && tt < L) acc = fmaf(w[c * k + j], cn_ld_elem<T>(x, b * L + tt, d, c), acc);
    }
    y[i] = acc;
}

// per-utterance sum and sum of squares over the L x d image (one workgroup per utterance)
__global__ __launch_bounds__(1024) void groupnorm_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int n) {
    __shared__ double s1[16], s2[16];
    const float* p = x + (long long)blockIdx.x * n;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double v = p[i];
        a += v;
        b += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s1[threadIdx.x >> 6] = a;
        s2[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 16; ++i) {
            ta += s1[i];
            tb += s2[i];
        }
        stats[2 * blockIdx.x] = ta;
        stats[2 * blockIdx.x + 1] = tb;
    }
}

// out = swish(gamma[c] * (x - mean_b) * rsqrt(var_b + eps) + beta[c]), biased variance (F.group_norm)
template <typename T>
__global__ void groupnorm_swish_kernel(const float* __restrict__ x, const double* __restrict__ stats, const float* __restrict__ gw,
                                       const float* __restrict__ gb, T* __restrict__ out, int B, int L, int d, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)L * d;
    if (i >= (long long)B * n) return;
    const int b = (int)(i / n), c = (int)(i % d);
    const double mean = stats[2 * b] / (double)n;
    const double var = stats[2 * b + 1] / (double)n - mean * mean;
    const float inv = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    const float v = (x[i] - (float)mean) * inv * gw[c] + gb[c];
    cn_st_elem<T>(out, i / d, d, c, v * (1.f / (1.f + __expf(-v))));
}

int launch_glu(int prec, const void* in, void* out, int M, int d, hipStream_t s) {
    if (M <= 0) return 0;
    const long long n = (long long)M * d;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (prec == CN_PREC_F32)
        hipLaunchKernelGGL(glu_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, (float*)out, (long long)M, d);
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(glu_kernel<split_t>, dim3(blocks), dim3(256), 0, s, (const split_t*)in, (split_t*)out, (long long)M, d);
    else
        hipLaunchKernelGGL(glu_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)in, (bf16*)out, (long long)M, d);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// Tiled form for the kernel sizes the recipes use: a thread owns one channel and TT consecutive frames; the K taps sit in
// registers and every input frame of the window is loaded once (TT + K - 1 loads per TT outputs instead of K per output).
// Each output still accumulates bias, tap 0, tap 1, ... in that order, so the result equals dwconv_kernel's bit for bit.
template <typename T, int K, int TT>
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int L, int d) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= d) return;
    const int tiles = (L + TT - 1) / TT;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x - b * tiles) * TT;
    constexpr int pad = (K - 1) / 2;
    float wk[K], acc[TT];
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j] = w[c * K + j];
    const float bz = bias[c];
#pragma unroll
    for (int o = 0; o < TT; ++o) acc[o] = bz;
#pragma unroll
    for (int tt = 0; tt < TT + K - 1; ++tt) {
        const int t = t0 + tt - pad;
        if (t >= 0 && t < L) {  // a frame outside the utterance contributes nothing (and is skipped by dwconv_kernel as well)
            const float v = cn_ld_elem<T>(x, (long long)b * L + t, d, c);
#pragma unroll
            for (int o = 0; o < TT; ++o) {
                const int j = tt - o;
                if (j >= 0 && j < K) acc[o] = fmaf(wk[j], v, acc[o]);
            }
        }
    }
    float* yb = y + (long long)b * L * d + c;
#pragma unroll
    for (int o = 0; o < TT; ++o)
        if (t0 + o < L) yb[(long long)(t0 + o) * d] = acc[o];
}

template <typename T, int K>
static void launch_dwconv_tiled(const void* x, const float* w, const float* bias, float* y, int B, int L, int d, hipStream_t s) {
    constexpr int TT = 32;
    const dim3 grid((unsigned)(B * ((L + TT - 1) / TT)), (unsigned)((d + 255) / 256));
    hipLaunchKernelGGL((dwconv_tiled_kernel<T, K, TT>), grid, dim3(256), 0, s, (const T*)x, w, bias, y, L, d);
}

int launch_dwconv(int prec, const void* x, const float* w, const float* bias, float* y, int B, int L, int d, int k, hipStream_t s) {
    if (B * L <= 0) return 0;
    static const bool naive = cn_exp_env("CASSNAT_DWCONV_NAIVE") != nullptr;
#define DW_CASE(KK)                                                                                   \
    case KK:                                                                                          \
        if (prec == CN_PREC_F32) launch_dwconv_tiled<float, KK>(x, w, bias, y, B, L, d, s);           \
        else if (prec == CN_PREC_X3) launch_dwconv_tiled<split_t, KK>(x, w, bias, y, B, L, d, s);     \
        else launch_dwconv_tiled<bf16, KK>(x, w, bias, y, B, L, d, s);                                \
        CN_HIP_CHECK(hipGetLastError());                                                              \
        return 0;
    if (!naive) switch (k) {
        DW_CASE(3) DW_CASE(7) DW_CASE(15) DW_CASE(31)
        default: break;
    }
#undef DW_CASE
    const long long n = (long long)B * L * d;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (prec == CN_PREC_F32)
        hipLaunchKernelGGL(dwconv_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, w, bias, y, B, L, d, k);
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(dwconv_kernel<split_t>, dim3(blocks), dim3(256), 0, s, (const split_t*)x, w, bias, y, B, L, d, k);
    else
        hipLaunchKernelGGL(dwconv_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, w, bias, y, B, L, d, k);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_groupnorm_swish(int prec, const float* x, double* stats, const float* gw, const float* gb, void* out, int B, int L,
                           int d, float eps, hipStream_t s) {
    if (B * L <= 0) return 0;
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(B), dim3(1024), 0, s, x, stats, L * d);
    const long long n = (long long)B * L * d;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (prec == CN_PREC_F32)
        hipLaunchKernelGGL(groupnorm_swish_kernel<float>, dim3(blocks), dim3(256), 0, s, x, stats, gw, gb, (float*)out, B, L, d, eps);
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(groupnorm_swish_kernel<split_t>, dim3(blocks), dim3(256), 0, s, x, stats, gw, gb, (split_t*)out, B, L, d, eps);
    else
        hipLaunchKernelGGL(groupnorm_swish_kernel<bf16>, dim3(blocks), dim3(256), 0, s, x, stats, gw, gb, (bf16*)out, B, L, d, eps);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
