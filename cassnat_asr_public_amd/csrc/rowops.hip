// Row-wise, HBM-bound kernels of the hot path: the reference's custom LayerNorm, the generator's
// log-softmax + argmax, the padding mask, query-table broadcast and precision converts.
#include "kernels.h"

// ---------------------------------------------------------------------------------------------
// LayerNorm exactly as src/models/modules/norm.py:15-18: unbiased std (divide by d-1), eps added to
// the std (not the variance).  One wave per row, 16 bytes per lane per step, two-pass in registers.
// ---------------------------------------------------------------------------------------------
template <typename TY>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ a2,
                                                        const float* __restrict__ b2, TY* __restrict__ y, int M, int d,
                                                        float eps, float out_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long long)row * d;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = (i * 64 + lane) * 4;
        if (idx < d) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + idx);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = (i * 64 + lane) * 4;
        if (idx < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = v[i][j] - mean;
                ss = fmaf(c, c, ss);
            }
        }
    }
    const float denom = sqrtf(wave_sum(ss) / (float)(d - 1)) + eps;
    TY* yr = y + (long long)row * d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = (i * 64 + lane) * 4;
        if (idx < d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a2 + idx);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + idx);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = g[j] * (v[i][j] - mean) / denom + bb[j];
            if constexpr (sizeof(TY) == 1) {  // e4m3fn at the consumer's per-tensor scale (config 5)
                unsigned int w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) w |= (unsigned int)cn_f32_to_fp8(o[j] * out_scale) << (8 * j);
                *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(yr) + idx) = w;
            } else if constexpr (sizeof(TY) == 2) {
                bf16x4 ob;
#pragma unroll
                for (int j = 0; j < 4; ++j) ob[j] = (bf16)o[j];
                *reinterpret_cast<bf16x4*>(yr + idx) = ob;
            } else if constexpr (__is_same(TY, split_t)) {  // hi / lo halves of the element's 32-group (common.h)
                bf16x4 hi, lo;
                cn_split4(o, hi, lo);
                unsigned char* yb = reinterpret_cast<unsigned char*>(yr) + cn_split_off((size_t)idx);
                *reinterpret_cast<bf16x4*>(yb) = hi;
                *reinterpret_cast<bf16x4*>(yb + 64) = lo;
            } else {
                f32x4 of;
#pragma unroll
                for (int j = 0; j < 4; ++j) of[j] = o[j];
                *reinterpret_cast<f32x4*>(yr + idx) = of;
            }
        }
    }
}

int launch_layernorm(int prec, const float* x, const float* a2, const float* b2, void* y, int y_f32, int M, int d,
                     float eps, hipStream_t s) {
    if (d % 4 != 0 || d > 1024 || d < 2) {
        cn_set_error("layernorm: d must be a multiple of 4 in [4, 1024]");
        return -1;
    }
    if (M <= 0) return 0;
    const dim3 grid(cn_ceil_div(M, 4));
    if (prec == CN_PREC_X3 && !y_f32 && d % 32 != 0) {
        cn_set_error("layernorm: split-bf16 rows need d % 32 == 0");
        return -1;
    }
    if (y_f32 || prec == CN_PREC_F32)
        hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, s, x, a2, b2, (float*)y, M, d, eps, 1.f);
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(layernorm_kernel<split_t>, grid, dim3(256), 0, s, x, a2, b2, (split_t*)y, M, d, eps, 1.f);
    else
        hipLaunchKernelGGL(layernorm_kernel<bf16>, grid, dim3(256), 0, s, x, a2, b2, (bf16*)y, M, d, eps, 1.f);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// LayerNorm straight into e4m3fn at `scale` (the A operand of an fp8 product)
int launch_layernorm_fp8(const float* x, const float* a2, const float* b2, void* y, int M, int d, float eps, float scale,
                         hipStream_t s) {
    if (d % 4 != 0 || d > 1024 || d < 2) {
        cn_set_error("layernorm: d must be a multiple of 4 in [4, 1024]");
        return -1;
    }
    if (M <= 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel<fp8_t>, dim3(cn_ceil_div(M, 4)), dim3(256), 0, s, x, a2, b2, (fp8_t*)y, M, d, eps, scale);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// bf16 [M][K] (row stride ld) -> e4m3fn [M][K] at `scale`, saturating: the A operand of an fp8 product whose producer
// writes bf16 (the attention context)
__global__ void quantize_fp8_kernel(const bf16* __restrict__ src, int ld, unsigned char* __restrict__ dst, int M, int K, float scale) {
    const long long n4 = (long long)M * (K / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / (K / 4);
        const int c = (int)(i - m * (K / 4)) * 4;
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + m * ld + c);
        unsigned int w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= (unsigned int)cn_f32_to_fp8((float)v[j] * scale) << (8 * j);
        *reinterpret_cast<unsigned int*>(dst + m * K + c) = w;
    }
}
int launch_quantize_fp8(const void* src_bf16, int ld, void* dst, int M, int K, float scale, hipStream_t s) {
    if (M <= 0 || K <= 0) return 0;
    if (K % 4 != 0 || ld % 4 != 0) {
        cn_set_error("quantize_fp8: K and the row stride must be multiples of 4");
        return -1;
    }
    const long long n4 = (long long)M * (K / 4);
    long long g = (n4 + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)g), dim3(256), 0, s, (const bf16*)src_bf16, ld, (unsigned char*)dst, M, K, scale);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Generator tail (src/models/cassnat.py:113 log_softmax, :378 argmax, :611 topk(1)): per row of
// fp32 logits the first-index argmax and its log-probability (x - max) - log(sum exp(x - max));
// optionally the row is rewritten in place as log-probabilities.  One workgroup per row.
// ---------------------------------------------------------------------------------------------
// temperature: Generator.forward(x, T) = log_softmax(proj(x) / T) (src/models/transformer.py:48-51); 1.0 = no division
__global__ __launch_bounds__(256) void logsoftmax_argmax_kernel(float* __restrict__ logits, int V, int ldl,
                                                                int* __restrict__ arg, float* __restrict__ maxlp,
                                                                int write_logp, float temperature) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_sum[4];
    float* p = logits + (long long)blockIdx.x * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (temperature != 1.0f) {
        for (int i = tid; i < V; i += 256) p[i] = p[i] / temperature;
        __syncthreads();
    }
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = p[i];
        if (v > best || (v == best && i < bidx)) {
            best = v;
            bidx = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ov > best || (ov == best && oi < bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if (lane == 0) {
        s_val[wave] = best;
        s_idx[wave] = bidx;
    }
    __syncthreads();
    best = s_val[0];
    bidx = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) {
            best = s_val[w];
            bidx = s_idx[w];
        }
    }
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf(p[i] - best);
    sum = wave_sum(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    const float lse = logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
    if (tid == 0) {
        arg[blockIdx.x] = bidx;
        maxlp[blockIdx.x] = -lse;
    }
    if (write_logp)
        for (int i = tid; i < V; i += 256) p[i] = (p[i] - best) - lse;
}

int launch_logsoftmax_argmax(float* logits, int M, int V, int ldl, int* arg, float* maxlp, int write_logp,
                             hipStream_t s) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(logsoftmax_argmax_kernel, dim3(M), dim3(256), 0, s, logits, V, ldl, arg, maxlp, write_logp, 1.0f);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_logsoftmax_temp(float* logits, int M, int V, int ldl, float temperature, int* arg, float* maxlp, hipStream_t s) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(logsoftmax_argmax_kernel, dim3(M), dim3(256), 0, s, logits, V, ldl, arg, maxlp, 1, temperature);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// per-row top-k of log-probs (beam_width > 1, src/models/cassnat.py:611). Row staged in LDS, k rounds.
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ logp, int V, int ldl, int k,
                                                   int* __restrict__ idx, float* __restrict__ val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = reinterpret_cast<float*>(smem);
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const float* p = logp + (long long)blockIdx.x * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < V; i += 256) row[i] = p[i];
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float v = row[i];
            if (v > best || (v == best && i < bidx)) {
                best = v;
                bidx = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bidx, o);
            if (ov > best || (ov == best && oi < bidx)) {
                best = ov;
                bidx = oi;
            }
        }
        if (lane == 0) {
            s_val[wave] = best;
            s_idx[wave] = bidx;
        }
        __syncthreads();
        best = s_val[0];
        bidx = s_idx[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) {
                best = s_val[w];
                bidx = s_idx[w];
            }
        }
        if (tid == 0) {
            idx[(long long)blockIdx.x * k + r] = bidx;
            val[(long long)blockIdx.x * k + r] = best;
            if (bidx < V) row[bidx] = -INFINITY;
        }
        __syncthreads();
    }
}

int launch_topk(const float* logp, int M, int V, int ldl, int k, int* idx, float* val, hipStream_t s) {
    if (k < 1 || k > 64 || V > 16384) {
        cn_set_error("topk: need 1 <= k <= 64 and V <= 16384");
        return -1;
    }
    if (M <= 0) return 0;
    hipLaunchKernelGGL(topk_kernel, dim3(M), dim3(256), (size_t)V * sizeof(float), s, logp, V, ldl, k, idx, val);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// Generator tail of the autoregressive step in one pass per row: log_softmax(logits / T) and its top-k (sorted descending,
// ties: lower index first).  Same per-element arithmetic as logsoftmax_argmax_kernel followed by topk_kernel - x / T, the
// row maximum, the sum of expf(x - max) in the same partition and order, (x - max) - lse per entry (in LDS) - without
// writing the (M, V) log-probabilities back and with the k selection rounds working on cached per-thread maxima (only the
// thread that owned the previous winner rescans its V / 256 entries).
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const float* __restrict__ logits, int V, int ldl, float temperature,
                                                              int k, int* __restrict__ idx, float* __restrict__ val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = reinterpret_cast<float*>(smem);
    __shared__ float s_val[4], s_sum[4];
    __shared__ int s_idx[4];
    const float* p = logits + (long long)blockIdx.x * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto local_best = [&](float& best, int& bidx) {
        best = -INFINITY;
        bidx = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float v = row[i];
            if (v > best || (v == best && i < bidx)) {
                best = v;
                bidx = i;
            }
        }
    };
    auto block_best = [&](float& best, int& bidx) {  // all threads return the winner
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bidx, o);
            if (ov > best || (ov == best && oi < bidx)) {
                best = ov;
                bidx = oi;
            }
        }
        if (lane == 0) {
            s_val[wave] = best;
            s_idx[wave] = bidx;
        }
        __syncthreads();
        best = s_val[0];
        bidx = s_idx[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) {
                best = s_val[w];
                bidx = s_idx[w];
            }
        __syncthreads();  // s_val / s_idx are rewritten by the next round
    };
    if (temperature != 1.0f)
        for (int i = tid; i < V; i += 256) row[i] = p[i] / temperature;
    else
        for (int i = tid; i < V; i += 256) row[i] = p[i];
    // (a thread only ever reads the entries it wrote until the first block_best: no barrier needed before it)
    float mybest;
    int myidx;
    local_best(mybest, myidx);
    float best = mybest;
    int bidx = myidx;
    block_best(best, bidx);
    const float rmax = best;
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf(row[i] - rmax);
    sum = wave_sum(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    const float lse = logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
    // the selection ranks the LOG-PROBABILITIES, as the two-kernel form and the reference do: two logits one ulp apart can
    // round to the same log-probability, and the tie then goes to the lower index
    for (int i = tid; i < V; i += 256) row[i] = (row[i] - rmax) - lse;
    local_best(mybest, myidx);
    for (int r = 0; r < k; ++r) {
        best = mybest;
        bidx = myidx;
        block_best(best, bidx);
        if (tid == 0) {
            idx[(long long)blockIdx.x * k + r] = bidx;
            val[(long long)blockIdx.x * k + r] = best;
        }
        if (bidx < V && (bidx & 255) == tid) {  // the owner retires the winner and finds its next candidate
            row[bidx] = -INFINITY;
            local_best(mybest, myidx);
        }
    }
}

int launch_logsoftmax_topk(const float* logits, int M, int V, int ldl, float temperature, int k, int* idx, float* val,
                           hipStream_t s) {
    if (k < 1 || k > 16 || V > 16384) {
        cn_set_error("logsoftmax_topk: need 1 <= k <= 16 and V <= 16384");
        return -1;
    }
    if (M <= 0) return 0;
    hipLaunchKernelGGL(logsoftmax_topk_kernel, dim3(M), dim3(256), (size_t)V * sizeof(float), s, logits, V, ldl, temperature, k, idx,
                       val);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Padding mask after 4x subsampling: src/tasks/cassnat_task.py:328 then embedding.py:121-122
// (mask[:, :, ::2][:, :, ::2]) => keymask[b][j] = feats[b][4j][0] != padding_idx.
// ---------------------------------------------------------------------------------------------
__global__ void keymask_kernel(const float* __restrict__ feats, int B, int T, int F, int Tp, int stride, float padding,
                               unsigned char* __restrict__ km) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tp) return;
    const int b = i / Tp, j = i - b * Tp;
    const int t = stride * j;
    km[i] = (t < T && feats[((long long)b * T + t) * F] != padding) ? 1 : 0;
}

int launch_keymask(const float* feats, int B, int T, int F, int Tp, int stride, float padding, unsigned char* km,
                   hipStream_t s) {
    hipLaunchKernelGGL(keymask_kernel, dim3(cn_ceil_div(B * Tp, 256)), dim3(256), 0, s, feats, B, T, F, Tp, stride,
                       padding, km);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// The per-utterance records of a merged engine pass (kernels.h: UttMeta) from the pass's list of reference batches.
__global__ void expand_meta_kernel(SubList subs, UttMeta* __restrict__ meta, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int lo = 0, k = 0;
    for (; k < subs.n - 1 && b >= lo + subs.rows[k]; ++k) lo += subs.rows[k];
    const int T = subs.frames[k];
    const int T1 = (T - 1) / 2 + 1;
    UttMeta mrec;
    mrec.frames = T;
    mrec.tp = (T1 - 1) / 2 + 1;
    mrec.sub_lo = lo;
    mrec.sub_hi = lo + subs.rows[k];
    meta[b] = mrec;
}

int launch_expand_meta(const SubList& subs, UttMeta* meta, int B, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(expand_meta_kernel, dim3(cn_ceil_div(B, 256)), dim3(256), 0, s, subs, meta, B);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// extractor queries: pe[:ymax] repeated over the batch (src/models/cassnat.py:481)
__global__ void fill_queries_kernel(const float* __restrict__ table, float* __restrict__ out, int B, int U, int d) {
    const long long n4 = (long long)B * U * d / 4;
    const long long per = (long long)U * d / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(table)[i % per];
}

int launch_fill_queries(const float* table, float* out, int B, int U, int d, hipStream_t s) {
    const long long n4 = (long long)B * U * d / 4;
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(fill_queries_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, table, out, B, U, d);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// use_unimask (src/models/cassnat.py:486-488): prepend a zero embedding, drop the last row
__global__ void shift_right_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int U, int d) {
    const long long n = (long long)B * U * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long u = (i / d) % U;
        y[i] = u == 0 ? 0.f : x[i - d];
    }
}

int launch_shift_right(const float* x, float* y, int B, int U, int d, hipStream_t s) {
    const long long n = (long long)B * U * d;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(shift_right_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, B, U, d);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
template <typename TD, typename TS>
__global__ void convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<TD>(to_f32(src[i]));
}

// flat fp32 <-> split-bf16 (n % 32 == 0: contiguous rows whose length is a multiple of 32)
__global__ void convert_to_split_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        const float a[4] = {v[0], v[1], v[2], v[3]};
        bf16x4 hi, lo;
        cn_split4(a, hi, lo);
        unsigned char* d = dst + cn_split_off(4 * i);
        *reinterpret_cast<bf16x4*>(d) = hi;
        *reinterpret_cast<bf16x4*>(d + 64) = lo;
    }
}
__global__ void convert_from_split_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned char* p = src + cn_split_off(4 * i);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p), lo = *reinterpret_cast<const bf16x4*>(p + 64);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (float)hi[j] + (float)lo[j];
        *reinterpret_cast<f32x4*>(dst + 4 * i) = o;
    }
}

static unsigned convert_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

int launch_convert(int prec, const float* src, void* dst, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    if (prec == CN_PREC_X3 && n % 32 != 0) {
        cn_set_error("convert: split-bf16 tensors hold a multiple of 32 elements");
        return -1;
    }
    if (prec == CN_PREC_F32)
        CN_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(convert_to_split_kernel, dim3(convert_grid(n / 4)), dim3(256), 0, s, src, (unsigned char*)dst, n / 4);
    else
        hipLaunchKernelGGL((convert_kernel<bf16, float>), dim3(convert_grid(n)), dim3(256), 0, s, src, (bf16*)dst, n);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_convert_back(int prec, const void* src, float* dst, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    if (prec == CN_PREC_X3 && n % 32 != 0) {
        cn_set_error("convert: split-bf16 tensors hold a multiple of 32 elements");
        return -1;
    }
    if (prec == CN_PREC_F32)
        CN_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    else if (prec == CN_PREC_X3)
        hipLaunchKernelGGL(convert_from_split_kernel, dim3(convert_grid(n / 4)), dim3(256), 0, s, (const unsigned char*)src, dst,
                           n / 4);
    else
        hipLaunchKernelGGL((convert_kernel<float, bf16>), dim3(convert_grid(n)), dim3(256), 0, s, (const bf16*)src, dst,
                           n);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// rows of `width` 4-byte words from a [rows][src_ld] buffer into a [rows][dst_ld] one (a kernel, not hipMemcpy2DAsync: the
// runtime's pitched device-to-device copy takes a staging path of its own with occasional tens-of-milliseconds stalls)
__global__ void copy_rows_kernel(unsigned* __restrict__ dst, int dst_ld, const unsigned* __restrict__ src, int src_ld, int width,
                                 long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / width;
        const int c = (int)(i - r * width);
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}
int launch_copy_rows(void* dst, int dst_ld, const void* src, int src_ld, int width, int rows, hipStream_t s) {
    const long long total = (long long)width * rows;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(convert_grid((size_t)total)), dim3(256), 0, s, (unsigned*)dst, dst_ld, (const unsigned*)src,
                       src_ld, width, total);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void fill_int_kernel(int* p, size_t n, int v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill_int(int* p, size_t n, int v, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_int_kernel, dim3(convert_grid(n)), dim3(256), 0, s, p, n, v);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- ESA sampled alignment: label of frame i = second best iff its draw is 1 and exp(best log-prob) < threshold ----------
__global__ void esa_paths_kernel(const int* __restrict__ top2_idx, const float* __restrict__ top2_val,
                                 const unsigned char* __restrict__ select, float threshold, int* __restrict__ best, int M, int n) {
    // n paths per frame: best[g][i] for draw set g (select: [n][M], null = the best path)
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (long long)M * n) return;
    const int i = (int)(j % M);
    const int pick = (select && select[j] && expf(top2_val[2 * i]) < threshold) ? 1 : 0;
    best[j] = top2_idx[2 * i + pick];
}
int launch_esa_paths(const int* top2_idx, const float* top2_val, const unsigned char* select, float threshold, int* best, int M,
                     int n, hipStream_t s) {
    if (M <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(esa_paths_kernel, dim3((unsigned)(((long long)M * n + 255) / 256)), dim3(256), 0, s, top2_idx, top2_val, select,
                       threshold, best, M, n);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- TransformerLM input: x[b][u] = lut[tok[b][u]] * sqrt(d) + pe[u]   (embedding.py:71-78, 29-31) -----------------------
__global__ void lm_embed_kernel(const int* __restrict__ tok, int ld, const float* __restrict__ lut, const float* __restrict__ pe,
                                float* __restrict__ x, int B, int U, int d, float scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * U * d) return;
    const int c = (int)(i % d);
    const long long bu = i / d;
    const int u = (int)(bu % U), b = (int)(bu / U);
    x[i] = lut[(long long)tok[(long long)b * ld + u] * d + c] * scale + pe[(long long)u * d + c];
}
int launch_lm_embed(const int* tok, int ld, const float* lut, const float* pe, float* x, int B, int U, int d, float scale,
                    hipStream_t s) {
    const long long n = (long long)B * U * d;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lm_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tok, ld, lut, pe, x, B, U, d, scale);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void gather_logp_kernel(const float* __restrict__ logp, int V, const int* __restrict__ tgt, int ld,
                                   float* __restrict__ out, int B, int U) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * U) return;
    const int b = i / U, u = i - b * U;
    out[(long long)b * ld + u] = logp[(long long)i * V + tgt[(long long)b * ld + u]];
}
int launch_gather_logp(const float* logp, int V, const int* tgt, int ld, float* out, int B, int U, hipStream_t s) {
    if (B * U <= 0) return 0;
    hipLaunchKernelGGL(gather_logp_kernel, dim3(cn_ceil_div(B * U, 256)), dim3(256), 0, s, logp, V, tgt, ld, out, B, U);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// Global CMVN of a padded batch on the device: x[b][t][f] <- float((double(x) - mean[f]) / std[f]) for the frames t < len[b] of
// utterance b, later frames (collate's padding) untouched.  The reference does this per utterance on the host, in numpy's
// float64 with float64 statistics, and rounds to float32 when it collates (src/data/speech_loader.py:109-115, 147-149, 340): the
// same two IEEE operations and the same single rounding here, so the values are the reference's bit for bit - at HBM speed
// instead of the test-set loader's (which it was 3/4 of).
__global__ void cmvn_kernel(float* __restrict__ x, const int* __restrict__ len, const double* __restrict__ mean,
                            const double* __restrict__ sd, int T, int F) {
    const int b = blockIdx.y;
    const int lb = len[b];  // (clamped to the batch's frames: a bad length must not write past the utterance's rows)
    const long long n = (long long)(lb < 0 ? 0 : lb > T ? T : lb) * F;
    float* xb = x + (long long)b * T * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        xb[i] = (float)(((double)xb[i] - mean[f]) / sd[f]);
    }
}

// The reader's hand-over (SuperviseLoader.collate_fn, src/data/speech_loader.py:327-356, and the global CMVN of :109-115, 147-149) on
// the device: the utterances of an engine pass arrive PACKED - their archive rows back to back, exactly as they lie in the .ark, one
// DMA - and are spread over the padded batch here: out[r][t][:] = t < len[r] ? norm(packed[off[r] + t][:]) : pad, with
// norm(x) = float((double(x) - mean) / std) when statistics are given (the dataset's float64 arithmetic, one rounding: bit for
// bit) and x itself otherwise.  Replaces the host-side padded collate, the per-batch device copies and the separate CMVN pass.
// One workgroup per (utterance, 32-frame chunk); 16-byte accesses when F % 4 == 0.
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float* __restrict__ packed, const int* __restrict__ off, const int* __restrict__ len,
                                                          float* __restrict__ out, int T, int F, float pad, const double* __restrict__ mean,
                                                          const double* __restrict__ sd) {
    const int r = blockIdx.y;
    const int n = len[r] < 0 ? 0 : (len[r] > T ? T : len[r]);
    const long long t0 = (long long)blockIdx.x * 32;
    if (t0 >= T) return;
    const int tn = (int)((T - t0) < 32 ? (T - t0) : 32);
    const float* src = packed + ((long long)off[r] + t0) * F;
    float* dst = out + ((long long)r * T + t0) * F;
    const int total = tn * F, valid = (int)((n - t0) <= 0 ? 0 : ((n - t0) < tn ? (n - t0) : tn)) * F;
    if ((F & 3) == 0 && (((size_t)src | (size_t)dst) & 15) == 0) {
        for (int i = 4 * threadIdx.x; i < total; i += 4 * 256) {
            f32x4 v = {pad, pad, pad, pad};
            if (i < valid) {  // (valid is a multiple of F, F of 4: a quad never straddles the utterance's end)
                v = *reinterpret_cast<const f32x4*>(src + i);
                if (mean) {
                    const int f = i % F;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)(((double)v[e] - mean[f + e]) / sd[f + e]);
                }
            }
            *reinterpret_cast<f32x4*>(dst + i) = v;
        }
    } else {
        for (int i = threadIdx.x; i < total; i += 256) {
            float v = pad;
            if (i < valid) {
                v = src[i];
                if (mean) v = (float)(((double)v - mean[i % F]) / sd[i % F]);
            }
            dst[i] = v;
        }
    }
}

int launch_unpack_rows(const float* packed, const int* off, const int* len, float* out, int rows, int T, int F, float pad,
                       const double* mean, const double* sd, hipStream_t s) {
    if (rows <= 0 || T <= 0 || F <= 0) return 0;
    if (rows > 65535) {
        cn_set_error("unpack_rows: more than 65535 utterances in one pass");
        return -1;
    }
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)cn_ceil_div(T, 32), (unsigned)rows), dim3(256), 0, s, packed, off, len, out, T, F,
                       pad, mean, sd);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_cmvn(float* x, const int* len, const double* mean, const double* sd, int B, int T, int F, hipStream_t s) {
    if (B <= 0 || T <= 0 || F <= 0) return 0;
    const int per = cn_ceil_div(T * F, 256);
    hipLaunchKernelGGL(cmvn_kernel, dim3((unsigned)(per < 64 ? per : 64), (unsigned)B), dim3(256), 0, s, x, len, mean, sd, T, F);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- fp16 engine: the features against the range its half-precision operands hold (model.hip: op16_feat_limit) -------------------
__global__ __launch_bounds__(256) void feature_range_kernel(const float* __restrict__ x, size_t n4, size_t n, const float* __restrict__ limit,
                                                            unsigned int* fault) {
    const float lim = *limit;
    if (!(lim > 0.f)) return;
    float m = 0.f;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    // what the 16-byte loads did not cover: the last n % 4 values - or everything, for a batch that does not start on a 16-byte
    // boundary (a view into a larger tensor with an odd feature width)
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    if (m > lim) *fault = 1u;  // (racing writers store the same word; page-locked host memory, read after the stream has drained)
}

int launch_feature_range(const float* x, size_t n, const float* limit, unsigned int* fault, hipStream_t s) {
    if (n == 0) return 0;
    const size_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) ? 0 : n / 4;
    const int grid = (int)std::min<size_t>(2048, ((n4 ? n4 : n) + 255) / 256 + 1);
    hipLaunchKernelGGL(feature_range_kernel, dim3(grid), dim3(256), 0, s, x, n4, n, limit, fault);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
