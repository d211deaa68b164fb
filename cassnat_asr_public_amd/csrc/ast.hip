// Kernels of the autoregressive (AST) decoder step with a KV cache - BASELINE config 4, SURVEY 8a row a18.
// Reference: Transformer.beam_decode (src/models/transformer.py:122-241) re-runs the whole decoder on the full prefix
// at every step; here only the NEW position of every live hypothesis is computed and the keys/values of the earlier
// positions are read from a cache.
//
// Cache layout: K and V of decoder layer l, position j, written by the hypothesis that occupied row ("slot") s of the live
// list at step j:  cache[l][j][s][d].  Every (j, s) cell is written exactly once (at step j), so a hypothesis never copies
// its parent's cache: it carries an ancestor table anc[j] = slot that wrote position j of its prefix, and the attention
// kernel gathers through it.  HBM-bound gather work - no MFMA (one query row per hypothesis).
#include "kernels.h"

// x[h][:] = lut[tok[h]][:] * sqrt(d) + pe[pos][:]        (TextEmbedding + PositionalEncoding, embedding.py:71-78, 29-31)
__global__ void ast_embed_kernel(const int* __restrict__ tok, const float* __restrict__ lut, const float* __restrict__ pe_row,
                                 float* __restrict__ x, int n, int d, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    const int h = i / d, c = i - h * d;
    x[i] = lut[(long long)tok[h] * d + c] * scale + pe_row[c];
}

int launch_ast_embed(const int* tok, const float* lut, const float* pe_row, float* x, int n, int d, float scale,
                     hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ast_embed_kernel, dim3(cn_ceil_div(n * d, 256)), dim3(256), 0, s, tok, lut, pe_row, x, n, d, scale);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// append the new position's K|V (columns d..3d of the fused projection) to the cache cell (pos, slot = row)
template <typename T>
__global__ void ast_kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ ck, T* __restrict__ cv, int n, int d,
                                     int slots, int pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // 16-byte chunks
    const int per = d * (int)sizeof(T) / 16;
    if (i >= n * per) return;
    const int h = i / per, c = i - h * per;
    const uint4* src = reinterpret_cast<const uint4*>(qkv + (long long)h * 3 * d);
    const long long cell = ((long long)pos * slots + h) * per + c;
    reinterpret_cast<uint4*>(ck)[cell] = src[per + c];
    reinterpret_cast<uint4*>(cv)[cell] = src[2 * per + c];
}

int launch_ast_kv_append(int prec, const void* qkv, void* ck, void* cv, int n, int d, int slots, int pos, hipStream_t s) {
    if (n <= 0) return 0;
    const int per = d * (int)cn_elem_size(prec) / 16;
    const dim3 grid(cn_ceil_div(n * per, 256));
    if (prec == CN_PREC_F32 || prec == CN_PREC_X3)  // (split-bf16: d columns are d * 4 contiguous bytes of a row, whole 128-byte groups)
        hipLaunchKernelGGL(ast_kv_append_kernel<float>, grid, dim3(256), 0, s, (const float*)qkv, (float*)ck, (float*)cv, n, d, slots, pos);
    else
        hipLaunchKernelGGL(ast_kv_append_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ck, (bf16*)cv, n, d, slots, pos);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Single-query multi-head attention with gathered key/value rows (d_k = 64).  One workgroup per hypothesis, one wave per
// head.  Phase 1: a lane owns a key (64-dim dot product with the query held in registers); masked keys get float-min
// exactly like MultiHeadedAttention (attention.py:19-21).  Phase 2: a lane owns an output dimension.
//   MODE 0 (self, cache):  key j in [0, nkeys): row (j*slots + anc[h][j]) of ck / cv (row length d), allowed iff keyok[h][j]
//   MODE 1 (source):       key j in [0, nkeys): row (utt[h]*nkeys + j) of the fused K|V matrix (row length 2d), allowed iff
//                          keymask[utt[h]][j]
// ---------------------------------------------------------------------------------------------
struct GatherAttnParams {
    const void* q;  // [n][ldq], head hd at column hd*64
    int ldq;
    const void* k;
    const void* v;
    void* o;  // [n][ldo]
    int ldo;
    int n, H, nkeys, slots, d, table_stride;
    const int* anc;              // MODE 0: [n][table_stride]
    const unsigned char* keyok;  // MODE 0: [n][table_stride]
    const int* utt;              // MODE 1: [n]
    const unsigned char* keymask;  // MODE 1: [B][nkeys]
    float scale;
    // MODE 0, append_pos >= 0: this step's K and V of hypothesis h are still in the projection buffer (columns d.. and 2d.. of
    // q's row): the workgroup writes them to cache row (append_pos, slot h) for the steps to come and reads key append_pos
    // from the projection buffer itself (no separate append launch, no read-after-write through the cache)
    int append_pos;
};

// element c of a row that starts at byte pointer `row` (c counted from the row's first column; for split-bf16 rows the start
// must be a multiple of 32 columns): fp32 / bf16 elements are contiguous, a split-bf16 element is a bf16 hi half at
// cn_split_off(c) and its lo half 64 bytes further
template <typename T> __device__ __forceinline__ float ga_load(const unsigned char* row, int c) {
    if constexpr (__is_same(T, split_t)) {
        const unsigned char* e = row + cn_split_off((size_t)c);
        return (float)*reinterpret_cast<const bf16*>(e) + (float)*reinterpret_cast<const bf16*>(e + 64);
    } else {
        return to_f32(reinterpret_cast<const T*>(row)[c]);
    }
}
template <typename T> __device__ __forceinline__ void ga_store(unsigned char* row, int c, float v) {
    if constexpr (__is_same(T, split_t)) {
        unsigned char* e = row + cn_split_off((size_t)c);
        const bf16 hi = (bf16)v;
        *reinterpret_cast<bf16*>(e) = hi;
        *reinterpret_cast<bf16*>(e + 64) = (bf16)(v - (float)hi);
    } else {
        reinterpret_cast<T*>(row)[c] = from_f32<T>(v);
    }
}

template <typename T, int MODE>
__global__ void ast_gather_attn_kernel(GatherAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int h = blockIdx.x, lane = threadIdx.x & 63, hd = threadIdx.x >> 6;
    float* sc = reinterpret_cast<float*>(smem) + (long long)hd * p.nkeys;
    constexpr int ES = __is_same(T, split_t) ? 4 : (int)sizeof(T);  // bytes per element (a split-bf16 pair is 4)
    // rows as byte pointers: head hd of a row is 64 consecutive columns = 64 * ES contiguous bytes in every layout
    const unsigned char* qrow = reinterpret_cast<const unsigned char*>(p.q) + ((long long)h * p.ldq + hd * 64) * ES;
    float q[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) q[i] = ga_load<T>(qrow, i);
    const int u = MODE == 1 ? p.utt[h] : 0;
    auto row_of = [&](int j) -> long long {
        if (MODE == 0) return (long long)j * p.slots + p.anc[(long long)h * p.table_stride + j];
        return (long long)u * p.nkeys + j;
    };
    const int kstride = MODE == 0 ? p.d : 2 * p.d;
    const bool append = MODE == 0 && p.append_pos >= 0;
    const unsigned char* new_k = qrow + (long long)p.d * ES;      // (head hd of this row's K | V in the fused projection)
    const unsigned char* new_v = qrow + (long long)2 * p.d * ES;
    const unsigned char* kb = reinterpret_cast<const unsigned char*>(p.k);
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(p.v);
    if (append) {  // the head's 64 * ES bytes of K and of V, ES bytes per lane (a byte copy: any element layout)
        const long long crow = (((long long)p.append_pos * p.slots + h) * p.d + hd * 64) * ES + lane * ES;
        if constexpr (ES == 4) {
            *reinterpret_cast<unsigned*>(const_cast<unsigned char*>(kb) + crow) = *reinterpret_cast<const unsigned*>(new_k + lane * 4);
            *reinterpret_cast<unsigned*>(const_cast<unsigned char*>(vb) + crow) = *reinterpret_cast<const unsigned*>(new_v + lane * 4);
        } else {
            *reinterpret_cast<unsigned short*>(const_cast<unsigned char*>(kb) + crow) = *reinterpret_cast<const unsigned short*>(new_k + lane * 2);
            *reinterpret_cast<unsigned short*>(const_cast<unsigned char*>(vb) + crow) = *reinterpret_cast<const unsigned short*>(new_v + lane * 2);
        }
    }
    float lmax = -INFINITY;
    for (int j = lane; j < p.nkeys; j += 64) {
        const unsigned char* kr = (append && j == p.append_pos) ? new_k : kb + (row_of(j) * kstride + hd * 64) * ES;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) dot = fmaf(q[i], ga_load<T>(kr, i), dot);
        const bool ok = MODE == 0 ? p.keyok[(long long)h * p.table_stride + j] != 0 : p.keymask[(long long)u * p.nkeys + j] != 0;
        const float s = ok ? dot * p.scale : CN_NEG_FILL;
        sc[j] = s;
        lmax = fmaxf(lmax, s);
    }
    lmax = wave_max(lmax);
    float lsum = 0.f;
    for (int j = lane; j < p.nkeys; j += 64) {
        const float e = expf(sc[j] - lmax);
        sc[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __builtin_amdgcn_wave_barrier();
    // phase 2: lane = output dimension; LDS writes above are visible to the whole wave (same wave, program order)
    const float inv = 1.f / lsum;
    float acc = 0.f;
    for (int j = 0; j < p.nkeys; ++j) {
        const unsigned char* vr = (append && j == p.append_pos) ? new_v : vb + (row_of(j) * kstride + hd * 64) * ES;
        acc = fmaf(sc[j], ga_load<T>(vr, lane), acc);
    }
    ga_store<T>(reinterpret_cast<unsigned char*>(p.o) + ((long long)h * p.ldo + hd * 64) * ES, lane, acc * inv);
}

int launch_ast_gather_attn(int prec, int mode, const GatherAttnArgs& a, hipStream_t s) {
    if (a.n <= 0) return 0;
    if (a.H < 1 || a.H > 16 || a.nkeys < 1) {
        cn_set_error("ast_gather_attn: need 1 <= heads <= 16 and at least one key");
        return -1;
    }
    GatherAttnParams p;
    p.q = a.q;
    p.ldq = a.ldq;
    p.k = a.k;
    p.v = a.v;
    p.o = a.o;
    p.ldo = a.ldo;
    p.n = a.n;
    p.H = a.H;
    p.nkeys = a.nkeys;
    p.slots = a.slots;
    p.d = a.d;
    p.table_stride = a.table_stride;
    p.anc = a.anc;
    p.keyok = a.keyok;
    p.utt = a.utt;
    p.keymask = a.keymask;
    p.scale = a.scale;
    p.append_pos = mode == 0 ? a.append_pos : -1;
    const size_t lds = (size_t)a.H * a.nkeys * sizeof(float);
    if (lds > 64 * 1024) {
        cn_set_error("ast_gather_attn: too many keys for the score buffer");
        return -1;
    }
    const dim3 grid(a.n), block(64 * a.H);
    if (prec == CN_PREC_F32) {
        if (mode == 0) hipLaunchKernelGGL((ast_gather_attn_kernel<float, 0>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ast_gather_attn_kernel<float, 1>), grid, block, lds, s, p);
    } else if (prec == CN_PREC_X3) {
        if (a.ldq % 32 || a.ldo % 32 || a.d % 32) {
            cn_set_error("ast_gather_attn: split-bf16 rows need strides that are multiples of 32 elements");
            return -1;
        }
        if (mode == 0) hipLaunchKernelGGL((ast_gather_attn_kernel<split_t, 0>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ast_gather_attn_kernel<split_t, 1>), grid, block, lds, s, p);
    } else {
        if (mode == 0) hipLaunchKernelGGL((ast_gather_attn_kernel<bf16, 0>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ast_gather_attn_kernel<bf16, 1>), grid, block, lds, s, p);
    }
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// CTC side of joint decoding (src/models/transformer.py:135-141, src/utils/ctc_prefix.py).
// ---------------------------------------------------------------------------------------------
#define CN_LOGZERO (-1e10f)

// ctc_out.masked_fill_(src_mask^T == 0, logzero); ctc_out[:, :, blank].masked_fill_(src_mask == 0, 0)
__global__ void ast_ctc_mask_kernel(float* __restrict__ logp, const unsigned char* __restrict__ keymask, int rows, int V,
                                    int blank) {
    const int r = blockIdx.x;
    if (r >= rows || keymask[r]) return;
    float* p = logp + (long long)r * V;
    for (int i = threadIdx.x; i < V; i += blockDim.x) p[i] = i == blank ? 0.f : CN_LOGZERO;
}

// CTCPrefixScore.initial_state: r0[b][t] = (logzero, cumsum_t x[b][t][blank])   (sequential fp32 sum, as torch.cumsum)
__global__ void ast_ctc_init_kernel(const float* __restrict__ logp, float* __restrict__ r0, int B, int Tp, int V, int blank) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float run = 0.f;
    for (int t = 0; t < Tp; ++t) {
        run += logp[((long long)b * Tp + t) * V + blank];
        r0[((long long)b * Tp + t) * 2 + 0] = CN_LOGZERO;
        r0[((long long)b * Tp + t) * 2 + 1] = run;
    }
}

int launch_ast_ctc_prepare(float* logp, const unsigned char* keymask, float* r0, int B, int Tp, int V, int blank,
                           hipStream_t s) {
    hipLaunchKernelGGL(ast_ctc_mask_kernel, dim3(B * Tp), dim3(256), 0, s, logp, keymask, B * Tp, V, blank);
    hipLaunchKernelGGL(ast_ctc_init_kernel, dim3(cn_ceil_div(B, 64)), dim3(64), 0, s, logp, r0, B, Tp, V, blank);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

__device__ __forceinline__ float lse2(float a, float b) {  // torch.logsumexp over two values
    const float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}

// CTCPrefixScore.__call__ (ctc_prefix.py:50-106): one thread per (hypothesis, candidate label), sequential over frames.
__global__ void ast_ctc_prefix_kernel(CtcPrefixArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n * a.K) return;
    const int h = i / a.K;
    const int c = a.cand[i];
    const int u = a.utt[h];
    const int Tp = a.Tp;
    const float* x = a.logp + (long long)u * Tp * a.V;
    const int ref = a.prev_ref[h];
    const float* rp = ref < 0 ? a.r0 + (long long)(-1 - ref) * Tp * 2 : a.r_prev + (long long)ref * Tp * 2;
    float* r = a.r_new + (long long)i * Tp * 2;
    const bool same = a.last_tok[h] == c;
    const int L = a.out_len;
    const int start = L > 1 ? L : 1;
    for (int t = 0; t < start; ++t) {
        r[2 * t] = CN_LOGZERO;
        r[2 * t + 1] = CN_LOGZERO;
    }
    if (L == 0) r[0] = x[c];
    // log_phi(t) = last(g) == c ? r_prev^b(t) : logsumexp(r_prev^n(t), r_prev^b(t))
    float rn = r[2 * (start - 1)], rb = r[2 * (start - 1) + 1];
    // log_psi = logsumexp over { r^n(start-1) } U { log_phi(t-1) + x_c(t), t in [start, T) }: streaming max/sum
    float pm = rn, ps = 1.f;
    // The recurrence is sequential in t, its inputs are not: the loads of 8 frames go out together (the one-frame-at-a-time
    // loop spent 0.7 us per frame waiting for them), then the 8 updates run in the reference's order.
    for (int t0 = start; t0 < Tp; t0 += 8) {
        float2 pv[8];
        float xcv[8], xbv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j < Tp ? t0 + j : Tp - 1;
            pv[j] = *reinterpret_cast<const float2*>(rp + 2 * (t - 1));
            xcv[j] = x[(long long)t * a.V + c];
            xbv[j] = x[(long long)t * a.V + a.blank];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            if (t < Tp) {
                const float p0 = pv[j].x, p1 = pv[j].y;
                const float phi = same ? p1 : lse2(p0, p1);
                const float xc = xcv[j], xb = xbv[j];
                const float nn = lse2(rn, phi) + xc;
                const float nb = lse2(rn, rb) + xb;
                *reinterpret_cast<float2*>(r + 2 * t) = make_float2(nn, nb);
                rn = nn;
                rb = nb;
                const float v = phi + xc;
                if (v > pm) {
                    ps = ps * expf(pm - v) + 1.f;
                    pm = v;
                } else {
                    ps += expf(v - pm);
                }
            }
        }
    }
    float psi = pm + logf(ps);
    if (c == a.eos) psi = lse2(rp[2 * (Tp - 1)], rp[2 * (Tp - 1) + 1]);
    if (c == a.blank) psi = CN_LOGZERO;
    a.score[i] = psi;
}

int launch_ast_ctc_prefix(const CtcPrefixArgs& a, hipStream_t s) {
    if (a.n <= 0 || a.K <= 0) return 0;
    hipLaunchKernelGGL(ast_ctc_prefix_kernel, dim3(cn_ceil_div(a.n * a.K, 64)), dim3(64), 0, s, a);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side beam bookkeeping of Transformer.beam_decode (src/models/transformer.py:157-240).  Hypothesis slot s =
// b * bw + j is also its KV-cache slot; every slot runs every step (finished / unused ones compute on dummies and are
// ignored), so nothing is compacted and nothing returns to the host inside the loop.  One workgroup per utterance:
// <= bw finished hypotheses are carried, every live hypothesis contributes its top-bw of K candidates (stable order,
// like the host path's stable argsort), all candidates are ranked by score + (len - 1) * length_penalty with ties in
// list order (Python's stable sort), the best bw become the next beam.  Arithmetic follows the reference's types:
// float32 for the per-candidate combination (no FMA contraction), Python-float (double) score accumulation and keys.
// ---------------------------------------------------------------------------------------------------------------
__global__ void ast_beam_init_kernel(AstBeamState st, int cur, int B, int bw, int L, int sos, int pad) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * bw) return;
    const int b = s / bw, j = s - b * bw;
    for (int t = 0; t < L; ++t) {
        st.tok[cur][(long long)s * L + t] = t == 0 ? sos : pad;
        st.anc[cur][(long long)s * L + t] = s;
        st.keyok[cur][(long long)s * L + t] = (t == 0 && sos != pad) ? 1 : 0;
    }
    st.len[cur][s] = 1;
    st.score[cur][s] = 0.0;
    st.valid[cur][s] = j == 0;
    st.ctc_ref[cur][s] = -1 - b;
    st.ctc_prev[cur][s] = 0.f;
    st.cur_tok[s] = sos;
    st.utt[s] = b;
    if (s == 0) *st.live = B;
}

constexpr int BEAM_MAXW = 16, BEAM_MAXC = BEAM_MAXW + BEAM_MAXW * BEAM_MAXW;

__global__ __launch_bounds__(128) void ast_beam_update_kernel(AstBeamState st, AstBeamStep q) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int bw = q.bw, K = q.K, L = q.L, cur = q.cur, nxt = cur ^ 1;
    __shared__ double ckey[BEAM_MAXC], cscore[BEAM_MAXC];
    __shared__ int cpar[BEAM_MAXC], ccand[BEAM_MAXC], ctok[BEAM_MAXC], newslot[BEAM_MAXW];
    __shared__ float cctc[BEAM_MAXC], loc[BEAM_MAXW][BEAM_MAXW];
    __shared__ int fin_idx[BEAM_MAXW], live_idx[BEAM_MAXW], nf_s, nl_s;
    if (tid == 0) {
        int nf = 0, nl = 0;
        for (int j = 0; j < bw; ++j) {
            const int s = b * bw + j;
            if (!st.valid[cur][s]) continue;
            const int last = st.tok[cur][(long long)s * L + st.len[cur][s] - 1];
            if (last == q.eos) fin_idx[nf++] = j; else live_idx[nl++] = j;
        }
        nf_s = nf;
        nl_s = nl;
    }
    if (tid < BEAM_MAXW) newslot[tid] = -1;
    __syncthreads();
    const int nf = nf_s, nl = nl_s;
    if (tid < nf) {  // finished hypotheses are carried as they are
        const int s = b * bw + fin_idx[tid];
        const double sc = st.score[cur][s];
        cscore[tid] = sc;
        ckey[tid] = q.use_lp ? sc + (double)(st.len[cur][s] - 1) * q.lp : sc;
        cpar[tid] = fin_idx[tid];
        ccand[tid] = -1;
    }
    for (int i = tid; i < nl * K; i += 128) {
        const int li = i / K, c = i - li * K;
        const int s = b * bw + live_idx[li];
        const float att = q.att[(long long)s * K + c];
        // local = ctc_weight * (ctc - prev) + (1 - ctc_weight) * att, float32, one rounding per operation (transformer.py:205-206)
        loc[li][c] = q.use_ctc ? __fadd_rn(__fmul_rn(q.w, __fsub_rn(q.ctc[(long long)s * K + c], st.ctc_prev[cur][s])), __fmul_rn(q.u, att))
                               : att;
    }
    __syncthreads();
    for (int i = tid; i < nl * K; i += 128) {
        const int li = i / K, c = i - li * K;
        const float v = loc[li][c];
        int r = 0;
        for (int c2 = 0; c2 < K; ++c2) r += (loc[li][c2] > v) || (loc[li][c2] == v && c2 < c);
        if (r < bw) {
            const int j = live_idx[li], s = b * bw + j, e = nf + li * bw + r;
            const double sc = st.score[cur][s] + (double)v;
            cscore[e] = sc;
            ckey[e] = q.use_lp ? sc + (double)st.len[cur][s] * q.lp : sc;  // new length - 1 = old length
            cpar[e] = j;
            ccand[e] = c;
            ctok[e] = q.idx[(long long)s * K + c];
            cctc[e] = q.use_ctc ? q.ctc[(long long)s * K + c] : 0.f;
        }
    }
    __syncthreads();
    const int ncand = nf + nl * (K < bw ? K : bw);
    for (int e = tid; e < ncand; e += 128) {
        const double k = ckey[e];
        int r = 0;
        for (int e2 = 0; e2 < ncand; ++e2) r += (ckey[e2] > k) || (ckey[e2] == k && e2 < e);
        if (r < bw) newslot[r] = e;
    }
    __syncthreads();
    int live_new = 0;
    for (int qn = 0; qn < bw; ++qn) {
        const int e = newslot[qn];
        const int sn = b * bw + qn;
        if (e < 0) {  // fewer candidates than beam slots: an unused slot with harmless inputs for the next step
            for (int t = tid; t < L; t += 128) st.anc[nxt][(long long)sn * L + t] = sn;
            if (tid == 0) {
                st.valid[nxt][sn] = 0;
                st.len[nxt][sn] = 1;
                st.score[nxt][sn] = 0.0;
                st.ctc_ref[nxt][sn] = -1 - b;
                st.ctc_prev[nxt][sn] = 0.f;
                st.cur_tok[sn] = q.sos;
            }
            continue;
        }
        const int so = b * bw + cpar[e];
        const int len_o = st.len[cur][so];
        const bool grown = ccand[e] >= 0;
        for (int t = tid; t < L; t += 128) {
            int tk = st.tok[cur][(long long)so * L + t], an = st.anc[cur][(long long)so * L + t];
            unsigned char ko = st.keyok[cur][(long long)so * L + t];
            if (grown) {
                if (t == len_o) {
                    tk = ctok[e];
                    ko = ctok[e] != q.pad;
                }
                if (t == q.pos) an = so;       // position pos was computed in the parent's slot this step
                if (t == q.pos + 1) an = sn;   // the next position will be computed in this slot
            }
            st.tok[nxt][(long long)sn * L + t] = tk;
            st.anc[nxt][(long long)sn * L + t] = an;
            st.keyok[nxt][(long long)sn * L + t] = ko;
        }
        if (tid == 0) {
            st.valid[nxt][sn] = 1;
            st.len[nxt][sn] = len_o + (grown ? 1 : 0);
            st.score[nxt][sn] = cscore[e];
            st.ctc_ref[nxt][sn] = grown ? so * K + ccand[e] : st.ctc_ref[cur][so];
            st.ctc_prev[nxt][sn] = grown ? cctc[e] : st.ctc_prev[cur][so];
            st.cur_tok[sn] = grown ? ctok[e] : q.eos;
        }
        live_new += grown && ctok[e] != q.eos;
    }
    if (tid == 0 && live_new) atomicAdd(st.live, live_new);
}

int launch_ast_beam_init(const AstBeamState& st, int cur, int B, int bw, int L, int sos, int pad, hipStream_t s) {
    hipLaunchKernelGGL(ast_beam_init_kernel, dim3(cn_ceil_div(B * bw, 64)), dim3(64), 0, s, st, cur, B, bw, L, sos, pad);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_ast_beam_update(const AstBeamState& st, const AstBeamStep& q, int B, hipStream_t s) {
    if (q.bw < 1 || q.bw > BEAM_MAXW || q.K < 1 || q.K > BEAM_MAXW) {
        cn_set_error("ast beam: beam_width and candidate count must be in 1..16");
        return -1;
    }
    CN_HIP_CHECK(hipMemsetAsync(st.live, 0, sizeof(int), s));
    hipLaunchKernelGGL(ast_beam_update_kernel, dim3(B), dim3(128), 0, s, st, q);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
