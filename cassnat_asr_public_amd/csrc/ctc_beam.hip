// CTC prefix beam search and CTC forced (Viterbi) alignment for gfx950: the device half of `decode_type: ctc_only / ctc_att`.
//
// Reference: ctc_beam_decode (src/utils/beam_decode.py:8-93, called from src/tasks/cassnat_task.py:335-341) and
// CassNAT.beam_path_align -> viterbi_align (src/models/cassnat.py:391-414, 272-353).
//
// ctc_beam_decode is a per-frame loop over a Python list of hypotheses: every kept hypothesis yields one "stay" candidate
// (blank or a repetition of its last label) and one candidate per pruned label of the frame; the candidates are NOT merged
// by prefix; a stable descending sort by score_ctc + score_lm + ctc_lp * len(hyp) keeps `ctc_beam` of them.  Frames past
// src_size and frames whose blank probability exceeds 0.95 are skipped.  All scores are Python floats (float64) made of
// float32 log-posteriors.  Here: one workgroup per utterance, the frame loop inside the kernel, candidates one per thread
// in float64 with numpy's logaddexp formula, the stable top-W by counting rank (key descending, list position ascending),
// hypotheses kept as (parent, label) back-pointers per processed frame and unrolled at the end.  Integer / ordering work:
// the hypotheses are those of the reference wherever two candidate scores are not within an ulp of each other.
//
// viterbi_align is the max-product CTC forward pass over the blank-augmented label sequence in float32 followed by a
// sequential back-trace; here one workgroup per utterance with the two live alpha rows in LDS and the back-pointers in
// global memory; same float32 operation order, first-maximum-wins among the three predecessors (torch.max).
#include "kernels.h"

#define CB_LOGZERO (-1e10)

__device__ __forceinline__ double cb_logaddexp(double x, double y) {  // numpy's npy_logaddexp
    if (x == y) return x + 0.693147180559945309417232121458176568;
    const double tmp = x - y;
    if (tmp > 0) return x + log1p(exp(-tmp));
    if (tmp <= 0) return y + log1p(exp(tmp));
    return tmp;
}

__global__ __launch_bounds__(256) void ctc_prefix_beam_kernel(CtcBeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int W = a.W, P = a.P, NC = W * (P + 1);
    // beam state
    double* pb = reinterpret_cast<double*>(smem);  // [W]
    double* pnb = pb + W;                          // [W]
    double* ckey = pnb + W;                        // [NC] candidates
    double* cpb = ckey + NC;
    double* cpnb = cpb + NC;
    double* ctot = cpnb + NC;
    double* ptot = ctot + NC;                      // [W] score_ctc of the kept hypotheses
    int* blen = reinterpret_cast<int*>(ptot + W);  // [W]
    int* blast = blen + W;                         // [W] last label (-1: empty hypothesis)
    int* cpar = blast + W;                         // [NC] parent slot (-1: not a candidate)
    int* ctok = cpar + NC;                         // [NC] appended label (-1: stay)
    int* clen = ctok + NC;
    int* clast = clen + NC;
    __shared__ int s_nb, s_steps;

    const float* logp = a.logp + (long long)b * a.Tp * a.V;
    const int* top = a.top_idx + (long long)b * a.Tp * P;
    unsigned char* hpar = a.hist_parent + (long long)b * a.Tp * W;
    int* htok = a.hist_tok + (long long)b * a.Tp * W;
    // src_size = (ratio * T').long(): fp32 product, truncation (beam_decode.py:22)
    const int ssz = (int)(long long)(a.size_ratio[b] * (float)a.Tp);
    if (tid == 0) {
        pb[0] = 0.0;  // logone
        pnb[0] = CB_LOGZERO;
        ptot[0] = 0.0;
        blen[0] = 0;
        blast[0] = -1;
        s_nb = 1;
        s_steps = 0;
    }
    __syncthreads();
    for (int t = 0; t < a.Tp; ++t) {
        if (t > ssz) continue;  // (the reference does process frame t == src_size)
        const float* row = logp + (long long)t * a.V;
        const float pblank = row[a.blank];
        // torch.exp of a float32, compared with the double 0.95
        if ((double)(float)exp((double)pblank) > 0.95) continue;
        const int nb = s_nb, step = s_steps;
        for (int i = tid; i < NC; i += 256) {
            const int k = i / (P + 1), j = i - k * (P + 1);
            int par = -1, tok = -1, nlen = 0, nlast = -1;
            double npb = CB_LOGZERO, npnb = CB_LOGZERO, tot = CB_LOGZERO;
            if (k < nb) {
                const double p_b = pb[k], p_nb = pnb[k];
                const int last = blast[k], len = blen[k];
                if (j == 0) {  // blank or repetition
                    npnb = len > 0 ? p_nb + (double)row[last] : CB_LOGZERO;
                    const double pt = (double)pblank;
                    npb = cb_logaddexp(p_b + pt, p_nb + pt);
                    tot = cb_logaddexp(npb, npnb);
                    par = k;
                    nlen = len;
                    nlast = last;
                } else {
                    const int c = top[(long long)t * P + (j - 1)];
                    if (c != a.blank) {
                        const double pt = (double)row[c];
                        npnb = (c != last) ? cb_logaddexp(p_b + pt, p_nb + pt) : p_b + pt;
                        npb = CB_LOGZERO;
                        tot = cb_logaddexp(npb, npnb);
                        par = k;
                        tok = c;
                        nlen = len + 1;
                        nlast = c;
                    }
                }
            }
            cpar[i] = par;
            ctok[i] = tok;
            clen[i] = nlen;
            clast[i] = nlast;
            cpb[i] = npb;
            cpnb[i] = npnb;
            ctot[i] = tot;
            ckey[i] = (tot + 0.0) + a.lp * (double)nlen;  // score_ctc + score_lm + ctc_lp * len(hyp)
        }
        __syncthreads();
        // stable descending order: rank = candidates that sort before this one
        const int ncand = nb * (P + 1);
        for (int i = tid; i < ncand; i += 256) {
            if (cpar[i] < 0) continue;
            const double key = ckey[i];
            int rank = 0;
            for (int j = 0; j < ncand; ++j) {
                if (cpar[j] < 0) continue;
                const double kj = ckey[j];
                rank += (kj > key || (kj == key && j < i)) ? 1 : 0;
            }
            if (rank < W) {
                pb[rank] = cpb[i];
                pnb[rank] = cpnb[i];
                ptot[rank] = ctot[i];
                blen[rank] = clen[i];
                blast[rank] = clast[i];
                hpar[(long long)step * W + rank] = (unsigned char)cpar[i];
                htok[(long long)step * W + rank] = ctok[i];
            }
        }
        // (pb / pnb / blen / blast were only read in the candidate phase above, behind the barrier: safe to overwrite)
        if (tid == 0) {
            int valid = 0;
            for (int j = 0; j < ncand; ++j) valid += cpar[j] >= 0 ? 1 : 0;
            s_nb = valid < W ? valid : W;
            s_steps = step + 1;
        }
        __syncthreads();
    }
    // unroll the back-pointers: one thread per kept hypothesis
    const int nb = s_nb, steps = s_steps;
    if (tid == 0) a.n_out[b] = nb;
    if (tid < W) {
        int* h = a.hyp + ((long long)b * W + tid) * a.Lmax;
        if (tid < nb) {
            const int len = blen[tid];
            int pos = len - 1, cur = tid;
            for (int st = steps - 1; st >= 0; --st) {
                const int tok = htok[(long long)st * W + cur];
                if (tok >= 0) {
                    if (pos >= 0 && pos < a.Lmax) h[pos] = tok;
                    --pos;
                }
                cur = hpar[(long long)st * W + cur];
            }
            for (int i = len; i < a.Lmax; ++i) h[i] = 0;
            a.hyp_len[b * W + tid] = len;
            a.score[b * W + tid] = ptot[tid];
            a.p_blk[b * W + tid] = pb[tid];
            a.p_nblk[b * W + tid] = pnb[tid];
        } else {
            for (int i = 0; i < a.Lmax; ++i) h[i] = 0;
            a.hyp_len[b * W + tid] = 0;
            a.score[b * W + tid] = CB_LOGZERO;
            a.p_blk[b * W + tid] = CB_LOGZERO;
            a.p_nblk[b * W + tid] = CB_LOGZERO;
        }
    }
}

int launch_ctc_prefix_beam(const CtcBeamArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (a.W < 1 || a.W > 32 || a.P < 0 || a.P > 32 || a.Lmax < 1) {
        cn_set_error("ctc_beam: need 1 <= ctc_beam <= 32 and 0 <= ctc_pruning <= 32");
        return -1;
    }
    const int NC = a.W * (a.P + 1);
    const size_t lds = (size_t)(3 * a.W + 4 * NC) * 8 + (size_t)(2 * a.W + 4 * NC) * 4 + 64;
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)ctc_prefix_beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL(ctc_prefix_beam_kernel, dim3(a.B), dim3(256), lds, s, a);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- forced alignment ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ctc_viterbi_kernel(ViterbiArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int S = 2 * a.ymax + 1, Tp = a.Tp;
    float* al0 = reinterpret_cast<float*>(smem);  // alpha[t]
    float* al1 = al0 + S;                         // alpha[t + 1]
    float* alx = al1 + S;                         // alpha[src_size]
    int* path = reinterpret_cast<int*>(alx + S);  // [S] blank-augmented labels
    const float* logp = a.logp + (long long)b * Tp * a.V;
    const unsigned char* km = a.keymask + (long long)b * Tp;
    unsigned char* bp = a.bp + (long long)b * Tp * S;  // state index minus predecessor index: 0, 1 or 2
    int* out = a.out_path + (long long)b * Tp;
    const int ylen = a.label_len[b];
    const int plen = 2 * ylen + 1;
    const int xb = (int)(long long)(a.size_ratio[b] * (float)Tp);
    const float LZ = -1e10f;
    for (int s2 = tid; s2 < S; s2 += 256) {
        const int u = s2 >> 1;
        path[s2] = (s2 & 1) ? (u < a.ymax ? a.labels[(long long)b * a.ld + u] : a.blank) : a.blank;
        if (u >= ylen && (s2 & 1)) path[s2] = a.blank;  // (the reference pads ys with zeros = blank)
        al0[s2] = s2 == 0 ? 0.f : LZ;
    }
    __syncthreads();
    if (xb == 0)
        for (int s2 = tid; s2 < S; s2 += 256) alx[s2] = al0[s2];
    for (int t = 0; t < Tp; ++t) {
        const bool ok = km[t] != 0;
        for (int s2 = tid; s2 < S; s2 += 256) {
            const float m0 = al0[s2];
            const float m1 = s2 >= 1 ? al0[s2 - 1] : LZ;
            float m2 = s2 >= 2 ? al0[s2 - 2] : LZ;
            if (s2 >= 2 && path[s2 - 2] == path[s2]) m2 = LZ;  // blank and repeated labels have two predecessors only
            float mx = m0;
            int ix = 0;
            if (m1 > mx) {
                mx = m1;
                ix = 1;
            }
            if (m2 > mx) {
                mx = m2;
                ix = 2;
            }
            if (s2 >= plen) mx = LZ;
            bp[(long long)t * S + s2] = (unsigned char)ix;
            const float lp = ok ? logp[(long long)t * a.V + path[s2]] : LZ;
            al1[s2] = mx + lp;
        }
        __syncthreads();
        for (int s2 = tid; s2 < S; s2 += 256) {
            al0[s2] = al1[s2];
            if (t + 1 == xb) alx[s2] = al1[s2];
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int t = 0; t < Tp; ++t) out[t] = a.blank;
        if (xb >= 1 && xb <= Tp) {
            const int i1 = plen - 1, i2 = plen - 2 >= 0 ? plen - 2 : S - 1;  // (python index -1 wraps to the last state)
            int cur = alx[i1] > alx[i2] ? i1 : i2;
            out[xb - 1] = path[cur];
            for (int t = xb - 1; t >= 1; --t) {
                cur = cur - (int)bp[(long long)t * S + cur];
                if (cur < 0) cur += S;
                out[t - 1] = path[cur];
            }
        }
    }
}

int launch_ctc_viterbi(const ViterbiArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    const int S = 2 * a.ymax + 1;
    const size_t lds = (size_t)S * 16 + 64;
    if (a.ymax < 1 || lds > 96 * 1024) {
        cn_set_error("ctc_viterbi: label length out of range");
        return -1;
    }
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)ctc_viterbi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL(ctc_viterbi_kernel, dim3(a.B), dim3(256), lds, s, a);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
