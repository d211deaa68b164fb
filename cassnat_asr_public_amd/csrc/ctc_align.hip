// CTC greedy alignment -> trigger intervals, and the greedy hypothesis pack.  Integer work, bit-exact.
//
// Reference: CassNAT.best_path_align (src/models/cassnat.py:378-389), align_to_mask (:355-365),
// expand_trigger_mask (:259-270), "& src_mask" (:468).  The reference materialises a dense
// (B, ymax+1, T') boolean mask; here each trigger row is kept as at most two frame intervals
// (s1,e1,s2,e2) - the run of frames whose running token count equals u, and the forced
// "src_size-1" frame of the EOS row - and the attention kernel ANDs them with the key mask.
//
// Exactness of the interval form: the dense row is {t : c[t]==u} & keymask, then (optionally) dilated
// by one frame left/right, then & keymask again.  c is non-decreasing, so {t : c[t]==u} is a run;
// taking the TIGHT bounds of (run & keymask) before dilating and re-masking afterwards gives the same
// set for any mask (holes inside the run are removed by the final mask either way).
#include "kernels.h"

__global__ __launch_bounds__(256) void ctc_align_kernel(AlignArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_scan = reinterpret_cast<int*>(smem);  // [256]
    int* s_carry = s_scan + 256;                 // [4] (padded)
    int* s_lo = s_carry + 4;                     // [Tp+1] first valid frame of row u
    int* s_hi = s_lo + (a.Tp + 1);               // [Tp+1] one past the last valid frame of row u
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Tw = a.Tp;  // row width of the call's buffers
    const int* best = a.best + (long long)b * Tw;
    const int bs = a.src_mod > 0 ? b % a.src_mod : b;
    const unsigned char* km = a.keymask + (long long)bs * Tw;
    int* shift = a.shift + (long long)b * Tw;
    int* iv = a.intervals + (long long)b * (Tw + 1) * 4;
    // merged pass: the utterance's own batch has fewer frames than the call - ITS width is what the reference's tensors had
    // (frames at or past it do not exist: keymask is 0 there, and the shift below must not carry a token onto them)
    const int Tp = a.utt_meta ? a.utt_meta[bs].tp : Tw;

    // rows start empty: (INT_MAX, 0) is the identity of (min, max)
    for (int u = tid; u <= Tw; u += 256) {
        s_lo[u] = 0x7fffffff;
        s_hi[u] = 0;
    }
    if (tid == 0) s_carry[0] = 0;
    __syncthreads();

    // path[t] = keymask ? argmax : 0 ; collapsed[t] = path[t]==path[t-1] ? 0 : path[t] (path[-1]=0)
    // shift[t] = collapsed[t-1], shift[0] = 0 ; c[t] = #{t' <= t : shift[t'] != blank}
    for (int base = 0; base < Tw; base += 256) {
        const int t = base + tid;
        int sh = 0;
        if (t < Tp && t >= 1) {
            const int p1 = (a.raw_path || km[t - 1]) ? best[t - 1] : 0;
            const int p2 = (t >= 2) ? ((a.raw_path || km[t - 2]) ? best[t - 2] : 0) : 0;
            sh = (p1 == p2) ? 0 : p1;
        }
        if (t < Tw) shift[t] = sh;
        const int nz = (t < Tp && sh != a.blank) ? 1 : 0;
        // inclusive block scan
        s_scan[tid] = nz;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int c = s_carry[0] + s_scan[tid];
        if (t < Tp && km[t]) {
            atomicMin(&s_lo[c], t);
            atomicMax(&s_hi[c], t + 1);
        }
        __syncthreads();
        if (tid == 255) s_carry[0] = c;
        __syncthreads();
    }
    // number of non-blank entries of shift - or the given label count (forced alignment: viterbi_align returns its input ylens)
    const int ylen0 = a.ylen_in ? a.ylen_in[b] : s_carry[0];
    // src_size = (ratio * T').long()  (src/models/cassnat.py:436): fp32 product, truncation toward zero
    const int ssz = (int)(long long)(a.size_ratio[bs] * (float)Tp);
    if (tid == 0) {
        a.src_size[b] = ssz;
        a.ylen[b] = ylen0 + (a.no_trigger ? 0 : 1);
        atomicMax(a.ymax, ylen0 + (a.no_trigger ? 0 : 1));
    }
    for (int u = tid; u <= Tw; u += 256) {
        int s1 = s_lo[u], e1 = s_hi[u], s2 = 0, e2 = 0;
        if (a.no_trigger) {  // trigger_mask = src_mask (cassnat.py:470): all frames of the utterance's own batch, & keymask downstream
            s1 = 0;
            e1 = Tp;
        } else if (s1 == 0x7fffffff) {
            s1 = 0;
            e1 = 0;
        } else {
            if (a.left > 0) s1 = s1 > 0 ? s1 - 1 : 0;
            if (a.right > 0) e1 = e1 < Tp ? e1 + 1 : Tp;
        }
        if (u == ylen0 && !a.no_trigger) {  // trigger_mask[b, ylens[b], src_size[b]-1] = 1, python negative index wraps
            int f = ssz - 1;
            if (f < 0) f += Tp;
            if (f >= 0 && f < Tp) {
                s2 = f;
                e2 = f + 1;
                if (a.left > 0) s2 = s2 > 0 ? s2 - 1 : 0;
                if (a.right > 0) e2 = e2 < Tp ? e2 + 1 : Tp;
            }
        }
        iv[4 * u + 0] = s1;
        iv[4 * u + 1] = e1;
        iv[4 * u + 2] = s2;
        iv[4 * u + 3] = e2;
    }
}

int launch_ctc_align(const AlignArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.Tp <= 0) return 0;
    CN_HIP_CHECK(hipMemsetAsync(a.ymax, 0, sizeof(int), s));
    const size_t lds = (256 + 4 + 2 * (size_t)(a.Tp + 1)) * sizeof(int);
    if (lds > 160 * 1024) {
        cn_set_error("ctc_align: T' too large for one workgroup's LDS");
        return -1;
    }
    static CnAttrOnce attr_once;
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        CN_HIP_CHECK(hipFuncSetAttribute((const void*)ctc_align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL(ctc_align_kernel, dim3(a.B), dim3(256), lds, s, a);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// CTC greedy hypothesis as a decoder input (Transformer.fast_decode_with_ctc, src/models/transformer.py:256-270): arg-max path
// zeroed on masked frames, repeats of the previous frame's label zeroed, the non-zero labels compacted behind sos; the row is
// padded with `pad`.  One wave per utterance; tgt [B][ld] (ld >= Tp + 1), len [B] (labels, without sos), maxlen [1].
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int* __restrict__ best, const unsigned char* __restrict__ km, int B, int Tp,
                                                          int sos, int pad, int ld, int* __restrict__ tgt, int* __restrict__ len,
                                                          int* __restrict__ keylen, int* __restrict__ maxlen) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    const int* p = best + (long long)b * Tp;
    const unsigned char* k = km + (long long)b * Tp;
    int* row = tgt + (long long)b * ld;
    int count = 0;
    for (int base = 0; base < Tp; base += 64) {
        const int t = base + lane;
        int tok = 0;
        if (t < Tp) {
            const int cur = k[t] ? p[t] : 0;
            const int prev = t >= 1 ? (k[t - 1] ? p[t - 1] : 0) : 0;
            tok = cur == prev ? 0 : cur;
        }
        const unsigned long long m = __ballot(tok != 0);
        if (tok != 0) row[1 + count + __popcll(m & ((1ull << lane) - 1ull))] = tok;
        count += __popcll(m);
    }
    for (int i = count + 1 + lane; i < ld; i += 64) row[i] = pad;
    if (lane == 0) {
        row[0] = sos;
        len[b] = count;
        keylen[b] = count + 1;  // keys the causal + padding mask allows: sos and the labels (tgt_input != padding_idx)
        atomicMax(maxlen, count);
    }
}

int launch_ctc_collapse(const int* best, const unsigned char* km, int B, int Tp, int sos, int pad, int ld, int* tgt, int* len, int* keylen,
                        int* maxlen, hipStream_t s) {
    if (B <= 0 || Tp <= 0 || ld < Tp + 1) return -1;
    CN_HIP_CHECK(hipMemsetAsync(maxlen, 0, sizeof(int), s));
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), 0, s, best, km, B, Tp, sos, pad, ld, tgt, len, keylen, maxlen);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

// beam_width == 1 case of src/models/cassnat.py:574-637: position i is consumed while i <= ylen[b]
// (ylen already includes the EOS row), the score is a sequential double sum of the float32 maxima.
// sub > 0: the batch is `B / sub` coalesced reference batches of `sub` utterances each; the row limit of an utterance is then
// the largest ylen of ITS batch (what U would have been had that batch been decoded alone), so its hypothesis is the same.
__global__ __launch_bounds__(64) void greedy_pack_kernel(const int* __restrict__ tok, const float* __restrict__ val,
                                                         const int* __restrict__ ylen, int B, int U, int sos, int hyp_stride,
                                                         int* __restrict__ hyp, int* __restrict__ hyp_len, double* __restrict__ score, int sub,
                                                         const UttMeta* __restrict__ utt_meta, const int* __restrict__ ymax_dev) {
    // one wave per utterance (round 2: one THREAD per utterance - 16 to 45 us for a merged pass): the lanes share the scan over
    // the batch's row counts and the token copy; the score stays ONE sequential double sum, in row order, on lane 0 (the
    // reference accumulates a Python float row by row: any other order differs in the last bits)
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    int ulim = U;
    if (ymax_dev && *ymax_dev < ulim) ulim = *ymax_dev;  // U was a prediction: the reference's row count is the true maximum
    if (utt_meta || (sub > 0 && sub < B)) {
        int b0, b1;
        if (utt_meta) {
            b0 = utt_meta[b].sub_lo;
            b1 = utt_meta[b].sub_hi;
        } else {
            b0 = (b / sub) * sub;
            b1 = b0 + sub < B ? b0 + sub : B;
        }
        int um = 0;
        for (int j = b0 + lane; j < b1; j += 64) um = ylen[j] > um ? ylen[j] : um;
        for (int o = 32; o > 0; o >>= 1) {
            const int v = __shfl_xor(um, o);
            um = v > um ? v : um;
        }
        if (um < ulim) ulim = um;
    }
    int n = ylen[b] + 1;
    if (n > ulim) n = ulim;
    if (n > hyp_stride - 1) n = hyp_stride - 1;
    int* h = hyp + (long long)b * hyp_stride;
    for (int i = lane; i < hyp_stride; i += 64) h[i] = i == 0 ? sos : (i <= n ? tok[(long long)b * U + i - 1] : 0);
    if (lane == 0) {
        double sc = 0.0;
        for (int i = 0; i < n; ++i) sc = sc + (double)val[(long long)b * U + i];
        hyp_len[b] = n + 1;
        score[b] = sc;
    }
}

int launch_greedy_pack(const int* tok, const float* val, const int* ylen, int B, int U, int sos, int hyp_stride,
                       int* hyp, int* hyp_len, double* score, hipStream_t s, int sub, const UttMeta* utt_meta, const int* ymax_dev) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(greedy_pack_kernel, dim3(B), dim3(64), 0, s, tok, val, ylen, B, U, sos,
                       hyp_stride, hyp, hyp_len, score, sub, utt_meta, ymax_dev);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
