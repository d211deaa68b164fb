// Host-side launchers of the hand-written gfx950 kernels (one .hip file each).
// Every launcher is stream-ordered, allocates nothing and returns 0 / negative error code.
#pragma once
#include "common.h"

// model precisions of the kernels (the public CN_PRECISION_FP8 = 2 is a CN_PREC_BF16 engine with fp8 encoder products)
enum { CN_PREC_F32 = 0, CN_PREC_BF16 = 1, CN_PREC_X3 = 3 };  // X3: split-bf16 elements (common.h split_t), 3 bf16 MFMAs per product
static inline size_t cn_elem_size(int prec) { return prec == CN_PREC_BF16 ? 2 : 4; }

// ---- GEMM:  C[M][N] = epi( A[M][K] . W[N][K]^T + bias[N] )            (gemm.hip)
enum { CN_EPI_RELU = 1, CN_EPI_RESID = 2, CN_EPI_EMBED = 4, CN_EPI_SWISH = 8 };
struct GemmArgs {
    const void* A = nullptr;  // activations, model precision, row stride lda (elements)
    int lda = 0;
    const void* W = nullptr;  // weights [N][K], model precision, K contiguous
    const float* bias = nullptr;
    void* C = nullptr;  // output, row stride ldc; fp32 when c_f32 else model precision
    int ldc = 0;
    int c_f32 = 0;
    int M = 0, N = 0, K = 0;
    int epi = 0;
    const float* resid = nullptr;  // CN_EPI_RESID: C = resid + resid_scale * (acc + bias); fp32 [M][ldr]; may alias C
    float resid_scale = 1.0f;      // SublayerConnection's `scale` (0.5 for the conformer's macaron feed-forward halves)
    int ldr = 0;
    const float* pe = nullptr;  // CN_EPI_EMBED: C = (acc + bias) * scale + pe[m % pe_period][n]  (pe may be null: scale only)
    int pe_period = 1;
    float scale = 1.f;
    // implicit-GEMM A operand for the second 3x3/stride-2 subsampling convolution:
    // A[m=(b,t2,f2)][k=(kh,kw,c)] = conv1[b][2*t2+kh-1][2*f2+kw-1][c] (channels-last), zero outside.
    // fp8 operands (BASELINE config 5): A and W are e4m3fn bytes carrying per-tensor scales; acc_scale = 1 / (scale_A * scale_W)
    // restores the product.  c_fp8: the output is e4m3fn too, scaled by c_scale after the epilogue (feeds the next fp8 product)
    int ab_fp8 = 0;
    float acc_scale = 1.f;
    const float* w_inv_scale = nullptr;  // device: 1 / (the weights' scale), multiplied into acc_scale (may be null)
    int c_fp8 = 0;
    float c_scale = 1.f;
    int conv = 0;
    int conv_halo = 0;  // the input image carries a one-cell zero halo: [B][T1 + 2][F1 + 2][C] (conv1 with halo = 1)
    int cB = 0, cT1 = 0, cF1 = 0, cC = 0, cT2 = 0, cF2 = 0;
};
int launch_gemm(int prec, const GemmArgs& a, hipStream_t s);

// ---- first subsampling convolution, 1 -> C channels, 3x3 stride 2 pad 1, + ReLU   (conv1.hip)
// x (B,T,F) fp32  ->  out (B,T1,F1,C) channels-last in model precision.  w is [9][C] (tap-major).
// utt_meta (may be null): per-utterance records of a MERGED engine pass (see UttMeta below) - utterance b then belongs to a
// reference batch of utt_meta[b].frames <= T frames: input frames at or past that count read as the convolution's zero padding
// and image rows at or past (frames - 1) / 2 + 1 are written as zeros (what the second convolution's padding is in a pass of
// that batch alone).
struct UttMeta {
    int frames;    // T of the utterance's own reference batch (collate pads a batch to ITS longest utterance, speech_loader.py:327-356)
    int tp;        // its subsampled length T' = ((frames - 1) / 2 + 1 - 1) / 2 + 1
    int sub_lo;    // first utterance of that batch in the merged call
    int sub_hi;    // one past its last utterance
};
int launch_conv1(int prec, const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1,
                 int F1, int C, int halo, hipStream_t s, const UttMeta* utt_meta = nullptr);
// split-bf16 engine: the image as two bordered bf16 planes (hi, then lo), for launch_conv2_x3; halo 1 / 2 as above
int launch_conv1_planes(const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1, int F1,
                        int C, int halo, hipStream_t s, const UttMeta* utt_meta = nullptr);
// fp8 engine (BASELINE config 5): the bordered image as e4m3fn bytes at `scale`, for launch_conv2_f8
int launch_conv1_mixplanes(const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1, int F1, int C,
                           int halo, float scale_l, float scale_q, hipStream_t s, const UttMeta* utt_meta = nullptr);
int launch_conv1_f8(const float* x, const float* w9c, const float* bias, void* out8, int B, int T, int F, int T1, int F1, int C,
                    int halo, float scale, hipStream_t s, const UttMeta* utt_meta = nullptr);
// bf16 engine: the bordered bf16 image on the matrix cores (the e4m3 form's kernel with 512-byte cells); C == 256, F1 + 2 >= 32
bool conv1_bordered_bf16_applies(int C, int F1);
int launch_conv1_bordered_bf16(const float* x, const float* w9c, const float* bias, void* out, int B, int T, int F, int T1, int F1,
                               int C, hipStream_t s, const UttMeta* utt_meta = nullptr);
// the UttMeta records of a merged pass from its (rows, frames) list: n_sub <= CN_MAX_SUB batches, given by value
constexpr int CN_MAX_SUB = 64;
struct SubList {
    int n;
    int rows[CN_MAX_SUB];
    int frames[CN_MAX_SUB];
};
int launch_expand_meta(const SubList& subs, UttMeta* meta, int B, hipStream_t s);

// ---- row kernels                                                                  (rowops.hip)
// y = a_2 * (x - mean) / (std_unbiased + eps) + b_2 ; x fp32 [M][d] ; y model precision (or fp32 if y_f32)
int launch_layernorm(int prec, const float* x, const float* a2, const float* b2, void* y, int y_f32, int M, int d,
                     float eps, hipStream_t s);
// fp8 operands of config 5: LayerNorm straight into e4m3fn at a per-tensor scale; bf16 -> e4m3fn of a finished tensor
int launch_layernorm_fp8(const float* x, const float* a2, const float* b2, void* y, int M, int d, float eps, float scale,
                         hipStream_t s);
int launch_quantize_fp8(const void* src_bf16, int ld, void* dst, int M, int K, float scale, hipStream_t s);
int launch_cmvn(float* x, const int* len, const double* mean, const double* sd, int B, int T, int F, hipStream_t s);
// fp16 build: *fault = 1 (device-visible host word) when some |x[i]| > *limit (device word; <= 0: no check)
int launch_feature_range(const float* x, size_t n, const float* limit, unsigned int* fault, hipStream_t s);
// packed archive rows of a pass -> padded (rows, T, F) batch, optional global CMVN in float64 (the reader's collate on the device)
int launch_unpack_rows(const float* packed, const int* off, const int* len, float* out, int rows, int T, int F, float pad,
                       const double* mean, const double* sd, hipStream_t s);
// per row of logits [M][V] (fp32): first-index argmax, max log-prob; optionally rewrites the row as log-softmax.
int launch_logsoftmax_argmax(float* logits, int M, int V, int ldl, int* arg, float* maxlp, int write_logp,
                             hipStream_t s);
// keymask[b][j] = feats[b][stride*j][0] != padding  (mask[:, :, ::2][:, :, ::2] of the reference, stride 4)
int launch_keymask(const float* feats, int B, int T, int F, int Tp, int stride, float padding, unsigned char* km,
                   hipStream_t s);
// out[b][u][:] = table[u][:]  (extractor queries), optional shift for use_unimask handled by caller
int launch_fill_queries(const float* table, float* out, int B, int U, int d, hipStream_t s);
// use_unimask: y[b][0] = 0 ; y[b][u] = x[b][u-1]
int launch_shift_right(const float* x, float* y, int B, int U, int d, hipStream_t s);
int launch_esa_paths(const int* top2_idx, const float* top2_val, const unsigned char* select, float threshold, int* best, int M,
                     int n, hipStream_t s);
int launch_lm_embed(const int* tok, int ld, const float* lut, const float* pe, float* x, int B, int U, int d, float scale,
                    hipStream_t s);
int launch_gather_logp(const float* logp, int V, const int* tgt, int ld, float* out, int B, int U, hipStream_t s);
int launch_fill_int(int* p, size_t n, int v, hipStream_t s);
int launch_copy_rows(void* dst, int dst_ld, const void* src, int src_ld, int width, int rows, hipStream_t s);  // 4-byte words
int launch_convert(int prec, const float* src, void* dst, size_t n, hipStream_t s);       // fp32 -> model precision
int launch_convert_back(int prec, const void* src, float* dst, size_t n, hipStream_t s);  // model precision -> fp32

// ---- fused multi-head attention (flash-style, d_k = 64)                           (attention.hip)
struct AttnArgs {
    const void* Q = nullptr;  // [B*Lq][ldq], head h at column h*64
    const void* K = nullptr;  // [B*Lk][ldk]
    const void* V = nullptr;  // [B*Lk][ldv]
    void* O = nullptr;        // [B*Lq][ldo] model precision
    int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
    int B = 0, H = 0, Lq = 0, Lk = 0;
    const unsigned char* keymask = nullptr;  // [B][Lk] or null (all valid)
    int kv_mod = 0;  // > 0: K / V / keymask of batch entry b are those of entry b % kv_mod (ESA: many alignments per utterance)
    const int* kv_index = nullptr;  // non-null: ... those of entry kv_index[b] (AST beam search: the row's utterance)
    const int* klen = nullptr;               // [B] or null: key j valid iff j < klen[b]
    // merged passes of batches with different frame counts: K / V entry e holds only kcap[e * kcap_stride] <= Lk keys of its own
    // batch; the rest are the merged pass's padding and get -inf like tile padding (NOT the float-min fill of a masked key: a
    // row whose keys are all masked attends uniformly over the keys its own batch has, attention.py:19-21).  Null: Lk keys
    const int* kcap = nullptr;
    int kcap_stride = 1;
    // bf16 only: Q (q_blocked) / K and V (kv_blocked) live in a blocked matrix (common.h: cn_blk16_off) of q_n / kv_n columns -
    // what the row-chain kernel's tail writes; Q / K / V then point at the matrix itself and q_col / k_col / v_col give the
    // first column of head 0 (ldq / ldk / ldv are unused for a blocked operand)
    int q_blocked = 0, kv_blocked = 0;
    int q_col = 0, k_col = 0, v_col = 0, q_n = 0, kv_n = 0;
    // bf16 only: O is written as the row-chain kernel reads its B operands - per 32-row block [k-step of 16 channels][lane = 32 *
    // (bit 3 of the channel) + row % 32][8 bf16], i.e. a blocked matrix of 64-byte-wide column tiles: byte offset of the chunk at
    // (row m, channel c) = ((m >> 5) * (ldo / 16) + (c >> 4)) * 1024 + (((c >> 3) & 1) * 32 + (m & 31)) * 16
    int o_blocked = 0;
    const int* intervals = nullptr;          // [B][iv_stride][4] (s1,e1,s2,e2) per query row, or null
    int iv_stride = 0;
    int causal = 0;  // key j allowed only if j <= i
    float scale = 0.125f;
    // relative-position self attention (RelMultiHeadedAttention, attention.py:68-147): score(i, j) = ((q_i + u) . k_j +
    // (q_i + v) . P[clamp(j - i, -R, R) + R]) * scale; rows with no allowed key yield 0.  rel_pos: [2R+1][ld_pos] fp32
    // projected position rows (head h at column h*64), rel_u / rel_v: [H*64] fp32
    const float* rel_pos = nullptr;
    const float *rel_u = nullptr, *rel_v = nullptr;
    int rel_R = 0, ld_pos = 0;
};
int launch_attention(int prec, const AttnArgs& a, hipStream_t s);
int attention_print_stamps();  // CASSNAT_ATTN_STAMPS: phase timestamps of the last launch's workgroup 0

// ---- conformer convolution module pieces (conformer.hip)
int launch_glu(int prec, const void* in, void* out, int M, int d, hipStream_t s);
int launch_dwconv(int prec, const void* x, const float* w, const float* bias, float* y, int B, int L, int d, int k, hipStream_t s);
int launch_groupnorm_swish(int prec, const float* x, double* stats, const float* gw, const float* gb, void* out, int B, int L,
                           int d, float eps, hipStream_t s);

// ---- CTC greedy alignment -> trigger intervals (integer, exact)                   (ctc_align.hip)
struct AlignArgs {
    const int* best = nullptr;               // [B][Tp] argmax of the CTC posteriors
    const unsigned char* keymask = nullptr;  // [B][Tp]
    const float* size_ratio = nullptr;       // [B] fp32 length ratios
    int B = 0, Tp = 0, blank = 0, left = 0, right = 0;
    int src_mod = 0;  // > 0: keymask / size_ratio of entry b are those of entry b % src_mod (ESA: best[] holds many paths per utterance)
    int* shift = nullptr;      // [B][Tp]  aligned_seq_shift
    int* src_size = nullptr;   // [B]
    int* ylen = nullptr;       // [B]   (token count + 1)
    int* ymax = nullptr;       // [1]   max ylen
    int* intervals = nullptr;  // [B][Tp+1][4]
    // forced alignment (decode_type ctc_att, src/models/cassnat.py:391-414): `best` holds the per-frame labels of the Viterbi
    // path, which are NOT zeroed on masked frames, and the label counts are the given ones (align_to_mask forces row ylens[b])
    int raw_path = 0;
    const int* ylen_in = nullptr;  // [B] or null
    // merged pass (null otherwise): utterance b's own batch has utt_meta[b].tp <= Tp frames - the width of its ctc_out in the
    // reference: the shift drops a token that starts on ITS last frame, src_size = (ratio * tp).long(), rows wrap at tp
    const UttMeta* utt_meta = nullptr;
    // args.use_trigger == False (src/models/cassnat.py:469-473): trigger_mask = src_mask - every row's interval is the utterance's
    // whole frame range (the attention kernel ANDs it with the key mask) - and ylen / ymax are best_path_align's own counts: no
    // EOS row is added (align_to_mask, which adds it, does not run)
    int no_trigger = 0;
};
int launch_ctc_align(const AlignArgs& a, hipStream_t s);
// CTC greedy hypothesis compacted behind sos (Transformer.fast_decode_with_ctc's decoder input)
int launch_ctc_collapse(const int* best, const unsigned char* km, int B, int Tp, int sos, int pad, int ld, int* tgt, int* len, int* keylen,
                        int* maxlen, hipStream_t s);

// ---- CTC prefix beam search + forced alignment (decode_type ctc_only / ctc_att)              (ctc_beam.hip)
struct CtcBeamArgs {
    const float* logp = nullptr;        // [B][Tp][V] CTC log-posteriors of ALL frames
    const int* top_idx = nullptr;       // [B][Tp][P] the P best labels per frame, best first
    const float* size_ratio = nullptr;  // [B]
    int B = 0, Tp = 0, V = 0, P = 0, W = 0, blank = 0, Lmax = 0;
    double lp = 0.0;                    // args.ctc_lp
    unsigned char* hist_parent = nullptr;  // [B][Tp][W] scratch
    int* hist_tok = nullptr;               // [B][Tp][W] scratch
    int* hyp = nullptr;       // [B][W][Lmax]
    int* hyp_len = nullptr;   // [B][W]
    double* score = nullptr;  // [B][W] score_ctc
    double* p_blk = nullptr;  // [B][W]
    double* p_nblk = nullptr;
    int* n_out = nullptr;     // [B] hypotheses kept
};
int launch_ctc_prefix_beam(const CtcBeamArgs& a, hipStream_t s);
struct ViterbiArgs {
    const float* logp = nullptr;             // [B][Tp][V]
    const unsigned char* keymask = nullptr;  // [B][Tp]
    const float* size_ratio = nullptr;       // [B]
    const int* labels = nullptr;             // [B][ld]
    const int* label_len = nullptr;          // [B]
    int B = 0, Tp = 0, V = 0, ld = 0, ymax = 0, blank = 0;
    unsigned char* bp = nullptr;  // [B][Tp][2 ymax + 1] scratch
    int* out_path = nullptr;      // [B][Tp] label of the aligned state per frame (blank past src_size)
};
int launch_ctc_viterbi(const ViterbiArgs& a, hipStream_t s);
// hyp[b] = [sos] + tok[b][0 .. min(ylen[b]+1, U)) ; score = sequential double sum of val
// sub > 0: equal-sized coalesced batches of `sub` utterances; utt_meta: the batches of a merged pass (any sizes); ymax_dev
// (may be null): the true row count of the call when U is a prediction (hypotheses are limited by min(U, *ymax_dev))
int launch_greedy_pack(const int* tok, const float* val, const int* ylen, int B, int U, int sos, int hyp_stride,
                       int* hyp, int* hyp_len, double* score, hipStream_t s, int sub = 0, const UttMeta* utt_meta = nullptr,
                       const int* ymax_dev = nullptr);
// per row top-k (k <= 16) of log-probs [M][V] -> idx/val [M][k], sorted descending (ties: lower index first)
int launch_topk(const float* logp, int M, int V, int ldl, int k, int* idx, float* val, hipStream_t s);
// log_softmax(logits / T) and its per-row top-k in one pass (the (M, V) log-probabilities are not written)
int launch_logsoftmax_topk(const float* logits, int M, int V, int ldl, float temperature, int k, int* idx, float* val,
                           hipStream_t s);

// ---- fused FFN sublayer, bf16 / d_model == 256                                     (fused.hip)
//   x <- x + W2 relu(W1 LN(x) + b1) + b2 ;  optionally xn_out <- LN_next(x) in bf16
struct FfnFusedArgs {
    float* x = nullptr;
    const float *ln_a = nullptr, *ln_b = nullptr;
    const void* w1p = nullptr;  // pack_ffn_w1 image
    const float* b1 = nullptr;
    const void* w2p = nullptr;  // pack_ffn_w2 image
    const float* b2 = nullptr;
    const float *nln_a = nullptr, *nln_b = nullptr;
    void* xn_out = nullptr;
    int M = 0, d = 0, dff = 0;
    float eps = 1e-6f;
    // d_ff split over `nslice` workgroups per row tile (few rows): W2 partial products go to partial[nslice][M][256] and
    // launch_ffn_reduce finishes the sublayer (x += b2 + sum of slices, optional next LayerNorm); x, b2 and the next norm
    // are not used by the split launch itself
    int nslice = 1;
    float* partial = nullptr;
};
int launch_ffn_fused(const FfnFusedArgs& a, hipStream_t s);
int launch_ffn_reduce(float* x, const float* partial, int nslice, const float* b2, const float* nln_a, const float* nln_b,
                      void* xn_out, int M, float eps, hipStream_t s);
void pack_ffn_w1(const float* w1, int dff, uint16_t* out);  // [dff][256] fp32 -> fragment stream (dff*256 bf16)
void pack_ffn_w2(const float* w2, int dff, uint16_t* out);  // [256][dff] fp32 -> fragment stream (dff*256 bf16)

// ---- the same sublayer in the split-bf16 (bf16x3) precision: hi + lo operands, three MFMAs per product   (fused_x3.hip)
struct FfnX3Args {
    float* x = nullptr;
    const float *ln_a = nullptr, *ln_b = nullptr;
    const void* wst = nullptr;  // pack_ffn_x3 stream
    bool mix = false;           // ... packed for the mixed arithmetic (half-precision hi x hi + e4m3 cross terms: fused_x3.hip MIXF)
    const float *b1 = nullptr, *b2 = nullptr;
    const float *nln_a = nullptr, *nln_b = nullptr;
    void* xn_out = nullptr;  // [M][256] split-bf16 (when nln_a)
    int M = 0, d = 0, dff = 0;
    float eps = 1e-6f;
    // row-chain form (the encoder / self-attention layer of the split-bf16 engine in one launch around the attention kernel):
    // ctx != null: x <- x + Wo . ctx + bo first (ctx split-bf16 [M][ldctx], wo_p = pack_proj_x3 stream of the 256 x 256 output
    // projection, bo its bias); tail_p != null: instead of writing LN_next(x), project it - tail_out (split-bf16 rows of
    // ld_tail elements) <- Wt . LN_next(x) + bt, tail_n columns (a multiple of 128, pack_proj_x3 stream); needs nln_a
    const void* ctx = nullptr;
    int ldctx = 256;
    const void* wo_p = nullptr;
    const float* bo = nullptr;
    const void* tail_p = nullptr;
    const float* tail_b = nullptr;
    void* tail_out = nullptr;
    int tail_n = 0, ld_tail = 0;
};
bool ffn_x3_applies(int d, int dff);
int launch_ffn_x3(const FfnX3Args& a, hipStream_t s);
size_t ffn_x3_stream_bytes(int dff);
void pack_ffn_x3(const float* w1, const float* w2, int dff, uint16_t* out, bool mix = false);
bool ffn_mix_applies();  // the split-bf16 engine's feed-forward sublayers run the mixed arithmetic (experiments: CASSNAT_NO_FFN_MIX)

// ---- waveform -> log-mel filterbank (+ CMVN), Kaldi compute-fbank-feats semantics with dither 0 (fbank.hip)
struct FbankOpts {
    float sample_rate = 16000.f, frame_length_ms = 25.f, frame_shift_ms = 10.f, preemph = 0.97f, low_freq = 20.f, high_freq = 0.f;
    int num_mel = 80, window_type = 0 /* 0 hamming, 1 povey, 2 hanning, 3 rectangular */, remove_dc = 1, use_power = 1, use_log = 1;
};
int fbank_num_frames(const FbankOpts& o, int num_samples);
int launch_fbank(const FbankOpts& o, const float* wave, const int* num_samples, int B, int max_samples, const float* cmvn_mean,
                 const float* cmvn_istd, float* out, int Tmax, float pad_value, hipStream_t s);

// ---- conv2 as an LDS-DMA implicit GEMM, bf16 / 256 -> 256 channels (conv2.hip); launch_gemm dispatches to it
bool conv2_dma_applies(int prec, int C, int N);
// split-bf16 form of the same kernel: image and weights as hi / lo bf16 planes, three K steps per K step of the bf16 loop
bool conv2_f8_applies(int C, int N);
// out8_scale > 0: the output rows are e4m3fn bytes at that scale ([M][256] bytes) instead of bf16 - the input of launch_linear256_f8
int launch_conv2_f8(const void* in8, const void* w8, const int* q8_dev, const float* bias, void* out, int B, int T1, int F1, int T2,
                    int F2, hipStream_t s, float out8_scale = 0.f);
bool linear256_f8_applies(int N, int K);
int launch_linear256_f8(const void* A8, const void* W8, const int* q8_dev, const float* bias, float* out, int M, int K, float scale,
                        const float* pe, int pe_period, hipStream_t s);
bool conv2_mix_applies(int prec, int C, int N);
int launch_conv2_mix(const void* img, const void* w_hi, const void* w8q, const void* w8l, const int* q8_dev, const float* bias, void* out,
                     int B, int T1, int F1, int T2, int F2, hipStream_t s);
bool conv2_x3_applies(int prec, int C, int N);
int launch_conv2_x3(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias, void* out, int B,
                    int T1, int F1, int T2, int F2, hipStream_t s);
bool linear256_dma_applies(int prec, int N, int K);
int launch_linear256_dma(const void* A, int lda, const void* W, const float* bias, float* out, int M, int K, float scale,
                         const float* pe, int pe_period, hipStream_t s);
int launch_conv2_dma(const void* in, const void* w, const float* bias, void* out, int B, int T1, int F1, int T2, int F2,
                     hipStream_t s);

// ---- row-chain kernel, bf16 / d_model == 256: out-projection + residual, FFN sublayer + residual, next pre-norm and the
// next attention's input projection for 128-row blocks, activations in registers, weights streamed once per block (chain.hip)
constexpr int CHAIN_TAB_FLOATS = 5120;
constexpr int CHAIN_UNIT_BYTES = 16384;
struct ChainArgs {
    float* x = nullptr;          // [M][256] fp32 residual stream, in place
    const void* ctx = nullptr;   // [M][ldctx] bf16 attention context; null = no output projection
    int ldctx = 0;
    int ctx_blocked = 0;         // ctx comes in the attention kernel's `o_blocked` layout (ldctx == 256): 1-KiB load instructions
    const void* wstream = nullptr;  // pack_chain units
    const float* tab = nullptr;     // pack_chain table (CHAIN_TAB_FLOATS)
    void* out = nullptr;            // [M][ldo] bf16: tail projection of LNn(x), or LNn(x) itself when tail_n == 0
    int ldo = 0;
    void* ln_out = nullptr;         // optional: LNn(x) itself [M][ld_ln] bf16 IN ADDITION to a tail projection
    int ld_ln = 0;
    int M = 0, d = 0, dff = 0, tail_n = 0, has_next = 0;
    // x layout: row-major [M][256], or blocked: 32-row blocks of [32 pieces i = 4 nt + g][64 lanes][4 floats] where lane =
    // (row % 32) + 32 half holds channels 32 nt + 8 g + 4 half + (0..3) - each load/store instruction moves 1 KiB contiguous
    // (buffer must hold ceil(M / 32) * 32 rows).  store_x = 0: x is not written back (nothing reads it afterwards)
    int x_in_blocked = 0, x_out_blocked = 0, store_x = 1;
    int out_blocked = 0;  // the tail projection `out` is written as a blocked bf16 matrix of ldo columns (common.h: cn_blk16_off)
    int swish = 0;  // feed-forward activation: x * sigmoid(x) instead of ReLU (conformer)
    int f8 = 0;     // the stream's feed-forward units are e4m3 (pack_chain with f8_q)
    const int* f8_q = nullptr;  // device: the four scale bytes pack_chain returned
    float eps = 1e-6f;
};
int launch_chain(const ChainArgs& a, hipStream_t s);
int chain_print_stamps();  // investigation aid (CASSNAT_CHAIN_STAMPS): phase times of workgroup 0 of the last launch
struct ChainWeights {  // host fp32, nn.Linear layout [out][in]; null members = stage absent
    const float *wo = nullptr, *bo = nullptr, *ln1_a = nullptr, *ln1_b = nullptr, *w1 = nullptr, *b1 = nullptr, *w2 = nullptr,
                *b2 = nullptr, *nln_a = nullptr, *nln_b = nullptr, *wt = nullptr, *bt = nullptr;
    int dff = 0, tail_n = 0;
};
// F8 form (BASELINE config 5): the two feed-forward products on e4m3 operands - weights at a per-tensor power-of-two scale
// chosen at pack time, LayerNorm outputs at x16 and ReLU outputs at x8 as in the unfused fp8 path (model.hip); d_ff % 256 == 0.
// pack_chain with f8_q != null writes such a stream and returns the four E8M0 scale bytes the kernel's products need.
constexpr float CHAIN_F8_S_LN = 16.f, CHAIN_F8_S_HID = 8.f;
size_t chain_stream_units(int has_outproj, int dff, int tail_n, int f8 = 0);
void pack_chain(const ChainWeights& w, uint16_t* stream, float* tab, int* f8_q = nullptr);
// float -> OCP e4m3fn byte, round to nearest even, saturating at +-448 (host side; the device's v_cvt_pk_fp8_f32 after a clamp)
unsigned char cn_f32_to_e4m3_host(float f);

// ---- d_model-deep projections of the split-bf16 engine (K = 256): C = A . W^T + bias, split-bf16 or fp32 (+ residual) output   (proj_x3.hip)
struct ProjX3Args {
    const void* A = nullptr;   // [M] rows of 256 split-bf16 elements, row stride lda elements
    int lda = 256;
    const void* wp = nullptr;  // pack_proj_x3 stream (N * 1024 bytes)
    const float* bias = nullptr;
    void* C = nullptr;         // fp32 [M][ldc] when c_f32, else split-bf16 rows of ldc elements
    int ldc = 0;
    int c_f32 = 0;
    const float* resid = nullptr;  // fp32 output only: C = resid + resid_scale * (A . W^T + bias); may alias C
    int ldr = 0;
    float resid_scale = 1.f;
    int M = 0, N = 0;
};
bool proj_x3_applies(int N, int K);
int launch_proj_x3(const ProjX3Args& a, hipStream_t s);
size_t proj_x3_off(int n, int k);  // byte offset of W[n][k]'s hi half in the packed stream; the lo half 1024 bytes further
void pack_proj_x3(const float* w, int N, unsigned char* out);

// ---- fused generator tail, bf16 or split-bf16 / d_model == 256: per-row argmax and max log-probability of
// log_softmax(W h + b)   (genmax.hip)
struct GenmaxArgs {
    const void* h = nullptr;   // [M][256] bf16 (split-bf16 when x3)
    const void* wp = nullptr;  // pack_genmax / pack_genmax_x3 weight fragments
    const float* bp = nullptr; // pack_genmax / pack_genmax_x3 biases
    bool x3 = false;           // split-bf16 operands, three MFMAs per product
    int* arg = nullptr;
    float* maxlp = nullptr;
    int M = 0, V = 0, d = 0;
    // language-model scoring: tgt != null -> tgt_lp[b * tgt_ld + u] = log_softmax(W h[b * tgt_U + u] + b)[tgt[b * tgt_ld + u]]
    // (arg / maxlp unused)
    const int* tgt = nullptr;
    float* tgt_lp = nullptr;
    int tgt_U = 0, tgt_ld = 0;
};
int launch_genmax(const GenmaxArgs& a, hipStream_t s);
int genmax_vtw(int V);  // vocabulary tiles per wave; packed sizes: weights 4*vtw*16 KiB, biases 4*vtw*32 floats
void pack_genmax(const float* w, const float* b, int V, uint16_t* wout, float* bout);
int genmax_x3_vtw(int V);  // split-bf16: tiles per wave (8 waves); packed sizes: weights 8*vtw*32 KiB, biases 8*vtw*32 floats
void pack_genmax_x3(const float* w, const float* b, int V, uint16_t* wout, float* bout);
bool genmax_applies(int prec, int d, int V);

// ---- autoregressive decoder step with KV cache + CTC prefix scorer (BASELINE config 4)             (ast.hip)
int launch_ast_embed(const int* tok, const float* lut, const float* pe_row, float* x, int n, int d, float scale, hipStream_t s);
int launch_ast_kv_append(int prec, const void* qkv, void* ck, void* cv, int n, int d, int slots, int pos, hipStream_t s);
struct GatherAttnArgs {
    const void* q = nullptr;
    int ldq = 0;
    const void* k = nullptr;
    const void* v = nullptr;
    void* o = nullptr;
    int ldo = 0;
    int n = 0, H = 0, nkeys = 0, slots = 0, d = 0, table_stride = 0;
    const int* anc = nullptr;
    const unsigned char* keyok = nullptr;
    const int* utt = nullptr;
    const unsigned char* keymask = nullptr;
    float scale = 0.125f;
    int append_pos = -1;  // mode 0: >= 0 appends this step's K | V (columns d.. / 2d.. of q's rows) to the cache at that position
};
int launch_ast_gather_attn(int prec, int mode, const GatherAttnArgs& a, hipStream_t s);  // mode 0: cache, 1: source memory
int launch_ast_ctc_prepare(float* logp, const unsigned char* keymask, float* r0, int B, int Tp, int V, int blank, hipStream_t s);
struct CtcPrefixArgs {
    const float* logp;   // [B][Tp][V] masked CTC log-posteriors
    const float* r0;     // [B][Tp][2] initial states
    const float* r_prev; // [*][Tp][2] states kept from the previous step
    float* r_new;        // [n*K][Tp][2]
    const int* utt;      // [n]
    const int* last_tok; // [n]
    const int* cand;     // [n][K]
    const int* prev_ref; // [n]: >= 0 row of r_prev, < 0: initial state of utterance -1-ref
    float* score;        // [n][K]
    int n, K, Tp, V, blank, eos, out_len;
};
int launch_ast_ctc_prefix(const CtcPrefixArgs& a, hipStream_t s);
// device-side beam state of the autoregressive search (double buffered: [cur], [cur ^ 1]); S = B * beam_width slots, L = max length
struct AstBeamState {
    int* tok[2];              // [S][L] hypothesis tokens (sos first)
    int* anc[2];              // [S][L] KV-cache slot that produced each position
    unsigned char* keyok[2];  // [S][L] token != padding_idx
    int* len[2];              // [S]
    double* score[2];         // [S]
    int* valid[2];            // [S]
    int* ctc_ref[2];          // [S] row of the previous step's CTC states (< 0: initial state of utterance -1-ref)
    float* ctc_prev[2];       // [S] CTC prefix score of the hypothesis
    int* cur_tok;             // [S] newest token = the next step's input
    int* utt;                 // [S] utterance of the slot
    int* live;                // [1] live hypotheses after the last update
};
struct AstBeamStep {
    const int* idx;    // [S][K] candidate tokens of this step
    const float* att;  // [S][K] their attention log-probabilities (sorted, best first)
    const float* ctc;  // [S][K] CTC prefix scores (use_ctc)
    int cur, pos, bw, K, L, eos, sos, pad, use_ctc, use_lp;
    float w, u;        // ctc_weight, 1 - ctc_weight as float32
    double lp;
};
int launch_ast_beam_init(const AstBeamState& st, int cur, int B, int bw, int L, int sos, int pad, hipStream_t s);
int launch_ast_beam_update(const AstBeamState& st, const AstBeamStep& q, int B, hipStream_t s);
int launch_logsoftmax_temp(float* logits, int M, int V, int ldl, float temperature, int* arg, float* maxlp, hipStream_t s);
