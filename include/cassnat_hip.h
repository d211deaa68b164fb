/* libcassnat_hip.so - C ABI of the MI355X-native CASS-NAT inference hot path.
 *
 * The reference (balaji1312/cassnat_asr_public) is pure Python/PyTorch and has no FFI of its own; the
 * boundary this library sits behind is the Python call
 *     CassNAT.beam_decode(src, x_mask, src_size, vocab, args, ...)      src/models/cassnat.py:420-637
 * made once per batch by CassNATTask.decode                             src/tasks/cassnat_task.py:326-343
 * on a model built by make_model(input_size, args)                      src/models/cassnat.py:21-89
 * whose parameters are loaded by name from {'model_state': ...}         src/tasks/base_task.py:45-54.
 * The Python shim (cassnat_asr_public_amd/models/cassnat.py) binds these entry points with ctypes.
 *
 * Conventions: every function returns 0 on success, a negative code on failure (cn_last_error() gives the
 * text); nothing throws across the ABI.  Pointers named *_dev are device (HBM) pointers owned by the
 * caller; the library owns weights and workspace.  Calls are ordered on the given hipStream_t (passed as
 * void*, NULL = default stream).  One cn_model per device; a handle is not thread-safe.
 */
#ifndef CASSNAT_HIP_H
#define CASSNAT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cn_model cn_model;

enum { CN_PRECISION_F32 = 0, CN_PRECISION_BF16 = 1, CN_PRECISION_FP8 = 2, CN_PRECISION_BF16X3 = 3, CN_PRECISION_F16 = 4 };
/* CN_PRECISION_F16: the bf16 engine's kernels with IEEE half-precision MFMA operands (11 significant bits instead of 8, same matrix
 * pipe and rate; range +-65504).  It lives in a second build of the same sources, libcassnat_hip_f16.so (-DCN_OP16_F16), which
 * exports this same interface and accepts no other precision; libcassnat_hip.so refuses CN_PRECISION_F16.  cn_operand16() names
 * a library's 16-bit operand ("bf16" / "fp16"). */
enum { CN_DTYPE_F32 = 0, CN_DTYPE_I32 = 1, CN_DTYPE_U8 = 2, CN_DTYPE_F64 = 3 };

/* Model hyper-parameters: the subset of the flat `args` bag that make_model reads for the transformer
 * NAST model (src/models/cassnat.py:41-66). */
typedef struct cn_config {
    int32_t input_size; /* feature dim after splicing (80) */
    int32_t d_model, n_head, d_encff, d_decff;
    int32_t n_enc, n_extra, n_self_dec, n_mix_dec;
    int32_t vocab_size;
    int32_t precision;  /* CN_PRECISION_F32: exact-f32 MFMA (parity gate); CN_PRECISION_BF16: throughput; CN_PRECISION_FP8:
                           the bf16 engine with the encoder layers' products on the e4m3fn MFMA (BASELINE config 5);
                           CN_PRECISION_BF16X3: split-bf16 - every value kept as a bf16 hi + bf16 lo pair (~17 significant
                           bits), every product three bf16 MFMAs (hi.hi + hi.lo + lo.hi, fp32 accumulation): meets the same
                           parity gate as F32 at several times its MFMA rate */
    int32_t max_batch;  /* workspace is sized for max_batch x max_frames */
    int32_t max_frames;
    int32_t device; /* HIP device ordinal */
    int32_t ast;    /* 1: autoregressive (AST) model: n_mix_dec = N_dec decoder layers + tgt_embed (src/models/transformer.py);
                       2: TransformerLM (src/models/lm.py): n_enc layers of width d_encff + text_embed + out_generator */
    /* conformer variants (src/models/cassnat.py:29-57, pos_type "relative"): macaron Swish FFNs, relative-position self
     * attention, convolution module.  conf_enc / conf_dec = args.use_conv_enc / use_conv_dec. */
    int32_t conf_enc, conf_dec;
    int32_t enc_max_rel, dec_max_rel; /* args.enc_max_relative_len / dec_max_relative_len (<= 31) */
    int32_t enc_kernel, dec_kernel;   /* args.enc_kernel_size / dec_kernel_size (odd) */
    int32_t d_ff;                     /* args.d_ff: width of the conformer extractor's FFN */
    int32_t esa_group;                /* ESA: sampled alignments per utterance one cn_esa_sample pass may take (0/1: one);
                                         sizes the decoder-side workspace (max_batch x esa_group query sets) */
    /* CN_PRECISION_FP8 only: which encoder-side products take e4m3 operands (every e4m3 product adds ~5 % of relative
     * noise to its output; the throughput each one buys differs - DESIGN.md 5d has the measured table).  Bits:
     * CN_FP8_CONV2 the second convolution (conv1 then writes its image in e4m3), CN_FP8_LINEAR linear_out (needs CONV2:
     * conv2 hands its rows on in e4m3), CN_FP8_FFN the two feed-forward products of the encoder layers
     * n >= fp8_ffn_first_layer.  0 = all three, every layer. */
    int32_t fp8_scope;
    int32_t fp8_ffn_first_layer;
} cn_config;
#define CN_FP8_CONV2 1
#define CN_FP8_LINEAR 2
#define CN_FP8_FFN 4

/* Decode-time switches read by beam_decode from `args` (src/models/cassnat.py:435-636). */
typedef struct cn_decode_opts {
    int32_t padding_idx; /* also the CTC blank id */
    int32_t sos;
    int32_t left_trigger, right_trigger;
    int32_t src_trigger;
    int32_t use_unimask;
    int32_t beam_width; /* 1: greedy finish on device; 2..16: per-position top-k kept for the host beam */
    int32_t capture;    /* debug: keep an fp32 copy of every stage tensor for cn_fetch */
    int32_t sub_batch;  /* > 0: the call carries B / sub_batch coalesced batches of that many utterances; the greedy finish limits
                           every hypothesis by the row count of its own batch (src/models/cassnat.py:580-637 reads
                           min(ylen + 1, U of the batch) rows), so each batch's hypotheses are what a call of its own gives.
                           Transformer blocks only (a conformer's GroupNorm sees the padded rows) */
    int32_t no_trigger; /* args.use_trigger == False (src/models/cassnat.py:469-473): the extractor attends over every valid frame
                           (trigger_mask = src_mask) and the row counts are best_path_align's own (no EOS row).  0 = use_trigger */
    int32_t reserved[6];
} cn_decode_opts;

const char* cn_last_error(void);
const char* cn_version(void);
const char* cn_operand16(void);

/* replaces models.cassnat.make_model (src/models/cassnat.py:21) */
int cn_model_create(const cn_config* cfg, cn_model** out);
/* A further handle on the SAME device copy of the packed weights as the finalized `donor` (reference counted: the last
 * handle to go frees them): own workspace sized by cfg->max_batch / max_frames / esa_group, ready to decode (no load /
 * finalize).  The model hyper-parameters, device and precision must be the donor's.  What the decode pipelines of one GPU
 * use (one engine per pipeline, one 118 MB blob per GPU - and one RCCL broadcast per rank), and what a handle rebuilt for
 * a larger workspace uses.  No reference counterpart: nn.Module parameters are shared by reference in Python. */
int cn_model_create_shared(const cn_config* cfg, cn_model* donor, cn_model** out);
void cn_model_destroy(cn_model* m);

/* replaces the per-parameter copy of BaseTask.load_test_model (src/tasks/base_task.py:50-54): called once per
 * state-dict entry with its reference name ("encoder.layers.0.self_attn.linears.0.weight", ...). fp32 host data. */
int cn_model_load_weights(cn_model* m, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* sinusoid table shared by src_embed.pos_enc.pe and CassNAT.pe (src/models/cassnat.py:91-99): rows x d_model fp32 */
int cn_model_load_pe(cn_model* m, const float* host_table, int32_t rows);
/* repack into MFMA-friendly layouts / model precision and upload; must follow the last load */
int cn_model_finalize(cn_model* m);
/* rank-0 -> all ranks weight hand-off for the multi-GPU path: the packed device blob that RCCL broadcasts */
int cn_model_weight_blob(cn_model* m, void** dev_ptr, int64_t* bytes);

/* replaces CassNAT.beam_decode for the greedy NAST configuration (use_trigger, sample_num <= 1, no LM).
 *   feats_dev      (B,T,F) fp32 contiguous, padded frames exactly == padding_idx in feature 0
 *   size_ratio_dev (B) fp32 length ratios (SuperviseLoader.collate_fn, src/data/speech_loader.py:354)
 *   hyp_out_dev    (B,hyp_stride) int32: [sos, tok...]; hyp_len_dev (B); score_dev (B) float64
 * All stages through the greedy pack run on `stream`; the call synchronises the stream once (the token
 * count U is data dependent).  Workspace: every buffer is checked against the call before anything is launched. */
int cn_decode_nast(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                   const cn_decode_opts* opts, int32_t* hyp_out_dev, int32_t hyp_stride, int32_t* hyp_len_dev,
                   double* score_dev, void* stream);

/* One engine pass over SEVERAL reference batches, each with its own frame count (the reference collates every batch to its own
 * longest utterance, src/data/speech_loader.py:327-356, and decodes batch after batch, src/tasks/cassnat_task.py:317-356):
 * feats_dev holds the batches one after the other, every utterance padded with padding_idx frames to the call's T = the largest
 * sub_frames; batch k has sub_rows_host[k] utterances of sub_frames_host[k] frames (size_ratio relative to THAT count, as
 * collate_fn computes it).  Every utterance's hypothesis and score are exactly those of a cn_decode_nast call on its own batch:
 * its frames past sub_frames are treated as the convolutions' zero padding, src_size = (ratio * T'_own).long()
 * (src/models/cassnat.py:436), the alignment's shift and the forced EOS frame (:355-365, 378-389) use T'_own, keys past T'_own do
 * not exist for the softmax (a row without any allowed key attends uniformly over T'_own keys, attention.py:19-21), and the
 * greedy finish is limited by the row count of the own batch.  n_sub == 0: a plain call (sub_rows / sub_frames unused).
 * Transformer blocks, beam_width 1, no capture.  At most 64 batches per pass.
 * The workspace is an AREA: a pass fits when every buffer holds it (B x T' rows etc. against max_batch x max_frames), so a pass of
 * short utterances may carry more of them than max_batch (up to 16 x max_batch); the error names the buffer that is too small.
 * u_hint > 0: the decoder side is launched on min(u_hint, T' + 1) rows WITHOUT waiting for the data-dependent row count (the
 * reference's `.item()` sync, src/models/cassnat.py:387): nothing in the call blocks the host.  Results equal the exact call's
 * whenever u_hint >= the true count; *ticket_out names the page-locked word that receives the true count - once the stream has
 * drained, cn_decode_ticket(ticket) returns it beside the rows used, and on rows_used < ymax the caller decodes the pass again
 * (u_hint 0 = exact: the call synchronises the stream once, as cn_decode_nast).  A ticket is the call's sequence number on this
 * handle; its counts live in one of FOUR words, so it stays valid until four further decode calls (cn_decode_nast included) have
 * been started on the handle - after that cn_decode_ticket fails ("expired") instead of returning another pass's counts. */
int cn_decode_nast_merged(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                          const cn_decode_opts* opts, int32_t n_sub, const int32_t* sub_rows_host, const int32_t* sub_frames_host,
                          int32_t u_hint, int32_t* hyp_out_dev, int32_t hyp_stride, int32_t* hyp_len_dev, double* score_dev,
                          void* stream, int32_t* ticket_out);
int cn_decode_ticket(cn_model* m, int32_t ticket, int32_t* ymax_host, int32_t* rows_used_host);

/* CN_PRECISION_F16 engines (libcassnat_hip_f16.so): half-precision operands have a range (+-65504).  What drives magnitudes from
 * outside is the scale of the features - everything behind linear_out is LayerNorm-ed in fp32 first -, so every pass compares them
 * with the largest |feature| for which neither subsampling convolution's output can leave half of that range (from the
 * convolutions' weight row sums; a word of the weight blob) and raises a sticky flag.  *fault = 1: a pass since the last call saw
 * such features - its results are not to be used (decode on a bf16 / bf16x3 engine, or normalise the features).  Valid once the
 * passes' stream work is done; clears the flag.  *feature_limit (may be null): that largest |feature|, 0 if unknown.
 * CN_PRECISION_BF16X3 engines use the same guard: their second convolution runs a mixed arithmetic whose e4m3 cross-term operands
 * hold conv1 outputs up to 448 (csrc/conv2.hip MIX); features beyond (448 - |b1|max) / (largest row sum of |w1|) would let them
 * saturate and the engine fall below its tolerance.  Engines of every other precision: *fault = 0. */
int cn_take_range_fault(cn_model* m, int32_t* fault, float* feature_limit);

/* stage-level entry: src_embed + encoder + ctc_generator + alignment only (src/models/cassnat.py:431-468) */
int cn_encode_align(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                    const cn_decode_opts* opts, int32_t* ymax_host, void* stream);

/* ---- autoregressive (AST) model, BASELINE config 4: device half of Transformer.beam_decode (src/models/transformer.py:122-241).
 * The host keeps the beam bookkeeping (as the reference does in Python); per step it passes, for every live hypothesis (row),
 * its last token, utterance index, ancestor table (slot that wrote each earlier position of its prefix) and the key mask of
 * its prefix (token != padding_idx), and receives the top-K of log_softmax(att logits / T).  Keys/values of earlier positions
 * come from a per-layer cache - the decoder is NOT re-run on the whole prefix as the reference does. */
int cn_ast_begin(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                 int32_t want_ctc, int32_t max_len, int32_t max_slots, int32_t ctc_beam, void* stream);
int cn_ast_step(cn_model* m, int32_t n_live, int32_t pos, const int32_t* tok_dev, const int32_t* utt_dev,
                const int32_t* anc_dev, const uint8_t* keyok_dev, int32_t table_stride, float temperature, int32_t K,
                int32_t* topk_idx_dev, float* topk_val_dev, void* stream);
/* CTCPrefixScore.__call__ (src/utils/ctc_prefix.py:50-106) for K candidate labels of every live hypothesis; the new states
 * of all n_live*K candidates are kept on the device (row h*K+c of the buffer of this step's parity) for the next step. */
int cn_ast_ctc_score(cn_model* m, int32_t n_live, int32_t out_len, const int32_t* utt_dev, const int32_t* last_tok_dev,
                     const int32_t* cand_dev, int32_t K, const int32_t* prev_ref_dev, int32_t parity, int32_t eos,
                     float* score_dev, void* stream);

/* The whole joint CTC / attention beam search of Transformer.beam_decode (src/models/transformer.py:122-241, lm_weight == 0)
 * on the device: no host round trip inside the step loop (the host polls a live-hypothesis counter every 8 steps).
 * hyp_out_dev [B][beam_width][max_len] int32 (sos first, padded with padding_idx), hyp_len_dev [B][beam_width],
 * score_dev [B][beam_width] double; beams best first, same ordering rules as the reference (stable ties). */
typedef struct cn_ast_opts {
    float ctc_weight;            /* > 0: joint scoring with the CTC prefix scorer over ctc_beam candidates */
    float temperature;           /* args.T */
    int32_t ctc_beam;
    int32_t beam_width;
    int32_t max_step;            /* int(max_decode_ratio * T') or T' */
    int32_t eos;
    int32_t use_length_penalty;  /* 0: args.length_penalty is None */
    float one_minus_ctc_weight;  /* float32(1 - ctc_weight) as the reference computes it (in double, then cast) */
    double length_penalty;
    int32_t reserved[4];
} cn_ast_opts;
int cn_decode_ast(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                  const cn_ast_opts* ast_opts, int32_t* hyp_out_dev, int32_t max_len, int32_t* hyp_len_dev, double* score_dev,
                  void* stream);

/* ---- ESA: error-based sampling of alignments + TransformerLM ranking (src/models/cassnat.py:370-376, 441-445, 499-561) --
 * cn_esa_begin: encoder + CTC generator once; the two best labels of every frame are kept.
 * cn_esa_sample: n_samples (<= cfg.esa_group) sampled alignments per utterance in one pass: in alignment g, frame t of
 * utterance b takes the second-best label iff select[g][b][t] != 0 and the best label's probability < threshold (all-zero
 * draws - or select_dev == NULL with n_samples == 1 - give the best path); then alignment -> extractor -> decoder ->
 * generator over B * n_samples query sets that share the B utterances' encoder outputs: tok_out / val_out
 * [n_samples][B][out_stride] = argmax token and its log-probability per decoder row, ylen_out [n_samples][B] (EOS row
 * included), *ymax_host = rows of this pass.  force_U: 0 = decode on this pass's own row count; > 0 = on that many rows
 * (conformer blocks: GroupNorm sees an utterance's padded rows, so every group must use the row count of ALL samples, as the
 * reference's single batch does); -1 = count only (outputs may be NULL).  The caller owns the random draws (the reference takes them from
 * torch.randint) and loops over groups of samples.  opts->beam_width must be 1. */
int cn_esa_begin(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts, void* stream);
int cn_esa_sample(cn_model* m, const uint8_t* select_dev, int32_t n_samples, float threshold, const float* size_ratio_dev,
                  const cn_decode_opts* opts, int32_t* tok_out_dev, float* val_out_dev, int32_t out_stride,
                  int32_t* ylen_out_dev, int32_t* ymax_host, int32_t force_U, void* stream);
/* TransformerLM (src/models/lm.py; model created with cfg.ast = 2: n_enc layers of width d_encff, parameters
 * text_embed.0.lut / encoder.* / out_generator.proj): score[b][u] = log p(tgt[b][u] | tok[b][0..u]) with key j allowed
 * iff j <= u and j < len[b].  tok / tgt / score are [B][ld] with ld >= U. */
int cn_lm_score(cn_model* m, const int32_t* tok_dev, const int32_t* tgt_dev, const int32_t* len_dev, int32_t B, int32_t U,
                int32_t ld, float* score_dev, void* stream);

/* ---- decode_type ctc_only / ctc_att (src/tasks/cassnat_task.py:335-341) ------------------------------------------------
 * cn_ctc_beam replaces utils.beam_decode.ctc_beam_decode (src/utils/beam_decode.py:8-93) without a language model: encoder,
 * CTC generator, the `pruning` best labels per frame, then the prefix beam search over the frames (frames past src_size and
 * frames with blank probability > 0.95 skipped, candidates not merged by prefix, stable sort by score_ctc + length_penalty *
 * len(hyp), float64 scores) on the device.  hyp_out_dev [B][beam][hyp_cap] labels (no sos; hyp_cap >= T' + 1), hyp_len_dev /
 * score_dev (score_ctc) / p_blk_dev / p_nblk_dev [B][beam], nbeam_dev [B] hypotheses kept, best first. */
int cn_ctc_beam(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                const cn_decode_opts* opts, int32_t beam, int32_t pruning, double length_penalty, int32_t* hyp_out_dev,
                int32_t hyp_cap, int32_t* hyp_len_dev, double* score_dev, double* p_blk_dev, double* p_nblk_dev, int32_t* nbeam_dev,
                void* stream);
/* CassNAT.beam_decode for decode_type 'ctc_att' with sample_num 1 (src/models/cassnat.py:446-448): the trigger mask comes from
 * the forced (Viterbi) alignment of labels_dev [B][ld] / label_len_dev [B] (beam_path_align -> viterbi_align, :391-414,
 * 272-353) instead of the greedy path; max_label_len = the largest label_len (the width of the reference's label tensor).
 * Outputs as cn_decode_nast. */
int cn_decode_nast_forced(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                          const cn_decode_opts* opts, const int32_t* labels_dev, const int32_t* label_len_dev, int32_t ld,
                          int32_t max_label_len, int32_t* hyp_out_dev, int32_t hyp_stride, int32_t* hyp_len_dev, double* score_dev,
                          void* stream);
/* ESA ranking with rank_model 'at_baseline' (src/models/cassnat.py:514-520, Transformer.forward_decoder): the autoregressive
 * model (cfg.ast = 1, cfg.esa_group >= n_per_utt) scores token rows teacher-forced: its encoder on the B utterances, then the
 * decoder on N = B * n_per_utt rows (row e belongs to utterance e % B) under the causal + length mask:
 * score[e][u] = log softmax(att_generator(dec_h[e][u]))[tgt[e][u]].  tok / tgt / score [N][ld], ld >= U. */
int cn_ast_teacher_score(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                         const int32_t* tok_dev, const int32_t* tgt_dev, const int32_t* len_dev, int32_t n_per_utt, int32_t U,
                         int32_t ld, float* score_dev, void* stream);

/* ArtTask decode_type 'ctc_correct' = Transformer.fast_decode_with_ctc (src/models/transformer.py:243-342, called at
 * src/tasks/art_task.py:254-255): the CTC greedy hypothesis (arg-max path zeroed on masked frames, repeats collapsed, blanks dropped)
 * behind sos is the teacher-forced decoder input under the causal + padding mask; returned are the hypothesis lengths len_out_dev [B]
 * and, for the U = longest + 1 decoder rows (*rows_host), the k best labels and log-probabilities of every row, tok_out_dev /
 * val_out_dev [B][U][k] (buffers of B * (T' + 1) * k entries): what the reference's finish loop (:276-341) consumes on the host. */
int cn_ast_ctc_correct(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts, int32_t k,
                       int32_t* tok_out_dev, float* val_out_dev, int32_t* len_out_dev, int32_t* rows_host, void* stream);

/* Copy a named internal / captured tensor to the host (synchronous; test + host-beam use).  Activations are
 * returned as fp32 whatever the model precision.  shape_out has room for 4 dims. */
int cn_fetch(cn_model* m, const char* name, void* host_dst, int64_t max_bytes, int64_t* shape_out, int32_t* ndim_out,
             int32_t* dtype_out);

/* Per-kernel timing with HIP events recorded on the launch stream around each tagged kernel.
 * tags: '|'-separated list ("conv2|self_attention") or NULL for every tag.  cn_profile_end synchronises the device and
 * writes {"tag": {"count": n, "ms": total, "flops": algorithmic flops, "bytes": algorithmic bytes}, ...}. */
int cn_profile_begin(cn_model* m, const char* tags);
int cn_profile_end(cn_model* m, char* json_out, int64_t cap);

/* Row-chain kernel (bf16, d_model 256), one launch for the non-attention half of a layer on 128-row blocks:
 * x += Wo.ctx + bo (skipped when ctx_dev is NULL); x += W2.relu(W1.LN1(x)+b1)+b2 (skipped when dff == 0);
 * when nln_a_host != NULL: out = Wt.LNn(x)+bt (tail_n columns) or out = LNn(x) itself (tail_n == 0), bf16 [M][ldo].
 * Replaces linears[3] + SublayerConnection + PositionwiseFeedForward + LayerNorm + linears[0..2]
 * (src/models/modules/attention.py:44-66, utils.py:23-32, positionff.py:15-16, norm.py:15-18).  Weights are host fp32
 * in nn.Linear layout and are packed on every call: a test entry point, the model keeps its packed copies.
 * x_mode bits: 1 = x_dev is read in the kernel's blocked layout, 2 = written in it, 4 = not written back, 8 = the feed-forward
 * activation is Swish (x * sigmoid(x), the conformer's macaron halves) instead of ReLU, 16 = the tail projection out_dev is
 * written as a blocked bf16 matrix of ldo columns (ceil(M / 32) * 32 rows; 32 x 32 tiles of 2 KiB, each [16-column half][lane =
 * 32 * (bit 3 of the column) + row % 32][8 bf16]: what the attention kernel reads as `blocked` Q / K / V).  Blocked x: 32-row
 * blocks of [32 pieces i][64 lanes][4 floats], lane = row % 32 + 32 h holding channels 32 (i / 4) + 8 (i % 4) + 4 h + (0..3);
 * the buffer then holds ceil(M / 32) * 32 rows.  32 = the two feed-forward products take e4m3fn operands (BASELINE config 5,
 * CN_PRECISION_FP8 engines; d_ff % 256 == 0, ReLU only): LN1(x) at x16 and the ReLU output at x8, both saturating at +-448,
 * W1 and W2 each at the largest power-of-two scale that keeps its largest magnitude <= 448; accumulation, bias, residual and
 * everything else of the launch as without the bit. */
int cn_op_chain(float* x_dev, const void* ctx_dev, int32_t ldctx, const float* wo_host, const float* bo_host,
                const float* ln1_a_host, const float* ln1_b_host, const float* w1_host, const float* b1_host,
                const float* w2_host, const float* b2_host, const float* nln_a_host, const float* nln_b_host,
                const float* wt_host, const float* bt_host, void* out_dev, int32_t ldo, int32_t M, int32_t dff,
                int32_t tail_n, float eps, int32_t x_mode, void* stream);

/* ---- front-end: waveform -> log-mel filterbank features (+ global CMVN), padded batch out ----------------------
 * What the reference leaves to Kaldi's compute-fbank-feats (egs/librispeech/conf/fbank.conf:1-6: hamming window, 16 kHz,
 * 80 mel bins, no energy; other options at Kaldi's defaults, dither = 0) followed by the (feat - mean) / std of
 * SpeechDataset._load_cmvn (src/data/speech_loader.py:109-115).  wave_dev: [B][max_samples] float32 on the int16 scale
 * (as Kaldi reads a wav); num_samples_dev: [B]; feats_dev: [B][Tmax][num_mel] float32, frame t of utterance b exists for
 * t < 1 + (num_samples[b] - frame_len) / frame_shift (snip_edges), later frames are filled with pad_value (the
 * decoder's padding_idx, so that its mask derivation feats[:,:,0] != padding_idx sees them as padding).
 * cmvn_mean_dev / cmvn_istd_dev: [num_mel] or NULL.  The output is directly cn_decode_nast's feats_dev. */
typedef struct cn_fbank_opts {
    float sample_rate, frame_length_ms, frame_shift_ms, preemph, low_freq, high_freq;
    int32_t num_mel, window_type /* 0 hamming, 1 povey, 2 hanning, 3 rectangular */, remove_dc, use_power, use_log;
    int32_t reserved[5];
} cn_fbank_opts;
void cn_fbank_default_opts(cn_fbank_opts* o);
int32_t cn_fbank_num_frames(const cn_fbank_opts* o, int32_t num_samples);
int cn_fbank(const cn_fbank_opts* o, const float* wave_dev, const int32_t* num_samples_dev, int32_t B, int32_t max_samples,
             const float* cmvn_mean_dev, const float* cmvn_istd_dev, float* feats_dev, int32_t Tmax, float pad_value,
             void* stream);

/* ---- single-kernel entry points (parity tests drive each hand-written kernel through the ABI) ---------- */
/* all pointers device; `precision` selects the element type of activations/weights: CN_PRECISION_F32, CN_PRECISION_BF16 or
 * CN_PRECISION_BF16X3 (split-bf16 elements: each group of 32 consecutive elements of a row is 128 bytes, 32 bf16 hi halves
 * then 32 bf16 lo halves; rows and row strides are multiples of 32 elements) */
/* fp32 <-> a flat tensor of n elements in the given precision's element type (to_f32 = 0: src fp32 -> dst; 1: back) */
int cn_op_convert(int32_t precision, const void* src, void* dst, int64_t n, int32_t to_f32, void* stream);
int cn_op_gemm(int32_t precision, const void* A, int32_t lda, const void* W, const float* bias, void* C, int32_t ldc,
               int32_t c_is_f32, int32_t M, int32_t N, int32_t K, int32_t relu, const float* resid, int32_t ldr,
               const float* pe, int32_t pe_period, float scale, void* stream);
int cn_op_conv1(int32_t precision, const float* x, const float* w9c, const float* bias, void* out, int32_t B, int32_t T,
                int32_t F, int32_t C, void* stream);
/* the bf16 engine's form of the same layer (src/models/modules/embedding.py:102-104): conv1 + ReLU written as the bordered bf16
 * image [B][T1+2][F1+2][C] (zero border) that the second convolution's tile kernel reads; computed on the matrix cores from
 * split-bf16 operands (16 significant bits in front of the bf16 rounding).  C == 256, (F-1)/2 + 3 >= 32. */
int cn_op_conv1_bordered(const float* x, const float* w9c, const float* bias, void* out, int32_t B, int32_t T, int32_t F,
                         int32_t C, void* stream);
/* fp8 engine's conv front-end (BASELINE config 5, src/models/modules/embedding.py:102-108 on e4m3fn operands): conv1 + ReLU written as
 * an e4m3fn image at img_scale (a power of two; saturating), conv2 + ReLU from it with w2 (HOST fp32, [C][3][3][C]: k = (kh*3+kw)*C + ci)
 * quantised at the largest power-of-two scale that keeps max|w| <= 448 (returned in *w_scale_out); out bf16 [B*T2*F2][C].  C == 256.
 * img8_out_dev (optional): the image as the second kernel reads it, [B][T1+2][F1+2][C] bytes with its border of zeros. */
int cn_op_conv_frontend_fp8(const float* x_dev, const float* w1_9c_dev, const float* b1_dev, const float* w2_host,
                            const float* b2_dev, void* out_dev, void* img8_out_dev, int32_t B, int32_t T, int32_t F, int32_t C,
                            float img_scale, float out8_scale, float* w_scale_out, void* stream);

/* The split-bf16 engine's conv front-end in its MIX arithmetic (csrc/conv2.hip): a product is half(a) half(b) + l_a q_b + q_a l_b
 * with l = e4m3((v - half(v)) S_l) and q = e4m3(v S_q) at fixed power-of-two scales - one half-precision MFMA plus two e4m3 MFMAs
 * at twice the rate where the split-bf16 form spends three bf16 MFMAs.  x fp32 [B][T][F] (device), w1 [9][C] / b1 / b2 device,
 * w2_host fp32 [C][3][3][C] (k = (kh * 3 + kw) * C + ci); out: split-bf16 rows [B * T2 * F2][C] (device).  C = 256. */
int cn_op_conv_frontend_mix(const float* x, const float* w1_9c, const float* b1, const float* w2_host, const float* b2, void* out,
                            void* img_out /* optional: conv1's planes, 4 bytes per bordered cell */, int32_t B, int32_t T, int32_t F,
                            int32_t C, void* stream);
/* (out8_scale > 0: out_dev receives e4m3fn bytes [B*T2*F2][C] at that scale instead of bf16 - the input of the next entry)
 * linear_out (src/models/modules/embedding.py:118-119) of the fp8 engine: a8_dev [M][K] e4m3fn at a_scale, K = 5120; w_host fp32
 * [256][K] quantised at the largest power-of-two scale in range (*w_scale_out); out fp32 [M][256] =
 * (a . w^T + bias) * out_scale + pe[m % pe_period] (pe_dev may be NULL). */
int cn_op_linear256_fp8(const void* a8_dev, const float* w_host, const float* bias_dev, float* out_dev, int32_t M, int32_t K,
                        float a_scale, float out_scale, const float* pe_dev, int32_t pe_period, float* w_scale_out, void* stream);
int cn_op_conv2(int32_t precision, const void* conv1_out, const void* w_khwc, const float* bias, void* out, int32_t B,
                int32_t T1, int32_t F1, int32_t C, void* stream);
int cn_op_layernorm(int32_t precision, const float* x, const float* a2, const float* b2, void* y, int32_t M, int32_t d,
                    float eps, void* stream);
int cn_op_attention(int32_t precision, const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V,
                    int32_t ldv, void* O, int32_t ldo, int32_t B, int32_t H, int32_t Lq, int32_t Lk,
                    const uint8_t* keymask, const int32_t* klen, const int32_t* intervals, int32_t iv_stride,
                    int32_t causal, float scale, void* stream);
int cn_op_logsoftmax_argmax(float* logits, int32_t M, int32_t V, int32_t* arg, float* maxlp, int32_t write_logp,
                            void* stream);
int cn_op_ctc_align(const int32_t* best, const uint8_t* keymask, const float* size_ratio, int32_t B, int32_t Tp,
                    int32_t blank, int32_t left, int32_t right, int32_t* shift, int32_t* src_size, int32_t* ylen,
                    int32_t* ymax, int32_t* intervals, void* stream);
/* CTC prefix beam search (src/utils/beam_decode.py:8-93, no LM) and forced alignment (src/models/cassnat.py:272-345: the label of
 * the aligned state per frame, before the collapse / shift that cn_op_ctc_align's kernel applies) on given log-posteriors
 * logp [B][Tp][V]; buffers as cn_ctc_beam / cn_decode_nast_forced */
int cn_op_ctc_prefix_beam(const float* logp, const float* size_ratio, int32_t B, int32_t Tp, int32_t V, int32_t beam, int32_t pruning,
                          double length_penalty, int32_t blank, int32_t* hyp, int32_t hyp_cap, int32_t* hyp_len, double* score,
                          double* p_blk, double* p_nblk, int32_t* nbeam, void* stream);
int cn_op_ctc_viterbi(const float* logp, const uint8_t* keymask, const float* size_ratio, const int32_t* labels,
                      const int32_t* label_len, int32_t B, int32_t Tp, int32_t V, int32_t ld, int32_t ymax, int32_t blank,
                      int32_t* out_path, void* stream);
int cn_op_greedy_pack(const int32_t* tok, const float* val, const int32_t* ylen, int32_t B, int32_t U, int32_t sos,
                      int32_t hyp_stride, int32_t* hyp, int32_t* hyp_len, double* score, void* stream);
/* fused LN -> W1 -> ReLU -> W2 -> residual (-> next LN) sublayer, bf16 / d_model 256.  x_dev fp32 [M][256] is updated
 * in place; weights are HOST fp32 nn.Linear matrices (w1 [dff][256], w2 [256][dff]) packed and uploaded by the call
 * (test entry: the model packs once at cn_model_finalize).  xn_out_dev (bf16 [M][256]) may be NULL.  nslice > 1: the
 * hidden units are split over nslice workgroups per row tile and a second kernel adds the slices (the form the
 * autoregressive decode step uses for its few rows); d_ff % (128 * nslice) == 0. */
int cn_op_ffn_fused(float* x_dev, const float* ln_a_dev, const float* ln_b_dev, const float* w1_host,
                    const float* b1_dev, const float* w2_host, const float* b2_dev, const float* nln_a_dev,
                    const float* nln_b_dev, void* xn_out_dev, int32_t M, int32_t dff, float eps, int32_t nslice,
                    void* stream);
/* the same sublayer in the split-bf16 precision (CN_PRECISION_BF16X3; fused_x3.hip): three MFMAs per product on hi + lo
 * operands; xn_out_dev is split-bf16 [M][256] (or NULL).  mix != 0: the sublayer's two products in the engine's mixed arithmetic
 * instead (what the engine runs) - half(a) half(b) + l_a q_b + q_a l_b with e4m3 l = (v - half(v)) 2^11 S and q = v S: one
 * half-precision MFMA per 16 k and one K = 64 e4m3 MFMA per 32 k, 2 MFMA units per product */
int cn_op_ffn_x3(float* x_dev, const float* ln_a_dev, const float* ln_b_dev, const float* w1_host, const float* b1_dev,
                 const float* w2_host, const float* b2_dev, const float* nln_a_dev, const float* nln_b_dev, void* xn_out_dev,
                 int32_t M, int32_t dff, float eps, int32_t mix, void* stream);
/* the row-chain form of the split-bf16 engine (fused_x3.hip): attention output projection + residual, feed-forward sublayer, next
 * LayerNorm and (wt_host != null) the next attention's projection of it, in one launch; ctx_dev split-bf16 [M][256] or null;
 * weight matrices on the host, vectors on the device (positionff.py:15-16, attention.py:57-66, norm.py:15-18) */
int cn_op_x3_chain(float* x_dev, const void* ctx_dev, const float* wo_host, const float* bo_dev, const float* ln_a_dev,
                   const float* ln_b_dev, const float* w1_host, const float* b1_dev, const float* w2_host, const float* b2_dev,
                   const float* nln_a_dev, const float* nln_b_dev, void* xn_out_dev, const float* wt_host, const float* bt_dev,
                   void* tail_out_dev, int32_t tail_n, int32_t M, int32_t dff, float eps, int32_t mix /* as cn_op_ffn_x3 */,
                   void* stream);
/* fused generator tail, bf16 / d_model 256: arg[m] = argmax_v, maxlp[m] = max_v of log_softmax(W h[m] + b); h_dev bf16 [M][256],
 * W/b HOST fp32 nn.Linear parameters (packed and uploaded by the call; the model packs once at cn_model_finalize). */
int cn_op_genmax(const void* h_dev, const float* w_host, const float* b_host, int32_t M, int32_t V, int32_t* arg_dev,
                 float* maxlp_dev, void* stream);
/* the same kernel with a target gather (TransformerLM scoring, src/models/cassnat.py:507-520): h_dev bf16 [B * U][256];
 * tgt_lp[b * ld + u] = log_softmax(W h[b * U + u] + b)[tgt[b * ld + u]] */
int cn_op_genmax_gather(const void* h_dev, const float* w_host, const float* b_host, int32_t B, int32_t U, int32_t V,
                        const int32_t* tgt_dev, int32_t ld, float* tgt_lp_dev, void* stream);
/* the generator tail in the split-bf16 precision (CN_PRECISION_BF16X3; three MFMAs per product): h_host fp32 [M][256] (split
 * and uploaded by the call).  tgt_dev == NULL: arg-max (and maxlp when maxlp_dev != NULL); else M = B * U rows and
 * tgt_lp[b * ld + u] = log_softmax(W h[b * U + u] + b)[tgt[b * ld + u]] */
int cn_op_genmax_x3(const float* h_host, const float* w_host, const float* b_host, int32_t M, int32_t V, int32_t* arg_dev,
                    float* maxlp_dev, const int32_t* tgt_dev, int32_t U, int32_t ld, float* tgt_lp_dev, void* stream);
/* d_model-deep projection of the split-bf16 engine (proj_x3.hip; the nn.Linear layers around the attention kernel,
 * src/models/modules/attention.py:57-66): a_host fp32 [M][256], w_host fp32 [N][256] (split into hi + lo halves, packed and
 * uploaded by the call), bias_host [N]; N a multiple of 32, at most 1024.  split_out == 0: c_dev fp32 [M][N] = A . W^T + bias,
 * or resid + resid_scale * (...) when resid_dev (fp32 [M][N], may alias c_dev);  split_out == 1: c_dev receives split-bf16
 * rows (M * N * 4 bytes: per 32 columns 64 bytes of bf16 hi halves, then 64 bytes of lo halves) */
int cn_op_proj_x3(const float* a_host, const float* w_host, const float* bias_host, const float* resid_dev, float resid_scale,
                  void* c_dev, int32_t M, int32_t N, int32_t split_out, void* stream);
int cn_op_topk(const float* logp, int32_t M, int32_t V, int32_t k, int32_t* idx, float* val, void* stream);
/* e4m3fn product (BASELINE config 5): A bf16 [M][lda] on the device is quantised at a_scale (saturating at 448 / a_scale),
 * W = HOST fp32 [N][K] at the largest power-of-two scale that fits (returned in *w_scale_out), as cn_model_finalize does for
 * the encoder layers of a CN_PRECISION_FP8 model; C fp32 [M][N] = relu?(A_q . W_q^T / (a_scale * w_scale) + bias) */
int cn_op_gemm_fp8(const void* a_bf16_dev, int32_t lda, const float* w_host, const float* bias_dev, float* c_dev, int32_t M,
                   int32_t N, int32_t K, float a_scale, int32_t relu, float* w_scale_out, void* stream);
/* bf16 [M][ld] -> e4m3fn bytes [M][K] at `scale` (round to nearest even, saturating at +-448): the activation quantiser */
/* Global CMVN of a padded (B, T, F) float32 batch in place: frames t < len[b] become float((double(x) - mean[f]) / std[f]) - the
 * reference's SpeechDataset arithmetic (src/data/speech_loader.py:109-115, 147-149: numpy float64 with float64 statistics, rounded to
 * float32 at collate) bit for bit; later frames (padding) are left alone.  mean / std: F doubles on the device. */
int cn_op_cmvn(float* feats_dev, const int32_t* len_dev, const double* mean_dev, const double* std_dev, int32_t B, int32_t T, int32_t F,
               void* stream);
/* The reader's collate on the device (SuperviseLoader.collate_fn, src/data/speech_loader.py:327-356; global CMVN :109-115, 147-149):
 * the utterances of an engine pass arrive packed - archive rows back to back, utterance r at row off_dev[r], len_dev[r] frames - and
 * are spread over the padded batch out_dev (rows, T, F): frames past an utterance's length are `pad`; with statistics (float64, [F])
 * a frame becomes float((double(x) - mean) / std), the dataset's arithmetic bit for bit. */
int cn_op_unpack_rows(const float* packed_dev, const int32_t* off_dev, const int32_t* len_dev, float* out_dev, int32_t rows, int32_t T,
                      int32_t F, float pad, const double* mean_dev, const double* std_dev, void* stream);
/* host side of the same reader: n byte ranges (an utterance's rows in the memory map of its archive) copied back to back into a
 * staging buffer (dst + dst_offsets[i]) by one GIL-free call; threads > 1 deals them over that many host threads */
int cn_host_gather(void* dst, const uint64_t* src_ptrs, const uint64_t* dst_offsets, const uint64_t* nbytes, int32_t n, int32_t threads);
int cn_op_quantize_fp8(const void* src_bf16_dev, int32_t ld, void* dst_dev, int32_t M, int32_t K, float scale, void* stream);
/* generator tail of the autoregressive step (src/models/transformer.py:48-51, 199-200): log_softmax(logits / T) and its per-row
 * top-k (sorted descending, ties: lower index) in one pass; the logits [M][V] are left untouched */
int cn_op_logsoftmax_topk(const float* logits, int32_t M, int32_t V, float temperature, int32_t k, int32_t* idx, float* val,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif
