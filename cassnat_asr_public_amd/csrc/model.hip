// C-ABI of libcassnat_hip.so: model handle, weight packing, workspace and the stream-ordered decode
// pipeline that replaces CassNAT.beam_decode (reference: src/models/cassnat.py:420-637) for the greedy
// NAST configuration.  See include/cassnat_hip.h for the contract.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cassnat_hip.h"
#include "kernels.h"

static thread_local std::string g_last_error;
void cn_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* cn_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* cn_version(void) { return "cassnat_hip 0.1 (gfx950, 16-bit operand " CN_OP16_NAME ")"; }
extern "C" const char* cn_operand16(void) { return CN_OP16_NAME; }

// The library exists in two builds that differ in the 16-bit MFMA operand (common.h): this one's engines of the 16-bit kind are
// CN_PRECISION_BF16 (+ FP8, BF16X3, F32) - or, built with -DCN_OP16_F16, CN_PRECISION_F16 and nothing else.  Returns the internal
// precision of a request, -1 (error set) when it belongs to the other build.
static int cn_own_precision(int32_t precision, const char* who) {
#ifdef CN_OP16_F16
    if (precision == CN_PRECISION_F16) return CN_PREC_BF16;  // (the 16-bit path; its operand type is this build's)
    cn_set_error(std::string(who) + ": this build of the library (libcassnat_hip_f16.so) holds the fp16 engine only (CN_PRECISION_F16)");
    return -1;
#else
    if (precision == CN_PRECISION_F16) {
        cn_set_error(std::string(who) + ": CN_PRECISION_F16 engines live in libcassnat_hip_f16.so (the -DCN_OP16_F16 build of these sources)");
        return -1;
    }
    return precision;
#endif
}

#define CN_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != 0) return _rc;   \
    } while (0)

namespace {

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct Linear {
    void* W = nullptr;  // [N][K] model precision
    float* b = nullptr;
    int N = 0, K = 0;
    void* W8 = nullptr;          // fp8 encoder mode (config 5): e4m3fn copy [N][K] at a per-tensor power-of-two scale ...
    float* w8_inv = nullptr;     // ... and 1 / scale (one float in the blob: it travels with a weight broadcast)
    void* px3 = nullptr;    // split-bf16 engine, K == 256: pack_proj_x3 fragment stream (proj_x3.hip)
    void* gm_w = nullptr;   // generators only (bf16, d_model 256): pack_genmax fragment stream ...
    float* gm_b = nullptr;  // ... and padded biases for the fused argmax kernel
};
struct Norm {
    float* a = nullptr;
    float* b = nullptr;
};
// conformer convolution module (conformer_related.py:15-44)
struct ConvMod {
    Linear pw1, pw2;            // pointwise convs as [2d][d] / [d][d] matrices
    float *dw_w = nullptr, *dw_b = nullptr;  // depthwise [d][k] fp32, bias [d]
    float *gn_w = nullptr, *gn_b = nullptr;  // GroupNorm(1, d) affine
    int k = 0;
};
// fp8 engine: scale of the e4m3 image conv1 writes for conv2's e4m3 form (ReLU outputs: x8, saturating at 56 - as the hidden
// activations of the feed-forward products)
constexpr float FP8_S_IMG = 8.f;
// conv2's MIX form (conv2.hip): fixed power-of-two scales of the image's e4m3 planes - q = e4m3(v 2^MIX_LG_AQ) holds conv1 outputs
// up to 448 (beyond it the cross term a_q w_l saturates: the product falls back towards half precision, not apart) and
// l = e4m3((v - half(v)) 2^MIX_LG_AL), |v - half(v)| <= 2^-11 |v|
constexpr int MIX_LG_AQ = 0, MIX_LG_AL = 11;
struct ChainRef {
    void* w = nullptr;
    float* tab = nullptr;
    int dff = 0, tail_n = 0;
    bool has_wo = false, has_next = false, swish = false;
    int* f8q = nullptr;  // e4m3 feed-forward units (chain.hip, F8 form): the four scale bytes of its products, in the blob
};
struct Layer {
    Norm n[5];
    // conformer blocks: w1/w2 are feed_forward1 (or the extractor's feed_forward), ff2_* feed_forward2; relative-position
    // self attention keeps the projected position rows P = table . W_pos^T [(2R+1)][d] and the biases u, v [d] in fp32
    Linear ff2_w1, ff2_w2;
    ConvMod conv;
    float *pos_proj = nullptr, *pos_u = nullptr, *pos_v = nullptr;
    int rel_R = 0;
    bool conformer = false;
    ChainRef cf_a, cf_b, cf_c, cf_d;  // conformer layer on the row-chain kernel (bf16 / d_model 256): see conf_layer_chains
    int kv_slot = -1;  // >= 0: this layer's cross-attention K|V are columns kv_slot * 2d.. of m->kv_all (written by the last encoder chain)
    Linear qkv;       // self-attention: fused Q|K|V projection
    Linear self_o;
    Linear src_q, src_kv, src_o;  // source attention: Q from the stream, K|V from the encoder memory
    Linear w1, w2;
    void *w1p = nullptr, *w2p = nullptr;  // fused-FFN fragment streams (bf16, d_model == 256); then w1/w2 hold biases only
    void* wx3 = nullptr;  // split-bf16 engine: the fused-FFN stream of fused_x3.hip (hi and lo fragments); w1/w2 hold biases only
    bool has_self = false, has_src = false;
};

// One packed stream of the row-chain kernel (chain.hip): [attention output projection] + [FFN] + [the NEXT sublayer's
// pre-norm and input projection: tail_n columns, 0 = the norm itself is the output]
// decoder-side sublayers in execution order (extractor: src; SAD: self; MAD: self, src), each followed by its chain
struct DecStep {
    int stack = 0;       // 0 extractor, 1 SAD, 2 MAD
    int layer = 0;
    bool self = false;   // self attention (else source attention)
    ChainRef chain;      // after this sublayer's attention
    ChainRef entry;      // pre-norm + input projection alone (first step; first MAD step when use_unimask shifts the stream)
};

struct ProfPending {
    const char* tag;
    hipEvent_t a, b;
    double flops, bytes;
};
struct ProfStat {
    long long count = 0;
    double ms = 0, flops = 0, bytes = 0;
};

struct Capture {
    void* p = nullptr;
    size_t bytes = 0;
    int dtype = CN_DTYPE_F32;
    std::vector<int64_t> shape;
    long long call = -1;  // the engine call (cn_model::call_id) that produced it
};

inline uint16_t f32_to_bf16_host(float f) { return cn_host_op16(f); }  // (the engine's 16-bit operand: common.h)

}  // namespace

struct cn_model {
    cn_config cfg;
    int prec = 0;
    int fp8_scope = 0;     // CN_FP8_* bits in force (cn_config.fp8_scope, 0 resolved to all)
    bool fp8_enc = false;  // CN_PRECISION_FP8: bf16 engine whose encoder-layer products run on the fp8 MFMA (BASELINE config 5)
    size_t es = 4;
    std::map<std::string, HostTensor> host;
    std::vector<float> pe_host;
    int pe_rows = 0;
    bool finalized = false;

    // packed weights: owned through a reference count, so that several handles (the decode pipelines of one GPU, or a
    // handle rebuilt for a larger workspace) use ONE device copy (cn_model_create_shared)
    struct BlobOwner {
        unsigned char* p = nullptr;
        ~BlobOwner() {
            if (p) (void)hipFree(p);
        }
    };
    std::shared_ptr<BlobOwner> blob_owner;
    bool blob_borrowed = false;  // the blob belongs to a donor handle: never re-packed or re-allocated here
    unsigned char* blob = nullptr;
    size_t blob_bytes = 0;

    // weight views into the blob
    float* conv1_w = nullptr;  // [9][C]
    float* conv1_b = nullptr;
    Linear conv2;       // [C][9C] (kh,kw,cin)
    void* conv2_f8w = nullptr;  // fp8 engine, 256 channels: the same matrix as e4m3fn bytes at a per-tensor power-of-two scale, and
    int* conv2_f8q = nullptr;   // the two E8M0 scale bytes of conv2.hip's e4m3 form {127 - log2(weight scale), 127 - log2(image scale)}
    void* linear_f8w = nullptr;  // fp8 engine: linear_out's (column-permuted) matrix as e4m3fn bytes, and its two scale bytes
    int* linear_f8q = nullptr;   // {127 - log2(weight scale), 127 - log2(scale of conv2's e4m3 output)}
    void* conv2_x3w = nullptr;  // split-bf16 engine, 256 channels: the same matrix as two bf16 planes (hi, then lo) for conv2.hip's X3 form
    // ... and for conv2.hip's MIX form (half-precision hi x hi + e4m3 cross terms: 2 MFMA units per product instead of 3): the
    // half-precision matrix, then the q = e4m3(w S_q) and l = e4m3((w - hi) S_l) bytes; conv2_mixq = the four E8M0 scale bytes
    void* conv2_mixw = nullptr;
    int* conv2_mixq = nullptr;
    Linear linear_out;  // [d][F2*C] (f,c)
    std::vector<Layer> enc, extra, sad, mad;
    std::vector<ChainRef> enc_chain;  // one per encoder layer when the row-chain path applies, else empty
    ChainRef enc_entry;               // LayerNorm + Q|K|V projection of encoder layer 0 (the rows come from linear_out)
    std::vector<DecStep> dec_steps;   // decoder-side sublayers with their chains when the path applies, else empty
    Norm enc_norm, dec_norm;
    Linear ctc_gen, att_gen;
    float* pe = nullptr;  // [pe_rows][d]

    // geometry of the workspace
    int maxB = 0, maxT = 0, maxT1 = 0, maxTp = 0, F1 = 0, F2 = 0;
    int maxU = 0;  // utterances a call may carry at most (buffers with one entry per utterance are sized for it): a merged pass
                   // of short utterances fits more of them than max_batch into the workspace's max_batch x max_frames area
    std::vector<void*> allocs;
    std::map<std::string, size_t> ws_cap;  // bytes allocated per workspace buffer: every call is checked against it (ws_check)
    // merged engine pass (cn_decode_nast_merged): per-utterance records of the reference batches it carries
    UttMeta* utt_meta = nullptr;
    bool ragged = false;       // the current call has them
    bool ragged_next = false;  // set by cn_decode_nast_merged right before its encoder stage
    bool u_predicted = false;  // the current call's decoder side runs on a predicted row count (>= the true one, or the ticket says so)
    // row-count prediction (cn_decode_nast_merged, u_hint): the decoder side ran on `ticket_U` rows before the true count was
    // known; the count lands in one of four page-locked words (a ticket names its word) for cn_decode_ticket
    int* ymax_ring = nullptr;  // [4], page-locked
    int ticket_U[4] = {0, 0, 0, 0};
    int ticket_id[4] = {-1, -1, -1, -1};  // the ticket (sequence number) whose pass owns the word now: an older ticket has expired
    long long ticket_seq = 0;
    unsigned char* keymask = nullptr;
    void *c1 = nullptr, *c2 = nullptr;
    float* x = nullptr;
    void *xn = nullptr, *qkv = nullptr, *ctx = nullptr, *hbuf = nullptr, *enc_h = nullptr, *kvm = nullptr, *qd = nullptr,
         *dec_h = nullptr;
    float *xd = nullptr, *xd2 = nullptr, *logits = nullptr, *scratch_f32 = nullptr;
    size_t scratch_elems = 0;
    int *best = nullptr, *shift = nullptr, *src_size = nullptr, *ylen = nullptr, *ymax = nullptr, *intervals = nullptr,
        *tok = nullptr, *topk_idx = nullptr;
    float *ctc_maxlp = nullptr, *val = nullptr, *topk_val = nullptr;
    void* kv_all = nullptr;   // [M][kv_cols] bf16: cross-attention K|V of every decoder-side layer, projected by the last encoder
    int kv_cols = 0;          // chain launch's tail (0: each layer projects its own into kvm)
    bool kv_ready = false;    // ... and that launch ran for the current batch
    bool kv_blocked = false;  // ... writing the blocked layout (common.h: cn_blk16_off)
    bool ctx_blocked = false; // m->ctx holds the attention kernel's o_blocked form (its reader is the row-chain kernel)
    long long call_id = 0;  // counts encoder passes: a captured tensor is only served for the call that wrote it
    int c1_halo_B = -1, c1_halo_T1 = -1;  // shape of the haloed conv1 image the buffer currently holds (-1: none)
    bool ctc_maxlp_valid = false;  // the fused arg-max-only CTC generator does not produce it

    void* cv_a = nullptr;      // conformer convolution module scratch
    float* cv_f = nullptr;
    double* gn_stats = nullptr;
    int* ymax_pinned = nullptr;  // page-locked host word for the one data-dependent readback per batch
    // fp16 build (CN_OP16_F16): half-precision operands have a range.  What drives magnitudes from outside is the scale of the
    // features (everything behind linear_out is LayerNorm-ed in fp32 first), so every pass checks them against the largest value
    // for which neither subsampling convolution's output can leave the half range (op16_feat_limit: a word of the weight blob,
    // from the convolutions' weight row sums) and raises a sticky flag in device-visible page-locked memory (cn_take_range_fault).
    // The other build uses the same guard for its split-bf16 engines: conv1's outputs against the e4m3 range of conv2's MIX form
    const float* op16_feat_limit = nullptr;
    float op16_feat_limit_host = -1.f;  // (read back once, on the first cn_take_range_fault)
    unsigned int* op16_fault = nullptr;
    // CTC prefix beam / forced alignment scratch (cn_ctc_beam, cn_decode_nast_forced): grown on demand
    std::map<std::string, std::pair<void*, size_t>> scratch;

    // autoregressive (AST) decoder state (cn_ast_*): token embedding, per-layer cross K|V, KV cache, CTC prefix states
    float* tgt_lut = nullptr;  // [V][d] fp32 (view into the blob)
    int ast_max_len = 0, ast_slots = 0, ast_ctc_beam = 0, ast_Tp_cap = 0, ast_blank = 0;
    std::vector<void*> ast_kvx, ast_ck, ast_cv;  // per decoder layer
    float *ast_logits = nullptr, *ast_r0 = nullptr, *ast_r[2] = {nullptr, nullptr}, *ast_maxlp = nullptr;
    int* ast_arg = nullptr;
    std::vector<void*> ast_allocs;
    AstBeamState beam;  // device-side beam search state (cn_decode_ast)
    int beam_S = 0, beam_L = 0, beam_K = 0;
    int *beam_idx = nullptr;
    float *beam_val = nullptr, *beam_ctc = nullptr;

    // last call
    int B = 0, T = 0, T1 = 0, Tp = 0, U = 0, last_k = 0;
    int dec_group = 1;  // decoder-side batch = B * dec_group (ESA: that many alignments per utterance in one pass), else 1
    std::map<std::string, Capture> captures;

    // per-kernel-tag timing with HIP events on the launch stream (cn_profile_begin / cn_profile_end)
    bool prof_on = false;
    std::string prof_filter;  // empty: every tag
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<ProfPending> prof_pending;
    std::map<std::string, ProfStat> prof_stats;
};

namespace {

int dev_alloc(cn_model* m, void** p, size_t bytes) {
    if (bytes == 0) bytes = 256;
    CN_HIP_CHECK(hipMalloc(p, bytes));
    m->allocs.push_back(*p);
    return 0;
}

// Brackets the launches of one kernel tag with two events on the stream when profiling is on.
struct ProfScope {
    cn_model* m;
    hipStream_t s;
    ProfPending pp;
    bool live = false;
    ProfScope(cn_model* m_, const char* tag, double flops, double bytes, hipStream_t s_) : m(m_), s(s_) {
        if (!m->prof_on) return;
        if (!m->prof_filter.empty() && m->prof_filter.find(std::string("|") + tag + "|") == std::string::npos) return;
        while (m->ev_pool.size() < m->ev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            m->ev_pool.push_back(e);
        }
        pp.tag = tag;
        pp.a = m->ev_pool[m->ev_used++];
        pp.b = m->ev_pool[m->ev_used++];
        pp.flops = flops;
        pp.bytes = bytes;
        live = hipEventRecord(pp.a, s) == hipSuccess;
    }
    ~ProfScope() {
        if (live && hipEventRecord(pp.b, s) == hipSuccess) m->prof_pending.push_back(pp);
    }
};

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
struct Packer {
    cn_model* m;
    bool fill;                        // false: layout only (weights arrive by broadcast)
    std::vector<unsigned char> host;  // staging image of the blob
    size_t off = 256;  // offset 0 is reserved so that a null view means "not present"
    std::string missing;

    size_t reserve(size_t bytes) {
        const size_t at = off;
        off = (off + bytes + 255) & ~(size_t)255;
        if (fill && host.size() < off) host.resize(off);
        return at;
    }
    const HostTensor* find(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = m->host.find(name);
        if (it == m->host.end()) {
            it = m->host.find("module." + name);
            if (it == m->host.end()) {
                if (missing.size() < 400) missing += name + " ";
                return nullptr;
            }
        }
        if (it->second.shape != std::vector<int64_t>(shape)) {
            missing += name + "(shape) ";
            return nullptr;
        }
        return &it->second;
    }
    void put_elem(size_t at, size_t idx, float v) {
        if (m->prec == CN_PREC_F32)
            std::memcpy(&host[at + idx * 4], &v, 4);
        else if (m->prec == CN_PREC_X3) {  // split-bf16: hi / lo halves of the element's 32-group (every packed row is a
            const uint16_t hi = f32_to_bf16_host(v);  // multiple of 32 long, so flat groups never straddle rows)
            uint32_t hb = (uint32_t)hi << 16;
            float hf;
            std::memcpy(&hf, &hb, 4);
            const uint16_t lo = f32_to_bf16_host(v - hf);
            const size_t o = at + cn_split_off(idx);
            std::memcpy(&host[o], &hi, 2);
            std::memcpy(&host[o + 64], &lo, 2);
        } else {
            const uint16_t h = f32_to_bf16_host(v);
            std::memcpy(&host[at + idx * 2], &h, 2);
        }
    }
    // fp32 vector made by concatenating several named vectors
    float* vec(std::initializer_list<std::string> names, int64_t each) {
        const size_t at = reserve(names.size() * each * 4);
        if (fill) {
            size_t k = 0;
            for (auto& n : names) {
                const HostTensor* t = find(n, {each});
                if (t) std::memcpy(&host[at + k * each * 4], t->data.data(), each * 4);
                ++k;
            }
        }
        return reinterpret_cast<float*>(at);
    }
    // [sum rows][K] matrix in model precision from several [rows][K] matrices, optional column permutation
    static unsigned char f32_to_e4m3(float f) { return cn_f32_to_e4m3_host(f); }
    // fp8 copy of a packed Linear (same row order as `linear` built it from `prefixes`): one power-of-two scale per tensor,
    // the largest that keeps max|w| * scale <= 448
    void quant8(Linear& l, std::initializer_list<std::string> prefixes, int64_t rows_each) {
        const size_t at = reserve((size_t)l.N * l.K);
        const size_t sat = reserve(4);
        if (fill) {
            float mx = 0.f;
            for (auto& pfx : prefixes) {
                const HostTensor* t = find(pfx + ".weight", {rows_each, (int64_t)l.K});
                if (t)
                    for (float v : t->data) mx = std::max(mx, std::fabs(v));
            }
            const float scale = mx > 0.f ? std::ldexp(1.f, (int)std::floor(std::log2(448.f / mx))) : 1.f;
            size_t r0 = 0;
            for (auto& pfx : prefixes) {
                const HostTensor* t = find(pfx + ".weight", {rows_each, (int64_t)l.K});
                if (t)
                    for (int64_t i = 0; i < rows_each * l.K; ++i) host[at + r0 * l.K + i] = f32_to_e4m3(t->data[i] * scale);
                r0 += rows_each;
            }
            const float inv = 1.f / scale;
            std::memcpy(&host[sat], &inv, 4);
        }
        l.W8 = reinterpret_cast<void*>(at);
        l.w8_inv = reinterpret_cast<float*>(sat);
    }
    Linear linear(std::initializer_list<std::string> prefixes, int64_t rows_each, int64_t K,
                  const std::vector<int>* colperm = nullptr) {
        Linear l;
        l.N = (int)(prefixes.size() * rows_each);
        l.K = (int)K;
        const size_t at = reserve((size_t)l.N * K * m->es);
        if (fill) {
            size_t r0 = 0;
            for (auto& pfx : prefixes) {
                const HostTensor* t = find(pfx + ".weight", {rows_each, K});
                if (t) {
                    for (int64_t r = 0; r < rows_each; ++r)
                        for (int64_t c = 0; c < K; ++c) {
                            const int64_t src_c = colperm ? (*colperm)[c] : c;
                            put_elem(at, (r0 + r) * K + c, t->data[r * K + src_c]);
                        }
                }
                r0 += rows_each;
            }
        }
        l.W = reinterpret_cast<void*>(at);
        if (m->prec == CN_PREC_X3 && proj_x3_applies(l.N, l.K)) {  // the same matrix as the projection kernel's fragment stream
            const size_t pat = reserve((size_t)l.N * 1024);
            if (fill) {
                size_t r0 = 0;
                for (auto& pfx : prefixes) {
                    const HostTensor* t = find(pfx + ".weight", {rows_each, K});
                    if (t)
                        for (int64_t r = 0; r < rows_each; ++r)
                            for (int64_t c = 0; c < K; ++c) {
                                const float v = t->data[r * K + (colperm ? (*colperm)[c] : c)];
                                const uint16_t hi = f32_to_bf16_host(v);
                                const uint32_t hb = (uint32_t)hi << 16;
                                float hf;
                                std::memcpy(&hf, &hb, 4);
                                const uint16_t lo = f32_to_bf16_host(v - hf);
                                const size_t o = pat + proj_x3_off((int)(r0 + r), (int)c);
                                std::memcpy(&host[o], &hi, 2);
                                std::memcpy(&host[o + 1024], &lo, 2);
                            }
                    r0 += rows_each;
                }
            }
            l.px3 = reinterpret_cast<void*>(pat);
        }
        size_t bat = reserve((size_t)l.N * 4);
        if (fill) {
            size_t k = 0;
            for (auto& pfx : prefixes) {
                const HostTensor* t = find(pfx + ".bias", {rows_each});
                if (t) std::memcpy(&host[bat + k * rows_each * 4], t->data.data(), rows_each * 4);
                ++k;
            }
        }
        l.b = reinterpret_cast<float*>(bat);
        return l;
    }
    // nn.Conv1d(kernel 1) weight [rows][K][1] as a Linear
    Linear pointwise(const std::string& prefix, int64_t rows, int64_t K) {
        Linear l;
        l.N = (int)rows;
        l.K = (int)K;
        const size_t at = reserve((size_t)rows * K * m->es);
        if (fill) {
            const HostTensor* t = find(prefix + ".weight", {rows, K, 1});
            if (t)
                for (int64_t i = 0; i < rows * K; ++i) put_elem(at, i, t->data[i]);
        }
        l.W = reinterpret_cast<void*>(at);
        l.b = vec({prefix + ".bias"}, rows);
        return l;
    }
    // fp32 copy of a named tensor of any shape with n elements
    float* raw(const std::string& name, std::initializer_list<int64_t> shape, int64_t n) {
        const size_t at = reserve((size_t)n * 4);
        if (fill) {
            const HostTensor* t = find(name, shape);
            if (t) std::memcpy(&host[at], t->data.data(), (size_t)n * 4);
        }
        return reinterpret_cast<float*>(at);
    }
    // plain (unfused) feed-forward pair
    void ffn_plain(Linear& w1, Linear& w2, const std::string& p, int64_t dff, int64_t d) {
        w1 = linear({p + ".w_1"}, dff, d);
        w2 = linear({p + ".w_2"}, d, dff);
    }
    ConvMod conv_module(const std::string& p, int64_t d, int64_t k) {
        ConvMod c;
        c.pw1 = pointwise(p + ".pointwise_conv1", 2 * d, d);
        c.dw_w = raw(p + ".depthwise_conv.weight", {d, 1, k}, d * k);
        c.dw_b = vec({p + ".depthwise_conv.bias"}, d);
        c.gn_w = vec({p + ".norm.weight"}, d);
        c.gn_b = vec({p + ".norm.bias"}, d);
        c.pw2 = pointwise(p + ".pointwise_conv2", d, d);
        c.k = (int)k;
        return c;
    }
    // relative-position self attention: fused Q|K|V, output projection, u / v, and P = table . W_pos^T (the position rows
    // are a frozen sinusoid table in the checkpoint, so their projection is a constant of the layer)
    void rel_attn(Layer& L, const std::string& att, const std::string& table, int64_t R, int64_t d, int64_t H) {
        L.has_self = true;
        L.qkv = linear({att + ".linears.0", att + ".linears.1", att + ".linears.2"}, d, d);
        L.self_o = linear({att + ".linears.3"}, d, d);
        L.pos_u = raw(att + ".pos_bias_u", {H, d / H}, d);
        L.pos_v = raw(att + ".pos_bias_v", {H, d / H}, d);
        const int64_t nr = 2 * R + 1;
        const size_t at = reserve((size_t)nr * d * 4);
        if (fill) {
            const HostTensor* tb = find(table, {nr, d});
            const HostTensor* wp = find(att + ".linear_pos.weight", {d, d});
            if (tb && wp)
                for (int64_t r = 0; r < nr; ++r)
                    for (int64_t o = 0; o < d; ++o) {
                        double acc = 0.0;
                        for (int64_t i = 0; i < d; ++i) acc += (double)tb->data[r * d + i] * (double)wp->data[o * d + i];
                        const float v = (float)acc;
                        std::memcpy(&host[at + (r * d + o) * 4], &v, 4);
                    }
        }
        L.pos_proj = reinterpret_cast<float*>(at);
        L.rel_R = (int)R;
    }
    // conformer self-attention block (fanat_conformer_blocks.py:9-38): feed_forward1, self_attn, conv_module, feed_forward2
    Layer conformer_layer(const std::string& p, const std::string& table, int64_t dff, int64_t k, int64_t R, int64_t d, int64_t H,
                          int nnorm) {
        Layer L;
        L.conformer = true;
        rel_attn(L, p + ".self_attn", table, R, d, H);
        ffn_plain(L.w1, L.w2, p + ".feed_forward1", dff, d);
        L.conv = conv_module(p + ".conv_module", d, k);
        ffn_plain(L.ff2_w1, L.ff2_w2, p + ".feed_forward2", dff, d);
        for (int i = 0; i < nnorm; ++i) L.n[i] = norm(p + ".sublayer." + std::to_string(i) + ".norm", d);
        return L;
    }
    // feed-forward weights: fragment streams for the fused kernel when it applies, plain matrices otherwise
    void ffn(Layer& L, const std::string& p, int64_t dff, int64_t d) {
        const bool fused = m->prec == CN_PREC_BF16 && d == 256 && dff % 128 == 0 && dff <= 2048;
        if (m->prec == CN_PREC_X3 && ffn_x3_applies((int)d, (int)dff)) {
            const size_t ax = reserve(ffn_x3_stream_bytes((int)dff));
            if (fill) {
                const HostTensor* t1 = find(p + ".feed_forward.w_1.weight", {dff, d});
                const HostTensor* t2 = find(p + ".feed_forward.w_2.weight", {d, dff});
                if (t1 && t2) pack_ffn_x3(t1->data.data(), t2->data.data(), (int)dff, reinterpret_cast<uint16_t*>(&host[ax]), ffn_mix_applies());
            }
            L.wx3 = reinterpret_cast<void*>(ax);
            L.w1.N = (int)dff;
            L.w1.K = (int)d;
            L.w1.b = vec({p + ".feed_forward.w_1.bias"}, dff);
            L.w2.N = (int)d;
            L.w2.K = (int)dff;
            L.w2.b = vec({p + ".feed_forward.w_2.bias"}, d);
            return;
        }
        if (!fused) {
            L.w1 = linear({p + ".feed_forward.w_1"}, dff, d);
            L.w2 = linear({p + ".feed_forward.w_2"}, d, dff);
            return;
        }
        const size_t a1 = reserve((size_t)dff * d * 2), a2 = reserve((size_t)dff * d * 2);
        if (fill) {
            const HostTensor* t1 = find(p + ".feed_forward.w_1.weight", {dff, d});
            const HostTensor* t2 = find(p + ".feed_forward.w_2.weight", {d, dff});
            if (t1) pack_ffn_w1(t1->data.data(), (int)dff, reinterpret_cast<uint16_t*>(&host[a1]));
            if (t2) pack_ffn_w2(t2->data.data(), (int)dff, reinterpret_cast<uint16_t*>(&host[a2]));
        }
        L.w1p = reinterpret_cast<void*>(a1);
        L.w2p = reinterpret_cast<void*>(a2);
        L.w1.N = (int)dff;
        L.w1.K = (int)d;
        L.w1.b = vec({p + ".feed_forward.w_1.bias"}, dff);
        L.w2.N = (int)d;
        L.w2.K = (int)dff;
        L.w2.b = vec({p + ".feed_forward.w_2.bias"}, d);
    }
    bool chain_ok(int64_t d, int64_t dff) const {
        return m->prec == CN_PREC_BF16 && d == 256 && dff % 32 == 0 && dff <= 2048;
    }
    // row-chain stream: out-projection `wo` ("" = none), FFN under `p` with pre-norm `ln1` (dff 0 = none), then norm
    // `nln` ("" = nothing follows) and the concatenated projections `tails` (each [d][d]) of whatever consumes the
    // stream next.  Seven spare units follow the stream: the kernel's dummy refills past the end read them.
    ChainRef chain(const std::string& wo, const std::string& ln1, const std::string& p, int64_t dff, const std::string& nln,
                   std::vector<std::string> tails, int64_t d, bool f8 = false) {
        ChainRef r;
        const int tail_n = (int)(tails.size() * d);
        const size_t units = chain_stream_units(!wo.empty(), (int)dff, tail_n, f8);
        const size_t aw = reserve((units + 7) * CHAIN_UNIT_BYTES), at = reserve((size_t)CHAIN_TAB_FLOATS * 4);
        const size_t aq = f8 ? reserve(16) : 0;
        if (fill) {
            ChainWeights w;
            bool ok = true;
            auto get = [&](const std::string& n, std::initializer_list<int64_t> shape) -> const float* {
                const HostTensor* t = find(n, shape);
                if (!t) ok = false;
                return t ? t->data.data() : nullptr;
            };
            if (!wo.empty()) {
                w.wo = get(wo + ".weight", {d, d});
                w.bo = get(wo + ".bias", {d});
            }
            if (dff) {
                w.ln1_a = get(ln1 + ".a_2", {d});
                w.ln1_b = get(ln1 + ".b_2", {d});
                w.w1 = get(p + ".feed_forward.w_1.weight", {dff, d});
                w.b1 = get(p + ".feed_forward.w_1.bias", {dff});
                w.w2 = get(p + ".feed_forward.w_2.weight", {d, dff});
                w.b2 = get(p + ".feed_forward.w_2.bias", {d});
            }
            if (!nln.empty()) {
                w.nln_a = get(nln + ".a_2", {d});
                w.nln_b = get(nln + ".b_2", {d});
            }
            std::vector<float> tw((size_t)tail_n * d), tb((size_t)tail_n);
            size_t k = 0;
            for (auto& tp : tails) {
                const float* a = get(tp + ".weight", {d, d});
                const float* b = get(tp + ".bias", {d});
                if (a && b) {
                    std::memcpy(&tw[k * d * d], a, (size_t)d * d * 4);
                    std::memcpy(&tb[k * d], b, (size_t)d * 4);
                }
                ++k;
            }
            w.wt = tw.data();
            w.bt = tb.data();
            w.dff = (int)dff;
            w.tail_n = tail_n;
            if (ok) pack_chain(w, reinterpret_cast<uint16_t*>(&host[aw]), reinterpret_cast<float*>(&host[at]),
                               f8 ? reinterpret_cast<int*>(&host[aq]) : nullptr);
        }
        r.w = reinterpret_cast<void*>(aw);
        r.tab = reinterpret_cast<float*>(at);
        r.f8q = f8 ? reinterpret_cast<int*>(aq) : nullptr;
        r.dff = (int)dff;
        r.tail_n = tail_n;
        r.has_wo = !wo.empty();
        r.has_next = !nln.empty();
        return r;
    }
    // The same stream for a conformer sublayer group (fanat_conformer_blocks.py:26-38): explicit tensor names, the
    // output projection / tail may be a pointwise convolution ([rows][d][1]), the feed-forward is a macaron half -
    // Swish, and its 0.5 residual scale folded into w_2 and b_2 (a power of two: exact in bf16)
    struct ConfChain {
        std::string wo;          // "" or prefix of a [d][d] Linear / [d][d][1] pointwise conv (+ ".bias")
        bool wo_conv = false;
        std::string ln1, ffn;    // FFN pre-norm and prefix (".w_1" / ".w_2"), "" = none
        int64_t dff = 0;
        float ffn_scale = 1.f;
        std::string nln;         // next norm ("" = none)
        std::vector<std::string> tails;  // each a prefix of [rows][d](,1) weights
        int64_t tail_rows = 0;   // rows per tail tensor
        bool tail_conv = false;
    };
    ChainRef conf_chain(const ConfChain& c, int64_t d) {
        ChainRef r;
        const int tail_n = (int)(c.tails.size() * c.tail_rows);
        const size_t units = chain_stream_units(!c.wo.empty(), (int)c.dff, tail_n);
        const size_t aw = reserve((units + 7) * CHAIN_UNIT_BYTES), at = reserve((size_t)CHAIN_TAB_FLOATS * 4);
        if (fill) {
            ChainWeights w;
            bool ok = true;
            auto get = [&](const std::string& n, std::initializer_list<int64_t> shape) -> const float* {
                const HostTensor* t = find(n, shape);
                if (!t) ok = false;
                return t ? t->data.data() : nullptr;
            };
            std::vector<float> w2s, b2s;
            if (!c.wo.empty()) {
                w.wo = c.wo_conv ? get(c.wo + ".weight", {d, d, 1}) : get(c.wo + ".weight", {d, d});
                w.bo = get(c.wo + ".bias", {d});
            }
            if (c.dff) {
                w.ln1_a = get(c.ln1 + ".a_2", {d});
                w.ln1_b = get(c.ln1 + ".b_2", {d});
                w.w1 = get(c.ffn + ".w_1.weight", {c.dff, d});
                w.b1 = get(c.ffn + ".w_1.bias", {c.dff});
                const float* w2 = get(c.ffn + ".w_2.weight", {d, c.dff});
                const float* b2 = get(c.ffn + ".w_2.bias", {d});
                if (w2 && b2) {
                    w2s.assign(w2, w2 + (size_t)d * c.dff);
                    b2s.assign(b2, b2 + d);
                    for (auto& v : w2s) v *= c.ffn_scale;
                    for (auto& v : b2s) v *= c.ffn_scale;
                    w.w2 = w2s.data();
                    w.b2 = b2s.data();
                }
            }
            if (!c.nln.empty()) {
                w.nln_a = get(c.nln + ".a_2", {d});
                w.nln_b = get(c.nln + ".b_2", {d});
            }
            std::vector<float> tw((size_t)tail_n * d), tb((size_t)tail_n);
            size_t k = 0;
            for (auto& tp : c.tails) {
                const float* a = c.tail_conv ? get(tp + ".weight", {c.tail_rows, d, 1}) : get(tp + ".weight", {c.tail_rows, d});
                const float* b = get(tp + ".bias", {c.tail_rows});
                if (a && b) {
                    std::memcpy(&tw[k * c.tail_rows * d], a, (size_t)c.tail_rows * d * 4);
                    std::memcpy(&tb[k * c.tail_rows], b, (size_t)c.tail_rows * 4);
                }
                ++k;
            }
            w.wt = tw.data();
            w.bt = tb.data();
            w.dff = (int)c.dff;
            w.tail_n = tail_n;
            if (ok) pack_chain(w, reinterpret_cast<uint16_t*>(&host[aw]), reinterpret_cast<float*>(&host[at]));
        }
        r.w = reinterpret_cast<void*>(aw);
        r.tab = reinterpret_cast<float*>(at);
        r.dff = (int)c.dff;
        r.tail_n = tail_n;
        r.has_wo = !c.wo.empty();
        r.has_next = !c.nln.empty();
        r.swish = c.dff > 0;
        return r;
    }
    // generator: the plain matrix (beam search / capture) plus the fused-argmax fragment stream when it applies
    Linear generator(const std::string& prefix, int64_t V, int64_t d) {
        Linear l = linear({prefix}, V, d);
        if (genmax_applies(m->prec, (int)d, (int)V)) {
            const bool x3 = m->prec == CN_PREC_X3;
            const int vtw = x3 ? genmax_x3_vtw((int)V) : genmax_vtw((int)V);
            const size_t aw = reserve(x3 ? (size_t)8 * vtw * 32 * 1024 : (size_t)4 * vtw * 16 * 1024);
            const size_t ab = reserve((size_t)(x3 ? 8 : 4) * vtw * 32 * 4);
            if (fill) {
                const HostTensor* tw = find(prefix + ".weight", {V, d});
                const HostTensor* tb = find(prefix + ".bias", {V});
                if (tw && tb)
                    (x3 ? pack_genmax_x3 : pack_genmax)(tw->data.data(), tb->data.data(), (int)V, reinterpret_cast<uint16_t*>(&host[aw]),
                                                        reinterpret_cast<float*>(&host[ab]));
            }
            l.gm_w = reinterpret_cast<void*>(aw);
            l.gm_b = reinterpret_cast<float*>(ab);
        }
        return l;
    }
    Norm norm(const std::string& prefix, int64_t d) {
        Norm n;
        n.a = vec({prefix + ".a_2"}, d);
        n.b = vec({prefix + ".b_2"}, d);
        return n;
    }
};

template <typename P> void rebase(P*& p, unsigned char* base) {
    if (p) p = reinterpret_cast<P*>(base + reinterpret_cast<size_t>(p));
}
void rebase_linear(Linear& l, unsigned char* base) {
    rebase(l.W, base);
    rebase(l.b, base);
    rebase(l.px3, base);
    rebase(l.gm_w, base);
    rebase(l.gm_b, base);
    rebase(l.W8, base);
    rebase(l.w8_inv, base);
}
void rebase_norm(Norm& n, unsigned char* base) {
    rebase(n.a, base);
    rebase(n.b, base);
}

int build_weights(cn_model* m) {
    const cn_config& c = m->cfg;
    const int64_t d = c.d_model, V = c.vocab_size, C = c.d_model;
    Packer pk;
    pk.m = m;
    pk.fill = !m->host.empty() && !m->blob_borrowed;
    const int64_t F2 = m->F2;
    // cfg.ast == 2: the TransformerLM that ranks ESA samples (src/models/lm.py): token embedding, encoder stack, generator
    const bool lm = c.ast == 2;

    // conv1: (C,1,3,3) -> [tap][C] fp32
    if (!lm) {
        const size_t at = pk.reserve(9 * C * 4);
        if (pk.fill) {
            const HostTensor* t = pk.find("src_embed.conv.0.weight", {C, 1, 3, 3});
            if (t)
                for (int64_t ch = 0; ch < C; ++ch)
                    for (int tap = 0; tap < 9; ++tap)
                        std::memcpy(&pk.host[at + (tap * C + ch) * 4], &t->data[ch * 9 + tap], 4);
        }
        m->conv1_w = reinterpret_cast<float*>(at);
        m->conv1_b = pk.vec({"src_embed.conv.0.bias"}, C);
    }
    // conv2: (Co,Ci,3,3) -> [Co][(kh,kw,ci)]
    if (!lm) {
        m->conv2.N = (int)C;
        m->conv2.K = (int)(9 * C);
        const size_t at = pk.reserve((size_t)C * 9 * C * m->es);
        if (pk.fill) {
            const HostTensor* t = pk.find("src_embed.conv.2.weight", {C, C, 3, 3});
            if (t)
                for (int64_t co = 0; co < C; ++co)
                    for (int64_t ci = 0; ci < C; ++ci)
                        for (int tap = 0; tap < 9; ++tap)
                            pk.put_elem(at, co * 9 * C + tap * C + ci, t->data[(co * C + ci) * 9 + tap]);
        }
        m->conv2.W = reinterpret_cast<void*>(at);
        if (conv2_x3_applies(m->prec, (int)C, (int)C)) {
            const size_t plane = (size_t)C * 9 * C * 2, pat = pk.reserve(2 * plane);
            if (pk.fill) {
                const HostTensor* t = pk.find("src_embed.conv.2.weight", {C, C, 3, 3});
                if (t)
                    for (int64_t co = 0; co < C; ++co)
                        for (int64_t ci = 0; ci < C; ++ci)
                            for (int tap = 0; tap < 9; ++tap) {
                                const float v = t->data[(co * C + ci) * 9 + tap];
                                const uint16_t hi = f32_to_bf16_host(v);
                                const uint32_t hb = (uint32_t)hi << 16;
                                float hf;
                                std::memcpy(&hf, &hb, 4);
                                const uint16_t lo = f32_to_bf16_host(v - hf);
                                const size_t o = pat + (size_t)(co * 9 * C + tap * C + ci) * 2;
                                std::memcpy(&pk.host[o], &hi, 2);
                                std::memcpy(&pk.host[o + plane], &lo, 2);
                            }
            }
            m->conv2_x3w = reinterpret_cast<void*>(pat);
        }
        if (conv2_mix_applies(m->prec, (int)C, (int)C)) {
            const size_t n = (size_t)C * 9 * C, pat = pk.reserve(4 * n), qat = pk.reserve(16);
            if (pk.fill) {
                const HostTensor* t = pk.find("src_embed.conv.2.weight", {C, C, 3, 3});
                if (t) {
                    float mx = 0.f;
                    for (float v : t->data) mx = std::max(mx, std::fabs(v));
                    // q: the largest power-of-two scale that keeps the weights inside e4m3; l: |w - half(w)| <= 2^-11 |w|
                    const int lg = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
                    const float sq = std::ldexp(1.f, lg), sl = std::ldexp(1.f, lg + 11);
                    for (int64_t co = 0; co < C; ++co)
                        for (int64_t ci = 0; ci < C; ++ci)
                            for (int tap = 0; tap < 9; ++tap) {
                                const float v = t->data[(co * C + ci) * 9 + tap];
                                const _Float16 h = (_Float16)v;
                                const size_t e = (size_t)(co * 9 * C + tap * C + ci);
                                std::memcpy(&pk.host[pat + 2 * e], &h, 2);
                                pk.host[pat + 2 * n + e] = cn_f32_to_e4m3_host(v * sq);
                                pk.host[pat + 3 * n + e] = cn_f32_to_e4m3_host((v - (float)h) * sl);
                            }
                    const int q[4] = {127 - lg, 127 - MIX_LG_AL, 127 - (lg + 11), 127 - MIX_LG_AQ};  // {W_q, A_l, W_l, A_q}
                    std::memcpy(&pk.host[qat], q, 16);
                }
            }
            m->conv2_mixw = reinterpret_cast<void*>(pat);
            m->conv2_mixq = reinterpret_cast<int*>(qat);
        }
        if (m->fp8_enc && (m->fp8_scope & CN_FP8_CONV2) && conv2_f8_applies((int)C, (int)C)) {  // config 5: the second convolution on e4m3 operands (conv2.hip, F8)
            const size_t fat = pk.reserve((size_t)C * 9 * C), qat = pk.reserve(16);
            if (pk.fill) {
                const HostTensor* t = pk.find("src_embed.conv.2.weight", {C, C, 3, 3});
                if (t) {
                    float mx = 0.f;
                    for (float v : t->data) mx = std::max(mx, std::fabs(v));
                    const int lg = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
                    const float scale = std::ldexp(1.f, lg);
                    for (int64_t co = 0; co < C; ++co)
                        for (int64_t ci = 0; ci < C; ++ci)
                            for (int tap = 0; tap < 9; ++tap)
                                pk.host[fat + (size_t)(co * 9 * C + tap * C + ci)] = cn_f32_to_e4m3_host(t->data[(co * C + ci) * 9 + tap] * scale);
                    const int q[4] = {127 - lg, 127 - (int)std::lround(std::log2(FP8_S_IMG)), 0, 0};
                    std::memcpy(&pk.host[qat], q, 16);
                }
            }
            m->conv2_f8w = reinterpret_cast<void*>(fat);
            m->conv2_f8q = reinterpret_cast<int*>(qat);
        }
        m->conv2.b = pk.vec({"src_embed.conv.2.bias"}, C);
    }
    if (!lm) {
        // |conv1 out| <= fmax A1 + B1 and |conv2 out| <= |conv1 out| A2 + B2 with A = the largest row sum of |weights|, B = the
        // largest |bias|: the feature magnitude that keeps both under half of 65504 (0: weights unknown here - no check)
        const size_t at = pk.reserve(16);
        if (pk.fill) {
            float lim = 0.f;
            const HostTensor* w1 = pk.find("src_embed.conv.0.weight", {C, 1, 3, 3});
            const HostTensor* b1 = pk.find("src_embed.conv.0.bias", {C});
            const HostTensor* w2 = pk.find("src_embed.conv.2.weight", {C, C, 3, 3});
            const HostTensor* b2 = pk.find("src_embed.conv.2.bias", {C});
            if (w1 && b1 && w2 && b2) {
                double A1 = 0, B1 = 0, A2 = 0, B2 = 0;
                for (int64_t ch = 0; ch < C; ++ch) {
                    double r1 = 0, r2 = 0;
                    for (int k = 0; k < 9; ++k) r1 += std::fabs((double)w1->data[ch * 9 + k]);
                    for (int64_t k = 0; k < 9 * C; ++k) r2 += std::fabs((double)w2->data[ch * 9 * C + k]);
                    A1 = std::max(A1, r1);
                    A2 = std::max(A2, r2);
                    B1 = std::max(B1, std::fabs((double)b1->data[ch]));
                    B2 = std::max(B2, std::fabs((double)b2->data[ch]));
                }
#ifdef CN_OP16_F16
                const double half_max = 65504.0 / 2;
                const double c1 = std::min(half_max, A2 > 0 ? (half_max - B2) / A2 : half_max);  // the bound on conv1's output
#else
                // (this build: the split-bf16 engine's conv2 in the mixed arithmetic - conv1's outputs have to stay inside the
                // e4m3 range of their q plane, 448 / 2^MIX_LG_AQ; beyond it the cross terms saturate and the product falls back
                // towards half precision: the engine would keep decoding, below its tolerance)
                (void)A2;
                (void)B2;
                const double c1 = 448.0 / std::ldexp(1.0, MIX_LG_AQ);
#endif
                lim = (float)std::max(0.0, A1 > 0 ? (c1 - B1) / A1 : 3e38);
            }
            std::memcpy(&pk.host[at], &lim, 4);
        }
        m->op16_feat_limit = reinterpret_cast<const float*>(at);
    }
    // linear_out: column c*F2+f of the reference (embedding.py:118) -> column f*C+c (the conv2 GEMM's natural output)
    if (!lm) {
        std::vector<int> perm((size_t)(C * F2));
        for (int64_t f = 0; f < F2; ++f)
            for (int64_t ch = 0; ch < C; ++ch) perm[(size_t)(f * C + ch)] = (int)(ch * F2 + f);
        m->linear_out = pk.linear({"src_embed.linear_out"}, d, C * F2, &perm);
        if (m->fp8_enc && (m->fp8_scope & CN_FP8_LINEAR) && m->conv2_f8w && linear256_f8_applies((int)d, (int)(C * F2))) {  // config 5: linear_out on e4m3 operands as well
            const size_t fat = pk.reserve((size_t)d * C * F2), qat = pk.reserve(16);
            if (pk.fill) {
                const HostTensor* t = pk.find("src_embed.linear_out.weight", {d, C * F2});
                if (t) {
                    float mx = 0.f;
                    for (float v : t->data) mx = std::max(mx, std::fabs(v));
                    const int lg = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
                    const float scale = std::ldexp(1.f, lg);
                    for (int64_t r = 0; r < d; ++r)
                        for (int64_t cc = 0; cc < C * F2; ++cc)
                            pk.host[fat + (size_t)(r * C * F2 + cc)] = cn_f32_to_e4m3_host(t->data[r * C * F2 + perm[(size_t)cc]] * scale);
                    const int q[4] = {127 - lg, 127 - (int)std::lround(std::log2(FP8_S_IMG)), 0, 0};
                    std::memcpy(&pk.host[qat], q, 16);
                }
            }
            m->linear_f8w = reinterpret_cast<void*>(fat);
            m->linear_f8q = reinterpret_cast<int*>(qat);
        }
    }
    auto self_layer = [&](const std::string& p, const std::string& att, int64_t dff, int nnorm) {
        Layer L;
        L.has_self = true;
        L.qkv = pk.linear({p + "." + att + ".linears.0", p + "." + att + ".linears.1", p + "." + att + ".linears.2"}, d, d);
        L.self_o = pk.linear({p + "." + att + ".linears.3"}, d, d);
        pk.ffn(L, p, dff, d);
        for (int i = 0; i < nnorm; ++i) L.n[i] = pk.norm(p + ".sublayer." + std::to_string(i) + ".norm", d);
        return L;
    };
    auto add_src = [&](Layer& L, const std::string& p) {
        L.has_src = true;
        L.src_q = pk.linear({p + ".src_attn.linears.0"}, d, d);
        L.src_kv = pk.linear({p + ".src_attn.linears.1", p + ".src_attn.linears.2"}, d, d);
        L.src_o = pk.linear({p + ".src_attn.linears.3"}, d, d);
    };
    m->enc.clear();
    m->extra.clear();
    m->sad.clear();
    m->mad.clear();
    const int64_t H = c.n_head;
    for (int n = 0; n < c.n_enc; ++n) {
        const std::string p = "encoder.layers." + std::to_string(n);
        if (c.conf_enc)
            m->enc.push_back(pk.conformer_layer(p, "src_embed.pos_enc.embedding.weight", c.d_encff, c.enc_kernel, c.enc_max_rel, d, H, 4));
        else {
            Layer L = self_layer(p, "self_attn", c.d_encff, 2);
            if (m->fp8_enc) {  // config 5: e4m3fn copies of the four products of an encoder layer
                pk.quant8(L.qkv, {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"}, d);
                pk.quant8(L.self_o, {p + ".self_attn.linears.3"}, d);
                pk.quant8(L.w1, {p + ".feed_forward.w_1"}, c.d_encff);
                pk.quant8(L.w2, {p + ".feed_forward.w_2"}, d);
            }
            m->enc.push_back(L);
        }
    }
    m->enc_norm = pk.norm("encoder.norm", d);
    // conformer layer = row-chain launches around the attention and the depthwise-convolution kernels:
    //   A: x += 0.5 FFN1(LN x);            next: self-attention pre-norm + Q|K|V
    //   B: x += W_o ctx;                   next: convolution pre-norm + pointwise conv 1 (2d columns)
    //   C: x += pointwise conv 2 (module); self layers: x += 0.5 FFN2(LN x) (+ `final_norm` -> output after the last one)
    //      mixed-attention layers: next: source-attention pre-norm + its query projection, then
    //   D: x += W_o' ctx;  x += 0.5 FFN2(LN x) (+ `final_norm`)
    auto conf_layer_chains = [&](Layer& L, const std::string& p, int64_t dff, bool mixed, const std::string& final_norm) {
        Packer::ConfChain a, b, cc, dd;
        a.ln1 = p + ".sublayer.0.norm";
        a.ffn = p + ".feed_forward1";
        a.dff = dff;
        a.ffn_scale = 0.5f;
        a.nln = p + ".sublayer.2.norm";
        a.tails = {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"};
        a.tail_rows = d;
        L.cf_a = pk.conf_chain(a, d);
        b.wo = p + ".self_attn.linears.3";
        b.nln = p + ".sublayer.1.norm";
        b.tails = {p + ".conv_module.pointwise_conv1"};
        b.tail_rows = 2 * d;
        b.tail_conv = true;
        L.cf_b = pk.conf_chain(b, d);
        cc.wo = p + ".conv_module.pointwise_conv2";
        cc.wo_conv = true;
        if (!mixed) {
            cc.ln1 = p + ".sublayer.3.norm";
            cc.ffn = p + ".feed_forward2";
            cc.dff = dff;
            cc.ffn_scale = 0.5f;
            cc.nln = final_norm;
        } else {
            cc.nln = p + ".sublayer.3.norm";
            cc.tails = {p + ".src_attn.linears.0"};
            cc.tail_rows = d;
            dd.wo = p + ".src_attn.linears.3";
            dd.ln1 = p + ".sublayer.4.norm";
            dd.ffn = p + ".feed_forward2";
            dd.dff = dff;
            dd.ffn_scale = 0.5f;
            dd.nln = final_norm;
            L.cf_d = pk.conf_chain(dd, d);
        }
        L.cf_c = pk.conf_chain(cc, d);
    };
    if (c.conf_enc && pk.chain_ok(d, c.d_encff) && c.d_encff % 128 == 0)
        for (int n = 0; n < c.n_enc; ++n)
            conf_layer_chains(m->enc[n], "encoder.layers." + std::to_string(n), c.d_encff, false,
                              n + 1 == c.n_enc ? "encoder.norm" : "");
    m->enc_chain.clear();
    m->kv_cols = 0;
    // fp8 engine: the same chains with the feed-forward units in e4m3 (chain.hip, F8 form) when d_ff allows, else the unfused
    // per-product path (run_enc_layer_fp8)
    static const bool f8_unfused = cn_exp_env("CASSNAT_FP8_UNFUSED") != nullptr;
    const bool f8c = m->fp8_enc && !lm && !f8_unfused && c.d_encff % 256 == 0;
    // CASSNAT_FP8_LAYERS: bit n set = layer n's feed-forward products in e4m3 (accuracy experiments: tools/fp8_accuracy.py)
    static const unsigned f8_layers = cn_exp_env("CASSNAT_FP8_LAYERS") ? (unsigned)strtoul(cn_exp_env("CASSNAT_FP8_LAYERS"), nullptr, 0) : ~0u;
    auto f8_at = [&](int n) {
        return f8c && (m->fp8_scope & CN_FP8_FFN) && n >= c.fp8_ffn_first_layer && ((f8_layers >> (n & 31)) & 1u);
    };
    if (!c.conf_enc && (!m->fp8_enc || f8c) && pk.chain_ok(d, c.d_encff))  // (the TransformerLM's layers are encoder layers: same chains)
        for (int n = 0; n < c.n_enc; ++n) {
            const std::string p = "encoder.layers." + std::to_string(n), q = "encoder.layers." + std::to_string(n + 1);
            if (n + 1 < c.n_enc)
                m->enc_chain.push_back(pk.chain(p + ".self_attn.linears.3", p + ".sublayer.1.norm", p, c.d_encff,
                                                q + ".sublayer.0.norm",
                                                {q + ".self_attn.linears.0", q + ".self_attn.linears.1", q + ".self_attn.linears.2"}, d, f8_at(n)));
            else {
                // the last launch also projects enc_h = encoder.norm(x) onto the cross-attention K|V of every decoder-side
                // layer (extractor, then mixed-attention decoder): three 8000-row GEMM launches per batch become the tail of
                // a launch that already holds enc_h in registers
                std::vector<std::string> kv_tails;
                if (!lm && !c.ast && !c.conf_dec && pk.chain_ok(d, c.d_decff) && (c.n_extra + c.n_mix_dec) * 2 * d <= 1536) {
                    for (int j = 0; j < c.n_extra; ++j)
                        for (int w = 1; w <= 2; ++w)
                            kv_tails.push_back("acembed_extractor.layers." + std::to_string(j) + ".src_attn.linears." + std::to_string(w));
                    for (int j = 0; j < c.n_mix_dec; ++j)
                        for (int w = 1; w <= 2; ++w)
                            kv_tails.push_back("decoder.layers." + std::to_string(j) + ".src_attn.linears." + std::to_string(w));
                }
                m->kv_cols = (int)kv_tails.size() * (int)d;
                m->enc_chain.push_back(
                    pk.chain(p + ".self_attn.linears.3", p + ".sublayer.1.norm", p, c.d_encff, "encoder.norm", kv_tails, d, f8_at(n)));
            }
        }
    m->enc_entry = ChainRef();
    if (!m->enc_chain.empty())
        m->enc_entry = pk.chain("", "", "", 0, "encoder.layers.0.sublayer.0.norm",
                                {"encoder.layers.0.self_attn.linears.0", "encoder.layers.0.self_attn.linears.1",
                                 "encoder.layers.0.self_attn.linears.2"}, d);
    const std::string dec_table = "acembed_extractor.layers.0.pos_enc.embedding.weight";
    for (int n = 0; n < c.n_extra; ++n) {
        const std::string p = "acembed_extractor.layers." + std::to_string(n);
        Layer L;
        add_src(L, p);
        if (c.conf_dec) {  // SrcAttLayer (fanat_conformer_blocks.py:41-60): one norm, Swish FFN of width d_ff
            L.conformer = true;
            pk.ffn_plain(L.w1, L.w2, p + ".feed_forward", c.d_ff, d);
            L.n[0] = pk.norm(p + ".sublayer.norm", d);
        } else {
            pk.ffn(L, p, c.d_decff, d);
            L.n[0] = pk.norm(p + ".sublayer.0.norm", d);
            L.n[1] = pk.norm(p + ".sublayer.1.norm", d);
        }
        m->extra.push_back(L);
    }
    for (int n = 0; n < c.n_self_dec; ++n) {
        const std::string p = "embed_mapper.layers." + std::to_string(n);
        if (c.conf_dec)
            m->sad.push_back(pk.conformer_layer(p, dec_table, c.d_decff, c.dec_kernel, c.dec_max_rel, d, H, 4));
        else
            m->sad.push_back(self_layer(p, "self_attn", c.d_decff, 2));
    }
    for (int n = 0; n < c.n_mix_dec; ++n) {
        const std::string p = "decoder.layers." + std::to_string(n);
        if (c.conf_dec) {
            Layer L = pk.conformer_layer(p, dec_table, c.d_decff, c.dec_kernel, c.dec_max_rel, d, H, 5);
            add_src(L, p);
            m->mad.push_back(L);
        } else {
            Layer L = self_layer(p, "self_attn", c.d_decff, 3);
            add_src(L, p);
            m->mad.push_back(L);
        }
    }
    if (c.conf_dec && pk.chain_ok(d, c.d_decff) && c.d_decff % 128 == 0) {
        for (int n = 0; n < c.n_self_dec; ++n)
            conf_layer_chains(m->sad[n], "embed_mapper.layers." + std::to_string(n), c.d_decff, false,
                              (c.n_mix_dec == 0 && n + 1 == c.n_self_dec) ? "decoder.norm" : "");
        for (int n = 0; n < c.n_mix_dec; ++n)
            conf_layer_chains(m->mad[n], "decoder.layers." + std::to_string(n), c.d_decff, true,
                              n + 1 == c.n_mix_dec ? "decoder.norm" : "");
    }
    if (m->kv_cols > 0) {
        for (size_t j = 0; j < m->extra.size(); ++j) m->extra[j].kv_slot = (int)j;
        for (size_t j = 0; j < m->mad.size(); ++j) m->mad[j].kv_slot = (int)(m->extra.size() + j);
    }
    if (!lm) m->dec_norm = pk.norm("decoder.norm", d);
    // decoder-side sublayers of the NAT model in execution order, each with the chain that follows its attention
    m->dec_steps.clear();
    if (!c.ast && !c.conf_dec && pk.chain_ok(d, c.d_decff)) {
        struct Sub {
            int stack, layer;
            bool self, ffn;
            std::string p, att, pre, ffn_norm;  // layer prefix, attention module, its pre-norm, the FFN's pre-norm
        };
        std::vector<Sub> subs;
        for (int n = 0; n < c.n_extra; ++n) {
            const std::string p = "acembed_extractor.layers." + std::to_string(n);
            subs.push_back({0, n, false, true, p, p + ".src_attn", p + ".sublayer.0.norm", p + ".sublayer.1.norm"});
        }
        for (int n = 0; n < c.n_self_dec; ++n) {
            const std::string p = "embed_mapper.layers." + std::to_string(n);
            subs.push_back({1, n, true, true, p, p + ".self_attn", p + ".sublayer.0.norm", p + ".sublayer.1.norm"});
        }
        for (int n = 0; n < c.n_mix_dec; ++n) {
            const std::string p = "decoder.layers." + std::to_string(n);
            subs.push_back({2, n, true, false, p, p + ".self_attn", p + ".sublayer.0.norm", ""});
            subs.push_back({2, n, false, true, p, p + ".src_attn", p + ".sublayer.1.norm", p + ".sublayer.2.norm"});
        }
        auto in_proj = [&](const Sub& u) -> std::vector<std::string> {
            if (u.self) return {u.att + ".linears.0", u.att + ".linears.1", u.att + ".linears.2"};
            return {u.att + ".linears.0"};
        };
        for (size_t k = 0; k < subs.size(); ++k) {
            const Sub& u = subs[k];
            DecStep st;
            st.stack = u.stack;
            st.layer = u.layer;
            st.self = u.self;
            const bool final = k + 1 == subs.size();
            st.chain = pk.chain(u.att + ".linears.3", u.ffn_norm, u.p, u.ffn ? c.d_decff : 0,
                                final ? std::string("decoder.norm") : subs[k + 1].pre,
                                final ? std::vector<std::string>{} : in_proj(subs[k + 1]), d);
            if (k == 0 || (u.stack == 2 && u.layer == 0 && u.self)) st.entry = pk.chain("", "", "", 0, u.pre, in_proj(u), d);
            m->dec_steps.push_back(st);
        }
    }
    if (c.ast) {
        const size_t at = pk.reserve((size_t)V * d * 4);
        if (pk.fill) {
            const HostTensor* t = pk.find(lm ? "text_embed.0.lut.weight" : "tgt_embed.0.lut.weight", {V, d});
            if (t) std::memcpy(&pk.host[at], t->data.data(), (size_t)V * d * 4);
        }
        m->tgt_lut = reinterpret_cast<float*>(at);
    } else {
        m->tgt_lut = nullptr;
    }
    if (!lm) m->ctc_gen = pk.generator("ctc_generator.proj", V, d);
    m->att_gen = pk.generator(lm ? "out_generator.proj" : "att_generator.proj", V, d);
    {
        const size_t at = pk.reserve((size_t)m->pe_rows * d * 4);
        if (pk.fill) std::memcpy(&pk.host[at], m->pe_host.data(), (size_t)m->pe_rows * d * 4);
        m->pe = reinterpret_cast<float*>(at);
    }
    if (!pk.missing.empty()) {
        cn_set_error("cn_model_finalize: missing or mis-shaped parameters: " + pk.missing);
        return -1;
    }
    if (m->blob_borrowed) {
        if (m->blob_bytes != pk.off) {
            cn_set_error("cn_model_create_shared: the donor's weight blob has another layout (" + std::to_string(m->blob_bytes) +
                         " vs " + std::to_string(pk.off) + " bytes): the model hyper-parameters must agree");
            return -1;
        }
    } else {
        if (m->blob && (m->blob_bytes != pk.off || m->blob_owner.use_count() > 1)) {  // (a blob others still use is left to them)
            m->blob_owner.reset();
            m->blob = nullptr;
        }
        if (!m->blob) {
            auto owner = std::make_shared<cn_model::BlobOwner>();
            CN_HIP_CHECK(hipMalloc((void**)&owner->p, pk.off));
            m->blob_owner = owner;
            m->blob = owner->p;
        }
        m->blob_bytes = pk.off;
        if (pk.fill) CN_HIP_CHECK(hipMemcpy(m->blob, pk.host.data(), pk.off, hipMemcpyHostToDevice));
    }

    unsigned char* base = m->blob;
    rebase(m->conv1_w, base);
    rebase(m->op16_feat_limit, base);
    rebase(m->conv1_b, base);
    rebase_linear(m->conv2, base);
    rebase(m->conv2_x3w, base);
    rebase(m->conv2_mixw, base);
    rebase(m->conv2_mixq, base);
    rebase(m->conv2_f8w, base);
    rebase(m->conv2_f8q, base);
    rebase(m->linear_f8w, base);
    rebase(m->linear_f8q, base);
    rebase_linear(m->linear_out, base);
    auto rebase_layers = [&](std::vector<Layer>& v) {
        for (auto& L : v) {
            for (int i = 0; i < 5; ++i) rebase_norm(L.n[i], base);
            rebase(L.cf_a.w, base);
            rebase(L.cf_a.tab, base);
            rebase(L.cf_b.w, base);
            rebase(L.cf_b.tab, base);
            rebase(L.cf_c.w, base);
            rebase(L.cf_c.tab, base);
            rebase(L.cf_d.w, base);
            rebase(L.cf_d.tab, base);
            rebase_linear(L.ff2_w1, base);
            rebase_linear(L.ff2_w2, base);
            rebase_linear(L.conv.pw1, base);
            rebase_linear(L.conv.pw2, base);
            rebase(L.conv.dw_w, base);
            rebase(L.conv.dw_b, base);
            rebase(L.conv.gn_w, base);
            rebase(L.conv.gn_b, base);
            rebase(L.pos_proj, base);
            rebase(L.pos_u, base);
            rebase(L.pos_v, base);
            rebase_linear(L.qkv, base);
            rebase_linear(L.self_o, base);
            rebase_linear(L.src_q, base);
            rebase_linear(L.src_kv, base);
            rebase_linear(L.src_o, base);
            rebase_linear(L.w1, base);
            rebase_linear(L.w2, base);
            rebase(L.w1p, base);
            rebase(L.w2p, base);
            rebase(L.wx3, base);
        }
    };
    auto rebase_chain = [&](ChainRef& r) {
        rebase(r.w, base);
        rebase(r.tab, base);
        rebase(r.f8q, base);
    };
    for (auto& r : m->enc_chain) rebase_chain(r);
    rebase_chain(m->enc_entry);
    for (auto& st : m->dec_steps) {
        rebase_chain(st.chain);
        rebase_chain(st.entry);
    }
    rebase_layers(m->enc);
    rebase_layers(m->extra);
    rebase_layers(m->sad);
    rebase_layers(m->mad);
    rebase_norm(m->enc_norm, base);
    rebase_norm(m->dec_norm, base);
    rebase_linear(m->ctc_gen, base);
    rebase_linear(m->att_gen, base);
    rebase(m->pe, base);
    rebase(m->tgt_lut, base);
    return 0;
}

// Bytes every workspace buffer needs for a call of B utterances (Bu: for the buffers with one entry per utterance) whose
// conv1 image has T1 rows and whose encoder has Tp frames, with G alignments per utterance on the decoder side.  The ONE place
// these sizes are written: build_workspace allocates them for the configured maxima, ws_check holds every call against what
// was allocated (round 2's GPU fault was a generator that wrote B x G x U rows into a buffer sized for B x (T' + 1): a
// capacity that only existed implicitly).
struct WsDims {
    size_t B, Bu, T1, Tp, G;
};
typedef std::vector<std::pair<const char*, size_t>> WsList;
void ws_needs(const cn_model* m, const WsDims& v, WsList& out) {
    const cn_config& c = m->cfg;
    const size_t es = m->es;
    const size_t B = v.B, T1 = v.T1, Tp = v.Tp, F1 = m->F1, F2 = m->F2, d = c.d_model, V = c.vocab_size, G = v.G;
    const size_t M0 = B * (Tp + 1);  // decoder rows can reach B*(T'+1) per alignment
    const size_t M = M0 * G;
    const size_t dff = std::max(std::max(c.d_encff, c.d_decff), c.d_ff);  // (d_ff: the conformer extractor's FFN width)
    out.clear();
    out.push_back({"keymask", B * Tp});
    if (c.ast != 2) {  // (the LM has no convolutional front-end)
        out.push_back({"c1", B * (T1 + 2) * (F1 + 2) * d * es});  // room for the zero halo the bf16 conv2 kernel wants
        out.push_back({"c2", B * Tp * F2 * d * es});
    }
    out.push_back({"x", (M + 32) * d * 4});  // + one 32-row block: the chain kernel's blocked layout rounds up
    out.push_back({"xn", M * d * es});
    out.push_back({"qkv", (M + 32) * 3 * d * es});  // (+ 32 rows: the chain kernel writes whole 32-row blocks of its blocked output)
    out.push_back({"ctx", (M + 32) * d * es});  // (+ 32 rows: whole 32-row blocks of the blocked form)
    out.push_back({"hbuf", M * dff * es});
    out.push_back({"enc_h", M0 * d * es});
    out.push_back({"kvm", M0 * 2 * d * es});
    if (m->kv_cols > 0) out.push_back({"kv_all", (M0 + 32) * (size_t)m->kv_cols * es});
    out.push_back({"qd", (M + 32) * d * es});
    out.push_back({"dec_h", M * d * es});
    out.push_back({"xd", (M + 32) * d * 4});
    out.push_back({"xd2", (M + 32) * d * 4});
    out.push_back({"logits", M0 * V * 4});
    out.push_back({"best", M * 4});
    out.push_back({"ctc_maxlp", M0 * 4});
    out.push_back({"shift", M * 4});
    out.push_back({"src_size", v.Bu * G * 4});
    out.push_back({"ylen", v.Bu * G * 4});
    out.push_back({"utt_meta", v.Bu * sizeof(UttMeta)});
    out.push_back({"ymax", 256});
    out.push_back({"intervals", B * G * (Tp + 1) * 16});
    out.push_back({"tok", M * 4});
    out.push_back({"val", M * 4});
    out.push_back({"topk_idx", M0 * 16 * 4});
    out.push_back({"topk_val", M0 * 16 * 4});
    if (c.conf_enc || c.conf_dec) {  // convolution module: pointwise-conv output [M][2d], depthwise output fp32, GroupNorm sums
        out.push_back({"cv_a", M * 2 * d * es});
        out.push_back({"cv_f", M * d * 4});
        out.push_back({"gn_stats", v.Bu * G * 2 * 8});
    }
}

int build_workspace(cn_model* m) {
    if (m->keymask) return 0;
    WsList needs;
    ws_needs(m, {(size_t)m->maxB, (size_t)m->maxU, (size_t)m->maxT1, (size_t)m->maxTp, (size_t)std::max(1, m->cfg.esa_group)}, needs);
    std::map<std::string, void*> at;
    for (auto& kv : needs) {
        void* p = nullptr;
        CN_TRY(dev_alloc(m, &p, kv.second));
        at[kv.first] = p;
        m->ws_cap[kv.first] = kv.second;
    }
    auto get = [&](const char* n) -> void* {
        auto it = at.find(n);
        return it == at.end() ? nullptr : it->second;
    };
    m->keymask = (unsigned char*)get("keymask");
    m->c1 = get("c1");
    m->c2 = get("c2");
    m->x = (float*)get("x");
    m->xn = get("xn");
    m->qkv = get("qkv");
    m->ctx = get("ctx");
    m->hbuf = get("hbuf");
    m->enc_h = get("enc_h");
    m->kvm = get("kvm");
    m->kv_all = get("kv_all");
    m->qd = get("qd");
    m->dec_h = get("dec_h");
    m->xd = (float*)get("xd");
    m->xd2 = (float*)get("xd2");
    m->logits = (float*)get("logits");
    m->best = (int*)get("best");
    m->ctc_maxlp = (float*)get("ctc_maxlp");
    m->shift = (int*)get("shift");
    m->src_size = (int*)get("src_size");
    m->ylen = (int*)get("ylen");
    m->utt_meta = (UttMeta*)get("utt_meta");
    m->ymax = (int*)get("ymax");
    m->intervals = (int*)get("intervals");
    m->tok = (int*)get("tok");
    m->val = (float*)get("val");
    m->topk_idx = (int*)get("topk_idx");
    m->topk_val = (float*)get("topk_val");
    m->cv_a = get("cv_a");
    m->cv_f = (float*)get("cv_f");
    m->gn_stats = (double*)get("gn_stats");
    CN_HIP_CHECK(hipHostMalloc((void**)&m->ymax_pinned, 64, hipHostMallocDefault));
    CN_HIP_CHECK(hipHostMalloc((void**)&m->ymax_ring, 64, hipHostMallocDefault));
    bool range_guard = false;
#ifdef CN_OP16_F16
    range_guard = true;
#else
    range_guard = conv2_mix_applies(m->prec, m->cfg.d_model, m->cfg.d_model) && m->cfg.ast != 2;
#endif
    if (range_guard) {
        CN_HIP_CHECK(hipHostMalloc((void**)&m->op16_fault, 64, hipHostMallocMapped));
        *m->op16_fault = 0;
    }
    return 0;
}

// Every buffer of the workspace against what a call of (B utterances, T frames, G alignments per utterance) needs: an error,
// never a write past a buffer.  The workspace is an AREA (max_batch x max_frames): more, shorter utterances fit as well.
int ws_check(cn_model* m, int B, int T, int G, const char* what) {
    const int T1 = (T - 1) / 2 + 1, Tp = (T1 - 1) / 2 + 1;
    WsList needs;
    ws_needs(m, {(size_t)B, (size_t)B, (size_t)T1, (size_t)Tp, (size_t)std::max(1, G)}, needs);
    for (auto& kv : needs) {
        auto it = m->ws_cap.find(kv.first);
        if (it == m->ws_cap.end() || kv.second > it->second) {
            cn_set_error(std::string(what) + ": workspace buffer '" + kv.first + "' holds " +
                         std::to_string(it == m->ws_cap.end() ? 0 : it->second) + " bytes, the call (B=" + std::to_string(B) + " T=" +
                         std::to_string(T) + " alignments=" + std::to_string(G) + ") needs " + std::to_string(kv.second) +
                         " (workspace: max_batch " + std::to_string(m->maxB) + " x max_frames " + std::to_string(m->maxT) + ")");
            return -1;
        }
    }
    if (Tp + 1 > m->pe_rows) {
        cn_set_error(std::string(what) + ": more subsampled frames than the positional table has rows");
        return -1;
    }
    return 0;
}

int capture(cn_model* m, const char* name, const void* src, bool model_prec, int dtype, std::vector<int64_t> shape,
            hipStream_t s) {
    size_t n = 1;
    for (auto v : shape) n *= (size_t)v;
    const size_t esz = dtype == CN_DTYPE_U8 ? 1 : (dtype == CN_DTYPE_F64 ? 8 : 4);
    Capture& cp = m->captures[name];
    if (cp.bytes < n * esz) {
        if (cp.p) (void)hipFree(cp.p);
        CN_HIP_CHECK(hipMalloc(&cp.p, n * esz));
        cp.bytes = n * esz;
    }
    cp.dtype = dtype;
    cp.shape = shape;
    cp.call = m->call_id;
    if (model_prec)
        CN_TRY(launch_convert_back(m->prec, src, (float*)cp.p, n, s));
    else
        CN_HIP_CHECK(hipMemcpyAsync(cp.p, src, n * esz, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// building blocks of the pipeline
// ---------------------------------------------------------------------------------------------
int run_linear(cn_model* m, const char* tag, const Linear& l, const void* A, int lda, void* C, int ldc, int c_f32, int M,
               int epi, const float* resid, int ldr, hipStream_t s) {
    ProfScope ps(m, tag, 2.0 * M * l.N * l.K,
                 (double)M * l.K * m->es + (double)l.N * l.K * m->es + (double)M * l.N * (c_f32 ? 4 : m->es), s);
    static const bool no_proj_x3 = cn_exp_env("CASSNAT_NO_PROJ_X3") != nullptr;
    if (l.px3 && m->prec == CN_PREC_X3 && !no_proj_x3 && (epi == 0 || epi == CN_EPI_RESID) && (epi == 0 || c_f32) && lda % 32 == 0 &&
        (c_f32 || ldc % 32 == 0) && (long long)M * std::max(ldc, ldr) * 4 < (1ll << 31)) {
        ProjX3Args a;
        a.A = A;
        a.lda = lda;
        a.wp = l.px3;
        a.bias = l.b;
        a.C = C;
        a.ldc = ldc;
        a.c_f32 = c_f32;
        a.resid = epi == CN_EPI_RESID ? resid : nullptr;
        a.ldr = ldr;
        a.M = M;
        a.N = l.N;
        return launch_proj_x3(a, s);
    }
    GemmArgs g;
    g.A = A;
    g.lda = lda;
    g.W = l.W;
    g.bias = l.b;
    g.C = C;
    g.ldc = ldc;
    g.c_f32 = c_f32;
    g.M = M;
    g.N = l.N;
    g.K = l.K;
    g.epi = epi;
    g.resid = resid;
    g.ldr = ldr;
    return launch_gemm(m->prec, g, s);
}

int run_ln(cn_model* m, const Norm& n, const float* x, void* y, int M, hipStream_t s) {
    ProfScope ps(m, "layernorm", 0, (double)M * m->cfg.d_model * (4 + m->es), s);
    return launch_layernorm(m->prec, x, n.a, n.b, y, 0, M, m->cfg.d_model, 1e-6f, s);
}

// x += FFN(LN(x)); when `next` is given, next_out <- LN_next(x) in model precision (fused into the FFN kernel on
// the bf16/d256 path, a separate LayerNorm launch otherwise)
int run_ffn(cn_model* m, const Layer& L, const Norm& n, float* x, int M, const Norm* next, void* next_out,
            hipStream_t s) {
    const int d = m->cfg.d_model;
    if (L.wx3) {  // split-bf16 engine: LN + w_1 + ReLU + w_2 + residual (+ next LN) in one launch, 3 MFMAs per product
        ProfScope ps(m, "ffn_fused_x3", 4.0 * M * (double)L.w1.N * d,
                     (double)M * d * 8 + 2.0 * L.w1.N * d * 4 + (next ? (double)M * d * 4 : 0.0), s);
        FfnX3Args a;
        a.x = x;
        a.ln_a = n.a;
        a.ln_b = n.b;
        a.wst = L.wx3;
        a.mix = ffn_mix_applies();
        a.b1 = L.w1.b;
        a.b2 = L.w2.b;
        if (next) {
            a.nln_a = next->a;
            a.nln_b = next->b;
            a.xn_out = next_out;
        }
        a.M = M;
        a.d = d;
        a.dff = L.w1.N;
        return launch_ffn_x3(a, s);
    }
    if (L.w1p) {
        ProfScope ps(m, "ffn_fused", 4.0 * M * (double)L.w1.N * d,
                     (double)M * d * 8 + 2.0 * L.w1.N * d * 2 + (next ? (double)M * d * 2 : 0.0), s);
        FfnFusedArgs a;
        a.x = x;
        a.ln_a = n.a;
        a.ln_b = n.b;
        a.w1p = L.w1p;
        a.b1 = L.w1.b;
        a.w2p = L.w2p;
        a.b2 = L.w2.b;
        if (next) {
            a.nln_a = next->a;
            a.nln_b = next->b;
            a.xn_out = next_out;
        }
        a.M = M;
        a.d = d;
        a.dff = L.w1.N;
        return launch_ffn_fused(a, s);
    }
    CN_TRY(run_ln(m, n, x, m->xn, M, s));
    CN_TRY(run_linear(m, "ffn_w1_relu", L.w1, m->xn, d, m->hbuf, L.w1.N, 0, M, CN_EPI_RELU, nullptr, 0, s));
    CN_TRY(run_linear(m, "ffn_w2_resid", L.w2, m->hbuf, L.w1.N, x, d, 1, M, CN_EPI_RESID, x, d, s));
    if (next) CN_TRY(run_ln(m, *next, x, next_out, M, s));
    return 0;
}

// Split-bf16 engine, the row-chain form of a self-attention layer's tail (fused_x3.hip, PRO / TAIL): in ONE launch behind the
// attention kernel  x += Wo . ctx + bo;  x += FFN(LN1 x);  then either LN_next(x) -> next_out (split-bf16 rows) or, with `tail`, the
// next attention's projection of it -> tail_out (the activations between the three products never leave LDS / registers).
bool x3_chain_applies(const cn_model* m, const Layer& L, const Linear* tail) {
    return m->prec == CN_PREC_X3 && L.wx3 && L.self_o.px3 && L.self_o.N == 256 && L.self_o.K == 256 &&
           (!tail || (tail->px3 && tail->K == 256 && tail->N % 128 == 0 && tail->N <= 1024));
}
int run_x3_chain(cn_model* m, const Layer& L, const Norm& n1, float* x, int M, const Norm& next, void* next_out, const Linear* tail,
                 void* tail_out, int ld_tail, const char* tag, hipStream_t s) {
    const int d = m->cfg.d_model, tn = tail ? tail->N : 0;
    const double macs = (double)d * d + 2.0 * d * L.w1.N + (double)d * tn;
    ProfScope ps(m, tag, 2.0 * M * macs, (double)M * d * (4 + 8) + (double)M * (tail ? tn : d) * 4 + 4.0 * macs, s);
    FfnX3Args a;
    a.x = x;
    a.ln_a = n1.a;
    a.ln_b = n1.b;
    a.wst = L.wx3;
    a.mix = ffn_mix_applies();
    a.b1 = L.w1.b;
    a.b2 = L.w2.b;
    a.nln_a = next.a;
    a.nln_b = next.b;
    a.xn_out = next_out;
    a.M = M;
    a.d = d;
    a.dff = L.w1.N;
    a.ctx = m->ctx;
    a.ldctx = d;
    a.wo_p = L.self_o.px3;
    a.bo = L.self_o.b;
    if (tail) {
        a.tail_p = tail->px3;
        a.tail_b = tail->b;
        a.tail_out = tail_out;
        a.tail_n = tn;
        a.ld_tail = ld_tail;
    }
    return launch_ffn_x3(a, s);
}

// ctx <- Attn(q, k, v) on the fused [M][3d] projection buffer m->qkv
int run_self_attn_core(cn_model* m, int B, int Lseq, const unsigned char* keymask, const int* klen, int causal,
                       hipStream_t s, bool blocked = false) {
    const int d = m->cfg.d_model, M = B * Lseq;
    AttnArgs a;
    const size_t es = m->es;
    a.Q = m->qkv;
    a.K = (const unsigned char*)m->qkv + (size_t)d * es;
    a.V = (const unsigned char*)m->qkv + (size_t)2 * d * es;
    if (blocked) {  // m->qkv as the row-chain kernel's tail wrote it: a blocked [M][3d] matrix
        a.K = a.V = m->qkv;
        a.q_blocked = a.kv_blocked = 1;
        a.q_col = 0;
        a.k_col = d;
        a.v_col = 2 * d;
        a.q_n = a.kv_n = 3 * d;
        a.o_blocked = 1;  // (blocked in = the row-chain path: its next launch reads ctx as B operands)
    }
    m->ctx_blocked = a.o_blocked != 0;
    a.O = m->ctx;
    a.ldq = a.ldk = a.ldv = 3 * d;
    a.ldo = d;
    a.B = B;
    a.H = m->cfg.n_head;
    a.Lq = a.Lk = Lseq;
    a.keymask = keymask;
    a.klen = klen;
    if (m->ragged && keymask == m->keymask) {  // encoder self-attention of a merged pass
        a.kcap = &m->utt_meta[0].tp;
        a.kcap_stride = (int)(sizeof(UttMeta) / sizeof(int));
    }
    a.causal = causal;
    a.scale = 1.0f / sqrtf((float)(d / m->cfg.n_head));
    ProfScope ps(m, "self_attention", 4.0 * B * a.H * (double)Lseq * Lseq * 64, (double)M * 4 * d * m->es, s);
    return launch_attention(m->prec, a, s);
}

// x += O(Attn(LN(x) Wq, LN(x) Wk, LN(x) Wv))
// `pre` == nullptr: m->xn already holds LN(x) (written by the preceding FFN)
int run_self_attn(cn_model* m, const Layer& L, const Norm* pre, float* x, int B, int Lseq, const unsigned char* keymask,
                  const int* klen, int causal, hipStream_t s) {
    const int d = m->cfg.d_model, M = B * Lseq;
    if (pre) CN_TRY(run_ln(m, *pre, x, m->xn, M, s));
    CN_TRY(run_linear(m, "qkv_proj", L.qkv, m->xn, d, m->qkv, 3 * d, 0, M, 0, nullptr, 0, s));
    CN_TRY(run_self_attn_core(m, B, Lseq, keymask, klen, causal, s));
    CN_TRY(run_linear(m, "out_proj_resid", L.self_o, m->ctx, d, x, d, 1, M, CN_EPI_RESID, x, d, s));
    return 0;
}

// row-chain launch: [x += Wo ctx + bo]; [x += FFN(LN1 x)]; [out <- tail projection of LNn(x) (or LNn(x) itself)].
// `with_next` false drops the norm/projection part of a stream that has one (use_unimask cuts the carry SAD -> MAD).
// x_mode: CHX_* bits - the fp32 stream is read / written row-major or in the kernel's blocked tile layout, or not written
enum { CHX_IN_BLK = 1, CHX_OUT_BLK = 2, CHX_NO_STORE = 4 };
// tag: the profile row of the launch ("row_chain": encoder-side, full-width launches; "row_chain_dec": the decoder side's - a
// tenth of the rows, a quarter of the CUs: bench.py prices the two apart)
int run_chain(cn_model* m, const ChainRef& r, float* x, int M, void* out, int ldo, bool with_next, int x_mode,
              hipStream_t s, void* ln_out = nullptr, int ld_ln = 0, bool out_blocked = false, const char* tag = "row_chain") {
    const int d = m->cfg.d_model;
    const int tail_n = with_next ? r.tail_n : 0;
    const double macs = (r.has_wo ? (double)d * d : 0.0) + 2.0 * d * r.dff + (double)d * tail_n;
    ProfScope ps(m, tag, 2.0 * M * macs, (double)M * d * (8 + 2) + (double)M * ldo * 2 + 2.0 * macs, s);
    ChainArgs a;
    a.x = x;
    a.ctx = r.has_wo ? m->ctx : nullptr;
    a.ldctx = d;
    a.ctx_blocked = m->ctx_blocked ? 1 : 0;  // (as the attention launch before this one wrote it)
    a.wstream = r.w;
    a.tab = r.tab;
    a.out = out;
    a.ldo = ldo;
    a.ln_out = ln_out;
    a.ld_ln = ld_ln;
    a.M = M;
    a.d = d;
    a.dff = r.dff;
    a.tail_n = tail_n;
    a.has_next = with_next && r.has_next;
    a.x_in_blocked = (x_mode & CHX_IN_BLK) != 0;
    a.x_out_blocked = (x_mode & CHX_OUT_BLK) != 0;
    a.store_x = (x_mode & CHX_NO_STORE) == 0;
    a.swish = r.swish ? 1 : 0;
    a.out_blocked = out_blocked ? 1 : 0;
    a.f8 = r.f8q ? 1 : 0;
    a.f8_q = r.f8q;
    return launch_chain(a, s);
}

// ---- encoder layer with e4m3fn products (BASELINE config 5; the layer of encoder.py / transformer_blocks.py) -------------
// Operands carry per-tensor power-of-two scales: the weights' is chosen at pack time (largest that keeps max|w| in range,
// stored beside them), the activations' are fixed - LayerNorm outputs and attention contexts x16 (saturating at 28), the
// ReLU hidden activations x8 (saturating at 56).  Accumulation, bias, residual and LayerNorm stay fp32; attention runs
// on the bf16 Q|K|V the first product writes.  LayerNorm and the first FFN product write their fp8 outputs directly; only
// the attention context needs a quantisation pass.
constexpr float FP8_S_LN = 16.f, FP8_S_CTX = 16.f, FP8_S_HID = 8.f;  // (FP8_S_IMG, the conv1 image's: at the top of the file)
int run_linear_fp8(cn_model* m, const char* tag, const Linear& l, const void* a8, float a_scale, void* C, int ldc, int c_f32,
                   int c_fp8, float c_scale, int M, int epi, const float* resid, int ldr, hipStream_t s) {
    ProfScope ps(m, tag, 2.0 * M * l.N * l.K, (double)M * l.K + (double)l.N * l.K + (double)M * l.N * (c_f32 ? 4 : (c_fp8 ? 1 : 2)), s);
    GemmArgs g;
    g.A = a8;
    g.lda = l.K;
    g.W = l.W8;
    g.bias = l.b;
    g.C = C;
    g.ldc = ldc;
    g.c_f32 = c_f32;
    g.c_fp8 = c_fp8;
    g.c_scale = c_scale;
    g.M = M;
    g.N = l.N;
    g.K = l.K;
    g.epi = epi;
    g.resid = resid;
    g.ldr = ldr;
    g.ab_fp8 = 1;
    g.acc_scale = 1.f / a_scale;
    g.w_inv_scale = l.w8_inv;
    return launch_gemm(CN_PREC_BF16, g, s);
}

int run_enc_layer_fp8(cn_model* m, const Layer& L, float* x, int B, int Tp, hipStream_t s) {
    const int d = m->cfg.d_model, M = B * Tp;
    void* q8 = m->xn;    // [M][d] e4m3fn (the bf16 buffer is twice as large)
    void* h8 = m->hbuf;  // [M][d_ff] e4m3fn
    {
        ProfScope ps(m, "layernorm_fp8", 0, (double)M * d * 5, s);
        CN_TRY(launch_layernorm_fp8(x, L.n[0].a, L.n[0].b, q8, M, d, 1e-6f, FP8_S_LN, s));
    }
    CN_TRY(run_linear_fp8(m, "qkv_proj_fp8", L.qkv, q8, FP8_S_LN, m->qkv, 3 * d, 0, 0, 1.f, M, 0, nullptr, 0, s));
    CN_TRY(run_self_attn_core(m, B, Tp, m->keymask, nullptr, 0, s));
    {
        ProfScope ps(m, "quantize_fp8", 0, (double)M * d * 3, s);
        CN_TRY(launch_quantize_fp8(m->ctx, d, q8, M, d, FP8_S_CTX, s));
    }
    CN_TRY(run_linear_fp8(m, "out_proj_fp8", L.self_o, q8, FP8_S_CTX, x, d, 1, 0, 1.f, M, CN_EPI_RESID, x, d, s));
    {
        ProfScope ps(m, "layernorm_fp8", 0, (double)M * d * 5, s);
        CN_TRY(launch_layernorm_fp8(x, L.n[1].a, L.n[1].b, q8, M, d, 1e-6f, FP8_S_LN, s));
    }
    CN_TRY(run_linear_fp8(m, "ffn_w1_fp8", L.w1, q8, FP8_S_LN, h8, L.w1.N, 0, 1, FP8_S_HID, M, CN_EPI_RELU, nullptr, 0, s));
    CN_TRY(run_linear_fp8(m, "ffn_w2_fp8", L.w2, h8, FP8_S_HID, x, d, 1, 0, 1.f, M, CN_EPI_RESID, x, d, s));
    return 0;
}

// ---- conformer sublayers (fanat_conformer_blocks.py, conformer_related.py, attention.py:68-147) ------------------
// x += scale * W2 . swish(W1 . LN(x) + b1) + ...   (SublayerConnection with the Swish feed-forward, scale 0.5 in the macaron halves)
int run_ffn_swish(cn_model* m, const Linear& w1, const Linear& w2, const Norm& n, float* x, int M, float scale, hipStream_t s) {
    const int d = m->cfg.d_model;
    CN_TRY(run_ln(m, n, x, m->xn, M, s));
    CN_TRY(run_linear(m, "ffn_w1_swish", w1, m->xn, d, m->hbuf, w1.N, 0, M, CN_EPI_SWISH, nullptr, 0, s));
    ProfScope ps(m, "ffn_w2_resid", 2.0 * M * w2.N * w2.K, (double)M * w2.K * m->es + (double)w2.N * w2.K * m->es + (double)M * w2.N * 8, s);
    GemmArgs g;
    g.A = m->hbuf;
    g.lda = w1.N;
    g.W = w2.W;
    g.bias = w2.b;
    g.C = x;
    g.ldc = d;
    g.c_f32 = 1;
    g.M = M;
    g.N = w2.N;
    g.K = w2.K;
    g.epi = CN_EPI_RESID;
    g.resid = x;
    g.ldr = d;
    g.resid_scale = scale;
    return launch_gemm(m->prec, g, s);
}

// x += O(RelAttn(LN(x))) with relative-position scores
// ctx <- relative-position attention on the fused [M][3d] projection buffer m->qkv
int run_rel_attn_core(cn_model* m, const Layer& L, int B, int Lseq, const unsigned char* keymask, const int* klen, hipStream_t s);

int run_rel_self_attn(cn_model* m, const Layer& L, const Norm& n, float* x, int B, int Lseq, const unsigned char* keymask,
                      const int* klen, hipStream_t s) {
    const int d = m->cfg.d_model, M = B * Lseq;
    CN_TRY(run_ln(m, n, x, m->xn, M, s));
    CN_TRY(run_linear(m, "qkv_proj", L.qkv, m->xn, d, m->qkv, 3 * d, 0, M, 0, nullptr, 0, s));
    CN_TRY(run_rel_attn_core(m, L, B, Lseq, keymask, klen, s));
    return run_linear(m, "out_proj_resid", L.self_o, m->ctx, d, x, d, 1, M, CN_EPI_RESID, x, d, s);
}

int run_rel_attn_core(cn_model* m, const Layer& L, int B, int Lseq, const unsigned char* keymask, const int* klen, hipStream_t s) {
    const int d = m->cfg.d_model, M = B * Lseq;
    AttnArgs a;
    const size_t es = m->es;
    a.Q = m->qkv;
    a.K = (const unsigned char*)m->qkv + (size_t)d * es;
    a.V = (const unsigned char*)m->qkv + (size_t)2 * d * es;
    m->ctx_blocked = false;  // (the relative-position kernel writes row-major)
    a.O = m->ctx;
    a.ldq = a.ldk = a.ldv = 3 * d;
    a.ldo = d;
    a.B = B;
    a.H = m->cfg.n_head;
    a.Lq = a.Lk = Lseq;
    a.keymask = keymask;
    a.klen = klen;
    a.scale = 1.0f / sqrtf((float)(d / m->cfg.n_head));
    a.rel_pos = L.pos_proj;
    a.rel_u = L.pos_u;
    a.rel_v = L.pos_v;
    a.rel_R = L.rel_R;
    a.ld_pos = d;
    ProfScope ps(m, "rel_self_attention", 4.0 * B * a.H * (double)Lseq * Lseq * 64, (double)M * 4 * d * m->es, s);
    return launch_attention(m->prec, a, s);
}

// x += ConvModule(LN(x))
int run_conv_module(cn_model* m, const Layer& L, const Norm& n, float* x, int B, int Lseq, hipStream_t s) {
    const int d = m->cfg.d_model, M = B * Lseq;
    CN_TRY(run_ln(m, n, x, m->xn, M, s));
    CN_TRY(run_linear(m, "conv_pointwise1", L.conv.pw1, m->xn, d, m->cv_a, 2 * d, 0, M, 0, nullptr, 0, s));
    {
        ProfScope ps(m, "conv_glu_depthwise_norm", 2.0 * M * d * L.conv.k, (double)M * d * (4 * m->es + 12), s);
        CN_TRY(launch_glu(m->prec, m->cv_a, m->xn, M, d, s));
        CN_TRY(launch_dwconv(m->prec, m->xn, L.conv.dw_w, L.conv.dw_b, m->cv_f, B, Lseq, d, L.conv.k, s));
        CN_TRY(launch_groupnorm_swish(m->prec, m->cv_f, m->gn_stats, L.conv.gn_w, L.conv.gn_b, m->xn, B, Lseq, d, 1e-5f, s));
    }
    return run_linear(m, "conv_pointwise2_resid", L.conv.pw2, m->xn, d, x, d, 1, M, CN_EPI_RESID, x, d, s);
}

// SelfAttLayer, relative branch (fanat_conformer_blocks.py:26-38): ff1 (0.5), attention, convolution, ff2 (0.5)
int run_conformer_self_layer(cn_model* m, const Layer& L, float* x, int B, int Lseq, const unsigned char* keymask,
                             const int* klen, hipStream_t s) {
    const int M = B * Lseq;
    CN_TRY(run_ffn_swish(m, L.w1, L.w2, L.n[0], x, M, 0.5f, s));
    CN_TRY(run_rel_self_attn(m, L, L.n[2], x, B, Lseq, keymask, klen, s));
    CN_TRY(run_conv_module(m, L, L.n[1], x, B, Lseq, s));
    return run_ffn_swish(m, L.ff2_w1, L.ff2_w2, L.n[3], x, M, 0.5f, s);
}

int run_src_attn(cn_model* m, const Layer& L, const Norm* pre, float* x, int B, int U, int Tp, const int* intervals,
                 hipStream_t s);
int run_src_attn_core(cn_model* m, const Layer& L, int B, int U, int Tp, const int* intervals, hipStream_t s, bool q_blocked = false);

// A conformer layer on the row-chain kernel (chains packed by conf_layer_chains in build_weights): A -> relative-position
// attention -> B -> GLU / depthwise conv / GroupNorm + Swish -> C [-> source attention -> D for a mixed-attention layer].
// x_in / x_out: CHX_* layout bits of the residual stream at the layer's ends (blocked between its own launches unless
// `rowmajor`); `final_out`: where the norm packed behind the last chain goes (null: none packed).
int run_conformer_layer_chain(cn_model* m, const Layer& L, float* x, int B, int Lseq, const unsigned char* keymask, const int* klen,
                              bool mixed, int Tp, const int* src_intervals, bool rowmajor, int x_in, int x_out, void* final_out,
                              hipStream_t s) {
    const int d = m->cfg.d_model, M = B * Lseq;
    const int mid_in = rowmajor ? 0 : CHX_IN_BLK, mid_out = rowmajor ? 0 : CHX_OUT_BLK;
    CN_TRY(run_chain(m, L.cf_a, x, M, m->qkv, 3 * d, true, x_in | mid_out, s));
    CN_TRY(run_rel_attn_core(m, L, B, Lseq, keymask, klen, s));
    CN_TRY(run_chain(m, L.cf_b, x, M, m->cv_a, 2 * d, true, mid_in | mid_out, s));
    {
        ProfScope ps(m, "conv_glu_depthwise_norm", 2.0 * M * d * L.conv.k, (double)M * d * (4 * m->es + 12), s);
        CN_TRY(launch_glu(m->prec, m->cv_a, m->xn, M, d, s));
        CN_TRY(launch_dwconv(m->prec, m->xn, L.conv.dw_w, L.conv.dw_b, m->cv_f, B, Lseq, d, L.conv.k, s));
        // (the module's output goes to m->ctx: the next chain's output projection is pointwise conv 2)
        CN_TRY(launch_groupnorm_swish(m->prec, m->cv_f, m->gn_stats, L.conv.gn_w, L.conv.gn_b, m->ctx, B, Lseq, d, 1e-5f, s));
        m->ctx_blocked = false;  // (row-major)
    }
    if (!mixed) return run_chain(m, L.cf_c, x, M, final_out, d, final_out != nullptr, mid_in | x_out, s);
    CN_TRY(run_chain(m, L.cf_c, x, M, m->qd, d, true, mid_in | mid_out, s));
    CN_TRY(run_src_attn_core(m, L, B, Lseq, Tp, src_intervals, s));
    return run_chain(m, L.cf_d, x, M, final_out, d, final_out != nullptr, mid_in | x_out, s);
}

// ctx <- Attn(m->qd, enc_h Wk, enc_h Wv) with the padding mask and (optionally) trigger intervals
int run_src_attn_core(cn_model* m, const Layer& L, int B, int U, int Tp, const int* intervals, hipStream_t s, bool q_blocked) {
    const int d = m->cfg.d_model;
    // (B counts query sets: dec_group of them share the keys / values of one utterance)
    AttnArgs a;
    if (m->kv_ready && L.kv_slot >= 0) {  // projected by the last encoder chain launch
        a.K = (const unsigned char*)m->kv_all + (size_t)L.kv_slot * 2 * d * m->es;
        a.ldk = a.ldv = m->kv_cols;
        if (m->kv_blocked) {  // ... into a blocked [M][kv_cols] matrix
            a.K = m->kv_all;
            a.kv_blocked = 1;
            a.k_col = L.kv_slot * 2 * d;
            a.v_col = a.k_col + d;
            a.kv_n = m->kv_cols;
        }
    } else {
        CN_TRY(run_linear(m, "src_kv_proj", L.src_kv, m->enc_h, d, m->kvm, 2 * d, 0, m->B * Tp, 0, nullptr, 0, s));
        a.K = m->kvm;
        a.ldk = a.ldv = 2 * d;
    }
    a.V = a.kv_blocked ? a.K : (const unsigned char*)a.K + (size_t)d * m->es;
    a.kv_mod = m->dec_group > 1 ? m->B : 0;
    a.Q = m->qd;
    if (q_blocked) {
        a.q_blocked = 1;
        a.q_col = 0;
        a.q_n = d;
        a.o_blocked = 1;
    }
    m->ctx_blocked = a.o_blocked != 0;
    a.O = m->ctx;
    a.ldq = d;
    a.ldo = d;
    a.B = B;
    a.H = m->cfg.n_head;
    a.Lq = U;
    a.Lk = Tp;
    a.keymask = m->keymask;
    if (m->ragged) {
        a.kcap = &m->utt_meta[0].tp;
        a.kcap_stride = (int)(sizeof(UttMeta) / sizeof(int));
    }
    a.intervals = intervals;
    a.iv_stride = Tp + 1;
    a.scale = 1.0f / sqrtf((float)(d / m->cfg.n_head));
    ProfScope ps(m, "src_attention", 4.0 * B * a.H * (double)U * Tp * 64,
                 ((double)B * U * 2 * d + (double)B * Tp * 2 * d) * m->es, s);
    return launch_attention(m->prec, a, s);
}

// x += O(Attn(LN(x) Wq, mem Wk, mem Wv))
int run_src_attn(cn_model* m, const Layer& L, const Norm* pre, float* x, int B, int U, int Tp, const int* intervals,
                 hipStream_t s) {
    const int d = m->cfg.d_model;
    if (pre) CN_TRY(run_ln(m, *pre, x, m->xn, B * U, s));
    CN_TRY(run_linear(m, "src_q_proj", L.src_q, m->xn, d, m->qd, d, 0, B * U, 0, nullptr, 0, s));
    CN_TRY(run_src_attn_core(m, L, B, U, Tp, intervals, s));
    CN_TRY(run_linear(m, "out_proj_resid", L.src_o, m->ctx, d, x, d, 1, B * U, CN_EPI_RESID, x, d, s));
    return 0;
}

// generator tail: argmax + max log-prob per row.  Fused kernel (no logits tensor) unless full rows are needed.
int run_generator(cn_model* m, const Linear& g, const void* h, int M, int* arg, float* maxlp, bool need_rows,
                  hipStream_t s) {
    const int d = m->cfg.d_model, V = m->cfg.vocab_size;
    if (g.gm_w && !need_rows) {
        const bool x3 = m->prec == CN_PREC_X3;
        ProfScope ps(m, "generator_argmax_fused", (x3 ? 6.0 : 2.0) * M * V * d, ((double)M * d + (double)V * d) * (x3 ? 4 : 2), s);
        GenmaxArgs a;
        a.x3 = x3;
        a.h = h;
        a.wp = g.gm_w;
        a.bp = g.gm_b;
        a.arg = arg;
        a.maxlp = maxlp;
        a.M = M;
        a.V = V;
        a.d = d;
        return launch_genmax(a, s);
    }
    // the logits buffer holds B x (T' + 1) rows (one alignment per utterance); the decoder side of an ESA group brings G times
    // that: the rows go through it in chunks (the engines without the fused kernel - fp32, bf16x3 - wrote past the buffer here)
    const size_t cap_rows = (size_t)m->maxB * (m->maxTp + 1);
    if ((size_t)M > cap_rows && need_rows) {
        cn_set_error("generator: full log-probability rows (capture / beam_width > 1) of more rows than one alignment per utterance");
        return -1;
    }
    for (size_t r0 = 0; r0 < (size_t)M; r0 += cap_rows) {
        const int rows = (int)std::min(cap_rows, (size_t)M - r0);
        CN_TRY(run_linear(m, "generator_proj", g, (const unsigned char*)h + r0 * d * m->es, d, m->logits, V, 1, rows, 0, nullptr, 0, s));
        ProfScope ps(m, "logsoftmax_argmax", 0, (double)rows * V * 4, s);
        CN_TRY(launch_logsoftmax_argmax(m->logits, rows, V, V, arg + r0, maxlp ? maxlp + r0 : nullptr, need_rows ? 1 : 0, s));
    }
    return 0;
}

// `area`: the call may carry more utterances than max_batch (or longer ones than max_frames) as long as every buffer holds
// it (the greedy NAT decode, whose kernels take any B x T; the other entry points keep the configured bounds)
int check_call(cn_model* m, int B, int T, int F, bool area = false) {
    if (!m || !m->finalized) {
        cn_set_error("model not finalized");
        return -1;
    }
    if (F != m->cfg.input_size || B < 1 || T < 1 || (!area && (B > m->maxB || T > m->maxT))) {
        cn_set_error("decode: batch/frames/feature dims outside the configured workspace (B=" + std::to_string(B) +
                     " T=" + std::to_string(T) + " F=" + std::to_string(F) + ")");
        return -1;
    }
    return ws_check(m, B, T, 1, "decode");
}

int stage_encode(cn_model* m, const float* feats, int B, int T, int F, const cn_decode_opts* o, hipStream_t s);

// src_embed + encoder + ctc_generator + best_path_align + align_to_mask  (cassnat.py:431-468)
int stage_encode_align(cn_model* m, const float* feats, const float* ratio, int B, int T, int F,
                       const cn_decode_opts* o, hipStream_t s) {
    CN_TRY(stage_encode(m, feats, B, T, F, o, s));
    const cn_config& c = m->cfg;
    const int Tp = m->Tp, M = B * Tp;
    const bool cap = o->capture != 0;
    // the alignment needs the arg-max only; the best path's log-probabilities are produced when a capture asks for rows
    m->ctc_maxlp_valid = cap || !m->ctc_gen.gm_w;
    CN_TRY(run_generator(m, m->ctc_gen, m->enc_h, M, m->best, m->ctc_maxlp_valid ? m->ctc_maxlp : nullptr, cap, s));
    if (cap) CN_TRY(capture(m, "ctc_out", m->logits, false, CN_DTYPE_F32, {B, Tp, c.vocab_size}, s));
    AlignArgs al;
    al.best = m->best;
    al.keymask = m->keymask;
    al.size_ratio = ratio;
    al.B = B;
    al.Tp = Tp;
    al.blank = o->padding_idx;
    al.left = o->left_trigger;
    al.right = o->right_trigger;
    al.shift = m->shift;
    al.src_size = m->src_size;
    al.ylen = m->ylen;
    al.ymax = m->ymax;
    al.intervals = m->intervals;
    al.utt_meta = m->ragged ? m->utt_meta : nullptr;
    al.no_trigger = o->no_trigger != 0;
    {
        ProfScope ps(m, "ctc_align", 0, (double)B * Tp * 9 + (double)B * (Tp + 1) * 16, s);
        CN_TRY(launch_ctc_align(al, s));
    }
    return 0;
}

// src_embed + encoder (shared by the NAST and AST paths): features -> enc_h (model precision), keymask
int stage_encode(cn_model* m, const float* feats, int B, int T, int F, const cn_decode_opts* o, hipStream_t s) {
    const cn_config& c = m->cfg;
    const int d = c.d_model;
    const int T1 = (T - 1) / 2 + 1, Tp = (T1 - 1) / 2 + 1, F1 = m->F1, F2 = m->F2;
    const int M = B * Tp;
    const bool cap = o->capture != 0;
    m->B = B;
    ++m->call_id;
    if (!m->ragged_next) m->ragged = false;  // (only cn_decode_nast_merged announces per-utterance records, for its own call)
    m->ragged_next = false;
    m->u_predicted = false;
    m->dec_group = 1;
    m->kv_ready = false;
    m->T = T;
    m->T1 = T1;
    m->Tp = Tp;
    m->U = 0;
    CN_TRY(launch_keymask(feats, B, T, F, Tp, 4, (float)o->padding_idx, m->keymask, s));
    // (fp16 engines: the features against the half range; split-bf16 engines: against the e4m3 range of conv2's mixed arithmetic)
    if (m->op16_fault && m->op16_feat_limit) CN_TRY(launch_feature_range(feats, (size_t)B * T * F, m->op16_feat_limit, m->op16_fault, s));
    // conv1 writes its image with a zero halo when the LDS-DMA conv2 kernel consumes it (captures want the plain image)
    // (the split-bf16 engine: two bordered bf16 planes, hi and lo, for the same kernel's X3 form)
    // (the split-bf16 engine's planes in the MIX arithmetic of conv2.hip where it applies: half-precision hi values + two e4m3 planes)
    const bool mix_planes = !cap && m->conv2_mixw && conv2_mix_applies(m->prec, d, d);
    const bool x3_planes = !mix_planes && !cap && m->conv2_x3w && conv2_x3_applies(m->prec, d, d);
    // (the fp8 engine: an e4m3 image for the same kernel's F8 form - config 5's "fp8 MFMA encoder GEMMs" include the largest one)
    // (also under capture - the accuracy of the mode is measured on captures; the e4m3 image itself is then not captured)
    const bool f8_img = m->fp8_enc && m->conv2_f8w && conv2_f8_applies(d, d);
    const bool f8_lin = f8_img && m->linear_f8w && linear256_f8_applies(d, F2 * d);  // conv2 then hands its rows on as e4m3 too
    const int halo = ((!cap && conv2_dma_applies(m->prec, d, d)) || x3_planes || mix_planes || f8_img) ? 1 : 0;
    {
        // the halo cells of the image buffer are zero already when the previous haloed image had this very shape (and nothing
        // else wrote the buffer since): conv1 then writes the interior only (halo mode 2)
        const bool same = halo && m->c1_halo_B == B && m->c1_halo_T1 == T1;
        ProfScope ps(m, "conv1", 2.0 * 9 * B * T1 * F1 * d, (double)B * T * F * 4 + (double)B * T1 * F1 * d * m->es, s);
        const UttMeta* um = m->ragged ? m->utt_meta : nullptr;
        if (mix_planes)
            CN_TRY(launch_conv1_mixplanes(feats, m->conv1_w, m->conv1_b, m->c1, B, T, F, T1, F1, d, same ? 2 : 1, std::ldexp(1.f, MIX_LG_AL),
                                          std::ldexp(1.f, MIX_LG_AQ), s, um));
        else if (x3_planes)
            CN_TRY(launch_conv1_planes(feats, m->conv1_w, m->conv1_b, m->c1, B, T, F, T1, F1, d, same ? 2 : 1, s, um));
        else if (f8_img)
            CN_TRY(launch_conv1_f8(feats, m->conv1_w, m->conv1_b, m->c1, B, T, F, T1, F1, d, same ? 2 : 1, FP8_S_IMG, s, um));
        else if (halo && m->prec == CN_PREC_BF16 && conv1_bordered_bf16_applies(d, F1))
            CN_TRY(launch_conv1_bordered_bf16(feats, m->conv1_w, m->conv1_b, m->c1, B, T, F, T1, F1, d, s, um));
        else
            CN_TRY(launch_conv1(m->prec, feats, m->conv1_w, m->conv1_b, m->c1, B, T, F, T1, F1, d, same ? 2 : halo, s, um));
        m->c1_halo_B = halo ? B : -1;
        m->c1_halo_T1 = halo ? T1 : -1;
    }
    if (cap && !f8_img) CN_TRY(capture(m, "conv1", m->c1, true, CN_DTYPE_F32, {B, T1, F1, d}, s));
    {
        GemmArgs g;
        g.A = m->c1;
        g.conv_halo = halo;
        g.W = m->conv2.W;
        g.bias = m->conv2.b;
        g.C = m->c2;
        g.ldc = d;
        g.M = M * F2;
        g.N = d;
        g.K = 9 * d;
        g.epi = CN_EPI_RELU;
        g.conv = 1;
        g.cB = B;
        g.cT1 = T1;
        g.cF1 = F1;
        g.cC = d;
        g.cT2 = Tp;
        g.cF2 = F2;
        ProfScope ps(m, "conv2", 2.0 * g.M * g.N * g.K,
                     ((double)B * T1 * F1 * d + (double)g.N * g.K + (double)g.M * g.N) * m->es, s);
        if (mix_planes) {
            const size_t wn = (size_t)d * 9 * d;
            const unsigned char* w = (const unsigned char*)m->conv2_mixw;
            CN_TRY(launch_conv2_mix(m->c1, w, w + 2 * wn, w + 3 * wn, m->conv2_mixq, m->conv2.b, m->c2, B, T1, F1, Tp, F2, s));
        } else if (x3_planes) {
            const size_t img = (size_t)B * (T1 + 2) * (F1 + 2) * d * 2, wpl = (size_t)d * 9 * d * 2;
            CN_TRY(launch_conv2_x3(m->c1, (const unsigned char*)m->c1 + img, m->conv2_x3w, (const unsigned char*)m->conv2_x3w + wpl,
                                   m->conv2.b, m->c2, B, T1, F1, Tp, F2, s));
        } else if (f8_img) {
            CN_TRY(launch_conv2_f8(m->c1, m->conv2_f8w, m->conv2_f8q, m->conv2.b, m->c2, B, T1, F1, Tp, F2, s, f8_lin ? FP8_S_IMG : 0.f));
        } else {
            CN_TRY(launch_gemm(m->prec, g, s));
        }
    }
    if (cap && !f8_lin) CN_TRY(capture(m, "conv2", m->c2, true, CN_DTYPE_F32, {B, Tp, F2, d}, s));
    if (f8_lin) {
        ProfScope ps(m, "linear_out_embed", 2.0 * M * d * (double)F2 * d, (double)M * F2 * d + (double)d * F2 * d + (double)M * d * 4, s);
        CN_TRY(launch_linear256_f8(m->c2, m->linear_f8w, m->linear_f8q, m->linear_out.b, m->x, M, F2 * d, sqrtf((float)d),
                                   c.conf_enc ? nullptr : m->pe, Tp, s));
    } else {
        GemmArgs g;
        g.A = m->c2;
        g.lda = F2 * d;
        g.W = m->linear_out.W;
        g.bias = m->linear_out.b;
        g.C = m->x;
        g.ldc = d;
        g.c_f32 = 1;
        g.M = M;
        g.N = d;
        g.K = F2 * d;
        g.epi = CN_EPI_EMBED;
        g.pe = c.conf_enc ? nullptr : m->pe;  // relative positions: x * sqrt(d) only (embedding.py:48-58, 119)
        g.pe_period = Tp;
        g.scale = sqrtf((float)d);
        ProfScope ps(m, "linear_out_embed", 2.0 * g.M * g.N * g.K,
                     ((double)g.M * g.K + (double)g.N * g.K) * m->es + (double)g.M * g.N * 4, s);
        CN_TRY(launch_gemm(m->prec, g, s));
    }
    if (cap) CN_TRY(capture(m, "x_embed", m->x, false, CN_DTYPE_F32, {B, Tp, d}, s));
    static const bool no_chain_c = cn_exp_env("CASSNAT_NO_CHAIN") != nullptr;
    if (c.conf_enc && !m->enc.empty() && m->enc[0].cf_a.w && !no_chain_c) {
        // conformer encoder on the row-chain kernel: per layer A -> relative-position attention -> B -> GLU / depthwise
        // conv / GroupNorm + Swish -> C (see build_weights); the residual stream stays in the blocked layout in between
        for (size_t n = 0; n < m->enc.size(); ++n) {
            const bool last = n + 1 == m->enc.size();
            const int xin = (cap || n == 0) ? 0 : CHX_IN_BLK, xout = cap ? 0 : (last ? CHX_NO_STORE : CHX_OUT_BLK);
            CN_TRY(run_conformer_layer_chain(m, m->enc[n], m->x, B, Tp, m->keymask, nullptr, false, 0, nullptr, cap, xin, xout,
                                             last ? m->enc_h : nullptr, s));
            if (cap) CN_TRY(capture(m, ("enc_layer" + std::to_string(n)).c_str(), m->x, false, CN_DTYPE_F32, {B, Tp, d}, s));
        }
        if (cap) CN_TRY(capture(m, "enc_h", m->enc_h, true, CN_DTYPE_F32, {B, Tp, d}, s));
        return 0;
    }
    if (c.conf_enc) {  // conformer encoder (fanat_conformer_blocks.py:141-170)
        for (size_t n = 0; n < m->enc.size(); ++n) {
            CN_TRY(run_conformer_self_layer(m, m->enc[n], m->x, B, Tp, m->keymask, nullptr, s));
            if (cap) CN_TRY(capture(m, ("enc_layer" + std::to_string(n)).c_str(), m->x, false, CN_DTYPE_F32, {B, Tp, d}, s));
        }
        CN_TRY(run_ln(m, m->enc_norm, m->x, m->enc_h, M, s));
        if (cap) CN_TRY(capture(m, "enc_h", m->enc_h, true, CN_DTYPE_F32, {B, Tp, d}, s));
        return 0;
    }
    static const bool no_chain = cn_exp_env("CASSNAT_NO_CHAIN") != nullptr;
    // BASELINE config 5.  With row chains (d_model 256, d_ff % 256 == 0) the feed-forward products - 80 % of a layer's
    // multiply-adds - run on e4m3 operands inside the chain kernel at twice the bf16 rate (chain.hip, F8 form) and the layer
    // goes down the bf16 engine's path below; without them, the four products of every layer as separate e4m3 launches
    if (m->fp8_enc && (m->enc_chain.size() != m->enc.size() || no_chain)) {
        for (size_t n = 0; n < m->enc.size(); ++n) {
            CN_TRY(run_enc_layer_fp8(m, m->enc[n], m->x, B, Tp, s));
            if (cap) CN_TRY(capture(m, ("enc_layer" + std::to_string(n)).c_str(), m->x, false, CN_DTYPE_F32, {B, Tp, d}, s));
        }
        CN_TRY(run_ln(m, m->enc_norm, m->x, m->enc_h, M, s));
        if (cap) CN_TRY(capture(m, "enc_h", m->enc_h, true, CN_DTYPE_F32, {B, Tp, d}, s));
        return 0;
    }
    const bool chain = !m->enc.empty() && m->enc_chain.size() == m->enc.size() && !no_chain;
    // the projections a chain launch writes for the attention kernel (Q|K|V, and the decoder side's K|V) go out in the blocked
    // layout: 1-KiB store instructions instead of thirty-two 32-byte row segments (the tail phase was store-bound)
    static const bool blk = cn_exp_env("CASSNAT_NO_BLOCKED_QKV") == nullptr;
    m->kv_blocked = false;
    if (chain) {  // bf16 / d_model 256: LN + QKV of layer 0 (an entry chain launch: no FFN, x left alone), then per layer
                  // attention -> row-chain kernel
        if (m->enc_entry.w) {
            CN_TRY(run_chain(m, m->enc_entry, m->x, M, m->qkv, 3 * d, true, CHX_NO_STORE, s, nullptr, 0, blk));
        } else {
            CN_TRY(run_ln(m, m->enc[0].n[0], m->x, m->xn, M, s));
            CN_TRY(run_linear(m, "qkv_proj", m->enc[0].qkv, m->xn, d, m->qkv, 3 * d, 0, M, 0, nullptr, 0, s));
        }
    }
    // split-bf16 engine: every encoder layer in the row-chain form (all layers or none: the Q|K|V hand-over is layer to layer)
    static const bool no_x3_chain = cn_exp_env("CASSNAT_NO_X3_CHAIN") != nullptr;
    bool x3_rows = !chain && !cap && !no_x3_chain && m->prec == CN_PREC_X3 && !m->enc.empty() && m->enc[0].qkv.px3;
    for (size_t n = 0; x3_rows && n < m->enc.size(); ++n)
        x3_rows = x3_chain_applies(m, m->enc[n], n + 1 < m->enc.size() ? &m->enc[n + 1].qkv : nullptr);
    for (size_t n = 0; n < m->enc.size(); ++n) {
        const Layer& L = m->enc[n];
        const bool last = n + 1 == m->enc.size();
        if (chain) {
            CN_TRY(run_self_attn_core(m, B, Tp, m->keymask, nullptr, 0, s, (n > 0 || m->enc_entry.w) && blk));
            // between chain launches the residual stream lives in the kernel's blocked layout; the last layer's x is
            // consumed by nobody (enc_h is the output).  Captures read x row-major.
            const int xm = cap ? 0 : ((n > 0 ? CHX_IN_BLK : 0) | (last ? CHX_NO_STORE : CHX_OUT_BLK));
            if (last && m->kv_cols > 0) {  // enc_h and, from the same registers, every decoder-side layer's K|V
                CN_TRY(run_chain(m, m->enc_chain[n], m->x, M, m->kv_all, m->kv_cols, true, xm, s, m->enc_h, d, blk));
                m->kv_ready = true;
                m->kv_blocked = blk;
            } else {
                CN_TRY(run_chain(m, m->enc_chain[n], m->x, M, last ? m->enc_h : m->qkv, last ? d : 3 * d, true, xm, s, nullptr, 0,
                                 blk && !last));
            }
        } else if (x3_rows) {
            // split-bf16 engine: attention, then ONE launch for the out-projection, the feed-forward sublayer and the next layer's
            // Q|K|V (the last layer: enc_h).  (A capture run keeps the unfused kernels: the fused path's reference in the tests.)
            if (n == 0) {
                CN_TRY(run_ln(m, L.n[0], m->x, m->xn, M, s));
                CN_TRY(run_linear(m, "qkv_proj", L.qkv, m->xn, d, m->qkv, 3 * d, 0, M, 0, nullptr, 0, s));
            }
            CN_TRY(run_self_attn_core(m, B, Tp, m->keymask, nullptr, 0, s));
            if (last)
                CN_TRY(run_x3_chain(m, L, L.n[1], m->x, M, m->enc_norm, m->enc_h, nullptr, nullptr, 0, "row_chain_x3", s));
            else
                CN_TRY(run_x3_chain(m, L, L.n[1], m->x, M, m->enc[n + 1].n[0], nullptr, &m->enc[n + 1].qkv, m->qkv, 3 * d, "row_chain_x3", s));
        } else {
            CN_TRY(run_self_attn(m, L, n == 0 ? &L.n[0] : nullptr, m->x, B, Tp, m->keymask, nullptr, 0, s));
            CN_TRY(run_ffn(m, L, L.n[1], m->x, M, last ? &m->enc_norm : &m->enc[n + 1].n[0], last ? m->enc_h : m->xn, s));
        }
        if (cap) CN_TRY(capture(m, ("enc_layer" + std::to_string(n)).c_str(), m->x, false, CN_DTYPE_F32, {B, Tp, d}, s));
    }
    if (m->enc.empty()) CN_TRY(run_ln(m, m->enc_norm, m->x, m->enc_h, M, s));
    if (cap) CN_TRY(capture(m, "enc_h", m->enc_h, true, CN_DTYPE_F32, {B, Tp, d}, s));
    return 0;
}

// acembed_extractor + embed_mapper + decoder + att_generator + greedy finish  (cassnat.py:475-497, 574-637)
int stage_decode_tail(cn_model* m, int U, const cn_decode_opts* o, int32_t* hyp, int hyp_stride, int32_t* hyp_len,
                      double* score, hipStream_t s);

int stage_decode(cn_model* m, int U, const cn_decode_opts* o, int32_t* hyp, int hyp_stride, int32_t* hyp_len,
                 double* score, hipStream_t s) {
    const cn_config& c = m->cfg;
    const int d = c.d_model, B = m->B * m->dec_group, Tp = m->Tp, MU = B * U;
    const bool cap = o->capture != 0;
    m->U = U;
    if (U > m->pe_rows || U < 1 || U > Tp + 1) {
        cn_set_error("decode: token count exceeds the positional table / the frames of the batch");
        return -1;
    }
    // every decoder-side buffer against B x alignments-per-utterance x rows (host-side, before anything is launched)
    CN_TRY(ws_check(m, m->B, m->T, m->dec_group, "decoder side"));
    CN_TRY(launch_fill_queries(m->pe, m->xd, B, U, d, s));
    // The LayerNorm that follows an FFN is produced by that FFN (-> m->xn); `pending` says whether the next
    // sublayer may skip its own LayerNorm.  use_unimask shifts the stream between SAD and MAD, so no carry there.
    const bool uni = o->use_unimask != 0;
    auto first_norm_after = [&](int stage, size_t idx) -> const Norm* {  // stage 0 extractor, 1 SAD, 2 MAD
        if (stage == 0 && idx + 1 < m->extra.size()) return &m->extra[idx + 1].n[0];
        if (stage <= 0 && !m->sad.empty()) return &m->sad[0].n[0];
        if (stage == 1 && idx + 1 < m->sad.size()) return &m->sad[idx + 1].n[0];
        if (stage <= 1) return (uni || m->mad.empty()) ? nullptr : &m->mad[0].n[0];
        if (idx + 1 < m->mad.size()) return &m->mad[idx + 1].n[0];
        return nullptr;
    };
    if (c.conf_dec) {
        if (uni) {
            cn_set_error("decode: use_unimask is not defined for the conformer decoder (the reference indexes a tuple there)");
            return -1;
        }
        float* x = m->xd;
        // ConAcExtra (fanat_conformer_blocks.py:41-60, 172-186): attention on the raw position queries, output projection
        // WITHOUT residual or pre-norm, scaled by sqrt(d); then x += FFN_swish(LN x)
        for (size_t i = 0; i < m->extra.size(); ++i) {
            const Layer& L = m->extra[i];
            CN_TRY(launch_convert(m->prec, x, m->xn, (size_t)MU * d, s));
            CN_TRY(run_linear(m, "src_q_proj", L.src_q, m->xn, d, m->qd, d, 0, MU, 0, nullptr, 0, s));
            CN_TRY(run_src_attn_core(m, L, B, U, Tp, m->intervals, s));
            {
                ProfScope ps(m, "out_proj_scaled", 2.0 * MU * d * d, (double)MU * d * (m->es + 4), s);
                GemmArgs g;
                g.A = m->ctx;
                g.lda = d;
                g.W = L.src_o.W;
                g.bias = L.src_o.b;
                g.C = x;
                g.ldc = d;
                g.c_f32 = 1;
                g.M = MU;
                g.N = d;
                g.K = d;
                g.epi = CN_EPI_EMBED;
                g.pe = nullptr;
                g.scale = sqrtf((float)d);
                CN_TRY(launch_gemm(m->prec, g, s));
            }
            CN_TRY(run_ffn_swish(m, L.w1, L.w2, L.n[0], x, MU, 1.0f, s));
        }
        if (cap) CN_TRY(capture(m, "ac_embed", x, false, CN_DTYPE_F32, {B, U, d}, s));
        static const bool no_chain_d = cn_exp_env("CASSNAT_NO_CHAIN") != nullptr;
        const bool dchain = !no_chain_d && ((!m->sad.empty() && m->sad[0].cf_a.w) || (!m->mad.empty() && m->mad[0].cf_a.w));
        if (dchain) {
            // the self- and mixed-attention conformer layers on the row-chain kernel (the extractor above stays generic);
            // the stream is row-major where something else reads it (extractor output, the pred_embed capture), blocked otherwise
            const size_t ns = m->sad.size(), nm = m->mad.size();
            for (size_t i = 0; i < ns + nm; ++i) {
                const bool mixed = i >= ns, last = i + 1 == ns + nm;
                const Layer& L = mixed ? m->mad[i - ns] : m->sad[i];
                const bool rm_in = cap || i == 0, rm_out = cap;
                const int xin = rm_in ? 0 : CHX_IN_BLK, xout = last ? CHX_NO_STORE : (rm_out ? 0 : CHX_OUT_BLK);
                CN_TRY(run_conformer_layer_chain(m, L, x, B, U, nullptr, m->ylen, mixed, Tp, o->src_trigger ? m->intervals : nullptr,
                                                 cap, xin, xout, last ? m->dec_h : nullptr, s));
                if (cap && !mixed && i + 1 == ns) CN_TRY(capture(m, "pred_embed", x, false, CN_DTYPE_F32, {B, U, d}, s));
            }
            if (cap && ns == 0) CN_TRY(capture(m, "pred_embed", x, false, CN_DTYPE_F32, {B, U, d}, s));
            if (ns + nm == 0) CN_TRY(run_ln(m, m->dec_norm, x, m->dec_h, MU, s));
            return stage_decode_tail(m, U, o, hyp, hyp_stride, hyp_len, score, s);
        }
        for (size_t i = 0; i < m->sad.size(); ++i) CN_TRY(run_conformer_self_layer(m, m->sad[i], x, B, U, nullptr, m->ylen, s));
        if (cap) CN_TRY(capture(m, "pred_embed", x, false, CN_DTYPE_F32, {B, U, d}, s));
        for (size_t i = 0; i < m->mad.size(); ++i) {  // MixAttLayer, relative branch (:85-97)
            const Layer& L = m->mad[i];
            CN_TRY(run_ffn_swish(m, L.w1, L.w2, L.n[0], x, MU, 0.5f, s));
            CN_TRY(run_rel_self_attn(m, L, L.n[2], x, B, U, nullptr, m->ylen, s));
            CN_TRY(run_conv_module(m, L, L.n[1], x, B, U, s));
            CN_TRY(run_src_attn(m, L, &L.n[3], x, B, U, Tp, o->src_trigger ? m->intervals : nullptr, s));
            CN_TRY(run_ffn_swish(m, L.ff2_w1, L.ff2_w2, L.n[4], x, MU, 0.5f, s));
        }
        CN_TRY(run_ln(m, m->dec_norm, x, m->dec_h, MU, s));
        return stage_decode_tail(m, U, o, hyp, hyp_stride, hyp_len, score, s);
    }
    static const bool no_chain = cn_exp_env("CASSNAT_NO_CHAIN") != nullptr;
    if (!m->dec_steps.empty() && !no_chain) {
        // bf16 / d_model 256: every sublayer is [attention] + one row-chain launch that also produces the next
        // sublayer's input projection (src Q -> m->qd, self Q|K|V -> m->qkv) or, after the last one, dec_h
        float* xdec = m->xd;
        const size_t n = m->dec_steps.size();
        auto proj_out = [&](const DecStep& st, void*& out, int& ldo) {
            out = st.self ? m->qkv : m->qd;
            ldo = st.self ? 3 * d : d;
        };
        for (size_t k = 0; k < n; ++k) {
            const DecStep& st = m->dec_steps[k];
            const Layer& L = st.stack == 0 ? m->extra[st.layer] : st.stack == 1 ? m->sad[st.layer] : m->mad[st.layer];
            const bool first_mad = st.stack == 2 && st.layer == 0 && st.self;
            if (first_mad) {
                if (cap) CN_TRY(capture(m, "pred_embed", m->xd, false, CN_DTYPE_F32, {B, U, d}, s));
                if (uni) {
                    CN_TRY(launch_shift_right(m->xd, m->xd2, B, U, d, s));
                    xdec = m->xd2;
                }
            }
            static const bool blkd = cn_exp_env("CASSNAT_NO_BLOCKED_QKV") == nullptr;
            if (k == 0 || (first_mad && uni)) {
                void* out;
                int ldo;
                proj_out(st, out, ldo);
                CN_TRY(run_chain(m, st.entry, xdec, MU, out, ldo, true, CHX_NO_STORE, s, nullptr, 0, blkd, "row_chain_dec"));  // reads row-major x, writes none
            }
            if (st.self)
                CN_TRY(run_self_attn_core(m, B, U, nullptr, m->ylen, (st.stack == 2 && uni) ? 1 : 0, s, blkd));
            else
                CN_TRY(run_src_attn_core(m, L, B, U, Tp,
                                         st.stack == 0 ? m->intervals : (o->src_trigger ? m->intervals : nullptr), s, blkd));
            const bool final = k + 1 == n;
            // use_unimask shifts the stream between the last SAD layer and the first MAD layer: nothing carries over
            const bool carry = final || !(uni && m->dec_steps[k + 1].stack == 2 && m->dec_steps[k + 1].layer == 0 &&
                                          m->dec_steps[k + 1].self);
            void* out = m->dec_h;
            int ldo = d;
            if (!final) proj_out(m->dec_steps[k + 1], out, ldo);
            // blocked between chain launches; row-major where something else touches the stream (queries in, the
            // use_unimask shift, captures); the last sublayer's x is consumed by nobody
            const bool in_rm = k == 0 || (first_mad && uni);
            const bool out_rm = !final && !carry;
            const int xm = cap ? 0 : ((in_rm ? 0 : CHX_IN_BLK) | (final ? CHX_NO_STORE : (out_rm ? 0 : CHX_OUT_BLK)));
            CN_TRY(run_chain(m, st.chain, xdec, MU, out, ldo, carry, xm, s, nullptr, 0, blkd && !final, "row_chain_dec"));
            if (cap && st.stack == 0 && (k + 1 == n || m->dec_steps[k + 1].stack != 0))
                CN_TRY(capture(m, "ac_embed", m->xd, false, CN_DTYPE_F32, {B, U, d}, s));
        }
        if (cap && m->mad.empty()) CN_TRY(capture(m, "pred_embed", m->xd, false, CN_DTYPE_F32, {B, U, d}, s));
        return stage_decode_tail(m, U, o, hyp, hyp_stride, hyp_len, score, s);
    }
    bool pending = false;
    for (size_t i = 0; i < m->extra.size(); ++i) {
        const Layer& L = m->extra[i];
        CN_TRY(run_src_attn(m, L, pending ? nullptr : &L.n[0], m->xd, B, U, Tp, m->intervals, s));
        const Norm* nx = first_norm_after(0, i);
        CN_TRY(run_ffn(m, L, L.n[1], m->xd, MU, nx, m->xn, s));
        pending = nx != nullptr;
    }
    if (cap) CN_TRY(capture(m, "ac_embed", m->xd, false, CN_DTYPE_F32, {B, U, d}, s));
    for (size_t i = 0; i < m->sad.size(); ++i) {
        const Layer& L = m->sad[i];
        CN_TRY(run_self_attn(m, L, pending ? nullptr : &L.n[0], m->xd, B, U, nullptr, m->ylen, 0, s));
        const Norm* nx = first_norm_after(1, i);
        CN_TRY(run_ffn(m, L, L.n[1], m->xd, MU, nx, m->xn, s));
        pending = nx != nullptr;
    }
    if (cap) CN_TRY(capture(m, "pred_embed", m->xd, false, CN_DTYPE_F32, {B, U, d}, s));
    float* xdec = m->xd;
    if (uni) {
        CN_TRY(launch_shift_right(m->xd, m->xd2, B, U, d, s));
        xdec = m->xd2;
        pending = false;
    }
    for (size_t i = 0; i < m->mad.size(); ++i) {
        const Layer& L = m->mad[i];
        const bool last = i + 1 == m->mad.size();
        CN_TRY(run_self_attn(m, L, pending ? nullptr : &L.n[0], xdec, B, U, nullptr, m->ylen, uni ? 1 : 0, s));
        CN_TRY(run_src_attn(m, L, &L.n[1], xdec, B, U, Tp, o->src_trigger ? m->intervals : nullptr, s));
        const Norm* nx = last ? &m->dec_norm : first_norm_after(2, i);
        CN_TRY(run_ffn(m, L, L.n[2], xdec, MU, nx, last ? m->dec_h : m->xn, s));
        pending = !last && nx != nullptr;
    }
    if (m->mad.empty()) CN_TRY(run_ln(m, m->dec_norm, xdec, m->dec_h, MU, s));
    return stage_decode_tail(m, U, o, hyp, hyp_stride, hyp_len, score, s);
}

// generator + argmax / top-k + hypothesis packing on m->dec_h
int stage_decode_tail(cn_model* m, int U, const cn_decode_opts* o, int32_t* hyp, int hyp_stride, int32_t* hyp_len,
                      double* score, hipStream_t s) {
    const cn_config& c = m->cfg;
    const int d = c.d_model, B = m->B * m->dec_group, MU = B * U;
    const bool cap = o->capture != 0;
    if (cap) CN_TRY(capture(m, "dec_h", m->dec_h, true, CN_DTYPE_F32, {B, U, d}, s));
    const int k = o->beam_width;
    CN_TRY(run_generator(m, m->att_gen, m->dec_h, MU, m->tok, m->val, cap || k > 1, s));
    if (cap) CN_TRY(capture(m, "att_out", m->logits, false, CN_DTYPE_F32, {B, U, c.vocab_size}, s));
    m->last_k = 0;
    if (k > 1) {
        CN_TRY(launch_topk(m->logits, MU, c.vocab_size, c.vocab_size, k, m->topk_idx, m->topk_val, s));
        m->last_k = k;
    }
    if (hyp)
        CN_TRY(launch_greedy_pack(m->tok, m->val, m->ylen, B, U, o->sos, hyp_stride, hyp, hyp_len, score, s, o->sub_batch,
                                  m->ragged ? m->utt_meta : nullptr, m->u_predicted ? m->ymax : nullptr));
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// exported functions
// ---------------------------------------------------------------------------------------------
extern "C" int cn_model_create(const cn_config* cfg, cn_model** out) {
    if (!cfg || !out) {
        cn_set_error("cn_model_create: null argument");
        return -1;
    }
    const cn_config& c = *cfg;
    if (c.d_model % 64 != 0 || c.n_head < 1 || c.d_model != 64 * c.n_head) {
        cn_set_error("cn_model_create: this build needs d_model == 64 * n_head (d_k = 64)");
        return -1;
    }
    if (c.d_encff % 64 || c.d_decff % 64 || c.d_model > 1024) {
        cn_set_error("cn_model_create: feed-forward widths must be multiples of 64 and d_model <= 1024");
        return -1;
    }
    if (c.precision != CN_PRECISION_F32 && c.precision != CN_PRECISION_BF16 && c.precision != CN_PRECISION_FP8 &&
        c.precision != CN_PRECISION_BF16X3 && c.precision != CN_PRECISION_F16) {
        cn_set_error("cn_model_create: unknown precision");
        return -1;
    }
    const int own_prec = cn_own_precision(c.precision, "cn_model_create");
    if (own_prec < 0) return -1;
    if (c.precision == CN_PRECISION_BF16X3 && (c.d_encff % 32 || c.d_decff % 32 || c.d_model % 32 || c.d_ff % 32)) {
        cn_set_error("cn_model_create: the split-bf16 (bf16x3) engine needs d_model and the feed-forward widths to be multiples of 32");
        return -1;
    }
    if (c.precision == CN_PRECISION_FP8 && (c.ast || c.conf_enc || c.d_model % 128 != 0 || c.d_encff % 128 != 0)) {
        cn_set_error("cn_model_create: the fp8 encoder mode covers the transformer-block NAT model with d_model, d_encff % 128 == 0");
        return -1;
    }
    if (c.fp8_scope < 0 || c.fp8_scope > 7 || ((c.fp8_scope & CN_FP8_LINEAR) && !(c.fp8_scope & CN_FP8_CONV2)) || c.fp8_ffn_first_layer < 0) {
        cn_set_error("cn_model_create: fp8_scope is a mask of CN_FP8_CONV2 | CN_FP8_LINEAR | CN_FP8_FFN (LINEAR needs CONV2), fp8_ffn_first_layer >= 0");
        return -1;
    }
    if (c.input_size < 4 || c.vocab_size < 4 || c.max_batch < 1 || c.max_frames < 4 || c.n_enc < 0 || c.n_extra < 0 ||
        c.n_self_dec < 0 || c.n_mix_dec < 0) {
        cn_set_error("cn_model_create: bad dimensions");
        return -1;
    }
    CN_HIP_CHECK(hipSetDevice(c.device));
    cn_model* m = new cn_model();
    m->cfg = c;
    // fp8: a bf16 engine (storage, decoder side, conv front-end) whose encoder-layer products take e4m3fn operands
    m->fp8_enc = c.precision == CN_PRECISION_FP8;
    m->fp8_scope = c.fp8_scope ? c.fp8_scope : (CN_FP8_CONV2 | CN_FP8_LINEAR | CN_FP8_FFN);
    m->prec = m->fp8_enc ? CN_PREC_BF16 : (c.precision == CN_PRECISION_BF16X3 ? CN_PREC_X3 : own_prec);
    m->es = cn_elem_size(m->prec);
    m->maxB = c.max_batch;
    m->maxT = c.max_frames;
    m->maxT1 = (c.max_frames - 1) / 2 + 1;
    m->maxTp = (m->maxT1 - 1) / 2 + 1;
    m->maxU = 16 * c.max_batch;
    m->F1 = (c.input_size - 1) / 2 + 1;
    m->F2 = (m->F1 - 1) / 2 + 1;
    *out = m;
    return 0;
}

extern "C" int cn_model_create_shared(const cn_config* cfg, cn_model* donor, cn_model** out) {
    if (!donor || !donor->finalized || !donor->blob) {
        cn_set_error("cn_model_create_shared: the donor must be a finalized model");
        return -1;
    }
    if (cfg->device != donor->cfg.device || cfg->precision != donor->cfg.precision) {
        cn_set_error("cn_model_create_shared: device and precision must be the donor's");
        return -1;
    }
    cn_model* m = nullptr;
    CN_TRY(cn_model_create(cfg, &m));
    m->blob_owner = donor->blob_owner;
    m->blob = donor->blob;
    m->blob_bytes = donor->blob_bytes;
    m->blob_borrowed = true;
    m->pe_rows = donor->pe_rows;
    const int rc = cn_model_finalize(m);  // layout pass over the donor's blob + this handle's own workspace
    if (rc != 0) {
        cn_model_destroy(m);
        return rc;
    }
    *out = m;
    return 0;
}

extern "C" void cn_model_destroy(cn_model* m) {
    if (!m) return;
    for (void* p : m->allocs) (void)hipFree(p);
    for (auto& kv : m->captures)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto e : m->ev_pool) (void)hipEventDestroy(e);
    if (m->ymax_pinned) (void)hipHostFree(m->ymax_pinned);
    if (m->ymax_ring) (void)hipHostFree(m->ymax_ring);
    if (m->op16_fault) (void)hipHostFree(m->op16_fault);
    for (void* q : m->ast_allocs) (void)hipFree(q);
    for (auto& kv : m->scratch)
        if (kv.second.first) (void)hipFree(kv.second.first);
    m->blob_owner.reset();  // the last handle that uses the blob frees it
    delete m;
}

extern "C" int cn_model_load_weights(cn_model* m, const char* name, const float* host_data, const int64_t* shape,
                                     int32_t ndim) {
    if (!m || !name || !host_data || !shape || ndim < 1 || ndim > 4) {
        cn_set_error("cn_model_load_weights: bad argument");
        return -1;
    }
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(host_data, host_data + n);
    m->host[name] = std::move(t);
    m->finalized = false;
    return 0;
}

extern "C" int cn_model_load_pe(cn_model* m, const float* host_table, int32_t rows) {
    if (!m || !host_table || rows < 1) {
        cn_set_error("cn_model_load_pe: bad argument");
        return -1;
    }
    m->pe_host.assign(host_table, host_table + (size_t)rows * m->cfg.d_model);
    m->pe_rows = rows;
    m->finalized = false;
    return 0;
}

extern "C" int cn_model_finalize(cn_model* m) {
    if (!m) {
        cn_set_error("cn_model_finalize: null model");
        return -1;
    }
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    m->op16_feat_limit_host = -1.f;  // (the range guard's bound is a word of the weights: read back again after this)
    if (m->pe_rows < m->maxTp + 1) {
        if (m->host.empty() && m->pe_rows == 0) {
            m->pe_rows = 5000;  // layout-only: table arrives with the broadcast blob (create_pe max_len)
        } else {
            cn_set_error("cn_model_finalize: positional table missing or shorter than max_frames/4 + 1");
            return -1;
        }
    }
    CN_TRY(build_weights(m));
    CN_TRY(build_workspace(m));
    m->finalized = true;
    return 0;
}

extern "C" int cn_model_weight_blob(cn_model* m, void** dev_ptr, int64_t* bytes) {
    if (!m || !m->blob) {
        cn_set_error("cn_model_weight_blob: finalize first");
        return -1;
    }
    *dev_ptr = m->blob;
    *bytes = (int64_t)m->blob_bytes;
    return 0;
}

extern "C" int cn_encode_align(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T,
                               int32_t F, const cn_decode_opts* opts, int32_t* ymax_host, void* stream) {
    CN_TRY(check_call(m, B, T, F));
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(stage_encode_align(m, feats_dev, size_ratio_dev, B, T, F, opts, s));
    CN_HIP_CHECK(hipMemcpyAsync(m->ymax_pinned, m->ymax, sizeof(int), hipMemcpyDeviceToHost, s));
    CN_HIP_CHECK(hipStreamSynchronize(s));
    const int ymax = *m->ymax_pinned;
    if (ymax_host) *ymax_host = ymax;
    return 0;
}

namespace {
// greedy NAT decode of one call; subs != null: a merged pass (see cn_decode_nast_merged); u_hint > 0: predicted row count
int decode_nast_impl(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                     const cn_decode_opts* opts, const SubList* subs, int32_t u_hint, int32_t* hyp_out_dev, int32_t hyp_stride,
                     int32_t* hyp_len_dev, double* score_dev, void* stream, int32_t* ticket_out) {
    CN_TRY(check_call(m, B, T, F, /*area=*/true));
    if (!opts || opts->beam_width < 1 || opts->beam_width > 16) {
        cn_set_error("cn_decode_nast: beam_width must be in [1,16]");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));  // callers may decode from several host threads (one handle each)
    if (subs) {
        if (m->cfg.conf_enc || m->cfg.conf_dec) {
            cn_set_error("cn_decode_nast_merged: transformer blocks only (a conformer's GroupNorm sees the padded rows of the merged pass)");
            return -1;
        }
        if (opts->capture || opts->beam_width != 1 || opts->sub_batch) {
            cn_set_error("cn_decode_nast_merged: greedy decode without capture; sub_batch is implied by the batch list");
            return -1;
        }
        CN_TRY(launch_expand_meta(*subs, m->utt_meta, B, s));
        m->ragged = true;
        m->ragged_next = true;
    }
    CN_TRY(stage_encode_align(m, feats_dev, size_ratio_dev, B, T, F, opts, s));
    const int slot = (int)(m->ticket_seq & 3);
    if (u_hint > 0 && opts->beam_width == 1 && !opts->capture) {
        // No mid-pass host sync: the decoder side is launched on the predicted row count.  Results do not depend on U as long
        // as U >= the true maximum (rows past an utterance's own count are masked keys that contribute exactly zero, and the
        // finish is limited by the device-side counts); the true maximum lands in the ticket's page-locked word and the caller
        // checks it once the stream has drained (cn_decode_ticket) - on a miss the pass is decoded again.
        const int U = std::min(std::max(u_hint, 1), m->Tp + 1);
        if (hyp_out_dev && hyp_stride < U + 1) {
            cn_set_error("cn_decode_nast: hyp_stride " + std::to_string(hyp_stride) + " < predicted rows + 1 = " + std::to_string(U + 1));
            return -1;
        }
        m->u_predicted = true;
        CN_HIP_CHECK(hipMemcpyAsync(m->ymax_ring + slot, m->ymax, sizeof(int), hipMemcpyDeviceToHost, s));
        m->ticket_U[slot] = U;
        CN_TRY(stage_decode(m, U, opts, hyp_out_dev, hyp_stride, hyp_len_dev, score_dev, s));
    } else {
        CN_HIP_CHECK(hipMemcpyAsync(m->ymax_ring + slot, m->ymax, sizeof(int), hipMemcpyDeviceToHost, s));
        CN_HIP_CHECK(hipStreamSynchronize(s));  // U is data dependent
        const int ymax = m->ymax_ring[slot];
        *m->ymax_pinned = ymax;
        if (ymax < 1 || ymax > m->Tp + 1) {
            cn_set_error("cn_decode_nast: alignment produced an impossible token count");
            return -3;
        }
        if (hyp_out_dev && hyp_stride < ymax + 1) {
            cn_set_error("cn_decode_nast: hyp_stride " + std::to_string(hyp_stride) + " < ymax+1 = " + std::to_string(ymax + 1));
            return -1;
        }
        m->ticket_U[slot] = ymax;
        CN_TRY(stage_decode(m, ymax, opts, hyp_out_dev, hyp_stride, hyp_len_dev, score_dev, s));
    }
    // a ticket is the call's sequence number (its word is seq & 3): cn_decode_ticket can tell a word that a later call on this
    // handle - cn_decode_nast included - has taken over, instead of handing out another pass's counts
    m->ticket_id[slot] = (int)(m->ticket_seq & 0x3fffffff);
    if (ticket_out) *ticket_out = m->ticket_id[slot];
    ++m->ticket_seq;
    return 0;
}
}  // namespace

extern "C" int cn_decode_nast(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T,
                              int32_t F, const cn_decode_opts* opts, int32_t* hyp_out_dev, int32_t hyp_stride,
                              int32_t* hyp_len_dev, double* score_dev, void* stream) {
    return decode_nast_impl(m, feats_dev, size_ratio_dev, B, T, F, opts, nullptr, 0, hyp_out_dev, hyp_stride, hyp_len_dev, score_dev,
                            stream, nullptr);
}

extern "C" int cn_decode_nast_merged(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                                     const cn_decode_opts* opts, int32_t n_sub, const int32_t* sub_rows_host,
                                     const int32_t* sub_frames_host, int32_t u_hint, int32_t* hyp_out_dev, int32_t hyp_stride,
                                     int32_t* hyp_len_dev, double* score_dev, void* stream, int32_t* ticket_out) {
    if (n_sub < 0 || n_sub > CN_MAX_SUB || (n_sub > 0 && (!sub_rows_host || !sub_frames_host))) {
        cn_set_error("cn_decode_nast_merged: between 0 and " + std::to_string(CN_MAX_SUB) + " batches per pass");
        return -1;
    }
    if (n_sub == 0)
        return decode_nast_impl(m, feats_dev, size_ratio_dev, B, T, F, opts, nullptr, u_hint, hyp_out_dev, hyp_stride, hyp_len_dev,
                                score_dev, stream, ticket_out);
    SubList subs;
    subs.n = n_sub;
    long long rows = 0;
    for (int k = 0; k < n_sub; ++k) {
        if (sub_rows_host[k] < 1 || sub_frames_host[k] < 1 || sub_frames_host[k] > T) {
            cn_set_error("cn_decode_nast_merged: every batch needs >= 1 utterance and 1 <= frames <= T");
            return -1;
        }
        subs.rows[k] = sub_rows_host[k];
        subs.frames[k] = sub_frames_host[k];
        rows += sub_rows_host[k];
    }
    if (rows != B) {
        cn_set_error("cn_decode_nast_merged: the batches' utterance counts must add up to B");
        return -1;
    }
    return decode_nast_impl(m, feats_dev, size_ratio_dev, B, T, F, opts, &subs, u_hint, hyp_out_dev, hyp_stride, hyp_len_dev, score_dev,
                            stream, ticket_out);
}

extern "C" int cn_decode_ticket(cn_model* m, int32_t ticket, int32_t* ymax_host, int32_t* rows_used_host) {
    if (!m || !m->ymax_ring || ticket < 0) {
        cn_set_error("cn_decode_ticket: bad ticket");
        return -1;
    }
    const int slot = ticket & 3;
    if (m->ticket_id[slot] != ticket) {
        cn_set_error("cn_decode_ticket: ticket " + std::to_string(ticket) + " has expired - more than three decode calls were started on "
                     "this handle since it was issued (at most four passes may be outstanding per handle)");
        return -1;
    }
    if (ymax_host) *ymax_host = m->ymax_ring[slot];
    if (rows_used_host) *rows_used_host = m->ticket_U[slot];
    return 0;
}

// fp16 engines: has a pass since the last call seen features beyond the range the engine's half-precision operands hold (then
// its results are not to be used)?  Valid once the passes' stream work is done; clears the flag.  Other engines: always 0.
extern "C" int cn_take_range_fault(cn_model* m, int32_t* fault_host, float* feature_limit_host) {
    if (!m || !fault_host) {
        cn_set_error("cn_take_range_fault: bad arguments");
        return -1;
    }
    *fault_host = 0;
    if (feature_limit_host) *feature_limit_host = 0.f;
    if (!m->op16_fault) return 0;
    *fault_host = (int32_t)__atomic_exchange_n(m->op16_fault, 0u, __ATOMIC_ACQ_REL);
    if (feature_limit_host && m->op16_feat_limit) {
        if (m->op16_feat_limit_host < 0.f) CN_HIP_CHECK(hipMemcpy(&m->op16_feat_limit_host, m->op16_feat_limit, 4, hipMemcpyDeviceToHost));
        *feature_limit_host = m->op16_feat_limit_host;
    }
    return 0;
}


// ---- decode_type ctc_only / ctc_att: CTC prefix beam search, and the NAT decode on the forced alignment of given labels ----
namespace {
int scratch_buf(cn_model* m, const char* name, size_t bytes, void** out) {
    auto& e = m->scratch[name];
    if (e.second < bytes) {
        if (e.first) (void)hipFree(e.first);
        e.first = nullptr;
        e.second = 0;
        CN_HIP_CHECK(hipMalloc(&e.first, bytes));
        e.second = bytes;
    }
    *out = e.first;
    return 0;
}
// encoder + CTC generator with the log-posteriors of ALL rows kept in m->logits (ctc_out of the reference)
int stage_encode_ctc_rows(cn_model* m, const float* feats, int B, int T, int F, const cn_decode_opts* o, hipStream_t s) {
    CN_TRY(stage_encode(m, feats, B, T, F, o, s));
    m->ctc_maxlp_valid = true;
    return run_generator(m, m->ctc_gen, m->enc_h, B * m->Tp, m->best, m->ctc_maxlp, true, s);
}
}  // namespace

// ctc_beam_decode (src/utils/beam_decode.py:8-93) without a language model: hyp_out_dev [B][beam][hyp_cap] labels (no sos),
// hyp_len_dev / score_dev (score_ctc, float64) / p_blk_dev / p_nblk_dev [B][beam], nbeam_dev [B] hypotheses kept (best first).
extern "C" int cn_ctc_beam(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                           const cn_decode_opts* opts, int32_t beam, int32_t pruning, double length_penalty, int32_t* hyp_out_dev,
                           int32_t hyp_cap, int32_t* hyp_len_dev, double* score_dev, double* p_blk_dev, double* p_nblk_dev,
                           int32_t* nbeam_dev, void* stream) {
    CN_TRY(check_call(m, B, T, F));
    // (a NAT model, or the autoregressive model with its CTC head: ArtTask decode_type 'ctc_only', src/tasks/art_task.py:252-253)
    if (!opts || m->cfg.ast == 2 || !m->ctc_gen.W || beam < 1 || beam > 32 || pruning < 0 || pruning > 32 || !hyp_out_dev || !hyp_len_dev ||
        !score_dev || !p_blk_dev || !p_nblk_dev || !nbeam_dev) {
        cn_set_error("cn_ctc_beam: needs a model with a CTC head, 1 <= ctc_beam <= 32, 0 <= ctc_pruning <= 32 and all output buffers");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(stage_encode_ctc_rows(m, feats_dev, B, T, F, opts, s));
    const int Tp = m->Tp, M = B * Tp, V = m->cfg.vocab_size;
    if (hyp_cap < Tp + 1) {
        cn_set_error("cn_ctc_beam: hyp_cap must be at least T' + 1");
        return -1;
    }
    void *top_idx = nullptr, *top_val = nullptr, *hpar = nullptr, *htok = nullptr;
    const int P = pruning > 0 ? pruning : 1;  // (a zero-width pruning still needs a buffer; the kernel then sees P = 0)
    CN_TRY(scratch_buf(m, "cb_top_idx", (size_t)M * P * 4, &top_idx));
    CN_TRY(scratch_buf(m, "cb_top_val", (size_t)M * P * 4, &top_val));
    CN_TRY(scratch_buf(m, "cb_hist_par", (size_t)M * beam, &hpar));
    CN_TRY(scratch_buf(m, "cb_hist_tok", (size_t)M * beam * 4, &htok));
    if (pruning > 0) {
        ProfScope ps(m, "ctc_topk", 0, (double)M * V * 4, s);
        CN_TRY(launch_topk(m->logits, M, V, V, pruning, (int*)top_idx, (float*)top_val, s));
    }
    CtcBeamArgs a;
    a.logp = m->logits;
    a.top_idx = (const int*)top_idx;
    a.size_ratio = size_ratio_dev;
    a.B = B;
    a.Tp = Tp;
    a.V = V;
    a.P = pruning;
    a.W = beam;
    a.blank = opts->padding_idx;
    a.Lmax = hyp_cap;
    a.lp = length_penalty;
    a.hist_parent = (unsigned char*)hpar;
    a.hist_tok = (int*)htok;
    a.hyp = hyp_out_dev;
    a.hyp_len = hyp_len_dev;
    a.score = score_dev;
    a.p_blk = p_blk_dev;
    a.p_nblk = p_nblk_dev;
    a.n_out = nbeam_dev;
    ProfScope ps(m, "ctc_prefix_beam", 0, (double)M * (pruning + 1) * 4, s);
    return launch_ctc_prefix_beam(a, s);
}

// CassNAT.beam_decode with decode_type 'ctc_att' and sample_num 1 (src/models/cassnat.py:446-448, 391-414): the trigger mask
// comes from the forced (Viterbi) alignment of the given label sequences - the best CTC beam hypotheses - instead of the greedy
// path; everything after align_to_mask is cn_decode_nast's.  labels_dev [B][ld] (no sos), label_len_dev [B].
extern "C" int cn_decode_nast_forced(cn_model* m, const float* feats_dev, const float* size_ratio_dev, int32_t B, int32_t T, int32_t F,
                                     const cn_decode_opts* opts, const int32_t* labels_dev, const int32_t* label_len_dev, int32_t ld,
                                     int32_t max_label_len, int32_t* hyp_out_dev, int32_t hyp_stride, int32_t* hyp_len_dev,
                                     double* score_dev, void* stream) {
    CN_TRY(check_call(m, B, T, F));
    if (!opts || m->cfg.ast || opts->beam_width < 1 || opts->beam_width > 16 || !labels_dev || !label_len_dev || max_label_len < 0 ||
        max_label_len > ld) {
        cn_set_error("cn_decode_nast_forced: bad argument");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(stage_encode_ctc_rows(m, feats_dev, B, T, F, opts, s));
    const int Tp = m->Tp, V = m->cfg.vocab_size;
    if (max_label_len > Tp) {
        cn_set_error("cn_decode_nast_forced: more labels than frames");
        return -1;
    }
    const bool cap = opts->capture != 0;
    if (cap) CN_TRY(capture(m, "ctc_out", m->logits, false, CN_DTYPE_F32, {B, Tp, V}, s));
    if (max_label_len == 0) {  // every hypothesis empty: aligned_seq_shift is all blank (cassnat.py:410-411)
        CN_TRY(launch_fill_int(m->best, (size_t)B * Tp, opts->padding_idx, s));
    } else {
        void* bp = nullptr;
        CN_TRY(scratch_buf(m, "vit_bp", (size_t)B * Tp * (2 * (size_t)max_label_len + 1), &bp));
        ViterbiArgs v;
        v.logp = m->logits;
        v.keymask = m->keymask;
        v.size_ratio = size_ratio_dev;
        v.labels = labels_dev;
        v.label_len = label_len_dev;
        v.B = B;
        v.Tp = Tp;
        v.V = V;
        v.ld = ld;
        v.ymax = max_label_len;
        v.blank = opts->padding_idx;
        v.bp = (unsigned char*)bp;
        v.out_path = m->best;
        ProfScope ps(m, "ctc_viterbi", 0, (double)B * Tp * (2 * max_label_len + 1) * 5, s);
        CN_TRY(launch_ctc_viterbi(v, s));
    }
    AlignArgs al;
    al.best = m->best;
    al.keymask = m->keymask;
    al.size_ratio = size_ratio_dev;
    al.B = B;
    al.Tp = Tp;
    al.blank = opts->padding_idx;
    al.left = opts->left_trigger;
    al.right = opts->right_trigger;
    al.shift = m->shift;
    al.src_size = m->src_size;
    al.ylen = m->ylen;
    al.ymax = m->ymax;
    al.intervals = m->intervals;
    al.raw_path = 1;
    al.ylen_in = label_len_dev;
    CN_TRY(launch_ctc_align(al, s));
    CN_HIP_CHECK(hipMemcpyAsync(m->ymax_pinned, m->ymax, sizeof(int), hipMemcpyDeviceToHost, s));
    CN_HIP_CHECK(hipStreamSynchronize(s));
    const int ymax = *m->ymax_pinned;
    if (ymax < 1 || ymax > Tp + 1 || (hyp_out_dev && hyp_stride < ymax + 1)) {
        cn_set_error("cn_decode_nast_forced: token count outside the output buffers");
        return -1;
    }
    return stage_decode(m, ymax, opts, hyp_out_dev, hyp_stride, hyp_len_dev, score_dev, s);
}

// ---- ESA: error-based sampling of alignments (src/models/cassnat.py:370-376, 441-445) ------------------------------
// cn_esa_begin runs the encoder and the CTC generator once and keeps the two best labels of every frame.  Every
// cn_esa_sample call then builds n_samples sampled alignments per utterance - in alignment g frame t of utterance b takes
// the second-best label iff its draw select[g][b][t] is 1 and the best label's probability is below `threshold` (all-zero
// draws, or select == NULL with n_samples == 1: the best path itself) - and runs the alignment + decoder side on all of
// them in ONE pass of B * n_samples query sets over the B utterances' encoder outputs (n_samples <= cfg.esa_group, which
// sizes the decoder-side workspace): tok_out / val_out [n_samples][B][out_stride] = argmax token and its log-probability per
// decoder row, ylen_out [n_samples][B] (EOS row included), *ymax_host = rows of this pass.  force_U: 0 = decode on this pass's
// own row count, > 0 = on that many rows, -1 = only count (no decode).  The random draws are the caller's (the reference takes
// them from torch.randint).
extern "C" int cn_esa_begin(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                            void* stream) {
    CN_TRY(check_call(m, B, T, F));
    if (!opts || m->cfg.ast) {
        cn_set_error("cn_esa_begin: needs a NAT model and decode options");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(stage_encode(m, feats_dev, B, T, F, opts, s));
    const int d = m->cfg.d_model, V = m->cfg.vocab_size, M = B * m->Tp;
    CN_TRY(run_linear(m, "generator_proj", m->ctc_gen, m->enc_h, d, m->logits, V, 1, M, 0, nullptr, 0, s));
    CN_TRY(launch_logsoftmax_argmax(m->logits, M, V, V, m->best, m->ctc_maxlp, 1, s));
    m->ctc_maxlp_valid = true;
    CN_TRY(launch_topk(m->logits, M, V, V, 2, m->topk_idx, m->topk_val, s));
    return 0;
}

extern "C" int cn_esa_sample(cn_model* m, const uint8_t* select_dev, int32_t n_samples, float threshold,
                             const float* size_ratio_dev, const cn_decode_opts* opts, int32_t* tok_out_dev, float* val_out_dev,
                             int32_t out_stride, int32_t* ylen_out_dev, int32_t* ymax_host, int32_t force_U, void* stream) {
    // beam_width > 1: the pass of the SELECTED alignments (src/models/cassnat.py:556-561 gathers them, :574-637 finishes with
    // top-k per row): the per-row top-k stays in the engine (cn_fetch "topk_idx" / "topk_val" / "ylen"), no token rows are copied
    const bool topk_pass = opts && opts->beam_width > 1;
    if (!m || !opts || !ymax_host || m->B < 1 || opts->beam_width < 1 || opts->beam_width > 16 ||
        (force_U >= 0 && !topk_pass && (!tok_out_dev || !val_out_dev || !ylen_out_dev))) {
        cn_set_error("cn_esa_sample: call cn_esa_begin first; 1 <= beam_width <= 16; beam_width 1 needs the output buffers");
        return -1;
    }
    if (n_samples < 1 || n_samples > std::max(1, m->cfg.esa_group) || (n_samples > 1 && !select_dev)) {
        cn_set_error("cn_esa_sample: n_samples must be in [1, cfg.esa_group] (and needs draws when > 1)");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    const int B = m->B, Tp = m->Tp, G = n_samples, BG = B * G;
    CN_TRY(launch_esa_paths(m->topk_idx, m->topk_val, select_dev, threshold, m->best, B * Tp, G, s));
    AlignArgs al;
    al.best = m->best;
    al.keymask = m->keymask;
    al.size_ratio = size_ratio_dev;
    al.B = BG;
    al.src_mod = B;  // entry g * B + b reads utterance b's mask and length
    al.Tp = Tp;
    al.blank = opts->padding_idx;
    al.left = opts->left_trigger;
    al.right = opts->right_trigger;
    al.shift = m->shift;
    al.src_size = m->src_size;
    al.ylen = m->ylen;
    al.ymax = m->ymax;
    al.intervals = m->intervals;
    CN_TRY(launch_ctc_align(al, s));
    CN_HIP_CHECK(hipMemcpyAsync(m->ymax_pinned, m->ymax, sizeof(int), hipMemcpyDeviceToHost, s));
    CN_HIP_CHECK(hipStreamSynchronize(s));
    int U = *m->ymax_pinned;
    if (force_U < 0) {  // count only: the caller wants the row count of these alignments (to take the maximum over all groups)
        *ymax_host = U;
        return 0;
    }
    if (force_U > 0) {
        // decode on force_U rows per alignment (>= this pass's own count): the conformer's GroupNorm runs over an utterance's
        // whole (channels x rows) image, padded rows included, so every group has to use the row count of ALL samples to give
        // what the reference's one big batch gives
        if (force_U < U) {
            cn_set_error("cn_esa_sample: force_U is smaller than the row count of this pass");
            return -3;
        }
        U = force_U;
    }
    if (U < 1 || U > Tp + 1 || (!topk_pass && U > out_stride)) {
        cn_set_error("cn_esa_sample: token count outside the output stride");
        return -3;
    }
    m->dec_group = G;
    const int rc = stage_decode(m, U, opts, nullptr, 0, nullptr, nullptr, s);
    m->dec_group = 1;
    if (rc) return rc;
    *ymax_host = U;
    if (topk_pass) return 0;
    CN_TRY(launch_copy_rows(tok_out_dev, out_stride, m->tok, U, U, BG, s));
    CN_TRY(launch_copy_rows(val_out_dev, out_stride, m->val, U, U, BG, s));
    CN_TRY(launch_copy_rows(ylen_out_dev, BG, m->ylen, BG, BG, 1, s));
    *ymax_host = U;
    return 0;
}

// ---- TransformerLM scoring (src/models/lm.py:52-56 as used at cassnat.py:507-523): score[b][u] = log p(tgt[b][u] | tok[b][..u])
// under the causal + length mask (key j allowed iff j <= u and j < len[b]).  Model created with cfg.ast == 2:
// n_enc encoder layers of width d_encff, parameters text_embed.0.lut / encoder.* / out_generator.proj.
extern "C" int cn_lm_score(cn_model* m, const int32_t* tok_dev, const int32_t* tgt_dev, const int32_t* len_dev, int32_t B,
                           int32_t U, int32_t ld, float* score_dev, void* stream) {
    if (!m || m->cfg.ast != 2 || !m->finalized || !m->tgt_lut) {
        cn_set_error("cn_lm_score: the model was not created with cfg.ast = 2 / not finalized");
        return -1;
    }
    if (B < 1 || U < 1 || ld < U || (size_t)B * U > (size_t)m->maxB * (m->maxTp + 1) || U > m->pe_rows) {
        cn_set_error("cn_lm_score: batch x length exceeds the workspace");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(build_workspace(m));
    const int d = m->cfg.d_model, V = m->cfg.vocab_size, M = B * U;
    float* x = m->x;
    CN_TRY(launch_lm_embed(tok_dev, ld, m->tgt_lut, m->pe, x, B, U, d, sqrtf((float)d), s));
    static const bool no_chain = cn_exp_env("CASSNAT_NO_CHAIN") != nullptr;
    if (!m->enc.empty() && m->enc_chain.size() == m->enc.size() && !no_chain) {
        // bf16 / d_model 256: per layer [causal + length-masked attention] + one row-chain launch, as in stage_encode
        if (m->enc_entry.w) {
            CN_TRY(run_chain(m, m->enc_entry, x, M, m->qkv, 3 * d, true, CHX_NO_STORE, s));
        } else {
            CN_TRY(run_ln(m, m->enc[0].n[0], x, m->xn, M, s));
            CN_TRY(run_linear(m, "qkv_proj", m->enc[0].qkv, m->xn, d, m->qkv, 3 * d, 0, M, 0, nullptr, 0, s));
        }
        for (size_t n = 0; n < m->enc.size(); ++n) {
            const bool last = n + 1 == m->enc.size();
            CN_TRY(run_self_attn_core(m, B, U, nullptr, len_dev, 1, s));
            const int xm = (n > 0 ? CHX_IN_BLK : 0) | (last ? CHX_NO_STORE : CHX_OUT_BLK);
            CN_TRY(run_chain(m, m->enc_chain[n], x, M, last ? m->enc_h : m->qkv, last ? d : 3 * d, true, xm, s));
        }
    } else {
        for (size_t n = 0; n < m->enc.size(); ++n) {
            const Layer& L = m->enc[n];
            CN_TRY(run_self_attn(m, L, &L.n[0], x, B, U, nullptr, len_dev, 1, s));
            CN_TRY(run_ffn(m, L, L.n[1], x, M, nullptr, nullptr, s));
        }
        CN_TRY(run_ln(m, m->enc_norm, x, m->enc_h, M, s));
    }
    static const bool no_fused = cn_exp_env("CASSNAT_LM_NO_FUSED_TAIL") != nullptr;
    if (m->att_gen.gm_w && !no_fused) {  // bf16 / d_model 256: generator + log-softmax + gather in one kernel, no (M, V) tensor
        const bool x3 = m->prec == CN_PREC_X3;
        ProfScope ps(m, "generator_gather_fused", (x3 ? 6.0 : 2.0) * M * V * d, ((double)M * d + (double)V * d) * (x3 ? 4 : 2), s);
        GenmaxArgs a;
        a.x3 = x3;
        a.h = m->enc_h;
        a.wp = m->att_gen.gm_w;
        a.bp = m->att_gen.gm_b;
        a.M = M;
        a.V = V;
        a.d = d;
        a.tgt = tgt_dev;
        a.tgt_lp = score_dev;
        a.tgt_U = U;
        a.tgt_ld = ld;
        return launch_genmax(a, s);
    }
    CN_TRY(run_linear(m, "generator_proj", m->att_gen, m->enc_h, d, m->logits, V, 1, M, 0, nullptr, 0, s));
    CN_TRY(launch_logsoftmax_argmax(m->logits, M, V, V, m->best, m->ctc_maxlp, 1, s));
    m->ctc_maxlp_valid = true;
    return launch_gather_logp(m->logits, V, tgt_dev, ld, score_dev, B, U, s);
}

// ESA ranking with rank_model == 'at_baseline' (src/models/cassnat.py:514-520; Transformer.forward_decoder,
// src/models/transformer.py:113-116): the autoregressive model (cfg.ast = 1) scores the NAT predictions teacher-forced.
// Its own encoder runs on the B utterances; the decoder then takes N = B * n_per_utt token rows (row e belongs to utterance
// e % B: the sample-major order of cn_esa_sample) under the causal + length mask, with cross attention to its utterance's
// encoder output under the padding mask: score[e][u] = log softmax(att_generator(dec_h))[e][u][tgt[e][u]] (the reference
// takes the probability itself: exp on the caller's side).  tok / tgt / score are [N][ld], ld >= U; cfg.esa_group >= n_per_utt.
extern "C" int cn_ast_teacher_score(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                                    const int32_t* tok_dev, const int32_t* tgt_dev, const int32_t* len_dev, int32_t n_per_utt,
                                    int32_t U, int32_t ld, float* score_dev, void* stream) {
    CN_TRY(check_call(m, B, T, F));
    if (!opts || m->cfg.ast != 1 || !m->tgt_lut || n_per_utt < 1 || n_per_utt > std::max(1, m->cfg.esa_group) || U < 1 || ld < U ||
        U > m->pe_rows) {
        cn_set_error("cn_ast_teacher_score: needs an autoregressive model (cfg.ast = 1) whose cfg.esa_group covers the samples per utterance");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(stage_encode(m, feats_dev, B, T, F, opts, s));
    const int d = m->cfg.d_model, V = m->cfg.vocab_size, Tp = m->Tp, N = B * n_per_utt, M = N * U;
    if ((size_t)M > (size_t)m->maxB * (m->maxTp + 1) * std::max(1, m->cfg.esa_group)) {
        cn_set_error("cn_ast_teacher_score: token rows exceed the workspace");
        return -1;
    }
    m->dec_group = n_per_utt;  // N query sets over the B utterances' encoder outputs
    float* x = m->xd;
    CN_TRY(launch_lm_embed(tok_dev, ld, m->tgt_lut, m->pe, x, N, U, d, sqrtf((float)d), s));
    for (size_t i = 0; i < m->mad.size(); ++i) {  // DecoderLayer: self attention, source attention, feed-forward
        const Layer& L = m->mad[i];
        CN_TRY(run_self_attn(m, L, &L.n[0], x, N, U, nullptr, len_dev, 1, s));
        CN_TRY(run_src_attn(m, L, &L.n[1], x, N, U, Tp, nullptr, s));
        CN_TRY(run_ffn(m, L, L.n[2], x, M, nullptr, nullptr, s));
    }
    CN_TRY(run_ln(m, m->dec_norm, x, m->dec_h, M, s));
    // generator in chunks of whole token rows that fit the logits buffer (B * (T' + 1) rows)
    const int cap_rows = m->maxB * (m->maxTp + 1);
    const int seq_per = std::max(1, cap_rows / U);
    for (int e0 = 0; e0 < N; e0 += seq_per) {
        const int ne = std::min(seq_per, N - e0), rows = ne * U;
        const void* h = (const unsigned char*)m->dec_h + (size_t)e0 * U * d * m->es;
        CN_TRY(run_linear(m, "generator_proj", m->att_gen, h, d, m->logits, V, 1, rows, 0, nullptr, 0, s));
        CN_TRY(launch_logsoftmax_argmax(m->logits, rows, V, V, m->tok, m->val, 1, s));
        CN_TRY(launch_gather_logp(m->logits, V, tgt_dev + (size_t)e0 * ld, ld, score_dev + (size_t)e0 * ld, ne, U, s));
    }
    m->dec_group = 1;
    return 0;
}

// ArtTask decode_type 'ctc_correct' (Transformer.fast_decode_with_ctc, src/models/transformer.py:243-342): the CTC greedy hypothesis
// of every utterance, behind sos, is the decoder's teacher-forced input under the causal + padding mask; the finish loop then reads
// the k best labels of row i while i <= length[b].  Device part: encoder, CTC arg-max + collapse, decoder, generator + log-softmax +
// top-k.  Outputs: len_out_dev [B] (CTC labels per utterance), tok_out_dev / val_out_dev [B][U][k] (U = longest + 1 rows, returned
// in *rows_host; the buffers hold B * (T' + 1) * k entries).  One host sync on the longest hypothesis (the reference's max(length)).
extern "C" int cn_ast_ctc_correct(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                                  int32_t k, int32_t* tok_out_dev, float* val_out_dev, int32_t* len_out_dev, int32_t* rows_host,
                                  void* stream) {
    CN_TRY(check_call(m, B, T, F));
    if (!opts || m->cfg.ast != 1 || !m->tgt_lut || !m->ctc_gen.W || k < 1 || k > 16 || !tok_out_dev || !val_out_dev || !len_out_dev ||
        !rows_host) {
        cn_set_error("cn_ast_ctc_correct: needs an autoregressive model (cfg.ast = 1) with a CTC head, 1 <= k <= 16 and all output buffers");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(stage_encode_ctc_rows(m, feats_dev, B, T, F, opts, s));
    const int d = m->cfg.d_model, V = m->cfg.vocab_size, Tp = m->Tp, ld = Tp + 1;
    if (ld > m->pe_rows) {
        cn_set_error("cn_ast_ctc_correct: more subsampled frames than rows of the positional table");
        return -1;
    }
    void *tgt = nullptr, *keylen = nullptr;
    CN_TRY(scratch_buf(m, "cc_tgt", (size_t)B * ld * 4, &tgt));
    CN_TRY(scratch_buf(m, "cc_keylen", (size_t)B * 4, &keylen));
    CN_TRY(launch_ctc_collapse(m->best, m->keymask, B, Tp, opts->sos, opts->padding_idx, ld, (int*)tgt, len_out_dev, (int*)keylen, m->ymax, s));
    CN_HIP_CHECK(hipMemcpyAsync(m->ymax_pinned, m->ymax, sizeof(int), hipMemcpyDeviceToHost, s));
    CN_HIP_CHECK(hipStreamSynchronize(s));  // the decoder's row count is data dependent (max_length + 1, transformer.py:266)
    const int U = *m->ymax_pinned + 1, M = B * U;
    if (U < 1 || U > ld) {
        cn_set_error("cn_ast_ctc_correct: impossible CTC hypothesis length");
        return -3;
    }
    m->dec_group = 1;
    float* x = m->xd;
    CN_TRY(launch_lm_embed((const int*)tgt, ld, m->tgt_lut, m->pe, x, B, U, d, sqrtf((float)d), s));
    for (size_t i = 0; i < m->mad.size(); ++i) {  // DecoderLayer: self attention (causal + padding mask), source attention, feed-forward
        const Layer& L = m->mad[i];
        CN_TRY(run_self_attn(m, L, &L.n[0], x, B, U, nullptr, (const int*)keylen, 1, s));
        CN_TRY(run_src_attn(m, L, &L.n[1], x, B, U, Tp, nullptr, s));
        CN_TRY(run_ffn(m, L, L.n[2], x, M, nullptr, nullptr, s));
    }
    CN_TRY(run_ln(m, m->dec_norm, x, m->dec_h, M, s));
    CN_TRY(run_linear(m, "generator_proj", m->att_gen, m->dec_h, d, m->logits, V, 1, M, 0, nullptr, 0, s));
    CN_TRY(launch_logsoftmax_argmax(m->logits, M, V, V, m->tok, m->val, 1, s));
    CN_TRY(launch_topk(m->logits, M, V, V, k, tok_out_dev, val_out_dev, s));
    *rows_host = U;
    return 0;
}

extern "C" int cn_profile_begin(cn_model* m, const char* tags) {
    if (!m) {
        cn_set_error("cn_profile_begin: null model");
        return -1;
    }
    m->prof_on = true;
    m->prof_filter = (tags && *tags) ? std::string("|") + tags + "|" : std::string();
    m->prof_pending.clear();
    m->prof_stats.clear();
    m->ev_used = 0;
    // events for a few thousand tagged launches exist before the caller's timed region starts (the pool still grows on
    // demand past that)
    while (m->ev_pool.size() < 4096) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        m->ev_pool.push_back(e);
    }
    return 0;
}

extern "C" int cn_profile_end(cn_model* m, char* json_out, int64_t cap) {
    if (!m) {
        cn_set_error("cn_profile_end: null model");
        return -1;
    }
    CN_HIP_CHECK(hipDeviceSynchronize());
    for (auto& pp : m->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pp.a, pp.b) != hipSuccess) continue;
        ProfStat& st = m->prof_stats[pp.tag];
        st.count += 1;
        st.ms += ms;
        st.flops += pp.flops;
        st.bytes += pp.bytes;
    }
    m->prof_pending.clear();
    m->prof_on = false;
    m->ev_used = 0;
    std::string js = "{";
    bool first = true;
    for (auto& kv : m->prof_stats) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s\"%s\": {\"count\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.count, kv.second.ms, kv.second.flops, kv.second.bytes);
        js += buf;
        first = false;
    }
    js += "}";
    if (json_out && cap > 0) {
        if ((int64_t)js.size() + 1 > cap) {
            cn_set_error("cn_profile_end: buffer too small");
            return -1;
        }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
    }
    return 0;
}

extern "C" int cn_fetch(cn_model* m, const char* name, void* host_dst, int64_t max_bytes, int64_t* shape_out,
                        int32_t* ndim_out, int32_t* dtype_out) {
    if (!m || !name) {
        cn_set_error("cn_fetch: bad argument");
        return -1;
    }
    const std::string n(name);
    const void* src = nullptr;
    int dtype = CN_DTYPE_F32;
    std::vector<int64_t> shape;
    bool model_prec = false;
    const int64_t B = m->B, Tp = m->Tp, U = m->U, d = m->cfg.d_model;
    auto it = m->captures.find(n);
    const bool stale = it != m->captures.end() && it->second.call != m->call_id;  // captured by an earlier call: not served
    if (it != m->captures.end() && !stale) {
        src = it->second.p;
        dtype = it->second.dtype;
        shape = it->second.shape;
    } else if (n == "best_paths") { src = m->best; dtype = CN_DTYPE_I32; shape = {B, Tp};
    } else if (n == "ctc_maxlp") {
        if (!m->ctc_maxlp_valid) {
            cn_set_error("cn_fetch: ctc_maxlp is only produced by capture runs (the fused CTC generator computes the arg-max alone)");
            return -1;
        }
        src = m->ctc_maxlp; shape = {B, Tp};
    } else if (n == "aligned_seq_shift") { src = m->shift; dtype = CN_DTYPE_I32; shape = {B, Tp};
    } else if (n == "keymask") { src = m->keymask; dtype = CN_DTYPE_U8; shape = {B, Tp};
    } else if (n == "src_size") { src = m->src_size; dtype = CN_DTYPE_I32; shape = {B};
    } else if (n == "ylen") { src = m->ylen; dtype = CN_DTYPE_I32; shape = {B};
    } else if (n == "ymax") { src = m->ymax; dtype = CN_DTYPE_I32; shape = {1};
    } else if (n == "intervals") { src = m->intervals; dtype = CN_DTYPE_I32; shape = {B, Tp + 1, 4};
    } else if (n == "tok") { src = m->tok; dtype = CN_DTYPE_I32; shape = {B, U};
    } else if (n == "val") { src = m->val; shape = {B, U};
    } else if (n == "topk_idx") { src = m->topk_idx; dtype = CN_DTYPE_I32; shape = {B, U, m->last_k};
    } else if (n == "topk_val") { src = m->topk_val; shape = {B, U, m->last_k};
    } else if (n == "enc_h_live") { src = m->enc_h; model_prec = true; shape = {B, Tp, d};
    } else if (stale) {
        cn_set_error("cn_fetch: '" + n + "' was captured by an earlier call, not by the last one (capture is a per-call option)");
        return -1;
    } else {
        cn_set_error("cn_fetch: unknown tensor '" + n + "' (captures need opts.capture=1)");
        return -1;
    }
    size_t cnt = 1;
    for (auto v : shape) cnt *= (size_t)v;
    const size_t esz = dtype == CN_DTYPE_U8 ? 1 : (dtype == CN_DTYPE_F64 ? 8 : 4);
    if (shape_out) {
        for (size_t i = 0; i < shape.size() && i < 4; ++i) shape_out[i] = shape[i];
    }
    if (ndim_out) *ndim_out = (int)shape.size();
    if (dtype_out) *dtype_out = dtype;
    if (!host_dst) return 0;  // shape query
    if ((int64_t)(cnt * esz) > max_bytes) {
        cn_set_error("cn_fetch: destination too small");
        return -1;
    }
    CN_HIP_CHECK(hipDeviceSynchronize());
    if (model_prec && m->prec != CN_PREC_F32) {
        float* tmp = nullptr;
        CN_HIP_CHECK(hipMalloc((void**)&tmp, cnt * 4));
        int rc = launch_convert_back(m->prec, src, tmp, cnt, 0);
        if (rc == 0 && hipMemcpy(host_dst, tmp, cnt * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = -2;
        (void)hipFree(tmp);
        if (rc) {
            cn_set_error("cn_fetch: convert/copy failed");
            return rc;
        }
        return 0;
    }
    CN_HIP_CHECK(hipMemcpy(host_dst, src, cnt * esz, hipMemcpyDeviceToHost));
    return 0;
}

// ---- single-kernel entry points ------------------------------------------------------------------
static FbankOpts fbank_opts_from(const cn_fbank_opts* o) {
    FbankOpts f;
    f.sample_rate = o->sample_rate;
    f.frame_length_ms = o->frame_length_ms;
    f.frame_shift_ms = o->frame_shift_ms;
    f.preemph = o->preemph;
    f.low_freq = o->low_freq;
    f.high_freq = o->high_freq;
    f.num_mel = o->num_mel;
    f.window_type = o->window_type;
    f.remove_dc = o->remove_dc;
    f.use_power = o->use_power;
    f.use_log = o->use_log;
    return f;
}

extern "C" void cn_fbank_default_opts(cn_fbank_opts* o) {
    if (!o) return;
    const FbankOpts f;
    std::memset(o, 0, sizeof(*o));
    o->sample_rate = f.sample_rate;
    o->frame_length_ms = f.frame_length_ms;
    o->frame_shift_ms = f.frame_shift_ms;
    o->preemph = f.preemph;
    o->low_freq = f.low_freq;
    o->high_freq = f.high_freq;
    o->num_mel = f.num_mel;
    o->window_type = f.window_type;
    o->remove_dc = f.remove_dc;
    o->use_power = f.use_power;
    o->use_log = f.use_log;
}

extern "C" int32_t cn_fbank_num_frames(const cn_fbank_opts* o, int32_t num_samples) {
    return o ? fbank_num_frames(fbank_opts_from(o), num_samples) : 0;
}

extern "C" int cn_fbank(const cn_fbank_opts* o, const float* wave_dev, const int32_t* num_samples_dev, int32_t B,
                        int32_t max_samples, const float* cmvn_mean_dev, const float* cmvn_istd_dev, float* feats_dev,
                        int32_t Tmax, float pad_value, void* stream) {
    if (!o || !wave_dev || !num_samples_dev || !feats_dev || B < 0 || max_samples < 0 || Tmax < 0) {
        cn_set_error("cn_fbank: bad argument");
        return -1;
    }
    return launch_fbank(fbank_opts_from(o), wave_dev, num_samples_dev, B, max_samples, cmvn_mean_dev, cmvn_istd_dev, feats_dev,
                        Tmax, pad_value, (hipStream_t)stream);
}

extern "C" int cn_op_gemm(int32_t precision, const void* A, int32_t lda, const void* W, const float* bias, void* C,
                          int32_t ldc, int32_t c_is_f32, int32_t M, int32_t N, int32_t K, int32_t relu,
                          const float* resid, int32_t ldr, const float* pe, int32_t pe_period, float scale,
                          void* stream) {
    if ((precision = cn_own_precision(precision, "cn_op_gemm")) < 0) return -1;
    GemmArgs g;
    g.A = A;
    g.lda = lda;
    g.W = W;
    g.bias = bias;
    g.C = C;
    g.ldc = ldc;
    g.c_f32 = c_is_f32;
    g.M = M;
    g.N = N;
    g.K = K;
    g.epi = (relu ? CN_EPI_RELU : 0) | (resid ? CN_EPI_RESID : 0) | (pe ? CN_EPI_EMBED : 0);
    g.resid = resid;
    g.ldr = ldr;
    g.pe = pe;
    g.pe_period = pe_period;
    g.scale = scale;
    return launch_gemm(precision, g, (hipStream_t)stream);
}

extern "C" int cn_op_convert(int32_t precision, const void* src, void* dst, int64_t n, int32_t to_f32, void* stream) {
    if ((precision = cn_own_precision(precision, "cn_op_convert")) < 0) return -1;
    if (n < 0 || !src || !dst) {
        cn_set_error("cn_op_convert: bad argument");
        return -1;
    }
    return to_f32 ? launch_convert_back(precision, src, (float*)dst, (size_t)n, (hipStream_t)stream)
                  : launch_convert(precision, (const float*)src, dst, (size_t)n, (hipStream_t)stream);
}

extern "C" int cn_op_conv1(int32_t precision, const float* x, const float* w9c, const float* bias, void* out, int32_t B,
                           int32_t T, int32_t F, int32_t C, void* stream) {
    if ((precision = cn_own_precision(precision, "cn_op_conv1")) < 0) return -1;
    return launch_conv1(precision, x, w9c, bias, out, B, T, F, (T - 1) / 2 + 1, (F - 1) / 2 + 1, C, 0, (hipStream_t)stream);
}

// the bf16 engine's bordered image ([B][T1 + 2][F1 + 2][C] bf16, zero border) as conv2's LDS-DMA kernel reads it, from the
// matrix-core kernel (C == 256, (F - 1) / 2 + 3 >= 32)
extern "C" int cn_op_conv1_bordered(const float* x, const float* w9c, const float* bias, void* out, int32_t B, int32_t T, int32_t F,
                                    int32_t C, void* stream) {
    return launch_conv1_bordered_bf16(x, w9c, bias, out, B, T, F, (T - 1) / 2 + 1, (F - 1) / 2 + 1, C, (hipStream_t)stream);
}

// conv front-end of the fp8 engine through the ABI (config 5): conv1 -> e4m3fn image at `img_scale` (bordered) -> conv2 on e4m3
// operands; w2_host fp32 [C][3][3][C] (k = (kh * 3 + kw) * C + ci) is quantised here at the largest power-of-two scale that keeps it
// in range.  img8_out_dev (optional): the bordered image [B][T1 + 2][F1 + 2][C] bytes.  out: bf16 [B * T2 * F2][C], or with
// out8_scale > 0 e4m3fn bytes at that scale (what linear_out's e4m3 form reads)
extern "C" int cn_op_conv_frontend_fp8(const float* x_dev, const float* w1_9c_dev, const float* b1_dev, const float* w2_host,
                                       const float* b2_dev, void* out_dev, void* img8_out_dev, int32_t B, int32_t T, int32_t F,
                                       int32_t C, float img_scale, float out8_scale, float* w_scale_out, void* stream) {
    if (!conv2_f8_applies(C, C)) {
        cn_set_error("cn_op_conv_frontend_fp8: 256 channels only");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int T1 = (T - 1) / 2 + 1, F1 = (F - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1, F2 = (F1 - 1) / 2 + 1;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)C * 9 * C; ++i) mx = std::max(mx, std::fabs(w2_host[i]));
    const int lg = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
    const float ws = std::ldexp(1.f, lg);
    if (w_scale_out) *w_scale_out = ws;
    std::vector<unsigned char> w8((size_t)C * 9 * C);
    for (size_t i = 0; i < w8.size(); ++i) w8[i] = cn_f32_to_e4m3_host(w2_host[i] * ws);
    const int q[4] = {127 - lg, 127 - (int)std::lround(std::log2(img_scale)), 0, 0};
    const size_t img_bytes = (size_t)B * (T1 + 2) * (F1 + 2) * C;
    void *dw = nullptr, *dq = nullptr, *img = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, w8.size()));
    CN_HIP_CHECK(hipMalloc(&dq, 16));
    CN_HIP_CHECK(hipMalloc(&img, img_bytes));
    CN_HIP_CHECK(hipMemcpy(dw, w8.data(), w8.size(), hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(dq, q, 16, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemsetAsync(img, 0xff, img_bytes, s));  // (NaN bytes: the kernel must write every cell, border included)
    int rc = launch_conv1_f8(x_dev, w1_9c_dev, b1_dev, img, B, T, F, T1, F1, C, 1, img_scale, s);
    if (rc == 0) rc = launch_conv2_f8(img, dw, (const int*)dq, b2_dev, out_dev, B, T1, F1, T2, F2, s, out8_scale);
    if (rc == 0 && img8_out_dev) CN_HIP_CHECK(hipMemcpyAsync(img8_out_dev, img, img_bytes, hipMemcpyDeviceToDevice, s));
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(dq);
    (void)hipFree(img);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_conv_frontend_fp8: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

// The split-bf16 engine's conv front-end in the MIX arithmetic through the ABI (conv1.hip MIXP planes + conv2.hip MIX): x fp32 [B][T][F],
// w2_host fp32 [C][3][3][C] (k = (kh * 3 + kw) * C + ci); out_dev: split-bf16 rows [B * T2 * F2][C] (cn_op_convert turns them into fp32);
// img_out_dev (optional): conv1's three bordered planes, 4 bytes per cell of [B][T1 + 2][F1 + 2][C] (half values, l bytes, q bytes)
extern "C" int cn_op_conv_frontend_mix(const float* x_dev, const float* w1_9c_dev, const float* b1_dev, const float* w2_host,
                                       const float* b2_dev, void* out_dev, void* img_out_dev, int32_t B, int32_t T, int32_t F, int32_t C,
                                       void* stream) {
    if (!conv2_mix_applies(CN_PREC_X3, C, C)) {
        cn_set_error("cn_op_conv_frontend_mix: 256 channels only (and not in the half-precision build of the library)");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int T1 = (T - 1) / 2 + 1, F1 = (F - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1, F2 = (F1 - 1) / 2 + 1;
    const size_t n = (size_t)C * 9 * C;
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w2_host[i]));
    const int lg = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
    const float sq = std::ldexp(1.f, lg), sl = std::ldexp(1.f, lg + 11);
    std::vector<unsigned char> w(4 * n);
    for (size_t i = 0; i < n; ++i) {
        const _Float16 h = (_Float16)w2_host[i];
        std::memcpy(&w[2 * i], &h, 2);
        w[2 * n + i] = cn_f32_to_e4m3_host(w2_host[i] * sq);
        w[3 * n + i] = cn_f32_to_e4m3_host((w2_host[i] - (float)h) * sl);
    }
    const int q[4] = {127 - lg, 127 - MIX_LG_AL, 127 - (lg + 11), 127 - MIX_LG_AQ};
    const size_t img_bytes = (size_t)B * (T1 + 2) * (F1 + 2) * C * 4;
    void *dw = nullptr, *dq = nullptr, *img = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, w.size()));
    CN_HIP_CHECK(hipMalloc(&dq, 16));
    CN_HIP_CHECK(hipMalloc(&img, img_bytes));
    CN_HIP_CHECK(hipMemcpy(dw, w.data(), w.size(), hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(dq, q, 16, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemsetAsync(img, 0xff, img_bytes, s));  // (NaN bytes: conv1 must write every cell of every plane, border included)
    int rc = launch_conv1_mixplanes(x_dev, w1_9c_dev, b1_dev, img, B, T, F, T1, F1, C, 1, std::ldexp(1.f, MIX_LG_AL), std::ldexp(1.f, MIX_LG_AQ), s);
    if (rc == 0)
        rc = launch_conv2_mix(img, dw, (const unsigned char*)dw + 2 * n, (const unsigned char*)dw + 3 * n, (const int*)dq, b2_dev, out_dev, B, T1, F1,
                              T2, F2, s);
    if (rc == 0 && img_out_dev) CN_HIP_CHECK(hipMemcpyAsync(img_out_dev, img, img_bytes, hipMemcpyDeviceToDevice, s));
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(dq);
    (void)hipFree(img);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_conv_frontend_mix: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

// linear_out of the fp8 engine through the ABI: a8_dev [M][K] e4m3fn at a_scale (K = 5120), w_host fp32 [256][K] quantised here at
// the largest power-of-two scale in range; out fp32 [M][256] = (a . w^T / (a_scale w_scale) + bias) * out_scale + pe[m % pe_period]
extern "C" int cn_op_linear256_fp8(const void* a8_dev, const float* w_host, const float* bias_dev, float* out_dev, int32_t M, int32_t K,
                                   float a_scale, float out_scale, const float* pe_dev, int32_t pe_period, float* w_scale_out,
                                   void* stream) {
    if (!linear256_f8_applies(256, K)) {
        cn_set_error("cn_op_linear256_fp8: K must be 5120");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)256 * K; ++i) mx = std::max(mx, std::fabs(w_host[i]));
    const int lg = mx > 0.f ? (int)std::floor(std::log2(448.f / mx)) : 0;
    const float ws = std::ldexp(1.f, lg);
    if (w_scale_out) *w_scale_out = ws;
    std::vector<unsigned char> w8((size_t)256 * K);
    for (size_t i = 0; i < w8.size(); ++i) w8[i] = cn_f32_to_e4m3_host(w_host[i] * ws);
    const int q[4] = {127 - lg, 127 - (int)std::lround(std::log2(a_scale)), 0, 0};
    void *dw = nullptr, *dq = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, w8.size()));
    CN_HIP_CHECK(hipMalloc(&dq, 16));
    CN_HIP_CHECK(hipMemcpy(dw, w8.data(), w8.size(), hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(dq, q, 16, hipMemcpyHostToDevice));
    int rc = launch_linear256_f8(a8_dev, dw, (const int*)dq, bias_dev, out_dev, M, K, out_scale, pe_dev, pe_period, s);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(dq);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_linear256_fp8: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

extern "C" int cn_op_conv2(int32_t precision, const void* conv1_out, const void* w_khwc, const float* bias, void* out,
                           int32_t B, int32_t T1, int32_t F1, int32_t C, void* stream) {
    if ((precision = cn_own_precision(precision, "cn_op_conv2")) < 0) return -1;
    GemmArgs g;
    const int T2 = (T1 - 1) / 2 + 1, F2 = (F1 - 1) / 2 + 1;
    g.A = conv1_out;
    g.W = w_khwc;
    g.bias = bias;
    g.C = out;
    g.ldc = C;
    g.M = B * T2 * F2;
    g.N = C;
    g.K = 9 * C;
    g.epi = CN_EPI_RELU;
    g.conv = 1;
    g.cB = B;
    g.cT1 = T1;
    g.cF1 = F1;
    g.cC = C;
    g.cT2 = T2;
    g.cF2 = F2;
    if (!conv2_dma_applies(precision, C, C)) return launch_gemm(precision, g, (hipStream_t)stream);
    // bf16, 256 channels: the LDS-DMA kernel reads an image with a one-cell zero halo (the model has conv1 write it that
    // way); this test entry pads a copy of the plain image
    const size_t cell = (size_t)C * 2, prow = (size_t)(F1 + 2) * cell;
    void* padded = nullptr;
    CN_HIP_CHECK(hipMalloc(&padded, (size_t)B * (T1 + 2) * prow));
    CN_HIP_CHECK(hipMemsetAsync(padded, 0, (size_t)B * (T1 + 2) * prow, (hipStream_t)stream));
    for (int b = 0; b < B; ++b)
        CN_HIP_CHECK(hipMemcpy2DAsync((unsigned char*)padded + ((size_t)b * (T1 + 2) + 1) * prow + cell, prow,
                                      (const unsigned char*)conv1_out + (size_t)b * T1 * F1 * cell, (size_t)F1 * cell,
                                      (size_t)F1 * cell, T1, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    g.A = padded;
    g.conv_halo = 1;
    int rc = launch_gemm(precision, g, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(padded);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_conv2: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

extern "C" int cn_op_layernorm(int32_t precision, const float* x, const float* a2, const float* b2, void* y, int32_t M,
                               int32_t d, float eps, void* stream) {
    if ((precision = cn_own_precision(precision, "cn_op_layernorm")) < 0) return -1;
    return launch_layernorm(precision, x, a2, b2, y, 0, M, d, eps, (hipStream_t)stream);
}

extern "C" int cn_op_attention(int32_t precision, const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V,
                               int32_t ldv, void* O, int32_t ldo, int32_t B, int32_t H, int32_t Lq, int32_t Lk,
                               const uint8_t* keymask, const int32_t* klen, const int32_t* intervals, int32_t iv_stride,
                               int32_t causal, float scale, void* stream) {
    if ((precision = cn_own_precision(precision, "cn_op_attention")) < 0) return -1;
    AttnArgs a;
    a.Q = Q;
    a.K = K;
    a.V = V;
    a.O = O;
    a.ldq = ldq;
    a.ldk = ldk;
    a.ldv = ldv;
    a.ldo = ldo;
    a.B = B;
    a.H = H;
    a.Lq = Lq;
    a.Lk = Lk;
    a.keymask = keymask;
    a.klen = klen;
    a.intervals = intervals;
    a.iv_stride = iv_stride;
    a.causal = causal;
    a.scale = scale;
    int rc = launch_attention(precision, a, (hipStream_t)stream);
    if (rc == 0 && cn_exp_env("CASSNAT_ATTN_STAMPS")) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)attention_print_stamps();
    }
    return rc;
}

extern "C" int cn_op_logsoftmax_argmax(float* logits, int32_t M, int32_t V, int32_t* arg, float* maxlp,
                                       int32_t write_logp, void* stream) {
    return launch_logsoftmax_argmax(logits, M, V, V, arg, maxlp, write_logp, (hipStream_t)stream);
}

extern "C" int cn_op_ctc_align(const int32_t* best, const uint8_t* keymask, const float* size_ratio, int32_t B,
                               int32_t Tp, int32_t blank, int32_t left, int32_t right, int32_t* shift,
                               int32_t* src_size, int32_t* ylen, int32_t* ymax, int32_t* intervals, void* stream) {
    AlignArgs a;
    a.best = best;
    a.keymask = keymask;
    a.size_ratio = size_ratio;
    a.B = B;
    a.Tp = Tp;
    a.blank = blank;
    a.left = left;
    a.right = right;
    a.shift = shift;
    a.src_size = src_size;
    a.ylen = ylen;
    a.ymax = ymax;
    a.intervals = intervals;
    return launch_ctc_align(a, (hipStream_t)stream);
}

// the two kernels of decode_type ctc_only / ctc_att on given log-posteriors (test entries; all pointers device)
extern "C" int cn_op_ctc_prefix_beam(const float* logp, const float* size_ratio, int32_t B, int32_t Tp, int32_t V, int32_t beam,
                                     int32_t pruning, double length_penalty, int32_t blank, int32_t* hyp, int32_t hyp_cap,
                                     int32_t* hyp_len, double* score, double* p_blk, double* p_nblk, int32_t* nbeam, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t M = (size_t)B * Tp;
    const int P = pruning > 0 ? pruning : 1;
    int* top_idx = nullptr;
    float* top_val = nullptr;
    unsigned char* hpar = nullptr;
    int* htok = nullptr;
    CN_HIP_CHECK(hipMalloc((void**)&top_idx, M * P * 4));
    CN_HIP_CHECK(hipMalloc((void**)&top_val, M * P * 4));
    CN_HIP_CHECK(hipMalloc((void**)&hpar, M * (size_t)std::max(beam, 1)));
    CN_HIP_CHECK(hipMalloc((void**)&htok, M * (size_t)std::max(beam, 1) * 4));
    int rc = pruning > 0 ? launch_topk(logp, (int)M, V, V, pruning, top_idx, top_val, s) : 0;
    if (rc == 0) {
        CtcBeamArgs a;
        a.logp = logp;
        a.top_idx = top_idx;
        a.size_ratio = size_ratio;
        a.B = B;
        a.Tp = Tp;
        a.V = V;
        a.P = pruning;
        a.W = beam;
        a.blank = blank;
        a.Lmax = hyp_cap;
        a.lp = length_penalty;
        a.hist_parent = hpar;
        a.hist_tok = htok;
        a.hyp = hyp;
        a.hyp_len = hyp_len;
        a.score = score;
        a.p_blk = p_blk;
        a.p_nblk = p_nblk;
        a.n_out = nbeam;
        rc = launch_ctc_prefix_beam(a, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(top_idx);
    (void)hipFree(top_val);
    (void)hipFree(hpar);
    (void)hipFree(htok);
    return rc;
}

extern "C" int cn_op_ctc_viterbi(const float* logp, const uint8_t* keymask, const float* size_ratio, const int32_t* labels,
                                 const int32_t* label_len, int32_t B, int32_t Tp, int32_t V, int32_t ld, int32_t ymax, int32_t blank,
                                 int32_t* out_path, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    unsigned char* bp = nullptr;
    CN_HIP_CHECK(hipMalloc((void**)&bp, (size_t)B * Tp * (2 * (size_t)ymax + 1)));
    ViterbiArgs v;
    v.logp = logp;
    v.keymask = keymask;
    v.size_ratio = size_ratio;
    v.labels = labels;
    v.label_len = label_len;
    v.B = B;
    v.Tp = Tp;
    v.V = V;
    v.ld = ld;
    v.ymax = ymax;
    v.blank = blank;
    v.bp = bp;
    v.out_path = out_path;
    const int rc = launch_ctc_viterbi(v, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(bp);
    return rc;
}

extern "C" int cn_op_greedy_pack(const int32_t* tok, const float* val, const int32_t* ylen, int32_t B, int32_t U,
                                 int32_t sos, int32_t hyp_stride, int32_t* hyp, int32_t* hyp_len, double* score,
                                 void* stream) {
    return launch_greedy_pack(tok, val, ylen, B, U, sos, hyp_stride, hyp, hyp_len, score, (hipStream_t)stream);
}

extern "C" int cn_op_topk(const float* logp, int32_t M, int32_t V, int32_t k, int32_t* idx, float* val, void* stream) {
    return launch_topk(logp, M, V, V, k, idx, val, (hipStream_t)stream);
}

// fp8 product through the ABI (config 5): A bf16 on the device is quantised at a_scale, W (HOST fp32 [N][K]) at the largest
// power-of-two scale that fits e4m3fn (as cn_model_finalize does); C = relu?(A_q . W_q^T / scales + bias), fp32 [M][N]
extern "C" int cn_op_gemm_fp8(const void* a_bf16_dev, int32_t lda, const float* w_host, const float* bias_dev, float* c_dev,
                              int32_t M, int32_t N, int32_t K, float a_scale, int32_t relu, float* w_scale_out, void* stream) {
    if (M < 1 || N < 1 || K < 128 || K % 128 != 0) {
        cn_set_error("cn_op_gemm_fp8: K must be a positive multiple of 128");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)N * K; ++i) mx = std::max(mx, std::fabs(w_host[i]));
    const float ws = mx > 0.f ? std::ldexp(1.f, (int)std::floor(std::log2(448.f / mx))) : 1.f;
    if (w_scale_out) *w_scale_out = ws;
    std::vector<unsigned char> w8((size_t)N * K);
    for (size_t i = 0; i < w8.size(); ++i) w8[i] = Packer::f32_to_e4m3(w_host[i] * ws);
    void *dw = nullptr, *da = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, w8.size()));
    CN_HIP_CHECK(hipMalloc(&da, (size_t)M * K));
    CN_HIP_CHECK(hipMemcpy(dw, w8.data(), w8.size(), hipMemcpyHostToDevice));
    int rc = launch_quantize_fp8(a_bf16_dev, lda, da, M, K, a_scale, s);
    if (rc == 0) {
        GemmArgs g;
        g.A = da;
        g.lda = K;
        g.W = dw;
        g.bias = bias_dev;
        g.C = c_dev;
        g.ldc = N;
        g.c_f32 = 1;
        g.M = M;
        g.N = N;
        g.K = K;
        g.epi = relu ? CN_EPI_RELU : 0;
        g.ab_fp8 = 1;
        g.acc_scale = 1.f / (a_scale * ws);
        rc = launch_gemm(CN_PREC_BF16, g, s);
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(da);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_gemm_fp8: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

extern "C" int cn_op_cmvn(float* feats_dev, const int32_t* len_dev, const double* mean_dev, const double* std_dev, int32_t B, int32_t T,
                          int32_t F, void* stream) {
    if (!feats_dev || !len_dev || !mean_dev || !std_dev) {
        cn_set_error("cn_op_cmvn: null argument");
        return -1;
    }
    return launch_cmvn(feats_dev, len_dev, mean_dev, std_dev, B, T, F, (hipStream_t)stream);
}

// Host side of the packed reader: n byte ranges (an utterance's rows inside the memory map of an archive) copied back to back into a
// page-locked staging buffer in ONE call - ctypes releases the GIL for its duration, so the decode pipelines' host threads copy side
// by side (numpy's slice assignment holds it: two threads took turns, 5 ms a turn).  threads > 1: the ranges are dealt over that
// many std::threads in equal byte shares (--load_data_workers on this path).
extern "C" int cn_host_gather(void* dst, const uint64_t* src_ptrs, const uint64_t* dst_offsets, const uint64_t* nbytes, int32_t n,
                              int32_t threads) {
    if (!dst || !src_ptrs || !dst_offsets || !nbytes || n < 0) {
        cn_set_error("cn_host_gather: null argument");
        return -1;
    }
    auto run = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i)
            memcpy(static_cast<unsigned char*>(dst) + dst_offsets[i], reinterpret_cast<const void*>(src_ptrs[i]), nbytes[i]);
    };
    const int nt = std::max(1, std::min<int>(threads, n));
    if (nt == 1) {
        run(0, n);
        return 0;
    }
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) total += nbytes[i];
    std::vector<std::thread> pool;
    int lo = 0;
    uint64_t acc = 0;
    for (int t = 0; t < nt; ++t) {
        int hi = lo;
        const uint64_t want = total * (uint64_t)(t + 1) / (uint64_t)nt;
        while (hi < n && (t + 1 == nt || acc + nbytes[hi] <= want || hi == lo)) acc += nbytes[hi++];
        if (t + 1 < nt) pool.emplace_back(run, lo, hi);
        else run(lo, hi);
        lo = hi;
    }
    for (auto& th : pool) th.join();
    return 0;
}

extern "C" int cn_op_unpack_rows(const float* packed_dev, const int32_t* off_dev, const int32_t* len_dev, float* out_dev, int32_t rows,
                                 int32_t T, int32_t F, float pad, const double* mean_dev, const double* std_dev, void* stream) {
    if (!packed_dev || !off_dev || !len_dev || !out_dev || (!mean_dev) != (!std_dev)) {
        cn_set_error("cn_op_unpack_rows: null argument (mean and std come together)");
        return -1;
    }
    return launch_unpack_rows(packed_dev, off_dev, len_dev, out_dev, rows, T, F, pad, mean_dev, std_dev, (hipStream_t)stream);
}

extern "C" int cn_op_quantize_fp8(const void* src_bf16_dev, int32_t ld, void* dst_dev, int32_t M, int32_t K, float scale,
                                  void* stream) {
    return launch_quantize_fp8(src_bf16_dev, ld, dst_dev, M, K, scale, (hipStream_t)stream);
}

extern "C" int cn_op_logsoftmax_topk(const float* logits, int32_t M, int32_t V, float temperature, int32_t k, int32_t* idx,
                                     float* val, void* stream) {
    return launch_logsoftmax_topk(logits, M, V, V, temperature, k, idx, val, (hipStream_t)stream);
}

extern "C" int cn_op_ffn_fused(float* x_dev, const float* ln_a_dev, const float* ln_b_dev, const float* w1_host,
                               const float* b1_dev, const float* w2_host, const float* b2_dev, const float* nln_a_dev,
                               const float* nln_b_dev, void* xn_out_dev, int32_t M, int32_t dff, float eps,
                               int32_t nslice, void* stream) {
    if (dff <= 0 || dff % 128 != 0 || dff > 2048) {
        cn_set_error("cn_op_ffn_fused: d_ff must be a positive multiple of 128, at most 2048");
        return -1;
    }
    const size_t n = (size_t)dff * 256;
    std::vector<uint16_t> h1(n), h2(n);
    pack_ffn_w1(w1_host, dff, h1.data());
    pack_ffn_w2(w2_host, dff, h2.data());
    void *d1 = nullptr, *d2 = nullptr;
    CN_HIP_CHECK(hipMalloc(&d1, n * 2));
    CN_HIP_CHECK(hipMalloc(&d2, n * 2));
    CN_HIP_CHECK(hipMemcpy(d1, h1.data(), n * 2, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(d2, h2.data(), n * 2, hipMemcpyHostToDevice));
    FfnFusedArgs a;
    a.x = x_dev;
    a.ln_a = ln_a_dev;
    a.ln_b = ln_b_dev;
    a.w1p = d1;
    a.b1 = b1_dev;
    a.w2p = d2;
    a.b2 = b2_dev;
    a.nln_a = nln_a_dev;
    a.nln_b = nln_b_dev;
    a.xn_out = xn_out_dev;
    a.M = M;
    a.d = 256;
    a.dff = dff;
    a.eps = eps;
    void* part = nullptr;
    if (nslice > 1) {  // d_ff split + reduce (the decode-step form)
        CN_HIP_CHECK(hipMalloc(&part, (size_t)nslice * M * 256 * 4));
        a.nslice = nslice;
        a.partial = (float*)part;
    }
    int rc = launch_ffn_fused(a, (hipStream_t)stream);
    if (rc == 0 && nslice > 1)
        rc = launch_ffn_reduce(x_dev, a.partial, nslice, b2_dev, nln_a_dev, nln_b_dev, xn_out_dev, M, eps, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d1);
    (void)hipFree(d2);
    if (part) (void)hipFree(part);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_ffn_fused: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

// the same sublayer in the split-bf16 precision (fused_x3.hip); xn_out_dev: split-bf16 [M][256] or NULL
extern "C" int cn_op_ffn_x3(float* x_dev, const float* ln_a_dev, const float* ln_b_dev, const float* w1_host, const float* b1_dev,
                            const float* w2_host, const float* b2_dev, const float* nln_a_dev, const float* nln_b_dev,
                            void* xn_out_dev, int32_t M, int32_t dff, float eps, int32_t mix, void* stream) {
    if (!ffn_x3_applies(256, dff)) {
        cn_set_error("cn_op_ffn_x3: d_ff must be a positive multiple of 128, at most 2048");
        return -1;
    }
    const size_t bytes = ffn_x3_stream_bytes(dff);
    std::vector<uint16_t> h(bytes / 2);
    pack_ffn_x3(w1_host, w2_host, dff, h.data(), mix != 0);
    void* dw = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, bytes));
    CN_HIP_CHECK(hipMemcpy(dw, h.data(), bytes, hipMemcpyHostToDevice));
    FfnX3Args a;
    a.x = x_dev;
    a.ln_a = ln_a_dev;
    a.ln_b = ln_b_dev;
    a.wst = dw;
    a.mix = mix != 0;
    a.b1 = b1_dev;
    a.b2 = b2_dev;
    a.nln_a = nln_a_dev;
    a.nln_b = nln_b_dev;
    a.xn_out = xn_out_dev;
    a.M = M;
    a.d = 256;
    a.dff = dff;
    a.eps = eps;
    int rc = launch_ffn_x3(a, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(dw);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_ffn_x3: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

// The row-chain form of the split-bf16 engine as one op (fused_x3.hip PRO / TAIL): x += Wo . ctx + bo; x += FFN(LN1 x); then
// LN_next(x) -> xn_out_dev (wt_host null) or its projection Wt . LN_next(x) + bt -> tail_out_dev (split-bf16 rows of tail_n
// elements).  ctx_dev: split-bf16 [M][256] or null (no output projection); weight matrices on the host (fp32, nn.Linear layout),
// vectors on the device.
extern "C" int cn_op_x3_chain(float* x_dev, const void* ctx_dev, const float* wo_host, const float* bo_dev, const float* ln_a_dev,
                              const float* ln_b_dev, const float* w1_host, const float* b1_dev, const float* w2_host,
                              const float* b2_dev, const float* nln_a_dev, const float* nln_b_dev, void* xn_out_dev,
                              const float* wt_host, const float* bt_dev, void* tail_out_dev, int32_t tail_n, int32_t M, int32_t dff,
                              float eps, int32_t mix, void* stream) {
    if (!ffn_x3_applies(256, dff) || (wt_host && !proj_x3_applies(tail_n, 256))) {
        cn_set_error("cn_op_x3_chain: d_ff a multiple of 128 (<= 2048), tail_n a multiple of 32 (<= 1024)");
        return -1;
    }
    std::vector<void*> dev;
    auto upload = [&](const void* h, size_t bytes, void** out) -> int {
        CN_HIP_CHECK(hipMalloc(out, bytes));
        dev.push_back(*out);
        CN_HIP_CHECK(hipMemcpy(*out, h, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    FfnX3Args a;
    int rc = 0;
    {
        std::vector<uint16_t> h(ffn_x3_stream_bytes(dff) / 2);
        pack_ffn_x3(w1_host, w2_host, dff, h.data(), mix != 0);
        void* dw = nullptr;
        rc = upload(h.data(), h.size() * 2, &dw);
        a.wst = dw;
        a.mix = mix != 0;
    }
    if (rc == 0 && ctx_dev) {
        std::vector<unsigned char> h((size_t)256 * 1024);
        pack_proj_x3(wo_host, 256, h.data());
        void* dw = nullptr;
        rc = upload(h.data(), h.size(), &dw);
        a.ctx = ctx_dev;
        a.wo_p = dw;
        a.bo = bo_dev;
    }
    if (rc == 0 && wt_host) {
        std::vector<unsigned char> h((size_t)tail_n * 1024);
        pack_proj_x3(wt_host, tail_n, h.data());
        void* dw = nullptr;
        rc = upload(h.data(), h.size(), &dw);
        a.tail_p = dw;
        a.tail_b = bt_dev;
        a.tail_out = tail_out_dev;
        a.tail_n = tail_n;
        a.ld_tail = tail_n;
    }
    a.x = x_dev;
    a.ln_a = ln_a_dev;
    a.ln_b = ln_b_dev;
    a.b1 = b1_dev;
    a.b2 = b2_dev;
    a.nln_a = nln_a_dev;
    a.nln_b = nln_b_dev;
    a.xn_out = xn_out_dev;
    a.M = M;
    a.d = 256;
    a.dff = dff;
    a.eps = eps;
    if (rc == 0) rc = launch_ffn_x3(a, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    for (void* q : dev) (void)hipFree(q);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_x3_chain: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

extern "C" int cn_op_chain(float* x_dev, const void* ctx_dev, int32_t ldctx, const float* wo_host, const float* bo_host,
                           const float* ln1_a_host, const float* ln1_b_host, const float* w1_host, const float* b1_host,
                           const float* w2_host, const float* b2_host, const float* nln_a_host, const float* nln_b_host,
                           const float* wt_host, const float* bt_host, void* out_dev, int32_t ldo, int32_t M, int32_t dff,
                           int32_t tail_n, float eps, int32_t x_mode, void* stream) {
    if (dff < 0 || dff % 32 != 0 || dff > 2048 || tail_n < 0 || tail_n % 32 != 0 || tail_n > 1536) {
        cn_set_error("cn_op_chain: d_ff and the tail width must be multiples of 32 (<= 2048 / <= 1536)");
        return -1;
    }
    ChainWeights w;
    w.wo = ctx_dev ? wo_host : nullptr;
    w.bo = bo_host;
    w.ln1_a = ln1_a_host;
    w.ln1_b = ln1_b_host;
    w.w1 = w1_host;
    w.b1 = b1_host;
    w.w2 = w2_host;
    w.b2 = b2_host;
    w.nln_a = nln_a_host;
    w.nln_b = nln_b_host;
    w.wt = wt_host;
    w.bt = bt_host;
    w.dff = dff;
    w.tail_n = tail_n;
    const int f8 = (x_mode & 32) != 0;
    if (f8 && (dff <= 0 || dff % 256 != 0 || (x_mode & 8))) {
        cn_set_error("cn_op_chain: the e4m3 feed-forward form (x_mode bit 32) needs d_ff % 256 == 0 and the ReLU activation");
        return -1;
    }
    const size_t units = chain_stream_units(ctx_dev != nullptr, dff, tail_n, f8);
    std::vector<uint16_t> hs((units + 7) * (CHAIN_UNIT_BYTES / 2));  // (seven spare units: the kernel's dummy refills read them)
    std::vector<float> ht(CHAIN_TAB_FLOATS);
    int hq[4] = {127, 127, 127, 127};
    pack_chain(w, hs.data(), ht.data(), f8 ? hq : nullptr);
    void *ds = nullptr, *dt = nullptr;
    CN_HIP_CHECK(hipMalloc(&ds, hs.size() * 2));
    CN_HIP_CHECK(hipMalloc(&dt, ht.size() * 4 + 16));
    CN_HIP_CHECK(hipMemcpy(ds, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy((char*)dt + ht.size() * 4, hq, 16, hipMemcpyHostToDevice));
    ChainArgs a;
    a.x = x_dev;
    a.ctx = ctx_dev;
    a.ldctx = ldctx;
    a.wstream = ds;
    a.tab = (const float*)dt;
    a.out = out_dev;
    a.ldo = ldo;
    a.M = M;
    a.d = 256;
    a.dff = dff;
    a.tail_n = tail_n;
    a.has_next = nln_a_host != nullptr;
    a.eps = eps;
    a.x_in_blocked = (x_mode & 1) != 0;
    a.x_out_blocked = (x_mode & 2) != 0;
    a.store_x = (x_mode & 4) == 0;
    a.swish = (x_mode & 8) != 0;
    a.out_blocked = (x_mode & 16) != 0;
    a.f8 = f8;
    a.f8_q = f8 ? reinterpret_cast<const int*>((char*)dt + ht.size() * 4) : nullptr;
    int rc = launch_chain(a, (hipStream_t)stream);
    if (const char* rep = cn_exp_env("CASSNAT_CHAIN_REPEAT")) {  // timing runs only: x keeps being updated
        // CASSNAT_CHAIN_STREAMS = n: the repeats go round-robin onto n private streams (how do concurrent launches share
        // the chip?); they race on x, which a timing run does not look at
        const int ns = cn_exp_env("CASSNAT_CHAIN_STREAMS") ? atoi(cn_exp_env("CASSNAT_CHAIN_STREAMS")) : 0;
        std::vector<hipStream_t> ss(ns > 0 ? ns : 0);
        for (auto& q : ss) (void)hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
        (void)hipStreamSynchronize((hipStream_t)stream);
        for (int i = 1; i < atoi(rep) && rc == 0; ++i) rc = launch_chain(a, ns > 0 ? ss[i % ns] : (hipStream_t)stream);
        for (auto& q : ss) {
            (void)hipStreamSynchronize(q);
            (void)hipStreamDestroy(q);
        }
    }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (cn_exp_env("CASSNAT_CHAIN_STAMPS")) (void)chain_print_stamps();
    (void)hipFree(ds);
    (void)hipFree(dt);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_chain: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

static int op_genmax_impl(const void* h_dev, const float* w_host, const float* b_host, int32_t M, int32_t V,
                          int32_t* arg_dev, float* maxlp_dev, const int32_t* tgt_dev, float* tgt_lp_dev, int32_t U, int32_t ld,
                          void* stream) {
    const int vtw = genmax_vtw(V);
    if (vtw > 48) {
        cn_set_error("cn_op_genmax: V too large");
        return -1;
    }
    std::vector<uint16_t> hw((size_t)4 * vtw * 16 * 512);
    std::vector<float> hb((size_t)4 * vtw * 32);
    pack_genmax(w_host, b_host, V, hw.data(), hb.data());
    void *dw = nullptr, *db = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, hw.size() * 2));
    CN_HIP_CHECK(hipMalloc(&db, hb.size() * 4));
    CN_HIP_CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    GenmaxArgs a;
    a.h = h_dev;
    a.wp = dw;
    a.bp = (const float*)db;
    a.arg = arg_dev;
    a.maxlp = maxlp_dev;
    a.M = M;
    a.V = V;
    a.d = 256;
    a.tgt = tgt_dev;
    a.tgt_lp = tgt_lp_dev;
    a.tgt_U = U;
    a.tgt_ld = ld;
    int rc = launch_genmax(a, (hipStream_t)stream);
    if (const char* rep = cn_exp_env("CASSNAT_GENMAX_REPEAT"))  // timing runs only
        for (int i = 1, n = atoi(rep); rc == 0 && i < n; ++i) rc = launch_genmax(a, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(dw);
    (void)hipFree(db);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_genmax: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

extern "C" int cn_op_genmax(const void* h_dev, const float* w_host, const float* b_host, int32_t M, int32_t V,
                            int32_t* arg_dev, float* maxlp_dev, void* stream) {
    return op_genmax_impl(h_dev, w_host, b_host, M, V, arg_dev, maxlp_dev, nullptr, nullptr, 0, 0, stream);
}

extern "C" int cn_op_genmax_gather(const void* h_dev, const float* w_host, const float* b_host, int32_t B, int32_t U, int32_t V,
                                   const int32_t* tgt_dev, int32_t ld, float* tgt_lp_dev, void* stream) {
    return op_genmax_impl(h_dev, w_host, b_host, B * U, V, nullptr, nullptr, tgt_dev, tgt_lp_dev, U, ld, stream);
}

// the split-bf16 form (CN_PRECISION_BF16X3): h_host fp32 [M][256] is split into hi + lo halves and uploaded by the call;
// tgt_dev == NULL: arg-max (+ maxlp when maxlp_dev), else the target gather with M = rows, U per sequence, ld the target stride
extern "C" int cn_op_genmax_x3(const float* h_host, const float* w_host, const float* b_host, int32_t M, int32_t V, int32_t* arg_dev,
                               float* maxlp_dev, const int32_t* tgt_dev, int32_t U, int32_t ld, float* tgt_lp_dev, void* stream) {
    if (!genmax_applies(CN_PREC_X3, 256, V) || M < 1) {
        cn_set_error("cn_op_genmax_x3: V too large");
        return -1;
    }
    const int vtw = genmax_x3_vtw(V);
    std::vector<uint16_t> hw((size_t)8 * vtw * 16 * 1024);
    std::vector<float> hb((size_t)8 * vtw * 32);
    pack_genmax_x3(w_host, b_host, V, hw.data(), hb.data());
    std::vector<unsigned char> hx((size_t)M * 1024);
    for (size_t r = 0; r < (size_t)M; ++r)
        for (size_t c = 0; c < 256; ++c) {
            const float v = h_host[r * 256 + c];
            const uint16_t hi = f32_to_bf16_host(v);
            const uint32_t hbits = (uint32_t)hi << 16;
            float hf;
            std::memcpy(&hf, &hbits, 4);
            const uint16_t lo = f32_to_bf16_host(v - hf);
            std::memcpy(&hx[r * 1024 + cn_split_off(c)], &hi, 2);
            std::memcpy(&hx[r * 1024 + cn_split_off(c) + 64], &lo, 2);
        }
    void *dw = nullptr, *db = nullptr, *dx = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, hw.size() * 2));
    CN_HIP_CHECK(hipMalloc(&db, hb.size() * 4));
    CN_HIP_CHECK(hipMalloc(&dx, hx.size()));
    CN_HIP_CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice));
    GenmaxArgs a;
    a.x3 = true;
    a.h = dx;
    a.wp = dw;
    a.bp = (const float*)db;
    a.arg = arg_dev;
    a.maxlp = maxlp_dev;
    a.M = M;
    a.V = V;
    a.d = 256;
    a.tgt = tgt_dev;
    a.tgt_lp = tgt_lp_dev;
    a.tgt_U = U;
    a.tgt_ld = ld;
    int rc = launch_genmax(a, (hipStream_t)stream);
    if (const char* rep = cn_exp_env("CASSNAT_GENMAX_REPEAT"))  // timing runs only
        for (int i = 1, n = atoi(rep); rc == 0 && i < n; ++i) rc = launch_genmax(a, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(dw);
    (void)hipFree(db);
    (void)hipFree(dx);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_genmax_x3: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

// d_model-deep projection of the split-bf16 engine (proj_x3.hip): a_host fp32 [M][256] and w_host fp32 [N][256] are split into
// hi + lo halves, packed and uploaded by the call; bias_host [N].  split_out == 0: c_dev fp32 [M][N] = (resid_dev ? resid +
// resid_scale * : ) (A . W^T + bias), resid_dev fp32 [M][N] may alias c_dev;  split_out == 1: c_dev receives split-bf16 rows
// (M * N * 4 bytes: per 32 columns 64 bytes of hi halves, then 64 bytes of lo halves)
extern "C" int cn_op_proj_x3(const float* a_host, const float* w_host, const float* bias_host, const float* resid_dev,
                             float resid_scale, void* c_dev, int32_t M, int32_t N, int32_t split_out, void* stream) {
    if (!proj_x3_applies(N, 256) || M < 1) {
        cn_set_error("cn_op_proj_x3: N must be a multiple of 32, at most 1024");
        return -1;
    }
    std::vector<unsigned char> hw((size_t)N * 1024), hx((size_t)M * 1024);
    pack_proj_x3(w_host, N, hw.data());
    for (size_t r = 0; r < (size_t)M; ++r)
        for (size_t c = 0; c < 256; ++c) {
            const float v = a_host[r * 256 + c];
            const uint16_t hi = f32_to_bf16_host(v);
            const uint32_t hbits = (uint32_t)hi << 16;
            float hf;
            std::memcpy(&hf, &hbits, 4);
            const uint16_t lo = f32_to_bf16_host(v - hf);
            std::memcpy(&hx[r * 1024 + cn_split_off(c)], &hi, 2);
            std::memcpy(&hx[r * 1024 + cn_split_off(c) + 64], &lo, 2);
        }
    void *dw = nullptr, *db = nullptr, *dx = nullptr;
    CN_HIP_CHECK(hipMalloc(&dw, hw.size()));
    CN_HIP_CHECK(hipMalloc(&db, (size_t)N * 4));
    CN_HIP_CHECK(hipMalloc(&dx, hx.size()));
    CN_HIP_CHECK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(db, bias_host, (size_t)N * 4, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice));
    ProjX3Args a;
    a.A = dx;
    a.lda = 256;
    a.wp = dw;
    a.bias = (const float*)db;
    a.C = c_dev;
    a.ldc = N;
    a.c_f32 = split_out ? 0 : 1;
    a.resid = resid_dev;
    a.ldr = N;
    a.resid_scale = resid_scale;
    a.M = M;
    a.N = N;
    int rc = launch_proj_x3(a, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(dw);
    (void)hipFree(db);
    (void)hipFree(dx);
    if (rc == 0 && e != hipSuccess) {
        cn_set_error(std::string("cn_op_proj_x3: ") + hipGetErrorString(e));
        rc = -2;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// AST (autoregressive) path: BASELINE config 4
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int ast_alloc(cn_model* m, void** p, size_t bytes) {
    CN_HIP_CHECK(hipMalloc(p, bytes ? bytes : 256));
    m->ast_allocs.push_back(*p);
    return 0;
}

int ast_prepare_buffers(cn_model* m, int max_len, int max_slots, int ctc_beam) {
    if (m->ast_max_len >= max_len && m->ast_slots >= max_slots && m->ast_ctc_beam >= ctc_beam && m->ast_Tp_cap >= m->maxTp)
        return 0;
    for (void* q : m->ast_allocs) (void)hipFree(q);
    m->ast_allocs.clear();
    m->beam_S = m->beam_L = m->beam_K = 0;  // the beam-search state lives in the same allocation list
    m->ast_kvx.clear();
    m->ast_ck.clear();
    m->ast_cv.clear();
    const size_t d = m->cfg.d_model, V = m->cfg.vocab_size, es = m->es;
    const size_t Mmem = (size_t)m->maxB * m->maxTp;
    for (size_t l = 0; l < m->mad.size(); ++l) {
        void *kvx = nullptr, *ck = nullptr, *cv = nullptr;
        CN_TRY(ast_alloc(m, &kvx, Mmem * 2 * d * es));
        CN_TRY(ast_alloc(m, &ck, (size_t)max_len * max_slots * d * es));
        CN_TRY(ast_alloc(m, &cv, (size_t)max_len * max_slots * d * es));
        m->ast_kvx.push_back(kvx);
        m->ast_ck.push_back(ck);
        m->ast_cv.push_back(cv);
    }
    CN_TRY(ast_alloc(m, (void**)&m->ast_logits, (size_t)max_slots * V * 4));
    CN_TRY(ast_alloc(m, (void**)&m->ast_arg, (size_t)max_slots * 4));
    CN_TRY(ast_alloc(m, (void**)&m->ast_maxlp, (size_t)max_slots * 4));
    CN_TRY(ast_alloc(m, (void**)&m->ast_r0, Mmem * 2 * 4));
    const int kb = ctc_beam > 0 ? ctc_beam : 1;
    for (int i = 0; i < 2; ++i) CN_TRY(ast_alloc(m, (void**)&m->ast_r[i], (size_t)max_slots * kb * m->maxTp * 2 * 4));
    m->ast_max_len = max_len;
    m->ast_slots = max_slots;
    m->ast_ctc_beam = ctc_beam;
    m->ast_Tp_cap = m->maxTp;
    return 0;
}
}  // namespace

extern "C" int cn_ast_begin(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                            int32_t want_ctc, int32_t max_len, int32_t max_slots, int32_t ctc_beam, void* stream) {
    CN_TRY(check_call(m, B, T, F));
    if (!m->cfg.ast || !m->tgt_lut) {
        cn_set_error("cn_ast_begin: the model was not created with cfg.ast = 1");
        return -1;
    }
    if (max_len < 2 || max_slots < 1 || (size_t)max_slots > (size_t)m->maxB * (m->maxTp + 1) || ctc_beam < 0 || ctc_beam > 16) {
        cn_set_error("cn_ast_begin: bad max_len / max_slots / ctc_beam");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CN_TRY(ast_prepare_buffers(m, max_len, max_slots, ctc_beam));
    m->ast_blank = opts->padding_idx;
    CN_TRY(stage_encode(m, feats_dev, B, T, F, opts, s));
    const int d = m->cfg.d_model, Mmem = B * m->Tp, V = m->cfg.vocab_size;
    for (size_t l = 0; l < m->mad.size(); ++l)  // cross-attention K|V of the memory, once per utterance batch
        CN_TRY(run_linear(m, "src_kv_proj", m->mad[l].src_kv, m->enc_h, d, m->ast_kvx[l], 2 * d, 0, Mmem, 0, nullptr, 0, s));
    if (want_ctc) {
        CN_TRY(run_linear(m, "generator_proj", m->ctc_gen, m->enc_h, d, m->logits, V, 1, Mmem, 0, nullptr, 0, s));
        CN_TRY(launch_logsoftmax_argmax(m->logits, Mmem, V, V, m->best, m->ctc_maxlp, 1, s));
    m->ctc_maxlp_valid = true;
        CN_TRY(launch_ast_ctc_prepare(m->logits, m->keymask, m->ast_r0, B, m->Tp, V, opts->padding_idx, s));
    }
    return 0;
}

namespace {
// decoder layers on the newest position of n hypotheses -> top-K (token, temperature log-softmax value) per hypothesis
// dense_bw > 0: the rows are the dense slots b * dense_bw + j of the device beam (utt[row] = row / dense_bw)
int ast_step_run(cn_model* m, int n, int pos, const int32_t* tok_dev, const int32_t* utt_dev, const int32_t* anc_dev,
                 const uint8_t* keyok_dev, int table_stride, float temperature, int K, int32_t* topk_idx_dev,
                 float* topk_val_dev, int dense_bw, hipStream_t s) {
    const cn_config& c = m->cfg;
    const int d = c.d_model, V = c.vocab_size, H = c.n_head;
    const float scale = 1.0f / sqrtf((float)(d / H));
    float* x = m->xd;
    CN_TRY(launch_ast_embed(tok_dev, m->tgt_lut, m->pe + (size_t)pos * d, x, n, d, sqrtf((float)d), s));
    // A decode step has few rows (live hypotheses, <= batch x beam): the fused feed-forward kernel would put them on
    // ceil(n / 64) workgroups that each stream the whole 2 MB of W1|W2.  Its d_ff split spreads the hidden units over 8
    // workgroups per row tile; the reduce kernel that adds the slices also writes the LayerNorm the next sublayer starts
    // with (the next layer's self-attention pre-norm, or decoder.norm after the last layer).
    static const bool no_split = cn_exp_env("CASSNAT_AST_NO_FFN_SPLIT") != nullptr;
    const int ffn_slices = (m->mad.empty() || !m->mad[0].w1p || no_split) ? 1 : (c.d_decff % (128 * 8) == 0 ? 8 : (c.d_decff % (128 * 4) == 0 ? 4 : 1));
    bool have_ln = false;  // m->xn already holds LN(x) for the sublayer about to start
    // (only while the rows are few, and the slices fit the hidden-activation buffer they borrow)
    const size_t hbuf_bytes = (size_t)m->maxB * (m->maxTp + 1) * std::max(1, c.esa_group) *
                              (size_t)std::max(std::max(c.d_encff, c.d_decff), c.d_ff) * m->es;
    const bool split_now = ffn_slices > 1 && n <= 2048 && (size_t)ffn_slices * n * d * 4 <= hbuf_bytes;
    for (size_t l = 0; l < m->mad.size(); ++l) {
        const Layer& L = m->mad[l];
        // x += O(SelfAttn(LN x)) over the cached prefix
        if (!have_ln) CN_TRY(run_ln(m, L.n[0], x, m->xn, n, s));
        have_ln = false;
        CN_TRY(run_linear(m, "qkv_proj", L.qkv, m->xn, d, m->qkv, 3 * d, 0, n, 0, nullptr, 0, s));
        GatherAttnArgs a;
        a.append_pos = pos;  // the cache attention appends this position's K | V itself (it is the only reader of that row now)
        a.q = m->qkv;
        a.ldq = 3 * d;
        a.k = m->ast_ck[l];
        a.v = m->ast_cv[l];
        a.o = m->ctx;
        a.ldo = d;
        a.n = n;
        a.H = H;
        a.nkeys = pos + 1;
        a.slots = m->ast_slots;
        a.d = d;
        a.table_stride = table_stride;
        a.anc = anc_dev;
        a.keyok = keyok_dev;
        a.scale = scale;
        {
            ProfScope ps(m, "ast_cache_attention", 4.0 * n * H * (pos + 1) * 64, 2.0 * n * (pos + 1) * d * m->es, s);
            CN_TRY(launch_ast_gather_attn(m->prec, 0, a, s));
        }
        CN_TRY(run_linear(m, "out_proj_resid", L.self_o, m->ctx, d, x, d, 1, n, CN_EPI_RESID, x, d, s));
        // x += O(SrcAttn(LN x, memory))
        CN_TRY(run_ln(m, L.n[1], x, m->xn, n, s));
        CN_TRY(run_linear(m, "src_q_proj", L.src_q, m->xn, d, m->qd, d, 0, n, 0, nullptr, 0, s));
        GatherAttnArgs b;
        b.q = m->qd;
        b.ldq = d;
        b.k = m->ast_kvx[l];
        b.v = (const unsigned char*)m->ast_kvx[l] + (size_t)d * m->es;
        b.o = m->ctx;
        b.ldo = d;
        b.n = n;
        b.H = H;
        b.nkeys = m->Tp;
        b.d = d;
        b.utt = utt_dev;
        b.keymask = m->keymask;
        b.scale = scale;
        static const bool no_fast_src = cn_exp_env("CASSNAT_AST_GATHER_SRC") != nullptr;
        if (m->prec == CN_PREC_BF16 && m->Tp <= 256 && !no_fast_src) {
            // every hypothesis row is a batch entry of one query whose keys / values are its utterance's: the LDS-resident
            // attention kernel (K|V of a (row, head) by LDS-DMA, MFMA products) instead of the scalar gather kernel
            ProfScope ps(m, "ast_src_attention", 4.0 * n * H * m->Tp * 64, 2.0 * n * m->Tp * d * m->es, s);
            AttnArgs a2;
            a2.Q = m->qd;
            a2.K = m->ast_kvx[l];
            a2.V = (const unsigned char*)m->ast_kvx[l] + (size_t)d * m->es;
            a2.O = m->ctx;
            a2.ldq = d;
            a2.ldk = a2.ldv = 2 * d;
            a2.ldo = d;
            a2.H = H;
            a2.Lk = m->Tp;
            a2.keymask = m->keymask;
            if (dense_bw > 0 && n % dense_bw == 0) {  // device beam: utterance b owns rows b * bw .. + bw - 1
                a2.B = n / dense_bw;
                a2.Lq = dense_bw;
            } else {  // caller-managed rows: every row is a batch entry of one query that names its utterance
                a2.B = n;
                a2.Lq = 1;
                a2.kv_index = utt_dev;
            }
            a2.scale = scale;
            CN_TRY(launch_attention(m->prec, a2, s));
        } else {
            ProfScope ps(m, "ast_src_attention", 4.0 * n * H * m->Tp * 64, 2.0 * n * m->Tp * d * m->es, s);
            CN_TRY(launch_ast_gather_attn(m->prec, 1, b, s));
        }
        CN_TRY(run_linear(m, "out_proj_resid", L.src_o, m->ctx, d, x, d, 1, n, CN_EPI_RESID, x, d, s));
        if (split_now) {
            const bool last = l + 1 == m->mad.size();
            const Norm& nx = last ? m->dec_norm : m->mad[l + 1].n[0];
            ProfScope ps(m, "ffn_split", 4.0 * n * (double)L.w1.N * d, 2.0 * L.w1.N * d * 2 + (double)n * d * (8 + 4 * ffn_slices), s);
            FfnFusedArgs a;
            a.x = x;
            a.ln_a = L.n[2].a;
            a.ln_b = L.n[2].b;
            a.w1p = L.w1p;
            a.b1 = L.w1.b;
            a.w2p = L.w2p;
            a.b2 = L.w2.b;
            a.M = n;
            a.d = d;
            a.dff = L.w1.N;
            a.nslice = ffn_slices;
            a.partial = reinterpret_cast<float*>(m->hbuf);  // [slices][n][256] fp32 <= the [rows][d_ff] hidden buffer
            CN_TRY(launch_ffn_fused(a, s));
            CN_TRY(launch_ffn_reduce(x, a.partial, ffn_slices, L.w2.b, nx.a, nx.b, last ? m->dec_h : m->xn, n, 1e-6f, s));
            have_ln = true;
        } else {
            CN_TRY(run_ffn(m, L, L.n[2], x, n, nullptr, nullptr, s));
        }
    }
    if (!have_ln) CN_TRY(run_ln(m, m->dec_norm, x, m->dec_h, n, s));
    CN_TRY(run_linear(m, "generator_proj", m->att_gen, m->dec_h, d, m->ast_logits, V, 1, n, 0, nullptr, 0, s));
    CN_TRY(launch_logsoftmax_topk(m->ast_logits, n, V, V, temperature, K, topk_idx_dev, topk_val_dev, s));
    return 0;
}
}  // namespace

extern "C" int cn_ast_step(cn_model* m, int32_t n, int32_t pos, const int32_t* tok_dev, const int32_t* utt_dev,
                           const int32_t* anc_dev, const uint8_t* keyok_dev, int32_t table_stride, float temperature,
                           int32_t K, int32_t* topk_idx_dev, float* topk_val_dev, void* stream) {
    if (!m || !m->cfg.ast || m->ast_slots == 0) {
        cn_set_error("cn_ast_step: call cn_ast_begin first");
        return -1;
    }
    if (n < 1 || n > m->ast_slots || pos < 0 || pos >= m->ast_max_len || pos >= m->pe_rows || K < 1 || K > 16 ||
        table_stride <= pos) {
        cn_set_error("cn_ast_step: live rows / position / K outside the configured cache");
        return -1;
    }
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    return ast_step_run(m, n, pos, tok_dev, utt_dev, anc_dev, keyok_dev, table_stride, temperature, K, topk_idx_dev, topk_val_dev, 0,
                        (hipStream_t)stream);
}

extern "C" int cn_ast_ctc_score(cn_model* m, int32_t n, int32_t out_len, const int32_t* utt_dev, const int32_t* last_tok_dev,
                                const int32_t* cand_dev, int32_t K, const int32_t* prev_ref_dev, int32_t parity, int32_t eos,
                                float* score_dev, void* stream) {
    if (!m || !m->cfg.ast || !m->ast_r0 || K < 1 || K > m->ast_ctc_beam || n < 1 || n > m->ast_slots) {
        cn_set_error("cn_ast_ctc_score: call cn_ast_begin with want_ctc and a ctc_beam >= K first");
        return -1;
    }
    CN_HIP_CHECK(hipSetDevice(m->cfg.device));
    CtcPrefixArgs a;
    a.logp = m->logits;
    a.r0 = m->ast_r0;
    a.r_prev = m->ast_r[(parity & 1) ^ 1];
    a.r_new = m->ast_r[parity & 1];
    a.utt = utt_dev;
    a.last_tok = last_tok_dev;
    a.cand = cand_dev;
    a.prev_ref = prev_ref_dev;
    a.score = score_dev;
    a.n = n;
    a.K = K;
    a.Tp = m->Tp;
    a.V = m->cfg.vocab_size;
    a.blank = m->ast_blank;
    a.eos = eos;
    a.out_len = out_len;
    return launch_ast_ctc_prefix(a, (hipStream_t)stream);
}

// Whole joint CTC/attention beam search on the device (transformer.py:122-241): encoder, then max_step iterations of
// [decoder step on every slot -> CTC prefix scores -> beam update kernel]; the host only polls the live-hypothesis
// counter every 8 steps to stop early.  Outputs (device pointers): hyp_out [B][beam_width][max_len] (sos first, padded
// with padding_idx), hyp_len [B][beam_width], score [B][beam_width] as double; beams best first.
extern "C" int cn_decode_ast(cn_model* m, const float* feats_dev, int32_t B, int32_t T, int32_t F, const cn_decode_opts* opts,
                             const cn_ast_opts* ao, int32_t* hyp_out_dev, int32_t max_len, int32_t* hyp_len_dev,
                             double* score_dev, void* stream) {
    if (!m || !opts || !ao || !hyp_out_dev || !hyp_len_dev || !score_dev) {
        cn_set_error("cn_decode_ast: null argument");
        return -1;
    }
    const int bw = ao->beam_width, want_ctc = ao->ctc_weight > 0.f ? 1 : 0;
    const int K = want_ctc ? ao->ctc_beam : bw;
    if (bw < 1 || bw > 16 || K < bw || K > 16 || ao->max_step < 1 || max_len < ao->max_step + 1) {
        cn_set_error("cn_decode_ast: need 1 <= beam_width <= ctc_beam <= 16 and max_len > max_step");
        return -1;
    }
    const int S = B * bw, L = max_len;
    CN_TRY(cn_ast_begin(m, feats_dev, B, T, F, opts, want_ctc, L, S, want_ctc ? K : 0, stream));
    hipStream_t s = (hipStream_t)stream;
    if (m->beam_S < S || m->beam_L < L || m->beam_K < K) {
        AstBeamState& st = m->beam;
        for (int i = 0; i < 2; ++i) {
            CN_TRY(ast_alloc(m, (void**)&st.tok[i], (size_t)S * L * 4));
            CN_TRY(ast_alloc(m, (void**)&st.anc[i], (size_t)S * L * 4));
            CN_TRY(ast_alloc(m, (void**)&st.keyok[i], (size_t)S * L));
            CN_TRY(ast_alloc(m, (void**)&st.len[i], (size_t)S * 4));
            CN_TRY(ast_alloc(m, (void**)&st.score[i], (size_t)S * 8));
            CN_TRY(ast_alloc(m, (void**)&st.valid[i], (size_t)S * 4));
            CN_TRY(ast_alloc(m, (void**)&st.ctc_ref[i], (size_t)S * 4));
            CN_TRY(ast_alloc(m, (void**)&st.ctc_prev[i], (size_t)S * 4));
        }
        CN_TRY(ast_alloc(m, (void**)&st.cur_tok, (size_t)S * 4));
        CN_TRY(ast_alloc(m, (void**)&st.utt, (size_t)S * 4));
        CN_TRY(ast_alloc(m, (void**)&st.live, 64));
        CN_TRY(ast_alloc(m, (void**)&m->beam_idx, (size_t)S * K * 4));
        CN_TRY(ast_alloc(m, (void**)&m->beam_val, (size_t)S * K * 4));
        CN_TRY(ast_alloc(m, (void**)&m->beam_ctc, (size_t)S * K * 4));
        m->beam_S = S;
        m->beam_L = L;
        m->beam_K = K;
    }
    const AstBeamState& st = m->beam;
    int cur = 0;
    CN_TRY(launch_ast_beam_init(st, cur, B, bw, L, opts->sos, opts->padding_idx, s));
    for (int pos = 0; pos < ao->max_step; ++pos) {
        CN_TRY(ast_step_run(m, S, pos, st.cur_tok, st.utt, st.anc[cur], st.keyok[cur], L, ao->temperature, K, m->beam_idx,
                            m->beam_val, bw, s));
        if (want_ctc)
            CN_TRY(cn_ast_ctc_score(m, S, pos, st.utt, st.cur_tok, m->beam_idx, K, st.ctc_ref[cur], pos & 1, ao->eos, m->beam_ctc,
                                    stream));
        AstBeamStep q;
        q.idx = m->beam_idx;
        q.att = m->beam_val;
        q.ctc = m->beam_ctc;
        q.cur = cur;
        q.pos = pos;
        q.bw = bw;
        q.K = K;
        q.L = L;
        q.eos = ao->eos;
        q.sos = opts->sos;
        q.pad = opts->padding_idx;
        q.use_ctc = want_ctc;
        q.use_lp = ao->use_length_penalty;
        q.w = ao->ctc_weight;
        q.u = ao->one_minus_ctc_weight;
        q.lp = ao->length_penalty;
        CN_TRY(launch_ast_beam_update(st, q, B, s));
        cur ^= 1;
        if ((pos & 7) == 7 && pos + 1 < ao->max_step) {  // every 8 steps: anything still alive?
            CN_HIP_CHECK(hipMemcpyAsync(m->ymax_pinned + 8, st.live, sizeof(int), hipMemcpyDeviceToHost, s));
            CN_HIP_CHECK(hipStreamSynchronize(s));
            if (m->ymax_pinned[8] == 0) break;
        }
    }
    CN_HIP_CHECK(hipMemcpyAsync(hyp_out_dev, st.tok[cur], (size_t)S * L * 4, hipMemcpyDeviceToDevice, s));
    CN_HIP_CHECK(hipMemcpyAsync(hyp_len_dev, st.len[cur], (size_t)S * 4, hipMemcpyDeviceToDevice, s));
    CN_HIP_CHECK(hipMemcpyAsync(score_dev, st.score[cur], (size_t)S * 8, hipMemcpyDeviceToDevice, s));
    return 0;
}
