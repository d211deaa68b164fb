// Fused multi-head attention for gfx950 (d_k = 64): QK^T -> scale -> mask -> softmax -> .V in one kernel,
// scores never reach HBM.  Replaces `attention` + the head split/merge copies of MultiHeadedAttention
// (reference: src/models/modules/attention.py:13-24, 44-66).
//
// Mask semantics are the reference's: a masked key gets the score float32-min (NOT -inf), so a query row
// whose keys are all masked attends uniformly over all Lk keys (attention.py:19-21).  Keys that exist only
// because the last tile is padded get -inf and never contribute.  Three mask sources are ANDed:
//   keymask[b][j] (padding mask after subsampling), klen[b] (j < klen: target mask u < ylen),
//   per-query frame intervals (s1,e1,s2,e2) (the CTC trigger mask, see ctc_align.hip), and `causal`.
//
// Structure: workgroup = 4 waves = 128 query rows of one (batch, head); each wave keeps its 32 query rows
// as MFMA operand fragments in registers.  K/V tiles of 64 keys are staged global -> registers -> LDS
// (next tile's loads are in flight during the current tile's MFMAs).  The score product is computed
// "swapped" (S^T = K.Q^T, keys on accumulator rows, the query on the lane) so that
//   * the softmax row reduction is 32 in-lane values + one cross-half exchange (wave shuffle), and
//   * the probability tile is already laid out as the B operand of the next MFMA (O^T = V^T.P^T):
//     no LDS round trip for P.  V^T fragments come from LDS with ds_read_b64_tr_b16 (bf16) or plain
//     ds_read_b32 (f32 path, v_mfma_f32_32x32x2_f32).
#include <cstdlib>

#include <cstdio>
#include <cstdlib>

#include "kernels.h"

struct AttnParams {
    const unsigned char* Q;
    const unsigned char* K;
    const unsigned char* V;
    void* O;
    long long ldq_b, ldk_b, ldv_b;  // row strides in bytes
    int ldo;
    int H, Lq, Lk;
    const unsigned char* keymask;
    int kv_mod;  // > 0: keys / values / keymask of batch entry b live at entry b % kv_mod (several query sets per source)
    int stamps;  // investigation aid (CASSNAT_ATTN_STAMPS): the last workgroup's thread 0 records s_memtime at its phase boundaries
    const int* kv_index;  // non-null: ... at entry kv_index[b] (beam search: every hypothesis row names its utterance)
    const int* klen;
    int o_blk;          // bf16: O in the row-chain kernel's B-operand order (AttnArgs.o_blocked)
    int q_blk, kv_blk;  // bf16: Q / K|V are blocked matrices (cn_blk16_off) of q_n / kv_n columns, head 0 at column q_col / k_col / v_col
    int q_col, k_col, v_col, q_n, kv_n;
    const int* kcap;  // keys the K / V entry's own batch has (merged passes; null: Lk) - later keys get -inf, not the float-min fill
    int kcap_stride;
    const int* iv;
    int iv_stride;
    int causal;
    float scale;
    const float* rel_pos;  // relative-position self attention (see AttnArgs); null otherwise
    const float* rel_u;
    const float* rel_v;
    int rel_R, ld_pos;
};

template <typename T> struct AttnCfg;
template <> struct AttnCfg<bf16> {
    static constexpr int KROW = 128;  // bytes per K row in LDS (64 bf16), 8 chunks, swizzle (row>>1)&7
    static constexpr int VROW = 192;  // 128 B of data + 64 B pad: 4 consecutive rows hit disjoint banks for tr reads
    static constexpr int CPR = 8;
    __device__ static int swz(int row) { return (row >> 1) & 7; }
};
template <> struct AttnCfg<float> {
    static constexpr int KROW = 256;  // 64 f32, 16 chunks, swizzle row&15
    static constexpr int VROW = 256;
    static constexpr int CPR = 16;
    __device__ static int swz(int row) { return row & 15; }
};

template <> struct AttnCfg<split_t> {   // split-bf16 rows: per head [32 hi][32 lo][32 hi][32 lo] = 256 B, 16 chunks
    static constexpr int KROW = 256;
    static constexpr int VROW = 192;  // per PLANE (hi / lo): 64 bf16 of data + 64 B pad, as the bf16 tile
    static constexpr int CPR = 16;
    __device__ static int swz(int row) { return row & 15; }
};

// NW waves per workgroup (32 query rows each): 4 -> 128-row tiles, 2 -> 64-row tiles (more, smaller workgroups when
// the grid would otherwise be ~1 workgroup per CU with nothing to overlap its barriers and load latency)
// RES (bf16, Lk <= 256): ALL keys/values of the (batch, head) are brought into LDS at once by LDS-DMA (the XOR swizzles
// are applied to the per-lane SOURCE address, the LDS image of a DMA piece is lane-linear), then every key tile is
// processed back to back: one barrier per workgroup instead of two per tile, and a single load round trip.
// REL: relative-position scores (RelMultiHeadedAttention): the query operand is q + u, and a per-query table
// bd[i][r] = (q_i + v) . P[r] (r = clamp(j - i) + R, at most 63 entries) is built in LDS and added to every raw score.
__device__ long long attn_stamps[8];
// (the LAST workgroup of the grid: with more workgroups than CUs it runs among others in every phase, not in the launch's first burst)
#define AT_STAMP(i) if (p.stamps && blockIdx.x == 0 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1 && threadIdx.x == 0) attn_stamps[i] = (long long)__builtin_amdgcn_s_memtime();

// (second launch bound = waves per SIMD the register budget must allow: the resident 8-wave form runs two workgroups per CU - one
// loading while the other computes - which needs <= 128 VGPRs; the allocator otherwise lets the seldom-taken full mask path
// push it to 132 and halves the occupancy: -2 % on the whole benchmark)
// (split-bf16, the staged 4-wave form - the split-bf16 engine's encoder self attention: left alone the allocator takes 300 registers
// = one wave per SIMD = ONE 256-thread workgroup per CU, with two barriers and a global load round trip per key tile and nothing to
// overlap them with; bounded to 256 registers two workgroups share a CU)
template <typename T, int NW, bool RES, bool REL = false>
__global__ __launch_bounds__(64 * NW, (RES && NW == 8) ? 4 : ((__is_same(T, split_t) && NW == 4 && !REL) ? 2 : 1)) void attention_kernel(AttnParams p) {
    typedef AttnCfg<T> Cfg;
    typedef typename Frag<T>::type frag_t;
    constexpr bool SPLIT = __is_same(T, split_t);
    static_assert(!SPLIT || !RES, "split-bf16 attention: the staged forms only");
    constexpr int KROW = Cfg::KROW, CPR = Cfg::CPR;
    constexpr int VROW = RES ? 128 : Cfg::VROW;  // RES: unpadded rows, 64-byte halves swapped on rows with (row>>1)&1
    constexpr int KT_BYTES = 64 * KROW, VT_BYTES = (SPLIT ? 2 : 1) * 64 * VROW, NRES = RES ? 4 : 1;
    constexpr int NF = SPLIT ? 4 : KROW / 32;  // fragments of one 64-wide head row per lane half (split: k-steps of 16, hi + lo each)
    constexpr int NT = 64 * NW;
    constexpr int ST_IT = 64 * CPR / NT;    // 16-byte chunks per thread per 64-key tile

    __shared__ __attribute__((aligned(16))) unsigned char Ks_all[NRES * KT_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Vs_all[NRES * VT_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned int Ms_all[NRES * 16];  // mask bytes: 0 masked, 1 allowed, 2 tile padding
    // per key tile: 1 = every key is allowed (mask-free fast path); | 2 = some keys lie past the entry's own batch (code 2: a
    // suffix, key >= kcap - the "cut" path: one compare + select per score); | 4 = some keys are masked (code 0: full mask logic)
    __shared__ unsigned int Mplain[NRES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    AT_STAMP(0)
    const int q_row = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool wave_active = (blockIdx.x * (32 * NW) + wave * 32) < p.Lq;
    const int qc = q_row < p.Lq ? q_row : p.Lq - 1;

    // ---- query fragments (B operand of S^T = K.Q^T): lane holds Q[q][16-byte chunk 2s+half]
    frag_t qf[NF];
    {
        const unsigned char* qp = p.Q + ((long long)b * p.Lq + qc) * p.ldq_b + (long long)h * KROW;
        if constexpr (SPLIT) {
#pragma unroll
            for (int s = 0; s < NF; ++s) {  // k-step s: group s >> 1, chunk 2 (s & 1) + half of its hi quarter; lo 64 B further
                const unsigned char* c = qp + (s >> 1) * 128 + (2 * (s & 1) + half) * 16;
                qf[s].hi = as_frag<bf16>(ld16(c));
                qf[s].lo = as_frag<bf16>(ld16(c + 64));
            }
        } else if (p.q_blk) {
            const long long qm = (long long)b * p.Lq + qc;
#pragma unroll
            for (int s = 0; s < NF; ++s) qf[s] = as_frag<T>(ld16(p.Q + cn_blk16_off(qm, p.q_col + h * 64 + (2 * s + half) * 8, p.q_n)));
        } else {
#pragma unroll
            for (int s = 0; s < NF; ++s) qf[s] = as_frag<T>(ld16(qp + (2 * s + half) * 16));
        }
    }
    constexpr int REL_N = REL ? 64 : 1;
    __shared__ float rel_tab[REL ? 64 * 64 : 1];        // P rows of this head: [2R+1][64]
    __shared__ float rel_bias[REL ? NW : 1][32][REL_N];  // bd[i][r] per wave
    if constexpr (REL) {
        constexpr int E = SPLIT ? 8 : Frag<T>::ELEMS;
        const int nr = 2 * p.rel_R + 1;
        for (int i = tid; i < nr * 64; i += NT) rel_tab[i] = p.rel_pos[(long long)(i >> 6) * p.ld_pos + h * 64 + (i & 63)];
        __syncthreads();
        // this lane holds half of the query row - dims (2s + half) * E + j (split rows: k-step s holds dims 32 (s >> 1) +
        // 16 (s & 1) + 8 half + j, hi and lo halves) - the partner lane (lane ^ 32) holds the rest
        auto dim_of = [&](int s, int j) -> int {
            if constexpr (SPLIT) return 32 * (s >> 1) + 16 * (s & 1) + 8 * half + j;
            else return (2 * s + half) * E + j;
        };
        float qv[NF][E];
#pragma unroll
        for (int s = 0; s < NF; ++s)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int dim = dim_of(s, j);
                if constexpr (SPLIT) {
                    const float q = (float)qf[s].hi[j] + (float)qf[s].lo[j];
                    qv[s][j] = q + p.rel_v[h * 64 + dim];
                    const float qu = q + p.rel_u[h * 64 + dim];
                    const bf16 hi = (bf16)qu;
                    qf[s].hi[j] = hi;
                    qf[s].lo[j] = (bf16)(qu - (float)hi);
                } else {
                    const float q = to_f32(qf[s][j]);
                    qv[s][j] = q + p.rel_v[h * 64 + dim];
                    qf[s][j] = from_f32<T>(q + p.rel_u[h * 64 + dim]);
                }
            }
        for (int r = 0; r < nr; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < NF; ++s)
#pragma unroll
                for (int j = 0; j < E; ++j) acc = fmaf(qv[s][j], rel_tab[r * 64 + dim_of(s, j)], acc);
            acc += __shfl_xor(acc, 32);
            if (half == 0) rel_bias[wave][l31][r] = acc;
        }
        __syncthreads();
    }
    int iv_s1 = 0, iv_e1 = 0, iv_s2 = 0, iv_e2 = 0;
    if (p.iv) {
        const int4 r = *reinterpret_cast<const int4*>(p.iv + ((long long)b * p.iv_stride + qc) * 4);
        iv_s1 = r.x;
        iv_e1 = r.y;
        iv_s2 = r.z;
        iv_e2 = r.w;
    }
    const int klen = p.klen ? p.klen[b] : p.Lk;
    unsigned char* Ks = Ks_all;
    unsigned char* Vs = Vs_all;
    unsigned int* Ms = Ms_all;

    // ---- staging registers for the next K/V tile
    uint4 k_reg[ST_IT], v_reg[ST_IT];
    const int bk = p.kv_index ? p.kv_index[b] : (p.kv_mod > 0 ? b % p.kv_mod : b);
    const int kcap = p.kcap ? p.kcap[(long long)bk * p.kcap_stride] : p.Lk;
    // key tiles that hold at least one key of the entry's own batch: a later tile's keys all carry probability exactly 0 (code
    // 2, -inf) - in a merged pass of ragged batches most tiles of a short utterance (round 3 still loaded and multiplied them)
    const int kvalid = kcap < p.Lk ? kcap : p.Lk;
    const int nkt = kvalid > 0 ? (kvalid + 63) / 64 : 1;
    const unsigned char* kbase = p.K + (long long)bk * p.Lk * p.ldk_b + (long long)h * KROW;
    const unsigned char* vbase = p.V + (long long)bk * p.Lk * p.ldv_b + (long long)h * KROW;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < ST_IT; ++i) {
            const int cidx = tid + NT * i;
            const int row = cidx / CPR, ch = cidx % CPR;
            const int key = kt * 64 + row;
            if (key < p.Lk) {
                if (p.kv_blk) {
                    const long long km = (long long)bk * p.Lk + key;
                    k_reg[i] = ld16(p.K + cn_blk16_off(km, p.k_col + h * 64 + ch * 8, p.kv_n));
                    v_reg[i] = ld16(p.V + cn_blk16_off(km, p.v_col + h * 64 + ch * 8, p.kv_n));
                } else {
                    k_reg[i] = ld16(kbase + (long long)key * p.ldk_b + ch * 16);
                    v_reg[i] = ld16(vbase + (long long)key * p.ldv_b + ch * 16);
                }
            } else {
                k_reg[i] = make_uint4(0, 0, 0, 0);
                v_reg[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < ST_IT; ++i) {
            const int cidx = tid + NT * i;
            const int row = cidx / CPR, ch = cidx % CPR;
            st16(Ks + row * KROW + ((ch ^ Cfg::swz(row)) << 4), k_reg[i]);
            if constexpr (SPLIT)  // chunk ch = 8 g + c: c < 4 hi, else lo, of head dims 32 g + 8 (c & 3) ..; one LDS plane each
                st16(Vs + ((ch >> 2) & 1) * 64 * VROW + row * VROW + ((((ch >> 3) << 2) | (ch & 3)) << 4), v_reg[i]);
            else
                st16(Vs + row * VROW + (ch << 4), v_reg[i]);
        }
        if (tid < 64) {
            const int key = kt * 64 + tid;
            unsigned char code = 2;
            if (key < kcap) {
                bool ok = key < klen;
                if (p.keymask) ok = ok && p.keymask[(long long)bk * p.Lk + key] != 0;
                code = ok ? 1 : 0;
            }
            reinterpret_cast<unsigned char*>(Ms)[tid] = code;
            const unsigned long long bad0 = __ballot(code == 0), bad2 = __ballot(code == 2);
            if (tid == 0) Mplain[0] = 1u | (bad2 != 0ull ? 2u : 0u) | (bad0 != 0ull ? 4u : 0u);
        }
    };

    f32x16 o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = CN_NEG_FILL, l_run = 0.f;

    if constexpr (RES) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int r8 = lane >> 3, cp = lane & 7;
        for (int piece = wave_u; piece < nkt * 16; piece += NW) {  // 1-KiB pieces: 8 rows x 128 B of K or V
            const int kt2 = piece >> 4, is_v = (piece >> 3) & 1, j = piece & 7;
            const int row = 8 * j + r8;
            int key = kt2 * 64 + row;
            if (key >= p.Lk) key = p.Lk - 1;  // finite filler; its probability is exactly 0 (mask code 2)
            const int chunk = is_v ? (cp ^ (((row >> 1) & 1) << 2)) : (cp ^ Cfg::swz(row));
            const unsigned char* src;
            if (p.kv_blk)
                src = (is_v ? p.V : p.K) + cn_blk16_off((long long)bk * p.Lk + key, (is_v ? p.v_col : p.k_col) + h * 64 + chunk * 8, p.kv_n);
            else
                src = is_v ? vbase + (long long)key * p.ldv_b + (chunk << 4) : kbase + (long long)key * p.ldk_b + (chunk << 4);
            unsigned char* dst = (is_v ? Vs_all + kt2 * VT_BYTES : Ks_all + kt2 * KT_BYTES) + j * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        AT_STAMP(1)
        if (tid < NRES) Mplain[tid] = 1;
        __syncthreads();
        for (int i = tid; i < nkt * 64; i += NT) {
            unsigned char code = 2;
            if (i < kcap) {
                bool ok = i < klen;
                if (p.keymask) ok = ok && p.keymask[(long long)bk * p.Lk + i] != 0;
                code = ok ? 1 : 0;
            }
            reinterpret_cast<unsigned char*>(Ms_all)[i] = code;
            if (code != 1) atomicOr(&Mplain[i >> 6], code == 2 ? 2u : 4u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        AT_STAMP(2)
    } else {
        load_tile(0);
    }
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (RES) {
            Ks = Ks_all + kt * KT_BYTES;
            Vs = Vs_all + kt * VT_BYTES;
            Ms = Ms_all + kt * 16;
        } else {
            __syncthreads();  // everyone is done reading the previous tile
            store_tile(kt);
            __syncthreads();
            if (kt + 1 < nkt) load_tile(kt + 1);
        }
        if (!wave_active) continue;

        // The 64-key tile is consumed as two 32-key sub-tiles, each with its own online-softmax step: 16 score registers live
        // instead of 32, which is what lets the bf16 kernel run at 128 VGPRs - two 8-wave workgroups per CU, one loading
        // while the other computes.
        const unsigned tile_class = Mplain[RES ? kt : 0];
        const bool simple = !REL && !p.iv && !p.causal && (tile_class & 4u) == 0;  // no masked key, no per-row limits
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (simple && kt * 64 + sub * 32 >= kcap) continue;  // (wave-uniform) nothing but keys of probability 0
            // ---- S^T[key][q] of the sub-tile
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
            {
                const int row = sub * 32 + l31;
#pragma unroll
                for (int s = 0; s < NF; ++s) {
                    if constexpr (SPLIT) {
                        const int chunk = (s >> 1) * 8 + 2 * (s & 1) + half;
                        split_frag kf;
                        kf.hi = as_frag<bf16>(ld16(Ks + row * KROW + ((chunk ^ Cfg::swz(row)) << 4)));
                        kf.lo = as_frag<bf16>(ld16(Ks + row * KROW + (((chunk + 4) ^ Cfg::swz(row)) << 4)));
                        sc = mfma_frag(kf, qf[s], sc);
                    } else {
                        const int chunk = 2 * s + half;
                        const frag_t kf = as_frag<T>(ld16(Ks + row * KROW + ((chunk ^ Cfg::swz(row)) << 4)));
                        sc = mfma_frag(kf, qf[s], sc);
                    }
                }
            }
            // ---- scale + mask + online softmax.  Fast path (wave-uniform): no key of this tile is masked and there are no
            // per-row intervals / causal limit -> one FMA + exp2 per score instead of ~15 VALU ops of mask logic.
            float psum = 0.f, alpha;
            // (wave-uniform) the sub-tile's 32 keys all belong to the entry's own batch
            const bool plain = simple && kt * 64 + sub * 32 + 32 <= kcap;
            if (simple) {
                float tmax;
                if (plain) {
                    tmax = sc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
                } else {  // cut: keys >= kcap get -inf (probability exactly 0), the others are allowed
                    const int k0 = kt * 64 + sub * 32 + 4 * half;
                    tmax = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sc[r] = k0 + 8 * (r >> 2) + (r & 3) < kcap ? sc[r] : -INFINITY;
                        tmax = fmaxf(tmax, sc[r]);
                    }
                }
                tmax = xhalf_max(tmax) * p.scale;
                const float m_new = fmaxf(m_run, tmax);
                alpha = __expf(m_run - m_new);
                m_run = m_new;
                const float c2 = p.scale * 1.44269504088896340736f, mb = m_new * 1.44269504088896340736f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -mb));
                    sc[r] = pv;
                    psum += pv;
                }
            } else {
                // Full mask logic.  An allowed key's probability is computed exactly as on the fast paths (one FMA on the raw
                // score + exp2), so that WHICH path a tile takes never shows in the result - a batch decoded alone and in a merged
                // pass classifies the keys past its own row count differently (absent / masked) and must still agree bit for bit.
                // sc[r] keeps the raw score of an allowed key; a masked key is marked by the fill value, a key past the entry's
                // batch by -inf
                float tmax = -INFINITY;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned int mw = Ms[sub * 8 + 2 * g + half];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const int key = kt * 64 + sub * 32 + 8 * g + 4 * half + e;
                        const unsigned int code = (mw >> (8 * e)) & 0xffu;
                        bool ok = code == 1u;
                        if (p.iv) ok = ok && ((key >= iv_s1 && key < iv_e1) || (key >= iv_s2 && key < iv_e2));
                        if (p.causal) ok = ok && key <= q_row;
                        float v = sc[r];
                        if constexpr (REL) {
                            int rd = key - q_row;
                            rd = rd < -p.rel_R ? -p.rel_R : (rd > p.rel_R ? p.rel_R : rd);
                            v += rel_bias[wave][l31][rd + p.rel_R];
                        }
                        v = ok ? v : CN_NEG_FILL;
                        v = code == 2u ? -INFINITY : v;
                        sc[r] = v;
                        tmax = fmaxf(tmax, ok ? v * p.scale : v);
                    }
                }
                tmax = xhalf_max(tmax);
                const float m_new = fmaxf(m_run, tmax);
                alpha = __expf(m_run - m_new);
                m_run = m_new;
                const float c2 = p.scale * 1.44269504088896340736f, mb = m_new * 1.44269504088896340736f;
                const float p_fill = __expf(CN_NEG_FILL - m_new);  // 1 while every key of the row so far is masked, else 0
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = sc[r];
                    float pv = __builtin_amdgcn_exp2f(fmaf(v, c2, -mb));
                    pv = v == CN_NEG_FILL ? p_fill : pv;
                    pv = v == -INFINITY ? 0.f : pv;
                    sc[r] = pv;
                    psum += pv;
                }
            }
            l_run = l_run * alpha + psum;
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
            }

            // ---- O^T[dk][q] += V^T[dk][key] . P^T[key][q]
            if constexpr (SPLIT) {
                // as the bf16 form below, on the hi and lo planes of V and the hi / lo halves of P: three MFMAs per product
                const int i16 = lane & 15, g1 = (lane >> 4) & 1;
                typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ph[j] = (bf16)sc[8 * s + j];
                        pl[j] = (bf16)(sc[8 * s + j] - (float)ph[j]);
                    }
                    const int key0 = sub * 32 + 16 * s + 4 * half + (i16 >> 2);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        bf16x8 vf[2];
#pragma unroll
                        for (int pl_ = 0; pl_ < 2; ++pl_) {
                            const unsigned char* a1 = Vs + pl_ * 64 * VROW + key0 * VROW + (32 * d + 16 * g1 + 4 * (i16 & 3)) * 2;
                            const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
                            const s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1 + 8 * VROW));
                            s16x8 cat;
                            cat[0] = r1[0]; cat[1] = r1[1]; cat[2] = r1[2]; cat[3] = r1[3];
                            cat[4] = r2[0]; cat[5] = r2[1]; cat[6] = r2[2]; cat[7] = r2[3];
                            vf[pl_] = __builtin_bit_cast(bf16x8, cat);
                        }
                        o_acc[d] = CN_MFMA16(vf[1], ph, o_acc[d], 0, 0, 0);
                        o_acc[d] = CN_MFMA16(vf[0], pl, o_acc[d], 0, 0, 0);
                        o_acc[d] = CN_MFMA16(vf[0], ph, o_acc[d], 0, 0, 0);
                    }
                }
            } else if constexpr (sizeof(T) == 2) {
                const int i16 = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 pb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = (bf16)sc[8 * s + j];
                    const int key0 = sub * 32 + 16 * s + 4 * half + (i16 >> 2);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const int dd = RES ? (d ^ ((i16 >> 3) & 1)) : d;  // RES: 64-byte halves swapped on rows with (row>>1)&1
                        const unsigned char* a1 = Vs + key0 * VROW + (32 * dd + 16 * g1 + 4 * (i16 & 3)) * 2;
                        const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(a1));
                        const s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(a1 + 8 * VROW));
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        s16x8 cat;
                        cat[0] = r1[0]; cat[1] = r1[1]; cat[2] = r1[2]; cat[3] = r1[3];
                        cat[4] = r2[0]; cat[5] = r2[1]; cat[6] = r2[2]; cat[7] = r2[3];
                        const bf16x8 vf = __builtin_bit_cast(bf16x8, cat);
                        o_acc[d] = CN_MFMA16(vf, pb, o_acc[d], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key_local = sub * 32 + acc_row(r, lane);
                    const float* vrow = reinterpret_cast<const float*>(Vs + key_local * VROW);
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * d + l31], sc[r], o_acc[d], 0, 0, 0);
                }
            }
        }
    }

    AT_STAMP(3)
    if (!wave_active) return;
    const float l_tot = xhalf_sum(l_run);
    // REL: softmax(...).masked_fill(mask == 0, 0) leaves a row without any allowed key at zero (attention.py:133-134)
    const float inv = (REL && m_run == CN_NEG_FILL) ? 0.f : 1.f / l_tot;
    if constexpr (sizeof(T) == 2) {
        // bf16: a lane holds channels 8 g + 4 half + (0..3) of its row - sixteen 8-byte stores.  The two half-waves trade their odd /
        // even groups (v_permlane32_swap: upper half of the first operand <-> lower half of the second), after which a lane owns
        // 8 consecutive channels: eight 16-byte stores (the tail is bound by store instructions, not bytes).  The swap needs all
        // 64 lanes, so it runs for rows past Lq too; only the stores are guarded.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        T* orow = reinterpret_cast<T*>(p.O) + ((long long)b * p.Lq + qc) * p.ldo + h * 64 + 8 * half;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                bf16x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = (bf16)(o_acc[d][8 * gp + e] * inv);
                    o1[e] = (bf16)(o_acc[d][8 * gp + 4 + e] * inv);
                }
                const uint2 lo_ = __builtin_bit_cast(uint2, o0), hi_ = __builtin_bit_cast(uint2, o1);
                const auto s0_ = __builtin_amdgcn_permlane32_swap(lo_.x, hi_.x, false, false);
                const auto s1_ = __builtin_amdgcn_permlane32_swap(lo_.y, hi_.y, false, false);
                if (q_row < p.Lq) {
                    if (p.o_blk) {  // channels h * 64 + 32 d + 16 gp + 8 half .. + 7 = k-step 4 h + 2 d + gp of the chain's B operand
                        const long long om = (long long)b * p.Lq + q_row;
                        unsigned char* ob = reinterpret_cast<unsigned char*>(p.O) +
                                            ((om >> 5) * (p.ldo >> 4) + (4 * h + 2 * d + gp)) * 1024 + ((half << 5) + (int)(om & 31)) * 16;
                        *reinterpret_cast<u32x4*>(ob) = u32x4{s0_[0], s1_[0], s0_[1], s1_[1]};
                    } else {
                        *reinterpret_cast<u32x4*>(orow + 32 * d + 16 * gp) = u32x4{s0_[0], s1_[0], s0_[1], s1_[1]};
                    }
                }
            }
        }
        return;
    }
    if (q_row < p.Lq) {
        T* orow = reinterpret_cast<T*>(p.O) + ((long long)b * p.Lq + q_row) * p.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dk0 = 32 * d + 8 * g + 4 * half;
                if constexpr (SPLIT) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o_acc[d][4 * g + e] * inv;
                    bf16x4 hi, lo;
                    cn_split4(o, hi, lo);
                    unsigned char* ob = reinterpret_cast<unsigned char*>(orow) + cn_split_off((size_t)dk0);
                    *reinterpret_cast<bf16x4*>(ob) = hi;
                    *reinterpret_cast<bf16x4*>(ob + 64) = lo;
                } else if constexpr (sizeof(T) == 2) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)(o_acc[d][4 * g + e] * inv);
                    *reinterpret_cast<bf16x4*>(orow + dk0) = o;
                } else {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o_acc[d][4 * g + e] * inv;
                    *reinterpret_cast<f32x4*>(orow + dk0) = o;
                }
            }
        }
    }
    AT_STAMP(4)
}

int attention_print_stamps() {
    long long h[8];
    CN_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(attn_stamps), sizeof(h)));
    static const char* names[] = {"", "Q fragments + K/V requests", "K/V landed + mask codes", "key-tile loop", "normalise + store"};
    for (int i = 1; i < 5; ++i) fprintf(stderr, "[attention stamps] %-28s %8lld ticks\n", names[i], h[i] - h[i - 1]);
    fprintf(stderr, "[attention stamps] total %lld ticks (s_memtime)\n", h[4] - h[0]);
    return 0;
}

template <typename T> static int run_attention(const AttnArgs& a, hipStream_t s) {
    AttnParams p;
    p.Q = (const unsigned char*)a.Q;
    p.K = (const unsigned char*)a.K;
    p.V = (const unsigned char*)a.V;
    p.O = a.O;
    p.ldq_b = (long long)a.ldq * sizeof(T);
    p.ldk_b = (long long)a.ldk * sizeof(T);
    p.ldv_b = (long long)a.ldv * sizeof(T);
    p.ldo = a.ldo;
    p.H = a.H;
    p.Lq = a.Lq;
    p.Lk = a.Lk;
    p.keymask = a.keymask;
    p.kv_mod = a.kv_mod;
    static const int stamps = cn_exp_env("CASSNAT_ATTN_STAMPS") ? 1 : 0;
    p.stamps = stamps;
    p.kv_index = a.kv_index;
    p.klen = a.klen;
    p.kcap = a.kcap;
    p.kcap_stride = a.kcap_stride;
    p.o_blk = a.o_blocked;
    p.q_blk = a.q_blocked;
    p.kv_blk = a.kv_blocked;
    p.q_col = a.q_col;
    p.k_col = a.k_col;
    p.v_col = a.v_col;
    p.q_n = a.q_n;
    p.kv_n = a.kv_n;
    if ((a.q_blocked || a.kv_blocked || a.o_blocked) && (sizeof(T) != 2 || __is_same(T, split_t) || a.rel_pos || (a.q_blocked && a.q_n % 32) ||
                                          (a.kv_blocked && a.kv_n % 32))) {
        cn_set_error("attention: blocked operands exist for the bf16 kernels without relative positions; column counts % 32 == 0");
        return -1;
    }
    p.iv = a.intervals;
    p.iv_stride = a.iv_stride;
    p.causal = a.causal;
    p.scale = a.scale;
    p.rel_pos = a.rel_pos;
    p.rel_u = a.rel_u;
    p.rel_v = a.rel_v;
    p.rel_R = a.rel_R;
    p.ld_pos = a.ld_pos;
    {
        if (a.rel_pos) {
            if (a.rel_R < 0 || a.rel_R > 31 || !a.rel_u || !a.rel_v || a.Lq != a.Lk) {
                cn_set_error("attention: relative positions need self attention and max_relative_len <= 31");
                return -1;
            }
            hipLaunchKernelGGL((attention_kernel<T, 2, false, true>), dim3(cn_ceil_div(a.Lq, 64), a.H, a.B), dim3(128), 0, s, p);
            CN_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    const long long big_grid = (long long)cn_ceil_div(a.Lq, 128) * a.H * a.B;
    if constexpr (sizeof(T) == 2) {
        static const int no_res = cn_exp_env("CASSNAT_ATTN_NO_RES") ? 1 : 0;
        if (a.Lk <= 256 && !no_res) {  // all keys/values of a (batch, head) resident in 64 KB of LDS
            // Query rows per workgroup: every workgroup loads the whole K/V of its (batch, head), so fewer, larger
            // workgroups mean less CU time per launch (what counts when several decode pipelines share the GPU).
            static const int nw = cn_exp_env("CASSNAT_ATTN_NW") ? atoi(cn_exp_env("CASSNAT_ATTN_NW")) : 0;
            const int use = nw ? nw : (a.Lq > 128 ? 8 : (a.Lq > 64 ? 4 : 2));
            if (use >= 8)
                hipLaunchKernelGGL((attention_kernel<T, 8, true>), dim3(cn_ceil_div(a.Lq, 256), a.H, a.B), dim3(512), 0, s, p);
            else if (use >= 4)
                hipLaunchKernelGGL((attention_kernel<T, 4, true>), dim3(cn_ceil_div(a.Lq, 128), a.H, a.B), dim3(256), 0, s, p);
            else
                hipLaunchKernelGGL((attention_kernel<T, 2, true>), dim3(cn_ceil_div(a.Lq, 64), a.H, a.B), dim3(128), 0, s, p);
            CN_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (big_grid >= 1024)
        hipLaunchKernelGGL((attention_kernel<T, 4, false>), dim3(cn_ceil_div(a.Lq, 128), a.H, a.B), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((attention_kernel<T, 2, false>), dim3(cn_ceil_div(a.Lq, 64), a.H, a.B), dim3(128), 0, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attention(int prec, const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.Lq <= 0 || a.Lk <= 0 || a.H <= 0) return 0;
    const size_t es = cn_elem_size(prec);
    if ((a.ldq * es) % 16 || (a.ldk * es) % 16 || (a.ldv * es) % 16 || (a.ldo * es) % 16) {
        cn_set_error("attention: row strides must keep rows 16-byte aligned");
        return -1;
    }
    if (prec == CN_PREC_X3) {
        if (a.ldq % 32 || a.ldk % 32 || a.ldv % 32 || a.ldo % 32) {
            cn_set_error("attention: split-bf16 rows need strides that are multiples of 32 elements");
            return -1;
        }
        return run_attention<split_t>(a, s);
    }
    return prec == CN_PREC_F32 ? run_attention<float>(a, s) : run_attention<bf16>(a, s);
}
