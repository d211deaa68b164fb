// Shared device/host definitions for libcassnat_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Experiment switches.  The product library reads NO environment variable: every kernel-selection / stamp / repeat switch the
// measurement tools use (tools/*.py, tools/scripts/ab_bench.sh) exists only in a -DCASSNAT_EXPERIMENTS build of the library
// (cassnat_asr_public_amd.build.build(extra_flags=["-DCASSNAT_EXPERIMENTS"], lib=...), loaded through CASSNAT_HIP_LIB - the one
// loader-level variable, read by hip.py).  In the product build the helper is a constant and the switched-off paths fold away.
#ifdef CASSNAT_EXPERIMENTS
static inline const char* cn_exp_env(const char* name) { return getenv(name); }
#else
static inline const char* cn_exp_env(const char*) { return nullptr; }
#endif

// The 16-bit MFMA operand of the fast engine.  These sources are built into TWO libraries: libcassnat_hip.so, where it is bfloat16
// (the engines "bf16" and "fp8", beside the split-bf16 and fp32 engines), and libcassnat_hip_f16.so (-DCN_OP16_F16), where it is
// IEEE half precision - 11 significant bits instead of 8 on the same matrix pipe at the same rate (v_mfma_f32_32x32x16_f16), the
// engine "fp16": the operand roundings are 8 x smaller (CTC log-posteriors within 1e-3 of the fp32 reference on the benchmark model,
// where bfloat16 operands give 5e-3), the price is the range (|v| <= 65504: LayerNorm outputs, ReLU activations, softmax
// probabilities, Q / K / V and weights of a model of this family sit many orders below it; the residual stream, every
// accumulator, LayerNorm and softmax stay fp32 in both).  Everything below - layouts, LDS images, fragment maps, schedules - is
// the same for both: the kernels name the type `bf16` throughout; in the -DCN_OP16_F16 build that name IS the half type.
#ifdef CN_OP16_F16
typedef _Float16 op16;
#define CN_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define CN_MFMA16_ASM "v_mfma_f32_32x32x16_f16 "
#define CN_OP16_NAME "fp16"
#else
typedef __bf16 op16;
#define CN_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define CN_MFMA16_ASM "v_mfma_f32_32x32x16_bf16 "
#define CN_OP16_NAME "bf16"
#endif
typedef op16 bf16;
typedef op16 bf16x4 __attribute__((ext_vector_type(4)));
typedef op16 bf16x8 __attribute__((ext_vector_type(8)));
// host side: fp32 -> the operand's 16 bits, round to nearest even (weight packing)
static inline uint16_t cn_host_op16(float f) {
#ifdef CN_OP16_F16
    const _Float16 h = (_Float16)f;  // (IEEE conversion: overflow gives infinity - a weight of a usable model is nowhere near)
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
#endif
}
static inline float cn_host_op16_value(uint16_t b) {
#ifdef CN_OP16_F16
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
#else
    const uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef long i64x2 __attribute__((ext_vector_type(2)));
// OCP e4m3fn byte (gfx950's fp8 MFMA format; not MI300's fnuz): max 448, no infinities
struct fp8_t {
    unsigned char bits;
};
#define CN_FP8_MAX 448.0f
// Split-bf16 element ("bf16x3" precision): a value v is kept as hi = bf16(v) and lo = bf16(v - hi); hi + lo carries ~17
// significant bits, and a product is three bf16 MFMAs (hi.hi + hi.lo + lo.hi, fp32 accumulation; the dropped lo.lo term is
// 2^-18 relative).  Storage of a row of K elements (K % 32 == 0): groups of 32 elements = 128 bytes, [32 x bf16 hi][32 x bf16 lo].
// sizeof == 4 on purpose: row strides, column offsets that are multiples of 32 elements (heads, Q|K|V thirds) and buffer
// sizes are those of the fp32 engine, and a 128-byte LDS slab row holds exactly one group.
struct split_t {
    unsigned int bits;
};
// byte offset of element c's hi half inside its row (the lo half sits 64 bytes further)
__host__ __device__ static inline size_t cn_split_off(size_t c) { return (c >> 5) * 128 + (c & 31) * 2; }

// Blocked bf16 matrix [M][N] (N % 32 == 0, the buffer holds ceil(M / 32) * 32 rows): 32-row x 32-column tiles of 2 KiB, each
// [16-column half gp][lane = 32 * h + (row & 31)][8 bf16] with h = bit 3 of the column - the order in which the row-chain kernel's
// lanes hold a projection tile after the half-wave exchange (chain.hip, S5), so that each of its store instructions writes 1 KiB
// contiguous instead of thirty-two 32-byte row segments.  Byte offset of the 16-byte chunk that starts at column c (c % 8 == 0):
__host__ __device__ static inline long long cn_blk16_off(long long m, int c, int N) {
    return ((((m >> 5) * (long long)(N >> 5) + (c >> 5)) << 1) + ((c >> 4) & 1)) * 1024 + ((((c >> 3) & 1) << 5) + (int)(m & 31)) * 16;
}

#define CN_WAVE 64
#define CN_NEG_FILL (-3.4028234663852886e38f) /* float32 min: the reference's masked_fill value */

// ----------------------------------------------------------------------------------------------
// 16-byte fragment of the GEMM K dimension as one wave lane sees it.
//   bf16: 8 consecutive k  -> one v_mfma_f32_32x32x16_bf16 (k = 8*(lane>>5) + j)
//   f32 : 4 consecutive k  -> four v_mfma_f32_32x32x2_f32  (step e uses k = 4*(lane>>5) + e)
//   fp8 : 16 consecutive k -> two v_mfma_f32_32x32x16_fp8_fp8 (low / high 8 bytes of the fragment)
// Both operands of a product use the same lane->k map, so the sum over k is complete; only the
// order of the fp32 additions differs from a sequential loop.
// ----------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16> {
    typedef bf16x8 type;
    static constexpr int ELEMS = 8;
};
template <> struct Frag<float> {
    typedef f32x4 type;
    static constexpr int ELEMS = 4;
};

template <> struct Frag<fp8_t> {
    typedef i64x2 type;
    static constexpr int ELEMS = 16;
};
struct split_frag {
    bf16x8 hi, lo;
};
template <> struct Frag<split_t> {
    typedef split_frag type;
    static constexpr int ELEMS = 8;
};
__device__ __forceinline__ f32x16 mfma_frag(const split_frag& a, const split_frag& b, f32x16 c) {
    c = CN_MFMA16(a.lo, b.hi, c, 0, 0, 0);  // (small terms first)
    c = CN_MFMA16(a.hi, b.lo, c, 0, 0, 0);
    c = CN_MFMA16(a.hi, b.hi, c, 0, 0, 0);
    return c;
}
// v -> (hi, lo); four consecutive elements -> 8 bytes of hi and 8 bytes of lo
__device__ __forceinline__ void cn_split4(const float (&v)[4], bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (bf16)v[j];
        lo[j] = (bf16)(v[j] - (float)hi[j]);
    }
}

__device__ __forceinline__ f32x16 mfma_frag(i64x2 a, i64x2 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[1], b[1], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x16 mfma_frag(bf16x8 a, bf16x8 b, f32x16 c) {
    return CN_MFMA16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_frag(f32x4 a, f32x4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

// Row of a 32x32 accumulator register: C/D layout of every 32x32 MFMA on gfx950.
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
// float -> e4m3fn, round to nearest even, saturating at +-448 (the value is expected to carry its tensor's scale already)
__device__ __forceinline__ unsigned char cn_f32_to_fp8(float v) {
    v = fminf(fmaxf(v, -CN_FP8_MAX), CN_FP8_MAX);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}
template <> __device__ __forceinline__ fp8_t from_f32<fp8_t>(float v) { return fp8_t{cn_f32_to_fp8(v)}; }
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
// element (row, c) of a matrix of `ld` columns in any element type (split-bf16: ld % 32 == 0; hi + lo halves)
template <typename T> __device__ __forceinline__ float cn_ld_elem(const void* base, long long row, int ld, int c) {
    if constexpr (__is_same(T, split_t)) {
        const unsigned char* e = reinterpret_cast<const unsigned char*>(base) + row * (long long)ld * 4 + cn_split_off((size_t)c);
        return (float)*reinterpret_cast<const bf16*>(e) + (float)*reinterpret_cast<const bf16*>(e + 64);
    } else {
        return to_f32(reinterpret_cast<const T*>(base)[row * ld + c]);
    }
}
template <typename T> __device__ __forceinline__ void cn_st_elem(void* base, long long row, int ld, int c, float v) {
    if constexpr (__is_same(T, split_t)) {
        unsigned char* e = reinterpret_cast<unsigned char*>(base) + row * (long long)ld * 4 + cn_split_off((size_t)c);
        const bf16 hi = (bf16)v;
        *reinterpret_cast<bf16*>(e) = hi;
        *reinterpret_cast<bf16*>(e + 64) = (bf16)(v - (float)hi);
    } else {
        reinterpret_cast<T*>(base)[row * ld + c] = from_f32<T>(v);
    }
}

// 16-byte global load / LDS store helpers on raw bytes.
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

template <typename T> __device__ __forceinline__ typename Frag<T>::type as_frag(uint4 v);
template <> __device__ __forceinline__ bf16x8 as_frag<bf16>(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
template <> __device__ __forceinline__ f32x4 as_frag<float>(uint4 v) { return __builtin_bit_cast(f32x4, v); }
template <> __device__ __forceinline__ i64x2 as_frag<fp8_t>(uint4 v) { return __builtin_bit_cast(i64x2, v); }

// Wave-wide reductions on the DPP cross-lane paths (no LDS round trip per step as with ds_bpermute: a LayerNorm row costs
// two of these back to back).  Steps: lane ^ 1 and lane ^ 2 (quad permutes), the other quad of the 8 (row_half_mirror: the
// quads are uniform by then), the other 8 of the row of 16 (row_mirror), lane 15 of the previous row into rows 1 and 3
// (row_bcast15), lane 31 into rows 2 and 3 (row_bcast31): lane 63 holds the result, read back as a wave-uniform value.
// All 64 lanes must be active.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float cn_dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += cn_dpp<0xB1, 0xf, true>(0.f, v);
    v += cn_dpp<0x4E, 0xf, true>(0.f, v);
    v += cn_dpp<0x141, 0xf, true>(0.f, v);
    v += cn_dpp<0x140, 0xf, true>(0.f, v);
    v += cn_dpp<0x142, 0xa, false>(0.f, v);
    v += cn_dpp<0x143, 0xc, false>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Combine a value with its partner in the other half-wave (lane ^ 32) on the VALU: v_permlane32_swap of two copies leaves
// {lo, lo} in one and {hi, hi} in the other, so every lane holds both halves' values - no LDS round trip as with ds_bpermute.
// (Written out: given the same value twice, hipcc folds the builtin's two results into one.  s_nop 1 = the two wait states
// between a VALU write of an operand and the swap.)  All 64 lanes must be active.
__device__ __forceinline__ void xhalf_pair(float v, float& lo, float& hi) {
    lo = v;
    hi = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ float xhalf_max(float v) {
    float lo, hi;
    xhalf_pair(v, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float xhalf_sum(float v) {
    float lo, hi;
    xhalf_pair(v, lo, hi);
    return lo + hi;
}

// N independent sums, step by step side by side: the DPP steps of one chain are each other's hazard distance for the next
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += cn_dpp<0xB1, 0xf, true>(0.f, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += cn_dpp<0x4E, 0xf, true>(0.f, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += cn_dpp<0x141, 0xf, true>(0.f, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += cn_dpp<0x140, 0xf, true>(0.f, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += cn_dpp<0x142, 0xa, false>(0.f, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += cn_dpp<0x143, 0xc, false>(0.f, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, cn_dpp<0xB1, 0xf, true>(v, v));
    v = fmaxf(v, cn_dpp<0x4E, 0xf, true>(v, v));
    v = fmaxf(v, cn_dpp<0x141, 0xf, true>(v, v));
    v = fmaxf(v, cn_dpp<0x140, 0xf, true>(v, v));
    v = fmaxf(v, cn_dpp<0x142, 0xa, false>(v, v));
    v = fmaxf(v, cn_dpp<0x143, 0xc, false>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ----------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------
#include <string>
void cn_set_error(const std::string& msg);
#define CN_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cn_set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                         std::to_string(__LINE__) + ")");                                         \
            return -2;                                                                            \
        }                                                                                         \
    } while (0)

#ifndef CN_TRY
#define CN_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != 0) return _rc;   \
    } while (0)
#endif

static inline int cn_ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and the launchers run on several host threads (one decode
// pipeline each): a per-kernel atomic bit mask of the devices already done.  Setting the attribute twice is harmless, so two
// threads racing on the first launch both set it and both publish the bit.
#include <atomic>
struct CnAttrOnce {
    std::atomic<unsigned long long> done{0};
    bool need(int* dev_out) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
        *dev_out = dev;
        return ((done.load(std::memory_order_acquire) >> dev) & 1ull) == 0;
    }
    void mark(int dev) { done.fetch_or(1ull << dev, std::memory_order_release); }
};
