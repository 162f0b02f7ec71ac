// Shared device helpers for the gfx950 kernels of libuclstm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit storage / MFMA operand type of activations and weight panels.  The dtype-dependent sources (igemm_fwd,
// igemm_wgrad, pointwise, pack) are compiled TWICE into the same library: once for bfloat16 (the default entry points) and
// once with -DUCLSTM_ACT_F16 for IEEE binary16, whose entry points carry the suffix _f16 (include/uclstm.h, "fp16 twins":
// BASELINE.json configs[3] asks for fp16 MFMA).  Accumulators, cell state, statistics and all gradients of parameters stay
// f32 in both.  Entry points that do not touch 16-bit data exist once (guarded by #ifndef UCLSTM_ACT_F16 in their files).
#if defined(UCLSTM_ACT_F16)
#define uclstm_igemm_fwd uclstm_igemm_fwd_f16
#define uclstm_igemm_fwd_group uclstm_igemm_fwd_group_f16
#define uclstm_igemm_wgrad uclstm_igemm_wgrad_f16
#define uclstm_pack_weights uclstm_pack_weights_f16
#define uclstm_pack_weights_batched uclstm_pack_weights_batched_f16
#define uclstm_splitk_finish uclstm_splitk_finish_f16
#define uclstm_bn_apply_relu uclstm_bn_apply_relu_f16
#define uclstm_bn_bwd_reduce uclstm_bn_bwd_reduce_f16
#define uclstm_bn_bwd_apply uclstm_bn_bwd_apply_f16
#define uclstm_bn_head_fwd uclstm_bn_head_fwd_f16
#define uclstm_bn_apply_relu_pool uclstm_bn_apply_relu_pool_f16
#define uclstm_bn_pool_bwd_reduce uclstm_bn_pool_bwd_reduce_f16
#define uclstm_bn_pool_bwd_apply uclstm_bn_pool_bwd_apply_f16
#define uclstm_bn_head_bwd_reduce uclstm_bn_head_bwd_reduce_f16
#define uclstm_bn_head_bwd_apply uclstm_bn_head_bwd_apply_f16
#define uclstm_maxpool2_fwd uclstm_maxpool2_fwd_f16
#define uclstm_maxpool2_bwd uclstm_maxpool2_bwd_f16
#define uclstm_lstm_bwd_pointwise uclstm_lstm_bwd_pointwise_f16
#define uclstm_lstm_fwd_pointwise uclstm_lstm_fwd_pointwise_f16
#define uclstm_lstm_fwd_pointwise_group uclstm_lstm_fwd_pointwise_group_f16
#define uclstm_nchw_to_nhwc uclstm_nchw_to_nhwc_f16
#define uclstm_nhwc_to_nchw uclstm_nhwc_to_nchw_f16
#define uclstm_nchw_grad_to_nhwc uclstm_nchw_grad_to_nhwc_f16
#define uclstm_im2col3x3_first uclstm_im2col3x3_first_f16
#define uclstm_outconv_fwd uclstm_outconv_fwd_f16
#define uclstm_outconv_bwd uclstm_outconv_bwd_f16
#define uclstm_colsum uclstm_colsum_f16
#define uclstm_attention_fwd uclstm_attention_fwd_f16
#define uclstm_attention_bwd uclstm_attention_bwd_f16
#endif
#include "../../include/uclstm.h"

#if defined(UCLSTM_ACT_F16)
typedef _Float16 act16;
typedef __attribute__((ext_vector_type(8))) _Float16 act16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 act16x4;
#define UCLSTM_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __bf16 act16;
typedef __attribute__((ext_vector_type(8))) __bf16 act16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 act16x4;
#define UCLSTM_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define UCLSTM_WAVE 64

// Launch and report only THIS launch's error: the per-thread "last error" may hold a stale, already
// handled code from another library's HIP call (observed: a launch right after torch's own copies).
inline int g_uclstm_last_hip_error = 0;      // hipError_t of the most recent failed launch (diagnostics); ONE variable for both compile passes
#define UCLSTM_LAUNCH(...)                                      \
    do {                                                        \
        (void)hipGetLastError();                                \
        hipLaunchKernelGGL(__VA_ARGS__);                        \
        const hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) {                                \
            g_uclstm_last_hip_error = (int)e__;                 \
            return UCLSTM_E_LAUNCH;                             \
        }                                                       \
    } while (0)

__device__ __forceinline__ float act_to_f32(act16 v) { return (float)v; }
__device__ __forceinline__ act16 f32_to_act(float v) { return (act16)v; }   // v_cvt_pk_bf16_f32 / v_cvt_f16_f32: RNE, NaN preserved

// 8 act16 <-> 8 floats through one 16-byte register quad
union Pack16 {
    uint4 u;
    act16x8 v;
    act16 e[8];
};
union Pack8 {
    uint2 u;
    act16x4 v;
    act16 e[4];
};

__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// 1 - 2/(e^{2x}+1): saturates correctly to +-1 for large |x|
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int32_t round_up32(int32_t a, int32_t b) { return (a + b - 1) / b * b; }

// Bijective XCD-aware remap of a 1-D grid (guide T1): blocks that share an XCD (b % 8) get a
// contiguous run of logical ids, so neighbouring tiles (which share a weight panel) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Grouped ("super-tile") order of a 2-D tile grid: groups of GROUP consecutive a-tiles; inside a group the a index runs
// fastest, then b.  The ~64 blocks resident on one XCD (consecutive ids after xcd_remap) then form a GROUP x 64/GROUP
// rectangle: every operand slab is shared by several resident blocks instead of a whole row or column of the grid
// streaming through the XCD's 4 MiB L2.  Bijective for any (na, nb).
__device__ __forceinline__ void grouped_tile(int id, int na, int nb, int group, int& a, int& b) {
    const int per_group = group * nb;
    const int g = id / per_group;
    const int first = g * group;
    const int gsz = min(na - first, group);
    const int r = id - g * per_group;
    b = r / gsz;
    a = first + (r - b * gsz);
}

// Division of a 31-bit dividend by a runtime constant: q = umulhi(m, magic) >> shift (Granlund-Montgomery
// round-up form; exact for every m < 2^31).  magic == 0 encodes d == 1.
struct FastDiv {
    uint32_t magic;
    uint32_t shift;
    uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) {
        f.magic = 0;
        f.shift = 0;
        return f;
    }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned s = 31 + l;
    f.magic = (uint32_t)(((1ull << s) + d - 1) / d);
    f.shift = s - 32;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t m, const FastDiv& f) {
    return f.magic ? (__umulhi(m, f.magic) >> f.shift) : m;
}
