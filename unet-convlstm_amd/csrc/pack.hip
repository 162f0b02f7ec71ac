// Weight panels: f32 reference layouts (OIHW conv, [in,out,2,2] transposed conv) <-> the
// [N][Ktot] panels the implicit-GEMM kernels read, plus bias reordering.  One generic index
// map (uclstm_pack_desc) covers forward panels, flipped/transposed input-gradient panels, the
// gate-interleaved ConvLSTM panel, tap-major ConvTranspose panels and the pre-gathered first
// layer.  Run once per optimiser step per weight; memory-bound, trivially parallel.
#include "common.h"

namespace {

struct Decoded {
    bool valid;
    int64_t off;
};

__device__ __forceinline__ bool decode_n(const uclstm_pack_desc& d, int n, int& n_ent, int& tapn) {
    tapn = 0;
    if (d.n_mode == UCLSTM_NMODE_IDENTITY) {
        n_ent = n;
        return n < d.n_valid;
    } else if (d.n_mode == UCLSTM_NMODE_LSTM) {
        const int hb = n >> 6, gate = (n & 63) >> 4, j = n & 15;
        const int hc = hb * 16 + j;
        n_ent = gate * d.n_valid + hc;
        return hc < d.n_valid;
    } else {
        tapn = n / d.n_cp;
        const int co = n - tapn * d.n_cp;
        n_ent = co;
        return co < d.n_valid;
    }
}

__device__ __forceinline__ Decoded decode(const uclstm_pack_desc& d, int n, int k) {
    Decoded r;
    r.valid = false;
    r.off = 0;
    int n_ent, tapn;
    if (!decode_n(d, n, n_ent, tapn)) return r;
    const int per_tap = d.kseg[0] + d.kseg[1];
    int tap = k / per_tap;
    const int kr = k - tap * per_tap;
    const int s = kr >= d.kseg[0] ? 1 : 0;
    const int c = s ? kr - d.kseg[0] : kr;
    int k_ent;
    if (d.k_mode == UCLSTM_KMODE_IDENTITY) {
        if (c >= d.cvalid[s]) return r;
        k_ent = d.choff[s] + c;
    } else if (d.k_mode == UCLSTM_KMODE_GATES) {
        const int gate = c / d.k_hdp;
        const int hc = c - gate * d.k_hdp;
        if (gate >= 4 || hc >= d.k_hd) return r;
        k_ent = d.choff[s] + gate * d.k_hd + hc;
    } else {
        const int tk = c / d.k_hd;
        if (tk >= d.k_hdp) return r;
        k_ent = c - tk * d.k_hd;
        tap = tk;
    }
    const int ntap_total = (d.k_mode == UCLSTM_KMODE_IM2COL) ? d.k_hdp : d.taps;
    const int tap_eff = d.tap_flip ? ntap_total - 1 - tap : tap;
    r.valid = true;
    r.off = (int64_t)n_ent * d.stride_n + (int64_t)k_ent * d.stride_k + (int64_t)tap_eff * d.stride_tap + (int64_t)tapn * d.stride_ntap;
    return r;
}

__global__ void pack_kernel(const uclstm_pack_desc d, const float* __restrict__ w, bf16* __restrict__ wp) {
    const int64_t total = (int64_t)d.N * d.Ktot;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / d.Ktot);
        const int k = (int)(idx - (int64_t)n * d.Ktot);
        const Decoded r = decode(d, n, k);
        wp[idx] = f32_to_bf16(r.valid ? w[r.off] : 0.f);
    }
}

__global__ void unpack_kernel(const uclstm_pack_desc d, const float* __restrict__ dwp, int nslab, int64_t slab, float* __restrict__ grad,
                              int accumulate) {
    const int64_t total = (int64_t)d.N * d.Ktot;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / d.Ktot);
        const int k = (int)(idx - (int64_t)n * d.Ktot);
        const Decoded r = decode(d, n, k);
        if (r.valid) {
            float v = dwp[idx];
            for (int sl = 1; sl < nslab; ++sl) v += dwp[sl * slab + idx];      // partial panels of the pixel ranges
            grad[r.off] = (accumulate ? grad[r.off] : 0.f) + v;
        }
    }
}

__global__ void pack_bias_kernel(const uclstm_pack_desc d, const float* __restrict__ b, float* __restrict__ bp) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= d.N) return;
    int n_ent, tapn;
    const bool ok = decode_n(d, n, n_ent, tapn);
    bp[n] = ok ? b[n_ent] : 0.f;
}

bool desc_ok(const uclstm_pack_desc* d) {
    if (!d || d->N <= 0 || d->Ktot <= 0 || d->taps <= 0 || d->nsrc < 1 || d->nsrc > 2) return false;
    if (d->Ktot != d->taps * (d->kseg[0] + d->kseg[1])) return false;
    if (d->n_mode < 0 || d->n_mode > 2 || d->k_mode < 0 || d->k_mode > 2) return false;
    if (d->n_mode == UCLSTM_NMODE_TAPMAJOR && d->n_cp <= 0) return false;
    if (d->k_mode != UCLSTM_KMODE_IDENTITY && (d->k_hd <= 0 || d->k_hdp <= 0)) return false;
    return true;
}

int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int32_t uclstm_pack_weights(const uclstm_pack_desc* d, const float* w, void* wp, void* stream) {
    if (!desc_ok(d) || !w || !wp) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(pack_kernel, dim3(grid_for((int64_t)d->N * d->Ktot)), dim3(256), 0, (hipStream_t)stream, *d, w, (bf16*)wp);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_unpack_wgrad(const uclstm_pack_desc* d, const float* dwp, int32_t nslab, int64_t slab, float* grad,
                                       int32_t accumulate, void* stream) {
    if (!desc_ok(d) || !dwp || !grad || nslab < 1 || (nslab > 1 && slab < (int64_t)d->N * d->Ktot)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(unpack_kernel, dim3(grid_for((int64_t)d->N * d->Ktot)), dim3(256), 0, (hipStream_t)stream, *d, dwp, nslab, slab, grad, accumulate);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_pack_bias(const uclstm_pack_desc* d, const float* b, float* bp, void* stream) {
    if (!desc_ok(d) || !b || !bp) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(pack_bias_kernel, dim3((d->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, b, bp);
    return UCLSTM_OK;
}
