// Loss (main.py:28-72) and optimiser (main.py:106-108) kernels: one pass each over f32 planes /
// flat parameter buffers, wave-shuffle + LDS block reduction, one f64 atomic per block.
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// reduce K doubles per thread across the block; thread 0 gets the totals
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* red /* [4][K] */) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = red[k] + red[K + k] + red[2 * K + k] + red[3 * K + k];
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// sums: [0] sum(ad*w*m) [1] sum(w*m) [2] sum(gd*mc) [3] sum(mc)
__global__ void loss_fwd_kernel(const float* __restrict__ yp, const float* __restrict__ y, const float* __restrict__ mask,
                                double* __restrict__ sums, int64_t total, FastDiv dHW, FastDiv dW, int H, int W) {
    __shared__ double red[16];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        const uint32_t pl = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - pl * HW;
        const int i = (int)fdiv(pix, dW);
        const int j = (int)pix - i * W;
        const float a = yp[idx], b = y[idx];
        const float m = mask ? mask[idx] : 1.f;
        const float ab = fabsf(b);
        const float w = 1.f + 4.f * ab * ab * ab;               // main.py:38
        acc[0] += (double)(fabsf(a - b) * w * m);
        acc[1] += (double)(w * m);
        if (i < H - 1 && j < W - 1) {                            // main.py:57-62 crop
            const float dxp = yp[idx + 1] - a, dyp = yp[idx + W] - a;
            const float dxg = y[idx + 1] - b, dyg = y[idx + W] - b;
            acc[2] += (double)((fabsf(dxp - dxg) + fabsf(dyp - dyg)) * m);
            acc[3] += (double)m;
        }
    }
    block_sum<4>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(sums + k, acc[k]);
}

__global__ void loss_bwd_kernel(const float* __restrict__ yp, const float* __restrict__ y, const float* __restrict__ mask,
                                const float* __restrict__ coefs, float* __restrict__ grad, int64_t total, FastDiv dHW, FastDiv dW,
                                int H, int W) {
    const int HW = dHW.d;
    const float c1 = coefs[0], c2 = coefs[1];
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        const uint32_t pl = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - pl * HW;
        const int i = (int)fdiv(pix, dW);
        const int j = (int)pix - i * W;
        const float a = yp[idx], b = y[idx];
        const float m = mask ? mask[idx] : 1.f;
        const float ab = fabsf(b);
        float g = c1 * sgn(a - b) * (1.f + 4.f * ab * ab * ab) * m;
        float gg = 0.f;
        // cell (i,j) itself: -sx(i,j) - sy(i,j)
        if (i < H - 1 && j < W - 1) {
            const float sx = sgn((yp[idx + 1] - a) - (y[idx + 1] - b));
            const float sy = sgn((yp[idx + W] - a) - (y[idx + W] - b));
            gg -= (sx + sy) * m;
        }
        // cell (i,j-1): +sx(i,j-1)
        if (j >= 1 && i < H - 1) {
            const float ml = mask ? mask[idx - 1] : 1.f;
            gg += sgn((a - yp[idx - 1]) - (b - y[idx - 1])) * ml;
        }
        // cell (i-1,j): +sy(i-1,j)
        if (i >= 1 && j < W - 1) {
            const float mu = mask ? mask[idx - W] : 1.f;
            gg += sgn((a - yp[idx - W]) - (b - y[idx - W])) * mu;
        }
        grad[idx] = g + c2 * gg;
    }
}

__global__ void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ double red[4];
    double acc[1] = {0.0};
    const int64_t n4 = n >> 2;
    const float4* g4 = (const float4*)g;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const float4 v = g4[i];
        acc[0] += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc[0] += (double)(v * v);
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, acc[0]);
}

__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                             int64_t n, const double* __restrict__ sumsq, float max_norm, float lr, float b1, float b2, float eps,
                             float wd, float inv_bc1, float inv_sqrt_bc2) {
    float coef = 1.f;
    if (sumsq) {
        const float total = (float)sqrt(*sumsq);
        coef = fminf(max_norm / (total + 1e-6f), 1.f);          // torch.nn.utils.clip_grad_norm_ (main.py:106)
    }
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gi = g[i] * coef;
        float w = p[i] * (1.f - lr * wd);                        // decoupled decay (AdamW, main.py:275)
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        w -= lr * inv_bc1 * mi / denom;
        p[i] = w;
        m[i] = mi;
        v[i] = vi;
    }
}

// The same update with every hyper-parameter and the step count read from DEVICE memory, so that the launch can sit in a
// captured HIP graph and still follow a learning-rate schedule: hyper = {lr, beta1, beta2, eps, weight_decay, max_norm
// (<= 0: no clipping), steps done so far (float, exact up to 2^24), reserved}.  adamw_advance_kernel bumps the count afterwards.
__global__ void adamw_dev_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                                 int64_t n, const double* __restrict__ sumsq, const float* __restrict__ hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], max_norm = hyper[5];
    const double step = (double)hyper[6] + 1.0;
    const float inv_bc1 = (float)(1.0 / (1.0 - pow((double)b1, step)));          // as the host computes them for adamw_kernel
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, step)));
    float coef = 1.f;
    if (sumsq && max_norm > 0.f) {
        const float total = (float)sqrt(*sumsq);
        coef = fminf(max_norm / (total + 1e-6f), 1.f);
    }
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gi = g[i] * coef;
        float w = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        w -= lr * inv_bc1 * mi / denom;
        p[i] = w;
        m[i] = mi;
        v[i] = vi;
    }
}
__global__ void adamw_advance_kernel(float* hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) hyper[6] += 1.f;
}

// fp16 training: the gradient buffer holds scale x the true gradient (loss scaling keeps fp16 activation gradients out of the
// subnormal range).  state = {scale, growth tracker, successful steps}.  Same update as adamw_kernel on g / scale; nothing is
// touched when the scaled sum of squares is not finite (an overflowed step is skipped, the scale backs off in
// loss_scale_update_kernel); Adam's bias-correction step is the device-side count of successful steps.
__global__ void adamw_scaled_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                                    int64_t n, const double* __restrict__ sumsq, float max_norm, float lr, float b1, float b2, float eps,
                                    float wd, const float* __restrict__ state) {
    const double ss = *sumsq;
    if (!(ss == ss) || ss > 1.0e300 || isinf(ss)) return;
    const float inv_scale = 1.f / state[0];
    const float step = state[2] + 1.f;
    const float inv_bc1 = 1.f / (1.f - exp2f(step * log2f(b1)));
    const float inv_sqrt_bc2 = rsqrtf(1.f - exp2f(step * log2f(b2)));
    float coef = inv_scale;
    if (max_norm > 0.f) {
        const float total = (float)sqrt(ss) * inv_scale;
        coef *= fminf(max_norm / (total + 1e-6f), 1.f);
    }
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gi = g[i] * coef;
        float w = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        w -= lr * inv_bc1 * mi / denom;
        p[i] = w;
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void loss_scale_update_kernel(float* __restrict__ state, const double* __restrict__ sumsq, float growth, float backoff,
                                         int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ss = *sumsq;
    if (!(ss == ss) || isinf(ss)) {
        state[0] = fmaxf(state[0] * backoff, 1.f);
        state[1] = 0.f;
    } else {
        state[2] += 1.f;
        state[1] += 1.f;
        if (state[1] >= (float)interval) {
            state[0] = fminf(state[0] * growth, 16777216.f);
            state[1] = 0.f;
        }
    }
}

// NPZSequenceDataset.__getitem__ for a batch (train/unet.py:273-304): mask from RAW channel 0 (> 1.1) before scaling,
// x / norm_const, y clipped -> asinh(y / scale) -> [-1, 1]
__global__ void dataset_transform_kernel(const float* __restrict__ xr, const float* __restrict__ yr, float* __restrict__ x,
                                         float* __restrict__ y, float* __restrict__ mask, int64_t total, FastDiv dHW, int C, float inv_norm,
                                         float min_vel, float max_vel, int clip, float inv_yscale, float tmin, float inv_trange) {
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        const uint32_t f = fdiv((uint32_t)idx, dHW);               // frame
        const uint32_t pix = (uint32_t)idx - f * HW;
        const float* xf = xr + (int64_t)f * C * HW + pix;
        mask[idx] = xf[0] > 1.1f ? 1.f : 0.f;
        for (int c = 0; c < C; ++c) x[(int64_t)f * C * HW + (int64_t)c * HW + pix] = xf[(int64_t)c * HW] * inv_norm;
        float v = yr[idx];
        if (clip) v = fminf(fmaxf(v, min_vel), max_vel);
        y[idx] = 2.f * (asinhf(v * inv_yscale) - tmin) * inv_trange - 1.f;
    }
}

// Epoch metrics of main.py:114-142 as running sums: de-normalise (train/unet.py:316-319) prediction and target,
// d = pred - target, sums += (sum |d| m, sum d^2 m, sum d m, sum m)
__global__ void metric_sums_kernel(const float* __restrict__ yp, const float* __restrict__ y, const float* __restrict__ mask,
                                   double* __restrict__ sums, int64_t n, float yscale, float tmin, float trange) {
    __shared__ double red[16];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float a = sinhf((yp[i] + 1.f) * 0.5f * trange + tmin) * yscale;
        const float b = sinhf((y[i] + 1.f) * 0.5f * trange + tmin) * yscale;
        const float m = mask ? (mask[i] != 0.f ? 1.f : 0.f) : 1.f;
        const double d = (double)(a - b);
        acc[0] += fabs(d) * m;
        acc[1] += d * d * m;
        acc[2] += d * m;
        acc[3] += m;
    }
    block_sum<4>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(sums + k, acc[k]);
}

int grid_for(int64_t items, int cap) {
    int64_t b = (items + NT - 1) / NT;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int32_t uclstm_loss_fwd(const float* y_pred, const float* y, const float* mask, double* sums, int64_t planes, int32_t H,
                                   int32_t W, void* stream) {
    if (!y_pred || !y || !sums || planes <= 0 || H <= 0 || W <= 0) return UCLSTM_E_BADARG;
    const int64_t total = planes * H * W;
    if (total >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(loss_fwd_kernel, dim3(grid_for(total, 1024)), dim3(NT), 0, (hipStream_t)stream, y_pred, y, mask, sums, total,
                       make_fastdiv(H * W), make_fastdiv(W), H, W);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_loss_bwd(const float* y_pred, const float* y, const float* mask, const float* coefs, float* grad,
                                   int64_t planes, int32_t H, int32_t W, void* stream) {
    if (!y_pred || !y || !grad || !coefs || planes <= 0 || H <= 0 || W <= 0) return UCLSTM_E_BADARG;
    const int64_t total = planes * H * W;
    if (total >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(loss_bwd_kernel, dim3(grid_for(total, 2048)), dim3(NT), 0, (hipStream_t)stream, y_pred, y, mask, coefs, grad,
                       total, make_fastdiv(H * W), make_fastdiv(W), H, W);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_sumsq(const float* g, int64_t n, double* out, void* stream) {
    if (!g || !out || n <= 0 || ((uintptr_t)g % 16)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(sumsq_kernel, dim3(grid_for((n + 3) / 4, 1024)), dim3(NT), 0, (hipStream_t)stream, g, n, out);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_adamw_step(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float max_norm,
                                     float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    if (!p || !m || !v || !g || n <= 0 || step < 1) return UCLSTM_E_BADARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    UCLSTM_LAUNCH(adamw_kernel, dim3(grid_for(n, 2048)), dim3(NT), 0, (hipStream_t)stream, p, m, v, g, n, sumsq, max_norm, lr, beta1,
                       beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_adamw_step_dev(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float* hyper,
                                         void* stream) {
    if (!p || !m || !v || !g || n <= 0 || !hyper || ((uintptr_t)hyper % 16)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(adamw_dev_kernel, dim3(grid_for(n, 2048)), dim3(NT), 0, (hipStream_t)stream, p, m, v, g, n, sumsq, (const float*)hyper);
    UCLSTM_LAUNCH(adamw_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_adamw_step_scaled(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float max_norm,
                                            float lr, float beta1, float beta2, float eps, float weight_decay, const float* scale_state,
                                            void* stream) {
    if (!p || !m || !v || !g || n <= 0 || !sumsq || !scale_state) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(adamw_scaled_kernel, dim3(grid_for(n, 2048)), dim3(NT), 0, (hipStream_t)stream, p, m, v, g, n, sumsq, max_norm, lr, beta1,
                  beta2, eps, weight_decay, scale_state);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_loss_scale_update(float* scale_state, const double* sumsq, float growth, float backoff, int32_t interval,
                                            void* stream) {
    if (!scale_state || !sumsq || growth < 1.f || backoff <= 0.f || backoff > 1.f || interval < 1) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale_state, sumsq, growth, backoff, interval);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_dataset_transform(const float* x_raw, const float* y_raw, float* x, float* y, float* mask, int64_t n_frames,
                                            int32_t C, int32_t HW, float norm_const, float min_vel, float max_vel, int32_t clip,
                                            float y_scale, float trans_min, float trans_max, void* stream) {
    if (!x_raw || !y_raw || !x || !y || !mask || n_frames <= 0 || C <= 0 || HW <= 0 || norm_const == 0.f || y_scale == 0.f ||
        trans_max == trans_min)
        return UCLSTM_E_BADARG;
    const int64_t total = n_frames * HW;
    if (total >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(dataset_transform_kernel, dim3(grid_for(total, 2048)), dim3(NT), 0, (hipStream_t)stream, x_raw, y_raw, x, y, mask, total,
                  make_fastdiv(HW), C, 1.0f / norm_const, min_vel, max_vel, clip, 1.0f / y_scale, trans_min,
                  1.0f / (trans_max - trans_min));
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_metric_sums(const float* y_pred, const float* y, const float* mask, double* sums, int64_t n, float y_scale,
                                      float trans_min, float trans_max, void* stream) {
    if (!y_pred || !y || !sums || n <= 0) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(metric_sums_kernel, dim3(grid_for(n, 1024)), dim3(NT), 0, (hipStream_t)stream, y_pred, y, mask, sums, n, y_scale,
                  trans_min, trans_max - trans_min);
    return UCLSTM_OK;
}

namespace {
__global__ void stream_spin_kernel(long ticks) {
    const long t0 = wall_clock64();                      // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" int32_t uclstm_stream_spin(int32_t microseconds, void* stream) {
    if (microseconds <= 0 || microseconds > 100000) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)microseconds * 100);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_abi_version(void) { return UCLSTM_ABI_VERSION; }
extern "C" const char* uclstm_build_arch(void) { return "gfx950"; }
extern "C" const char* uclstm_last_error_string(void) { return hipGetErrorString((hipError_t)g_uclstm_last_hip_error); }
