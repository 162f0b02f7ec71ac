// HBM-bound kernels around the MFMA GEMMs: BatchNorm(+ReLU) finalize/apply/backward, MaxPool2d(2),
// the ConvLSTM backward point-wise part, layout conversion at the module boundary, OutConv 1x1 and
// column sums.  All move 16 bytes (8 act16 channels) per lane per access on NHWC tensors; reductions
// over pixels keep a fixed channel chunk per thread, reduce across the block in LDS and finish with
// one f32 atomic per (block, channel).
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
    Pack16 p;
    p.u = u;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = act_to_f32(p.e[i]);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    Pack16 p;
#pragma unroll
    for (int i = 0; i < 8; ++i) p.e[i] = f32_to_act(f[i]);
    return p.u;
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void store8f(float* p, const float (&f)[8]) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

int ew_grid(int64_t items) {
    int64_t b = (items + NT - 1) / NT;
    if (b > 256 * 8) b = 256 * 8;
    return (int)(b < 1 ? 1 : b);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm finalize
// ---------------------------------------------------------------------------------------------
// Pass 1: one block per (group, 64-channel slab), 16 tile lanes x 64 channels: sum the conv epilogue's per-tile partials in
// f64 and leave (mean, biased variance) in the group's tile-0 slot of the SAME buffer (each (group, channel) column is read
// only by its own block, and written after the block's barrier).
__global__ __launch_bounds__(1024) void bn_reduce_kernel(float* __restrict__ stats, int tpg, int Cp, double inv_cnt) {
    __shared__ double r1[16][64], r2[16][64];
    const int g = blockIdx.x;
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < Cp) {
        for (int t = lane; t < tpg; t += 16) {
            const float2 v = *(const float2*)(stats + (((long)g * tpg + t) * Cp + c) * 2);
            s1 += v.x;
            s2 += v.y;
        }
    }
    r1[lane][cl] = s1;
    r2[lane][cl] = s2;
    __syncthreads();
    if (lane == 0 && c < Cp) {
        s1 = 0.0;
        s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s1 += r1[k][cl];
            s2 += r2[k][cl];
        }
        const double m = s1 * inv_cnt;
        double var = s2 * inv_cnt - m * m;
        var = var < 0.0 ? 0.0 : var;
        *(float2*)(stats + ((long)g * tpg * Cp + c) * 2) = make_float2((float)m, (float)var);
    }
}

// Training-mode forward, critical part in ONE launch: the reduction above plus scale / shift / mean / rstd of the block's own
// (group, channel) entries (no ordering needed between groups).  The running statistics -- the only sequential part -- are left to
// bn_running_kernel, which nothing in the forward or backward pass waits for (the host runs it on the second stream).
__global__ __launch_bounds__(1024) void bn_stats_kernel(float* __restrict__ stats, int tpg, int Cp, int C, double inv_cnt,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                                        float* __restrict__ rstd_o) {
    __shared__ double r1[16][64], r2[16][64];
    const int g = blockIdx.x;
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < Cp) {
        for (int t = lane; t < tpg; t += 16) {
            const float2 v = *(const float2*)(stats + (((long)g * tpg + t) * Cp + c) * 2);
            s1 += v.x;
            s2 += v.y;
        }
    }
    r1[lane][cl] = s1;
    r2[lane][cl] = s2;
    __syncthreads();
    if (lane == 0 && c < Cp) {
        s1 = 0.0;
        s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s1 += r1[k][cl];
            s2 += r2[k][cl];
        }
        const double m = s1 * inv_cnt;
        double var = s2 * inv_cnt - m * m;
        var = var < 0.0 ? 0.0 : var;
        const float mv = (float)m, vv = (float)var;                // exactly what bn_finalize_kernel reads back
        *(float2*)(stats + ((long)g * tpg * Cp + c) * 2) = make_float2(mv, vv);
        const bool real = c < C;
        const float rs = real ? (float)(1.0 / sqrt((double)vv + (double)eps)) : 0.f;
        const float sc = real ? gamma[c] * rs : 0.f;
        const long o = (long)g * Cp + c;
        scale[o] = sc;
        shift[o] = real ? beta[c] - mv * sc : 0.f;
        if (mean_o) mean_o[o] = real ? mv : 0.f;
        if (rstd_o) rstd_o[o] = rs;
    }
}

// running_mean / running_var: one momentum step per group IN ORDER (train/unet.py:179,:196 call BatchNorm once per timestep),
// from the (mean, biased variance) pairs bn_stats_kernel left in the tile-0 slots.
__global__ void bn_running_kernel(const float* __restrict__ stats, int groups, int tpg, int Cp, int C, double unbias,
                                  float* __restrict__ rmean, float* __restrict__ rvar, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = rmean[c], rv = rvar[c];
    for (int g = 0; g < groups; ++g) {
        const float2 mv = *(const float2*)(stats + ((long)g * tpg * Cp + c) * 2);
        const float mom = momentum >= 0.f ? momentum : 1.f / (-momentum + (float)g);
        rm = (1.f - mom) * rm + mom * mv.x;
        rv = (1.f - mom) * rv + mom * (float)((double)mv.y * unbias);
    }
    rmean[c] = rm;
    rvar[c] = rv;
}

// Pass 2: per channel, groups IN ORDER (running statistics are a sequential momentum recursion).
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int groups, int tpg, int Cp, int C,
                                   double unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                   float* __restrict__ rstd_o) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cp) return;
    const bool real = c < C;
    const float ga = real ? gamma[c] : 0.f, be = real ? beta[c] : 0.f;
    float rm = real ? rmean[c] : 0.f, rv = real ? rvar[c] : 1.f;
    for (int g = 0; g < groups; ++g) {
        float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
        if (real) {
            if (stats) {
                const float2 mv = *(const float2*)(stats + ((long)g * tpg * Cp + c) * 2);
                const double var = (double)mv.y;
                mu = mv.x;
                rs = (float)(1.0 / sqrt(var + (double)eps));
                // running stats: one momentum step per group, in group order (train/unet.py:179,:196).  momentum < 0 encodes
                // BatchNorm2d(momentum=None): cumulative average, factor 1/(batches tracked so far + 1), -momentum = that count
                // for the first group
                const float mom = momentum >= 0.f ? momentum : 1.f / (-momentum + (float)g);
                rm = (1.f - mom) * rm + mom * mu;
                rv = (1.f - mom) * rv + mom * (float)(var * unbias);
            } else {
                mu = rm;
                rs = 1.0f / sqrtf(rv + eps);
            }
            sc = ga * rs;
            sh = be - mu * sc;
        }
        const long o = (long)g * Cp + c;
        scale[o] = sc;
        shift[o] = sh;
        if (mean_o) mean_o[o] = mu;
        if (rstd_o) rstd_o[o] = rs;
    }
    if (real && stats) {
        rmean[c] = rm;
        rvar[c] = rv;
    }
}

// a = relu(z*scale + shift)
__global__ void bn_apply_relu_kernel(const uint4* __restrict__ z, uint4* __restrict__ a, const float* __restrict__ scale,
                                     const float* __restrict__ shift, int64_t chunks, FastDiv dcpc, FastDiv dppg, int Cp) {
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t pix = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - pix * dcpc.d;
        const uint32_t g = fdiv(pix, dppg);
        const long so = (long)g * Cp + cc * 8;
        float v[8], sc[8], sh[8];
        unpack8(z[idx], v);
        load8f(scale + so, sc);
        load8f(shift + so, sh);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i] * sc[i] + sh[i], 0.f);
        a[idx] = pack8(v);
    }
}

// Column-reduction geometry: a block of NT threads covers `rows` pixel rows x cpc chunk columns per sweep.
struct ColGeom {
    int cpc;       // 16-byte chunks per pixel
    int rows;      // pixel rows per sweep = NT / cpc (>= 1)
    int active;    // rows * cpc
};
static ColGeom col_geom(int Cp) {
    ColGeom g;
    g.cpc = Cp / 8;
    g.rows = NT / g.cpc;
    if (g.rows < 1) g.rows = 1;
    g.active = g.rows * g.cpc;
    return g;
}

// partials[g*blocks_per_group + bi][c][0..1] = this block's (sum g_, sum g_*xhat).  No atomics: the run-to-run order of f32
// atomic adds here changed the whole gradient by ~1e-3 (the sums feed dz, and BatchNorm backward at random init amplifies
// 1e-7 perturbations through its act16 roundings layer after layer); bn_bwd_sum_kernel adds the rows in a fixed order.
// (no launch bounds on purpose: at 128 VGPRs / four waves per SIMD the kernel spills 36 bytes per lane and runs 4.2 TB/s; with
// __launch_bounds__(256) it does not spill, drops to two or three waves per SIMD and runs 3.3 TB/s)
__global__ void bn_bwd_reduce_kernel(const uint4* __restrict__ z, const uint4* __restrict__ da, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, float* __restrict__ partials, int64_t ppg, int Cp, ColGeom cg,
                                     int blocks_per_group, int64_t pix_per_block) {
    extern __shared__ float red[];     // [rows][cpc*16]
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    // A block takes every blocks_per_group-th SWEEP (4*rows consecutive pixels) of its group, not a contiguous pixel range: the
    // blocks of a group then read next to each other at every moment (one moving front per tensor, as a grid-stride loop has)
    // instead of from blocks_per_group distant places.  The block -> pixel mapping is fixed, so the sums stay reproducible.
    const int64_t gbeg = (int64_t)g * ppg, gend = gbeg + ppg;
    const int64_t sweep = 4 * (int64_t)cg.rows;
    (void)pix_per_block;
    // thread t handles chunk column t % cpc of rows t / cpc + k*rows; columns beyond 256 threads loop
    for (int cbase = 0; cbase < cg.cpc; cbase += NT) {
        const int cc = cbase + (cg.cpc >= NT ? threadIdx.x : threadIdx.x % cg.cpc);
        const int prow = cg.cpc >= NT ? 0 : threadIdx.x / cg.cpc;
        const bool act = (cg.cpc >= NT) ? (cc < cg.cpc) : (threadIdx.x < cg.active);
        float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (act) {
            const long so = (long)g * Cp + cc * 8;
            float sc[8], sh[8], mu[8];
            load8f(scale + so, sc);
            load8f(shift + so, sh);
            load8f(mean + so, mu);
            // s2 = rstd * sum g_*(z - mean): rstd is a per-channel constant, applied once after the loop (one fused multiply-add per
            // element instead of subtract, two multiplies and an add; eight registers fewer inside the loop)
            int64_t p = gbeg + (int64_t)bi * sweep + prow;
            const int64_t step = sweep * blocks_per_group;
            for (; p + 3 * cg.rows < gend; p += step) {             // four pixel rows (eight 16-byte loads) in flight
                uint4 zq[4], gq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    zq[u] = z[(p + u * cg.rows) * cg.cpc + cc];
                    gq[u] = da[(p + u * cg.rows) * cg.cpc + cc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float zv[8], gv[8];
                    unpack8(zq[u], zv);
                    unpack8(gq[u], gv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float g0 = (zv[i] * sc[i] + sh[i] > 0.f) ? gv[i] : 0.f;
                        s1[i] += g0;
                        s2[i] = fmaf(g0, zv[i] - mu[i], s2[i]);
                    }
                }
            }
            // the group's last, partial sweep (at most one per thread: p advanced past every full one)
            for (int u = 0; u < 4 && p + u * cg.rows < gend; ++u) {
                float zv[8], gv[8];
                unpack8(z[(p + u * cg.rows) * cg.cpc + cc], zv);
                unpack8(da[(p + u * cg.rows) * cg.cpc + cc], gv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float gg = (zv[i] * sc[i] + sh[i] > 0.f) ? gv[i] : 0.f;
                    s1[i] += gg;
                    s2[i] = fmaf(gg, zv[i] - mu[i], s2[i]);
                }
            }
            float rs[8];
            load8f(rstd + so, rs);
#pragma unroll
            for (int i = 0; i < 8; ++i) s2[i] *= rs[i];
        }
        const int width = min(cg.cpc, NT) * 16;
        if (act) {
            float* r = red + prow * width + (cc - cbase) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                r[i * 2] = s1[i];
                r[i * 2 + 1] = s2[i];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < width; j += NT) {
            float t = 0.f;
            for (int r = 0; r < cg.rows; ++r) t += red[r * width + j];
            const int ch = (cbase * 8) + (j >> 1);
            if (ch < Cp) partials[((long)blockIdx.x * Cp + ch) * 2 + (j & 1)] = t;
        }
        __syncthreads();
    }
}

// Same result as bn_bwd_apply_kernel below with the per-(group, channel) constants held in registers: a block owns a pixel
// range of ONE group, a thread one 16-byte channel chunk of every `rows`-th pixel row, so
//     dz = sc*(g_ - s1/n - xhat*s2/n) = sc*g_ + k1*z + k0,   k1 = -sc*rs*s2/n,  k0 = -sc*s1/n - k1*mu
// costs two loads, one store and a handful of FMAs per chunk (the generic kernel re-loads ten float4 of constants per chunk).
// Fixed-order sum of the partial rows: block = (group, 32 consecutive (channel, j) entries), 8 row lanes each adding every
// 8th row in f64, then lane 0 adds the 8 lane sums in order.
__global__ __launch_bounds__(256) void bn_bwd_sum_kernel(const float* __restrict__ partials, float* __restrict__ sums, int blocks_per_group,
                                                         int Cp) {
    __shared__ double red[8][32];
    const int g = blockIdx.y;
    const int e = blockIdx.x * 32 + (threadIdx.x & 31);          // entry (c*2 + j) of the group
    const int lane = threadIdx.x >> 5;
    double t = 0.0;
    if (e < Cp * 2)
        for (int b = lane; b < blocks_per_group; b += 8) t += (double)partials[((long)(g * blocks_per_group + b) * Cp) * 2 + e];
    red[lane][threadIdx.x & 31] = t;
    __syncthreads();
    if (lane == 0 && e < Cp * 2) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
        sums[(long)g * Cp * 2 + e] = (float)s;
    }
}

__global__ void bn_bwd_apply_cols_kernel(const uint4* __restrict__ z, const uint4* __restrict__ da, const float* __restrict__ scale,
                                         const float* __restrict__ shift, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ sums, uint4* __restrict__ dz,
                                         int64_t ppg, int Cp, ColGeom cg, int blocks_per_group, int64_t pix_per_block, float inv_n) {
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t p0 = (int64_t)g * ppg + (int64_t)bi * pix_per_block;
    const int64_t p1 = min((int64_t)(g + 1) * ppg, p0 + pix_per_block);
    if ((int)threadIdx.x >= cg.active) return;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    const long so = (long)g * Cp + cc * 8;
    float sc[8], sh[8], k1[8], k0[8];
    {
        float mu[8], rs[8];
        load8f(scale + so, sc);
        load8f(shift + so, sh);
        load8f(mean + so, mu);
        load8f(rstd + so, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s1 = sums[(so + i) * 2], s2 = sums[(so + i) * 2 + 1];
            k1[i] = -sc[i] * rs[i] * s2 * inv_n;
            k0[i] = -sc[i] * s1 * inv_n - k1[i] * mu[i];
        }
    }
    int64_t p = p0 + prow;
    for (; p + cg.rows < p1; p += 2 * cg.rows) {
        const int64_t ia = p * cg.cpc + cc, ib = (p + cg.rows) * cg.cpc + cc;
        const uint4 za = z[ia], ga = da[ia], zb = z[ib], gb = da[ib];
        float zv[8], gv[8], zw[8], gw[8], oa[8], ob[8];
        unpack8(za, zv);
        unpack8(ga, gv);
        unpack8(zb, zw);
        unpack8(gb, gw);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            oa[i] = ((zv[i] * sc[i] + sh[i] > 0.f) ? sc[i] * gv[i] : 0.f) + k1[i] * zv[i] + k0[i];
            ob[i] = ((zw[i] * sc[i] + sh[i] > 0.f) ? sc[i] * gw[i] : 0.f) + k1[i] * zw[i] + k0[i];
        }
        dz[ia] = pack8(oa);
        dz[ib] = pack8(ob);
    }
    for (; p < p1; p += cg.rows) {
        const int64_t ia = p * cg.cpc + cc;
        float zv[8], gv[8], oa[8];
        unpack8(z[ia], zv);
        unpack8(da[ia], gv);
#pragma unroll
        for (int i = 0; i < 8; ++i) oa[i] = ((zv[i] * sc[i] + sh[i] > 0.f) ? sc[i] * gv[i] : 0.f) + k1[i] * zv[i] + k0[i];
        dz[ia] = pack8(oa);
    }
}

__global__ void bn_bwd_apply_kernel(const uint4* __restrict__ z, const uint4* __restrict__ da, const float* __restrict__ scale,
                                    const float* __restrict__ shift, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ sums, uint4* __restrict__ dz,
                                    int64_t chunks, FastDiv dcpc, FastDiv dppg, int Cp, float inv_n) {
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t pix = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - pix * dcpc.d;
        const uint32_t g = fdiv(pix, dppg);
        const long so = (long)g * Cp + cc * 8;
        float zv[8], gv[8], sc[8], sh[8], mu[8], rs[8], o[8];
        unpack8(z[idx], zv);
        unpack8(da[idx], gv);
        load8f(scale + so, sc);
        load8f(shift + so, sh);
        load8f(mean + so, mu);
        load8f(rstd + so, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gg = (zv[i] * sc[i] + sh[i] > 0.f) ? gv[i] : 0.f;
            const float xh = (zv[i] - mu[i]) * rs[i];
            const float s1 = sums[(so + i) * 2], s2 = sums[(so + i) * 2 + 1];
            o[i] = sc[i] * (gg - s1 * inv_n - xh * s2 * inv_n);
        }
        dz[idx] = pack8(o);
    }
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(2)
// ---------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const uint4* __restrict__ a, uint4* __restrict__ p, int64_t chunks, FastDiv dcpc, FastDiv dWo,
                                   FastDiv dHo, int H, int W) {
    const int cpc = dcpc.d, Wo = dWo.d, Ho = dHo.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t opix = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - opix * cpc;
        const uint32_t t = fdiv(opix, dWo);
        const uint32_t xo = opix - t * Wo;
        const uint32_t img = fdiv(t, dHo);
        const uint32_t yo = t - img * Ho;
        const int64_t base = (((int64_t)img * H + 2 * yo) * W + 2 * xo) * cpc + cc;
        float v0[8], v1[8], v2[8], v3[8], o[8];
        unpack8(a[base], v0);
        unpack8(a[base + cpc], v1);
        unpack8(a[base + (int64_t)W * cpc], v2);
        unpack8(a[base + (int64_t)W * cpc + cpc], v3);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaxf(fmaxf(v0[i], v1[i]), fmaxf(v2[i], v3[i]));
        p[idx] = pack8(o);
    }
}

// add (may be null): a second gradient of the pooled tensor's INPUT (the skip connection of the UNet: train/unet.py:166-169
// feed x_k both to the next Down and to the decoder), summed here instead of by a separate elementwise kernel.
__global__ void maxpool_bwd_kernel(const uint4* __restrict__ a, const uint4* __restrict__ dp, const uint4* __restrict__ add,
                                   uint4* __restrict__ da, int64_t chunks, FastDiv dcpc, FastDiv dWo, FastDiv dHo, int H, int W) {
    const int cpc = dcpc.d, Wo = dWo.d, Ho = dHo.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t opix = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - opix * cpc;
        const uint32_t t = fdiv(opix, dWo);
        const uint32_t xo = opix - t * Wo;
        const uint32_t img = fdiv(t, dHo);
        const uint32_t yo = t - img * Ho;
        const int64_t b0 = (((int64_t)img * H + 2 * yo) * W + 2 * xo) * cpc + cc;
        const int64_t offs[4] = {0, cpc, (int64_t)W * cpc, (int64_t)W * cpc + cpc};
        float v[4][8], g[8], o[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) unpack8(a[b0 + offs[k]], v[k]);
        unpack8(dp[idx], g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // first maximum in scan order (0,0),(0,1),(1,0),(1,1): strict '>' keeps the earlier one
            int best = 0;
            float m = v[0][i];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][i] > m) {
                    m = v[k][i];
                    best = k;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][i] = (k == best) ? g[i] : 0.f;
        }
        if (add) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s8[8];
                unpack8(add[b0 + offs[k]], s8);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[k][i] += s8[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) da[b0 + offs[k]] = pack8(o[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(2) fused into the BatchNorm stage that feeds it (the second stage of inc / down1..3: train/unet.py:81, :166-169)
// ---------------------------------------------------------------------------------------------
// Every encoder block output is used twice: by the next block's pooling and by the decoder's skip connection.  Unfused, the
// forward pass re-reads the activation to pool it, and the backward pass runs a kernel that scatters the pooled gradient,
// adds the skip gradient and WRITES the sum, which the BatchNorm backward reduction and apply then each read again.
// Fused: bn_apply_relu writes the pooled tensor too; the BatchNorm backward kernels take (skip gradient, pooled gradient),
// recompute the stored activation a = act16(relu(z*scale + shift)) of the 2x2 window from z to find the arg-max (first maximum in
// scan order, strict '>': ATen's rule, as maxpool_bwd_kernel) and form da = act16(skip + scatter(dp)) on the fly.
// Thread = one 16-byte channel chunk of `rows` interleaved 2x2 WINDOWS of one BatchNorm group (H, W even).
struct PoolGeom {
    int H, W, Ho, Wo;
    FastDiv dWo, dHoWo;
    int64_t wpg;           // windows per group = images per group * Ho * Wo
};
__device__ __forceinline__ int64_t pool_base(const PoolGeom& pg, int64_t wi, int cpc, int cc) {
    const uint32_t img = fdiv((uint32_t)wi, pg.dHoWo);
    const uint32_t r = (uint32_t)wi - img * pg.dHoWo.d;
    const uint32_t yo = fdiv(r, pg.dWo);
    const uint32_t xo = r - yo * pg.Wo;
    return (((int64_t)img * pg.H + 2 * yo) * pg.W + 2 * xo) * cpc + cc;
}

__global__ void bn_apply_relu_pool_kernel(const uint4* __restrict__ z, uint4* __restrict__ a, uint4* __restrict__ pl,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int Cp, ColGeom cg,
                                          PoolGeom pg, int blocks_per_group, int64_t win_per_block) {
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t w0 = (int64_t)g * pg.wpg + (int64_t)bi * win_per_block;
    const int64_t w1 = min((int64_t)(g + 1) * pg.wpg, w0 + win_per_block);
    if ((int)threadIdx.x >= cg.active) return;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    const long so = (long)g * Cp + cc * 8;
    float sc[8], sh[8];
    load8f(scale + so, sc);
    load8f(shift + so, sh);
    const int64_t offs[4] = {0, cg.cpc, (int64_t)pg.W * cg.cpc, (int64_t)pg.W * cg.cpc + cg.cpc};
    for (int64_t wi = w0 + prow; wi < w1; wi += cg.rows) {
        const int64_t b0 = pool_base(pg, wi, cg.cpc, cc);
        uint4 zq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) zq[k] = z[b0 + offs[k]];
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // post-ReLU values are >= 0: zero is the identity of this max
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[8];
            unpack8(zq[k], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = act_to_f32(f32_to_act(fmaxf(v[i] * sc[i] + sh[i], 0.f)));       // the STORED activation
                m[i] = fmaxf(m[i], v[i]);
            }
            a[b0 + offs[k]] = pack8(v);
        }
        pl[wi * cg.cpc + cc] = pack8(m);
    }
}

// da of the window's four pixels: act16(skip + (k == argmax ? dp : 0)), as maxpool_bwd_kernel stores it
__device__ __forceinline__ void pool_window_grad(const uint4 (&zq)[4], const uint4 (&sq)[4], bool has_skip, const uint4& dq, const float (&sc)[8],
                                                 const float (&sh)[8], float (&zv)[4][8], float (&da)[4][8]) {
    float av[4][8], g[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unpack8(zq[k], zv[k]);
#pragma unroll
        for (int i = 0; i < 8; ++i) av[k][i] = act_to_f32(f32_to_act(fmaxf(zv[k][i] * sc[i] + sh[i], 0.f)));
    }
    unpack8(dq, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int best = 0;
        float m = av[0][i];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (av[k][i] > m) {
                m = av[k][i];
                best = k;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) da[k][i] = (k == best) ? g[i] : 0.f;
    }
    if (has_skip) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s8[8];
            unpack8(sq[k], s8);
#pragma unroll
            for (int i = 0; i < 8; ++i) da[k][i] = act_to_f32(f32_to_act(da[k][i] + s8[i]));
        }
    }
}

__global__ __launch_bounds__(NT) void bn_pool_bwd_reduce_kernel(const uint4* __restrict__ z, const uint4* __restrict__ dskip, const uint4* __restrict__ dp,
                                          const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, float* __restrict__ partials, int Cp, ColGeom cg, PoolGeom pg,
                                          int blocks_per_group, int64_t win_per_block) {
    extern __shared__ float red[];     // [rows][cpc*16]
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t w0 = (int64_t)g * pg.wpg + (int64_t)bi * win_per_block;
    const int64_t w1 = min((int64_t)(g + 1) * pg.wpg, w0 + win_per_block);
    const bool act = (int)threadIdx.x < cg.active;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
        const long so = (long)g * Cp + cc * 8;
        float sc[8], sh[8], mu[8];
        load8f(scale + so, sc);
        load8f(shift + so, sh);
        load8f(mean + so, mu);
        const int64_t offs[4] = {0, cg.cpc, (int64_t)pg.W * cg.cpc, (int64_t)pg.W * cg.cpc + cg.cpc};
        for (int64_t wi = w0 + prow; wi < w1; wi += cg.rows) {
            const int64_t b0 = pool_base(pg, wi, cg.cpc, cc);
            uint4 zq[4], sq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                zq[k] = z[b0 + offs[k]];
                sq[k] = dskip ? dskip[b0 + offs[k]] : make_uint4(0, 0, 0, 0);
            }
            const uint4 dq = dp[wi * cg.cpc + cc];
            float zv[4][8], da[4][8];
            pool_window_grad(zq, sq, dskip != nullptr, dq, sc, sh, zv, da);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float g0 = (zv[k][i] * sc[i] + sh[i] > 0.f) ? da[k][i] : 0.f;
                    s1[i] += g0;
                    s2[i] = fmaf(g0, zv[k][i] - mu[i], s2[i]);
                }
        }
        float rs[8];                                   // rstd once per channel after the loop, as in bn_bwd_reduce_kernel
        load8f(rstd + so, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) s2[i] *= rs[i];
    }
    const int width = cg.cpc * 16;
    if (act) {
        float* r = red + prow * width + cc * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r[i * 2] = s1[i];
            r[i * 2 + 1] = s2[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < width; j += NT) {
        float t = 0.f;
        for (int r = 0; r < cg.rows; ++r) t += red[r * width + j];
        const int ch = j >> 1;
        if (ch < Cp) partials[((long)blockIdx.x * Cp + ch) * 2 + (j & 1)] = t;
    }
}

__global__ __launch_bounds__(NT) void bn_pool_bwd_apply_kernel(const uint4* __restrict__ z, const uint4* __restrict__ dskip, const uint4* __restrict__ dp,
                                         const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ sums, uint4* __restrict__ dz, int Cp,
                                         ColGeom cg, PoolGeom pg, int blocks_per_group, int64_t win_per_block, float inv_n) {
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t w0 = (int64_t)g * pg.wpg + (int64_t)bi * win_per_block;
    const int64_t w1 = min((int64_t)(g + 1) * pg.wpg, w0 + win_per_block);
    if ((int)threadIdx.x >= cg.active) return;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    const long so = (long)g * Cp + cc * 8;
    float sc[8], sh[8], k1[8], k0[8];
    {
        float mu[8], rs[8];
        load8f(scale + so, sc);
        load8f(shift + so, sh);
        load8f(mean + so, mu);
        load8f(rstd + so, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s1 = sums[(so + i) * 2], s2 = sums[(so + i) * 2 + 1];
            k1[i] = -sc[i] * rs[i] * s2 * inv_n;
            k0[i] = -sc[i] * s1 * inv_n - k1[i] * mu[i];
        }
    }
    const int64_t offs[4] = {0, cg.cpc, (int64_t)pg.W * cg.cpc, (int64_t)pg.W * cg.cpc + cg.cpc};
    for (int64_t wi = w0 + prow; wi < w1; wi += cg.rows) {
        const int64_t b0 = pool_base(pg, wi, cg.cpc, cc);
        uint4 zq[4], sq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            zq[k] = z[b0 + offs[k]];
            sq[k] = dskip ? dskip[b0 + offs[k]] : make_uint4(0, 0, 0, 0);
        }
        const uint4 dq = dp[wi * cg.cpc + cc];
        float zv[4][8], da[4][8];
        pool_window_grad(zq, sq, dskip != nullptr, dq, sc, sh, zv, da);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = ((zv[k][i] * sc[i] + sh[i] > 0.f) ? sc[i] * da[k][i] : 0.f) + k1[i] * zv[k][i] + k0[i];
            dz[b0 + offs[k]] = pack8(o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Output head fused into the last BatchNorm stage (train/unet.py:70-71 of up0's second conv + OutConv, :101-107, Co = 1)
// ---------------------------------------------------------------------------------------------
// The activation of the last DoubleConv stage feeds ONLY the 1x1 output convolution.  Unfused, a training step writes it
// (bn_apply_relu), reads it twice (outconv forward, outconv weight gradient), and writes + reads twice its gradient (outconv
// input gradient -> BatchNorm backward reduce / apply): six passes over the largest tensor of the model.  Fused, the activation
// and its gradient never exist in memory: forward  y[p] = b + sum_c w[c] * act16(relu(z[p][c]*scale + shift)),  backward works
// from z and the one-channel dy: da[p][c] = act16(dy[p] * w[c]) exactly as outconv_bwd_da would have stored it.
// Thread = one 16-byte channel chunk column of `rows` interleaved pixel rows; constants in registers; the cpc lanes of a pixel
// are adjacent (cpc a power of two <= 64).
__global__ void bn_head_fwd_kernel(const uint4* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y, int64_t ppg, int Cp,
                                   int C, ColGeom cg, int blocks_per_group, int64_t pix_per_block) {
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t p0 = (int64_t)g * ppg + (int64_t)bi * pix_per_block;
    const int64_t p1 = min((int64_t)(g + 1) * ppg, p0 + pix_per_block);
    const bool act = (int)threadIdx.x < cg.active;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    const long so = (long)g * Cp + cc * 8;
    float sc[8], sh[8], wr[8];
    load8f(scale + so, sc);
    load8f(shift + so, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = cc * 8 + i;
        const float t = w[c < C ? c : 0];
        wr[i] = c < C ? t : 0.f;
    }
    const float bias = b ? b[0] : 0.f;
    // every lane walks the same number of rows (the shuffles below need all lanes of a pixel's group alive)
    const int64_t n_it = (p1 - p0 + cg.rows - 1) / cg.rows;
    for (int64_t it = 0; it < n_it; ++it) {
        const int64_t p = p0 + it * cg.rows + prow;
        const bool ok = act && p < p1;
        float v[8];
        unpack8(z[(ok ? p : p0) * cg.cpc + cc], v);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += act_to_f32(f32_to_act(fmaxf(v[i] * sc[i] + sh[i], 0.f))) * wr[i];
        for (int o = cg.cpc >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (ok && cc == 0) y[p] = acc + bias;
    }
}

// partials as bn_bwd_reduce_kernel; additionally dw[c] += sum_p dy[p] * a[p][c], db += sum_p dy[p] (one f32 atomic per block
// and element, as outconv_bwd_dw_kernel)
__global__ void bn_head_bwd_reduce_kernel(const uint4* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ scale,
                                          const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                                          const float* __restrict__ w, float* __restrict__ partials, float* __restrict__ dw,
                                          float* __restrict__ db, int64_t ppg, int Cp, int C, ColGeom cg, int blocks_per_group,
                                          int64_t pix_per_block) {
    extern __shared__ float red[];     // [rows][cpc*24 + 1]: (s1, s2) pairs, then dw, then db
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t p0 = (int64_t)g * ppg + (int64_t)bi * pix_per_block;
    const int64_t p1 = min((int64_t)(g + 1) * ppg, p0 + pix_per_block);
    const bool act = (int)threadIdx.x < cg.active;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    const int width = cg.cpc * 24 + 1;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.f;
    if (act) {
        const long so = (long)g * Cp + cc * 8;
        // (rstd stays inside the loop here: hoisted as in bn_bwd_reduce_kernel the allocator spills 36 B instead of 12 at the 128-VGPR
        // cap and the kernel is 7 % slower in the step, 90 -> 96 us; profiles/round3_bn_instep_ab.txt)
        float sc[8], sh[8], mu[8], rs[8], wr[8];
        load8f(scale + so, sc);
        load8f(shift + so, sh);
        load8f(mean + so, mu);
        load8f(rstd + so, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cc * 8 + i;
            const float t = w[c < C ? c : 0];
            wr[i] = c < C ? t : 0.f;
        }
        int64_t p = p0 + prow;
        for (; p + 3 * cg.rows < p1; p += 4 * cg.rows) {        // four pixel rows in flight
            uint4 zq[4];
            float gq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                zq[u] = z[(p + u * cg.rows) * cg.cpc + cc];
                gq[u] = dy[p + u * cg.rows];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float zv[8];
                unpack8(zq[u], zv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float yv = zv[i] * sc[i] + sh[i];
                    const float g0 = yv > 0.f ? act_to_f32(f32_to_act(gq[u] * wr[i])) : 0.f;
                    s1[i] += g0;
                    s2[i] += g0 * (zv[i] - mu[i]) * rs[i];
                    sw[i] += gq[u] * act_to_f32(f32_to_act(fmaxf(yv, 0.f)));
                }
                sb += gq[u];
            }
        }
        for (; p < p1; p += cg.rows) {
            float zv[8];
            unpack8(z[p * cg.cpc + cc], zv);
            const float gd = dy[p];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float yv = zv[i] * sc[i] + sh[i];
                const float g0 = yv > 0.f ? act_to_f32(f32_to_act(gd * wr[i])) : 0.f;
                s1[i] += g0;
                s2[i] += g0 * (zv[i] - mu[i]) * rs[i];
                sw[i] += gd * act_to_f32(f32_to_act(fmaxf(yv, 0.f)));
            }
            sb += gd;
        }
        float* r = red + prow * width;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r[cc * 16 + i * 2] = s1[i];
            r[cc * 16 + i * 2 + 1] = s2[i];
            r[cg.cpc * 16 + cc * 8 + i] = sw[i];
        }
        if (cc == 0) r[cg.cpc * 24] = sb;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < width; j += NT) {
        float t = 0.f;
        for (int r = 0; r < cg.rows; ++r) t += red[r * width + j];
        if (j < cg.cpc * 16) {
            const int ch = j >> 1;
            if (ch < Cp) partials[((long)blockIdx.x * Cp + ch) * 2 + (j & 1)] = t;
        } else if (j < cg.cpc * 24) {
            const int ch = j - cg.cpc * 16;
            if (ch < C) atomicAdd(dw + ch, t);
        } else {
            atomicAdd(db, t);
        }
    }
}

__global__ void bn_head_bwd_apply_kernel(const uint4* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ scale,
                                         const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ sums, const float* __restrict__ w, uint4* __restrict__ dz, int64_t ppg,
                                         int Cp, int C, ColGeom cg, int blocks_per_group, int64_t pix_per_block, float inv_n) {
    const int g = blockIdx.x / blocks_per_group;
    const int bi = blockIdx.x - g * blocks_per_group;
    const int64_t p0 = (int64_t)g * ppg + (int64_t)bi * pix_per_block;
    const int64_t p1 = min((int64_t)(g + 1) * ppg, p0 + pix_per_block);
    if ((int)threadIdx.x >= cg.active) return;
    const int cc = threadIdx.x % cg.cpc;
    const int prow = threadIdx.x / cg.cpc;
    const long so = (long)g * Cp + cc * 8;
    float sc[8], sh[8], k1[8], k0[8], wr[8];
    {
        float mu[8], rs[8];
        load8f(scale + so, sc);
        load8f(shift + so, sh);
        load8f(mean + so, mu);
        load8f(rstd + so, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s1 = sums[(so + i) * 2], s2 = sums[(so + i) * 2 + 1];
            k1[i] = -sc[i] * rs[i] * s2 * inv_n;
            k0[i] = -sc[i] * s1 * inv_n - k1[i] * mu[i];
            const int c = cc * 8 + i;
            const float t = w[c < C ? c : 0];
            wr[i] = c < C ? t : 0.f;
        }
    }
    int64_t p = p0 + prow;
    for (; p + cg.rows < p1; p += 2 * cg.rows) {
        const int64_t ia = p * cg.cpc + cc, ib = (p + cg.rows) * cg.cpc + cc;
        const uint4 za = z[ia], zb = z[ib];
        const float ga = dy[p], gb = dy[p + cg.rows];
        float zv[8], zw[8], oa[8], ob[8];
        unpack8(za, zv);
        unpack8(zb, zw);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            oa[i] = ((zv[i] * sc[i] + sh[i] > 0.f) ? sc[i] * act_to_f32(f32_to_act(ga * wr[i])) : 0.f) + k1[i] * zv[i] + k0[i];
            ob[i] = ((zw[i] * sc[i] + sh[i] > 0.f) ? sc[i] * act_to_f32(f32_to_act(gb * wr[i])) : 0.f) + k1[i] * zw[i] + k0[i];
        }
        dz[ia] = pack8(oa);
        dz[ib] = pack8(ob);
    }
    for (; p < p1; p += cg.rows) {
        const int64_t ia = p * cg.cpc + cc;
        float zv[8], oa[8];
        unpack8(z[ia], zv);
        const float ga = dy[p];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            oa[i] = ((zv[i] * sc[i] + sh[i] > 0.f) ? sc[i] * act_to_f32(f32_to_act(ga * wr[i])) : 0.f) + k1[i] * zv[i] + k0[i];
        dz[ia] = pack8(oa);
    }
}

// ---------------------------------------------------------------------------------------------
// ConvLSTM backward, point-wise part
// ---------------------------------------------------------------------------------------------
// Split-K form of the cell forward: pre-activations arrive as f32 [pixels][N] in gate-interleaved panel-row order
// (n = hb*64 + gate*16 + j <-> hidden channel hb*16 + j).  One thread = 4 hidden channels of one pixel.
__device__ __forceinline__ void lstm_fwd_pw_body(float* pre, int nslab, int64_t slab, int clear, const float* __restrict__ pre_add,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ c_prev, float* __restrict__ c_out, act16* __restrict__ h_out,
                                                 act16* __restrict__ gates_out, int64_t items, FastDiv dq, int Hd_p, int N, int blk, int nblk) {
    for (int64_t idx = (int64_t)blk * NT + threadIdx.x; idx < items; idx += (int64_t)nblk * NT) {
        const uint32_t pix = fdiv((uint32_t)idx, dq);
        const int hc = ((uint32_t)idx - pix * dq.d) * 4;                 // first hidden channel of the quad
        const int nb = (hc >> 4) * 64 + (hc & 15);
        const float* pp = pre + (int64_t)pix * N + nb;
        float g4[4][4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            // hoisted x half of the gate convolution (one GEMM over all timesteps) + the split-K slabs of W_h * h_{t-1}
            float4 v = pre_add ? *(const float4*)(pre_add + (int64_t)pix * N + nb + gate * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int sl = 0; sl < nslab; ++sl) {          // split-K slabs (plain stores of the K ranges)
                const float4 u = *(const float4*)(pp + (int64_t)sl * slab + gate * 16);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            // consume-and-clear: an atomic split-K GEMM of the next timestep accumulates into this buffer again
            if (clear && nslab > 0) *(float4*)(const_cast<float*>(pp) + gate * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = bias ? *(const float4*)(bias + nb + gate * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            g4[gate][0] = v.x + b.x; g4[gate][1] = v.y + b.y; g4[gate][2] = v.z + b.z; g4[gate][3] = v.w + b.w;
        }
        float cp[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t so = (int64_t)pix * Hd_p + hc;
        if (c_prev) { const float4 t = *(const float4*)(c_prev + so); cp[0] = t.x; cp[1] = t.y; cp[2] = t.z; cp[3] = t.w; }
        float cn[4];
        Pack8 hi, gi, gf, gg, go;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vi = fast_sigmoid(g4[0][r]), vf = fast_sigmoid(g4[1][r]);
            const float vg = fast_tanh(g4[2][r]), vo = fast_sigmoid(g4[3][r]);
            cn[r] = vf * cp[r] + vi * vg;
            hi.e[r] = f32_to_act(vo * fast_tanh(cn[r]));
            gi.e[r] = f32_to_act(vi); gf.e[r] = f32_to_act(vf); gg.e[r] = f32_to_act(vg); go.e[r] = f32_to_act(vo);
        }
        *(float4*)(c_out + so) = make_float4(cn[0], cn[1], cn[2], cn[3]);
        *(uint2*)(h_out + so) = hi.u;
        if (gates_out) {
            act16* gp = gates_out + (int64_t)pix * 4 * Hd_p + hc;
            *(uint2*)(gp) = gi.u;
            *(uint2*)(gp + Hd_p) = gf.u;
            *(uint2*)(gp + 2 * Hd_p) = gg.u;
            *(uint2*)(gp + 3 * Hd_p) = go.u;
        }
    }
}

__global__ void lstm_fwd_pw_kernel(float* pre, int nslab, int64_t slab, int clear, const float* __restrict__ pre_add,
                                   const float* __restrict__ bias,
                                   const float* __restrict__ c_prev, float* __restrict__ c_out, act16* __restrict__ h_out,
                                   act16* __restrict__ gates_out, int64_t items, FastDiv dq, int Hd_p, int N) {
    lstm_fwd_pw_body(pre, nslab, slab, clear, pre_add, bias, c_prev, c_out, h_out, gates_out, items, dq, Hd_p, N, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ void lstm_bwd_pw_body(const uint4* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ c_new,
                                                 const uint4* __restrict__ dh_a, const void* dh_b, int dh_b_is_f32, int dh_b_nslab, int64_t dh_b_slab,
                                                 float* __restrict__ dc_io, int dc_is_zero, uint4* __restrict__ dgates, int64_t chunks, FastDiv dcpc,
                                                 int blk, int nblk) {
    const int cpc = dcpc.d;
    for (int64_t idx = (int64_t)blk * NT + threadIdx.x; idx < chunks; idx += (int64_t)nblk * NT) {
        const uint32_t pix = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - pix * cpc;
        const int64_t gb = (int64_t)pix * 4 * cpc + cc;
        float gi[8], gf[8], gg[8], go[8], cp[8], cn[8], dh[8], dc[8];
        unpack8(gates[gb], gi);
        unpack8(gates[gb + cpc], gf);
        unpack8(gates[gb + 2 * cpc], gg);
        unpack8(gates[gb + 3 * cpc], go);
        if (c_prev) load8f(c_prev + idx * 8, cp);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) cp[i] = 0.f;
        }
        load8f(c_new + idx * 8, cn);
#pragma unroll
        for (int i = 0; i < 8; ++i) dh[i] = 0.f;
        if (dh_a) unpack8(dh_a[idx], dh);
        if (dh_b) {
            float t[8];
            if (dh_b_is_f32) {
                float* p32 = (float*)const_cast<void*>(dh_b) + idx * 8;
                load8f(p32, t);
                for (int sl = 1; sl < dh_b_nslab; ++sl) {      // split-K slabs
                    float u[8];
                    load8f(p32 + sl * dh_b_slab, u);
#pragma unroll
                    for (int i = 0; i < 8; ++i) t[i] += u[i];
                }
                if (dh_b_is_f32 == 2) {     // consume-and-clear (split-K accumulator reused two timesteps later)
                    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    store8f(p32, z8);
                }
            } else {
                unpack8(((const uint4*)dh_b)[idx], t);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) dh[i] += t[i];
        }
        if (dc_is_zero) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dc[i] = 0.f;
        } else {
            load8f(dc_io + idx * 8, dc);
        }
        float di[8], df[8], dg[8], dO[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tc = fast_tanh(cn[i]);
            dO[i] = dh[i] * tc * go[i] * (1.f - go[i]);
            const float dct = dc[i] + dh[i] * go[i] * (1.f - tc * tc);
            df[i] = dct * cp[i] * gf[i] * (1.f - gf[i]);
            di[i] = dct * gg[i] * gi[i] * (1.f - gi[i]);
            dg[i] = dct * gi[i] * (1.f - gg[i] * gg[i]);
            dc[i] = dct * gf[i];
        }
        dgates[gb] = pack8(di);
        dgates[gb + cpc] = pack8(df);
        dgates[gb + 2 * cpc] = pack8(dg);
        dgates[gb + 3 * cpc] = pack8(dO);
        store8f(dc_io + idx * 8, dc);
    }
}

__global__ void lstm_bwd_pw_kernel(const uint4* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ c_new,
                                   const uint4* __restrict__ dh_a, const void* dh_b, int dh_b_is_f32, int dh_b_nslab, int64_t dh_b_slab,
                                   float* __restrict__ dc_io, int dc_is_zero, uint4* __restrict__ dgates, int64_t chunks, FastDiv dcpc) {
    lstm_bwd_pw_body(gates, c_prev, c_new, dh_a, dh_b, dh_b_is_f32, dh_b_nslab, dh_b_slab, dc_io, dc_is_zero, dgates, chunks, dcpc, (int)blockIdx.x,
                     (int)gridDim.x);
}

// The forward point-wise kernels of several independent ConvLSTMs (one timestep of each) as ONE launch: block ranges per member.
constexpr int PW_GROUP_MAX = 4;
struct LstmFwdPwGroup {
    int n;
    int first[PW_GROUP_MAX + 1];
    uclstm_lstm_fwd_pw_args a[PW_GROUP_MAX];
    FastDiv dq[PW_GROUP_MAX];
};
__global__ void lstm_fwd_pw_group_kernel(const LstmFwdPwGroup g) {
    const int b = (int)blockIdx.x;
#pragma unroll
    for (int j = 0; j < PW_GROUP_MAX; ++j) {
        if (j < g.n && b >= g.first[j] && b < g.first[j + 1]) {
            const uclstm_lstm_fwd_pw_args& a = g.a[j];
            lstm_fwd_pw_body(a.pre, a.nslab, a.slab, a.clear, a.pre_add, a.bias, a.c_prev, a.c_out, (act16*)a.h_out, (act16*)a.gates_out,
                             a.pixels * (a.Hd_p / 4), g.dq[j], a.Hd_p, 64 * ((a.Hd_p + 15) / 16), b - g.first[j], g.first[j + 1] - g.first[j]);
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Layout conversion (module boundary)
// ---------------------------------------------------------------------------------------------
// thread <-> (chunk column, pixel) with the pixel fastest: NCHW reads are coalesced.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, uint4* __restrict__ out, int64_t items, int C, int cpc, FastDiv dHW,
                                    FastDiv dcpc, FastDiv dinner, int64_t inner_stride, int64_t outer_stride) {
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * NT) {
        // idx = (img*cpc + cc)*HW + pix
        const uint32_t t = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - t * HW;
        const uint32_t img = fdiv(t, dcpc);
        const uint32_t cc = t - img * cpc;
        const uint32_t o = fdiv(img, dinner);
        const uint32_t in = img - o * dinner.d;
        const float* src = x + (int64_t)in * outer_stride + (int64_t)o * inner_stride;   // image i = o*inner + in reads [in][o]
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // load from a clamped address, then select: a load under a condition becomes a branch with its own s_waitcnt
            // vmcnt(0) -- eight round trips per thread instead of eight loads in flight
            const int c = cc * 8 + i;
            const float t8 = src[(int64_t)(c < C ? c : 0) * HW + pix];
            v[i] = c < C ? t8 : 0.f;
        }
        out[((int64_t)img * HW + pix) * cpc + cc] = pack8(v);
    }
}

__global__ void nhwc_to_nchw_kernel(const uint4* __restrict__ a, float* __restrict__ out, int64_t items, int C, int cpc, FastDiv dHW,
                                    FastDiv dcpc) {
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * NT) {
        const uint32_t t = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - t * HW;
        const uint32_t img = fdiv(t, dcpc);
        const uint32_t cc = t - img * cpc;
        float v[8];
        unpack8(a[((int64_t)img * HW + pix) * cpc + cc], v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cc * 8 + i;
            if (c < C) out[((int64_t)img * C + c) * HW + pix] = v[i];
        }
    }
}

__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t items, int C, int Cp, FastDiv dHW,
                                        FastDiv dCp) {
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * NT) {
        // idx = (img*Cp + c)*HW + pix
        const uint32_t t = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - t * HW;
        const uint32_t img = fdiv(t, dCp);
        const uint32_t c = t - img * Cp;
        out[((int64_t)img * HW + pix) * Cp + c] = (int)c < C ? x[((int64_t)img * C + c) * HW + pix] : 0.f;
    }
}
__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ a, float* __restrict__ out, int64_t items, int C, int Cp, FastDiv dHW,
                                        FastDiv dC) {
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * NT) {
        // idx = (img*C + c)*HW + pix
        const uint32_t t = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - t * HW;
        const uint32_t img = fdiv(t, dC);
        const uint32_t c = t - img * C;
        out[idx] = a[((int64_t)img * HW + pix) * Cp + c];
    }
}

__global__ void im2col_first_kernel(const float* __restrict__ x, uint4* __restrict__ out, int64_t items, int C, int kpc, int H, int W,
                                    FastDiv dHW, FastDiv dW, FastDiv dkpc, FastDiv dinner, FastDiv dC, int64_t inner_stride,
                                    int64_t outer_stride) {
    const int HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * NT) {
        // idx = (img*kpc + kc)*HW + pix
        const uint32_t t = fdiv((uint32_t)idx, dHW);
        const uint32_t pix = (uint32_t)idx - t * HW;
        const uint32_t img = fdiv(t, dkpc);
        const uint32_t kc = t - img * kpc;
        const int y = (int)fdiv(pix, dW);
        const int xx = (int)pix - y * W;
        const uint32_t o = fdiv(img, dinner);
        const uint32_t in = img - o * dinner.d;
        const float* src = x + (int64_t)in * outer_stride + (int64_t)o * inner_stride;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t k = kc * 8 + i;
            const uint32_t tap = fdiv(k, dC);
            const int c = (int)(k - tap * C);
            const int ys = y + (int)(tap / 3) - 1, xs = xx + (int)(tap % 3) - 1;
            const bool ok = tap < 9 && (unsigned)ys < (unsigned)H && (unsigned)xs < (unsigned)W;
            const float t8 = src[ok ? (int64_t)c * HW + ys * W + xs : 0];          // branch-free, as in nchw_to_nhwc_kernel
            v[i] = ok ? t8 : 0.f;
        }
        out[((int64_t)img * HW + pix) * kpc + kc] = pack8(v);
    }
}

// ---------------------------------------------------------------------------------------------
// OutConv 1x1
// ---------------------------------------------------------------------------------------------
__global__ void outconv_fwd_kernel(const uint4* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b,
                                   float* __restrict__ y, int64_t pixels, FastDiv dHW, int cpc, int C, int Co) {
    const int HW = dHW.d;
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * NT) {
        const uint32_t img = fdiv((uint32_t)p, dHW);
        const uint32_t pix = (uint32_t)p - img * HW;
        for (int co = 0; co < Co; ++co) {
            float acc = b ? b[co] : 0.f;
            for (int cc = 0; cc < cpc; ++cc) {
                float v[8];
                unpack8(a[p * cpc + cc], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = cc * 8 + i;
                    if (c < C) acc += v[i] * w[co * C + c];
                }
            }
            y[((int64_t)img * Co + co) * HW + pix] = acc;
        }
    }
}

// Same result with LPP = cpc (a power of two <= 64) lanes per pixel: every lane loads ONE 16-byte chunk (a pixel row is one
// coalesced run over its lanes), keeps its 8 weights in registers and the lanes of a pixel combine by xor-shuffles.
template <int LPP>
__global__ void outconv_fwd_lanes_kernel(const uint4* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b,
                                         float* __restrict__ y, int64_t pixels, FastDiv dHW, int C, int Co) {
    const int HW = dHW.d;
    const int cc = threadIdx.x % LPP;
    const int slot = threadIdx.x / LPP;
    constexpr int PPB = NT / LPP;
    for (int co = 0; co < Co; ++co) {
        float wr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[i] = (cc * 8 + i < C) ? w[co * C + cc * 8 + i] : 0.f;
        const float bias = b ? b[co] : 0.f;
        for (int64_t p = (int64_t)blockIdx.x * PPB + slot; p < pixels; p += (int64_t)gridDim.x * PPB) {
            float v[8];
            unpack8(a[p * LPP + cc], v);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i] * wr[i];
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (cc == 0) {
                const uint32_t img = fdiv((uint32_t)p, dHW);
                const uint32_t pix = (uint32_t)p - img * HW;
                y[((int64_t)img * Co + co) * HW + pix] = acc + bias;
            }
        }
    }
}

__global__ void outconv_bwd_da_kernel(const float* __restrict__ w, const float* __restrict__ dy, uint4* __restrict__ da, int64_t chunks,
                                      FastDiv dcpc, FastDiv dHW, int C, int Co) {
    const int cpc = dcpc.d, HW = dHW.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t p = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - p * cpc;
        const uint32_t img = fdiv(p, dHW);
        const uint32_t pix = p - img * HW;
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int co = 0; co < Co; ++co) {
            const float g = dy[((int64_t)img * Co + co) * HW + pix];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cc * 8 + i;
                const float wv = w[co * C + (c < C ? c : 0)];                      // branch-free, as in nchw_to_nhwc_kernel
                o[i] += g * (c < C ? wv : 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((int)(cc * 8 + i) >= C) o[i] = 0.f;                               // padding channels stay exactly zero (g may be inf)
        da[idx] = pack8(o);
    }
}

// dw[co][c] += sum_p dy[p][co]*a[p][c]; db[co] += sum_p dy[p][co]   (one co per blockIdx.y)
__global__ void outconv_bwd_dw_kernel(const uint4* __restrict__ a, const float* __restrict__ dy, float* __restrict__ dw,
                                      float* __restrict__ db, int64_t pixels, FastDiv dHW, ColGeom cg, int C, int Co,
                                      int64_t pix_per_block) {
    extern __shared__ float red[];   // [rows][cpc*8 + 1]
    const int co = blockIdx.y;
    const int HW = dHW.d;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(pixels, p0 + pix_per_block);
    const int width = cg.cpc * 8 + 1;
    for (int j = threadIdx.x; j < cg.rows * width; j += NT) red[j] = 0.f;
    __syncthreads();
    if ((int)threadIdx.x < cg.active) {
        const int cc = threadIdx.x % cg.cpc, prow = threadIdx.x / cg.cpc;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.f;
        for (int64_t p = p0 + prow; p < p1; p += cg.rows) {
            const uint32_t img = fdiv((uint32_t)p, dHW);
            const uint32_t pix = (uint32_t)p - img * HW;
            const float g = dy[((int64_t)img * Co + co) * HW + pix];
            float v[8];
            unpack8(a[p * cg.cpc + cc], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += g * v[i];
            sb += g;
        }
        float* r = red + prow * width;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[cc * 8 + i] = s[i];
        if (cc == 0) r[cg.cpc * 8] = sb;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < width; j += NT) {
        float t = 0.f;
        for (int r = 0; r < cg.rows; ++r) t += red[r * width + j];
        if (j < cg.cpc * 8) {
            if (j < C) atomicAdd(dw + co * C + j, t);
        } else {
            atomicAdd(db + co, t);
        }
    }
}

// Column sums of a [pixels][Cp] act16 tensor (bias gradients).  grid.x = pixel ranges, grid.y = groups of NT 16-byte
// column chunks; a thread owns one chunk column of `rows` interleaved pixel rows and keeps four loads in flight.
__global__ void colsum_kernel(const uint4* __restrict__ a, float* __restrict__ out, int64_t pixels, ColGeom cg, int Cp,
                              int64_t pix_per_block) {
    extern __shared__ float red[];   // [rows][min(cpc,NT)*8]
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(pixels, p0 + pix_per_block);
    const int cbase = blockIdx.y * NT;
    const int cc = cbase + (cg.cpc >= NT ? threadIdx.x : threadIdx.x % cg.cpc);
    const int prow = cg.cpc >= NT ? 0 : threadIdx.x / cg.cpc;
    const bool act = (cg.cpc >= NT) ? (cc < cg.cpc) : ((int)threadIdx.x < cg.active);
    const int width = min(cg.cpc, NT) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
        const int64_t st = cg.rows;
        int64_t p = p0 + prow;
        for (; p + 7 * st < p1; p += 8 * st) {                     // eight 16-byte loads in flight (a pure-read kernel: ~3.9 TB/s is the ceiling)
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = a[(p + u * st) * cg.cpc + cc];
            float v[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) unpack8(q[u], v[u]);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += ((v[0][i] + v[1][i]) + (v[2][i] + v[3][i])) + ((v[4][i] + v[5][i]) + (v[6][i] + v[7][i]));
        }
        for (; p + 3 * st < p1; p += 4 * st) {
            const uint4 q0 = a[p * cg.cpc + cc], q1 = a[(p + st) * cg.cpc + cc], q2 = a[(p + 2 * st) * cg.cpc + cc],
                        q3 = a[(p + 3 * st) * cg.cpc + cc];
            float v0[8], v1[8], v2[8], v3[8];
            unpack8(q0, v0);
            unpack8(q1, v1);
            unpack8(q2, v2);
            unpack8(q3, v3);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
        }
        for (; p < p1; p += st) {
            float v[8];
            unpack8(a[p * cg.cpc + cc], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
        float* r = red + prow * width + (cc - cbase) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = s[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < width; j += NT) {
        float t = 0.f;
        for (int r = 0; r < cg.rows; ++r) t += red[r * width + j];
        const int ch = cbase * 8 + j;
        if (ch < Cp && (cg.cpc >= NT ? (cbase + j / 8) < cg.cpc : true)) atomicAdd(out + ch, t);
    }
}

bool aligned16(const void* p) { return p && ((uintptr_t)p % 16) == 0; }


// ---------------------------------------------------------------------------------------------
// SpatialAttention (train/unet.py:113-125): channel mean & max -> k x k conv (2 -> 1, no bias) -> sigmoid -> scale
// ---------------------------------------------------------------------------------------------
// Small maps (the bottleneck: 4x4 .. 32x32 pixels, C = 16 * base_ch channels), HBM-bound on x; five kernels forward + backward
// per call, each one pass.  desc[pixel] = (mean_c x, max_c x) in f32, arg[pixel] = first channel that attains the max.
__global__ __launch_bounds__(256) void attn_desc_kernel(const uint4* __restrict__ x, float* __restrict__ desc, int* __restrict__ arg,
                                                        int64_t pixels, int Cp, int C) {
    const int lane = threadIdx.x & 63;
    const int cpc = Cp >> 3;
    for (int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < pixels; pix += (int64_t)gridDim.x * 4) {
        float sum = 0.f, mx = -INFINITY;
        int am = 0x7fffffff;
        for (int ch = lane; ch < cpc; ch += 64) {
            float f[8];
            unpack8(x[pix * cpc + ch], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = ch * 8 + i;
                if (c < C) {
                    sum += f[i];
                    if (f[i] > mx) { mx = f[i]; am = c; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o, 64);
            const float omx = __shfl_xor(mx, o, 64);
            const int oam = __shfl_xor(am, o, 64);
            if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
        }
        if (lane == 0) {
            desc[pix * 2] = sum / (float)C;
            desc[pix * 2 + 1] = mx;
            arg[pix] = am;
        }
    }
}

// att[p] = sigmoid(sum_{j,ky,kx} w[j][ky][kx] * desc[p + (ky - k/2, kx - k/2)][j])   (zero padding)
__global__ void attn_map_kernel(const float* __restrict__ desc, const float* __restrict__ w, float* __restrict__ att, int64_t pixels, int H,
                                int W, int k) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= pixels) return;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t img0 = pix - (int64_t)yy * W - xx;
    const int r = k / 2;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int y2 = yy + ky - r;
        if ((unsigned)y2 >= (unsigned)H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int x2 = xx + kx - r;
            if ((unsigned)x2 >= (unsigned)W) continue;
            const float2 dsc = *(const float2*)(desc + (img0 + (int64_t)y2 * W + x2) * 2);
            acc += w[ky * k + kx] * dsc.x + w[k * k + ky * k + kx] * dsc.y;
        }
    }
    att[pix] = 1.f / (1.f + __expf(-acc));
}

// out = x * att[pixel]
__global__ void attn_scale_kernel(const uint4* __restrict__ x, const float* __restrict__ att, uint4* __restrict__ out, int64_t chunks,
                                  FastDiv dcpc) {
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t pix = fdiv((uint32_t)idx, dcpc);
        const float a = att[pix];
        float f[8];
        unpack8(x[idx], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] *= a;
        out[idx] = pack8(f);
    }
}

// dpre[p] = (sum_c dout[p][c] * x[p][c]) * att (1 - att)
__global__ __launch_bounds__(256) void attn_bwd_dpre_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dout,
                                                            const float* __restrict__ att, float* __restrict__ dpre, int64_t pixels,
                                                            int Cp) {
    const int lane = threadIdx.x & 63;
    const int cpc = Cp >> 3;
    for (int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < pixels; pix += (int64_t)gridDim.x * 4) {
        float acc = 0.f;
        for (int ch = lane; ch < cpc; ch += 64) {
            float a[8], b[8];
            unpack8(x[pix * cpc + ch], a);
            unpack8(dout[pix * cpc + ch], b);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += a[i] * b[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            const float a = att[pix];
            dpre[pix] = acc * a * (1.f - a);
        }
    }
}

// ddesc[q][j] = sum_{ky,kx} w[j][ky][kx] * dpre[q - (ky - r, kx - r)]  (transposed convolution of the 2 -> 1 conv)
__global__ void attn_bwd_ddesc_kernel(const float* __restrict__ dpre, const float* __restrict__ w, float* __restrict__ ddesc,
                                      int64_t pixels, int H, int W, int k) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= pixels) return;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t img0 = pix - (int64_t)yy * W - xx;
    const int r = k / 2;
    float d0 = 0.f, d1 = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int y2 = yy - (ky - r);
        if ((unsigned)y2 >= (unsigned)H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int x2 = xx - (kx - r);
            if ((unsigned)x2 >= (unsigned)W) continue;
            const float g = dpre[img0 + (int64_t)y2 * W + x2];
            d0 += w[ky * k + kx] * g;
            d1 += w[k * k + ky * k + kx] * g;
        }
    }
    ddesc[pix * 2] = d0;
    ddesc[pix * 2 + 1] = d1;
}

// dw[j][ky][kx] (+)= sum_p dpre[p] * desc[p + (ky - r, kx - r)][j]: one block per weight element, fixed-order block reduction
__global__ __launch_bounds__(256) void attn_bwd_dw_kernel(const float* __restrict__ dpre, const float* __restrict__ desc, float* __restrict__ dw,
                                                          int64_t pixels, int H, int W, int k, int accumulate) {
    __shared__ double red[256];
    const int e = blockIdx.x;
    const int j = e / (k * k), ky = (e / k) % k, kx = e % k;
    const int r = k / 2;
    double acc = 0.0;
    for (int64_t pix = threadIdx.x; pix < pixels; pix += 256) {
        const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
        const int y2 = yy + ky - r, x2 = xx + kx - r;
        if ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W)
            acc += (double)(dpre[pix] * desc[(pix - (int64_t)yy * W - xx + (int64_t)y2 * W + x2) * 2 + j]);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) dw[e] = (accumulate ? dw[e] : 0.f) + (float)red[0];
}

// dx[p][c] = dout[p][c] * att[p] + ddesc[p][0] / C + (c == arg[p]) * ddesc[p][1]     (c < C)
__global__ void attn_bwd_dx_kernel(const uint4* __restrict__ dout, const float* __restrict__ att, const float* __restrict__ ddesc,
                                   const int* __restrict__ arg, uint4* __restrict__ dx, int64_t chunks, FastDiv dcpc, int C) {
    const float invC = 1.f / (float)C;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t pix = fdiv((uint32_t)idx, dcpc);
        const int c0 = (int)((uint32_t)idx - pix * dcpc.d) * 8;
        const float a = att[pix];
        const float2 dd = *(const float2*)(ddesc + (int64_t)pix * 2);
        const int am = arg[pix];
        float f[8];
        unpack8(dout[idx], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i;
            f[i] = c < C ? f[i] * a + dd.x * invC + (c == am ? dd.y : 0.f) : 0.f;
        }
        dx[idx] = pack8(f);
    }
}
}  // namespace

// =============================================================================================
#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_bn_finalize(float* stats, int32_t groups, int32_t tiles_per_group, int32_t Cp, int32_t C,
                                      int64_t count_per_group, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                                      float* rstd, void* stream) {
    if (groups <= 0 || Cp <= 0 || C <= 0 || C > Cp || !gamma || !beta || !running_mean || !running_var || !scale || !shift)
        return UCLSTM_E_BADARG;
    if (stats && (tiles_per_group <= 0 || count_per_group <= 0)) return UCLSTM_E_BADARG;
    const double unb = (stats && count_per_group > 1) ? (double)count_per_group / (double)(count_per_group - 1) : 1.0;
    if (stats) {
        // the partial-sum buffer is consumed (tile-0 slots are overwritten with mean/variance)
        UCLSTM_LAUNCH(bn_reduce_kernel, dim3(groups, (Cp + 63) / 64), dim3(1024), 0, (hipStream_t)stream, const_cast<float*>(stats),
                      tiles_per_group, Cp, 1.0 / (double)count_per_group);
    }
    UCLSTM_LAUNCH(bn_finalize_kernel, dim3((Cp + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, groups, tiles_per_group, Cp, C,
                  unb, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_stats_fwd(float* stats, int32_t groups, int32_t tiles_per_group, int32_t Cp, int32_t C, int64_t count_per_group,
                                       const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd,
                                       void* stream) {
    if (!stats || groups <= 0 || tiles_per_group <= 0 || Cp <= 0 || C <= 0 || C > Cp || count_per_group <= 0 || !gamma || !beta || !scale || !shift)
        return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(bn_stats_kernel, dim3(groups, (Cp + 63) / 64), dim3(1024), 0, (hipStream_t)stream, stats, tiles_per_group, Cp, C,
                  1.0 / (double)count_per_group, gamma, beta, eps, scale, shift, mean, rstd);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_running_stats(const float* stats, int32_t groups, int32_t tiles_per_group, int32_t Cp, int32_t C,
                                           int64_t count_per_group, float* running_mean, float* running_var, float momentum, void* stream) {
    if (!stats || groups <= 0 || tiles_per_group <= 0 || Cp <= 0 || C <= 0 || C > Cp || count_per_group <= 0 || !running_mean || !running_var)
        return UCLSTM_E_BADARG;
    const double unb = count_per_group > 1 ? (double)count_per_group / (double)(count_per_group - 1) : 1.0;
    UCLSTM_LAUNCH(bn_running_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, groups, tiles_per_group, Cp, C, unb,
                  running_mean, running_var, momentum);
    return UCLSTM_OK;
}
#endif

extern "C" int32_t uclstm_bn_apply_relu(const void* z, void* a, const float* scale, const float* shift, int64_t pixels,
                                        int64_t pixels_per_group, int32_t Cp, void* stream) {
    if (!aligned16(z) || !aligned16(a) || !scale || !shift || pixels <= 0 || pixels_per_group <= 0 || Cp <= 0 || (Cp % 8))
        return UCLSTM_E_BADARG;
    const int64_t chunks = pixels * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(bn_apply_relu_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, (uint4*)a, scale,
                       shift, chunks, make_fastdiv(Cp / 8), make_fastdiv((uint32_t)pixels_per_group), Cp);
    return UCLSTM_OK;
}

namespace {
inline int bn_bwd_blocks_per_group(int64_t pixels_per_group, int groups) {
    // 1024 blocks in total (four per CU).  Alone, on cold tensors, the reduction takes 142 us instead of 156 at 4096 on the 64-channel
    // full-resolution stage and 77 instead of 86 on the 128-channel one (profiles/round3_bn_reduce_blocks.txt).  Inside the step the
    // reduction kernel itself is 6 % slower than with 4096 (its inputs were just written) and the gain is in bn_bwd_sum_kernel, which
    // adds a quarter of the rows (8.2 -> 5.5 us), and in the head variant (profiles/round3_bn_reduce_blocks_instep.txt); backward phase
    // -0.08 ms in three of three same-box pairs
    static const int total = [] { const char* e = getenv("UCLSTM_BN_BWD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    // at least 32 pixel rows per block (128 until round 3: the 8x8 and 4x4 stages of the headline step then ran on 320 and 80 blocks;
    // in the serialised step bn_bwd_apply_cols 68.2 -> 63.6 us on average, the reduction no slower; profiles/round3_bn_instep_ab.txt)
    static const int min_rows = [] { const char* e = getenv("UCLSTM_BN_BWD_MIN_ROWS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 32; }();
    int bpg = (int)((pixels_per_group + min_rows - 1) / min_rows);
    // rounded DOWN: the reductions hold four blocks per CU (122 VGPRs), i.e. 1024 at once; with 20 groups a cap of ceil(1024 / 20) = 52
    // made 1040 blocks, and the last 16 ran as a second round on an empty chip (~10 us of tail on a 150-us launch)
    const int cap = total / groups;
    if (bpg > cap) bpg = cap;
    return bpg < 1 ? 1 : bpg;
}
}  // namespace

#ifndef UCLSTM_ACT_F16
extern "C" int64_t uclstm_bn_bwd_reduce_rows(int64_t pixels, int64_t pixels_per_group) {
    if (pixels <= 0 || pixels_per_group <= 0 || (pixels % pixels_per_group)) return UCLSTM_E_BADARG;
    const int groups = (int)(pixels / pixels_per_group);
    return (int64_t)groups * bn_bwd_blocks_per_group(pixels_per_group, groups);
}
#endif

extern "C" int32_t uclstm_bn_bwd_reduce(const void* z, const void* da, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, float* partials, float* sums, int64_t pixels, int64_t pixels_per_group,
                                        int32_t Cp, void* stream) {
    if (!aligned16(z) || !aligned16(da) || !scale || !shift || !mean || !rstd || !partials || !sums || pixels <= 0 ||
        pixels_per_group <= 0 || (pixels % pixels_per_group) || Cp <= 0 || (Cp % 8))
        return UCLSTM_E_BADARG;
    const ColGeom cg = col_geom(Cp);
    const int groups = (int)(pixels / pixels_per_group);
    const int bpg = bn_bwd_blocks_per_group(pixels_per_group, groups);
    const int64_t ppb = (pixels_per_group + bpg - 1) / bpg;
    const size_t lds = (size_t)cg.rows * (cg.cpc < NT ? cg.cpc : NT) * 16 * sizeof(float);
    UCLSTM_LAUNCH(bn_bwd_reduce_kernel, dim3(groups * bpg), dim3(NT), lds, (hipStream_t)stream, (const uint4*)z, (const uint4*)da,
                       scale, shift, mean, rstd, partials, pixels_per_group, Cp, cg, bpg, ppb);
    UCLSTM_LAUNCH(bn_bwd_sum_kernel, dim3((Cp * 2 + 31) / 32, groups), dim3(256), 0, (hipStream_t)stream, partials, sums, bpg, Cp);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_bwd_apply(const void* z, const void* da, const float* scale, const float* shift, const float* mean,
                                       const float* rstd, const float* sums, void* dz, int64_t pixels, int64_t pixels_per_group,
                                       int32_t Cp, void* stream) {
    if (!aligned16(z) || !aligned16(da) || !aligned16(dz) || !scale || !shift || !mean || !rstd || !sums || pixels <= 0 ||
        pixels_per_group <= 0 || Cp <= 0 || (Cp % 8))
        return UCLSTM_E_BADARG;
    const int64_t chunks = pixels * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    const ColGeom cg = col_geom(Cp);
    if (cg.cpc <= NT && (pixels % pixels_per_group) == 0) {
        const int groups = (int)(pixels / pixels_per_group);
        // at least 32 pixel rows per block (128 until round 3: the 8x8 and 4x4 stages of the headline step then ran on 320 and 80 blocks;
    // in the serialised step bn_bwd_apply_cols 68.2 -> 63.6 us on average, the reduction no slower; profiles/round3_bn_instep_ab.txt)
    static const int min_rows = [] { const char* e = getenv("UCLSTM_BN_BWD_MIN_ROWS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 32; }();
    int bpg = (int)((pixels_per_group + min_rows - 1) / min_rows);
        const int cap = (4096 + groups - 1) / groups;
        if (bpg > cap) bpg = cap;
        if (bpg < 1) bpg = 1;
        const int64_t ppb = (pixels_per_group + bpg - 1) / bpg;
        UCLSTM_LAUNCH(bn_bwd_apply_cols_kernel, dim3(groups * bpg), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, (const uint4*)da,
                      scale, shift, mean, rstd, sums, (uint4*)dz, pixels_per_group, Cp, cg, bpg, ppb,
                      (float)(1.0 / (double)pixels_per_group));
        return UCLSTM_OK;
    }
    UCLSTM_LAUNCH(bn_bwd_apply_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, (const uint4*)da,
                       scale, shift, mean, rstd, sums, (uint4*)dz, chunks, make_fastdiv(Cp / 8),
                       make_fastdiv((uint32_t)pixels_per_group), Cp, (float)(1.0 / (double)pixels_per_group));
    return UCLSTM_OK;
}

// ---- MaxPool2d(2) fused into the BatchNorm stage that feeds it (see bn_apply_relu_pool_kernel) ----
static bool pool_plan(int64_t n_img, int H, int W, int Cp, int groups, ColGeom& cg, PoolGeom& pg, int& bpg, int64_t& wpb) {
    if (n_img <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || groups <= 0 || (n_img % groups) || Cp <= 0 || (Cp % 8) || Cp / 8 > NT) return false;
    if (n_img * H * W * (Cp / 8) >= ((int64_t)1 << 31)) return false;
    cg = col_geom(Cp);
    pg.H = H; pg.W = W; pg.Ho = H / 2; pg.Wo = W / 2;
    pg.dWo = make_fastdiv((uint32_t)pg.Wo);
    pg.dHoWo = make_fastdiv((uint32_t)(pg.Ho * pg.Wo));
    pg.wpg = (n_img / groups) * (int64_t)pg.Ho * pg.Wo;
    bpg = (int)((pg.wpg + 31) / 32);                     // >= 32 windows (128 pixels) per thread row
    // total block count rounded DOWN per group (see bn_bwd_blocks_per_group); UCLSTM_BN_POOL_BLOCKS for A/B runs
    static const int total = [] { const char* e = getenv("UCLSTM_BN_POOL_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();
    const int cap = total / groups;
    if (bpg > cap) bpg = cap;
    if (bpg < 1) bpg = 1;
    wpb = (pg.wpg + bpg - 1) / bpg;
    return true;
}

#ifndef UCLSTM_ACT_F16
extern "C" int64_t uclstm_bn_pool_bwd_rows(int64_t n_img, int32_t H, int32_t W, int32_t Cp, int32_t groups) {
    ColGeom cg; PoolGeom pg; int bpg; int64_t wpb;
    if (!pool_plan(n_img, H, W, Cp, groups, cg, pg, bpg, wpb)) return UCLSTM_E_BADARG;
    return (int64_t)groups * bpg;
}
#endif

extern "C" int32_t uclstm_bn_apply_relu_pool(const void* z, void* a, void* p, const float* scale, const float* shift, int64_t n_img,
                                             int32_t H, int32_t W, int32_t Cp, int32_t groups, void* stream) {
    ColGeom cg; PoolGeom pg; int bpg; int64_t wpb;
    if (!aligned16(z) || !aligned16(a) || !aligned16(p) || !scale || !shift || !pool_plan(n_img, H, W, Cp, groups, cg, pg, bpg, wpb))
        return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(bn_apply_relu_pool_kernel, dim3(groups * bpg), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, (uint4*)a, (uint4*)p, scale,
                  shift, Cp, cg, pg, bpg, wpb);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_pool_bwd_reduce(const void* z, const void* dskip, const void* dp, const float* scale, const float* shift,
                                             const float* mean, const float* rstd, float* partials, float* sums, int64_t n_img, int32_t H,
                                             int32_t W, int32_t Cp, int32_t groups, void* stream) {
    ColGeom cg; PoolGeom pg; int bpg; int64_t wpb;
    if (!aligned16(z) || !aligned16(dp) || (dskip && !aligned16(dskip)) || !scale || !shift || !mean || !rstd || !partials || !sums ||
        !pool_plan(n_img, H, W, Cp, groups, cg, pg, bpg, wpb))
        return UCLSTM_E_BADARG;
    const size_t lds = (size_t)cg.rows * cg.cpc * 16 * sizeof(float);
    UCLSTM_LAUNCH(bn_pool_bwd_reduce_kernel, dim3(groups * bpg), dim3(NT), lds, (hipStream_t)stream, (const uint4*)z, (const uint4*)dskip,
                  (const uint4*)dp, scale, shift, mean, rstd, partials, Cp, cg, pg, bpg, wpb);
    UCLSTM_LAUNCH(bn_bwd_sum_kernel, dim3((Cp * 2 + 31) / 32, groups), dim3(256), 0, (hipStream_t)stream, partials, sums, bpg, Cp);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_pool_bwd_apply(const void* z, const void* dskip, const void* dp, const float* scale, const float* shift,
                                            const float* mean, const float* rstd, const float* sums, void* dz, int64_t n_img, int32_t H,
                                            int32_t W, int32_t Cp, int32_t groups, void* stream) {
    ColGeom cg; PoolGeom pg; int bpg; int64_t wpb;
    if (!aligned16(z) || !aligned16(dp) || !aligned16(dz) || (dskip && !aligned16(dskip)) || !scale || !shift || !mean || !rstd || !sums ||
        !pool_plan(n_img, H, W, Cp, groups, cg, pg, bpg, wpb))
        return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(bn_pool_bwd_apply_kernel, dim3(groups * bpg), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, (const uint4*)dskip,
                  (const uint4*)dp, scale, shift, mean, rstd, sums, (uint4*)dz, Cp, cg, pg, bpg, wpb,
                  (float)(1.0 / (double)((n_img / groups) * (int64_t)H * W)));
    return UCLSTM_OK;
}

// ---- output head fused into the last BatchNorm stage (Co = 1; see bn_head_fwd_kernel) ----
static bool head_geom_ok(int Cp, int C) {
    const int cpc = Cp / 8;
    return Cp > 0 && (Cp % 8) == 0 && C > 0 && C <= Cp && cpc <= 64 && (cpc & (cpc - 1)) == 0;
}
static void head_blocks(int64_t pixels_per_group, int groups, int& bpg, int64_t& ppb) {
    bpg = (int)((pixels_per_group + 127) / 128);
    const int cap = (4096 + groups - 1) / groups;
    if (bpg > cap) bpg = cap;
    if (bpg < 1) bpg = 1;
    ppb = (pixels_per_group + bpg - 1) / bpg;
}

extern "C" int32_t uclstm_bn_head_fwd(const void* z, const float* scale, const float* shift, const float* w, const float* b, float* y,
                                      int64_t pixels, int64_t pixels_per_group, int32_t Cp, int32_t C, void* stream) {
    if (!aligned16(z) || !scale || !shift || !w || !y || pixels <= 0 || pixels_per_group <= 0 || (pixels % pixels_per_group) ||
        !head_geom_ok(Cp, C))
        return UCLSTM_E_BADARG;
    if (pixels * (Cp / 8) >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    const ColGeom cg = col_geom(Cp);
    const int groups = (int)(pixels / pixels_per_group);
    int bpg;
    int64_t ppb;
    head_blocks(pixels_per_group, groups, bpg, ppb);
    UCLSTM_LAUNCH(bn_head_fwd_kernel, dim3(groups * bpg), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, scale, shift, w, b, y,
                  pixels_per_group, Cp, C, cg, bpg, ppb);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_head_bwd_reduce(const void* z, const float* dy, const float* scale, const float* shift, const float* mean,
                                             const float* rstd, const float* w, float* partials, float* sums, float* dw, float* db,
                                             int64_t pixels, int64_t pixels_per_group, int32_t Cp, int32_t C, void* stream) {
    if (!aligned16(z) || !dy || !scale || !shift || !mean || !rstd || !w || !partials || !sums || !dw || !db || pixels <= 0 ||
        pixels_per_group <= 0 || (pixels % pixels_per_group) || !head_geom_ok(Cp, C))
        return UCLSTM_E_BADARG;
    const ColGeom cg = col_geom(Cp);
    const int groups = (int)(pixels / pixels_per_group);
    const int bpg = bn_bwd_blocks_per_group(pixels_per_group, groups);          // = uclstm_bn_bwd_reduce_rows / groups
    const int64_t ppb = (pixels_per_group + bpg - 1) / bpg;
    const size_t lds = (size_t)cg.rows * (cg.cpc * 24 + 1) * sizeof(float);
    UCLSTM_LAUNCH(bn_head_bwd_reduce_kernel, dim3(groups * bpg), dim3(NT), lds, (hipStream_t)stream, (const uint4*)z, dy, scale, shift, mean,
                  rstd, w, partials, dw, db, pixels_per_group, Cp, C, cg, bpg, ppb);
    UCLSTM_LAUNCH(bn_bwd_sum_kernel, dim3((Cp * 2 + 31) / 32, groups), dim3(256), 0, (hipStream_t)stream, partials, sums, bpg, Cp);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_bn_head_bwd_apply(const void* z, const float* dy, const float* scale, const float* shift, const float* mean,
                                            const float* rstd, const float* sums, const float* w, void* dz, int64_t pixels,
                                            int64_t pixels_per_group, int32_t Cp, int32_t C, void* stream) {
    if (!aligned16(z) || !aligned16(dz) || !dy || !scale || !shift || !mean || !rstd || !sums || !w || pixels <= 0 ||
        pixels_per_group <= 0 || (pixels % pixels_per_group) || !head_geom_ok(Cp, C))
        return UCLSTM_E_BADARG;
    if (pixels * (Cp / 8) >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    const ColGeom cg = col_geom(Cp);
    const int groups = (int)(pixels / pixels_per_group);
    int bpg;
    int64_t ppb;
    head_blocks(pixels_per_group, groups, bpg, ppb);
    UCLSTM_LAUNCH(bn_head_bwd_apply_kernel, dim3(groups * bpg), dim3(NT), 0, (hipStream_t)stream, (const uint4*)z, dy, scale, shift, mean,
                  rstd, sums, w, (uint4*)dz, pixels_per_group, Cp, C, cg, bpg, ppb, (float)(1.0 / (double)pixels_per_group));
    return UCLSTM_OK;
}

namespace {
__global__ void bn_bwd_param_grads_kernel(const float* __restrict__ sums, int groups, int Cp, int C, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int g = 0; g < groups; ++g) {
        const float2 v = *(const float2*)(sums + ((long)g * Cp + c) * 2);
        s1 += v.x;
        s2 += v.y;
    }
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s1;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s2;
}
}  // namespace

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_bn_bwd_param_grads(const float* sums, int32_t groups, int32_t Cp, int32_t C, float* dgamma, float* dbeta,
                                             int32_t accumulate, void* stream) {
    if (!sums || !dgamma || !dbeta || groups <= 0 || Cp <= 0 || C <= 0 || C > Cp) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(bn_bwd_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, groups, Cp, C, dgamma,
                  dbeta, accumulate);
    return UCLSTM_OK;
}
#endif

extern "C" int32_t uclstm_maxpool2_fwd(const void* a, void* p, int32_t n_img, int32_t H, int32_t W, int32_t Cp, void* stream) {
    if (!aligned16(a) || !aligned16(p) || n_img <= 0 || H < 2 || W < 2 || Cp <= 0 || (Cp % 8)) return UCLSTM_E_BADARG;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t chunks = (int64_t)n_img * Ho * Wo * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(maxpool_fwd_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, (const uint4*)a, (uint4*)p, chunks,
                       make_fastdiv(Cp / 8), make_fastdiv(Wo), make_fastdiv(Ho), H, W);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_maxpool2_bwd(const void* a, const void* dp, const void* add, void* da, int32_t n_img, int32_t H, int32_t W,
                                       int32_t Cp, void* stream) {
    if (!aligned16(a) || !aligned16(dp) || !aligned16(da) || n_img <= 0 || H < 2 || W < 2 || Cp <= 0 || (Cp % 8)) return UCLSTM_E_BADARG;
    if (add && (!aligned16(add) || (H % 2) || (W % 2))) return UCLSTM_E_BADARG;      // fused add: every input pixel is covered
    const int Ho = H / 2, Wo = W / 2;
    const int64_t chunks = (int64_t)n_img * Ho * Wo * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(maxpool_bwd_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)dp,
                       (const uint4*)add, (uint4*)da, chunks, make_fastdiv(Cp / 8), make_fastdiv(Wo), make_fastdiv(Ho), H, W);
    return UCLSTM_OK;
}

namespace {
// Finish of a split-K convolution with the STORE epilogue (inference on few pixels: a 32 x 32 bottleneck gives 32 tiles for 256
// CUs, so the GEMM runs as K ranges that store f32 partial tiles): out = act16(relu?((sum of slabs + bias) * scale + shift)),
// the same expression, in the same order, as the GEMM's own epilogue.  Thread = 8 channels of a pixel.
__global__ void splitk_finish_kernel(const float* __restrict__ pre, int nslab, int64_t slab, int ld, const float* __restrict__ bias,
                                     const float* __restrict__ scale, const float* __restrict__ shift, int relu, uint4* __restrict__ out,
                                     int64_t chunks, FastDiv dcpc) {
    const int cpc = dcpc.d;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * NT) {
        const uint32_t p = fdiv((uint32_t)idx, dcpc);
        const uint32_t cc = (uint32_t)idx - p * cpc;
        const float* src = pre + (int64_t)p * ld + cc * 8;
        float4 a0 = *(const float4*)src, a1 = *(const float4*)(src + 4);
        for (int sl = 1; sl < nslab; ++sl) {
            const float4 t0 = *(const float4*)(src + sl * slab), t1 = *(const float4*)(src + sl * slab + 4);
            a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
            a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
        }
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float bs[8], sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bs[i] = bias ? bias[cc * 8 + i] : 0.f;
            sc[i] = scale ? scale[cc * 8 + i] : 1.f;
            sh[i] = shift ? shift[cc * 8 + i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = (v[i] + bs[i]) * sc[i] + sh[i];
            if (relu) x = fmaxf(x, 0.f);
            v[i] = x;
        }
        out[idx] = pack8(v);
    }
}
}  // namespace

extern "C" int32_t uclstm_splitk_finish(const float* pre, int32_t nslab, int64_t slab, int32_t ld, const float* bias, const float* col_scale,
                                        const float* col_shift, int32_t relu, void* out, int64_t pixels, int32_t C, void* stream) {
    if (!pre || !out || !aligned16(pre) || !aligned16(out) || nslab < 1 || pixels <= 0 || C <= 0 || (C % 8) || ld < C || (ld % 4)) return UCLSTM_E_BADARG;
    if (nslab > 1 && (slab < pixels * ld || (slab % 4))) return UCLSTM_E_BADARG;
    const int64_t chunks = pixels * (C / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(splitk_finish_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, pre, nslab, slab, ld, bias, col_scale, col_shift,
                  relu, (uint4*)out, chunks, make_fastdiv(C / 8));
    return UCLSTM_OK;
}

static bool lstm_fwd_pw_ok(float* pre, int32_t nslab, int64_t slab, const float* pre_add, const float* bias, const float* c_prev, float* c_out,
                           void* h_out, void* gates_out, int64_t pixels, int32_t Hd_p) {
    if (!aligned16(c_out) || !aligned16(h_out) || pixels <= 0 || Hd_p <= 0 || (Hd_p % 8)) return false;
    if (nslab < 0 || (nslab > 0 && (!pre || !aligned16(pre))) || (nslab == 0 && !pre_add) || (pre_add && !aligned16(pre_add))) return false;
    if ((c_prev && !aligned16(c_prev)) || (gates_out && !aligned16(gates_out)) || (bias && !aligned16(bias))) return false;
    if (pixels * (Hd_p / 4) >= ((int64_t)1 << 31)) return false;
    if (nslab > 1 && (slab <= 0 || (slab % 4))) return false;
    return true;
}
static bool lstm_bwd_pw_ok(const void* gates, const float* c_prev, const float* c_new, const void* dh_a, const void* dh_b, int32_t dh_b_is_f32,
                           int32_t dh_b_nslab, int64_t dh_b_slab, float* dc_io, void* dgates, int64_t pixels, int32_t Hd_p) {
    if (dh_b && dh_b_is_f32 && (dh_b_nslab < 1 || (dh_b_nslab > 1 && (dh_b_slab <= 0 || (dh_b_slab % 4))))) return false;
    if (!aligned16(gates) || !aligned16(c_new) || !aligned16(dc_io) || !aligned16(dgates) || pixels <= 0 || Hd_p <= 0 || (Hd_p % 8)) return false;
    if ((c_prev && !aligned16(c_prev)) || (dh_a && !aligned16(dh_a)) || (dh_b && !aligned16(dh_b))) return false;
    return pixels * (Hd_p / 8) < ((int64_t)1 << 31);
}

extern "C" int32_t uclstm_lstm_fwd_pointwise_group(const uclstm_lstm_fwd_pw_args* args, int32_t n, void* stream) {
    if (!args || n < 1 || n > PW_GROUP_MAX) return UCLSTM_E_BADARG;
    LstmFwdPwGroup g{};
    g.n = n;
    int at = 0;
    for (int i = 0; i < n; ++i) {
        const uclstm_lstm_fwd_pw_args& a = args[i];
        if (!lstm_fwd_pw_ok(a.pre, a.nslab, a.slab, a.pre_add, a.bias, a.c_prev, a.c_out, a.h_out, a.gates_out, a.pixels, a.Hd_p)) return UCLSTM_E_BADARG;
        g.a[i] = a;
        g.dq[i] = make_fastdiv(a.Hd_p / 4);
        g.first[i] = at;
        at += ew_grid(a.pixels * (a.Hd_p / 4));
    }
    for (int i = n; i <= PW_GROUP_MAX; ++i) g.first[i] = at;
    UCLSTM_LAUNCH(lstm_fwd_pw_group_kernel, dim3(at), dim3(NT), 0, (hipStream_t)stream, g);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_lstm_fwd_pointwise(float* pre, int32_t nslab, int64_t slab, int32_t clear, const float* pre_add, const float* bias,
                                             const float* c_prev, float* c_out, void* h_out, void* gates_out, int64_t pixels, int32_t Hd_p,
                                             void* stream) {
    if (!lstm_fwd_pw_ok(pre, nslab, slab, pre_add, bias, c_prev, c_out, h_out, gates_out, pixels, Hd_p)) return UCLSTM_E_BADARG;
    const int64_t items = pixels * (Hd_p / 4);
    const int N = 64 * ((Hd_p + 15) / 16);
    UCLSTM_LAUNCH(lstm_fwd_pw_kernel, dim3(ew_grid(items)), dim3(NT), 0, (hipStream_t)stream, pre, nslab, slab, clear, pre_add, bias, c_prev, c_out, (act16*)h_out,
                  (act16*)gates_out, items, make_fastdiv(Hd_p / 4), Hd_p, N);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_lstm_bwd_pointwise(const void* gates, const float* c_prev, const float* c_new, const void* dh_a,
                                             const void* dh_b, int32_t dh_b_is_f32, int32_t dh_b_nslab, int64_t dh_b_slab, float* dc_io,
                                             int32_t dc_is_zero, void* dgates, int64_t pixels, int32_t Hd_p, void* stream) {
    if (!lstm_bwd_pw_ok(gates, c_prev, c_new, dh_a, dh_b, dh_b_is_f32, dh_b_nslab, dh_b_slab, dc_io, dgates, pixels, Hd_p)) return UCLSTM_E_BADARG;
    const int64_t chunks = pixels * (Hd_p / 8);
    UCLSTM_LAUNCH(lstm_bwd_pw_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, (const uint4*)gates, c_prev, c_new,
                       (const uint4*)dh_a, dh_b, dh_b_is_f32, dh_b_nslab, dh_b_slab, dc_io, dc_is_zero, (uint4*)dgates, chunks,
                       make_fastdiv(Hd_p / 8));
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_nchw_to_nhwc(const float* x, void* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W,
                                       int32_t inner, int64_t inner_stride, int64_t outer_stride, void* stream) {
    if (!x || !aligned16(out) || n_img <= 0 || C <= 0 || Cp < C || (Cp % 8) || H <= 0 || W <= 0 || inner <= 0 || (n_img % inner))
        return UCLSTM_E_BADARG;
    const int64_t items = (int64_t)n_img * (Cp / 8) * H * W;
    if (items >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(nchw_to_nhwc_kernel, dim3(ew_grid(items)), dim3(NT), 0, (hipStream_t)stream, x, (uint4*)out, items, C, Cp / 8,
                       make_fastdiv(H * W), make_fastdiv(Cp / 8), make_fastdiv(inner), inner_stride, outer_stride);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_nchw_grad_to_nhwc(const float* g, void* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W,
                                            void* stream) {
    return uclstm_nchw_to_nhwc(g, out, n_img, C, Cp, H, W, n_img, 0, (int64_t)C * H * W, stream);
}

extern "C" int32_t uclstm_nhwc_to_nchw(const void* a, float* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W,
                                       void* stream) {
    if (!aligned16(a) || !out || n_img <= 0 || C <= 0 || Cp < C || (Cp % 8) || H <= 0 || W <= 0) return UCLSTM_E_BADARG;
    const int64_t items = (int64_t)n_img * (Cp / 8) * H * W;
    if (items >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(nhwc_to_nchw_kernel, dim3(ew_grid(items)), dim3(NT), 0, (hipStream_t)stream, (const uint4*)a, out, items, C, Cp / 8,
                       make_fastdiv(H * W), make_fastdiv(Cp / 8));
    return UCLSTM_OK;
}

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_nchw_to_nhwc_f32(const float* x, float* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W,
                                           void* stream) {
    if (!x || !out || n_img <= 0 || C <= 0 || Cp < C || H <= 0 || W <= 0) return UCLSTM_E_BADARG;
    const int64_t items = (int64_t)n_img * Cp * H * W;
    if (items >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(nchw_to_nhwc_f32_kernel, dim3(ew_grid(items)), dim3(NT), 0, (hipStream_t)stream, x, out, items, C, Cp,
                       make_fastdiv(H * W), make_fastdiv(Cp));
    return UCLSTM_OK;
}
#endif

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_nhwc_to_nchw_f32(const float* a, float* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W,
                                           void* stream) {
    if (!a || !out || n_img <= 0 || C <= 0 || Cp < C || H <= 0 || W <= 0) return UCLSTM_E_BADARG;
    const int64_t items = (int64_t)n_img * C * H * W;
    if (items >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(nhwc_to_nchw_f32_kernel, dim3(ew_grid(items)), dim3(NT), 0, (hipStream_t)stream, a, out, items, C, Cp,
                       make_fastdiv(H * W), make_fastdiv(C));
    return UCLSTM_OK;
}
#endif

extern "C" int32_t uclstm_im2col3x3_first(const float* x, void* out, int32_t n_img, int32_t C, int32_t Kp, int32_t H, int32_t W,
                                          int32_t inner, int64_t inner_stride, int64_t outer_stride, void* stream) {
    if (!x || !aligned16(out) || n_img <= 0 || C <= 0 || Kp < 9 * C || (Kp % 8) || H <= 0 || W <= 0 || inner <= 0 || (n_img % inner))
        return UCLSTM_E_BADARG;
    const int64_t items = (int64_t)n_img * (Kp / 8) * H * W;
    if (items >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(im2col_first_kernel, dim3(ew_grid(items)), dim3(NT), 0, (hipStream_t)stream, x, (uint4*)out, items, C, Kp / 8, H, W,
                       make_fastdiv(H * W), make_fastdiv(W), make_fastdiv(Kp / 8), make_fastdiv(inner), make_fastdiv(C), inner_stride,
                       outer_stride);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_outconv_fwd(const void* a, const float* w, const float* b, float* y, int64_t n_img, int32_t HW, int32_t Cp,
                                      int32_t C, int32_t Co, void* stream) {
    if (!aligned16(a) || !w || !y || n_img <= 0 || HW <= 0 || Cp < C || (Cp % 8) || C <= 0 || Co <= 0) return UCLSTM_E_BADARG;
    const int64_t pixels = n_img * HW;
    if (pixels >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    const int cpc = Cp / 8;
    const int grid = ew_grid(pixels * cpc);
    const FastDiv dHW = make_fastdiv(HW);
    hipStream_t st = (hipStream_t)stream;
#define UCLSTM_OUTCONV_LANES(L_) \
    UCLSTM_LAUNCH(outconv_fwd_lanes_kernel<L_>, dim3(grid), dim3(NT), 0, st, (const uint4*)a, w, b, y, pixels, dHW, C, Co)
    switch (cpc) {
        case 2: UCLSTM_OUTCONV_LANES(2); break;
        case 4: UCLSTM_OUTCONV_LANES(4); break;
        case 8: UCLSTM_OUTCONV_LANES(8); break;
        case 16: UCLSTM_OUTCONV_LANES(16); break;
        case 32: UCLSTM_OUTCONV_LANES(32); break;
        default:
            UCLSTM_LAUNCH(outconv_fwd_kernel, dim3(ew_grid(pixels)), dim3(NT), 0, st, (const uint4*)a, w, b, y, pixels, dHW, cpc, C, Co);
    }
#undef UCLSTM_OUTCONV_LANES
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_outconv_bwd(const void* a, const float* w, const float* dy, void* da, float* dw, float* db, int64_t n_img,
                                      int32_t HW, int32_t Cp, int32_t C, int32_t Co, void* stream) {
    if (!aligned16(a) || !w || !dy || n_img <= 0 || HW <= 0 || Cp < C || (Cp % 8) || C <= 0 || Co <= 0 || Cp / 8 > NT)
        return UCLSTM_E_BADARG;
    const int64_t pixels = n_img * HW;
    const int64_t chunks = pixels * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    if (da) {
        if (!aligned16(da)) return UCLSTM_E_BADARG;
        UCLSTM_LAUNCH(outconv_bwd_da_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, (hipStream_t)stream, w, dy, (uint4*)da, chunks,
                           make_fastdiv(Cp / 8), make_fastdiv(HW), C, Co);
    }
    if (dw && db) {
        const ColGeom cg = col_geom(Cp);
        int nb = (int)((pixels + 1023) / 1024);
        if (nb > 1024) nb = 1024;
        const int64_t ppb = (pixels + nb - 1) / nb;
        const size_t lds = (size_t)cg.rows * (cg.cpc * 8 + 1) * sizeof(float);
        UCLSTM_LAUNCH(outconv_bwd_dw_kernel, dim3(nb, Co), dim3(NT), lds, (hipStream_t)stream, (const uint4*)a, dy, dw, db, pixels,
                           make_fastdiv(HW), cg, C, Co, ppb);
    }
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_colsum(const void* a, float* out, int64_t pixels, int32_t Cp, void* stream) {
    if (!aligned16(a) || !out || pixels <= 0 || Cp <= 0 || (Cp % 8)) return UCLSTM_E_BADARG;
    const ColGeom cg = col_geom(Cp);
    const int ncg = (cg.cpc + NT - 1) / NT;                    // column groups (grid.y)
    // ~2048 blocks in all (8 per CU), but at least 32 pixels per thread row (four rounds of eight loads; small atomic tail)
    int64_t nb = 2048 / ncg;
    const int64_t maxb = (pixels + 32 * cg.rows - 1) / (32 * cg.rows);
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    const int64_t ppb = (pixels + nb - 1) / nb;
    const size_t lds = (size_t)cg.rows * (cg.cpc < NT ? cg.cpc : NT) * 8 * sizeof(float);
    UCLSTM_LAUNCH(colsum_kernel, dim3((unsigned)nb, (unsigned)ncg), dim3(NT), lds, (hipStream_t)stream, (const uint4*)a, out, pixels, cg, Cp, ppb);
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_attention_fwd(const void* x, const float* w, void* out, float* att, float* desc, int32_t* argmax, int32_t n_img,
                                        int32_t H, int32_t W, int32_t Cp, int32_t C, int32_t k, void* stream) {
    if (!aligned16(x) || !aligned16(out) || !w || !att || !desc || !argmax || n_img <= 0 || H <= 0 || W <= 0 || Cp <= 0 || (Cp % 8) || C <= 0 ||
        C > Cp || k < 1 || !(k & 1) || k > 15 || ((uintptr_t)desc % 8))
        return UCLSTM_E_BADARG;
    const int64_t pixels = (int64_t)n_img * H * W, chunks = pixels * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    UCLSTM_LAUNCH(attn_desc_kernel, dim3((unsigned)std::min<int64_t>((pixels + 3) / 4, 4096)), dim3(256), 0, st, (const uint4*)x, desc, argmax, pixels,
                  Cp, C);
    UCLSTM_LAUNCH(attn_map_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, st, desc, w, att, pixels, H, W, k);
    UCLSTM_LAUNCH(attn_scale_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, st, (const uint4*)x, att, (uint4*)out, chunks, make_fastdiv(Cp / 8));
    return UCLSTM_OK;
}

extern "C" int32_t uclstm_attention_bwd(const void* x, const void* dout, const float* w, const float* att, const float* desc,
                                        const int32_t* argmax, void* dx, float* dw, int32_t dw_accumulate, float* scratch, int32_t n_img,
                                        int32_t H, int32_t W, int32_t Cp, int32_t C, int32_t k, void* stream) {
    if (!aligned16(x) || !aligned16(dout) || !aligned16(dx) || !w || !att || !desc || !argmax || !dw || !scratch || n_img <= 0 || H <= 0 ||
        W <= 0 || Cp <= 0 || (Cp % 8) || C <= 0 || C > Cp || k < 1 || !(k & 1) || k > 15 || ((uintptr_t)scratch % 8))
        return UCLSTM_E_BADARG;
    const int64_t pixels = (int64_t)n_img * H * W, chunks = pixels * (Cp / 8);
    if (chunks >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    float* dpre = scratch;                     // [pixels]
    float* ddesc = scratch + ((pixels + 1) & ~(int64_t)1);   // [pixels][2], 8-byte aligned
    UCLSTM_LAUNCH(attn_bwd_dpre_kernel, dim3((unsigned)std::min<int64_t>((pixels + 3) / 4, 4096)), dim3(256), 0, st, (const uint4*)x,
                  (const uint4*)dout, att, dpre, pixels, Cp);
    UCLSTM_LAUNCH(attn_bwd_ddesc_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, st, dpre, w, ddesc, pixels, H, W, k);
    UCLSTM_LAUNCH(attn_bwd_dw_kernel, dim3(2 * k * k), dim3(256), 0, st, dpre, desc, dw, pixels, H, W, k, dw_accumulate);
    UCLSTM_LAUNCH(attn_bwd_dx_kernel, dim3(ew_grid(chunks)), dim3(NT), 0, st, (const uint4*)dout, att, ddesc, argmax, (uint4*)dx, chunks,
                  make_fastdiv(Cp / 8), C);
    return UCLSTM_OK;
}
