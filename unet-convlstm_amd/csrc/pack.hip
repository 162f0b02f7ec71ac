// Weight panels: f32 reference layouts (OIHW conv, [in,out,2,2] transposed conv) <-> the
// [N][Ktot] panels the implicit-GEMM kernels read, plus bias reordering.  One generic index
// map (uclstm_pack_desc) covers forward panels, flipped/transposed input-gradient panels, the
// gate-interleaved ConvLSTM panel, tap-major ConvTranspose panels and the pre-gathered first
// layer.  Run once per optimiser step per weight; memory-bound, trivially parallel.
#include "common.h"
#include <cstdlib>

namespace {

struct Decoded {
    bool valid;
    int64_t off;
};

// Magic-number divisors of a descriptor (computed once on the host): the per-element kernels are bound by their index
// decode, and runtime 32/64-bit integer division costs tens of instructions each on this ISA.
struct PackDiv {
    FastDiv ktot, per_tap, n_cp, k_hdp, k_hd;
};
PackDiv make_pack_div(const uclstm_pack_desc& d) {
    PackDiv v;
    v.ktot = make_fastdiv((uint32_t)d.Ktot);
    v.per_tap = make_fastdiv((uint32_t)(d.kseg[0] + d.kseg[1]));
    v.n_cp = make_fastdiv((uint32_t)(d.n_cp > 0 ? d.n_cp : 1));
    v.k_hdp = make_fastdiv((uint32_t)(d.k_hdp > 0 ? d.k_hdp : 1));
    v.k_hd = make_fastdiv((uint32_t)(d.k_hd > 0 ? d.k_hd : 1));
    return v;
}

__device__ __forceinline__ bool decode_n(const uclstm_pack_desc& d, const PackDiv& dv, int n, int& n_ent, int& tapn) {
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
        tapn = (int)fdiv((uint32_t)n, dv.n_cp);
        const int co = n - tapn * d.n_cp;
        n_ent = co;
        return co < d.n_valid;
    }
}

__device__ __forceinline__ Decoded decode(const uclstm_pack_desc& d, const PackDiv& dv, int n, int k) {
    Decoded r;
    r.valid = false;
    r.off = 0;
    int n_ent, tapn;
    if (!decode_n(d, dv, n, n_ent, tapn)) return r;
    const int per_tap = d.kseg[0] + d.kseg[1];
    int tap = (int)fdiv((uint32_t)k, dv.per_tap);
    const int kr = k - tap * per_tap;
    const int s = kr >= d.kseg[0] ? 1 : 0;
    const int c = s ? kr - d.kseg[0] : kr;
    int k_ent;
    if (d.k_mode == UCLSTM_KMODE_IDENTITY) {
        if (c >= d.cvalid[s]) return r;
        k_ent = d.choff[s] + c;
    } else if (d.k_mode == UCLSTM_KMODE_GATES) {
        const int gate = (int)fdiv((uint32_t)c, dv.k_hdp);
        const int hc = c - gate * d.k_hdp;
        if (gate >= 4 || hc >= d.k_hd) return r;
        k_ent = d.choff[s] + gate * d.k_hd + hc;
    } else {
        const int tk = (int)fdiv((uint32_t)c, dv.k_hd);
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

__device__ __forceinline__ void pack_generic_body(const uclstm_pack_desc& d, const PackDiv& dv, const float* __restrict__ w,
                                                  act16* __restrict__ wp, uint32_t block, uint32_t nblocks) {
    const uint32_t total = (uint32_t)d.N * (uint32_t)d.Ktot;       // < 2^31 (checked by the launcher)
    for (uint32_t idx = block * blockDim.x + threadIdx.x; idx < total; idx += nblocks * blockDim.x) {
        const int n = (int)fdiv(idx, dv.ktot);
        const int k = (int)(idx - (uint32_t)n * (uint32_t)d.Ktot);
        const Decoded r = decode(d, dv, n, k);
        wp[idx] = f32_to_act(r.valid ? w[r.off] : 0.f);
    }
}
__global__ void pack_kernel(const uclstm_pack_desc d, const PackDiv dv, const float* __restrict__ w, act16* __restrict__ wp) {
    pack_generic_body(d, dv, w, wp, blockIdx.x, gridDim.x);
}

// grid.y > 1 (accumulate only): slab group blockIdx.y adds its share of the slabs with one f32 atomic per element -- for the
// small panels whose weight gradient used hundreds of pixel ranges (first layer: 4096 elements x ~1000 slabs), where one
// thread per element would walk all the slabs serially.
__global__ void unpack_kernel(const uclstm_pack_desc d, const PackDiv dv, const float* __restrict__ dwp, int nslab, int64_t slab,
                              float* __restrict__ grad, int accumulate) {
    const uint32_t total = (uint32_t)d.N * (uint32_t)d.Ktot;
    const int per = (nslab + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * per, s1 = min(nslab, s0 + per);
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int n = (int)fdiv(idx, dv.ktot);
        const int k = (int)(idx - (uint32_t)n * (uint32_t)d.Ktot);
        const Decoded r = decode(d, dv, n, k);
        if (r.valid && s0 < s1) {
            float v = 0.f;
            for (int sl = s0; sl < s1; ++sl) v += dwp[sl * slab + idx];      // partial panels of the pixel ranges
            if (gridDim.y > 1) atomicAdd(grad + r.off, v);
            else grad[r.off] = (accumulate ? grad[r.off] : 0.f) + v;
        }
    }
}

// ---- LDS-transposing kernels (the two layout families that hold 95 % of the weights) --------------------------------
// The per-element kernels above gather with a 36-byte stride (tap is the innermost dimension of OIHW weights, the panel's
// K axis is tap-major): 18 cache lines per wave load, ~1.3 TB/s.  Here the reference-layout side is always touched in
// long contiguous runs and the permutation happens in LDS.
//
// "Row" family (k_mode IDENTITY, stride_k == taps, stride_tap == 1: forward conv / ConvLSTM panels, every weight-gradient
// unpack, ConvTranspose input-gradient panels): for one panel row the (channel, tap) block of a source is ONE contiguous
// run of channels*taps floats.  Block = (panel row, source, 256-channel chunk).
template <int TAPS, bool UNPACK>
__device__ __forceinline__ void pack_rows_body(const uclstm_pack_desc& d, const PackDiv& dv, const float* __restrict__ w,
                                               act16* __restrict__ wp, const float* __restrict__ dwp, int nslab, int64_t slab,
                                               float* __restrict__ grad, int accumulate, int chunks0, int bx, int by) {
    __shared__ float buf[256 * TAPS];
    const int n = by;
    const int s = bx >= chunks0 ? 1 : 0;
    const int c0 = (bx - (s ? chunks0 : 0)) * 256;
    const int seg = s ? d.kseg[1] : d.kseg[0];
    const int per_tap = d.kseg[0] + d.kseg[1];
    int n_ent, tapn;
    const bool ok_n = decode_n(d, dv, n, n_ent, tapn);
    int nval = (s ? d.cvalid[1] : d.cvalid[0]) - c0;
    nval = nval < 0 ? 0 : (nval > 256 ? 256 : nval);
    const int count = ok_n ? nval * TAPS : 0;
    const int64_t roff = (int64_t)n_ent * d.stride_n + (int64_t)tapn * d.stride_ntap + (int64_t)((s ? d.choff[1] : d.choff[0]) + c0) * TAPS;
    const int cl = threadIdx.x;
    const bool col = c0 + cl < seg;                                   // a panel column of this chunk (valid or padding)
    const int64_t pbase = (int64_t)n * d.Ktot + (s ? d.kseg[0] : 0) + c0 + cl;
    if constexpr (!UNPACK) {
        {   // count <= 256*TAPS: exactly TAPS strided passes, all loads issued before the first LDS store
            float v[TAPS];
#pragma unroll
            for (int i = 0; i < TAPS; ++i) {
                const int e = i * 256 + threadIdx.x;
                const float x = w[e < count ? roff + e : 0];         // branch-free: see pack_transposed_body
                v[i] = e < count ? x : 0.f;
            }
#pragma unroll
            for (int i = 0; i < TAPS; ++i) {
                const int e = i * 256 + threadIdx.x;
                if (e < count) buf[e] = v[i];
            }
        }
        __syncthreads();
        if (col) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int ts = d.tap_flip ? TAPS - 1 - t : t;
                wp[pbase + (int64_t)t * per_tap] = f32_to_act((ok_n && cl < nval) ? buf[cl * TAPS + ts] : 0.f);
            }
        }
    } else {
        if (count == 0) return;
        if (cl < nval) {
            float v[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[t] = dwp[pbase + (int64_t)t * per_tap];
            int sl = 1;
            for (; sl + 3 <= nslab; sl += 3) {            // three slabs = 3 * TAPS independent loads per round trip, added in slab order
                float a[3][TAPS];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float* ps = dwp + (sl + j) * slab + pbase;
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) a[j][t] = ps[(int64_t)t * per_tap];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) v[t] += a[j][t];
            }
            for (; sl < nslab; ++sl) {
                const float* ps = dwp + sl * slab + pbase;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) v[t] += ps[(int64_t)t * per_tap];
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) buf[cl * TAPS + (d.tap_flip ? TAPS - 1 - t : t)] = v[t];
        }
        __syncthreads();
        float g[TAPS];
#pragma unroll
        for (int i = 0; i < TAPS; ++i) {
            const int e = i * 256 + threadIdx.x;
            g[i] = (accumulate && e < count) ? grad[roff + e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TAPS; ++i) {
            const int e = i * 256 + threadIdx.x;
            if (e < count) grad[roff + e] = g[i] + buf[e];
        }
    }
}

template <int TAPS, bool UNPACK>
__global__ __launch_bounds__(256) void pack_rows_kernel(const uclstm_pack_desc d, const PackDiv dv, const float* __restrict__ w,
                                                        act16* __restrict__ wp, const float* __restrict__ dwp, int nslab, int64_t slab,
                                                        float* __restrict__ grad, int accumulate, int chunks0) {
    pack_rows_body<TAPS, UNPACK>(d, dv, w, wp, dwp, nslab, slab, grad, accumulate, chunks0, (int)blockIdx.x, (int)blockIdx.y);
}

// "Transposed" family (n_mode IDENTITY, stride_n == taps, stride_tap == 1: conv / ConvLSTM input-gradient panels; the panel
// row is the INPUT channel, the K column an output channel or gate channel): for one K column the (row, tap) block of 16
// consecutive panel rows is one contiguous run of 16*taps floats.  Block = 16 panel rows x 64 K columns.
// (64-row tiles -- 2.3-KiB runs, 512 threads, 72 KiB of LDS -- were tried and are no faster: 2.3-2.9 TB/s against 2.4-3.1 TB/s,
// tools/bench_boundary.py; what held the first version at 1.4 TB/s of reads was not the run length but loads under a condition.)
template <int TAPS>
__device__ __forceinline__ void pack_transposed_body(const uclstm_pack_desc& d, const PackDiv& dv, const float* __restrict__ w,
                                                     act16* __restrict__ wp, int bx, int by) {
    constexpr int ROWS = 16, NTHR = 256;
    constexpr int RUN = ROWS * TAPS;
    constexpr int PITCH = RUN + 2;                                    // in act16: an odd number of dwords per column -> conflict-free column walks
    constexpr int NLD = 64 * RUN / NTHR;                              // loads per thread
    __shared__ act16 buf[64 * PITCH];                                  // 18 KiB (already rounded: the panel is act16), small enough
                                                                      // to share a CU with a 128-KiB weight-gradient block
    __shared__ int64_t kbase[64];
    const int kc0 = bx * 64;
    const int n0 = by * ROWS;
    const int per_tap = d.kseg[0] + d.kseg[1];
    if (threadIdx.x < 64) {
        const int c = kc0 + threadIdx.x;                              // single source (nsrc == 1 in this family)
        int64_t kb = -1;
        if (d.k_mode == UCLSTM_KMODE_IDENTITY) {
            if (c < d.cvalid[0]) kb = (int64_t)(d.choff[0] + c) * d.stride_k;
        } else {
            const int gate = (int)fdiv((uint32_t)c, dv.k_hdp);
            const int hc = c - gate * d.k_hdp;
            if (gate < 4 && hc < d.k_hd) kb = (int64_t)(d.choff[0] + gate * d.k_hd + hc) * d.stride_k;
        }
        kbase[threadIdx.x] = kb;
    }
    __syncthreads();
    int nrow = d.n_valid - n0;
    nrow = nrow < 0 ? 0 : (nrow > ROWS ? ROWS : nrow);
    const int run = nrow * TAPS;
    // ALL of the block's loads are issued before the first LDS store (the first version kept four in flight per thread and was
    // latency-bound at 0.6 TB/s: 4096 blocks x 9 dependent load batches).  Consecutive lanes read consecutive floats of a
    // column's contiguous (ROWS rows x taps) run.
    float v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int e = u * NTHR + threadIdx.x;
        const int kl = e / RUN, r = e - kl * RUN;
        const int64_t kb = kbase[kl];
        // unconditional load from a clamped address + select: a load under a condition compiles to a branch with its own
        // s_waitcnt vmcnt(0), i.e. one round trip per element instead of all loads in flight (reads at 1.4 TB/s)
        const bool ok = r < run && kb >= 0;
        const float x = w[ok ? kb + (int64_t)n0 * TAPS + r : 0];
        v[u] = ok ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int e = u * NTHR + threadIdx.x;
        const int kl = e / RUN, r = e - kl * RUN;
        buf[kl * PITCH + r] = f32_to_act(v[u]);
    }
    __syncthreads();
    // output: panel row (n0 + nl), tap t, 64 consecutive K columns = 128 contiguous bytes; a lane gathers 8 columns from LDS
    // and stores 16 bytes (the first version stored 2 bytes per lane: 36 store instructions per thread instead of 4.5)
    const int g8 = threadIdx.x & 7;
    const bool full = kc0 + 64 <= per_tap;
    for (int sgm = threadIdx.x >> 3; sgm < RUN; sgm += NTHR / 8) {
        const int nl = sgm / TAPS, t = sgm - nl * TAPS;
        if (n0 + nl >= d.N) continue;
        const int ts = d.tap_flip ? TAPS - 1 - t : t;
        Pack16 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = nl < nrow ? buf[(g8 * 8 + j) * PITCH + nl * TAPS + ts] : f32_to_act(0.f);
        act16* dst = wp + (int64_t)(n0 + nl) * d.Ktot + (int64_t)t * per_tap + kc0 + g8 * 8;
        if (full) {
            *(uint4*)dst = o.u;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kc0 + g8 * 8 + j < per_tap) dst[j] = o.e[j];
        }
    }
}

template <int TAPS>
__global__ __launch_bounds__(256) void pack_transposed_kernel(const uclstm_pack_desc d, const PackDiv dv, const float* __restrict__ w,
                                                              act16* __restrict__ wp) {
    pack_transposed_body<TAPS>(d, dv, w, wp, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- batched packing: ALL panels of one kernel family in one launch ------------------------------------------------------
// A training step repacks ~100 panels; as separate 3-15 us launches on one stream they are a latency chain of ~0.7-1 ms that the
// first convolutions of the step wait for (profiles/round2_notes.md).  The host fills a job table once (descriptors, pointers and
// block ranges are the same every step) and launches one kernel per family; a block finds its job by binary search over the
// block offsets and runs the same body as the single-panel kernel.
struct JobView {
    uclstm_pack_desc d;
    PackDiv dv;
    const float* w;
    act16* wp;
    int lb, gx, nblocks;
};
__device__ __forceinline__ JobView find_job(const uclstm_pack_job* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const uclstm_pack_job& J = jobs[lo];
    JobView v;
    v.d = J.d;
    const uint32_t* q = J.div;
    v.dv.ktot = FastDiv{q[0], q[1], q[2]};
    v.dv.per_tap = FastDiv{q[3], q[4], q[5]};
    v.dv.n_cp = FastDiv{q[6], q[7], q[8]};
    v.dv.k_hdp = FastDiv{q[9], q[10], q[11]};
    v.dv.k_hd = FastDiv{q[12], q[13], q[14]};
    v.w = J.w;
    v.wp = (act16*)J.wp;
    v.lb = (int)blockIdx.x - J.block0;
    v.gx = J.gx;
    v.nblocks = J.nblocks;
    return v;
}
template <int TAPS>
__global__ __launch_bounds__(256) void pack_rows_batched_kernel(const uclstm_pack_job* __restrict__ jobs, int njobs) {
    const JobView v = find_job(jobs, njobs);
    const int by = v.lb / v.gx, bx = v.lb - by * v.gx;
    pack_rows_body<TAPS, false>(v.d, v.dv, v.w, v.wp, nullptr, 0, 0, nullptr, 0, (v.d.kseg[0] + 255) / 256, bx, by);
}
template <int TAPS>
__global__ __launch_bounds__(256) void pack_transposed_batched_kernel(const uclstm_pack_job* __restrict__ jobs, int njobs) {
    const JobView v = find_job(jobs, njobs);
    const int by = v.lb / v.gx, bx = v.lb - by * v.gx;
    pack_transposed_body<TAPS>(v.d, v.dv, v.w, v.wp, bx, by);
}
__global__ void pack_generic_batched_kernel(const uclstm_pack_job* __restrict__ jobs, int njobs) {
    const JobView v = find_job(jobs, njobs);
    pack_generic_body(v.d, v.dv, v.w, v.wp, (uint32_t)v.lb, (uint32_t)v.nblocks);
}

// Many-slab panels (C_out <= 64 layers: 64 x 576 elements x ~170 pixel-range slabs): slab 0 += slabs 1..nslab-1 in slab order
// (deterministic), one float4 column per thread with eight independent loads in flight; the unpack then reads one slab.
__global__ __launch_bounds__(256) void slab_fold_kernel(float* __restrict__ dwp, int nslab, int64_t slab, int64_t quads) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    float4* base = (float4*)dwp + q;
    const int64_t sq = slab / 4;
    float4 acc = base[0];
    int sl = 1;
    for (; sl + 8 <= nslab; sl += 8) {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = base[(sl + j) * sq];
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += t[j].x; acc.y += t[j].y; acc.z += t[j].z; acc.w += t[j].w; }
    }
    for (; sl < nslab; ++sl) {
        const float4 t = base[sl * sq];
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    base[0] = acc;
}

__global__ void pack_bias_kernel(const uclstm_pack_desc d, const PackDiv dv, const float* __restrict__ b, float* __restrict__ bp) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= d.N) return;
    int n_ent, tapn;
    const bool ok = decode_n(d, dv, n, n_ent, tapn);
    bp[n] = ok ? b[n_ent] : 0.f;
}

bool desc_ok(const uclstm_pack_desc* d) {
    if (!d || d->N <= 0 || d->Ktot <= 0 || d->taps <= 0 || d->nsrc < 1 || d->nsrc > 2) return false;
    if (d->Ktot != d->taps * (d->kseg[0] + d->kseg[1])) return false;
    if (d->n_mode < 0 || d->n_mode > 2 || d->k_mode < 0 || d->k_mode > 2) return false;
    if (d->n_mode == UCLSTM_NMODE_TAPMAJOR && d->n_cp <= 0) return false;
    if (d->k_mode != UCLSTM_KMODE_IDENTITY && (d->k_hd <= 0 || d->k_hdp <= 0)) return false;
    if ((int64_t)d->N * d->Ktot >= ((int64_t)1 << 31)) return false;
    return true;
}

int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

namespace {
inline bool staged_off() {
    static const bool off = [] { const char* e = getenv("UCLSTM_PACK_GENERIC"); return e && e[0] == '1'; }();
    return off;
}
inline bool rows_family(const uclstm_pack_desc& d) {
    return !staged_off() && d.k_mode == UCLSTM_KMODE_IDENTITY && d.stride_tap == 1 && d.stride_k == d.taps && (d.taps == 9 || d.taps == 4) &&
           d.N <= 65535;
}
inline bool transposed_family(const uclstm_pack_desc& d) {
    return !staged_off() && d.n_mode == UCLSTM_NMODE_IDENTITY && d.nsrc == 1 && d.stride_tap == 1 && d.stride_n == d.taps &&
           (d.taps == 9 || d.taps == 4) && (d.k_mode == UCLSTM_KMODE_IDENTITY || d.k_mode == UCLSTM_KMODE_GATES) && (d.N + 15) / 16 <= 65535;
}
}  // namespace

extern "C" int32_t uclstm_pack_weights(const uclstm_pack_desc* d, const float* w, void* wp, void* stream) {
    if (!desc_ok(d) || !w || !wp) return UCLSTM_E_BADARG;
    if (rows_family(*d)) {
        const int ch0 = (d->kseg[0] + 255) / 256, ch1 = (d->kseg[1] + 255) / 256;
        const dim3 grid(ch0 + ch1, d->N);
        if (d->taps == 9)
            UCLSTM_LAUNCH((pack_rows_kernel<9, false>), grid, dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), w, (act16*)wp, nullptr, 0,
                          (int64_t)0, nullptr, 0, ch0);
        else
            UCLSTM_LAUNCH((pack_rows_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), w, (act16*)wp, nullptr, 0,
                          (int64_t)0, nullptr, 0, ch0);
        return UCLSTM_OK;
    }
    if (transposed_family(*d)) {
        const dim3 grid((d->kseg[0] + d->kseg[1] + 63) / 64, (d->N + 15) / 16);
        if (d->taps == 9) UCLSTM_LAUNCH(pack_transposed_kernel<9>, grid, dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), w, (act16*)wp);
        else UCLSTM_LAUNCH(pack_transposed_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), w, (act16*)wp);
        return UCLSTM_OK;
    }
    UCLSTM_LAUNCH(pack_kernel, dim3(grid_for((int64_t)d->N * d->Ktot)), dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), w, (act16*)wp);
    return UCLSTM_OK;
}


#ifndef UCLSTM_ACT_F16
// family of a descriptor: 0 generic, 1 rows (9 taps), 2 rows (4 taps), 3 transposed (9 taps), 4 transposed (4 taps)
static int pack_family(const uclstm_pack_desc& d) {
    if (rows_family(d)) return d.taps == 9 ? 1 : 2;
    if (transposed_family(d)) return d.taps == 9 ? 3 : 4;
    return 0;
}

extern "C" int32_t uclstm_pack_job_init(uclstm_pack_job* job, const uclstm_pack_desc* d, const float* w, void* wp, int32_t block0) {
    if (!job || !desc_ok(d) || !w || !wp || block0 < 0) return UCLSTM_E_BADARG;
    job->d = *d;
    job->w = w;
    job->wp = wp;
    job->block0 = block0;
    job->family = pack_family(*d);
    if (job->family == 1 || job->family == 2) {
        job->gx = (d->kseg[0] + 255) / 256 + (d->kseg[1] + 255) / 256;
        job->nblocks = job->gx * d->N;
    } else if (job->family >= 3) {
        job->gx = (d->kseg[0] + d->kseg[1] + 63) / 64;
        job->nblocks = job->gx * ((d->N + 15) / 16);
    } else {
        job->gx = grid_for((int64_t)d->N * d->Ktot);
        job->nblocks = job->gx;
    }
    const PackDiv v = make_pack_div(*d);
    const FastDiv fs[5] = {v.ktot, v.per_tap, v.n_cp, v.k_hdp, v.k_hd};
    for (int i = 0; i < 5; ++i) {
        job->div[3 * i] = fs[i].magic;
        job->div[3 * i + 1] = fs[i].shift;
        job->div[3 * i + 2] = fs[i].d;
    }
    job->pad_ = 0;
    return job->family;
}
#endif

extern "C" int32_t uclstm_pack_weights_batched(const uclstm_pack_job* jobs_dev, int32_t njobs, int32_t family, int32_t total_blocks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || family < 0 || family > 4) return UCLSTM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    switch (family) {
        case 1: UCLSTM_LAUNCH(pack_rows_batched_kernel<9>, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs); break;
        case 2: UCLSTM_LAUNCH(pack_rows_batched_kernel<4>, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs); break;
        case 3: UCLSTM_LAUNCH(pack_transposed_batched_kernel<9>, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs); break;
        case 4: UCLSTM_LAUNCH(pack_transposed_batched_kernel<4>, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs); break;
        default: UCLSTM_LAUNCH(pack_generic_batched_kernel, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs); break;
    }
    return UCLSTM_OK;
}

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_unpack_wgrad(const uclstm_pack_desc* d, const float* dwp, int32_t nslab, int64_t slab, float* grad,
                                       int32_t accumulate, void* stream) {
    if (!desc_ok(d) || !dwp || !grad || nslab < 1 || (nslab > 1 && slab < (int64_t)d->N * d->Ktot)) return UCLSTM_E_BADARG;
    if (rows_family(*d) && nslab > 64 && (slab % 4) == 0 && ((int64_t)d->N * d->Ktot % 4) == 0 && ((uintptr_t)dwp % 16) == 0) {
        // hundreds of small slabs: fold them into slab 0 first (the slabs are scratch of the weight-gradient GEMM and are
        // CONSUMED here), then unpack a single slab
        const int64_t quads = (int64_t)d->N * d->Ktot / 4;
        UCLSTM_LAUNCH(slab_fold_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(dwp), nslab, slab,
                      quads);
        nslab = 1;
    }
    if (rows_family(*d) && nslab <= 64) {
        const int ch0 = (d->kseg[0] + 255) / 256, ch1 = (d->kseg[1] + 255) / 256;
        const dim3 grid(ch0 + ch1, d->N);
        if (d->taps == 9)
            UCLSTM_LAUNCH((pack_rows_kernel<9, true>), grid, dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), nullptr, nullptr, dwp, nslab,
                          slab, grad, accumulate, ch0);
        else
            UCLSTM_LAUNCH((pack_rows_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), nullptr, nullptr, dwp, nslab,
                          slab, grad, accumulate, ch0);
        return UCLSTM_OK;
    }
    // few elements, many slabs: spread the slabs over grid.y (atomic accumulate) until the launch has ~1024 blocks
    const int gx = grid_for((int64_t)d->N * d->Ktot);
    int gy = 1;
    if (accumulate && nslab >= 16 && gx < 512) {
        gy = 1024 / gx;
        if (gy > nslab / 4) gy = nslab / 4;
        if (gy < 1) gy = 1;
    }
    UCLSTM_LAUNCH(unpack_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), dwp, nslab, slab, grad, accumulate);
    return UCLSTM_OK;
}
#endif

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_pack_bias(const uclstm_pack_desc* d, const float* b, float* bp, void* stream) {
    if (!desc_ok(d) || !b || !bp) return UCLSTM_E_BADARG;
    UCLSTM_LAUNCH(pack_bias_kernel, dim3((d->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, make_pack_div(*d), b, bp);
    return UCLSTM_OK;
}
#endif
