// Weight-gradient implicit GEMM on MFMA (gfx950).
//
//   dWp[n][k] += sum_m dY[m][n] * A[m][k]      reduction over output pixels m
//
// Both operands are pixel-major in HBM (NHWC), i.e. K-strided for the MFMA.  They are staged
// as they lie ([64 pixels][128 channels], 288-byte pitch) and the K-contiguous fragments are
// produced by the gfx950 transposing LDS read ds_read_b64_tr_b16 (guide T10): a 16-lane group
// reads a 4-pixel x 16-channel block and each lane receives the 4 pixels of ITS channel.
// The 8 k-values a lane feeds to v_mfma_f32_16x16x32_bf16 are pixels {4q..4q+3} and
// {16+4q..16+4q+3} of the 32-pixel sub-step (q = lane>>4) for BOTH operands -- a permutation
// of k, which a sum over k does not see -- so that with the 288-byte pitch (72 dwords = 8 mod
// 64 banks) the eight 32-byte row pieces of each 32-lane half land on distinct banks.
// One block owns a 128 (panel rows) x 128 (channels of one (tap, source) K segment) tile of
// dWp for one of `splits` pixel ranges and adds it with f32 atomics (256-byte runs per wave
// instruction; MI355X_MICROARCH "Global float atomics").
#include "common.h"

namespace {

constexpr int TN = 128;   // panel rows per tile
constexpr int TC = 128;   // K columns (channels) per tile
constexpr int TP = 64;    // pixels per stage
constexpr int PITCH = 288;
constexpr int TILE_BYTES = TP * PITCH;            // 18432
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;       // 73728

struct WDerived {
    int kseg0, kseg1;
    int nb0, nb1;        // 128-column blocks per source
    int n_kt, n_nt;
    long M;              // total pixels
    long chunk;          // pixels per split (multiple of 64)
    FastDiv dHW, dW;
};

typedef __attribute__((address_space(3))) short4v* lds_s4p;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base) {
    // two transposing reads, 16 pixel rows apart
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(base));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(base + 16 * PITCH));
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void igemm_wgrad_kernel(const uclstm_wgrad_desc d, const WDerived dv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1;     // along panel rows
    const int wk = wave & 1;      // along K columns
    const int l15 = lane & 15;
    const int lq = lane >> 4;

    const int per_split = dv.n_kt * dv.n_nt;
    const int lid = xcd_remap(blockIdx.x, per_split * d.splits);
    const int sp = lid / per_split;
    const int rr = lid - sp * per_split;
    const int kt = rr / dv.n_nt;
    const int nt = rr - kt * dv.n_nt;
    const int per_tap = dv.nb0 + dv.nb1;
    const int tap = kt / per_tap;
    const int kr = kt - tap * per_tap;
    const int s = kr >= dv.nb0 ? 1 : 0;
    const int c_base = (s ? kr - dv.nb0 : kr) * TC;
    const int n0 = nt * TN;
    const uclstm_src S = d.src[s];
    const int tdy = tap / d.ktap;
    const int dy = tdy - d.pad - S.offY;
    const int dx = (tap - tdy * d.ktap) - d.pad - S.offX;

    const long m_begin = (long)sp * dv.chunk;
    if (m_begin >= dv.M) return;     // block-uniform: trailing split with no pixels
    const long m_end = min(dv.M, m_begin + dv.chunk);
    const int nsteps = (int)((m_end - m_begin + TP - 1) / TP);
    const int HW = d.H * d.W;

    // staging role: 16-byte chunk column lch of rows lrow0 + 16*i
    const int lch = tid & 15;
    const int lrow0 = tid >> 4;
    // dY column chunk -> segment (fixed for the whole loop)
    const int ny = n0 + lch * 8;
    int sgi = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < d.nseg && ny < d.N && ny >= d.seg[i].n_begin && ny < d.seg[i].n_end) sgi = i;
    const uclstm_seg G = d.seg[sgi < 0 ? 0 : sgi];
    const bool yvalid = sgi >= 0;
    const int ycol = G.c_off + (ny - G.n_begin);
    const int xc = c_base + lch * 8;
    const bool xvalid = xc < S.C;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ry[4], rx[4];
    int lstep = 0;
    auto issue_loads = [&]() {
        const long mb = m_begin + (long)lstep * TP;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = mb + lrow0 + 16 * i;
            ry[i] = make_uint4(0u, 0u, 0u, 0u);
            rx[i] = make_uint4(0u, 0u, 0u, 0u);
            if (m < m_end) {
                const uint32_t mu = (uint32_t)m;
                const uint32_t img = fdiv(mu, dv.dHW);
                const uint32_t rem = mu - img * (uint32_t)HW;
                const int y = (int)fdiv(rem, dv.dW);
                const int x = (int)rem - y * d.W;
                if (yvalid) {
                    const int yd = y * G.scale + G.oy, xd = x * G.scale + G.ox;
                    if ((unsigned)yd < (unsigned)G.Hd && (unsigned)xd < (unsigned)G.Wd)
                        ry[i] = *(const uint4*)((const bf16*)G.ptr + (((long)img * G.Hd + yd) * G.Wd + xd) * (long)G.C + ycol);
                }
                if (xvalid) {
                    const int ys = y * d.scale + dy, xs = x * d.scale + dx;
                    if ((unsigned)ys < (unsigned)S.Hs && (unsigned)xs < (unsigned)S.Ws)
                        rx[i] = *(const uint4*)((const bf16*)S.ptr + (((long)img * S.Hs + ys) * S.Ws + xs) * (long)S.C + xc);
                }
            }
        }
        ++lstep;
    };
    auto stage_store = [&](int buf) {
        unsigned char* Y = smem + buf * STAGE_BYTES;
        unsigned char* X = Y + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (lrow0 + 16 * i) * PITCH + lch * 16;
            *(uint4*)(Y + off) = ry[i];
            *(uint4*)(X + off) = rx[i];
        }
    };
    // lane's address inside a 4-pixel x 16-channel block: pixel row (l15>>2), channels 4*(l15&3)
    const int tr_off = (4 * lq + (l15 >> 2)) * PITCH + (l15 & 3) * 8;
    auto compute = [&](int buf) {
        const unsigned char* Y = smem + buf * STAGE_BYTES + tr_off;
        const unsigned char* X = Y + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 yf[4], xf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) yf[a] = tr_frag(Y + ks * 32 * PITCH + (wc * 64 + a * 16) * 2);
#pragma unroll
            for (int b = 0; b < 4; ++b) xf[b] = tr_frag(X + ks * 32 * PITCH + (wk * 64 + b * 16) * 2);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        issue_loads();
        stage_store(0);
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const bool more = step + 1 < nsteps;
            if (more) issue_loads();
            compute(step & 1);
            if (more) stage_store((step + 1) & 1);
            __syncthreads();
        }
    }

    // ---- accumulate the tile: lane owns panel rows n..n+3 (registers) of K column kcol ----
    const int kseg = s ? dv.kseg1 : dv.kseg0;
    const long koff = (long)tap * (dv.kseg0 + dv.kseg1) + (s ? dv.kseg0 : 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int kcol = c_base + wk * 64 + b * 16 + l15;
        if (kcol >= kseg) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + wc * 64 + a * 16 + lq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < d.N) {
                    float* dst = d.dwp + (long)(n + r) * d.Ktot + koff + kcol;
                    __hip_atomic_fetch_add(dst, acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

bool wsrc_ok(const uclstm_src& s) {
    return s.ptr && s.C > 0 && (s.C % 8) == 0 && s.Hs > 0 && s.Ws > 0 && ((uintptr_t)s.ptr % 16) == 0;
}

}  // namespace

extern "C" int32_t uclstm_igemm_wgrad(const uclstm_wgrad_desc* dp, void* stream) {
    if (!dp) return UCLSTM_E_BADARG;
    const uclstm_wgrad_desc& d = *dp;
    if (d.n_img <= 0 || d.H <= 0 || d.W <= 0) return UCLSTM_E_BADARG;
    if (d.ktap < 1 || d.ktap > 3 || d.scale < 1 || d.scale > 2 || d.pad < 0 || d.pad > 1) return UCLSTM_E_BADARG;
    if (d.nsrc < 1 || d.nsrc > 2 || !d.dwp || d.N <= 0 || (d.N % 8) || d.splits < 1) return UCLSTM_E_BADARG;
    if (d.nseg < 1 || d.nseg > 4) return UCLSTM_E_BADARG;
    for (int s = 0; s < d.nsrc; ++s)
        if (!wsrc_ok(d.src[s])) return UCLSTM_E_BADARG;
    for (int i = 0; i < d.nseg; ++i) {
        const uclstm_seg& sg = d.seg[i];
        if (!sg.ptr || (sg.n_begin % 8) || (sg.n_end % 8) || sg.n_end <= sg.n_begin || (sg.C % 8) || (sg.c_off % 8) ||
            sg.c_off + (sg.n_end - sg.n_begin) > sg.C || sg.Hd <= 0 || sg.Wd <= 0 || sg.scale < 1 || ((uintptr_t)sg.ptr % 16))
            return UCLSTM_E_BADARG;
    }
    WDerived dv;
    dv.kseg0 = round_up32(d.src[0].C, 64);
    dv.kseg1 = d.nsrc > 1 ? round_up32(d.src[1].C, 64) : 0;
    const int taps = d.ktap * d.ktap;
    if (d.Ktot != taps * (dv.kseg0 + dv.kseg1)) return UCLSTM_E_BADARG;
    dv.nb0 = (d.src[0].C + TC - 1) / TC;
    dv.nb1 = d.nsrc > 1 ? (d.src[1].C + TC - 1) / TC : 0;
    dv.n_kt = taps * (dv.nb0 + dv.nb1);
    dv.n_nt = (d.N + TN - 1) / TN;
    dv.M = (long)d.n_img * d.H * d.W;
    if (dv.M >= ((long)1 << 31)) return UCLSTM_E_BADARG;
    long chunk = (dv.M + d.splits - 1) / d.splits;
    chunk = (chunk + TP - 1) / TP * TP;
    dv.chunk = chunk;
    dv.dHW = make_fastdiv((uint32_t)(d.H * d.W));
    dv.dW = make_fastdiv((uint32_t)d.W);
    const int64_t nblk = (int64_t)dv.n_kt * dv.n_nt * d.splits;
    if (nblk <= 0 || nblk > 0x7fffffff) return UCLSTM_E_BADARG;

    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_done = true;
    }
    UCLSTM_LAUNCH(igemm_wgrad_kernel, dim3((unsigned)nblk), dim3(256), SMEM_BYTES, (hipStream_t)stream, d, dv);
    return UCLSTM_OK;
}
