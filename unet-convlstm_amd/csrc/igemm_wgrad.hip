// Weight-gradient implicit GEMM on MFMA (gfx950).
//
//   dWp[n][k] += sum_m dY[m][n] * A[m][k]      reduction over output pixels m
//
// Both operands are pixel-major in HBM (NHWC), i.e. K-strided for the MFMA.  They are staged AS THEY LIE
// ([64 pixels][128 channels] = 256-byte rows) by direct-to-LDS loads and the K-contiguous fragments are produced by
// the gfx950 transposing LDS read ds_read_b64_tr_b16 (guide T10): a 16-lane group reads a 4-pixel x 16-channel block
// and each lane receives the 4 pixels of ITS channel.
//   * k-permutation: the 8 k-values a lane feeds to v_mfma_f32_16x16x32_bf16 are pixels {4q..4q+3} and {16+4q..16+4q+3}
//     of the 32-pixel sub-step (q = lane>>4) for BOTH operands; a sum over k does not see the permutation, and it makes
//     the eight row pieces of each 32-lane half come from eight different pixel rows mod 8;
//   * swizzle: the 32-byte granule g of pixel row r is stored at granule position g ^ (r & 7) -- applied on the SOURCE
//     address of the DMA (rule 21) and on the read -- so those eight 32-byte pieces cover all 64 banks: conflict-free.
// One block owns a 128 (panel rows) x 128 (consecutive columns of the packed K axis, any mix of taps / sources) tile of
// dWp for one of `splits` pixel ranges; per lane the (tap, source, channel) of its 16-byte chunk is fixed for the whole
// loop.  PLAIN instantiation (dY and the sources are dense same-size NHWC tensors: every 3x3 / ConvLSTM weight
// gradient): operand offsets are linear in the pixel index, only the tap-validity needs (y, x).  The tile is added to
// dWp with f32 atomics staged through LDS (256 contiguous bytes per wave instruction).
#include "common.h"

namespace {

constexpr int TN = 128;   // panel rows per tile
constexpr int TC = 128;   // packed-K columns per tile
constexpr int TP = 64;    // pixels per stage
constexpr int PITCH = 256;
constexpr int TILE_BYTES = TP * PITCH;            // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;       // 64 KiB

__device__ uint4 g_wzero_page[2];

struct WDerived {
    int kseg0, kseg1;
    int n_kt, n_nt;
    long M;              // total pixels
    long chunk;          // pixels per split (multiple of 64)
    FastDiv dHW, dW, dPerTap;
};

typedef __attribute__((address_space(3))) short4v* lds_s4p;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p + 16 * PITCH));
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

// addr = ok ? addr : zero (branch-free: the DMA below must stay ONE full-wave instruction)
__device__ __forceinline__ uint64_t select_addr(uint64_t a, uint32_t ok, uint64_t zero) {
    uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
    asm volatile("v_cmp_ne_u32 vcc, 0, %2\n\tv_cndmask_b32 %0, %3, %0, vcc\n\tv_cndmask_b32 %1, %4, %1, vcc"
                 : "+v"(lo), "+v"(hi)
                 : "v"(ok), "v"((uint32_t)zero), "v"((uint32_t)(zero >> 32))
                 : "vcc");
    return ((uint64_t)hi << 32) | lo;
}

template <bool PLAIN>
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
    const int n0 = nt * TN;
    const int kbase = kt * TC;

    const long m_begin = (long)sp * dv.chunk;
    if (m_begin >= dv.M) return;     // block-uniform: trailing split with no pixels
    const long m_end = min(dv.M, m_begin + dv.chunk);
    const int nsteps = (int)((m_end - m_begin + TP - 1) / TP);
    const int HW = d.H * d.W;
    const uint64_t zero_addr = (uint64_t)(const void*)g_wzero_page;

    // ---- staging role: DMA instruction j of wave w fills pixel rows 16*j + 4*w + (lane>>4), chunk position lane&15 ----
    const int lrow0 = 4 * wave + lq;
    const int cp = l15;
    const int sc = 2 * ((cp >> 1) ^ (lrow0 & 7)) + (cp & 1);     // source 16-byte chunk stored at this position

    // dY operand: column chunk -> segment (fixed for the loop)
    const int ny = n0 + sc * 8;
    int sgi = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < d.nseg && ny < d.N && ny >= d.seg[i].n_begin && ny < d.seg[i].n_end) sgi = i;
    const uclstm_seg G = d.seg[sgi < 0 ? 0 : sgi];
    const uint32_t yok = sgi >= 0 ? 1u : 0u;
    const int ycol = G.c_off + (ny - G.n_begin);

    // X operand: packed-K column chunk -> (tap, source, channel) (fixed for the loop)
    const int kx = kbase + sc * 8;
    const int xtap = (int)fdiv((uint32_t)kx, dv.dPerTap);
    const int kr = kx - xtap * (dv.kseg0 + dv.kseg1);
    const int xs_ = kr >= dv.kseg0 ? 1 : 0;
    const int xc = xs_ ? kr - dv.kseg0 : kr;
    const uclstm_src S = d.src[(xs_ < d.nsrc) ? xs_ : 0];
    const uint32_t xok = (kx < d.Ktot && xs_ < d.nsrc && xc < S.C) ? 1u : 0u;
    const int tdy = xtap / d.ktap;
    const int ddy = tdy - d.pad - S.offY;                      // source row = y*scale + ddy
    const int ddx = (xtap - tdy * d.ktap) - d.pad - S.offX;

    // PLAIN: element offsets linear in the pixel index
    const int ystride = G.C, xstride = S.C;
    const int xshift = (ddy * S.Ws + ddx) * S.C + xc;           // (m*C + xshift) for a dense same-size source

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int lstep = 0;
    auto issue_loads = [&](int buf) {
        unsigned char* Y = smem + buf * STAGE_BYTES;
        unsigned char* X = Y + TILE_BYTES;
        const uint32_t mb = (uint32_t)(m_begin + (long)lstep * TP);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t m = mb + lrow0 + 16 * j;
            const uint32_t inr = (long)m < m_end ? 1u : 0u;
            const uint32_t img = fdiv(m, dv.dHW);
            const uint32_t rem = m - img * (uint32_t)HW;
            const int y = (int)fdiv(rem, dv.dW);
            const int x = (int)rem - y * d.W;
            uint64_t ay, ax;
            uint32_t oky = yok & inr, okx = xok & inr;
            if constexpr (PLAIN) {
                ay = (uint64_t)G.ptr + (uint64_t)((long)(int)(m * (uint32_t)ystride + (uint32_t)ycol) * 2);
                ax = (uint64_t)S.ptr + (uint64_t)((long)((int)(m * (uint32_t)xstride) + xshift) * 2);
                okx &= ((unsigned)(y + ddy) < (unsigned)S.Hs && (unsigned)(x + ddx) < (unsigned)S.Ws) ? 1u : 0u;
            } else {
                const int yd = y * G.scale + G.oy, xd = x * G.scale + G.ox;
                oky &= ((unsigned)yd < (unsigned)G.Hd && (unsigned)xd < (unsigned)G.Wd) ? 1u : 0u;
                ay = (uint64_t)G.ptr + (uint64_t)((long)((((int)img * G.Hd + yd) * G.Wd + xd) * G.C + ycol) * 2);
                const int ys = y * d.scale + ddy, xs = x * d.scale + ddx;
                okx &= ((unsigned)ys < (unsigned)S.Hs && (unsigned)xs < (unsigned)S.Ws) ? 1u : 0u;
                ax = (uint64_t)S.ptr + (uint64_t)((long)((((int)img * S.Hs + ys) * S.Ws + xs) * S.C + xc) * 2);
            }
            ay = select_addr(ay, oky, zero_addr);
            ax = select_addr(ax, okx, zero_addr);
            const int ldso = (16 * j + 4 * wave) * PITCH;      // wave-uniform; the DMA adds lane*16
            __builtin_amdgcn_global_load_lds((gbl_ptr)ay, (lds_ptr)(Y + ldso), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr)ax, (lds_ptr)(X + ldso), 16, 0, 0);
        }
        ++lstep;
    };

    // transposing fragment reads: lane's row inside a 4-pixel block and its swizzled granule per column block
    const int trow = 4 * lq + (l15 >> 2);                       // pixel row (mod 32 sub-step), also + 16 for the hi half
    const int rsw = trow & 7;
    const int tr_base = trow * PITCH + (l15 & 3) * 8;
    int goffy[4], goffx[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        goffy[a] = (((wc * 4 + a) ^ rsw) << 5) + tr_base;
        goffx[a] = (((wk * 4 + a) ^ rsw) << 5) + tr_base;
    }
    auto compute = [&](int buf) {
        const unsigned char* Y = smem + buf * STAGE_BYTES;
        const unsigned char* X = Y + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 yf[4], xf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) yf[a] = tr_frag(Y + ks * 32 * PITCH + goffy[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) xf[b] = tr_frag(X + ks * 32 * PITCH + goffx[b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
        }
    };

    issue_loads(0);
    __syncthreads();                 // drains the DMA (vmcnt(0)) and orders it before the reads
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) issue_loads((step + 1) & 1);
        compute(step & 1);
        __syncthreads();
    }

    // ---- accumulate the tile into dWp: LDS-staged, 64 consecutive floats of one panel row per wave instruction ----
    constexpr int AP = TC + 4;
    float* At = (float*)smem;                // [64 panel rows][AP]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wc == half) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) At[(a * 16 + lq * 4 + r) * AP + wk * 64 + b * 16 + l15] = acc[a][b][r];
        }
        __syncthreads();
        const int k = kbase + (tid & 127);
        if (k < d.Ktot) {
            for (int pr = tid >> 7; pr < 64; pr += 2) {
                const int n = n0 + half * 64 + pr;
                if (n < d.N)
                    __hip_atomic_fetch_add(d.dwp + (long)n * d.Ktot + k, At[pr * AP + (tid & 127)], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
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
    bool plain = d.scale == 1;
    for (int s = 0; s < d.nsrc; ++s) {
        if (!wsrc_ok(d.src[s])) return UCLSTM_E_BADARG;
        if ((int64_t)d.n_img * d.src[s].Hs * d.src[s].Ws * d.src[s].C >= ((int64_t)1 << 31) - (1 << 20)) return UCLSTM_E_BADARG;
        plain = plain && d.src[s].Hs == d.H && d.src[s].Ws == d.W && d.src[s].offY == 0 && d.src[s].offX == 0;
    }
    for (int i = 0; i < d.nseg; ++i) {
        const uclstm_seg& sg = d.seg[i];
        if (!sg.ptr || (sg.n_begin % 8) || (sg.n_end % 8) || sg.n_end <= sg.n_begin || (sg.C % 8) || (sg.c_off % 8) ||
            sg.c_off + (sg.n_end - sg.n_begin) > sg.C || sg.Hd <= 0 || sg.Wd <= 0 || sg.scale < 1 || ((uintptr_t)sg.ptr % 16))
            return UCLSTM_E_BADARG;
        if ((int64_t)d.n_img * sg.Hd * sg.Wd * sg.C >= ((int64_t)1 << 31) - (1 << 20)) return UCLSTM_E_BADARG;
        plain = plain && sg.scale == 1 && sg.oy == 0 && sg.ox == 0 && sg.Hd == d.H && sg.Wd == d.W;
    }
    WDerived dv;
    dv.kseg0 = round_up32(d.src[0].C, 64);
    dv.kseg1 = d.nsrc > 1 ? round_up32(d.src[1].C, 64) : 0;
    const int taps = d.ktap * d.ktap;
    if (d.Ktot != taps * (dv.kseg0 + dv.kseg1)) return UCLSTM_E_BADARG;
    dv.n_kt = (d.Ktot + TC - 1) / TC;
    dv.n_nt = (d.N + TN - 1) / TN;
    dv.M = (long)d.n_img * d.H * d.W;
    if (dv.M >= ((long)1 << 31) - 4096) return UCLSTM_E_BADARG;
    long chunk = (dv.M + d.splits - 1) / d.splits;
    chunk = (chunk + TP - 1) / TP * TP;
    dv.chunk = chunk;
    dv.dHW = make_fastdiv((uint32_t)(d.H * d.W));
    dv.dW = make_fastdiv((uint32_t)d.W);
    dv.dPerTap = make_fastdiv((uint32_t)(dv.kseg0 + dv.kseg1));
    const int64_t nblk = (int64_t)dv.n_kt * dv.n_nt * d.splits;
    if (nblk <= 0 || nblk > 0x7fffffff) return UCLSTM_E_BADARG;

    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_done = true;
    }
    if (plain)
        UCLSTM_LAUNCH(igemm_wgrad_kernel<true>, dim3((unsigned)nblk), dim3(256), SMEM_BYTES, (hipStream_t)stream, d, dv);
    else
        UCLSTM_LAUNCH(igemm_wgrad_kernel<false>, dim3((unsigned)nblk), dim3(256), SMEM_BYTES, (hipStream_t)stream, d, dv);
    return UCLSTM_OK;
}
