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
#include <cstdlib>

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
    int kt_per_tap;      // K-column tiles per tap when a tap is a whole number of tiles, else 0 (no tap-minor order)
    FastDiv dHW, dW, dPerTap;
};

typedef __attribute__((address_space(3))) short4v* lds_s4p;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

__device__ __forceinline__ act16x8 tr_frag(const unsigned char* p) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p + 16 * PITCH));
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(act16x8, v);
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
            act16x8 yf[4], xf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) yf[a] = tr_frag(Y + ks * 32 * PITCH + goffy[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) xf[b] = tr_frag(X + ks * 32 * PITCH + goffx[b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = UCLSTM_MFMA_16x16x32(yf[a], xf[b], acc[a][b], 0, 0, 0);
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
                    {
                    float* dst = d.dwp + (long)sp * d.slab + (long)n * d.Ktot + k;
                    if (d.slab > 0) *dst = At[pr * AP + (tid & 127)];        // this pixel range's own slab (added up by the unpack kernel)
                    else __hip_atomic_fetch_add(dst, At[pr * AP + (tid & 127)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Fast path: dense same-size operands (PLAIN), H and W powers of two, one dY segment, 3x3/pad 1 or 1x1/pad 0.
//
// Staging goes through BUFFER descriptors (buffer_load_dwordx4 ... lds).  Measured on gfx950 (tools/probes/): a lane
// whose voffset + soffset is >= num_records writes 16 ZERO bytes into LDS.  So
//   * a padded tap is one v_cndmask (offset or 0x80000000) instead of a 64-bit select against a zero page;
//   * the per-lane voffset is loop invariant: the pixel range advances through the SGPR soffset, no address VALU at all;
//   * a lane whose (tap, channel) column or dY column does not exist carries 0x80000000 for the whole loop.
// The X descriptor starts (W+1) pixels BEFORE the tensor so that voffset is never negative (those bytes are never
// fetched: every tap that would reach them is masked).  Tap validity without division: a stage is 64 consecutive pixels
// starting at a multiple of 64, so with W, H powers of two the row r of the stage has
//     x = (mb & (W-1)) + (r & (W-1)),    y = ((mb >> lw) + (r >> lw)) & (H-1)
// and "x + ddx outside" <=> x == one bad value: both tests become a compare of a per-stage SCALAR with a per-lane
// constant (3 VALU per staged row instead of ~35 in the generic kernel, which was VALU-issue bound at 4.3 VALU/MFMA).
//
// LDS image: every operand tile is a set of PLANES of [64 pixels][64 columns] (128-byte rows, 8 KiB).  One DMA
// instruction fills 8 rows of one plane, so all its lanes share one 64-column segment = one (tap, source): the
// descriptor is wave-uniform.  32-byte granule g of row r sits at position g ^ ((r>>1)&3); two 128-byte rows share a
// 256-byte bank row, so the eight row pieces of a transposing read (rows r..r+7) cover all 64 banks.
// Tile shapes: WN = 2: 128 panel rows x 128 K columns (waves 2x2);  WN = 1: 64 x 256 (waves 1x4) for C_out <= 64;
// WN = 3: 64 x 192 (waves 1x4, 48 columns = three MFMA column tiles each) for C_out <= 64 when 192 | Ktot -- the K = 576 / 1152
// layers of the full-resolution level, where 256-column tiles are 25 % / 10 % padding.
struct P2 {
    int lw, lh;            // log2 W, log2 H
    uint32_t ybytes;       // dY tensor bytes
    uint32_t xbytes[2];    // source bytes + bias
    uint32_t xbias[2];     // bytes the descriptor base lies before the tensor
};

constexpr int PLANE = TP * 128;          // 8 KiB
constexpr uint32_t OOB = 0x80000000u;

template <int WN>
struct WShape {
    static constexpr int WNR = WN == 2 ? 2 : 1;               // waves (= dY planes) along panel rows
    static constexpr int WK = 4 / WNR;                        // waves along K columns
    static constexpr int XP = WN == 3 ? 3 : WK;               // X planes (64 K columns each) per stage
    static constexpr int NB = WN == 3 ? 3 : 4;                // MFMA column tiles (16 columns) per wave
    static constexpr int WCOLS = 16 * NB;                     // K columns per wave
    static constexpr int TN_ = 64 * WNR, TC_ = 64 * XP;
    static constexpr int NPL = WNR + XP;                      // planes per stage (dY planes first)
    static constexpr int STAGE = NPL * PLANE;                 // 32 KiB / 40 KiB
    static constexpr int SMEM = 2 * STAGE;
    static constexpr int NI = 2 * NPL;                        // DMA instructions per wave per stage (8 per plane / 4 waves)
};

__device__ __forceinline__ act16x8 tr_frag128(const unsigned char* p) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p + 16 * 128));
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(act16x8, v);
}

template <int WN, int NSRC>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_p2_kernel(const uclstm_wgrad_desc d, const WDerived dv, const P2 p2) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-resource type does not exist in the host pass (the stub needs no body)
    using SH = WShape<WN>;
    constexpr int WK = SH::WK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WK;     // along panel rows
    const int wk = wave % WK;     // along K columns
    const int l15 = lane & 15;
    const int lq = lane >> 4;

    const int per_split = dv.n_kt * dv.n_nt;
    const int lid = xcd_remap(blockIdx.x, per_split * d.splits);
    const int sp = lid / per_split;
    const int rr = lid - sp * per_split;
    // resident blocks of an XCD = 8 panel-row tiles x 8 K-column tiles; the K-column tiles are visited TAP-MINOR (the taps
    // of one channel range are neighbours and read the same activation lines, shifted)
    int nt, kts;
    grouped_tile(rr, dv.n_nt, dv.n_kt, 8, nt, kts);
    int kt = kts;
    if (dv.kt_per_tap > 0) {
        const int ntaps = d.ktap * d.ktap;
        const int cg = kts / ntaps;
        kt = (kts - cg * ntaps) * dv.kt_per_tap + cg;
    }
    const int n0 = nt * SH::TN_;
    const int kbase = kt * SH::TC_;

    const long m_begin = (long)sp * dv.chunk;
    if (m_begin >= dv.M) return;     // block-uniform: trailing split with no pixels
    const long m_end = min(dv.M, m_begin + dv.chunk);
    const int nsteps = (int)((m_end - m_begin) / TP);        // M % 64 == 0 on this path

    const uclstm_seg G = d.seg[0];
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)G.ptr, 0, p2.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)d.src[0].ptr - p2.xbias[0]), 0, p2.xbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx1 =
        NSRC > 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)d.src[1].ptr - p2.xbias[1]), 0, p2.xbytes[1], 0x00020000)
                 : rsx0;

    // ---- staging role: instruction j of wave w = row group 4*(j&1) + w of plane j>>1; lane = (row lane>>3, position lane&7)
    const int lrow = lane >> 3;
    const int cp = lane & 7;
    const int sc = 2 * ((cp >> 1) ^ ((lrow >> 1) & 3)) + (cp & 1);      // source 16-byte chunk stored at this position
    const int Wm = d.W - 1, Hm = d.H - 1;
    const int kseg = dv.kseg0 + dv.kseg1;

    uint32_t voff[SH::NI];            // loop-invariant byte offsets (dY planes, then X planes)
    uint32_t kxj[2 * SH::XP], kyj[2 * SH::XP];
    int xsrc[SH::XP];                 // wave-uniform source of each X plane
#pragma unroll
    for (int j = 0; j < SH::NI; ++j) {
        const int pl = j >> 1;
        const int r = 8 * (4 * (j & 1) + wave) + lrow;        // pixel row of the stage
        if (pl < SH::WNR) {
            const int ny = n0 + pl * 64 + sc * 8;
            const bool yok = ny < d.N && ny >= G.n_begin && ny < G.n_end;
            voff[j] = yok ? (uint32_t)(2 * (r * G.C + G.c_off + (ny - G.n_begin))) : OOB;
        } else {
            const int xp = pl - SH::WNR;
            const int k0 = kbase + xp * 64;                   // first column of the plane: one (tap, source)
            const int xtap = (int)fdiv((uint32_t)k0, dv.dPerTap);
            const int kr = k0 - xtap * kseg;
            const int xs_ = (NSRC > 1 && kr >= dv.kseg0) ? 1 : 0;
            const int xc = (xs_ ? kr - dv.kseg0 : kr) + sc * 8;
            const uclstm_src S = d.src[xs_];
            const bool xok = k0 < d.Ktot && xc < S.C;
            const int tdy = xtap / d.ktap;
            const int ddy = tdy - d.pad;
            const int ddx = (xtap - tdy * d.ktap) - d.pad;
            const int badx = ddx < 0 ? 0 : (ddx > 0 ? Wm : -1);
            const int bady = ddy < 0 ? 0 : (ddy > 0 ? Hm : -1);
            if ((j & 1) == 0) xsrc[xp] = xs_;
            voff[j] = xok ? (uint32_t)(2 * (r * S.C + (ddy * S.Ws + ddx) * S.C + xc) + (int)p2.xbias[xs_]) : OOB;
            // bad column <=> (mb & (W-1)) == kxj ; bad row <=> ((mb >> lw) & (H-1)) == kyj   (0xffffffff: never)
            const int jj = j - 2 * SH::WNR;
            if (badx < 0) kxj[jj] = 0xffffffffu;
            else if (d.W >= 64) kxj[jj] = (uint32_t)(badx - r);        // equal only if a non-negative multiple of 64
            else kxj[jj] = ((r & Wm) == badx) ? 0u : 0xffffffffu;
            kyj[jj] = bady < 0 ? 0xffffffffu : (uint32_t)((bady - (r >> p2.lw)) & Hm);
        }
    }
    const uint32_t ystep = (uint32_t)(2 * G.C), x0step = (uint32_t)(2 * d.src[0].C), x1step = (uint32_t)(2 * d.src[NSRC - 1].C);

    constexpr int NB = SH::NB;
    f32x4 acc[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint32_t mb = (uint32_t)m_begin;      // first pixel of the NEXT stage to load (wave-uniform)
    auto issue_loads = [&](int buf) {
        unsigned char* St = smem + buf * SH::STAGE;
        const uint32_t sy = (mb >> p2.lw) & (uint32_t)Hm;
        const uint32_t sx = mb & (uint32_t)Wm;
        const uint32_t ysoff = mb * ystep, x0soff = mb * x0step, x1soff = mb * x1step;
#pragma unroll
        for (int j = 0; j < SH::NI; ++j) {
            lds_ptr dst = (lds_ptr)(St + (j >> 1) * PLANE + (4 * (j & 1) + wave) * 1024);
            if (j < 2 * SH::WNR) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, dst, 16, voff[j], ysoff, 0, 0);
            } else {
                const int jj = j - 2 * SH::WNR;
                const bool ok = (sx != kxj[jj]) & (sy != kyj[jj]);
                const uint32_t off = ok ? voff[j] : OOB;
                if (NSRC == 1 || xsrc[jj >> 1] == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx0, dst, 16, off, x0soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx1, dst, 16, off, x1soff, 0, 0);
            }
        }
        mb += TP;
    };

    // transposing fragment reads (k-permutation as in the generic kernel): same offsets inside every plane
    const int trow = 4 * lq + (l15 >> 2);
    int goff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) goff[a] = trow * 128 + ((a ^ ((trow >> 1) & 3)) << 5) + (l15 & 3) * 8;
    // this wave's K-column tiles: tile ct = wk * NB + b (16 columns each) lives in X plane ct / 4, granule ct % 4
    int xoff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ct = wk * NB + b;
        xoff[b] = (SH::WNR + (ct >> 2)) * PLANE + trow * 128 + (((ct & 3) ^ ((trow >> 1) & 3)) << 5) + (l15 & 3) * 8;
    }
    auto compute = [&](int buf) {
        const unsigned char* St = smem + buf * SH::STAGE;
        const unsigned char* Y = St + wc * PLANE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            act16x8 yf[4], xf[NB];
#pragma unroll
            for (int a = 0; a < 4; ++a) yf[a] = tr_frag128(Y + ks * 32 * 128 + goff[a]);
#pragma unroll
            for (int b = 0; b < NB; ++b) xf[b] = tr_frag128(St + ks * 32 * 128 + xoff[b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = UCLSTM_MFMA_16x16x32(yf[a], xf[b], acc[a][b], 0, 0, 0);
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
    constexpr int TCc = SH::TC_;
    constexpr int AP = TCc + 4;
    float* At = (float*)smem;                // [64 panel rows][AP]
#pragma unroll
    for (int half = 0; half < SH::WNR; ++half) {
        if (wc == half) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) At[(a * 16 + lq * 4 + r) * AP + wk * SH::WCOLS + b * 16 + l15] = acc[a][b][r];
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * TCc; idx += 256) {          // consecutive threads = consecutive K columns of one panel row
            const int pr = idx / TCc, col = idx - pr * TCc;
            const int k = kbase + col;
            const int n = n0 + half * 64 + pr;
            if (k < d.Ktot && n < d.N) {
                float* dst = d.dwp + (long)sp * d.slab + (long)n * d.Ktot + k;
                if (d.slab > 0) *dst = At[pr * AP + col];        // this pixel range's own slab (added up by the unpack kernel)
                else __hip_atomic_fetch_add(dst, At[pr * AP + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves, "8-phase" schedule (guide section 5; tools/probes/gemm256_8phase.hip reaches 1.15-1.25 PFLOP/s
// with this structure on a plain GEMM where the 128x128 two-barrier loop plateaus near 0.9).  Same operand handling as
// the fast path above (planes of [64 pixels][64 columns], buffer-addressed zero-filling DMA, scalar-vs-constant tap
// validity); what changes is the pipeline:
//   * a K-tile (64 pixels) is staged as four HALF-tiles of two planes (16 KiB): dY-lo, X-lo, X-hi, dY-hi, in the order the
//     phases consume them; every phase issues one half-tile LEAD (7, or 6) halves ahead and waits with a COUNTED vmcnt, so
//     LEAD - 2 half-tiles stay in flight across the raw s_barriers (never vmcnt(0) in the steady state);
//   * a wave owns 128 x 64 of the tile as four 64 x 32 quadrants, one quadrant (16 MFMA) per phase; its rows/columns are
//     interleaved over the halves so that each half is read in exactly one phase: dY-lo + X-lo in phase 0, X-hi in 1,
//     dY-hi in 2, none in 3 -> with LEAD 6 a half is re-staged at least two phases after its last read (write-after-read
//     safe with the stagger below); with LEAD 7 dY-lo is re-staged ONE phase after its reads, which are therefore retired
//     (lgkmcnt(0)) before the phase-0 barrier;
//   * waves 4-7 run one barrier behind waves 0-3 (they share the SIMDs pairwise): one group's MFMA section overlaps the
//     other group's LDS reads + DMA issue.
// One block per CU (128 KiB LDS).  Used when C_out >= 256; pixel splits keep the grid a multiple of the 256 CUs.
constexpr int P3_HALF = 2 * PLANE;            // 16 KiB
constexpr int P3_BUF = 4 * P3_HALF;           // 64 KiB
constexpr int P3_SMEM = 2 * P3_BUF;           // 128 KiB

template <int NSRC, int P3_LEAD>
__global__ __launch_bounds__(512, 1) void igemm_wgrad_p3_kernel(const uclstm_wgrad_desc d, const WDerived dv, const P2 p2) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;     // along panel rows (also the stagger group)
    const int wc = wave & 3;      // along K columns
    const int l15 = lane & 15;
    const int lq = lane >> 4;

    const int per_split = dv.n_kt * dv.n_nt;
    const int lid = xcd_remap(blockIdx.x, per_split * d.splits);
    const int sp = lid / per_split;
    const int rr = lid - sp * per_split;
    int nt, kts;
    grouped_tile(rr, dv.n_nt, dv.n_kt, 4, nt, kts);
    int kt = kts;
    if (dv.kt_per_tap > 0) {
        const int ntaps = d.ktap * d.ktap;
        const int cg = kts / ntaps;
        kt = (kts - cg * ntaps) * dv.kt_per_tap + cg;
    }
    const int n0 = nt * 256;
    const int kbase = kt * 256;

    const long m_begin = (long)sp * dv.chunk;
    const long m_end = min(dv.M, m_begin + dv.chunk);
    const int KT = m_begin < dv.M ? (int)((m_end - m_begin) / TP) : 0;      // block-uniform; 0: nothing to do (but every
    if (KT == 0) return;                                                      // wave of the block leaves together)
    const int total_halves = 4 * KT;

    const uclstm_seg G = d.seg[0];
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)G.ptr, 0, p2.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)d.src[0].ptr - p2.xbias[0]), 0, p2.xbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx1 =
        NSRC > 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)d.src[1].ptr - p2.xbias[1]), 0, p2.xbytes[1], 0x00020000)
                 : rsx0;

    // ---- staging role: DMA instruction i of a half fills rows 8*wave + (lane>>3) of its plane i ----
    const int lrow = lane >> 3;
    const int cp = lane & 7;
    const int sc = 2 * ((cp >> 1) ^ ((lrow >> 1) & 3)) + (cp & 1);      // source 16-byte chunk stored at this position
    const int r = 8 * wave + lrow;                                       // pixel row of the stage
    const int Wm = d.W - 1, Hm = d.H - 1;
    const int kseg = dv.kseg0 + dv.kseg1;
    uint32_t yvoff[4], xvoff[4], kxj[4], kyj[4];
    int xsrc[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
        const int ny = n0 + pl * 64 + sc * 8;
        const bool yok = ny < d.N && ny >= G.n_begin && ny < G.n_end;
        yvoff[pl] = yok ? (uint32_t)(2 * (r * G.C + G.c_off + (ny - G.n_begin))) : OOB;
        const int k0 = kbase + pl * 64;
        const int xtap = (int)fdiv((uint32_t)k0, dv.dPerTap);
        const int kr = k0 - xtap * kseg;
        const int xs_ = (NSRC > 1 && kr >= dv.kseg0) ? 1 : 0;
        const int xc = (xs_ ? kr - dv.kseg0 : kr) + sc * 8;
        const uclstm_src S = d.src[xs_];
        const bool xok = k0 < d.Ktot && xc < S.C;
        const int tdy = xtap / d.ktap;
        const int ddy = tdy - d.pad;
        const int ddx = (xtap - tdy * d.ktap) - d.pad;
        const int badx = ddx < 0 ? 0 : (ddx > 0 ? Wm : -1);
        const int bady = ddy < 0 ? 0 : (ddy > 0 ? Hm : -1);
        xsrc[pl] = xs_;
        xvoff[pl] = xok ? (uint32_t)(2 * (r * S.C + (ddy * S.Ws + ddx) * S.C + xc) + (int)p2.xbias[xs_]) : OOB;
        if (badx < 0) kxj[pl] = 0xffffffffu;
        else if (d.W >= 64) kxj[pl] = (uint32_t)(badx - r);
        else kxj[pl] = ((r & Wm) == badx) ? 0u : 0xffffffffu;
        kyj[pl] = bady < 0 ? 0xffffffffu : (uint32_t)((bady - (r >> p2.lw)) & Hm);
    }
    const uint32_t ystep = (uint32_t)(2 * G.C), x0step = (uint32_t)(2 * d.src[0].C), x1step = (uint32_t)(2 * d.src[NSRC - 1].C);
    const uint32_t mb0 = (uint32_t)m_begin;

    // half-tile `slot` of K-tile `kt_`: 0 dY planes 0-1, 1 X planes 0-1, 2 X planes 2-3, 3 dY planes 2-3
    auto issue_half = [&](int kt_, int slot) {
        unsigned char* dst = smem + (kt_ & 1) * P3_BUF + slot * P3_HALF + wave * 1024;
        const uint32_t mb = mb0 + (uint32_t)kt_ * TP;
        if (slot == 0 || slot == 3) {
            const uint32_t ysoff = mb * ystep;
            const int p0 = slot == 0 ? 0 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (lds_ptr)(dst + i * PLANE), 16, slot == 0 ? yvoff[i] : yvoff[2 + i], ysoff, 0, 0);
            (void)p0;
        } else {
            const uint32_t sy = (mb >> p2.lw) & (uint32_t)Hm;
            const uint32_t sx = mb & (uint32_t)Wm;
            const uint32_t x0soff = mb * x0step, x1soff = mb * x1step;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = (slot == 1 ? 0 : 2) + i;
                const bool ok = (sx != kxj[pl]) & (sy != kyj[pl]);
                const uint32_t off = ok ? xvoff[pl] : OOB;
                if (NSRC == 1 || xsrc[pl] == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx0, (lds_ptr)(dst + i * PLANE), 16, off, x0soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx1, (lds_ptr)(dst + i * PLANE), 16, off, x1soff, 0, 0);
            }
        }
    };

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposing fragment reads: offsets inside a plane (granule g of pixel row r at g ^ ((r>>1)&3))
    const int trow = 4 * lq + (l15 >> 2);
    const int tsw = (trow >> 1) & 3;
    int goffA[4], goffB[2];
#pragma unroll
    for (int a = 0; a < 4; ++a) goffA[a] = wr * PLANE + trow * 128 + ((a ^ tsw) << 5) + (l15 & 3) * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) goffB[j] = (wc >> 1) * PLANE + trow * 128 + ((((wc & 1) * 2 + j) ^ tsw) << 5) + (l15 & 3) * 8;

    act16x8 af[4][2], b0f[2][2], b1f[2][2];

    // prologue: six half-tiles in flight, the first two (dY-lo, X-lo of K-tile 0) landed before the first phase
#pragma unroll
    for (int q = 0; q < P3_LEAD; ++q)
        if (q < total_halves) issue_half(q >> 2, q & 3);
    // in flight after a wait: LEAD - 2 half-tiles = 2*(LEAD-2) DMA instructions of this wave
    if (total_halves >= P3_LEAD) {
        if constexpr (P3_LEAD == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();            // stagger: group 1 runs one barrier behind
    __builtin_amdgcn_s_barrier();

    int g = 0;
    for (int ktile = 0; ktile < KT; ++ktile) {
        const unsigned char* base = smem + (ktile & 1) * P3_BUF;
#pragma unroll
        for (int p = 0; p < 4; ++p, ++g) {
            if (p == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) b0f[j][ks] = tr_frag128(base + 1 * P3_HALF + ks * 32 * 128 + goffB[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) af[i][ks] = tr_frag128(base + 0 * P3_HALF + ks * 32 * 128 + goffA[i]);
            } else if (p == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) b1f[j][ks] = tr_frag128(base + 2 * P3_HALF + ks * 32 * 128 + goffB[j]);
            } else if (p == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) af[i][ks] = tr_frag128(base + 3 * P3_HALF + ks * 32 * 128 + goffA[i]);
            }
            if (g + P3_LEAD < total_halves) {
                issue_half(ktile + ((p + P3_LEAD) >> 2), (p + P3_LEAD) & 3);
                if constexpr (P3_LEAD == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // LEAD 7 re-stages dY-lo ONE phase after its reads (phase 0): those reads must have retired before this barrier,
            // because the other wave group runs a barrier behind (guide: "1 phase after when an lgkmcnt retired those reads")
            if (P3_LEAD == 7 && p == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            {
                const int mh = (p >= 2) ? 1 : 0;
                const int nh = (p == 1 || p == 2) ? 1 : 0;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const act16x8 bb = nh ? b1f[j][ks] : b0f[j][ks];
                            acc[mh][i][nh][j] = UCLSTM_MFMA_16x16x32(af[i][ks], bb, acc[mh][i][nh][j], 0, 0, 0);
                        }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();            // balance the stagger: both groups are past their last LDS read
    __syncthreads();

    // ---- accumulate into dWp: four passes of 64 panel rows, LDS-staged so that a wave instruction adds 64 consecutive floats
    constexpr int AP = 256 + 4;
    float* At = (float*)smem;                // [64 panel rows][AP] = 65 KiB
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int mh = pass >> 1, pwr = pass & 1;
        if (wr == pwr) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            At[(i * 16 + lq * 4 + q) * AP + nh * 128 + wc * 32 + j * 16 + l15] = mh ? acc[1][i][nh][j][q] : acc[0][i][nh][j][q];
        }
        __syncthreads();
        const int col = tid & 255;
        const int k = kbase + col;
        if (k < d.Ktot) {
            for (int pr = tid >> 8; pr < 64; pr += 2) {
                const int n = n0 + mh * 128 + pwr * 64 + pr;
                if (n < d.N)
                    {
                    float* dst = d.dwp + (long)sp * d.slab + (long)n * d.Ktot + k;
                    if (d.slab > 0) *dst = At[pr * AP + col];        // this pixel range's own slab (added up by the unpack kernel)
                    else __hip_atomic_fetch_add(dst, At[pr * AP + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }
#endif
}

// Splits for the one-block-per-CU kernel, two regimes (uclstm_wgrad_desc::overlapped):
//  * stand-alone launch: whole rounds of the 256 CUs, every extra slab priced (written here, read back by the unpack);
//  * launch that runs on a side stream next to the input-gradient chain: what the step needs from it is not a short kernel
//    but little interference -- few slabs and a grid that does not have to cover every CU (a 128-KiB block excludes every
//    64-KiB GEMM block from its CU).  Same-box sweeps of the STEP time: the stand-alone plan gave 16.49 k frames/s, 0.35x its
//    split counts 16.74 k.  Rule: the fewest splits that still give UCLSTM_P3_MIN_BLOCKS (default 128) blocks.
int auto_splits256(int64_t tiles, int64_t stages, bool slabs, int64_t panel_elems, bool overlapped) {
    const int64_t smax = stages / 8 > 1 ? stages / 8 : 1;
    if (overlapped) {
        static const int min_blocks = [] { const char* e = getenv("UCLSTM_P3_MIN_BLOCKS"); return e ? atoi(e) : 128; }();
        int64_t s = (min_blocks + tiles - 1) / tiles;
        if (s > smax) s = smax;
        return (int)(s < 1 ? 1 : s);
    }
    const double slab_cost = slabs ? (double)panel_elems * 8.0 / 4e12 / 1.9e-6 : 0.0;      // see auto_splits; a stage is ~1.9 us here
    const int64_t fixed = slabs ? 8 : 28;     // 256 KiB per block: ~50 us of float atomics vs ~11 us of stores
    int best = 1;
    int64_t best_cost = -1;
    for (int64_t s = 1; s <= smax && tiles * s <= 8192; ++s) {
        const int64_t per = (stages + s - 1) / s;
        const int64_t used = (stages + per - 1) / per;
        const int64_t rounds = (tiles * used + 255) / 256;
        const int64_t cost = rounds * (per + fixed) + (int64_t)(slab_cost * (double)used);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = (int)s;
        }
    }
    return best;
}

// Pixel-range splits: minimise (rounds of the 512 resident blocks) x (stages per block + fixed cost of a block's
// prologue and its 64-KiB atomic epilogue, ~8 stages' worth).
int auto_splits(int64_t tiles, int64_t stages, bool slabs, int64_t panel_elems) {
    // every extra slab is written here and read back by the unpack kernel: ~8 B per panel element at ~4 TB/s, in units of a
    // block-stage (~1.3 us for this kernel)
    const double slab_cost = slabs ? (double)panel_elems * 8.0 / 4e12 / 1.3e-6 : 0.0;
    const int64_t fixed = slabs ? 4 : 8;      // a block's prologue + epilogue in stages' worth (atomics cost ~2x the stores)
    int best = 1;
    int64_t best_cost = -1;
    const int64_t smax = stages / 4 > 1 ? stages / 4 : 1;
    for (int64_t s = 1; s <= smax && tiles * s <= 16384; ++s) {
        const int64_t per = (stages + s - 1) / s;
        const int64_t used = (stages + per - 1) / per;      // splits that actually get pixels
        const int64_t rounds = (tiles * used + 511) / 512;
        const int64_t cost = rounds * (per + fixed) + (int64_t)(slab_cost * (double)used);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = (int)s;
        }
    }
    return best;
}

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }


// ------------------------------------------------------------------------------------------------------------------
// Ring-staged weight gradient of the 64-channel full-resolution layers (C_out = 64, sources of 64 channels, 3x3 / pad 1,
// images a multiple of 64 wide and of 4 high): the p2 kernel above re-stages every activation plane once per tap and the dY
// plane once per K tile -- 48 flop per staged byte, staging-bound at ~0.6 PFLOP/s.  Here, as in igemm_fwd_c64_kernel, a
// persistent block walks 4-row x 64-column tiles of one image strip; the activation rows live in a ring of ten image rows
// (each staged ONCE, halo columns included), the dY tile (4 planes of [64 px][64 ch]) is double-buffered, and all nine taps are
// SHIFTED transposing reads of the ring: ~295 flop per staged byte.
//   out[n][tap][c] += sum_p dY[p][n] * x[p + tap][c]:  MFMA A = dY^T (16 n x 32 pixels), B = x (32 pixels x 16 c), both
//   produced by ds_read_b64_tr_b16 from pixel-major rows (same k-permutation on both, see the top of the file).
//   Wave w owns input-channel tile w (16 channels) x all 9 taps x all 4 n tiles: 36 accumulator tiles (144 VGPRs); per
//   (image row, 32-pixel half) it reads 4 dY fragments + 9 shifted x fragments for 36 MFMAs.
//   Ring rows are 72 LDS rows apart (66 used: 64 pixels + 2 halo columns): a multiple of 8, so the granule swizzle
//   g ^ ((row >> 1) & 3) does not depend on the slot and a tap shift of dx rows stays conflict-free (any 8 consecutive rows
//   cover all 64 banks).
// One launch per source; every block accumulates its whole tile range in registers and stores ONE slab (grid = slab count).
constexpr int WR_PITCH = 72 * 128;                 // bytes between ring slots
constexpr int WR_SLOTS = 10;
constexpr int WR_RING = WR_SLOTS * WR_PITCH;       // 92160
constexpr int WR_DZ = 4 * PLANE;                   // one dY tile: 4 image rows x [64 px][64 ch]
constexpr int WR_SMEM = WR_RING + 2 * WR_DZ;       // 157696

__global__ __launch_bounds__(256, 1) void igemm_wgrad_c64_kernel(const uclstm_wgrad_desc d, const int src, const int kcol0, const int per_tap,
                                                                  const int tiles_total, const int tiles_per_block, const uint32_t xbytes,
                                                                  const uint32_t ybytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ring = smem;
    unsigned char* Dz = smem + WR_RING;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lq = lane >> 4;
    const int H = d.H, W = d.W;
    const int strips = W >> 6;
    const int n_seq = d.n_img * strips;
    const int tiles_per_seq = H >> 2;
    const int t_begin = blockIdx.x * tiles_per_block;
    const int t_end = min(tiles_total, t_begin + tiles_per_block);
    if (t_begin >= t_end) return;

    const uclstm_seg G = d.seg[0];
    const uclstm_src S = d.src[src];
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)G.ptr, 0, ybytes, 0x00020000);

    // zero the halo columns (ring positions 0 and 65 of every slot) once: permanent when the image is one strip wide.
    // With more than one strip EVERY staged row brings its halo pixels by DMA (real pixels or out-of-range zeros), and these
    // plain stores must not exist: nothing orders a ds_write of wave 2 against the first halo DMAs of waves 0 / 1 to the same
    // bytes (different counters, no barrier in between) -- a late wave zeroed halo pixels that had already landed, ~1 in 4
    // processes, one tensor off by 1e-4 .. 6e-4 (found by the 256 x 256 reproducibility test of round 3).
    if (strips == 1 && tid < WR_SLOTS * 2 * 8) {
        const int sl = tid >> 4, side = (tid >> 3) & 1, ch = tid & 7;
        *(uint4*)(Ring + sl * WR_PITCH + side * 65 * 128 + ch * 16) = make_uint4(0, 0, 0, 0);
    }

    // ---- staging roles.  A DMA instruction of a wave covers 8 LDS rows x 128 B: lane = (row lane>>3, 16-byte position lane&7);
    // position cp of LDS row r holds source chunk 2*((cp>>1) ^ ((r>>1)&3)) + (cp&1).
    const int lrow = lane >> 3, cp = lane & 7;
    // activation row: pixels 8*(wave + 4i) + lrow, i = 0, 1, at ring positions pixel + 1
    uint32_t xvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = 8 * (wave + 4 * i) + lrow;
        const int r = px + 1;
        const int sc = 2 * ((cp >> 1) ^ ((r >> 1) & 3)) + (cp & 1);
        xvoff[i] = (uint32_t)(2 * (px * 64 + sc * 8));
    }
    // the DMA of instruction i lands at LDS rows 8*(wave+4i)+1 .. +8 of the slot: NOT 8-row aligned, so it is issued per row
    // piece through the lane's own address (M0-relative LDS addressing writes lane l at base + 16*l): base = row (8*(wave+4i)+1)
    // halo pixels (strips > 1): waves 0 / 1, lanes 0..7 -> ring position 0 / 65
    const int hpos = wave == 0 ? 0 : 65;
    const uint32_t halo_voff = (uint32_t)(2 * ((2 * (((lane & 7) >> 1) ^ ((hpos >> 1) & 3)) + (lane & 1)) * 8));
    // dY plane rows: instruction j of wave w covers pixel rows 8*(w + 4*(j&1)) + lrow of plane j>>1
    uint32_t yvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = 8 * (wave + 4 * i) + lrow;
        const int sc = 2 * ((cp >> 1) ^ ((px >> 1) & 3)) + (cp & 1);
        yvoff[i] = (uint32_t)(2 * (px * G.C + G.c_off + sc * 8));
    }

    int lk = 0, ly = 0, lslot = 0, loaded = -2, limg = 0, lstrip = 0;
#define WR_ISSUE_NEXT_ROW()                                                                                                 \
    {                                                                                                                       \
        const bool zero_ = ly == H || lk < 0 || lk >= n_seq;                                                                \
        const uint32_t soff_ = zero_ ? 0u : (uint32_t)((limg * H + ly) * W + lstrip * 64) * 128u;                           \
        unsigned char* dst_ = Ring + lslot * WR_PITCH + 128;                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst_ + wave * 1024), 16, zero_ ? OOB : xvoff[0], soff_, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst_ + (wave + 4) * 1024), 16, zero_ ? OOB : xvoff[1], soff_, 0, 0);  \
        if (strips > 1 && wave < 2) {                                                                                       \
            const bool ok_ = !zero_ && (wave == 0 ? lstrip > 0 : lstrip < strips - 1);                                      \
            const uint32_t hs_ = ok_ ? (wave == 0 ? soff_ - 128u : soff_ + 64u * 128u) : 0u;                                \
            if (lane < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(Ring + lslot * WR_PITCH + hpos * 128), 16, \
                                                                   ok_ ? halo_voff : OOB, hs_, 0, 0);                       \
        }                                                                                                                   \
        ++loaded;                                                                                                           \
        lslot = lslot == WR_SLOTS - 1 ? 0 : lslot + 1;                                                                      \
        if (ly == H) {                                                                                                      \
            ly = 0;                                                                                                         \
            ++lk;                                                                                                           \
            if (++lstrip == strips) { lstrip = 0; ++limg; }                                                                 \
        } else {                                                                                                            \
            ++ly;                                                                                                           \
        }                                                                                                                   \
    }
    // the dY tile of tile (sequence k_, row group tr_) into buffer b_
#define WR_ISSUE_DZ(k_, tr_, b_)                                                                                            \
    {                                                                                                                       \
        const int img_ = (k_) / strips, st_ = (k_) - img_ * strips;                                                         \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 4; ++pl_) {                                                               \
            const uint32_t so_ = (uint32_t)((img_ * H + 4 * (tr_) + pl_) * W + st_ * 64) * (uint32_t)(2 * G.C);             \
            unsigned char* dp_ = Dz + (b_) * WR_DZ + pl_ * PLANE;                                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (lds_ptr)(dp_ + wave * 1024), 16, yvoff[0], so_, 0, 0);           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (lds_ptr)(dp_ + (wave + 4) * 1024), 16, yvoff[1], so_, 0, 0);     \
        }                                                                                                                   \
    }

    // ---- fragment read offsets: pixel row trow of a 16-row group, 8 bytes (4 channels) at (l15 & 3)
    const int trow = 4 * lq + (l15 >> 2);
    int yoff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) yoff[a] = trow * 128 + ((a ^ ((trow >> 1) & 3)) << 5) + (l15 & 3) * 8;
    int xoff[3];                 // this wave's 16-channel granule, window shifted by dx rows (ring position = pixel + dx)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int r = dx + trow;
        xoff[dx] = r * 128 + ((wave ^ ((r >> 1) & 3)) << 5) + (l15 & 3) * 8;
    }

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int k = t_begin / tiles_per_seq;
    int tr = t_begin - k * tiles_per_seq;
    WR_ISSUE_DZ(k, tr, 0)
    for (int tt = t_begin; tt < t_end; ++tt) {
        const int v0 = k * (H + 1) + 4 * tr;
        if (tt == t_begin) {      // cursor at virtual row v0 - 1
            const int v = v0 - 1;
            if (v < 0) { lk = -1; ly = H; limg = 0; lstrip = -1; } else { lk = v / (H + 1); ly = v - lk * (H + 1); limg = lk / strips; lstrip = lk - limg * strips; }
            lslot = (v + 1) % WR_SLOTS;
            loaded = v - 1;
        }
        // After a strip boundary ONE row was not prefetched (the look-ahead is capped at the four free slots); it lands in the
        // slot of the previous tile's first row, which a slower wave may still be reading -- no barrier inside a tile, and on
        // the weight-gradient stream another kernel shares the CU, so the waves drift (seen as one tensor off by 1e-4 .. 6e-4
        // in ~1 of 3 processes at 256 x 256).  Everybody first; block-uniform condition, once per image strip.
        if (tt != t_begin && loaded < v0 + 4) __syncthreads();
        while (loaded < v0 + 4) WR_ISSUE_NEXT_ROW()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int k1 = k, tr1 = tr + 1;
        if (tr1 == tiles_per_seq) { tr1 = 0; ++k1; }
        if (tt + 1 < t_end) {          // next tile: its dY tile into the other buffer, up to four new rows into the free slots
            WR_ISSUE_DZ(k1, tr1, (tt + 1 - t_begin) & 1)
            const int upto = min(k1 * (H + 1) + 4 * tr1 + 4, loaded + 4);
            while (loaded < upto) WR_ISSUE_NEXT_ROW()
        }
        const unsigned char* Y = Dz + ((tt - t_begin) & 1) * WR_DZ;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const unsigned char* rowp[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rowp[dy] = Ring + ((v0 + rr + dy) % WR_SLOTS) * WR_PITCH;      // virtual row v0 + rr + dy - 1
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                act16x8 yf[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) yf[a] = tr_frag128(Y + rr * PLANE + ks * 32 * 128 + yoff[a]);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const act16x8 xf = tr_frag128(rowp[t / 3] + ks * 32 * 128 + xoff[t % 3]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[t][a] = UCLSTM_MFMA_16x16x32(yf[a], xf, acc[t][a], 0, 0, 0);
                }
            }
        }
        k = k1;
        tr = tr1;
    }
    // ---- one slab per block: dWp[n][tap * per_tap + kcol0 + 16*wave + l15], n = a*16 + lq*4 + r
    float* out = d.dwp + (long)blockIdx.x * d.slab + kcol0 + wave * 16 + l15;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = a * 16 + lq * 4 + r;
                if (n < d.N) out[(long)n * d.Ktot + t * per_tap] = acc[t][a][r];
            }
#endif
}
#undef WR_ISSUE_NEXT_ROW
#undef WR_ISSUE_DZ

// launch conditions of igemm_wgrad_c64_kernel
inline bool wgrad_c64_ok(const uclstm_wgrad_desc& d, bool plain) {
    static const bool off = [] { const char* e = getenv("UCLSTM_WGRAD_RING"); return e && e[0] == '0'; }();
    if (off || !plain || d.ktap != 3 || d.pad != 1 || d.nseg != 1 || d.slab <= 0) return false;
    if (d.N > 64 || d.N <= 0 || (d.W & 63) || (d.H & 3)) return false;
    const uclstm_seg& g = d.seg[0];
    if (g.C != 64 || g.c_off != 0 || g.n_begin != 0) return false;
    for (int s = 0; s < d.nsrc; ++s)
        if (d.src[s].C != 64) return false;
    if (d.Ktot != 9 * 64 * d.nsrc) return false;
    if ((int64_t)d.n_img * d.H * d.W * 128 >= ((int64_t)1 << 31) - (1 << 22)) return false;
    if ((int64_t)d.n_img * (d.W / 64) * (d.H + 1) >= ((int64_t)1 << 30)) return false;
    return true;
}
inline int wgrad_c64_grid(const uclstm_wgrad_desc& d, int& per) {
    const int tiles_total = d.n_img * (d.W / 64) * (d.H / 4);
    const int cap = d.overlapped ? 128 : 256;
    const int blocks = tiles_total < cap ? tiles_total : cap;
    per = (tiles_total + blocks - 1) / blocks;
    return (tiles_total + per - 1) / per;
}

template <int WN, int NSRC>
int32_t launch_p2(const uclstm_wgrad_desc& d, const WDerived& dv, const P2& p2, int64_t nblk, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_p2_kernel<WN, NSRC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   WShape<WN>::SMEM);
        attr_done = true;
    }
    UCLSTM_LAUNCH((igemm_wgrad_p2_kernel<WN, NSRC>), dim3((unsigned)nblk), dim3(256), WShape<WN>::SMEM, st, d, dv, p2);
    return UCLSTM_OK;
}

bool wsrc_ok(const uclstm_src& s) {
    return s.ptr && s.C > 0 && (s.C % 8) == 0 && s.Hs > 0 && s.Ws > 0 && ((uintptr_t)s.ptr % 16) == 0;
}

}  // namespace

namespace {
// plan_only: validate, choose kernel + splits and return the split count without launching (dwp may be null)
// query: 0 = launch, 1 = return the pixel-range count (uclstm_igemm_wgrad_splits), 2 = return which kernel would run
// (uclstm_igemm_wgrad_shape: 3 = 256 x 256 8-phase, 2 = 128 x 128, 1 = 64 x 256 for C_out <= 64, 0 = generic addressing)
int32_t wgrad_run(const uclstm_wgrad_desc* dp, void* stream, int query) {
    const bool plan_only = query != 0;
    if (!dp) return UCLSTM_E_BADARG;
    const uclstm_wgrad_desc& d = *dp;
    if (d.n_img <= 0 || d.H <= 0 || d.W <= 0) return UCLSTM_E_BADARG;
    if (d.ktap < 1 || d.ktap > 7 || d.scale < 1 || d.scale > 2 || d.pad < 0 || d.pad > 3) return UCLSTM_E_BADARG;
    if (d.nsrc < 1 || d.nsrc > 2 || (!d.dwp && !plan_only) || d.N <= 0 || (d.N % 8) || d.splits < 0 || d.slab < 0) return UCLSTM_E_BADARG;
    if (d.slab > 0 && d.slab < (int64_t)d.N * d.Ktot) return UCLSTM_E_BADARG;
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
    if (wgrad_c64_ok(d, plain)) {            // ring-staged kernel of the 64-channel full-resolution layers: grid = slab count
        int per = 0;
        const int grid = wgrad_c64_grid(d, per);
        if (query == 2) return 4;
        if (query == 1) return grid;
        if (d.splits != grid) return UCLSTM_E_BADARG;
        static bool attr_r = false;
        if (!attr_r) {
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WR_SMEM);
            attr_r = true;
        }
        const int tiles_total = d.n_img * (d.W / 64) * (d.H / 4);
        const uint32_t ybytes = (uint32_t)((int64_t)d.n_img * d.H * d.W * d.seg[0].C * 2);
        for (int sidx = 0; sidx < d.nsrc; ++sidx) {
            const uint32_t xbytes = (uint32_t)((int64_t)d.n_img * d.H * d.W * 128);
            UCLSTM_LAUNCH(igemm_wgrad_c64_kernel, dim3(grid), dim3(256), WR_SMEM, (hipStream_t)stream, d, sidx, sidx * 64, 64 * d.nsrc, tiles_total,
                          per, xbytes, ybytes);
        }
        return UCLSTM_OK;
    }
    dv.M = (long)d.n_img * d.H * d.W;
    if (dv.M >= ((long)1 << 31) - 4096) return UCLSTM_E_BADARG;
    dv.dHW = make_fastdiv((uint32_t)(d.H * d.W));
    dv.dW = make_fastdiv((uint32_t)d.W);
    dv.dPerTap = make_fastdiv((uint32_t)(dv.kseg0 + dv.kseg1));

    // fast path (buffer-addressed, division-free): see igemm_wgrad_p2_kernel
    static const bool no_fast = [] { const char* e = getenv("UCLSTM_WGRAD_GENERIC"); return e && e[0] == '1'; }();
    bool fast = plain && !no_fast && d.nseg == 1 && is_pow2(d.H) && is_pow2(d.W) && (dv.M % TP) == 0 &&
                ((d.ktap == 3 && d.pad == 1) || (d.ktap == 1 && d.pad == 0));
    P2 p2{};
    if (fast) {
        const int64_t lim = ((int64_t)1 << 31) - (1 << 22);
        const int64_t yb = dv.M * d.seg[0].C * 2;
        fast = yb < lim;
        p2.ybytes = (uint32_t)yb;
        for (int s = 0; s < d.nsrc && fast; ++s) {
            const int64_t bias = (int64_t)(d.src[s].Ws + 1) * d.src[s].C * 2;
            const int64_t xb = dv.M * d.src[s].C * 2 + bias;
            fast = xb < lim;
            p2.xbias[s] = (uint32_t)bias;
            p2.xbytes[s] = (uint32_t)xb;
        }
        p2.lw = ilog2(d.W);
        p2.lh = ilog2(d.H);
    }
    static const bool no_p3 = [] { const char* e = getenv("UCLSTM_WGRAD_NO256"); return e && e[0] == '1'; }();
    const bool big = fast && !no_p3 && d.N >= 256;              // 256 x 256 tile, 8-phase pipeline
    static const bool no_192 = [] { const char* e = getenv("UCLSTM_WGRAD_NO192"); return e && e[0] == '1'; }();
    const int wn = (fast && d.N <= 64) ? ((!no_192 && d.Ktot % 192 == 0) ? 3 : 1) : 2;      // 3: 64 x 192 tile (no padding at K = 576 / 1152)
    const int tn = big ? 256 : (wn == 2 ? 128 : 64), tc = big ? 256 : (wn == 2 ? 128 : (wn == 3 ? 192 : 256));
    dv.n_kt = (d.Ktot + tc - 1) / tc;
    dv.n_nt = (d.N + tn - 1) / tn;
    dv.kt_per_tap = (fast && ((dv.kseg0 + dv.kseg1) % tc) == 0) ? (dv.kseg0 + dv.kseg1) / tc : 0;
    uclstm_wgrad_desc dd = d;
    const bool slabs = d.slab > 0;
    if (dd.splits <= 0)
        dd.splits = big ? auto_splits256((int64_t)dv.n_kt * dv.n_nt, (dv.M + TP - 1) / TP, slabs, (int64_t)d.N * d.Ktot, d.overlapped != 0)
                        : auto_splits((int64_t)dv.n_kt * dv.n_nt, (dv.M + TP - 1) / TP, slabs, (int64_t)d.N * d.Ktot);
    long chunk = (dv.M + dd.splits - 1) / dd.splits;
    chunk = (chunk + TP - 1) / TP * TP;
    dv.chunk = chunk;
    dd.splits = (int)((dv.M + chunk - 1) / chunk);          // every split owns pixels (slab mode: every slab is written)
    if (query == 2) return big ? 3 : (fast ? (wn == 2 ? 2 : 1) : 0);
    if (plan_only) return dd.splits;
    const int64_t nblk = (int64_t)dv.n_kt * dv.n_nt * dd.splits;
    if (nblk <= 0 || nblk > 0x7fffffff) return UCLSTM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;

    if (big) {
        static bool attr3 = false;
        // seven half-tiles ahead (five in flight across a barrier) is the default: +11 % on the 4096 x 18432 gradient over six
        // (same box A/B); UCLSTM_P3_LEAD=6 selects the shallower ring
        static const bool lead7 = [] { const char* e = getenv("UCLSTM_P3_LEAD"); return !(e && e[0] == '6'); }();
        if (!attr3) {
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_p3_kernel<1, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_p3_kernel<2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_p3_kernel<1, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_p3_kernel<2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
            attr3 = true;
        }
        if (lead7) {
            if (d.nsrc == 1) UCLSTM_LAUNCH((igemm_wgrad_p3_kernel<1, 7>), dim3((unsigned)nblk), dim3(512), P3_SMEM, st, dd, dv, p2);
            else UCLSTM_LAUNCH((igemm_wgrad_p3_kernel<2, 7>), dim3((unsigned)nblk), dim3(512), P3_SMEM, st, dd, dv, p2);
        } else {
            if (d.nsrc == 1) UCLSTM_LAUNCH((igemm_wgrad_p3_kernel<1, 6>), dim3((unsigned)nblk), dim3(512), P3_SMEM, st, dd, dv, p2);
            else UCLSTM_LAUNCH((igemm_wgrad_p3_kernel<2, 6>), dim3((unsigned)nblk), dim3(512), P3_SMEM, st, dd, dv, p2);
        }
        return UCLSTM_OK;
    }
    if (fast) {
        if (wn == 1) return d.nsrc == 1 ? launch_p2<1, 1>(dd, dv, p2, nblk, st) : launch_p2<1, 2>(dd, dv, p2, nblk, st);
        if (wn == 3) return d.nsrc == 1 ? launch_p2<3, 1>(dd, dv, p2, nblk, st) : launch_p2<3, 2>(dd, dv, p2, nblk, st);
        return d.nsrc == 1 ? launch_p2<2, 1>(dd, dv, p2, nblk, st) : launch_p2<2, 2>(dd, dv, p2, nblk, st);
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_done = true;
    }
    if (plain)
        UCLSTM_LAUNCH(igemm_wgrad_kernel<true>, dim3((unsigned)nblk), dim3(256), SMEM_BYTES, st, dd, dv);
    else
        UCLSTM_LAUNCH(igemm_wgrad_kernel<false>, dim3((unsigned)nblk), dim3(256), SMEM_BYTES, st, dd, dv);
    return UCLSTM_OK;
}
}  // namespace

extern "C" int32_t uclstm_igemm_wgrad(const uclstm_wgrad_desc* d, void* stream) { return wgrad_run(d, stream, 0); }

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_igemm_wgrad_splits(const uclstm_wgrad_desc* d) { return wgrad_run(d, nullptr, 1); }
#endif
#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_igemm_wgrad_shape(const uclstm_wgrad_desc* d) { return wgrad_run(d, nullptr, 2); }
#endif
