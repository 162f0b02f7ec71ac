// Implicit-GEMM convolution on MFMA (gfx950), pixel-major forward family.
//
//   D[n][m] = sum_k Wp[n][k] * A[m][k]        n = panel row (output channel), m = output pixel
//
// A is never materialised: for K-step (tap, source, 64-channel slice) row m of A is the 128
// contiguous bytes src[s][img][y*scale+dy-pad][x*scale+dx-pad][c0:c0+64] (zeros outside the
// image), so torch.cat([x,h]) / cat([skip,up]) / F.pad of the reference (train/unet.py:28,
// :95-98) never exist in memory.  4 waves per block, each wave a 64x64 output sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 with the PANEL as the MFMA A operand: a lane then owns 4 consecutive
// output channels of one pixel, which makes the LSTM gate quadruple (i,f,g,o of one hidden
// channel = 4 M-subtiles of a wave) lane-local and lets the epilogue pack 8-byte channel runs.
// Block shapes of the per-tap loop: 128 panel rows x 128 pixels (waves 2x2) and, for C_out <= 64 layers,
// 64 panel rows x 256 pixels (waves 1x4) so that no MFMA work is spent on absent channels.
// A third shape (128 x 256, 8 waves, one block per CU) runs the PATCH loop that most 3x3 launches
// take: the activations of a 64-channel chunk are staged once and the nine taps are shifted LDS
// reads of that staging (see the comment at `if constexpr (SHP == 2)`).
// LDS rows are 128 B, XOR-swizzled by (row & 7) on the 16-byte chunk (conflict-free
// ds_read_b128, guide T2).  Operands go global->LDS directly through BUFFER descriptors
// (buffer_load_dwordx4 ... lds, 16 B/lane).  Measured on gfx950 (tools/probes/): a lane whose
// voffset + soffset is >= num_records writes 16 ZERO bytes into LDS, so padding costs no zero
// page and no 64-bit select.  Per staged row the byte offset of tap (0,0) and a 9-bit "tap
// outside the image" mask are computed once; a K-step then needs 2 VALU per activation row
// (shift the mask bit of this tap into bit 31, OR it into the offset) and none for the weight
// rows -- the (tap, channel) advance rides in the SGPR soffset.  (The register-staged first
// version was VALU-issue bound at 4.4 VALU per MFMA.)  Two LDS stages, one barrier per K-step.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 64;

constexpr uint32_t OOB = 0x80000000u;      // >= num_records of every descriptor (tensors are < 2 GiB here)

struct Derived {
    int Mg;          // pixels per statistic group
    int tpg;         // M tiles per group
    int n_mtiles;    // groups * tpg
    int n_ntiles;
    int kseg0, kseg1;
    int ksteps;      // total K steps
    int ksplit;      // K ranges (UCLSTM_EPI_ATOMIC only, else 1)
    int kper;        // K steps per range
    FastDiv dHW, dW; // pixel index -> (image, y, x)
    FastDiv dPHW, dW2;   // shared-zero padded index -> (image, y, x): divisors (H+1)(W+1) and W+1   (patch shape only)
    uint32_t xbias[2];   // bytes the source descriptor starts before the tensor (tap (0,0) of a border pixel is "negative")
    uint32_t xbytes[2];  // descriptor size: tensor bytes + bias
    uint32_t wbytes;     // panel bytes
    // patch shape on images wider than 64 pixels: a tile is a STRIP block of 4 image rows x 64 columns (below)
    int strip;           // 0: tile = 256 consecutive pixels of the flattened (image, y, x) order; 1: 4 rows x 64 columns
    int strips;          // W / 64
    int tiles_per_img;   // (H / 4) * (W / 64)
};

// Block shapes (SHP): 0 = 128 rows x 128 pixels, waves 2x2;  1 = 64 x 256, waves 1x4 (C_out <= 64);
// 2 = 128 rows x 256 pixels, waves 2x4, ONE block per CU, the "patch" K loop (below).
// (A 128 x 256 / 512-thread shape with a 3-stage ring and counted vmcnt over the per-tap staging measured 5-12 % slower on
// every layer and was removed: profiles/round1_notes.md.)
template <int SHP>
struct Shape {
    static constexpr int WN = SHP == 1 ? 1 : 2;  // waves along panel rows
    static constexpr int WM = SHP == 0 ? 2 : 4;  // waves along pixels
    static constexpr int NT = 64 * WN * WM;      // threads per block
    static constexpr int MINB = SHP == 2 ? 1 : 2;    // blocks per CU the register budget is sized for
    static constexpr int RS = NT / 8;            // rows covered by one DMA round of the whole block
    static constexpr int TBN = 64 * WN;          // panel rows per tile
    static constexpr int TBM = 64 * WM;          // pixels per tile
    static constexpr int XR = TBM / RS;          // X rows staged per thread
    static constexpr int WR = TBN / RS;          // W rows staged per thread
    static constexpr int XBYTES = TBM * BK * 2;
    static constexpr int WBYTES = TBN * BK * 2;
    static constexpr int STAGE = XBYTES + WBYTES;
    static constexpr int STAGES = 2;
    // patch shape: two activation patches of PROWS padded pixels (128 B each) + a ring of three weight tiles
    static constexpr int PROWS = 448;
    static constexpr int PBYTES = PROWS * BK * 2;             // 57344: 2 patches + 3 weight tiles = 160 KiB exactly
    static constexpr int PROUNDS = (PROWS + RS - 1) / RS;     // 7 DMA rounds of 64 rows
    static constexpr int WSLOTS = 3;
    static constexpr int SMEM = SHP == 2 ? 2 * PBYTES + WSLOTS * WBYTES : STAGES * STAGE;  // 64 KiB / 80 KiB / 160 KiB
    static constexpr int OT_PITCH = TBN * 2 + 16;
};

// Sum over the 16 lanes of a DPP row with four v_add_f32 + DPP (xor 1, xor 2, half-row mirror, row mirror); every lane of the
// row ends up with the total.  (__shfl_xor would be four ds_bpermute round trips per value.)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));     // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));     // row_mirror
    return v;
}

// KT = widest kernel the per-tap loop's tap masks cover: 3 (one 32-bit word, the hot path) or 7 (two words: ConvLSTM cells
// with 5x5 / 7x7 gate convolutions, 128 x 128 shape only).
// The kernel body as a device function of (descriptor, derived constants, block id within this launch plan): the plain
// kernel below passes blockIdx.x, the grouped kernel (several independent GEMMs in ONE launch) the block's index inside its
// member's sub-grid.
template <int EPI, int SHP, int NSRC, int KT = 3>
__device__ __forceinline__ void igemm_fwd_body(const uclstm_igemm_desc& d, const Derived& dv, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-resource type does not exist in the host pass (the stub needs no body)
    using SH = Shape<SHP>;
    constexpr int TBN = SH::TBN, TBM = SH::TBM, XR = SH::XR, WR = SH::WR, NT = SH::NT, RS = SH::RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / SH::WM;                    // wave position along panel rows
    const int wpx = wave - wc * SH::WM;              // wave position along pixels
    const int l15 = lane & 15;
    const int lq = lane >> 4;

    const int per_split = dv.n_mtiles * dv.n_ntiles;
    const int lid0 = xcd_remap(bid, per_split * dv.ksplit);
    const int ks = lid0 / per_split;                 // K range of this block (0 unless split-K)
    const int lid = lid0 - ks * per_split;
    int nt, mt;
    grouped_tile(lid, dv.n_ntiles, dv.n_mtiles, 8, nt, mt);      // 8 panel-row tiles x 8 pixel tiles resident per XCD
    const int kstep_begin = ks * dv.kper;
    const int kstep_end = min(dv.ksteps, kstep_begin + dv.kper);
    const int g = mt / dv.tpg;
    const int tile = mt - g * dv.tpg;
    const int m_local0 = tile * TBM;
    const long m0 = (long)g * dv.Mg + m_local0;
    const int n0 = nt * TBN;
    const int HW = d.H * d.W;
    // strip tiles (patch shape, images wider than 64): tile mt = (image, 4-row band, 64-column strip); tile-local pixel r is
    // image row y0 + (r >> 6), column x0 + (r & 63).  Everything below that needs a pixel's linear index goes through PIX().
    const bool strip = SHP == 2 && dv.strip;
    int s_img = 0, s_y0 = 0, s_x0 = 0;
    if (strip) {
        s_img = mt / dv.tiles_per_img;
        const int rem = mt - s_img * dv.tiles_per_img;
        const int band = rem / dv.strips;
        s_y0 = band * 4;
        s_x0 = (rem - band * dv.strips) * 64;
    }
    const int rows_valid = strip ? TBM : min(TBM, dv.Mg - m_local0);
#define PIX(r_) (strip ? ((long)(s_img * d.H + s_y0 + ((r_) >> 6)) * d.W + s_x0 + ((r_) & 63)) : (m0 + (r_)))

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) void* lds_ptr;

    if constexpr (SHP == 2) {
        // ---- patch K loop (3x3 / pad 1 / stride 1, every source the size of the output, C % 64 == 0) ----------------
        // The per-tap loop below stages the SAME activation rows nine times per 64-channel chunk (once per tap, shifted).
        // Skipping eight of the nine stagings (wrong results, timing only) ran the 3x3 layers 12-30 % faster: the LDS-DMA
        // issue (60-180 cycles of the issuing wave each, guide) is what the per-tap loop is bound by, not the MFMAs.  Here a
        // chunk's activations are staged ONCE, as a patch: the tile's 256 consecutive pixels and their halo, laid out by a
        // PADDED-FLAT index that shares its zeros: q(img,y,x) = img*(H+1)(W+1) + y*(W+1) + x, where column W of every row and
        // row H of every image are zero (out-of-range DMA).  Column W is at once the right border of its row and the left
        // border of the next one, row H the bottom border of its image and the top border of the next.  Tap (dy,dx) of pixel
        // p is patch row prow(p) + (dy-1)*(W+1) + (dx-1): nine shifted LDS reads of one staging.  256 pixels span 307-406
        // rows for 64x64 ... 4x4 images.  Per chunk a wave issues 7 patch pieces + 9 x 2 weight pieces instead of 9 x 8.
        //   LDS: 2 patches x 448 rows x 128 B (the next chunk's patch lands while this one is multiplied) + a ring of three
        //   128 x 128-B weight tiles (two K-steps in flight, counted vmcnt, one raw s_barrier per K-step).
        constexpr int PROUNDS = SH::PROUNDS, PBYTES = SH::PBYTES, WSLOT = SH::WBYTES;
        unsigned char* const Wring = smem + 2 * PBYTES;
        // Images wider than 64 pixels: 256 consecutive pixels would span (2..4 + 2) full image rows of W + 1 patch rows each
        // -- more than the 448 rows the two patch buffers hold.  There the tile is a 4-row x 64-column block of ONE image and
        // the patch its (4 + 2) x (64 + 2) neighbourhood = 396 rows of pitch 66: halo rows / columns are real pixels of the
        // neighbouring band / strip, or zeros (out-of-range DMA) at the image border.  Same K order, same arithmetic.
        const int W1 = strip ? 66 : d.W + 1;
        const int PHW = (d.H + 1) * (d.W + 1);
        const int lrow0 = tid >> 3;                        // row within a 64-row DMA round
        const int lchunk = (tid & 7) ^ (lrow0 & 7);        // linear destination, swizzled source (as the per-tap loop)
        const __amdgpu_buffer_rsrc_t rsx0 = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[0].ptr, 0, dv.xbytes[0], 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx1 =
            NSRC > 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)d.src[1].ptr, 0, dv.xbytes[1], 0x00020000) : rsx0;
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.wp, 0, dv.wbytes, 0x00020000);

        // patch row 0 = tap (0,0) of the tile's first pixel = q(first) - (W+1) - 1 (negative only for the first tile)
        int q0 = 0;
        if (!strip) {
            const uint32_t m = (uint32_t)m0;
            const int img = (int)fdiv(m, dv.dHW);
            const uint32_t rem = m - (uint32_t)img * (uint32_t)HW;
            const int y = (int)fdiv(rem, dv.dW);
            q0 = img * PHW + y * W1 + ((int)rem - y * d.W) - W1 - 1;
        }
        // patch rows this lane stages (row lrow0 + 64*i): source PIXEL index, or "outside" (shared zeros / beyond the batch)
        uint32_t ppix[PROUNDS];
        uint32_t pout = 0;
#pragma unroll
        for (int i = 0; i < PROUNDS; ++i) {
            if (strip) {
                const int j = lrow0 + SH::RS * i;              // patch row = (band row pr, column pc) of the 6 x 66 neighbourhood
                const int pr = (j * 993) >> 16;                // j / 66 for j < 448
                const int yy = s_y0 - 1 + pr, xx = s_x0 - 1 + (j - pr * 66);
                const bool in = pr < 6 && (unsigned)yy < (unsigned)d.H && (unsigned)xx < (unsigned)d.W;
                ppix[i] = (uint32_t)((s_img * d.H + yy) * d.W + xx);
                pout |= in ? 0u : (1u << i);
            } else {
                const int q = q0 + lrow0 + SH::RS * i;
                const uint32_t qu = (uint32_t)max(q, 0);
                const int img = (int)fdiv(qu, dv.dPHW);
                const uint32_t rem = qu - (uint32_t)img * (uint32_t)PHW;
                const int yy = (int)fdiv(rem, dv.dW2);
                const int xx = (int)rem - yy * W1;
                const bool in = q >= 0 && yy < d.H && xx < d.W && img < d.n_img;
                ppix[i] = (uint32_t)((img * d.H + yy) * d.W + xx);
                pout |= in ? 0u : (1u << i);
            }
        }
        // patch rows this lane READS: the CENTRE tap of pixel wpx*64 + b*16 + l15
        int prow[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (strip) {
                prow[b] = (wpx + 1) * 66 + b * 16 + l15 + 1;   // tile-local pixel (row wpx, column b*16 + l15)
            } else {
                const uint32_t m = (uint32_t)(m0 + min(wpx * 64 + b * 16 + l15, rows_valid - 1));
                const int img = (int)fdiv(m, dv.dHW);
                const uint32_t rem = m - (uint32_t)img * (uint32_t)HW;
                const int y = (int)fdiv(rem, dv.dW);
                prow[b] = img * PHW + y * W1 + ((int)rem - y * d.W) - q0;
            }
        }
        uint32_t wvoff[SH::WR];
#pragma unroll
        for (int i = 0; i < SH::WR; ++i) {
            const int nrow = n0 + lrow0 + SH::RS * i;
            wvoff[i] = nrow < d.N ? (uint32_t)(2 * (nrow * d.Ktot + lchunk * 8)) : OOB;
        }
        const int spt = (dv.kseg0 + dv.kseg1) / BK;
        const int s0steps = dv.kseg0 / BK;
        const int wrow_lds = wave * 8 * 128;
        const int nsteps = kstep_end - kstep_begin;              // a multiple of 9, starting on a chunk boundary (host)
        const int chunk_begin = kstep_begin / 9;
        const int nchunks = nsteps / 9;

        // one DMA round (64 rows) of the patch of chunk `chunk` into patch buffer `pb`
#define PATCH_ROUND(i_, chunk_, pb_)                                                                                      \
        do {                                                                                                              \
            if ((i_) * SH::RS + wave * 8 < SH::PROWS) {                                                                    \
                const bool s1_ = NSRC > 1 && (chunk_) >= s0steps;                                                          \
                const int c0_ = ((chunk_) - (s1_ ? s0steps : 0)) * BK;                                                     \
                const int C_ = s1_ ? d.src[1].C : d.src[0].C;                                                              \
                const uint32_t off_ = ((pout >> (i_)) & 1u) ? OOB : ppix[i_] * (uint32_t)(2 * C_) + (uint32_t)(lchunk * 16); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s1_ ? rsx1 : rsx0,                                                \
                                                         (lds_ptr)(smem + (pb_) * PBYTES + (i_) * SH::RS * 128 + wrow_lds), 16, \
                                                         off_, (uint32_t)(2 * c0_), 0, 0);                                  \
            }                                                                                                             \
        } while (0)
        // the weight tile of K-step (chunk, tap) into ring slot `slot`
#define WTILE(chunk_, tap_, slot_)                                                                                        \
        do {                                                                                                              \
            const uint32_t koff_ = (uint32_t)((tap_) * spt + (chunk_)) * (BK * 2);                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < SH::WR; ++i_)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(Wring + (slot_) * WSLOT + i_ * SH::RS * 128 + wrow_lds), 16, \
                                                         wvoff[i_], koff_, 0, 0);                                          \
        } while (0)

        // prologue: first patch, weight tiles of steps 0 and 1 (issue order matters for the counted waits below)
#pragma unroll
        for (int i = 0; i < PROUNDS; ++i) PATCH_ROUND(i, chunk_begin, 0);
        WTILE(chunk_begin, 0, 0);
        WTILE(chunk_begin, 1, 1);

        int slot = 0;                         // ring slot of the current K-step
        for (int ci = 0; ci < nchunks; ++ci) {
            const int chunk = chunk_begin + ci;
            const bool more = ci + 1 < nchunks;
            const unsigned char* const P = smem + (ci & 1) * PBYTES;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // everything but the two youngest pieces (the weight tile of the NEXT step) has landed: this step's weight
                // tile and, at tap 0, the whole patch
                if (more || tap < 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // every wave is past its reads of the previous step: its ring slot and the other patch buffer are free
                if (more && tap < PROUNDS) PATCH_ROUND(tap, chunk + 1, (ci + 1) & 1);
                {
                    const int t2 = tap + 2 >= 9 ? tap + 2 - 9 : tap + 2;
                    const int c2 = tap + 2 >= 9 ? chunk + 1 : chunk;
                    int s2 = slot + 2;
                    s2 = s2 >= 3 ? s2 - 3 : s2;
                    if (more || tap + 2 < 9) WTILE(c2, t2, s2);
                }
                const int toff = (tap / 3 - 1) * W1 + (tap % 3 - 1);
                const unsigned char* const Wt = Wring + slot * WSLOT;
                act16x8 wf[4], xf[4];
                int xaddr[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = prow[b] + toff;
                    xaddr[b] = r * 128 + ((lq ^ (r & 7)) << 4);
                }
                const int choffw = (lq ^ (l15 & 7)) << 4;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) wf[a] = *(const act16x8*)(Wt + (wc * 64 + a * 16 + l15) * 128 + (choffw ^ (h << 6)));
#pragma unroll
                    for (int b = 0; b < 4; ++b) xf[b] = *(const act16x8*)(P + (xaddr[b] ^ (h << 6)));
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = UCLSTM_MFMA_16x16x32(wf[a], xf[b], acc[a][b], 0, 0, 0);
                }
                slot = slot + 1 >= 3 ? 0 : slot + 1;
            }
        }
#undef PATCH_ROUND
#undef WTILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {

    // ---- operand staging: direct-to-LDS buffer loads (16 B per lane, guide section 5) ----
    // DMA instruction i of wave w fills LDS rows RS*i + 8*w + (lane>>3), 16-byte position lane&7 (1 KiB contiguous per
    // wave instruction).  The XOR swizzle sits on the SOURCE side: position p of row r receives channel chunk
    // p ^ (r&7) (rule 21: linear destination, swizzled source, swizzled read).  Rows whose tap falls outside the image,
    // or beyond the tile's last pixel, get bit 31 set in their offset: out of range -> the DMA writes zeros.
    const int lrow0 = tid >> 3;                        // = 8*wave + (lane>>3)
    const int lchunk = (tid & 7) ^ (lrow0 & 7);        // channel chunk that belongs at this lane's LDS position
    const __amdgpu_buffer_rsrc_t rsx0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)d.src[0].ptr - dv.xbias[0]), 0, dv.xbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx1 =
        NSRC > 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)d.src[1].ptr - dv.xbias[1]), 0, dv.xbytes[1], 0x00020000)
                 : rsx0;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.wp, 0, dv.wbytes, 0x00020000);

    // per staged row: byte offset of tap (0,0) in each source (descriptor-relative, never negative), and a bit mask of the
    // taps that fall OUTSIDE the image (all ones for rows beyond the tile's last pixel)
    // (KT = 7: two 32-bit mask words for up to 7x7 = 49 taps -- ConvLSTMCell accepts any odd kernel_size, train/unet.py:15-19)
    constexpr int XRH = KT > 3 ? XR : 1;
    uint32_t roff0[XR], roff1[XR];
    uint32_t nv0[XR], nv1[XR], nv0h[XRH], nv1h[XRH];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int r = lrow0 + RS * i;
        const bool rvalid = r < rows_valid;
        const uint32_t m = (uint32_t)(m0 + (rvalid ? r : 0));
        const int img = (int)fdiv(m, dv.dHW);
        const uint32_t rem = m - (uint32_t)img * (uint32_t)HW;
        const int y = (int)fdiv(rem, dv.dW);
        const int x = (int)rem - y * d.W;
#pragma unroll
        for (int sidx = 0; sidx < NSRC; ++sidx) {
            const uclstm_src S = d.src[sidx];
            const int ys0 = y * d.scale - d.pad - S.offY;
            const int xs0 = x * d.scale - d.pad - S.offX;
            const uint32_t ro = (uint32_t)(2 * (((img * S.Hs + ys0) * S.Ws + xs0) * S.C + lchunk * 8) + (int)dv.xbias[sidx]);
            uint64_t mk = 0;
            if (rvalid) {
                // tap (j,k) is inside the image iff row j and column k are: ktap + ktap compares, no division
                uint64_t colm = 0;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < d.ktap && (unsigned)(xs0 + k) < (unsigned)S.Ws) colm |= 1ull << k;
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    if (j < d.ktap && (unsigned)(ys0 + j) < (unsigned)S.Hs) mk |= colm << (j * d.ktap);
            }
            // a lane whose channel chunk lies beyond a narrow source (C < 64) is never valid
            if (lchunk * 8 >= S.C) mk = 0;
            if (sidx == 0) { roff0[i] = ro; nv0[i] = ~(uint32_t)mk; } else { roff1[i] = ro; nv1[i] = ~(uint32_t)mk; }
            if constexpr (KT > 3) {
                if (sidx == 0) nv0h[i] = ~(uint32_t)(mk >> 32); else nv1h[i] = ~(uint32_t)(mk >> 32);
            }
        }
    }
    // weight panel rows: loop-invariant offsets, the K-step advances through soffset (rows >= N read zeros)
    uint32_t wvoff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int nrow = n0 + lrow0 + RS * i;
        wvoff[i] = nrow < d.N ? (uint32_t)(2 * (nrow * d.Ktot + lchunk * 8)) : OOB;
    }

    // K-step cursor of the NEXT load.  The panel's K axis is (tap, source, channel); the loop walks it CHANNEL-CHUNK MAJOR,
    // TAP MINOR: the taps of one 64-channel chunk are consecutive steps, so the nine shifted reads of the same activation
    // cache lines come back-to-back and hit in L2.  (Tap-major order re-read every activation line once per tap from
    // beyond L2: PMC FETCH_SIZE showed 9-14x the algorithmic bytes on the 3x3 layers, profiles/round1_notes.md.)
    // Linear step j -> chunk = j / taps (source s, channel offset c0), tap = j % taps; panel column block = tap*spt + chunk.
    const int ntaps = d.ktap * d.ktap;
    const int spt = (dv.kseg0 + dv.kseg1) / BK;          // 64-channel chunks per tap (both sources)
    const int s0steps = dv.kseg0 / BK;
    int tap, s, c0, chunk;
    {
        chunk = kstep_begin / ntaps;
        tap = kstep_begin - chunk * ntaps;
        s = (NSRC > 1 && chunk >= s0steps) ? 1 : 0;
        c0 = (s ? chunk - s0steps : chunk) * BK;
    }

    const int wrow_lds = wave * 8 * 128;               // this wave's first row in every RS-row group (bytes)

    auto issue_src = [&](unsigned char* X) {
        // NSRC == 2: the source of this K-step is wave-uniform (`s`), the per-row values are picked with scalar-condition
        // selects (never by pointing a shared body at one of two register arrays: that sends the arrays to scratch)
        const bool s1 = NSRC > 1 && s != 0;
        const uclstm_src S = d.src[s1 ? 1 : 0];
        const __amdgpu_buffer_rsrc_t rs = s1 ? rsx1 : rsx0;
        const int tdy = tap / d.ktap;
        const uint32_t tapoff = (uint32_t)(2 * ((tdy * S.Ws + (tap - tdy * d.ktap)) * S.C + c0));      // wave-uniform -> soffset
        const bool hiw = KT > 3 && tap >= 32;                    // wave-uniform: second mask word (7x7 kernels only)
        const uint32_t sh = 31u - (uint32_t)(KT > 3 ? (tap & 31) : tap);
        // channels beyond a source whose width is not a multiple of 64: only the last K-step of its segment can see them
        uint32_t cbad = 0;
        if ((S.C & 63) && c0 + 64 > S.C) cbad = (c0 + lchunk * 8 >= S.C) ? OOB : 0u;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const uint32_t ro = (NSRC > 1) ? (s1 ? roff1[i] : roff0[i]) : roff0[i];
            uint32_t nv = (NSRC > 1) ? (s1 ? nv1[i] : nv0[i]) : nv0[i];
            if constexpr (KT > 3) {
                const uint32_t nvh = (NSRC > 1) ? (s1 ? nv1h[i] : nv0h[i]) : nv0h[i];
                nv = hiw ? nvh : nv;
            }
            // bit `tap` of the outside-mask -> bit 31 of the offset: out of range, the DMA writes zeros
            const uint32_t off = (((nv << sh) & OOB) | ro) | cbad;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(X + i * RS * 128 + wrow_lds), 16, off, tapoff, 0, 0);
        }
    };
    auto issue_loads = [&](int buf) {
        unsigned char* X = smem + buf * SH::STAGE;
        unsigned char* Wt = X + SH::XBYTES;
        issue_src(X);
        const uint32_t koff = (uint32_t)(tap * spt + chunk) * (BK * 2);
#pragma unroll
        for (int i = 0; i < WR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(Wt + i * RS * 128 + wrow_lds), 16, wvoff[i], koff, 0, 0);
        // advance cursor: next tap of this chunk, then the next chunk (scalar selects, no branch)
        ++tap;
        const bool wrap = tap >= ntaps;
        tap = wrap ? 0 : tap;
        chunk += wrap ? 1 : 0;
        c0 += wrap ? BK : 0;
        if constexpr (NSRC > 1) {
            const bool next_src = wrap && s == 0 && chunk >= s0steps;
            s = next_src ? 1 : s;
            c0 = next_src ? 0 : c0;
        }
    };
    // One K-step.  Program order: fragments of the first 32-deep half, then (ISSUE) the address math + DMA of the NEXT
    // step, then the MFMAs -- the sched_group_barrier pattern asks the scheduler to slot those VALU/SALU/DMA
    // instructions into the shadows of the 16-cycle MFMAs instead of running them as a serial prologue of the step
    // (guide T19; in-order issue per wave means a serial prologue is pure MFMA idle time for this wave).
    auto step_body = [&](int buf, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        const unsigned char* X = smem + buf * SH::STAGE;
        const unsigned char* Wt = X + SH::XBYTES;
        act16x8 wf[4], xf[4];
        const int choff0 = ((0 * 4 + lq) ^ (l15 & 7)) << 4;
        const int choff1 = ((1 * 4 + lq) ^ (l15 & 7)) << 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[a] = *(const act16x8*)(Wt + (wc * 64 + a * 16 + l15) * 128 + choff0);
#pragma unroll
        for (int b = 0; b < 4; ++b) xf[b] = *(const act16x8*)(X + (wpx * 64 + b * 16 + l15) * 128 + choff0);
        if constexpr (ISSUE) issue_loads(buf ^ 1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = UCLSTM_MFMA_16x16x32(wf[a], xf[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[a] = *(const act16x8*)(Wt + (wc * 64 + a * 16 + l15) * 128 + choff1);
#pragma unroll
        for (int b = 0; b < 4; ++b) xf[b] = *(const act16x8*)(X + (wpx * 64 + b * 16 + l15) * 128 + choff1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = UCLSTM_MFMA_16x16x32(wf[a], xf[b], acc[a][b], 0, 0, 0);
        if constexpr (ISSUE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // 4 VALU
                __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);     // 3 SALU
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read (DMA)
            }
        }
    };

    const int nsteps = kstep_end - kstep_begin;      // >= 1 by construction of ksplit
    // ---- 2-stage: DMA of step s+1 is in flight while step s computes; __syncthreads() drains it (vmcnt(0)) ----
    issue_loads(0);
    __syncthreads();
    for (int step = 0; step + 1 < nsteps; ++step) {
        step_body(step & 1, std::true_type{});
        __syncthreads();
    }
    step_body((nsteps - 1) & 1, std::false_type{});
    __syncthreads();

    }   // per-tap loop (SHP 0 / 1)

    // ---- epilogue ----
    if constexpr (EPI == UCLSTM_EPI_LSTM) {
        const int hc = ((n0 >> 6) + wc) * 16 + lq * 4;     // first of this lane's 4 hidden channels
        if (hc < d.Hd_p) {
            float bg[4][4];
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) {
                const int nb = n0 + wc * 64 + gate * 16 + lq * 4;
                if (d.bias) {
                    const float4 t = *(const float4*)(d.bias + nb);
                    bg[gate][0] = t.x; bg[gate][1] = t.y; bg[gate][2] = t.z; bg[gate][3] = t.w;
                } else {
                    bg[gate][0] = bg[gate][1] = bg[gate][2] = bg[gate][3] = 0.f;
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int prow = wpx * 64 + b * 16 + l15;
                if (prow < rows_valid) {
                    const long pix = PIX(prow);
                    float cp[4] = {0.f, 0.f, 0.f, 0.f};
                    if (d.c_prev) {
                        const float4 t = *(const float4*)(d.c_prev + pix * d.Hd_p + hc);
                        cp[0] = t.x; cp[1] = t.y; cp[2] = t.z; cp[3] = t.w;
                    }
                    float cn[4];
                    Pack8 hi, gi, gf, gg, go;
                    float xa[4][4];                 // hoisted W_x * x_t of this pixel's four gate quads (or zeros)
#pragma unroll
                    for (int gate = 0; gate < 4; ++gate) {
                        const float4 t = d.pre_add ? *(const float4*)(d.pre_add + pix * d.N + n0 + wc * 64 + gate * 16 + lq * 4)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                        xa[gate][0] = t.x; xa[gate][1] = t.y; xa[gate][2] = t.z; xa[gate][3] = t.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float vi = fast_sigmoid(acc[0][b][r] + xa[0][r] + bg[0][r]);
                        const float vf = fast_sigmoid(acc[1][b][r] + xa[1][r] + bg[1][r]);
                        const float vg = fast_tanh(acc[2][b][r] + xa[2][r] + bg[2][r]);
                        const float vo = fast_sigmoid(acc[3][b][r] + xa[3][r] + bg[3][r]);
                        cn[r] = vf * cp[r] + vi * vg;                 // train/unet.py:34
                        hi.e[r] = f32_to_act(vo * fast_tanh(cn[r])); // train/unet.py:35
                        gi.e[r] = f32_to_act(vi);
                        gf.e[r] = f32_to_act(vf);
                        gg.e[r] = f32_to_act(vg);
                        go.e[r] = f32_to_act(vo);
                    }
                    *(float4*)(d.c_out + pix * d.Hd_p + hc) = make_float4(cn[0], cn[1], cn[2], cn[3]);
                    *(uint2*)((act16*)d.h_out + pix * d.Hd_p + hc) = hi.u;
                    if (d.gates_out) {
                        act16* gp = (act16*)d.gates_out + pix * 4 * d.Hd_p + hc;
                        *(uint2*)(gp) = gi.u;
                        *(uint2*)(gp + d.Hd_p) = gf.u;
                        *(uint2*)(gp + 2 * d.Hd_p) = gg.u;
                        *(uint2*)(gp + 3 * d.Hd_p) = go.u;
                    }
                }
            }
        }
    } else if constexpr (EPI == UCLSTM_EPI_ATOMIC) {
        // split-K partial tile, staged through LDS so that every wave instruction touches 64 consecutive floats of one pixel
        // row (256-byte runs): plain stores into this K range's slab (acc_slab > 0, the form the ConvLSTM path uses), or f32
        // atomic adds into one shared buffer (acc_slab == 0; ~4.6x slower per byte on this chip)
        constexpr int AP = TBN + 4;                      // floats per staged pixel row
        constexpr int RPI = NT / TBN;                    // pixel rows per sweep of the block
        float* At = (float*)smem;                        // [64 pixels][AP]
#pragma unroll
        for (int blk = 0; blk < SH::WM; ++blk) {
            if (wpx == blk) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        *(float4*)(At + (b * 16 + l15) * AP + wc * 64 + a * 16 + lq * 4) =
                            make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
            }
            __syncthreads();
            const int col = tid % TBN;
            const int n = n0 + col;
            if (n < d.N) {
                for (int pr = tid / TBN; pr < 64; pr += RPI) {
                    const int prow = blk * 64 + pr;
                    if (prow < rows_valid) {
                        float* dst = d.acc_out + (long)ks * d.acc_slab + PIX(prow) * (long)d.acc_ld + n;
                        if (d.acc_slab > 0) *dst = At[pr * AP + col];          // this K range's own slab: plain 256-byte runs
                        else __hip_atomic_fetch_add(dst, At[pr * AP + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        constexpr int OT_PITCH = SH::OT_PITCH;
        unsigned char* Ot = smem;    // [TBM pixels][OT_PITCH]; all K-loop LDS reads retired by the last barrier
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = wc * 64 + a * 16 + lq * 4;
            const int n = n0 + col;
            float bs[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < d.N) {
                if (d.bias) { const float4 t = *(const float4*)(d.bias + n); bs[0] = t.x; bs[1] = t.y; bs[2] = t.z; bs[3] = t.w; }
                if (d.col_scale) { const float4 t = *(const float4*)(d.col_scale + n); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
                if (d.col_shift) { const float4 t = *(const float4*)(d.col_shift + n); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }
            }
            float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int prow = wpx * 64 + b * 16 + l15;
                Pack8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = (acc[a][b][r] + bs[r]) * sc[r] + sh[r];
                    if (d.relu) v = fmaxf(v, 0.f);
                    o.e[r] = f32_to_act(v);
                    if constexpr (SHP == 2) {      // statistics of the ROUNDED values (what BatchNorm will read), from registers
                        const float q = prow < rows_valid ? act_to_f32(o.e[r]) : 0.f;
                        ps1[r] += q;
                        ps2[r] += q * q;
                    }
                }
                *(uint2*)(Ot + prow * OT_PITCH + col * 2) = o.u;
            }
            if constexpr (SHP == 2) {
                // with one block per CU nothing hides a serial 128-row LDS walk per channel: reduce over the wave's 64 pixels
                // with DPP row sums, leave one value per (wave, channel) in LDS for the two-wave add below
                if (d.stats) {
                    float* Red = (float*)(smem + TBM * OT_PITCH);       // [WM][TBN][2]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t1 = row16_sum(ps1[r]), t2 = row16_sum(ps2[r]);
                        if (l15 == 0) {
                            Red[(wpx * TBN + col + r) * 2] = t1;
                            Red[(wpx * TBN + col + r) * 2 + 1] = t2;
                        }
                    }
                }
            }
        }
        __syncthreads();

        // the patch shape's 256-pixel tile fills the statistics rows of the two 128-pixel tiles it covers, so the buffer
        // layout (uclstm_igemm_tiles_per_group) does not depend on which of the two kernels ran
        constexpr int HALVES = SHP == 2 ? 2 : 1;
        if (d.stats && tid < TBN * HALVES) {
            const int half = tid / TBN;
            const int col = tid - half * TBN;
            const int n = n0 + col;
            if (n < d.N) {
                float s1 = 0.f, s2 = 0.f;
                if constexpr (SHP == 2) {
                    const float* Red = (const float*)(smem + TBM * OT_PITCH);
                    s1 = Red[((2 * half) * TBN + col) * 2] + Red[((2 * half + 1) * TBN + col) * 2];
                    s2 = Red[((2 * half) * TBN + col) * 2 + 1] + Red[((2 * half + 1) * TBN + col) * 2 + 1];
                } else {
                    for (int r = 0; r < rows_valid; ++r) {
                        const float v = act_to_f32(*(const act16*)(Ot + r * OT_PITCH + col * 2));
                        s1 += v;
                        s2 += v * v;
                    }
                }
                float* sp = d.stats + ((long)(mt * HALVES + half) * d.N + n) * 2;
                sp[0] = s1;
                sp[1] = s2;
            }
        }

        constexpr int CPR = TBN / 8;                      // 16-byte chunks per staged pixel row
        for (int q = tid; q < TBM * CPR; q += NT) {
            const int r = q / CPR;
            const int cc = q - r * CPR;
            const int n = n0 + cc * 8;
            if (r >= rows_valid || n >= d.N) continue;
            int img, y, x;
            if (strip) {
                img = s_img;
                y = s_y0 + (r >> 6);
                x = s_x0 + (r & 63);
            } else {
                const uint32_t m = (uint32_t)(m0 + r);
                img = (int)fdiv(m, dv.dHW);
                const uint32_t rem = m - (uint32_t)img * (uint32_t)HW;
                y = (int)fdiv(rem, dv.dW);
                x = (int)rem - y * d.W;
            }
#pragma unroll
            for (int si = 0; si < 4; ++si) {
                if (si < d.nseg && n >= d.seg[si].n_begin && n < d.seg[si].n_end) {
                    const uclstm_seg sg = d.seg[si];
                    const int yd = y * sg.scale + sg.oy;
                    const int xd = x * sg.scale + sg.ox;
                    if ((unsigned)yd < (unsigned)sg.Hd && (unsigned)xd < (unsigned)sg.Wd) {
                        act16* dst = (act16*)sg.ptr + (((long)img * sg.Hd + yd) * sg.Wd + xd) * (long)sg.C + sg.c_off + (n - sg.n_begin);
                        *(uint4*)dst = *(const uint4*)(Ot + r * OT_PITCH + cc * 16);
                    }
                }
            }
        }
    }
#undef PIX
#endif
}

template <int EPI, int SHP, int NSRC, int KT = 3>
__global__ __launch_bounds__(Shape<SHP>::NT, Shape<SHP>::MINB) void igemm_fwd_kernel(const uclstm_igemm_desc d, const Derived dv) {
    igemm_fwd_body<EPI, SHP, NSRC, KT>(d, dv, (int)blockIdx.x);
}

// ---- several independent GEMMs of the patch shape in ONE launch ------------------------------------------------------
// The three ConvLSTMs of the model (bottleneck + two skip LSTMs, train/unet.py:185-191) are independent recurrences whose
// per-timestep GEMMs each fill the 256 CUs for one or two rounds only; launched one after the other every one of them pays
// its own drain and fill.  A group launch is one grid whose block ranges belong to different descriptors (members ordered by
// the host longest block first: the hardware hands out blocks in id order, i.e. longest-processing-time-first scheduling), each
// with its own epilogue (fused cell update or split-K slabs).  A member's first block id is a multiple of 8 so that its
// XCD-aware tile order is what a launch of its own would have.
constexpr int GROUP_MAX = 4;
struct FwdGroup {
    int n;
    int first[GROUP_MAX + 1];      // block ranges [first[i], first[i+1]); blocks beyond a member's own count do nothing
    int nblk[GROUP_MAX];
    uclstm_igemm_desc d[GROUP_MAX];
    Derived dv[GROUP_MAX];
};

template <int NSRC>
__global__ __launch_bounds__(Shape<2>::NT, 1) void igemm_fwd_group_kernel(const FwdGroup g) {
    const int b = (int)blockIdx.x;
    // constant member index in every access: all fields are read straight from the kernel-argument segment (a runtime index
    // into the by-value struct would send the whole 2-KiB argument to scratch memory)
#define UCLSTM_GROUP_MEMBER(j_)                                                                            \
    if (j_ < g.n && b >= g.first[j_] && b < g.first[j_ + 1]) {                                             \
        const int bid = b - g.first[j_];                                                                   \
        if (bid >= g.nblk[j_]) return;                                                                     \
        if (g.d[j_].epi == UCLSTM_EPI_LSTM) igemm_fwd_body<UCLSTM_EPI_LSTM, 2, NSRC>(g.d[j_], g.dv[j_], bid);   \
        else igemm_fwd_body<UCLSTM_EPI_ATOMIC, 2, NSRC>(g.d[j_], g.dv[j_], bid);                           \
        return;                                                                                            \
    }
    UCLSTM_GROUP_MEMBER(0)
    UCLSTM_GROUP_MEMBER(1)
    UCLSTM_GROUP_MEMBER(2)
    UCLSTM_GROUP_MEMBER(3)
#undef UCLSTM_GROUP_MEMBER
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 / pad 1 convolution with C_in = C_out = 64 on images whose width is a multiple of 64 (the full-resolution level of
// the UNet; at the 64x64 configurations: 2.6 M pixels, K = 576 -- nine K-steps per tile, where the generic kernel spends as long in its
// prologue/epilogue as in the loop and re-stages every activation row nine times through the LDS-DMA path: 545 TFLOP/s).
// Persistent blocks, one per CU, 4 waves:
//   * the whole weight panel (9 taps x 64 rows x 128 B = 72 KiB) stays resident in LDS;
//   * activations live in a RING of ten image rows (66 pixels x 128 B each: the two halo columns are permanent zeros,
//     rows above/below an image are zero-filled by out-of-range DMA); a tile is four image rows = 256 pixels, one row per
//     wave; it reads six ring rows, and while it computes, the four new rows of the next tile are DMA'd into the four
//     free slots -- every activation row is staged ONCE and all nine taps are shifted LDS reads;
//   * the tile loop body (288 MFMA per wave) has no barrier; two barriers per tile (rows landed, statistics);
//   * epilogue straight from registers: a lane holds 4 consecutive channels of a pixel, a wave store writes 32-byte
//     sector-aligned pieces; BatchNorm partial sums by 16-lane shuffles + a 2-KiB LDS exchange.
// Virtual row numbering: image k, row y -> v = k*(H+1) + y; v = k*(H+1) + H is the zero row shared by images k and k+1;
// ring slot of v = (v + 1) mod 10.

constexpr int C64_WBYTES = 9 * 64 * 128;            // 73728
constexpr int C64_ROW = 66 * 128;                   // 8448
constexpr int C64_SLOTS = 10;
constexpr int C64_RED = C64_WBYTES + C64_SLOTS * C64_ROW;      // 158208: [4 waves][64 channels][2] floats
constexpr int C64_SMEM = C64_RED + 4 * 64 * 2 * 4;            // 160256

__global__ __launch_bounds__(256, 1) void igemm_fwd_c64_kernel(const uclstm_igemm_desc d, const int tiles_total, const int tiles_per_block,
                                                                const uint32_t xbytes, const int tiles_per_group) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    unsigned char* Wl = smem;
    unsigned char* Ring = smem + C64_WBYTES;
    float* Red = (float*)(smem + C64_RED);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lq = lane >> 4;
    const int H = d.H;
    const int tiles_per_img = H >> 2;
    const int t_begin = blockIdx.x * tiles_per_block;
    const int t_end = min(tiles_total, t_begin + tiles_per_block);
    if (t_begin >= t_end) return;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[0].ptr, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.wp, 0, (uint32_t)(64 * 576 * 2), 0x00020000);

    // permanent zero halo columns (positions 0 and 65 of every ring row) -- only when the image is one strip wide: with more
    // strips every staged row brings its halo pixels by DMA (real pixels or out-of-range zeros), and a plain store here would
    // race with the first of those DMAs (no ordering between one wave's ds_write and another wave's LDS-DMA before the first
    // barrier; see igemm_wgrad_c64_kernel)
    if ((d.W >> 6) == 1 && tid < C64_SLOTS * 2 * 8) {
        const int sl = tid >> 4, side = (tid >> 3) & 1, ch = tid & 7;
        *(uint4*)(Ring + sl * C64_ROW + side * 65 * 128 + ch * 16) = make_uint4(0, 0, 0, 0);
    }
    // resident weight panel: instruction q covers tap q>>3, rows (q&7)*8 + (lane>>3); chunk position lane&7 <- chunk pos^(row&7)
    {
        const int srow = lane >> 3;
        const int sc = (lane & 7) ^ srow;
#pragma unroll
        for (int it = 0; it < 18; ++it) {
            const int q = it * 4 + wave;
            const int tp = q >> 3, nrow = (q & 7) * 8 + srow;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(Wl + q * 1024), 16, (uint32_t)(2 * (nrow * 576 + sc * 8)), (uint32_t)(tp * 128), 0,
                                                     0);
        }
    }
    // activation row staging: wave w moves pixels 8*(w + 4*i) + (lane>>3), i = 0,1, of a row; position p = x'+1 holds chunk pos^(p&7)
    uint32_t rvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int xp = 8 * (wave + 4 * i) + (lane >> 3);
        const int sc = (lane & 7) ^ ((xp + 1) & 7);
        rvoff[i] = (uint32_t)(2 * (xp * 64 + sc * 8));
    }
    // Images wider than 64 pixels are cut into 64-pixel column STRIPS; a "sequence" is one strip of one image (index
    // image*strips + strip), tiles walk down a sequence.  With more than one strip the halo columns are real pixels of the
    // neighbouring strip (or zero at the image border): waves 0 and 1 re-stage them with every row (8 pixels each, of which
    // the ring keeps one: the DMA granule is a whole wave).
    const int strips = d.W >> 6;
    const int n_seq = d.n_img * strips;
    const uint32_t px_bytes = 128u;                                  // 64 channels
    // cursor of the next virtual row to stage: (sequence lk, row ly in 0..H where H = the zero row), its ring slot.  Plain
    // locals advanced by a macro: as by-reference lambda captures they ended up in scratch memory.
    int lk = 0, ly = 0, lslot = 0, loaded = -2;
    int limg = 0, lstrip = 0;
#define C64_ISSUE_NEXT_ROW()                                                                                              \
    {                                                                                                                     \
        const bool zero_ = ly == H || lk < 0 || lk >= n_seq;                                                              \
        const uint32_t soff_ = zero_ ? 0u : (uint32_t)((limg * H + ly) * d.W + lstrip * 64) * px_bytes;                   \
        unsigned char* dst_ = Ring + lslot * C64_ROW + 128;                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst_ + wave * 1024), 16, zero_ ? OOB : rvoff[0], soff_, 0, 0);       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst_ + (wave + 4) * 1024), 16, zero_ ? OOB : rvoff[1], soff_, 0, 0); \
        if (strips > 1 && wave < 2) {                                                                                     \
            const bool ok_ = !zero_ && (wave == 0 ? lstrip > 0 : lstrip < strips - 1);                                    \
            const uint32_t hs_ = ok_ ? (wave == 0 ? soff_ - px_bytes : soff_ + 64u * px_bytes) : 0u;                      \
            unsigned char* hd_ = Ring + lslot * C64_ROW + (wave == 0 ? 0 : 65 * 128);                                     \
            if (lane < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)hd_, 16, ok_ ? halo_voff : OOB, hs_, 0, 0);          \
        }                                                                                                                 \
        ++loaded;                                                                                                         \
        lslot = lslot == C64_SLOTS - 1 ? 0 : lslot + 1;                                                                   \
        if (ly == H) {                                                                                                    \
            ly = 0;                                                                                                       \
            ++lk;                                                                                                         \
            if (++lstrip == strips) { lstrip = 0; ++limg; }                                                               \
        } else {                                                                                                          \
            ++ly;                                                                                                         \
        }                                                                                                                 \
    }
    // halo pixel: position 0 (left, wave 0) or 65 (right, wave 1) holds chunk (lane&7) ^ (position & 7)
    const uint32_t halo_voff = (uint32_t)(2 * ((((lane & 7) ^ (wave == 0 ? 0 : 1)) & 7) * 8));

    // fragment read offsets
    int xoff[4][3][2];          // [b][dx][ksub]: inside a ring row
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int pos = b * 16 + l15 + dx;
                xoff[b][dx][kk] = pos * 128 + (((kk * 4 + lq) ^ (pos & 7)) << 4);
            }
    int woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) woff[kk] = l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4);

    const uclstm_seg sg = d.seg[0];
    // epilogue constants of this lane's channels a*16 + lq*4 .. +3 (loop invariant)
    float bs[4][4], scl[4][4], sft[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = a * 16 + lq * 4;
        const float4 t0 = d.bias ? *(const float4*)(d.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 t1 = d.col_scale ? *(const float4*)(d.col_scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 t2 = d.col_shift ? *(const float4*)(d.col_shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        bs[a][0] = t0.x; bs[a][1] = t0.y; bs[a][2] = t0.z; bs[a][3] = t0.w;
        scl[a][0] = t1.x; scl[a][1] = t1.y; scl[a][2] = t1.z; scl[a][3] = t1.w;
        sft[a][0] = t2.x; sft[a][1] = t2.y; sft[a][2] = t2.z; sft[a][3] = t2.w;
    }
    // BatchNorm partial sums: a lane keeps RUNNING sums of its channels over the block's consecutive tiles of one statistic
    // group; only at the end of the run (group boundary or end of the block) are they reduced across lanes and waves and
    // written into that tile's row -- the other tiles of the run get zero rows (every row of the partial-sum buffer must be
    // written: the reduction kernel adds all rows of a group).  The cross-lane reduction per tile cost as much as the MFMAs.
    float gs1[4][4], gs2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) gs1[a][r] = gs2[a][r] = 0.f;
    int k = t_begin / tiles_per_img;
    int tr = t_begin - k * tiles_per_img;
    for (int tt = t_begin; tt < t_end; ++tt) {
        const int v0 = k * (H + 1) + 4 * tr;
        if (tt == t_begin) {      // cursor at virtual row v0 - 1
            const int v = v0 - 1;
            if (v < 0) { lk = -1; ly = H; limg = 0; lstrip = -1; } else { lk = v / (H + 1); ly = v - lk * (H + 1); limg = lk / strips; lstrip = lk - limg * strips; }
            lslot = (v + 1) % C64_SLOTS;
            loaded = v - 1;
        }
        // rows this tile needs that were not prefetched (six at the start of the block, one after an image boundary).  The one
        // after a boundary lands in the slot of the PREVIOUS tile's first row, which a slower wave may still be reading (there
        // is no barrier inside a tile, and with another kernel sharing the CU the waves drift): wait for everybody first.  The
        // condition is block-uniform; it holds once per image strip.
        if (tt != t_begin && loaded < v0 + 4) __syncthreads();
        while (loaded < v0 + 4) C64_ISSUE_NEXT_ROW()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // prefetch for the next tile into the free slots (at most four rows: ten slots minus the six in use)
        int k1 = k, tr1 = tr + 1;
        if (tr1 == tiles_per_img) { tr1 = 0; ++k1; }
        if (tt + 1 < t_end) {
            const int upto = min(k1 * (H + 1) + 4 * tr1 + 4, loaded + 4);
            while (loaded < upto) C64_ISSUE_NEXT_ROW()
        }

        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // 18 steps (tap, 32-channel half), software-pipelined by hand: the fragments of step s+1 are read while the 16 MFMAs of
        // step s issue (one wave per SIMD: nobody else hides an LDS round trip), one ds_read slotted after each of the first
        // eight MFMAs of a step
        const unsigned char* rowp[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rowp[dy] = Ring + ((v0 + wave + dy) % C64_SLOTS) * C64_ROW;      // virtual row v0 + wave + dy - 1
        act16x8 wf[2][4], xf[2][4];
        auto load_step = [&](int st, int set) {
            const int tp = st >> 1, kk = st & 1;
            const int dy = tp / 3, dx = tp - dy * 3;
#pragma unroll
            for (int a = 0; a < 4; ++a) wf[set][a] = *(const act16x8*)(Wl + tp * 8192 + a * 2048 + woff[kk]);
#pragma unroll
            for (int b = 0; b < 4; ++b) xf[set][b] = *(const act16x8*)(rowp[dy] + xoff[b][dx][kk]);
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) load_step(st + 1, (st + 1) & 1);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = UCLSTM_MFMA_16x16x32(wf[st & 1][a], xf[st & 1][b], acc[a][b], 0, 0, 0);
            if (st + 1 < 18) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);         // the other 8 MFMAs
            }
        }

        // ---- epilogue from registers: wave = image row y, lane = (pixel b*16 + l15, channels a*16 + lq*4 .. +3) ----
        const int y = 4 * tr + wave;
        const int img = k / strips, strip = k - img * strips;
        act16* orow = (act16*)sg.ptr + ((long)(img * H + y) * d.W + strip * 64) * (long)sg.C + sg.c_off;
        float s1[4][4], s2[4][4];
        Pack8 ov[4][4];           // [a][b]: this lane's 4 channels of pixel b*16 + l15, act16
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[a][r] = s2[a][r] = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = (acc[a][b][r] + bs[a][r]) * scl[a][r] + sft[a][r];
                    if (d.relu) v = fmaxf(v, 0.f);
                    ov[a][b].e[r] = f32_to_act(v);
                    const float q = act_to_f32(ov[a][b].e[r]);
                    s1[a][r] += q;
                    s2[a][r] += q * q;
                }
            }
        }
        // 16-byte stores (guide T21: a row-per-lane epilogue of 8-byte stores is store-ISSUE bound): v_permlane16_swap trades
        // quads between lane l (even 16-lane row, lq = 0/2) and l+16 (odd row) of the same pixel, so that the even lane
        // holds 8 consecutive channels of row tile a and the odd lane 8 consecutive channels of row tile a+1.
        const int nst0 = (lq & 1) ? 16 + (lq - 1) * 4 : lq * 4;       // + a*16 for the pair (a, a+1)
#pragma unroll
        for (int a = 0; a < 4; a += 2)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const auto p0 = __builtin_amdgcn_permlane16_swap(ov[a][b].u.x, ov[a + 1][b].u.x, false, false);
                const auto p1 = __builtin_amdgcn_permlane16_swap(ov[a][b].u.y, ov[a + 1][b].u.y, false, false);
                const int n = a * 16 + nst0;
                if (n < sg.n_end) *(uint4*)(orow + (long)(b * 16 + l15) * sg.C + n) = make_uint4(p0[0], p1[0], p0[1], p1[1]);
            }
        if (d.stats) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gs1[a][r] += s1[a][r];
                    gs2[a][r] += s2[a][r];
                }
            const bool run_ends = tt + 1 == t_end || (tt + 1) / tiles_per_group != tt / tiles_per_group;      // block-uniform
            if (!run_ends) {
                if (tid < 128) d.stats[(long)tt * d.N * 2 + tid] = 0.f;
            } else {
                // sum over the 16 pixels of a lane row, then over the 4 waves through LDS
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gs1[a][r] = row16_sum(gs1[a][r]);
                        gs2[a][r] = row16_sum(gs2[a][r]);
                    }
                if (l15 == 0) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = a * 16 + lq * 4 + r;
                            Red[(wave * 64 + n) * 2] = gs1[a][r];
                            Red[(wave * 64 + n) * 2 + 1] = gs2[a][r];
                        }
                }
                __syncthreads();
                if (tid < 128) {
                    const int n = tid >> 1, j = tid & 1;
                    d.stats[((long)tt * d.N + n) * 2 + j] =
                        Red[(0 * 64 + n) * 2 + j] + Red[(1 * 64 + n) * 2 + j] + Red[(2 * 64 + n) * 2 + j] + Red[(3 * 64 + n) * 2 + j];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gs1[a][r] = gs2[a][r] = 0.f;
                // (Red is rewritten only after the next run's tiles, each of which starts with a barrier)
            }
        }
        k = k1;
        tr = tr1;
    }
#endif
}

#undef C64_ISSUE_NEXT_ROW

// The launch conditions of igemm_fwd_c64_kernel (everything else takes the generic kernel).
inline bool c64_ok(const uclstm_igemm_desc& d) {
    static const bool off = [] { const char* e = getenv("UCLSTM_FWD_C64"); return e && e[0] == '0'; }();
    if (off || d.epi != UCLSTM_EPI_STORE || d.nsrc != 1 || d.ktap != 3 || d.pad != 1 || d.scale != 1) return false;
    const uclstm_src& S = d.src[0];
    if (S.C != 64 || d.N != 64 || d.Ktot != 576 || (d.W & 63) || (d.H & 3) || S.Hs != d.H || S.Ws != d.W || S.offY || S.offX) return false;
    if (d.nseg != 1) return false;
    const uclstm_seg& g = d.seg[0];
    if (g.scale != 1 || g.oy || g.ox || g.Hd != d.H || g.Wd != d.W || g.n_begin != 0 || g.n_end > 64 || (g.n_end % 8) || (g.C % 8) ||
        (g.c_off % 8))
        return false;
    if ((int64_t)d.n_img * d.H * d.W * 64 * 2 >= ((int64_t)1 << 31) - (1 << 22)) return false;
    if ((int64_t)d.n_img * (d.W / 64) * (d.H + 1) >= ((int64_t)1 << 30)) return false;
    return true;
}

bool src_ok(const uclstm_src& s) {
    return s.ptr && s.C > 0 && (s.C % 8) == 0 && s.Hs > 0 && s.Ws > 0 && ((uintptr_t)s.ptr % 16) == 0;
}

// Block shape for a launch (also fixes the row count of the BatchNorm partial-sum buffer, so it depends only on what
// uclstm_igemm_tiles_per_group is told): narrow panels -> 64x256, else 128x128.
inline int pick_shape(int N, int64_t /*mg*/, int /*groups*/, int epi) {
    if (epi == UCLSTM_EPI_LSTM) return 0;
    return N <= 64 ? 1 : 0;
}
inline int shape_pixels(int shp) { return shp == 0 ? 128 : 256; }
inline int shape_rows(int shp) { return shp == 1 ? 64 : 128; }

// Rows a 256-pixel tile's patch can span in the shared-zero padded index space (see the patch K loop): 255 steps between
// its first and last pixel, +1 per image-row boundary crossed, +(W+2) per image boundary crossed, plus the reach of the
// corner taps on either side.
inline int patch_rows_max(int H, int W, int64_t mg) {
    const int64_t HW = (int64_t)H * W;
    int64_t R, I;
    if (mg % 256 == 0 && 256 % W == 0 && (HW % 256 == 0 || 256 % HW == 0)) {      // tiles start on image-row boundaries
        R = 256 / W - 1;
        I = HW >= 256 ? 0 : 256 / HW - 1;
    } else {
        R = (254 + W) / W;
        I = (254 + HW) / HW;
    }
    return (int)(255 + (R - I) + (W + 2) * I + 2 * (W + 2) + 1);
}

// The launch conditions of the patch shape (SHP 2); everything else takes the per-tap loop.  Returns 0 (no), 1 (tiles of 256
// consecutive pixels: images up to 64 wide) or 2 (4-row x 64-column strip tiles: wider images with 64 | W and 4 | H).
inline int patch_ok(const uclstm_igemm_desc& d, int64_t mg) {
    static const bool off = [] { const char* e = getenv("UCLSTM_FWD_PATCH"); return e && e[0] == '0'; }();
    static const bool strip_off = [] { const char* e = getenv("UCLSTM_FWD_STRIP"); return e && e[0] == '0'; }();
    if (off || d.ktap != 3 || d.pad != 1 || d.scale != 1 || d.N <= 64 || (mg % 256)) return 0;
    if (d.Ktot < 2 * 9 * BK) return 0;       // one chunk: nothing to amortise the patch over (K = 576 measured 11 % slower)
    for (int s = 0; s < d.nsrc; ++s) {
        const uclstm_src& S = d.src[s];
        if ((S.C % 64) || S.Hs != d.H || S.Ws != d.W || S.offY || S.offX) return 0;
    }
    if ((int64_t)d.n_img * (d.H + 1) * (d.W + 1) >= ((int64_t)1 << 30)) return 0;
    if (patch_rows_max(d.H, d.W, mg) <= Shape<2>::PROWS) return 1;
    if (!strip_off && (d.W % 64) == 0 && (d.H % 4) == 0) return 2;
    return 0;
}

template <int EPI, int SHP, int NSRC, int KT = 3>
int32_t launch_n(const uclstm_igemm_desc& d, const Derived& dv, int64_t nblk, hipStream_t st) {
    static bool attr_done = false;          // per process: one process per GPU
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_fwd_kernel<EPI, SHP, NSRC, KT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   Shape<SHP>::SMEM);
        attr_done = true;
    }
    UCLSTM_LAUNCH((igemm_fwd_kernel<EPI, SHP, NSRC, KT>), dim3((unsigned)nblk), dim3(Shape<SHP>::NT), Shape<SHP>::SMEM, st, d, dv);
    return UCLSTM_OK;
}
template <int EPI, int SHP>
int32_t launch(const uclstm_igemm_desc& d, const Derived& dv, int64_t nblk, hipStream_t st) {
    if constexpr (SHP == 0) {
        if (d.ktap > 3) return d.nsrc == 1 ? launch_n<EPI, 0, 1, 7>(d, dv, nblk, st) : launch_n<EPI, 0, 2, 7>(d, dv, nblk, st);
    }
    return d.nsrc == 1 ? launch_n<EPI, SHP, 1>(d, dv, nblk, st) : launch_n<EPI, SHP, 2>(d, dv, nblk, st);
}

}  // namespace

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_igemm_tiles_per_group(int32_t n_img, int32_t H, int32_t W, int32_t groups, int32_t N) {
    if (groups <= 0 || n_img <= 0 || n_img % groups || N <= 0) return UCLSTM_E_BADARG;
    const int64_t mg = (int64_t)(n_img / groups) * H * W;
    const int bm = shape_pixels(pick_shape(N, mg, groups, UCLSTM_EPI_STORE));
    return (int32_t)((mg + bm - 1) / bm);
}
#endif

// K ranges of a split-K launch are whole (source, 64-channel) CHUNKS -- ktap*ktap consecutive K-steps -- so that every range
// can run the patch loop (a chunk's activations are staged once for all its taps): steps per range = taps * ceil(chunks / ksplit).
static inline int ksplit_kper(int ksteps, int taps, int ksplit) {
    const int chunks = ksteps / taps;
    return taps * ((chunks + ksplit - 1) / ksplit);
}
#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_igemm_ksplit_used(int32_t Ktot, int32_t ktap, int32_t ksplit) {
    if (Ktot <= 0 || (Ktot % BK) || ksplit < 1 || ktap < 1 || ktap > 7 || (Ktot / BK) % (ktap * ktap)) return UCLSTM_E_BADARG;
    const int ksteps = Ktot / BK;
    const int kper = ksplit_kper(ksteps, ktap * ktap, ksplit);
    return (ksteps + kper - 1) / kper;
}
#endif

// Validation + launch plan shared by uclstm_igemm_fwd and uclstm_igemm_fwd_shape: derived constants, block shape, grid.
static int32_t plan_fwd(const uclstm_igemm_desc& d, Derived& dv, int& shp, int64_t& nblk, int64_t& mg_out) {
    if (d.n_img <= 0 || d.H <= 0 || d.W <= 0 || d.groups <= 0 || d.n_img % d.groups) return UCLSTM_E_BADARG;
    if (d.ktap < 1 || d.ktap > 7 || d.scale < 1 || d.scale > 2 || d.pad < 0 || d.pad > 3) return UCLSTM_E_BADARG;
    if (d.nsrc < 1 || d.nsrc > 2 || !d.wp || d.N <= 0 || (d.N % 8)) return UCLSTM_E_BADARG;
    if (d.ktap > 3 && d.epi == UCLSTM_EPI_STORE && d.stats) return UCLSTM_E_BADARG;      // statistics rows are laid out for the 3x3 shapes
    for (int s = 0; s < d.nsrc; ++s)
        if (!src_ok(d.src[s])) return UCLSTM_E_BADARG;
    dv = Derived{};
    const int64_t lim = ((int64_t)1 << 31) - (1 << 22);      // descriptors are addressed with bit 31 = "out of range"
    for (int s = 0; s < d.nsrc; ++s) {
        const uclstm_src& S = d.src[s];
        const int64_t by = d.pad + S.offY, bx = d.pad + S.offX;
        const int64_t bias = 2 * ((by > 0 ? by : 0) * S.Ws + (bx > 0 ? bx : 0) + 1) * S.C;
        const int64_t bytes = (int64_t)d.n_img * S.Hs * S.Ws * S.C * 2 + bias;
        if (bytes >= lim) return UCLSTM_E_BADARG;
        dv.xbias[s] = (uint32_t)bias;
        dv.xbytes[s] = (uint32_t)bytes;
    }
    if ((int64_t)d.N * d.Ktot * 2 >= lim) return UCLSTM_E_BADARG;
    dv.wbytes = (uint32_t)((int64_t)d.N * d.Ktot * 2);
    dv.kseg0 = round_up32(d.src[0].C, BK);
    dv.kseg1 = d.nsrc > 1 ? round_up32(d.src[1].C, BK) : 0;
    const int taps = d.ktap * d.ktap;
    if (d.Ktot != taps * (dv.kseg0 + dv.kseg1)) return UCLSTM_E_BADARG;
    dv.ksteps = d.Ktot / BK;
    const int64_t mg = (int64_t)(d.n_img / d.groups) * d.H * d.W;
    if (mg * d.groups >= ((int64_t)1 << 31)) return UCLSTM_E_BADARG;
    dv.dHW = make_fastdiv((uint32_t)(d.H * d.W));
    dv.dW = make_fastdiv((uint32_t)d.W);
    shp = d.ktap > 3 ? 0 : pick_shape(d.N, mg, d.groups, d.epi);      // kernels wider than 3x3: the 128 x 128 shape only
    mg_out = mg;
    int patch = patch_ok(d, mg);
    if (d.epi == UCLSTM_EPI_ATOMIC && d.ksplit < 1) return UCLSTM_E_BADARG;      // (K ranges are whole chunks: ksplit_kper)
    if (patch) shp = 2;
    dv.strip = patch == 2;
    dv.strips = d.W / 64;
    dv.tiles_per_img = (d.H / 4) * (d.W / 64);
    dv.dPHW = make_fastdiv((uint32_t)((d.H + 1) * (d.W + 1)));
    dv.dW2 = make_fastdiv((uint32_t)(d.W + 1));
    const int bm = shape_pixels(shp), bn = shape_rows(shp);
    dv.Mg = (int)mg;
    dv.tpg = (int)((mg + bm - 1) / bm);
    dv.n_mtiles = d.groups * dv.tpg;
    dv.n_ntiles = (d.N + bn - 1) / bn;
    dv.ksplit = 1;
    dv.kper = dv.ksteps;
    if (d.epi == UCLSTM_EPI_ATOMIC) {
        if (!d.acc_out || d.acc_ld < d.N || d.ksplit < 1 || d.acc_slab < 0) return UCLSTM_E_BADARG;
        if (d.acc_slab > 0 && d.acc_slab < mg * d.groups * (int64_t)d.acc_ld) return UCLSTM_E_BADARG;
        dv.kper = ksplit_kper(dv.ksteps, taps, d.ksplit);
        dv.ksplit = (dv.ksteps + dv.kper - 1) / dv.kper;      // every K range is non-empty
    }
    nblk = (int64_t)dv.n_mtiles * dv.n_ntiles * dv.ksplit;
    if (nblk <= 0 || nblk > 0x7fffffff) return UCLSTM_E_BADARG;

    if (d.epi == UCLSTM_EPI_ATOMIC) {
        // validated above
    } else if (d.epi == UCLSTM_EPI_LSTM) {
        if (d.Hd_p <= 0 || (d.Hd_p % 8) || (d.N % 64) || d.N != 64 * ((d.Hd_p + 15) / 16)) return UCLSTM_E_BADARG;
        if (!d.c_out || !d.h_out || (d.pre_add && ((uintptr_t)d.pre_add % 16))) return UCLSTM_E_BADARG;
    } else if (d.epi == UCLSTM_EPI_STORE) {
        if (d.nseg < 1 || d.nseg > 4) return UCLSTM_E_BADARG;
        for (int i = 0; i < d.nseg; ++i) {
            const uclstm_seg& sg = d.seg[i];
            if (!sg.ptr || (sg.n_begin % 8) || (sg.n_end % 8) || sg.n_end <= sg.n_begin || (sg.C % 8) || (sg.c_off % 8) ||
                sg.c_off + (sg.n_end - sg.n_begin) > sg.C || sg.Hd <= 0 || sg.Wd <= 0 || sg.scale < 1 ||
                ((uintptr_t)sg.ptr % 16))
                return UCLSTM_E_BADARG;
        }
    } else {
        return UCLSTM_E_BADARG;
    }

    return UCLSTM_OK;
}

#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_igemm_fwd_shape(const uclstm_igemm_desc* dp) {
    if (!dp) return UCLSTM_E_BADARG;
    Derived dv;
    int shp = 0;
    int64_t nblk = 0, mg = 0;
    const int32_t rc = plan_fwd(*dp, dv, shp, nblk, mg);
    if (rc != UCLSTM_OK) return rc;
    return (c64_ok(*dp) && mg % 256 == 0) ? 3 : shp;
}
#endif

extern "C" int32_t uclstm_igemm_fwd(const uclstm_igemm_desc* dp, void* stream) {
    if (!dp) return UCLSTM_E_BADARG;
    const uclstm_igemm_desc& d = *dp;
    Derived dv;
    int shp = 0;
    int64_t nblk = 0, mg = 0;
    const int32_t rc = plan_fwd(d, dv, shp, nblk, mg);
    if (rc != UCLSTM_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (c64_ok(d) && mg % 256 == 0) {        // its 256-pixel tiles are the generic 64x256 shape's tiles: same statistics rows
        static bool attr64 = false;
        if (!attr64) {
            (void)hipFuncSetAttribute((const void*)igemm_fwd_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C64_SMEM);
            attr64 = true;
        }
        const int tiles_total = d.n_img * (d.W / 64) * (d.H / 4);
        const int blocks = tiles_total < 256 ? tiles_total : 256;
        const int per = (tiles_total + blocks - 1) / blocks;
        const int grid = (tiles_total + per - 1) / per;
        UCLSTM_LAUNCH(igemm_fwd_c64_kernel, dim3(grid), dim3(256), C64_SMEM, st, d, tiles_total, per,
                      (uint32_t)((int64_t)d.n_img * d.H * d.W * 64 * 2), dv.tpg);
        return UCLSTM_OK;
    }
    if (d.epi == UCLSTM_EPI_LSTM) return shp == 2 ? launch<UCLSTM_EPI_LSTM, 2>(d, dv, nblk, st) : launch<UCLSTM_EPI_LSTM, 0>(d, dv, nblk, st);
    if (d.epi == UCLSTM_EPI_ATOMIC)
        return shp == 2 ? launch<UCLSTM_EPI_ATOMIC, 2>(d, dv, nblk, st)
                        : shp == 1 ? launch<UCLSTM_EPI_ATOMIC, 1>(d, dv, nblk, st) : launch<UCLSTM_EPI_ATOMIC, 0>(d, dv, nblk, st);
    if (shp == 2) return launch<UCLSTM_EPI_STORE, 2>(d, dv, nblk, st);
    if (shp == 1) return launch<UCLSTM_EPI_STORE, 1>(d, dv, nblk, st);
    return launch<UCLSTM_EPI_STORE, 0>(d, dv, nblk, st);
}

// Several independent GEMMs in one launch (igemm_fwd_group_kernel).  Every member must be a descriptor that uclstm_igemm_fwd
// would run on the patch shape (uclstm_igemm_fwd_shape == 2) with the fused-cell or the split-K epilogue, and all members must
// have the same number of sources.  query != 0: validate only and return the block count of the launch.
static int32_t fwd_group_run(const uclstm_igemm_desc* descs, int32_t n, void* stream, int query) {
    if (!descs || n < 1 || n > GROUP_MAX) return UCLSTM_E_BADARG;
    FwdGroup g{};
    int64_t work[GROUP_MAX];
    int order[GROUP_MAX];
    Derived dvs[GROUP_MAX];
    int64_t nblks[GROUP_MAX];
    for (int i = 0; i < n; ++i) {
        const uclstm_igemm_desc& d = descs[i];
        int shp = 0;
        int64_t mg = 0;
        const int32_t rc = plan_fwd(d, dvs[i], shp, nblks[i], mg);
        if (rc != UCLSTM_OK) return rc;
        if (shp != 2 || (c64_ok(d) && mg % 256 == 0)) return UCLSTM_E_BADARG;
        if (d.epi != UCLSTM_EPI_LSTM && d.epi != UCLSTM_EPI_ATOMIC) return UCLSTM_E_BADARG;
        if (d.nsrc != descs[0].nsrc) return UCLSTM_E_BADARG;
        work[i] = dvs[i].kper;              // K-steps per block: longest blocks get the lowest block ids
        order[i] = i;
    }
    for (int a = 1; a < n; ++a)             // insertion sort, stable, descending work
        for (int b = a; b > 0 && work[order[b]] > work[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    g.n = n;
    int64_t at = 0;
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        g.first[k] = (int)at;
        g.nblk[k] = (int)nblks[i];
        g.d[k] = descs[i];
        g.dv[k] = dvs[i];
        at += (nblks[i] + 7) / 8 * 8;
        if (at > 0x7fffffff) return UCLSTM_E_BADARG;
    }
    for (int k = n; k <= GROUP_MAX; ++k) g.first[k] = (int)at;
    if (query) return (int32_t)at;
    hipStream_t st = (hipStream_t)stream;
    if (descs[0].nsrc == 1) {
        static bool a1 = false;
        if (!a1) { (void)hipFuncSetAttribute((const void*)igemm_fwd_group_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, Shape<2>::SMEM); a1 = true; }
        UCLSTM_LAUNCH((igemm_fwd_group_kernel<1>), dim3((unsigned)at), dim3(Shape<2>::NT), Shape<2>::SMEM, st, g);
    } else {
        static bool a2 = false;
        if (!a2) { (void)hipFuncSetAttribute((const void*)igemm_fwd_group_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, Shape<2>::SMEM); a2 = true; }
        UCLSTM_LAUNCH((igemm_fwd_group_kernel<2>), dim3((unsigned)at), dim3(Shape<2>::NT), Shape<2>::SMEM, st, g);
    }
    return UCLSTM_OK;
}
extern "C" int32_t uclstm_igemm_fwd_group(const uclstm_igemm_desc* descs, int32_t n, void* stream) { return fwd_group_run(descs, n, stream, 0); }
#ifndef UCLSTM_ACT_F16
extern "C" int32_t uclstm_igemm_fwd_group_blocks(const uclstm_igemm_desc* descs, int32_t n) { return fwd_group_run(descs, n, nullptr, 1); }
#endif
