// Probe: 256x256x64 bf16 GEMM, 8 waves, LDS-DMA staging in half-tiles with counted vmcnt across raw barriers, two wave
// groups staggered by one barrier (the "8-phase" structure of cdna_hip_programming.md section 5).  C = A[M][K] * B[N][K]^T.
// Purpose: measure what this structure reaches on this machine before building the weight-gradient / split-K kernels on it.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/_bin/gemm256 tools/probes/gemm256_8phase.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int HALF = 128 * 128;          // bytes: 128 rows x 64 k x 2 B
constexpr int BUF = 4 * HALF;            // A-lo, B-lo, B-hi, A-hi  (consumption order)
constexpr int SMEM = 2 * BUF;            // 128 KiB

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int VARIANT>
__global__ __launch_bounds__(512, 1) void gemm256(const bf16* __restrict__ A, const bf16* __restrict__ B, float* __restrict__ C, int M,
                                                   int N, int K) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;

    const int ntn = N / 256;
    const int lid = xcd_remap(blockIdx.x, (M / 256) * ntn);
    const int tm = lid / ntn, tn = lid - tm * ntn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int KT = K / 64;
    const int total_halves = 4 * KT;

    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (uint32_t)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (uint32_t)((size_t)N * K * 2), 0x00020000);

    // staging role: instruction i of wave w fills rows (8i + w)*8 + (lane>>3) of a half, position lane&7 <- chunk pos ^ (row&7)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;            // row & 7 == lane>>3 for every instruction
    uint32_t voffA[2][2], voffB[2][2];               // [half][instr]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = h * 128 + (8 * i + wave) * 8 + srow;
            voffA[h][i] = (uint32_t)(((m0 + row) * K + schunk * 8) * 2);
            voffB[h][i] = (uint32_t)(((n0 + row) * K + schunk * 8) * 2);
        }

    // half index q -> (k-tile, slot): slot 0 A-lo, 1 B-lo, 2 B-hi, 3 A-hi
    auto issue_half = [&](int kt, int slot) {
        unsigned char* dst = smem + (kt & 1) * BUF + slot * HALF;
        const uint32_t soff = (uint32_t)kt * 128u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_ptr d = (lds_ptr)(dst + (8 * i + wave) * 1024);
            if (slot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, d, 16, voffA[0][i], soff, 0, 0);
            else if (slot == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, d, 16, voffB[0][i], soff, 0, 0);
            else if (slot == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, d, 16, voffB[1][i], soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, d, 16, voffA[1][i], soff, 0, 0);
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

    // fragment read offsets inside a half
    const int rsw = l15 & 7;
    int aoff[4], boff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = (wr * 64 + i * 16 + l15) * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[j] = (wc * 32 + j * 16 + l15) * 128;
    const int ch0 = ((0 * 4 + lq) ^ rsw) << 4, ch1 = ((1 * 4 + lq) ^ rsw) << 4;

    bf16x8 af[4][2], b0f[2][2], b1f[2][2];

    constexpr int LEAD = (VARIANT == 3) ? 7 : 6;
    constexpr bool STAGGER = VARIANT != 2;
    // prologue
#pragma unroll
    for (int q = 0; q < LEAD; ++q)
        if (q < total_halves) issue_half(q >> 2, q & 3);
    if (LEAD == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // halves 0,1 landed (this wave's part)
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();            // stagger: group 1 runs one barrier behind
    __builtin_amdgcn_s_barrier();

    int g = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* base = smem + (kt & 1) * BUF;
#pragma unroll
        for (int p = 0; p < 4; ++p, ++g) {
            // ---- load section: fragment reads of this phase, one half-tile prefetch, counted wait ----
            if (p == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b0f[j][0] = *(const bf16x8*)(base + 1 * HALF + boff[j] + ch0);
                    b0f[j][1] = *(const bf16x8*)(base + 1 * HALF + boff[j] + ch1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i][0] = *(const bf16x8*)(base + 0 * HALF + aoff[i] + ch0);
                    af[i][1] = *(const bf16x8*)(base + 0 * HALF + aoff[i] + ch1);
                }
            } else if (p == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b1f[j][0] = *(const bf16x8*)(base + 2 * HALF + boff[j] + ch0);
                    b1f[j][1] = *(const bf16x8*)(base + 2 * HALF + boff[j] + ch1);
                }
            } else if (p == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i][0] = *(const bf16x8*)(base + 3 * HALF + aoff[i] + ch0);
                    af[i][1] = *(const bf16x8*)(base + 3 * HALF + aoff[i] + ch1);
                }
            }
            if (g + LEAD < total_halves) {
                issue_half(kt + ((p + LEAD) >> 2), (p + LEAD) & 3);
                if (LEAD == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA section: one quadrant x K=64 ----
            if (VARIANT == 0) __builtin_amdgcn_s_setprio(1);
            {
                constexpr int dummy = 0;
                (void)dummy;
                const int mh = (p >= 2) ? 1 : 0;
                const int nh = (p == 1 || p == 2) ? 1 : 0;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const bf16x8 bb = nh ? b1f[j][s] : b0f[j][s];
                            acc[mh][i][nh][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bb, acc[mh][i][nh][j], 0, 0, 0);
                        }
            }
            if (VARIANT == 0) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();            // balance the stagger

    // epilogue: plain f32 stores (probe only)
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + nh * 128 + wc * 32 + j * 16 + l15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + mh * 128 + wr * 64 + i * 16 + lq * 4 + r;
                        C[(size_t)row * N + col] = acc[mh][i][nh][j][r];
                    }
                }
#endif
}

static float bf2f(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

template <int V>
static double run(const bf16* dA, const bf16* dB, float* dC, int M, int N, int K, int iters) {
    (void)hipFuncSetAttribute((const void*)gemm256<V>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int nblk = (M / 256) * (N / 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) gemm256<V><<<nblk, 512, SMEM>>>(dA, dB, dC, M, N, K);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) gemm256<V><<<nblk, 512, SMEM>>>(dA, dB, dC, M, N, K);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    bf16 *dA, *dB;
    float* dC;
    (void)hipMalloc(&dA, hA.size() * 2);
    (void)hipMalloc(&dB, hB.size() * 2);
    (void)hipMalloc(&dC, (size_t)M * N * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * N * K;
    const double ms0 = run<0>(dA, dB, dC, M, N, K, 20);
    // check a sample of entries
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
        s = s * 1664525u + 1013904223u;
        const int r = (s >> 8) % M;
        s = s * 1664525u + 1013904223u;
        const int c = (s >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
        const double e = fabs(ref - hC[(size_t)r * N + c]) / (1.0 + fabs(ref));
        if (e > maxerr) maxerr = e;
    }
    const double ms1 = run<1>(dA, dB, dC, M, N, K, 20);
    const double ms2 = run<2>(dA, dB, dC, M, N, K, 20);
    const double ms3 = run<3>(dA, dB, dC, M, N, K, 20);
    printf("M=%d N=%d K=%d  base: %.3f ms %.0f TF/s | no-setprio %.0f | no-stagger %.0f | lead7 %.0f   max rel err (400 samples) %.2e\n", M, N, K,
           ms0, fl / ms0 / 1e9, fl / ms1 / 1e9, fl / ms2 / 1e9, fl / ms3 / 1e9, maxerr);
    return maxerr < 1e-3 ? 0 : 1;
}
