// Probe: what does a READ-ONLY stream reach on this chip, next to a copy of the same buffer?
// The BatchNorm backward reductions (two reads, no write) run at 3.9 - 4.2 TB/s; this says whether that is the ceiling.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/_bin/bw_read tools/probes/bw_read.hip ; run on the GPU box: bw_read [MiB].
// Output: one line per (kernel, blocks per CU, loads in flight): GB/s over 20 launches after 3 warm-up launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// U independent 16-byte loads in flight per lane, grid-stride; the sum keeps the loads alive, one atomic per block.
template <int U>
__global__ void __launch_bounds__(256) read_kernel(const uint4* __restrict__ a, size_t n16, unsigned* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc == 0x9e3779b9u) atomicAdd(sink, 1u);   // practically never taken; defeats dead-code removal
}

// two read streams, as the reductions have (z and the gradient)
template <int U>
__global__ void __launch_bounds__(256) read2_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n16,
                                                    unsigned* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = a[i + u * stride]; w[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (v[u].x ^ w[u].y) + (v[u].z ^ w[u].w) + (v[u].y ^ w[u].x) + (v[u].w ^ w[u].z);
    }
    for (; i < n16; i += stride) { uint4 v = a[i], w = b[i]; acc += (v.x ^ w.y) + (v.z ^ w.w) + (v.y ^ w.x) + (v.w ^ w.z); }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc == 0x9e3779b9u) atomicAdd(sink, 1u);
}

template <int U>
__global__ void __launch_bounds__(256) copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) b[i] = a[i];
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / 20.0;
}

int main(int argc, char** argv) {
    // default 1.34 GB: far beyond L2 + MALL.  `bw_read 320` = 335.5 MB, the largest activation of the headline step.
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : (size_t)1280) << 20;
    const size_t n16 = bytes / 16;
    uint4 *a, *b; unsigned* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0x5a, bytes)); CK(hipMemset(b, 0x3c, bytes)); CK(hipMemset(sink, 0, 4));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, buffer %.2f GB\n", p.gcnArchName, cus, bytes / 1e9);
    const int per_cu[] = {2, 4, 8, 16};
    for (int pc : per_cu) {
        const int g = cus * pc;
#define ROW(name_, U_, call_, nbytes_) do { double ms = time_ms([&] { call_; }); \
            printf("%-6s blocks/CU %2d  loads in flight %d : %8.1f GB/s  (%.3f ms)\n", name_, pc, U_, (nbytes_) / ms / 1e6, ms); } while (0)
        ROW("read",  4, (read_kernel<4><<<g, 256>>>(a, n16, sink)), (double)bytes);
        ROW("read",  8, (read_kernel<8><<<g, 256>>>(a, n16, sink)), (double)bytes);
        ROW("read2", 4, (read2_kernel<4><<<g, 256>>>(a, b, n16, sink)), 2.0 * bytes);
        ROW("copy",  4, (copy_kernel<4><<<g, 256>>>(a, b, n16)), 2.0 * bytes);
        ROW("copy",  8, (copy_kernel<8><<<g, 256>>>(a, b, n16)), 2.0 * bytes);
    }
    unsigned h = 0; CK(hipMemcpy(&h, sink, 4, hipMemcpyDeviceToHost));
    printf("sink %u\n", h);
    return 0;
}
