// Probe: does an out-of-range lane of buffer_load_dwordx4 ... lds write ZEROS into LDS, or leave LDS untouched?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/buf_lds_oob tools/probes/buf_lds_oob.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void probe(const uint32_t* src, uint32_t nbytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // even lanes in range, odd lanes far out of range
    uint32_t voff = (lane & 1) ? 0x7ffffff0u : (uint32_t)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
    std::vector<uint32_t> h(64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000u + (uint32_t)i;
    uint32_t *d, *o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, (uint32_t)(h.size() * 4), o);
    std::vector<uint32_t> r(64 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int zeros = 0, stale = 0, good = 0, other = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            uint32_t v = r[l * 4 + i];
            if (l & 1) { if (v == 0) ++zeros; else if (v == 0xdeadbeefu) ++stale; else ++other; }
            else { if (v == 0x1000u + (uint32_t)(l * 4 + i)) ++good; else ++other; }
        }
    printf("in-range dwords correct: %d/128; OOB dwords: zero=%d stale=%d other=%d  (lane1: %08x %08x)\n", good, zeros, stale, other, r[4], r[5]);
    return 0;
}
