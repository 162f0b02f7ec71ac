// Probe: buffer_load_dwordx4 ... lds with an SGPR soffset: which bytes are fetched and how the range check treats soffset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void probe(const uint32_t* src, uint32_t nrec, uint32_t soff, uint32_t vbase, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    uint32_t voff = vbase + (uint32_t)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
    const size_t n = 1 << 16;   // dwords: 256 KiB
    std::vector<uint32_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)i;     // value = dword index
    uint32_t *d, *o;
    (void)hipMalloc(&d, n * 4);
    (void)hipMalloc(&o, 256 * 4);
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    struct Case { uint32_t nrec, soff, vbase; const char* what; } cases[] = {
        {(uint32_t)(n * 4), 4096, 0, "in range, soffset 4096"},
        {8192, 4096, 0, "nrec 8192, soff 4096, voff 0..1008 (voff+soff < nrec)"},
        {8192, 4096, 4096 - 512, "nrec 8192, soff 4096, voff 3584..4592 (voff<nrec; voff+soff crosses nrec at lane 32)"},
        {8192, 4096, 8192 - 512, "nrec 8192, soff 4096, voff 7680..8688 (voff crosses nrec at lane 32)"},
        {8192, 4096, 0x80000000u, "marker 0x80000000 + soff"},
        {0x80000000u, 0x7ffff000u, 0x80000000u, "nrec 2^31, marker 0x80000000 + soff 0x7ffff000 (sum wraps past 2^32?)"},
    };
    for (auto& c : cases) {
        probe<<<1, 64>>>(d, c.nrec, c.soff, c.vbase, o);
        std::vector<uint32_t> r(256);
        (void)hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int good = 0, zero = 0, stale = 0, other = 0, first_zero = -1;
        for (int l = 0; l < 64; ++l) {
            uint32_t expect = (c.soff + c.vbase + l * 16) / 4;
            uint32_t v = r[l * 4];
            if (v == expect && r[l * 4 + 3] == expect + 3) ++good;
            else if (v == 0) { ++zero; if (first_zero < 0) first_zero = l; }
            else if (v == 0xdeadbeefu) ++stale;
            else ++other;
        }
        printf("%-90s good=%2d zero=%2d (first zero lane %d) stale=%d other=%d\n", c.what, good, zero, first_zero, stale, other);
    }
    return 0;
}
