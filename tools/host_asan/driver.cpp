// Host-side sanitizer sweep (AddressSanitizer + UBSan on the HOST code of libuclstm only; GPU ASan is not available on this pool).
// Calls every entry point that runs entirely on the host -- the launch planners and the pack job table -- over a few thousand
// descriptors, valid and invalid, and checks that each call returns a plan or a negative UCLSTM_E_* code.  Built and run by
// tools/host_asan/run.py; no GPU is needed (nothing is launched).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "uclstm.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}
static int pick(const int* v, int n) { return v[rnd() % n]; }
static int roundup(int v, int m) { return (v + m - 1) / m * m; }

int main() {
    if (uclstm_abi_version() != UCLSTM_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch\n");
        return 1;
    }
    static float dummy[64];
    const int sizes[] = {1, 2, 3, 4, 5, 7, 8, 12, 13, 16, 24, 31, 32, 64, 96, 128, 192, 256, 320, 512};
    const int chans[] = {8, 16, 24, 40, 64, 72, 128, 136, 256, 512, 1024, 2048, -8, 0, 4};
    const int imgs[] = {1, 2, 3, 4, 8, 20, 32, 48, 640, 0, -1};
    const int taps[] = {1, 2, 3, 5, 7, 0, 9};
    long n_ok = 0, n_bad = 0, n_wg = 0, n_pk = 0;
    for (int it = 0; it < 20000; ++it) {
        uclstm_igemm_desc d;
        memset(&d, 0, sizeof d);
        const bool wild = rnd() % 8 == 0;                 // one descriptor in eight gets arbitrary (mostly invalid) fields
        d.n_img = wild ? pick(imgs, 11) : pick(imgs, 9);
        d.H = pick(sizes, 20);
        d.W = (rnd() & 1) ? d.H : pick(sizes, 20);
        d.groups = 1;
        if (rnd() % 4 == 0 && d.n_img > 0) {
            const int cand[] = {2, 4, 8, 20};
            const int g = pick(cand, 4);
            if (d.n_img % g == 0 || wild) d.groups = g;
        }
        d.ktap = wild ? pick(taps, 7) : pick(taps, 5);
        d.scale = 1;
        d.pad = d.ktap / 2;
        if (d.ktap == 2) { d.scale = 2; d.pad = 0; }
        if (wild && (rnd() & 1)) d.pad = (int)(rnd() % 5) - 1;
        d.nsrc = 1 + (rnd() % 3 == 0);
        if (wild && rnd() % 8 == 0) d.nsrc = (int)(rnd() % 4);
        int kseg = 0;
        for (int s = 0; s < 2; ++s) {
            d.src[s].ptr = dummy;
            d.src[s].C = wild ? pick(chans, 15) : pick(chans, 12);
            d.src[s].Hs = (wild && rnd() % 4 == 0) ? pick(sizes, 20) : (d.ktap == 2 ? 2 * d.H : d.H);
            d.src[s].Ws = (wild && rnd() % 4 == 0) ? pick(sizes, 20) : (d.ktap == 2 ? 2 * d.W : d.W);
            d.src[s].offY = (rnd() % 16) ? 0 : (int)(rnd() % 3);
            d.src[s].offX = (rnd() % 16) ? 0 : (int)(rnd() % 3);
            if (s < d.nsrc && d.src[s].C > 0) kseg += roundup(d.src[s].C, 64);
        }
        d.wp = dummy;
        d.N = wild ? pick(chans, 15) : pick(chans, 12);
        d.Ktot = wild && (rnd() & 1) ? (int)(rnd() % 4096) : d.ktap * d.ktap * kseg;
        d.epi = (int)(rnd() % 3);
        if (d.epi == UCLSTM_EPI_LSTM) d.N = roundup(d.N > 0 ? d.N : 64, 64);
        if (wild && rnd() % 8 == 0) d.epi = 7;
        d.nseg = 1;
        d.seg[0].ptr = dummy;
        d.seg[0].n_begin = 0;
        d.seg[0].n_end = d.N;
        d.seg[0].C = d.N > 0 ? d.N : 8;
        d.seg[0].Hd = d.H * d.scale;
        d.seg[0].Wd = d.W * d.scale;
        d.seg[0].scale = d.scale;
        d.Hd_p = d.N > 0 ? d.N / 4 : 8;
        d.c_out = dummy;
        d.h_out = dummy;
        d.acc_out = dummy;
        d.acc_ld = d.N;
        d.ksplit = 1 + (int)(rnd() % 17);
        d.acc_slab = (rnd() & 1) ? ((int64_t)d.n_img * d.H * d.W * (d.N > 0 ? d.N : 8)) : 0;
        const int shp = uclstm_igemm_fwd_shape(&d);
        if (shp < UCLSTM_E_NODEVICE || shp > 3) {
            fprintf(stderr, "igemm_fwd_shape returned %d\n", shp);
            return 1;
        }
        (shp >= 0 ? n_ok : n_bad)++;
        if (d.Ktot > 0) (void)uclstm_igemm_ksplit_used(d.Ktot, d.ktap, d.ksplit);
        if (d.n_img > 0 && d.groups > 0) (void)uclstm_igemm_tiles_per_group(d.n_img, d.H, d.W, d.groups, d.N);

        uclstm_wgrad_desc w;
        memset(&w, 0, sizeof w);
        w.n_img = d.n_img;
        w.H = d.H;
        w.W = d.W;
        w.ktap = d.ktap;
        w.scale = d.scale;
        w.pad = d.pad;
        w.nsrc = d.nsrc;
        w.src[0] = d.src[0];
        w.src[1] = d.src[1];
        w.N = d.N;
        w.Ktot = d.Ktot;
        w.nseg = 1;
        w.seg[0] = d.seg[0];
        w.splits = (rnd() & 1) ? 0 : (int)(rnd() % 300);
        w.overlapped = rnd() & 1;
        w.slab = (rnd() & 1) ? (int64_t)(d.N > 0 ? d.N : 8) * (d.Ktot > 0 ? d.Ktot : 64) : 0;
        const int ws = uclstm_igemm_wgrad_shape(&w);
        const int sp = uclstm_igemm_wgrad_splits(&w);
        n_wg += ws >= 0;
        if (ws > 4 || (ws >= 0 && sp < 1)) {
            fprintf(stderr, "wgrad shape %d splits %d\n", ws, sp);
            return 1;
        }

        uclstm_pack_desc p;
        memset(&p, 0, sizeof p);
        p.taps = d.ktap * d.ktap;
        p.nsrc = d.nsrc;
        p.kseg[0] = d.src[0].C > 0 ? roundup(d.src[0].C, 64) : 0;
        p.kseg[1] = d.nsrc > 1 && d.src[1].C > 0 ? roundup(d.src[1].C, 64) : 0;
        p.cvalid[0] = d.src[0].C > 0 ? d.src[0].C - (int)(rnd() % 8) : 0;
        p.cvalid[1] = p.kseg[1] ? d.src[1].C : 0;
        p.choff[1] = p.cvalid[0];
        p.N = d.N;
        p.Ktot = (!wild || (rnd() & 1)) ? p.taps * (p.kseg[0] + p.kseg[1]) : (int)(rnd() % 999);
        p.n_mode = (int)(rnd() % (wild ? 4 : 3));
        p.n_valid = d.N > 0 ? d.N - (int)(rnd() % 8) : 0;
        p.n_cp = d.N > 0 ? d.N / 4 : 0;
        p.k_mode = (int)(rnd() % (wild ? 4 : 3));
        p.k_hdp = 8 * (1 + (int)(rnd() % 8));
        p.k_hd = p.k_hdp - (int)(rnd() % 8);
        p.tap_flip = rnd() & 1;
        p.stride_n = (rnd() & 1) ? p.taps : (int64_t)(p.cvalid[0] + p.cvalid[1]) * p.taps;
        p.stride_k = (rnd() & 1) ? p.taps : (int64_t)d.N * p.taps;
        p.stride_tap = rnd() % 4 ? 1 : 0;
        p.stride_ntap = rnd() & 1;
        uclstm_pack_job job;
        const int fam = uclstm_pack_job_init(&job, &p, dummy, dummy, (int)(rnd() % 1000));
        n_pk += fam >= 0;
        if (fam > 4 || (fam >= 0 && (job.nblocks < 1 || job.gx < 1 || job.family != fam))) {
            fprintf(stderr, "pack_job_init returned %d (nblocks %d gx %d)\n", fam, job.nblocks, job.gx);
            return 1;
        }
        if (d.n_img > 0 && d.H > 0 && d.W > 0) {
            const int64_t pixels = (int64_t)d.n_img * d.H * d.W;
            (void)uclstm_bn_bwd_reduce_rows(pixels, pixels / (d.groups > 0 && d.n_img % d.groups == 0 ? d.groups : 1));
        }
    }
    printf("host sanitizer sweep: 20000 descriptors, %ld forward plans, %ld refused, %ld weight-gradient plans, %ld pack jobs, no sanitizer report\n",
           n_ok, n_bad, n_wg, n_pk);
    return 0;
}
