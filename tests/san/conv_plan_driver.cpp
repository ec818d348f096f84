// CPU driver for tests/test_conv_plan_sanitized.py: the host-side planning arithmetic of the convolution launches
// (liso_amd/csrc/conv_plan.h: make_plan, plan_roles) over a sweep of pseudo-random and degenerate descriptors, built with
// -fsanitize=address,undefined.  Every accepted plan must satisfy the invariants the kernels rely on.
#include <stdio.h>
#include <string.h>

#include "../../liso_amd/csrc/conv_plan.h"

static unsigned g_s = 12345;
static int rnd(int n) {
    g_s = g_s * 1664525u + 1013904223u;
    return (int)((g_s >> 8) % (unsigned)n);
}

static int check(const liso_conv_desc& d, const Plan& p, long it) {
    const FwdArgs& a = p.a;
    const int bnt = 32 * p.nj, th = 4 * p.mi;
#define REQUIRE(c)                                                                     \
    if (!(c)) {                                                                        \
        printf("FAIL descriptor %ld: %s (mi %d nj %d cs %d lds %d roles %d)\n", it, #c, p.mi, p.nj, p.cs, p.lds, a.roles); \
        return 1;                                                                      \
    }
    REQUIRE(p.lds > 0 && p.lds <= 160 * 1024);
    REQUIRE(p.mi >= 1 && p.mi <= 2 && p.nj >= 1 && p.nj <= 3);
    REQUIRE(a.total > 0 && a.n_nt >= 1 && a.tiles_x >= 1 && a.tiles_y >= 1);
    REQUIRE((long)a.tiles_x * 32 >= d.wv && (long)a.tiles_y * th >= d.hv);
    REQUIRE((long)a.n_nt * bnt >= d.co);
    REQUIRE(a.total == (long)d.n_classes * d.batch * a.tiles_y * a.tiles_x * a.n_nt);
    REQUIRE(a.ci_pad >= d.ci && a.ci_pad % 16 == 0 && a.co_pad >= d.co && a.co_pad % 64 == 0);
    if (a.roles) {
        REQUIRE(d.n_taps == 9 && d.n_classes == 1);
        for (int pos = 0; pos < 9; pos++) REQUIRE((int)((a.roles_tapw >> (4 * pos)) & 15ull) < d.w_taps);
        REQUIRE(a.cs == (d.mode == LISO_CONV_F32X3 ? 16 : 32));
    } else {
        REQUIRE(a.g_taps >= 1 && a.x_plane_bytes % 16 == 0 && a.x_plane_bytes > 0);
    }
    return 0;
}

int main() {
    long tried = 0, planned = 0, roles = 0;
    for (long it = 0; it < 200000; it++) {
        liso_conv_desc d;
        memset(&d, 0, sizeof d);
        static const int ks[] = {1, 2, 3, 5, 7};
        const int k = ks[rnd(5)], st = 1 + rnd(2);
        d.batch = 1 + rnd(8);
        d.hi = 1 + rnd(1100);
        d.wi = 1 + rnd(1100);
        d.ci = 4 * (1 + rnd(110));
        d.x_pix_stride = d.ci + 8 * rnd(3);
        d.ho = (d.hi + 2 * (k / 2) - k) / st + 1;
        d.wo = (d.wi + 2 * (k / 2) - k) / st + 1;
        if (d.ho < 1 || d.wo < 1) continue;
        d.co = 1 + rnd(420);
        d.y_pix_stride = d.co + 8 * rnd(3);
        d.y_ch_off = 8 * rnd(2);
        d.hv = d.ho;
        d.wv = d.wo;
        d.isy = d.isx = st;
        d.osy = d.osx = 1;
        d.n_classes = 1;
        d.n_taps = k * k;
        d.class_tap_begin[0] = 0;
        d.class_tap_begin[1] = d.n_taps;
        const int mirror = rnd(2);
        for (int t = 0; t < d.n_taps; t++) {
            d.tap_dy[t] = t / k - k / 2;
            d.tap_dx[t] = t % k - k / 2;
            d.tap_w[t] = mirror ? d.n_taps - 1 - t : t;
        }
        d.w_taps = d.n_taps;
        d.mode = rnd(3);
        d.out_f32 = rnd(2);
        tried++;
        Plan p;
        memset(&p, 0, sizeof p);
        if (!make_plan(d, &p)) continue;
        planned++;
        roles += p.a.roles;
        if (check(d, p, it)) return 1;
    }
    // 4-class descriptors (stride-2 data gradients / transposed convolutions): 2 x 2 taps per class
    for (long it = 0; it < 20000; it++) {
        liso_conv_desc d;
        memset(&d, 0, sizeof d);
        d.batch = 1 + rnd(4);
        d.hi = 1 + rnd(300);
        d.wi = 1 + rnd(300);
        d.ci = 8 * (1 + rnd(48));
        d.x_pix_stride = d.ci;
        d.ho = 2 * d.hi;
        d.wo = 2 * d.wi;
        d.co = 1 + rnd(300);
        d.y_pix_stride = d.co;
        d.hv = d.hi;
        d.wv = d.wi;
        d.isy = d.isx = 1;
        d.osy = d.osx = 2;
        d.n_classes = 4;
        d.n_taps = 4;
        for (int c = 0; c < 4; c++) {
            d.class_tap_begin[c] = c;
            d.class_ooy[c] = c / 2;
            d.class_oox[c] = c % 2;
            d.tap_w[c] = c;
        }
        d.class_tap_begin[4] = 4;
        d.w_taps = 4;
        d.mode = rnd(3);
        tried++;
        Plan p;
        memset(&p, 0, sizeof p);
        if (!make_plan(d, &p)) continue;
        planned++;
        if (check(d, p, it)) return 1;
    }
    // degenerate descriptors are refused, not planned
    liso_conv_desc z;
    memset(&z, 0, sizeof z);
    Plan p;
    if (make_plan(z, &p)) { printf("FAIL: all-zero descriptor accepted\n"); return 1; }
    z.batch = 1; z.ci = 4; z.co = 4; z.n_classes = LISO_CONV_MAX_CLASSES + 1;
    if (make_plan(z, &p)) { printf("FAIL: too many classes accepted\n"); return 1; }
    z.n_classes = 1; z.n_taps = LISO_CONV_MAX_TAPS + 1;
    if (make_plan(z, &p)) { printf("FAIL: too many taps accepted\n"); return 1; }
    z.n_taps = 1; z.class_tap_begin[1] = 1; z.tap_w[0] = 7; z.w_taps = 1; z.x_pix_stride = 4; z.mode = LISO_CONV_F32X3;
    if (make_plan(z, &p)) { printf("FAIL: tap index beyond the packed weights accepted\n"); return 1; }
    printf("OK %ld plans of %ld descriptors (%ld for conv_roles_kernel)\n", planned, tried, roles);
    return planned > 1000 && roles > 100 ? 0 : 1;
}
