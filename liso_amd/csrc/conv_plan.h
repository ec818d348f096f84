// Host-side planning of the forward / data-gradient convolution launches (pure C++: no HIP): the kernel argument block, the
// tile / slab / stage choice of conv_igemm_kernel (make_plan) and of conv_roles_kernel (plan_roles).  Included by conv_mfma.hip;
// compiled on its own with -fsanitize=address,undefined by tests/test_conv_plan_sanitized.py (GPU sanitizers are not available).
#ifndef LISO_CONV_PLAN_H
#define LISO_CONV_PLAN_H
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liso_conv.h"

namespace {

constexpr int kThreads = 256;

struct FwdArgs {
    const void* x;
    const void* w;
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    void* y;
    float* stats;
    const float* stats_shift;
    int ci_pad, co_pad;
    int cs;        // channels per slab
    int g_taps;    // taps per weight stage
    int cls_dy0[LISO_CONV_MAX_CLASSES], cls_dx0[LISO_CONV_MAX_CLASSES];
    int cls_inh[LISO_CONV_MAX_CLASSES], cls_inw[LISO_CONV_MAX_CLASSES];
    int tiles_x, tiles_y, n_nt, total;
    int x_plane_bytes;  // LDS bytes of one plane of the input tile (max over classes), multiple of 16
    int pipelined;      // stage-granular software pipeline: the loads of stage t + 1 (G taps of a slab) overlap the MFMAs of stage t
    int slab_pipelined; // one register batch holds a whole slab (tile + the panels of all taps): what the SK = 2 kernel needs
    int group_bytes;    // LDS bytes of one split-K group's tile + panels (SK = 2 instantiations)
    const float* occ;   // optional [batch, hi, wi]: 0 = the input pixel is exactly zero in every channel (sparse BEV canvases)
    int dbg;            // (unused: experiment switches of removed kernels)
    int roles;          // conv_roles_kernel (loader waves + MFMA waves, double-buffered LDS): 3x3 / 1x1, stride 1, one class
    int wide_out;       // roles: 16-byte output stores through an LDS patch (channel count and strides allow them)
    unsigned long long roles_tapw;  // roles: 4 bits per window position (ty * 3 + tx): the tap's index inside the packed weights
    int direct1x1;      // conv_1x1_kernel: one tap, fragments straight from global memory (no LDS staging, no barrier in the reduction)
    int direct_taps;    // conv_taps_kernel: <= 8 input channels, any window: two taps per 16-wide MFMA step, fragments from global memory
};

int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Plan {
    int mi, nj, cs, g, lds, sk;
    FwdArgs a;
};

int g_shared_gpu = 0;  // liso_conv_set_option(LISO_CONV_OPT_SHARED_GPU)
int g_roles_cus = 0;   // liso_conv_set_option(LISO_CONV_OPT_ROLES_CUS): 0 = every compute unit

// conv_roles_kernel: 3x3 windows (any tap order: forward and mirrored data-gradient taps), stride 1, one class, bf16 / F32X3.
// Tile shape (MI, NJ) by a cycle model of one block per CU: rounds of 256 blocks x slabs x max(MFMA cycles, loader cycles per slab).
void plan_roles(const liso_conv_desc& d, Plan* p) {
    FwdArgs& a = p->a;
    a.roles = 0;
    static const int roles_env = getenv("LISO_CONV_ROLES") ? atoi(getenv("LISO_CONV_ROLES")) : 1;
    if (!roles_env || d.mode == LISO_CONV_F32) return;
    if (d.n_classes != 1 || d.isy != 1 || d.isx != 1 || d.osy != 1 || d.osx != 1 || d.n_taps != 9) return;
    const bool x3 = d.mode == LISO_CONV_F32X3;
    if (!x3 && d.ci % 32) return;  // (bf16 slabs are 32 channels: no half-filled last slab in the loaders' branch-free loop)
    int y0 = 1 << 30, x0 = 1 << 30, y1 = -(1 << 30), x1 = -(1 << 30);
    for (int t = 0; t < 9; t++) {
        y0 = d.tap_dy[t] < y0 ? d.tap_dy[t] : y0;
        y1 = d.tap_dy[t] > y1 ? d.tap_dy[t] : y1;
        x0 = d.tap_dx[t] < x0 ? d.tap_dx[t] : x0;
        x1 = d.tap_dx[t] > x1 ? d.tap_dx[t] : x1;
    }
    if (y1 - y0 != 2 || x1 - x0 != 2) return;
    int seen = 0;
    a.roles_tapw = 0ull;
    for (int t = 0; t < 9; t++) {
        const int pos = (d.tap_dy[t] - y0) * 3 + (d.tap_dx[t] - x0);
        if (d.tap_w[t] < 0 || d.tap_w[t] > 15) return;
        seen |= 1 << pos;
        a.roles_tapw |= (unsigned long long)d.tap_w[t] << (4 * pos);
    }
    if (seen != 0x1ff) return;
    const int cs = x3 ? 16 : 32, ks = cs / 16, nslab = (d.ci + cs - 1) / cs;
    // (round 5, before the blocks became persistent: shallow layers -- 2 slabs -- lost to conv_igemm_kernel, 76 vs 67 us on 32 -> 32 at
    // 8 x 256^2; with the next tile staged under the epilogue: 58 vs 65 us.  LISO_ROLES_MIN_SLABS: experiments)
    static const int min_slabs = getenv("LISO_ROLES_MIN_SLABS") ? atoi(getenv("LISO_ROLES_MIN_SLABS")) : 1;
    if (nslab < min_slabs || nslab < 2) return;  // (one slab per tile: the deferred statistics flush would race with the next epilogue)
    static const int force_mi = getenv("LISO_ROLES_MI") ? atoi(getenv("LISO_ROLES_MI")) : 0;
    static const int force_nj = getenv("LISO_ROLES_NJ") ? atoi(getenv("LISO_ROLES_NJ")) : 0;
    double best = 1e300;
    int bmi = 0, bnj = 0;
    for (int mi = 1; mi <= 2; mi++)
        for (int nj = 1; nj <= (x3 ? 3 : 2); nj++) {
            if (mi == 2 && nj == 3) continue;  // (two buffers would not fit 160 KB)
            if ((force_mi && mi != force_mi) || (force_nj && nj != force_nj)) continue;
            const long blocks = (long)d.batch * ((d.hv + 4 * mi - 1) / (4 * mi)) * ((d.wv + 31) / 32) * ((d.co + 32 * nj - 1) / (32 * nj));
            const long rounds = (blocks + 255) / 256;
            const int npix = (4 * mi + 2) * 34, xb = (npix * 4 + 255) / 256, wb = (9 * (x3 ? 2 : 1) * (cs / 8) * 32 * nj + 255) / 256;
            const double mfma = 9.0 * ks * mi * nj * (x3 ? 3 : 1) * 32.0;
            const double load = 40.0 * (xb * (x3 ? 6 : 2) + wb * 2) + 200.0;
            const double cost = (double)rounds * (nslab * (mfma > load ? mfma : load) + 3000.0) - 1e-3 * mi * nj;
            if (cost < best) {
                best = cost;
                bmi = mi;
                bnj = nj;
            }
        }
    if (!bmi) return;
    a.roles = 1;
    p->mi = bmi;
    p->nj = bnj;
    p->sk = 1;
    p->cs = cs;
    a.cs = cs;
    a.cls_dy0[0] = y0;
    a.cls_dx0[0] = x0;
    const int bnt = 32 * bnj, th = 4 * bmi;
    a.n_nt = (d.co + bnt - 1) / bnt;
    a.tiles_x = (d.wv + 31) / 32;
    a.tiles_y = (d.hv + th - 1) / th;
    a.total = d.batch * a.tiles_y * a.tiles_x * a.n_nt;
    const int planes = x3 ? 2 : 1;
    const int buf = ((th + 2) * 34 * (cs * 2 + 16) * planes + 9 * planes * (cs / 8) * bnt * 16 + 15) / 16 * 16;
    p->lds = 4096 + 2 * buf + 4 * 2048;  // (+ the epilogue's 2-KB patch per MFMA wave)
    const bool of32 = x3 || d.out_f32;
    const int cv = of32 ? 4 : 8;  // channels per 16-byte store
    static const int wide_env = getenv("LISO_ROLES_WIDE") ? atoi(getenv("LISO_ROLES_WIDE")) : 1;
    a.wide_out = (wide_env && d.co % cv == 0 && d.y_pix_stride % cv == 0 && d.y_ch_off % cv == 0) ? 1 : 0;
}

// panel width (in 32-filter tiles) of the direct kernels: the fewest (blocks per pixel tile) x (per-step cost ~ 2 + nj): 128 filters as
// 2 x 64 rather than 96 + 32 (the block of the narrow remainder pays the same A loads and the same latency per step)
inline int direct_panel_width(int co) {
    int best = 1, best_cost = 1 << 30;
    for (int nj = 1; nj <= 3; nj++) {
        const int cost = ((co + 32 * nj - 1) / (32 * nj)) * (2 + nj);
        if (cost < best_cost) {
            best_cost = cost;
            best = nj;
        }
    }
    return best;
}

// conv_1x1_kernel: F32X3, one class, ONE tap (1x1 convolutions of any stride: the residual shortcuts of liso/slim/model/extractor.py and
// conv_stat_corr1 of update.py:49).  conv_igemm_kernel runs them as 3-13 channel slabs of load -> LDS -> barrier -> one tap of MFMAs ->
// barrier (14-29 us for 0.1-0.9 GFLOP); here every wave reads its A fragments (32 pixels x 8 channels per lane half) and the weight
// fragments (packed in fragment order already) straight from global memory, one k-step ahead, and nothing synchronises before the epilogue.
void plan_1x1(const liso_conv_desc& d, Plan* p) {
    FwdArgs& a = p->a;
    a.direct1x1 = 0;
    static const int env = getenv("LISO_CONV_1X1") ? atoi(getenv("LISO_CONV_1X1")) : 1;
    if (!env || (d.mode != LISO_CONV_F32X3 && d.mode != LISO_CONV_BF16)) return;
    if (d.n_classes > 1) {  // tap classes: transposed convolutions with kernel = stride -- one tap per class (tap index = class index)
        if (d.n_taps != d.n_classes) return;
        for (int c = 0; c < d.n_classes; c++)
            if (d.class_tap_begin[c] != c) return;
    } else if (d.osy != 1 || d.osx != 1) {
        return;
    }
    // (the kernel walks any tap list; measured on the heads' 256 -> 6 output layer, 3x3 at 4 x 64 x 64: 47.7 us against 21.2 us on
    // conv_roles_kernel -- re-reading the pixels per tap through L1, 32 lines per load instruction, costs more than staging the tile
    // once even for a single 32-filter panel.  LISO_CONV_1X1=3 selects it for <= 16 filters on >= 64 channels: experiments)
    const bool narrow = env == 3 && d.mode == LISO_CONV_F32X3 && d.n_classes == 1 && d.n_taps >= 2 && d.n_taps <= 9 && d.co <= 16 && d.ci >= 64;
    // kernel = stride (the k = 2 / stride-2 deblock, rpn.py:70-104): the taps tile the input without overlap -- every input pixel is read
    // by exactly one tap, nothing is re-read: a 1x1 problem on s * s * ci channels
    bool cell = d.n_classes == 1 && d.n_taps > 1 && d.n_taps == d.isy * d.isx;
    if (cell) {
        unsigned seen = 0;
        for (int t = 0; t < d.n_taps && cell; t++) {
            if (d.tap_dy[t] < 0 || d.tap_dy[t] >= d.isy || d.tap_dx[t] < 0 || d.tap_dx[t] >= d.isx) cell = false;
            else seen |= 1u << (d.tap_dy[t] * d.isx + d.tap_dx[t]);
        }
        cell = cell && d.n_taps <= 16 && seen == (1u << d.n_taps) - 1u;
    }
    // 3x3 / stride-2 layers on pixels of <= 256 bytes (64 fp32 / 128 bf16 channels): every input pixel is read by 2.25 taps on average,
    // its lines stay in L1 between them -- measured on the encoders' four such layers (4 pairs): 141 -> 105 us against conv_igemm_kernel
    // (LISO_CONV_1X1=2: one tap only)
    const bool strided = env != 2 && d.n_classes == 1 && d.n_taps == 9 && d.isy == 2 && d.isx == 2 && d.ci > 8 &&  // (<= 8: conv_taps_kernel)
                         d.ci * (d.mode == LISO_CONV_F32X3 ? 4 : 2) <= 256;
    // (experiment, LISO_CONV_1X1=5: 3x3 / stride-1 layers on pixels of <= 128 bytes -- 32 fp32 channels, the encoders' first stage: 9
    // reads per pixel through L1.  Measured 114 vs 63.7 us (8 x 256^2) and 47.7 vs 29.9 us (4 x 256^2) against conv_roles_kernel: off)
    const bool small_px = env == 5 && d.n_classes == 1 && d.n_taps == 9 && d.isy == 1 && d.isx == 1 && d.ci > 8 &&
                          d.ci * (d.mode == LISO_CONV_F32X3 ? 4 : 2) <= 128;
    if (d.n_classes == 1 && d.n_taps != 1 && !narrow && !cell && !strided && !small_px) return;
    if (a.roles && !narrow && !small_px) return;
    for (int t = 0; t < d.n_taps; t++)
        if (d.tap_w[t] < 0 || d.tap_w[t] >= d.w_taps) return;
    if (d.in_affine_batch_stride & 3) return;
    a.roles = 0;
    const int nj = direct_panel_width(d.co);
    const long tiles4 = (long)d.n_classes * d.batch * ((d.hv + 3) / 4) * ((d.wv + 31) / 32) * ((d.co + 32 * nj - 1) / (32 * nj));
    const int mi = tiles4 >= 1024 ? 2 : 1;  // (8-row tiles once 4-row tiles would give every CU four blocks anyway)
    a.direct1x1 = 1;
    p->mi = mi;
    p->nj = nj;
    p->sk = 1;
    const int bnt = 32 * nj, th = 4 * mi;
    a.n_nt = (d.co + bnt - 1) / bnt;
    a.tiles_x = (d.wv + 31) / 32;
    a.tiles_y = (d.hv + th - 1) / th;
    a.total = d.n_classes * d.batch * a.tiles_y * a.tiles_x * a.n_nt;
}

// conv_taps_kernel: F32X3, one class, at most 8 input channels, 2+ taps (the motion encoder's 7x7 convolutions on flow / logits,
// update.py:54-59: 2-8 channels).  conv_igemm_kernel pads the channels to a 16-channel slab per tap -- half (8 channels) to seven
// eighths (2) of every MFMA multiply zeros, 49 tap stages of load -> LDS -> barrier each; here one 16-wide MFMA step holds TWO taps x 8
// channels (lane half h takes tap 2 s + h: its pixel is shifted by that tap), fragments straight from global memory.
void plan_taps(const liso_conv_desc& d, Plan* p) {
    FwdArgs& a = p->a;
    a.direct_taps = 0;
    static const int env = getenv("LISO_CONV_TAPS") ? atoi(getenv("LISO_CONV_TAPS")) : 1;
    if (!env || a.roles || a.direct1x1 || d.mode != LISO_CONV_F32X3 || d.n_classes != 1 || d.n_taps < 2 || d.ci > 8 || d.osy != 1 || d.osx != 1)
        return;
    if (d.in_affine_batch_stride & 3) return;
    for (int t = 0; t < d.n_taps; t++)
        if (d.tap_w[t] < 0 || d.tap_w[t] >= d.w_taps) return;
    const int nj = direct_panel_width(d.co);
    a.direct_taps = 1;
    p->mi = 1;
    p->nj = nj;
    p->sk = 1;
    const int bnt = 32 * nj;
    a.n_nt = (d.co + bnt - 1) / bnt;
    a.tiles_x = (d.wv + 31) / 32;
    a.tiles_y = (d.hv + 3) / 4;
    a.total = d.batch * a.tiles_y * a.tiles_x * a.n_nt;
}

bool make_plan(const liso_conv_desc& d, Plan* p) {
    if (d.batch <= 0 || d.ci <= 0 || d.co <= 0 || d.n_classes < 1 || d.n_classes > LISO_CONV_MAX_CLASSES) return false;
    if (d.n_taps < 1 || d.n_taps > LISO_CONV_MAX_TAPS || d.class_tap_begin[0] != 0 || d.class_tap_begin[d.n_classes] != d.n_taps)
        return false;
    if (d.mode != LISO_CONV_BF16 && d.mode != LISO_CONV_F32X3 && d.mode != LISO_CONV_F32) return false;
    const bool x3 = d.mode == LISO_CONV_F32X3, f32 = d.mode == LISO_CONV_F32, fin = x3 || f32;
    const int vec = fin ? 4 : 8;
    if (d.ci % vec || d.x_pix_stride % vec || d.x_pix_stride < d.ci) return false;
    const int planes = x3 ? 2 : 1;
    auto pix_bytes = [&](int cs) { return (f32 ? cs * 4 : cs * 2) + 16; };                       // LDS bytes per tile pixel and plane
    auto panel_bytes = [&](int cs, int bnt) { return f32 ? (cs / 4) * bnt * 16 : planes * (cs / 8) * bnt * 16; };  // one tap
    FwdArgs& a = p->a;
    a.ci_pad = round_up(d.ci, 16);
    a.co_pad = round_up(d.co, 64);
    p->nj = d.co <= 32 ? 1 : 2;
    {
        // small maps (the RAFT update block of one sample: 64 x 64 pixels = 32 tiles): 64-channel panels leave most CUs
        // without a block -> 32-channel panels double the block count (the tile is re-staged from L2 by twice as many blocks)
        const long tiles4 = (long)d.n_classes * d.batch * ((d.hv + 3) / 4) * ((d.wv + 31) / 32);
        if (p->nj == 2 && tiles4 * ((d.co + 63) / 64) < 160) p->nj = 1;
        if (p->nj == 2 && x3) {
            // F32X3 launches of a few hundred blocks are bound by the MFMA stream of the busiest CU: blocks per CU x work per block.
            // 64-channel panels on 384 blocks (ConvGRU z|r, 4 pairs: 128 tiles x 3 panels) give half the CUs two blocks of work 2;
            // 32-channel panels give every CU three blocks of work 1 (+ the tile staged twice as often: charged as 0.3 per block)
            const long b2 = tiles4 * ((d.co + 63) / 64), b1 = tiles4 * ((d.co + 31) / 32);
            const double c2 = (double)((b2 + 255) / 256) * 2.3, c1 = (double)((b1 + 255) / 256) * 1.3;
            // MEASURED (round 4): no gain in the loop (5.70 vs 5.69 ms), SLIM train step slower (15.5 vs 14.9 ms): off unless asked for
            static const bool auto_nj = getenv("LISO_CONV_NJ_AUTO") != nullptr && atoi(getenv("LISO_CONV_NJ_AUTO")) != 0;
            if (auto_nj && b1 <= 2048 && c1 < c2) p->nj = 1;
        }
        // F32X3 layers with 65-96 filters (ConvGRU q 304->96, the motion encoder's 160->80, the encoders' 96->96 stage): one 96-wide
        // panel instead of two 64-wide ones -- no padded filter columns through the matrix cores (96 -> 128: a quarter of the MFMAs,
        // 80 -> 128: three eighths) and the input tile staged once
        // MEASURED (round 4): correct on every test, no gain -- loop 5.10 vs 5.10 ms, SLIM step 13.99 vs 13.88 ms (the 96-wide panel
        // leaves room for 3 instead of 5 taps per weight stage in the 79-KB plans).  Off unless LISO_CONV_NJ3=1.
        static const bool nj3 = getenv("LISO_CONV_NJ3") != nullptr && atoi(getenv("LISO_CONV_NJ3")) != 0;
        if (nj3 && x3 && p->nj == 2 && d.co > 64 && d.co <= 96) p->nj = 3;
        if (const char* e = getenv("LISO_CONV_NJ")) {  // experiments: 1 | 2 force the panel width
            if (atoi(e) == 1) p->nj = 1;
            if (atoi(e) == 2 && d.co > 32) p->nj = 2;
        }
    }
    const int bnt = 32 * p->nj;
    a.n_nt = (d.co + bnt - 1) / bnt;
    auto blocks = [&](int th) { return (long)d.n_classes * d.batch * ((d.hv + th - 1) / th) * ((d.wv + 31) / 32) * a.n_nt; };
    int max_taps = 1;
    for (int c = 0; c < d.n_classes; c++) {
        const int nt = d.class_tap_begin[c + 1] - d.class_tap_begin[c];
        if (nt < 1) return false;
        max_taps = nt > max_taps ? nt : max_taps;
        for (int t = d.class_tap_begin[c]; t < d.class_tap_begin[c + 1]; t++)
            if (d.tap_w[t] < 0 || d.tap_w[t] >= d.w_taps) return false;
    }
    auto tile_pixels = [&](int th, int* inh, int* inw, int* y0s, int* x0s) {
        int max_pix = 0;
        for (int c = 0; c < d.n_classes; c++) {
            int y0 = 1 << 30, y1 = -(1 << 30), x0 = 1 << 30, x1 = -(1 << 30);
            for (int t = d.class_tap_begin[c]; t < d.class_tap_begin[c + 1]; t++) {
                y0 = d.tap_dy[t] < y0 ? d.tap_dy[t] : y0;
                y1 = d.tap_dy[t] > y1 ? d.tap_dy[t] : y1;
                x0 = d.tap_dx[t] < x0 ? d.tap_dx[t] : x0;
                x1 = d.tap_dx[t] > x1 ? d.tap_dx[t] : x1;
            }
            y0s[c] = y0;
            x0s[c] = x0;
            inh[c] = (th - 1) * d.isy + (y1 - y0) + 1;
            inw[c] = 31 * d.isx + (x1 - x0) + 1;
            const int np = inh[c] * inw[c];
            max_pix = np > max_pix ? np : max_pix;
        }
        return max_pix;
    };
    // Choose (rows per tile, slab width, taps per weight stage): the configuration with the most MFMAs between two barriers
    // among those that leave room for 2 blocks per CU (79 KB); one block per CU (158 KB) only if nothing else fits.
    // 8-row tiles only when they still give every CU 2 blocks.
    int mi_first = blocks(8) >= 512 ? 2 : 1;
    if (const char* e = getenv("LISO_CONV_MI")) mi_first = atoi(e) == 2 ? 2 : atoi(e) == 1 ? 1 : mi_first;  // experiments
    int cs_opts[2] = {fin ? 32 : 64, fin ? 16 : 32};
    if (const char* e = getenv("LISO_CONV_CS")) {  // experiments: force the slab width (bf16: 64 | 32; fp32 tensors: 32 | 16)
        const int v = atoi(e);
        if (v == cs_opts[0] || v == cs_opts[1]) cs_opts[0] = cs_opts[1] = v;
    }
    long best = -1;
    // at most one block per CU anyway: spend its whole LDS -- unless other streams' kernels share the GPU (the LISO loop's three
    // pipeline stages): a block that owns all 160 KB keeps every other kernel's blocks off its CU while its own 4 waves mostly wait
    // (loop, 60 steps: 6.16 -> 6.04 ms per step with the 79-KB plans)
    bool few_blocks = !g_shared_gpu && blocks(4 * mi_first) <= 256;
    if (const char* e = getenv("LISO_CONV_FEW")) few_blocks = few_blocks && atoi(e) != 0;  // experiments
    for (int pass = 0; pass < 2 && best < 0; pass++) {
        int cap = (pass == 0 && !few_blocks ? 79 : 158) * 1024;
        if (const char* e = getenv("LISO_CONV_LDS_KB")) { if (pass == 0 && atoi(e) >= 16) cap = atoi(e) * 1024; }  // experiments
        for (int mi = mi_first; mi >= 1; mi--) {
            int inh[LISO_CONV_MAX_CLASSES], inw[LISO_CONV_MAX_CLASSES], y0s[LISO_CONV_MAX_CLASSES], x0s[LISO_CONV_MAX_CLASSES];
            const int max_pix = tile_pixels(4 * mi, inh, inw, y0s, x0s);
            for (int k = 0; k < 2; k++) {
                const int cs = cs_opts[k];
                if (k == 0 && cs_opts[1] >= a.ci_pad) continue;  // do not stage channels that do not exist
                const int xb = round_up(max_pix * pix_bytes(cs), 16);
                const int panel = panel_bytes(cs, bnt);
                int g = (cap - 512 - xb * planes) / panel;
                if (g < 1) continue;
                g = g > max_taps ? max_taps : g;
                const long score = ((long)g * (cs / 16) * mi * 1000) + cs + (mi == mi_first ? 500000 : 0);
                if (score > best) {
                    best = score;
                    p->mi = mi;
                    p->cs = cs;
                    p->g = g;
                    a.x_plane_bytes = xb;
                    p->lds = 512 + xb * planes + g * panel;
                    for (int c = 0; c < d.n_classes; c++) {
                        a.cls_dy0[c] = y0s[c];
                        a.cls_dx0[c] = x0s[c];
                        a.cls_inh[c] = inh[c];
                        a.cls_inw[c] = inw[c];
                    }
                }
            }
            if (best >= 0) break;  // (prefer the taller tile whenever it fits)
        }
    }
    if (best < 0) return false;
    if (p->lds < 4096) p->lds = 4096;  // the statistics epilogue reuses the front of the buffer
    const int th = 4 * p->mi;
    a.cs = p->cs;
    a.g_taps = p->g;
    {
        int max_pix = 0;
        for (int c = 0; c < d.n_classes; c++) max_pix = a.cls_inh[c] * a.cls_inw[c] > max_pix ? a.cls_inh[c] * a.cls_inw[c] : max_pix;
        const int cpp = fin ? p->cs / 4 : p->cs / 8;
        const int x_chunks = (max_pix * cpp + kThreads - 1) / kThreads;          // per thread
        const int w_chunks = (max_taps * (panel_bytes(p->cs, bnt) / 16) + kThreads - 1) / kThreads;
        // stage-granular software pipeline: the halo tile within one register batch (12 x 16 B per thread) and a weight stage within
        // one (10 x 16 B per thread = 40 KB): the stage shrinks to the taps that fit
        const int panel = panel_bytes(p->cs, bnt);
        int g_pipe = (10 * kThreads * 16) / panel;
        g_pipe = g_pipe > p->g ? p->g : g_pipe;
        a.slab_pipelined = (p->g >= max_taps && x_chunks <= 12 && w_chunks <= 10) ? 1 : 0;  // (round 4's condition: the SK = 2 kernel needs it)
        a.pipelined = (x_chunks <= 12 && g_pipe >= 1) ? 1 : 0;
        if (const char* e = getenv("LISO_CONV_PIPE")) {  // experiments: 0 = never, 1 = round 4's whole-slab condition, 2 = stage-granular (default)
            const int v = atoi(e);
            if (v == 0) a.pipelined = 0;
            if (v == 1) a.pipelined = a.slab_pipelined;
        }
        if (a.pipelined && !(a.slab_pipelined && p->g >= max_taps)) {
            p->g = g_pipe;
            a.g_taps = g_pipe;
            p->lds = 512 + a.x_plane_bytes * planes + g_pipe * panel;
            if (p->lds < 4096) p->lds = 4096;
        }
    }
    a.tiles_x = (d.wv + 31) / 32;
    a.tiles_y = (d.hv + th - 1) / th;
    a.total = (int)blocks(th);
    // split-K inside the block (two groups of 4 waves on alternate slabs): F32X3, one wave tile per wave, at most one block per CU
    // anyway, at least two slabs, and both groups' buffers + the accumulator hand-over fit the CU's LDS
    p->sk = 1;
    a.group_bytes = round_up(a.x_plane_bytes * planes + max_taps * panel_bytes(p->cs, bnt), 16);
    if (x3 && a.slab_pipelined && p->mi == 1 && p->nj == 1 && a.total <= 256 && d.ci > p->cs && 512 + 2 * a.group_bytes <= 160 * 1024 &&
        a.group_bytes >= 16 * kThreads * 4)
        p->sk = 2;
    if (const char* e = getenv("LISO_CONV_SK")) p->sk = (atoi(e) >= 2 && p->sk == 2) ? 2 : 1;  // experiments
    if (p->sk == 2) p->lds = 512 + 2 * a.group_bytes;
    // (rounds 3-4 kept an 8-wave variant with LDS-DMA weight stages, conv_igemm8_kernel, behind LISO_CONV_A8: correct, never faster
    // than this kernel -- one block per CU, every wave both loads and multiplies; round 5's conv_roles_kernel is the role-split form
    // that works, and the experiment was removed)
    plan_roles(d, p);
    plan_1x1(d, p);
    plan_taps(d, p);
    return true;
}

}  // namespace

#endif
