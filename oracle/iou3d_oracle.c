/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the iou3d_nms rows (A1-A5).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  The product path (liso_amd/) never does.
 *
 * Plain-C restatement of the reference algorithm; every function cites the
 * reference lines it follows (paths relative to /root/reference).  Arithmetic
 * order and types follow the reference's *host* twin (iou3d_nms/src/iou3d_cpu.cpp)
 * so that, built with -ffp-contract=off against the same libm, it is
 * bit-identical to oracle/_ref/libiou3d_ref.so (the unmodified reference TU).
 *
 * Pinning: tests/test_oracle_iou3d.py checks this file against oracle/_ref (when
 * built) and against tests/golden/iou3d_*.npz, which were generated from
 * oracle/_ref by tests/golden/make_iou3d_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define ORACLE_EPS 1e-8f /* iou3d_cpu.cpp:38 / iou3d_nms_kernel.cu:14 */

typedef struct {
    float x, y;
} pt_t;

/* iou3d_cpu.cpp:30-36 (host min/max on floats) */
static float fmin_ref(float a, float b) { return a > b ? b : a; }
static float fmax_ref(float a, float b) { return a > b ? a : b; }

/* iou3d_cpu.cpp:63-65  cross(p1,p2,p0) */
static float cross3(pt_t p1, pt_t p2, pt_t p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* iou3d_cpu.cpp:59-61  cross(a,b) */
static float cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; }

/* iou3d_cpu.cpp:67-73 */
static int rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2) {
    return fmin_ref(p1.x, p2.x) <= fmax_ref(q1.x, q2.x) && fmin_ref(q1.x, q2.x) <= fmax_ref(p1.x, p2.x) &&
           fmin_ref(p1.y, p2.y) <= fmax_ref(q1.y, q2.y) && fmin_ref(q1.y, q2.y) <= fmax_ref(p1.y, p2.y);
}

/* iou3d_cpu.cpp:75-86  corner-in-box test with 1e-2 margin */
static int in_box2d(const float *box, pt_t p) {
    const float margin = 1e-2f;
    float cx = box[0], cy = box[1];
    float c = cosf(-box[6]), s = sinf(-box[6]);
    float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    float ry = (p.x - cx) * s + (p.y - cy) * c;
    return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

/* iou3d_cpu.cpp:88-117  segment/segment intersection */
static int seg_isect(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t *ans) {
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > ORACLE_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* iou3d_cpu.cpp:119-123 */
static pt_t rot_about(pt_t ctr, float c, float s, pt_t p) {
    pt_t r;
    r.x = (p.x - ctr.x) * c + (p.y - ctr.y) * (-s) + ctr.x;
    r.y = (p.x - ctr.x) * s + (p.y - ctr.y) * c + ctr.y;
    return r;
}

/* iou3d_cpu.cpp:125-127  point_cmp: strictly greater polar angle about centre */
static int angle_gt(pt_t a, pt_t b, pt_t ctr) {
    return atan2f(a.y - ctr.y, a.x - ctr.x) > atan2f(b.y - ctr.y, b.x - ctr.x);
}

static void corners_of(const float *box, pt_t out[5]) {
    /* iou3d_cpu.cpp:134-165 */
    float ang = box[6];
    float hx = box[3] / 2, hy = box[4] / 2;
    float x1 = box[0] - hx, y1 = box[1] - hy;
    float x2 = box[0] + hx, y2 = box[1] + hy;
    pt_t ctr = {box[0], box[1]};
    float c = cosf(ang), s = sinf(ang);
    pt_t raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; k++) out[k] = rot_about(ctr, c, s, raw[k]);
    out[4] = out[0];
}

/* iou3d_cpu.cpp:128-220  area of the intersection polygon of two rotated rectangles */
float oracle_box_overlap(const float *box_a, const float *box_b) {
    pt_t ca[5], cb[5];
    corners_of(box_a, ca);
    corners_of(box_b, cb);

    pt_t poly[16 + 8]; /* the reference declares 16; >16 would overflow there too */
    pt_t ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            pt_t hit;
            if (seg_isect(ca[i + 1], ca[i], cb[j + 1], cb[j], &hit)) {
                poly[cnt] = hit;
                ctr.x = ctr.x + hit.x;
                ctr.y = ctr.y + hit.y;
                cnt++;
            }
        }
    for (int k = 0; k < 4; k++) {
        if (in_box2d(box_a, cb[k])) {
            ctr.x = ctr.x + cb[k].x;
            ctr.y = ctr.y + cb[k].y;
            poly[cnt++] = cb[k];
        }
        if (in_box2d(box_b, ca[k])) {
            ctr.x = ctr.x + ca[k].x;
            ctr.y = ctr.y + ca[k].y;
            poly[cnt++] = ca[k];
        }
    }
    ctr.x /= cnt; /* cnt==0 -> NaN, unused: both loops below are empty */
    ctr.y /= cnt;

    /* iou3d_cpu.cpp:199-209  bubble sort, swap when left angle > right angle */
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (angle_gt(poly[i], poly[i + 1], ctr)) {
                pt_t t = poly[i];
                poly[i] = poly[i + 1];
                poly[i + 1] = t;
            }

    /* iou3d_cpu.cpp:211-217  shoelace fan about poly[0] */
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        pt_t u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
        pt_t v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
        area += cross2(u, v);
    }
    return (float)(fabsf(area) / 2.0);
}

/* iou3d_cpu.cpp:222-229 */
float oracle_iou_bev(const float *box_a, const float *box_b) {
    float sa = box_a[3] * box_a[4];
    float sb = box_b[3] * box_b[4];
    float ov = oracle_box_overlap(box_a, box_b);
    return ov / fmaxf(sa + sb - ov, ORACLE_EPS);
}

/* iou3d_nms_kernel.cu:314-326  axis-aligned IoU (heading ignored) */
float oracle_iou_normal(const float *a, const float *b) {
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, ORACLE_EPS);
}

/* iou3d_cpu.cpp:232-252 */
int oracle_boxes_iou_bev(const float *a, int n, const float *b, int m, float *out) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) out[(size_t)i * m + j] = oracle_iou_bev(a + i * 7, b + j * 7);
    return 1;
}

/* iou3d_nms_kernel.cu:236-249 (same geometry, host arithmetic) */
int oracle_boxes_overlap_bev(const float *a, int n, const float *b, int m, float *out) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) out[(size_t)i * m + j] = oracle_box_overlap(a + i * 7, b + j * 7);
    return 1;
}

/*
 * Suppression bit-matrix, iou3d_nms_kernel.cu:267-311 (rotated) / :328-372 (normal):
 * word (row, colblk) has bit i set iff iou(row, colblk*64+i) > thresh; inside the
 * diagonal tile only columns > row are tested.  Only words colblk >= row/64 are
 * filled (the greedy pass never reads the others, iou3d_nms.cpp:127-129).
 */
static void oracle_mask(const float *boxes, int n, float thresh, int normal, uint64_t *mask) {
    int cb = (n + 63) / 64;
    memset(mask, 0, (size_t)n * cb * sizeof(uint64_t));
    for (int r = 0; r < n; r++)
        for (int c = r + 1; c < n; c++) {
            float v = normal ? oracle_iou_normal(boxes + r * 7, boxes + c * 7) : oracle_iou_bev(boxes + r * 7, boxes + c * 7);
            if (v > thresh) mask[(size_t)r * cb + c / 64] |= 1ULL << (c % 64);
        }
}

/* host greedy sweep, iou3d_nms.cpp:113-132 (and :162-181) */
static int oracle_greedy(const uint64_t *mask, int n, int64_t *keep) {
    int cb = (n + 63) / 64;
    uint64_t *remv = (uint64_t *)calloc(cb > 0 ? cb : 1, sizeof(uint64_t));
    int kept = 0;
    for (int i = 0; i < n; i++) {
        int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[kept++] = i;
            const uint64_t *p = mask + (size_t)i * cb;
            for (int j = nb; j < cb; j++) remv[j] |= p[j];
        }
    }
    free(remv);
    return kept;
}

/* nms_gpu semantics, iou3d_nms.cpp:90-136; boxes must be sorted by descending score */
int oracle_nms(const float *boxes, int n, float thresh, int64_t *keep) {
    if (n <= 0) return 0;
    int cb = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)malloc((size_t)n * cb * sizeof(uint64_t));
    oracle_mask(boxes, n, thresh, 0, mask);
    int k = oracle_greedy(mask, n, keep);
    free(mask);
    return k;
}

/* nms_normal_gpu semantics, iou3d_nms.cpp:139-186 */
int oracle_nms_normal(const float *boxes, int n, float thresh, int64_t *keep) {
    if (n <= 0) return 0;
    int cb = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)malloc((size_t)n * cb * sizeof(uint64_t));
    oracle_mask(boxes, n, thresh, 1, mask);
    int k = oracle_greedy(mask, n, keep);
    free(mask);
    return k;
}

/* greedy NMS driven by an externally supplied IoU matrix (used to turn the
 * reference's own IoU output, oracle/_ref, into a keep list) */
int oracle_nms_from_iou(const float *iou, int n, float thresh, int64_t *keep) {
    if (n <= 0) return 0;
    int cb = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)calloc((size_t)n * cb, sizeof(uint64_t));
    for (int r = 0; r < n; r++)
        for (int c = r + 1; c < n; c++)
            if (iou[(size_t)r * n + c] > thresh) mask[(size_t)r * cb + c / 64] |= 1ULL << (c % 64);
    int k = oracle_greedy(mask, n, keep);
    free(mask);
    return k;
}
