/*
 * orb_oracle_extract.c -- CPU ORACLE (test infrastructure, NOT product code; see orb_oracle.h).
 *
 * Restates ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc) and the OpenCV 2.4
 * primitives it calls.  Compile with -ffp-contract=off.
 * Parity unpinned at the OpenCV boundary (no reference fixtures exist).
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime under -std=c11 */
#include "orb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ A1 rounding */
/* cvRound on x86-64 = cvtsd2si under round-to-nearest-even. */
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { return (int)floor(v); }
static inline int cv_ceil(double v) { return (int)ceil(v); }

/* A6: cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

static const int8_t k_pattern[1024] = {
#include "../include/orbgpu_pattern.inc"
};

const int8_t *ora_get_pattern(void) { return k_pattern; }

/* ------------------------------------------------------------------ state */
typedef struct {
    ora_corner *v;
    int n, cap;
} corner_vec;

static void cv_push(corner_vec *c, ora_corner k)
{
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 1024;
        c->v = (ora_corner *)realloc(c->v, (size_t)c->cap * sizeof(ora_corner));
    }
    c->v[c->n++] = k;
}

struct ora_extractor {
    int nfeatures;
    double scale_factor; /* ORBextractor.h:98 stores the ctor's float in a double member */
    int nlevels, ini_th, min_th;
    float scale[ORA_MAX_LEVELS], inv_scale[ORA_MAX_LEVELS];
    float sigma2[ORA_MAX_LEVELS], inv_sigma2[ORA_MAX_LEVELS];
    int quota[ORA_MAX_LEVELS];
    int umax[16];
    /* per-call state */
    int w[ORA_MAX_LEVELS], h[ORA_MAX_LEVELS], pitch[ORA_MAX_LEVELS];
    uint8_t *pyr[ORA_MAX_LEVELS];
    uint8_t *blur[ORA_MAX_LEVELS];
    int blurred[ORA_MAX_LEVELS];
    corner_vec cand[ORA_MAX_LEVELS];
    corner_vec sel[ORA_MAX_LEVELS];
    /* CPU-baseline aid (bench.py): seconds spent per stage since creation / the last read */
    double stage_s[ORA_STAGE_COUNT];
};

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ E0 */
/* ORBextractor.cc:410-470 */
ora_extractor *ora_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > ORA_MAX_LEVELS)
        return NULL;
    ora_extractor *e = (ora_extractor *)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures;
    e->scale_factor = (double)scale_factor;
    e->nlevels = nlevels;
    e->ini_th = ini_th;
    e->min_th = min_th;

    e->scale[0] = 1.0f;
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->scale[i] = (float)((double)e->scale[i - 1] * e->scale_factor); /* :421 float*double */
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }

    float factor = (float)(1.0 / e->scale_factor); /* :435, 1.0f/double -> double -> float */
    float desired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->quota[level] = cv_round(desired);
        sum += e->quota[level];
        desired *= factor;
    }
    e->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;

    /* umax, :452-469 (HALF_PATCH_SIZE = 15) */
    const int HP = 15;
    int v, v0;
    int vmax = cv_floor(HP * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HP * sqrtf(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v)
        e->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1])
            ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

void ora_extractor_destroy(ora_extractor *e)
{
    if (!e)
        return;
    for (int l = 0; l < ORA_MAX_LEVELS; l++) {
        free(e->pyr[l]);
        free(e->blur[l]);
        free(e->cand[l].v);
        free(e->sel[l].v);
    }
    free(e);
}

int ora_get_levels(const ora_extractor *e) { return e->nlevels; }
const float *ora_get_scale_factors(const ora_extractor *e) { return e->scale; }
const float *ora_get_inv_scale_factors(const ora_extractor *e) { return e->inv_scale; }
const float *ora_get_sigma2(const ora_extractor *e) { return e->sigma2; }
const float *ora_get_inv_sigma2(const ora_extractor *e) { return e->inv_sigma2; }
const int *ora_get_quotas(const ora_extractor *e) { return e->quota; }
const int *ora_get_umax(const ora_extractor *e) { return e->umax; }

/* ------------------------------------------------------------------ A2 resize */
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void ora_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw, int dh,
                          size_t dstride)
{
    /* cv::resize: inv_scale = (double)dsize/ssize; scale = 1./inv_scale */
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *rows[2];
    rows[0] = (int *)malloc(sizeof(int) * (size_t)dw);
    rows[1] = (int *)malloc(sizeof(int) * (size_t)dw);

    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx >= sw - 1) {
            fx = 0;
            sx = sw - 1;
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short(cv_round((1.f - fx) * 2048));
        ialpha[dx * 2 + 1] = sat_short(cv_round(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        short b0 = sat_short(cv_round((1.f - fy) * 2048));
        short b1 = sat_short(cv_round(fy * 2048));
        for (int k = 0; k < 2; k++) {
            int r = sy + k;
            r = r < 0 ? 0 : (r < sh ? r : sh - 1);
            const uint8_t *S = src + (size_t)r * sstride;
            int *D = rows[k];
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                int sx1 = sx + 1 < sw ? sx + 1 : sw - 1; /* weight 0 when clamped */
                D[dx] = S[sx] * ialpha[dx * 2] + S[sx1] * ialpha[dx * 2 + 1];
            }
        }
        uint8_t *d = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            d[x] = (uint8_t)((((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs);
    free(ialpha);
    free(rows[0]);
    free(rows[1]);
}

/* ------------------------------------------------------------------ A6 border */
void ora_border_reflect101_u8(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride,
                              int border)
{
    for (int y = -border; y < h + border; y++) {
        const uint8_t *S = src + (size_t)reflect101(y, h) * sstride;
        uint8_t *D = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; x++)
            D[x + border] = S[reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------ A3 blur */
void ora_gauss7_u8(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride)
{
    /* getGaussianKernel(7, 2, CV_32F) then x256 via cvRound (8-bit fixed-point path) */
    float kf[7];
    int ki[7];
    {
        double sigma = 2.0, scale2x = -0.5 / (sigma * sigma), sum = 0;
        for (int i = 0; i < 7; i++) {
            double x = i - 3.0;
            kf[i] = (float)exp(scale2x * x * x);
            sum += kf[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) {
            kf[i] = (float)(kf[i] * sum);
            ki[i] = cv_round(kf[i] * 256.f);
        }
    }
    /* same sums as the definition (row pass to int32, column pass, (s + 32768) >> 16), written so that the
     * compiler vectorises the interior: the reflect-101 index is only evaluated at the borders */
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * (size_t)h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        int *T = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            if (x >= 3 && x < w - 3)
                continue;
            int s = 0;
            for (int k = -3; k <= 3; k++)
                s += ki[k + 3] * S[reflect101(x + k, w)];
            T[x] = s;
        }
        for (int x = 3; x < w - 3; x++)
            T[x] = ki[0] * S[x - 3] + ki[1] * S[x - 2] + ki[2] * S[x - 1] + ki[3] * S[x] + ki[4] * S[x + 1] +
                   ki[5] * S[x + 2] + ki[6] * S[x + 3];
    }
    for (int y = 0; y < h; y++) {
        uint8_t *D = dst + (size_t)y * dstride;
        const int *R[7];
        for (int k = -3; k <= 3; k++)
            R[k + 3] = tmp + (size_t)reflect101(y + k, h) * w;
        for (int x = 0; x < w; x++) {
            int s = ki[0] * R[0][x] + ki[1] * R[1][x] + ki[2] * R[2][x] + ki[3] * R[3][x] + ki[4] * R[4][x] +
                    ki[5] * R[5][x] + ki[6] * R[6][x];
            int v = (s + 32768) >> 16;
            D[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------ A4 FAST-9/16 */
static const int k_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int k_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16>: largest threshold for which the pixel stays a corner. */
static int fast_corner_score(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++)
        d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0)
            continue;
        for (int j = 4; j <= 8; j++)
            a = a < d[k + j] ? a : d[k + j];
        int m = a < d[k] ? a : d[k];
        a0 = a0 > m ? a0 : m;
        m = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > m ? a0 : m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; j++)
            b = b > d[k + j] ? b : d[k + j];
        if (b >= b0)
            continue;
        for (int j = 6; j <= 8; j++)
            b = b > d[k + j] ? b : d[k + j];
        int m = b > d[k] ? b : d[k];
        b0 = b0 < m ? b0 : m;
        m = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < m ? b0 : m;
    }
    return -b0 - 1;
}

int ora_fast9_16(const uint8_t *img, int w, int h, size_t stride, int threshold, ora_corner *out, int cap)
{
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++)
        pixel[k] = k_ring_dx[k] + k_ring_dy[k] * (int)stride;
    for (int k = 16; k < 25; k++)
        pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++)
        tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    if (w < 7 || h < 7)
        return 0;
    /* score rows: non-corners are 0 (OpenCV keeps a 3-row ring; a full map is equivalent) */
    uint8_t *score = (uint8_t *)calloc((size_t)w * (size_t)h, 1);
    for (int i = 3; i < h - 3; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = score + (size_t)i * w;
        for (int j = 3; j < w - 3; j++, ptr++) {
            int v = ptr[0];
            const uint8_t *t = &tab[0] - v + 255;
            int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
            if (d == 0)
                continue;
            d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
            d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
            d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
            if (d == 0)
                continue;
            d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
            d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
            d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
            d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
            int is_corner = 0;
            if (d & 1) {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] < vt) {
                        if (++count > K) {
                            is_corner = 1;
                            break;
                        }
                    } else
                        count = 0;
                }
            }
            if (!is_corner && (d & 2)) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] > vt) {
                        if (++count > K) {
                            is_corner = 1;
                            break;
                        }
                    } else
                        count = 0;
                }
            }
            if (is_corner)
                curr[j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
            /* a corner whose score is 0 (only possible with threshold 0) can never pass the strict
             * '>' test below, exactly as in OpenCV, so storing 0 for it is equivalent. */
        }
    }
    int n = 0;
    for (int i = 3; i < h - 3; i++) {
        const uint8_t *prev = score + (size_t)i * w;
        const uint8_t *pprev = prev - w;
        const uint8_t *curr = prev + w;
        for (int j = 3; j < w - 3; j++) {
            int s = prev[j];
            if (s == 0)
                continue; /* not a corner, or corner with score 0 (never survives NMS) */
            if (s > prev[j + 1] && s > prev[j - 1] && s > pprev[j - 1] && s > pprev[j] && s > pprev[j + 1] &&
                s > curr[j - 1] && s > curr[j] && s > curr[j + 1]) {
                if (n < cap) {
                    out[n].x = j;
                    out[n].y = i;
                    out[n].response = s;
                }
                n++;
            }
        }
    }
    free(score);
    return n;
}

/* ------------------------------------------------------------------ E3 quadtree */
/* DistributeOctTree, ORBextractor.cc:539-763, with ExtractorNode::DivideNode :481-537.
 * std::list is modelled by an index-linked list over a node pool; node key sets are stable
 * segments of a permutation array (children keep the parent's key order, like push_back).
 * Tie-break convention for the (size, pointer) sort at :684 (the reference compares heap
 * addresses, which is not deterministic): the node created LATER compares greater. */
typedef struct {
    int ulx, uly, brx, bry;
    int first, count; /* segment of perm[] */
    int prev, next;
    int nomore;
    int seq;
} qnode;

typedef struct {
    qnode *n;
    int used, cap;
    int head, tail, size;
    int seq;
} qlist;

static int q_new(qlist *L)
{
    if (L->used == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 256;
        L->n = (qnode *)realloc(L->n, (size_t)L->cap * sizeof(qnode));
    }
    int id = L->used++;
    memset(&L->n[id], 0, sizeof(qnode));
    L->n[id].seq = L->seq++;
    L->n[id].prev = L->n[id].next = -1;
    return id;
}
static void q_push_front(qlist *L, int id)
{
    L->n[id].prev = -1;
    L->n[id].next = L->head;
    if (L->head >= 0)
        L->n[L->head].prev = id;
    L->head = id;
    if (L->tail < 0)
        L->tail = id;
    L->size++;
}
static void q_push_back(qlist *L, int id)
{
    L->n[id].next = -1;
    L->n[id].prev = L->tail;
    if (L->tail >= 0)
        L->n[L->tail].next = id;
    L->tail = id;
    if (L->head < 0)
        L->head = id;
    L->size++;
}
static void q_erase(qlist *L, int id)
{
    int p = L->n[id].prev, nx = L->n[id].next;
    if (p >= 0)
        L->n[p].next = nx;
    else
        L->head = nx;
    if (nx >= 0)
        L->n[nx].prev = p;
    else
        L->tail = p;
    L->size--;
}

/* DivideNode: creates up to 4 children (only non-empty ones are linked by the caller).
 * child ids returned in c[4] (-1 for empty). */
static void q_divide(qlist *L, int id, const ora_corner *keys, int *perm, int *scratch, int c[4])
{
    qnode P = L->n[id];
    const int halfX = (int)ceilf((float)(P.brx - P.ulx) / 2);
    const int halfY = (int)ceilf((float)(P.bry - P.uly) / 2);
    const int midx = P.ulx + halfX, midy = P.uly + halfY;
    int cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < P.count; i++) {
        const ora_corner *k = &keys[perm[P.first + i]];
        int q = (k->x < midx) ? ((k->y < midy) ? 0 : 2) : ((k->y < midy) ? 1 : 3);
        cnt[q]++;
    }
    int off[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
    int pos[4] = {off[0], off[1], off[2], off[3]};
    for (int i = 0; i < P.count; i++) {
        int ki = perm[P.first + i];
        const ora_corner *k = &keys[ki];
        int q = (k->x < midx) ? ((k->y < midy) ? 0 : 2) : ((k->y < midy) ? 1 : 3);
        scratch[pos[q]++] = ki;
    }
    memcpy(perm + P.first, scratch, sizeof(int) * (size_t)P.count);
    const int bx[4][2] = {{P.ulx, midx}, {midx, P.brx}, {P.ulx, midx}, {midx, P.brx}};
    const int by[4][2] = {{P.uly, midy}, {P.uly, midy}, {midy, P.bry}, {midy, P.bry}};
    for (int q = 0; q < 4; q++) {
        if (cnt[q] == 0) {
            c[q] = -1;
            continue;
        }
        int ch = q_new(L);
        qnode *C = &L->n[ch];
        C->ulx = bx[q][0];
        C->brx = bx[q][1];
        C->uly = by[q][0];
        C->bry = by[q][1];
        C->first = P.first + off[q];
        C->count = cnt[q];
        C->nomore = cnt[q] == 1;
        c[q] = ch;
    }
}

static _Thread_local qlist *g_sort_list; /* per thread: the CPU baseline runs one extractor per thread */
static int q_cmp_size_seq(const void *a, const void *b)
{
    const qnode *A = &g_sort_list->n[*(const int *)a], *B = &g_sort_list->n[*(const int *)b];
    if (A->count != B->count)
        return A->count < B->count ? -1 : 1;
    return A->seq < B->seq ? -1 : (A->seq > B->seq ? 1 : 0);
}

int ora_distribute_octtree(const ora_corner *keys, int nkeys, int minX, int maxX, int minY, int maxY, int N,
                           ora_corner *out, int cap)
{
    /* :542 round() = half away from zero */
    const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1)
        return -1;
    const float hX = (float)(maxX - minX) / (float)nIni;

    qlist L;
    memset(&L, 0, sizeof(L));
    L.head = L.tail = -1;
    int *perm = (int *)malloc(sizeof(int) * (size_t)(nkeys > 0 ? nkeys : 1));
    int *scratch = (int *)malloc(sizeof(int) * (size_t)(nkeys > 0 ? nkeys : 1));
    int *ini_cnt = (int *)calloc((size_t)nIni, sizeof(int));
    int *ini_of = (int *)malloc(sizeof(int) * (size_t)(nkeys > 0 ? nkeys : 1));

    /* :567-571 vpIniNodes[kp.pt.x/hX] (float division, truncation) */
    for (int i = 0; i < nkeys; i++) {
        int b = (int)((float)keys[i].x / hX);
        if (b < 0)
            b = 0;
        if (b >= nIni)
            b = nIni - 1; /* unreachable for x < maxX-minX; guards the oracle against UB */
        ini_of[i] = b;
        ini_cnt[b]++;
    }
    int *ini_first = (int *)malloc(sizeof(int) * (size_t)nIni);
    int acc = 0;
    for (int b = 0; b < nIni; b++) {
        ini_first[b] = acc;
        acc += ini_cnt[b];
    }
    {
        int *pos = (int *)malloc(sizeof(int) * (size_t)nIni);
        memcpy(pos, ini_first, sizeof(int) * (size_t)nIni);
        for (int i = 0; i < nkeys; i++)
            perm[pos[ini_of[i]]++] = i;
        free(pos);
    }
    /* :552-563 initial nodes (push_back), then :573-585 drop empty ones, flag singletons */
    for (int b = 0; b < nIni; b++) {
        if (ini_cnt[b] == 0)
            continue;
        int id = q_new(&L);
        qnode *n = &L.n[id];
        n->ulx = (int)(hX * (float)b);
        n->brx = (int)(hX * (float)(b + 1));
        n->uly = 0;
        n->bry = maxY - minY;
        n->first = ini_first[b];
        n->count = ini_cnt[b];
        n->nomore = ini_cnt[b] == 1;
        q_push_back(&L, id);
    }
    free(ini_cnt);
    free(ini_of);
    free(ini_first);

    int *vsz = NULL, vsz_n = 0, vsz_cap = 0;
    int *vprev = NULL, vprev_cap = 0;
#define VSZ_PUSH(id)                                                                                               \
    do {                                                                                                           \
        if (vsz_n == vsz_cap) {                                                                                    \
            vsz_cap = vsz_cap ? vsz_cap * 2 : 256;                                                                 \
            vsz = (int *)realloc(vsz, sizeof(int) * (size_t)vsz_cap);                                              \
        }                                                                                                          \
        vsz[vsz_n++] = (id);                                                                                       \
    } while (0)

    int finish = 0;
    while (!finish) {
        int prevSize = L.size;
        int nToExpand = 0;
        vsz_n = 0;
        int it = L.head;
        while (it >= 0) {
            if (L.n[it].nomore) {
                it = L.n[it].next;
                continue;
            }
            int c[4];
            q_divide(&L, it, keys, perm, scratch, c);
            for (int q = 0; q < 4; q++) {
                if (c[q] < 0)
                    continue;
                q_push_front(&L, c[q]);
                if (L.n[c[q]].count > 1) {
                    nToExpand++;
                    VSZ_PUSH(c[q]);
                }
            }
            int nx = L.n[it].next;
            q_erase(&L, it);
            it = nx;
        }
        if (L.size >= N || L.size == prevSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                if (vsz_n > vprev_cap) {
                    vprev_cap = vsz_n;
                    vprev = (int *)realloc(vprev, sizeof(int) * (size_t)vprev_cap);
                }
                int np = vsz_n;
                memcpy(vprev, vsz, sizeof(int) * (size_t)np);
                vsz_n = 0;
                g_sort_list = &L;
                qsort(vprev, (size_t)np, sizeof(int), q_cmp_size_seq); /* total order: no instability */
                for (int j = np - 1; j >= 0; j--) {
                    int c[4];
                    q_divide(&L, vprev[j], keys, perm, scratch, c);
                    for (int q = 0; q < 4; q++) {
                        if (c[q] < 0)
                            continue;
                        q_push_front(&L, c[q]);
                        if (L.n[c[q]].count > 1)
                            VSZ_PUSH(c[q]);
                    }
                    q_erase(&L, vprev[j]);
                    if (L.size >= N)
                        break;
                }
                if (L.size >= N || L.size == prevSize)
                    finish = 1;
            }
        }
    }
#undef VSZ_PUSH

    /* :741-760 keep the best response of every node (first maximum in key order) */
    int nout = 0;
    for (int it = L.head; it >= 0; it = L.n[it].next) {
        const qnode *n = &L.n[it];
        int best = perm[n->first];
        for (int k = 1; k < n->count; k++) {
            int ki = perm[n->first + k];
            if (keys[ki].response > keys[best].response)
                best = ki;
        }
        if (nout < cap)
            out[nout] = keys[best];
        nout++;
    }
    free(L.n);
    free(perm);
    free(scratch);
    free(vsz);
    free(vprev);
    return nout;
}

/* ------------------------------------------------------------------ E5 orientation */
/* A5: cv::fastAtan2 (OpenCV 2.4 polynomial), degrees in [0,360). */
float ora_fast_atan2(float y, float x)
{
    static const float s = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}

/* IC_Angle, ORBextractor.cc:77-104 */
float ora_ic_angle(const uint8_t *center, int step, const int *umax)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -15; u <= 15; ++u)
        m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return ora_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------ E7 descriptor */
/* ORBextractor.cc:112-113:  float a = (float)cos(angle), b = (float)sin(angle);  with a float `angle`.  The file has
 * `using namespace cv; using namespace std;` (:66-67), so overload resolution takes std::cos(float) / std::sin(float),
 * which are libm's cosf / sinf -- NOT the double functions.  cosf / sinf are not correctly rounded, so their last bit
 * belongs to the libm (and CPU variant) of the host the reference runs on.  ORA_TRIG_LIBM_FLOAT (default) calls this
 * host's cosf / sinf, i.e. what the reference computes on this machine; ORA_TRIG_ROUNDED_DOUBLE is the correctly rounded
 * value (float)cos((double)angle) -- the restatement rounds 1 to 3 had -- kept so that the residual between the two
 * can be counted (tests/test_trig.py) and liborbgpu's ORBGPU_TRIG_ROUNDED_DOUBLE mode checked. */
static int g_trig_mode = ORA_TRIG_LIBM_FLOAT;
void ora_set_trig_mode(int mode) { g_trig_mode = mode == ORA_TRIG_ROUNDED_DOUBLE ? ORA_TRIG_ROUNDED_DOUBLE : ORA_TRIG_LIBM_FLOAT; }
int ora_get_trig_mode(void) { return g_trig_mode; }
void ora_descriptor_trig(float angle_rad, float *a, float *b)
{
    if (g_trig_mode == ORA_TRIG_ROUNDED_DOUBLE) {
        *a = (float)cos((double)angle_rad);
        *b = (float)sin((double)angle_rad);
    } else {
        *a = cosf(angle_rad);
        *b = sinf(angle_rad);
    }
}

void ora_descriptor_trig_array(const float *angle_rad, int n, float *a, float *b)
{
    for (int i = 0; i < n; i++)
        ora_descriptor_trig(angle_rad[i], a + i, b + i);
}

/* computeOrbDescriptor, ORBextractor.cc:108-147 */
void ora_orb_descriptor(const uint8_t *center, int step, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f); /* :107 */
    float angle = angle_deg * factorPI;
    float a, b;
    ora_descriptor_trig(angle, &a, &b);
    const int8_t *p = k_pattern;
    for (int i = 0; i < 32; ++i, p += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            float x0 = (float)p[4 * k], y0 = (float)p[4 * k + 1];
            float x1 = (float)p[4 * k + 2], y1 = (float)p[4 * k + 3];
            int t0 = center[cv_round(x0 * b + y0 * a) * step + cv_round(x0 * a - y0 * b)];
            int t1 = center[cv_round(x1 * b + y1 * a) * step + cv_round(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------ E1 pyramid */
/* ComputePyramid, ORBextractor.cc:1107-1132 */
static void compute_pyramid(ora_extractor *e, const uint8_t *gray, int w, int h, size_t stride)
{
    for (int level = 0; level < e->nlevels; ++level) {
        float scale = e->inv_scale[level];
        int lw = cv_round((float)w * scale), lh = cv_round((float)h * scale);
        int pitch = lw + 2 * ORA_EDGE;
        e->w[level] = lw;
        e->h[level] = lh;
        e->pitch[level] = pitch;
        free(e->pyr[level]);
        e->pyr[level] = (uint8_t *)malloc((size_t)pitch * (size_t)(lh + 2 * ORA_EDGE));
        uint8_t *origin = e->pyr[level] + (size_t)ORA_EDGE * pitch + ORA_EDGE;
        if (level != 0) {
            const uint8_t *prev = e->pyr[level - 1] + (size_t)ORA_EDGE * e->pitch[level - 1] + ORA_EDGE;
            uint8_t *tmp = (uint8_t *)malloc((size_t)lw * (size_t)lh);
            ora_resize_linear_u8(prev, e->w[level - 1], e->h[level - 1], (size_t)e->pitch[level - 1], tmp, lw, lh,
                                 (size_t)lw);
            ora_border_reflect101_u8(tmp, lw, lh, (size_t)lw, e->pyr[level], (size_t)pitch, ORA_EDGE);
            free(tmp);
        } else {
            ora_border_reflect101_u8(gray, w, h, stride, e->pyr[level], (size_t)pitch, ORA_EDGE);
        }
        (void)origin;
    }
}

/* ------------------------------------------------------------------ E2..E5 */
/* ComputeKeyPointsOctTree, ORBextractor.cc:765-853 (orientation is done by the caller) */
static void compute_keypoints_level(ora_extractor *e, int level)
{
    const float W = 30;
    const int minBorderX = ORA_EDGE - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = e->w[level] - ORA_EDGE + 3;
    const int maxBorderY = e->h[level] - ORA_EDGE + 3;
    corner_vec *cand = &e->cand[level];
    corner_vec *sel = &e->sel[level];
    cand->n = 0;
    sel->n = 0;
    double t_begin = now_s();

    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1)
        return;
    const int wCell = (int)ceilf(width / nCols);
    const int hCell = (int)ceilf(height / nRows);

    const int pitch = e->pitch[level];
    const uint8_t *origin = e->pyr[level] + (size_t)ORA_EDGE * pitch + ORA_EDGE;
    int cellcap = (wCell + 6) * (hCell + 6);
    ora_corner *cell = (ora_corner *)malloc(sizeof(ora_corner) * (size_t)cellcap);

    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3)
            continue;
        if (maxY > maxBorderY)
            maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6)
                continue;
            if (maxX > maxBorderX)
                maxX = (float)maxBorderX;
            int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
            const uint8_t *sub = origin + (size_t)y0 * pitch + x0;
            int n = ora_fast9_16(sub, x1 - x0, y1 - y0, (size_t)pitch, e->ini_th, cell, cellcap);
            if (n == 0)
                n = ora_fast9_16(sub, x1 - x0, y1 - y0, (size_t)pitch, e->min_th, cell, cellcap);
            for (int k = 0; k < n; k++) {
                ora_corner c = cell[k];
                c.x += j * wCell;
                c.y += i * hCell;
                cv_push(cand, c);
            }
        }
    }
    free(cell);
    e->stage_s[ORA_STAGE_FAST] += now_s() - t_begin;
    t_begin = now_s();

    int cap = cand->n > 0 ? cand->n : 1;
    if (sel->cap < cap) {
        sel->cap = cap;
        sel->v = (ora_corner *)realloc(sel->v, sizeof(ora_corner) * (size_t)cap);
    }
    int n = ora_distribute_octtree(cand->v, cand->n, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                   e->quota[level], sel->v, cap);
    sel->n = n < 0 ? 0 : n;
    e->stage_s[ORA_STAGE_QUADTREE] += now_s() - t_begin;
}

/* ------------------------------------------------------------------ E8 operator() */
int ora_extract(ora_extractor *e, const uint8_t *gray, int w, int h, size_t stride, ora_keypoint *kps,
                uint8_t *desc, int cap)
{
    if (w <= 0 || h <= 0 || !gray)
        return 0; /* :1046 empty image */
    double t0 = now_s();
    compute_pyramid(e, gray, w, h, stride);
    e->stage_s[ORA_STAGE_PYRAMID] += now_s() - t0;
    int total = 0;
    for (int level = 0; level < e->nlevels; level++) {
        compute_keypoints_level(e, level);
        total += e->sel[level].n;
    }
    if (total > cap)
        return -1;

    int offset = 0;
    for (int level = 0; level < e->nlevels; level++) {
        e->blurred[level] = 0;
        int n = e->sel[level].n;
        if (n == 0)
            continue;
        const int pitch = e->pitch[level];
        const uint8_t *origin = e->pyr[level] + (size_t)ORA_EDGE * pitch + ORA_EDGE;
        int lw = e->w[level], lh = e->h[level];
        /* :1085-1086 clone (drops the border) + GaussianBlur(7x7, 2, 2, REFLECT_101) */
        free(e->blur[level]);
        e->blur[level] = (uint8_t *)malloc((size_t)lw * (size_t)lh);
        t0 = now_s();
        uint8_t *clone = (uint8_t *)malloc((size_t)lw * (size_t)lh);
        for (int y = 0; y < lh; y++)
            memcpy(clone + (size_t)y * lw, origin + (size_t)y * pitch, (size_t)lw);
        ora_gauss7_u8(clone, lw, lh, (size_t)lw, e->blur[level], (size_t)lw);
        free(clone);
        e->blurred[level] = 1;
        e->stage_s[ORA_STAGE_BLUR] += now_s() - t0;

        const int scaledPatchSize = (int)(31 * e->scale[level]); /* :837 */
        const float scale = e->scale[level];
        for (int k = 0; k < n; k++) {
            ora_corner c = e->sel[level].v[k];
            ora_keypoint kp;
            kp.x = (float)c.x + (float)(ORA_EDGE - 3); /* :843-844 */
            kp.y = (float)c.y + (float)(ORA_EDGE - 3);
            kp.size = (float)scaledPatchSize;
            kp.response = (float)c.response;
            kp.octave = level;
            kp.class_id = -1;
            int px = cv_round(kp.x), py = cv_round(kp.y);
            /* :851-852 orientation on the un-blurred level */
            t0 = now_s();
            kp.angle = ora_ic_angle(origin + (size_t)py * pitch + px, pitch, e->umax);
            const double t1 = now_s();
            e->stage_s[ORA_STAGE_ORIENT] += t1 - t0;
            /* :1089-1090 descriptor on the blurred level */
            ora_orb_descriptor(e->blur[level] + (size_t)py * lw + px, lw, kp.angle, desc + (size_t)(offset + k) * 32);
            e->stage_s[ORA_STAGE_DESCRIBE] += now_s() - t1;
            if (level != 0) { /* :1095-1101 */
                kp.x *= scale;
                kp.y *= scale;
            }
            kps[offset + k] = kp;
        }
        offset += n;
    }
    return total;
}

void ora_extractor_stage_seconds(ora_extractor *e, double out[ORA_STAGE_COUNT], int reset)
{
    for (int i = 0; i < ORA_STAGE_COUNT; i++) {
        out[i] = e->stage_s[i];
        if (reset)
            e->stage_s[i] = 0.0;
    }
}

const uint8_t *ora_pyramid_level(const ora_extractor *e, int level, int *w, int *h, int *pitch)
{
    *w = e->w[level];
    *h = e->h[level];
    *pitch = e->pitch[level];
    return e->pyr[level];
}
const uint8_t *ora_blurred_level(const ora_extractor *e, int level, int *w, int *h)
{
    *w = e->w[level];
    *h = e->h[level];
    return e->blurred[level] ? e->blur[level] : NULL;
}
int ora_level_candidates(const ora_extractor *e, int level, const ora_corner **out)
{
    *out = e->cand[level].v;
    return e->cand[level].n;
}
int ora_level_selected(const ora_extractor *e, int level, const ora_corner **out)
{
    *out = e->sel[level].v;
    return e->sel[level].n;
}
