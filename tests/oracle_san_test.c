/* Sanitizer run of the CPU oracle (test infrastructure): built by tests/test_sanitizers.py with
 * gcc -fsanitize=address,undefined against oracle/\*.c and executed on small seeded inputs covering the extractor,
 * the matchers, the dense map and the vocabulary code.  Exit code 0 = no finding (ASAN / UBSAN abort otherwise). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

static unsigned s_rng = 12345u;
static unsigned rnd(void)
{
    s_rng = s_rng * 1664525u + 1013904223u;
    return s_rng >> 8;
}

int main(void)
{
    const int w = 320, h = 240;
    uint8_t *img = (uint8_t *)malloc((size_t)w * h);
    for (int i = 0; i < w * h; i++)
        img[i] = (uint8_t)(100 + rnd() % 30);
    for (int r = 0; r < 120; r++) { /* rectangles: FAST corners */
        int x0 = rnd() % (w - 20), y0 = rnd() % (h - 20), ww = 4 + rnd() % 16, hh = 4 + rnd() % 16, v = 40 + rnd() % 160;
        for (int y = y0; y < y0 + hh; y++)
            for (int x = x0; x < x0 + ww; x++)
                img[y * w + x] = (uint8_t)v;
    }
    ora_extractor *e = ora_extractor_create(500, 1.2f, 8, 20, 7);
    const int cap = 700;
    ora_keypoint *kps = (ora_keypoint *)malloc(sizeof(ora_keypoint) * cap);
    uint8_t *desc = (uint8_t *)malloc((size_t)cap * 32);
    const int n = ora_extract(e, img, w, h, (size_t)w, kps, desc, cap);
    if (n <= 50) {
        fprintf(stderr, "only %d key points\n", n);
        return 1;
    }
    /* brute-force matcher against a bit-noisy copy */
    uint8_t *desc2 = (uint8_t *)malloc((size_t)n * 32);
    float *ang = (float *)malloc(sizeof(float) * n);
    for (int i = 0; i < n; i++) {
        ang[i] = kps[i].angle;
        memcpy(desc2 + (size_t)i * 32, desc + (size_t)i * 32, 32);
        desc2[(size_t)i * 32 + rnd() % 32] ^= (uint8_t)(1u << (rnd() % 8));
    }
    int32_t *match = (int32_t *)malloc(sizeof(int32_t) * n);
    const int nm = ora_match_bf(desc, ang, NULL, n, desc2, ang, n, 50, 0.9f, 1, match);
    /* vocabulary: k = 4, L = 3 (85 nodes), random descriptors */
    const int k = 4, L = 3, nn = 1 + 4 + 16 + 64;
    int32_t *parent = (int32_t *)calloc(nn, sizeof(int32_t));
    uint8_t *leaf = (uint8_t *)calloc(nn, 1), *vdesc = (uint8_t *)malloc((size_t)nn * 32);
    double *weight = (double *)calloc(nn, sizeof(double));
    for (int i = 1; i < nn; i++) {
        parent[i] = (i - 1) / 4;
        leaf[i] = i >= 21;
        weight[i] = leaf[i] ? 0.5 + (rnd() % 100) / 50.0 : 0.0;
        for (int b = 0; b < 32; b++)
            vdesc[(size_t)i * 32 + b] = (uint8_t)rnd();
    }
    ora_vocabulary *voc = ora_vocabulary_create(k, L, nn, parent, leaf, vdesc, weight, 0, 0);
    int32_t *wid = (int32_t *)malloc(sizeof(int32_t) * n), *nid = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *bid = (int32_t *)malloc(sizeof(int32_t) * n), *fvn = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *fvs = (int32_t *)malloc(sizeof(int32_t) * (n + 1)), *fvi = (int32_t *)malloc(sizeof(int32_t) * n);
    double *wgt = (double *)malloc(sizeof(double) * n), *bval = (double *)malloc(sizeof(double) * n);
    int32_t nb = 0, nf = 0;
    ora_bow_transform(voc, desc, n, 1, wid, wgt, nid, bid, bval, &nb, fvn, fvs, fvi, &nf);
    int32_t *match2 = (int32_t *)malloc(sizeof(int32_t) * n);
    const int nbw = ora_search_by_bow(desc, ang, NULL, nf, fvn, fvs, fvi, desc, ang, n, nf, fvn, fvs, fvi, 50, 0.9f, 1, match2);
    /* dense map: random points, two leaf sizes, overflow rule */
    const int np = 20000;
    ora_point *pts = (ora_point *)malloc(sizeof(ora_point) * np), *out = (ora_point *)malloc(sizeof(ora_point) * np);
    for (int i = 0; i < np; i++) {
        pts[i].x = (float)(rnd() % 4000) / 1000.f - 2.f;
        pts[i].y = (float)(rnd() % 3000) / 1000.f;
        pts[i].z = (float)(rnd() % 5000) / 1000.f;
        pts[i].rgba = rnd();
    }
    int ov = 0;
    const int v1 = ora_voxel_filter(pts, np, 0.05f, out, &ov);
    pts[0].x = 5000.f;
    const int v2 = ora_voxel_filter(pts, np, 0.001f, out, &ov);
    const int nsor = np < 1500 ? np : 1500;
    float *md = (float *)malloc(sizeof(float) * (size_t)nsor);
    const int v3 = ora_statistical_outlier_removal(pts, nsor, 50, 1.0, out, md);
    free(md);
    printf("sanitizer run ok: %d key points, %d bf matches, %d bow words, %d bow matches, %d / %d voxels (overflow %d), %d of %d "
           "points kept by the outlier filter\n", n, nm, nb, nbw, v1, v2, ov, v3, nsor);
    ora_vocabulary_destroy(voc);
    ora_extractor_destroy(e);
    free(img), free(kps), free(desc), free(desc2), free(ang), free(match), free(match2), free(parent), free(leaf);
    free(vdesc), free(weight), free(wid), free(nid), free(bid), free(fvn), free(fvs), free(fvi), free(wgt), free(bval);
    free(pts), free(out);
    return 0;
}
