/*
 * orb_oracle_match.c -- CPU ORACLE (test infrastructure, NOT product code; see orb_oracle.h).
 *
 * Restates ORB_SLAM2::ORBmatcher (reference src/ORBmatcher.cc) and the Frame helpers it reads
 * (reference src/Frame.cc).  Compile with -ffp-contract=off.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ M0 */
/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:1647-1663 (bit-twiddling popcount) */
int ora_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1601-1642 (on bin sizes) */
static void three_maxima(const int *histo, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            *ind3 = *ind2;
            *ind2 = *ind1;
            *ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            *ind3 = *ind2;
            *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        *ind3 = -1;
    }
}

/* rotation bin, ORBmatcher.cc:238-243 / 1433-1438 */
static int rot_bin(float angle_a, float angle_b)
{
    const float factor = 1.0f / ORA_HISTO_LENGTH;
    float rot = angle_a - angle_b;
    if (rot < 0.0)
        rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORA_HISTO_LENGTH)
        bin = 0;
    return bin;
}

/* ------------------------------------------------------------------ M4 */
int ora_match_bf(const uint8_t *desc_a, const float *angle_a, const uint8_t *valid_a, int na,
                 const uint8_t *desc_b, const float *angle_b, int nb, int th_low, float nnratio,
                 int check_orientation, int32_t *match_b)
{
    int nmatches = 0;
    int *bin_of = (int *)malloc(sizeof(int) * (size_t)(nb > 0 ? nb : 1));
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    for (int j = 0; j < nb; j++) {
        match_b[j] = -1;
        bin_of[j] = -1;
    }
    for (int i = 0; i < na; i++) {
        if (valid_a && !valid_a[i])
            continue; /* :195-199 pMP NULL or bad */
        const uint8_t *da = desc_a + (size_t)i * 32;
        int best1 = 256, bestIdx = -1, best2 = 256;
        for (int j = 0; j < nb; j++) {
            if (match_b[j] >= 0)
                continue; /* :209-210 */
            int dist = ora_descriptor_distance(da, desc_b + (size_t)j * 32);
            if (dist < best1) {
                best2 = best1;
                best1 = dist;
                bestIdx = j;
            } else if (dist < best2) {
                best2 = dist;
            }
        }
        /* bestIdx < 0: every B row is taken.  With the reference's TH_LOW = 50 the test below already fails (best1 is
         * still 256); the guard only matters for thresholds >= 256, which this entry point accepts */
        if (best1 <= th_low && bestIdx >= 0) {
            if ((float)best1 < nnratio * (float)best2) {
                match_b[bestIdx] = i;
                if (check_orientation) {
                    int bin = rot_bin(angle_a[i], angle_b[bestIdx]);
                    bin_of[bestIdx] = bin;
                    histo[bin]++;
                }
                nmatches++;
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(histo, ORA_HISTO_LENGTH, &i1, &i2, &i3);
        for (int j = 0; j < nb; j++) {
            int b = bin_of[j];
            if (b < 0 || b == i1 || b == i2 || b == i3)
                continue;
            match_b[j] = -1;
            nmatches--;
        }
    }
    free(bin_of);
    return nmatches;
}

/* ------------------------------------------------------------------ M7 */
/* Frame::PosInGrid Frame.cc:382-392 + AssignFeaturesToGrid :230-245 */
void ora_assign_features_to_grid(int n, const float *kp_x, const float *kp_y, float min_x, float min_y,
                                 float inv_w, float inv_h, int32_t *cell_start, int32_t *cell_items)
{
    const int NC = ORA_GRID_COLS * ORA_GRID_ROWS;
    int *cell_of = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    memset(cell_start, 0, sizeof(int32_t) * (size_t)(NC + 1));
    for (int i = 0; i < n; i++) {
        int px = (int)roundf((kp_x[i] - min_x) * inv_w);
        int py = (int)roundf((kp_y[i] - min_y) * inv_h);
        if (px < 0 || px >= ORA_GRID_COLS || py < 0 || py >= ORA_GRID_ROWS) {
            cell_of[i] = -1;
            continue;
        }
        cell_of[i] = px * ORA_GRID_ROWS + py;
        cell_start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < NC; c++)
        cell_start[c + 1] += cell_start[c];
    int *pos = (int *)malloc(sizeof(int) * (size_t)NC);
    for (int c = 0; c < NC; c++)
        pos[c] = cell_start[c];
    for (int i = 0; i < n; i++)
        if (cell_of[i] >= 0)
            cell_items[pos[cell_of[i]]++] = i;
    free(pos);
    free(cell_of);
}

/* Frame::GetFeaturesInArea, Frame.cc:327-380 */
int ora_get_features_in_area(const ora_frame_view *f, float x, float y, float r, int min_level, int max_level,
                             int32_t *out)
{
    int n = 0;
    int c0 = (int)floorf((x - f->min_x - r) * f->grid_inv_w);
    const int nMinCellX = c0 > 0 ? c0 : 0;
    if (nMinCellX >= ORA_GRID_COLS)
        return 0;
    int c1 = (int)ceilf((x - f->min_x + r) * f->grid_inv_w);
    const int nMaxCellX = c1 < ORA_GRID_COLS - 1 ? c1 : ORA_GRID_COLS - 1;
    if (nMaxCellX < 0)
        return 0;
    int r0 = (int)floorf((y - f->min_y - r) * f->grid_inv_h);
    const int nMinCellY = r0 > 0 ? r0 : 0;
    if (nMinCellY >= ORA_GRID_ROWS)
        return 0;
    int r1 = (int)ceilf((y - f->min_y + r) * f->grid_inv_h);
    const int nMaxCellY = r1 < ORA_GRID_ROWS - 1 ? r1 : ORA_GRID_ROWS - 1;
    if (nMaxCellY < 0)
        return 0;

    const int bCheckLevels = (min_level > 0) || (max_level >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            int c = ix * ORA_GRID_ROWS + iy;
            for (int j = f->cell_start[c]; j < f->cell_start[c + 1]; j++) {
                int idx = f->cell_items[j];
                if (bCheckLevels) {
                    if (f->kp_octave[idx] < min_level)
                        continue;
                    if (max_level >= 0)
                        if (f->kp_octave[idx] > max_level)
                            continue;
                }
                const float distx = f->kp_x[idx] - x;
                const float disty = f->kp_y[idx] - y;
                if (fabsf(distx) < r && fabsf(disty) < r)
                    out[n++] = idx;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------ M2 */
static int kp_is_claimed(const int32_t *kp_to_mp, int idx, const uint8_t *obs_pos)
{
    int v = kp_to_mp[idx];
    if (v == -1)
        return 0;
    if (v == -2)
        return 1;
    return obs_pos ? obs_pos[v] != 0 : 1;
}

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), ORBmatcher.cc:45-129 */
int ora_search_by_projection(const ora_frame_view *f, const ora_mappoint_view *mp, float th, float nnratio,
                             int32_t *kp_to_mp)
{
    int nmatches = 0;
    const int bFactor = th != 1.0;
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(f->n > 0 ? f->n : 1));
    for (int i = 0; i < mp->m; i++) {
        if (!mp->in_view[i])
            continue;
        if (mp->bad && mp->bad[i])
            continue;
        const int lvl = mp->level[i];
        if (lvl < 0 || lvl >= f->nlevels) {
            free(vIndices);
            return -1; /* H5: the reference would index mvScaleFactors out of range */
        }
        float r = mp->view_cos[i] > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos :131-137 */
        if (bFactor)
            r *= th;
        const float rs = r * f->scale_factors[lvl];
        int nc = ora_get_features_in_area(f, mp->proj_x[i], mp->proj_y[i], rs, lvl - 1, lvl, vIndices);
        if (nc == 0)
            continue;
        const uint8_t *dmp = mp->desc + (size_t)i * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = vIndices[c];
            if (kp_is_claimed(kp_to_mp, idx, mp->obs_pos))
                continue;
            if (f->u_right[idx] > 0) {
                const float er = fabsf(mp->proj_xr[i] - f->u_right[idx]);
                if (er > rs)
                    continue;
            }
            const int dist = ora_descriptor_distance(dmp, f->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = f->kp_octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = f->kp_octave[idx];
                bestDist2 = dist;
            }
        }
        if (bestDist <= ORA_TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)
                continue;
            kp_to_mp[bestIdx] = i;
            nmatches++;
        }
    }
    free(vIndices);
    return nmatches;
}

/* ------------------------------------------------------------------ M3 */
/* cv::Mat 3x3 * 3x1 + 3x1 for CV_32F takes cv::gemm's small-matrix path (OpenCV 2.4
 * matmul.cpp): float products summed left to right, then one add of the C term.  Adopted. */
static void rt_apply(const float *T /*4x4 row-major*/, const float *p, float *out)
{
    for (int i = 0; i < 3; i++) {
        float t0 = T[4 * i + 0] * p[0] + T[4 * i + 1] * p[1] + T[4 * i + 2] * p[2];
        out[i] = t0 + T[4 * i + 3];
    }
}
/* -R^T * t : general gemm path with double accumulators, alpha = -1. */
static void minus_rt_t(const float *T, float *out)
{
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++)
            s += (double)T[4 * k + i] * (double)T[4 * k + 3];
        out[i] = (float)(s * -1.0);
    }
}

int ora_search_by_projection_last(const ora_frame_view *cur, const float *cur_Tcw, float fx, float fy, float cx,
                                  float cy, float mbf, float mb, const ora_lastframe_view *last, float th,
                                  int mono, int check_orientation, int32_t *kp_to_mp)
{
    int nmatches = 0;
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    /* rotHist[bin].push_back(bestIdx2) replayed as a flat push list (a keypoint can be pushed
     * more than once when an earlier claimant has Observations()==0, e.g. the RGB-D temporal
     * points of Tracking::UpdateLastFrame) */
    int *push_idx = (int *)malloc(sizeof(int) * (size_t)(last->n > 0 ? last->n : 1));
    int *push_bin = (int *)malloc(sizeof(int) * (size_t)(last->n > 0 ? last->n : 1));
    int npush = 0;
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cur->n > 0 ? cur->n : 1));

    float twc[3], tlc[3];
    minus_rt_t(cur_Tcw, twc);
    rt_apply(last->Tcw, twc, tlc);
    const int bForward = tlc[2] > mb && !mono;
    const int bBackward = -tlc[2] > mb && !mono;

    for (int i = 0; i < last->n; i++) {
        if (!last->has_mp[i])
            continue;
        if (last->outlier && last->outlier[i])
            continue;
        float xc3[3];
        rt_apply(cur_Tcw, last->world_pos + 3 * (size_t)i, xc3);
        const float xc = xc3[0], yc = xc3[1];
        const float invzc = (float)(1.0 / (double)xc3[2]);
        if (invzc < 0)
            continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < cur->min_x || u > cur->max_x)
            continue;
        if (v < cur->min_y || v > cur->max_y)
            continue;
        int nLastOctave = last->kp_octave[i];
        if (nLastOctave < 0 || nLastOctave >= cur->nlevels) {
            nmatches = -1;
            goto done;
        }
        float radius = th * cur->scale_factors[nLastOctave];
        int nc;
        if (bForward)
            nc = ora_get_features_in_area(cur, u, v, radius, nLastOctave, -1, vIndices);
        else if (bBackward)
            nc = ora_get_features_in_area(cur, u, v, radius, 0, nLastOctave, vIndices);
        else
            nc = ora_get_features_in_area(cur, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices);
        if (nc == 0)
            continue;
        const uint8_t *dmp = last->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = vIndices[c];
            if (kp_is_claimed(kp_to_mp, i2, last->obs_pos))
                continue;
            if (cur->u_right[i2] > 0) {
                const float ur = u - mbf * invzc;
                const float er = fabsf(ur - cur->u_right[i2]);
                if (er > radius)
                    continue;
            }
            const int dist = ora_descriptor_distance(dmp, cur->desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= ORA_TH_HIGH) {
            kp_to_mp[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                int bin = rot_bin(last->kp_angle[i], cur->kp_angle[bestIdx2]);
                push_idx[npush] = bestIdx2;
                push_bin[npush++] = bin;
                histo[bin]++;
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(histo, ORA_HISTO_LENGTH, &i1, &i2, &i3);
        for (int k = 0; k < npush; k++) {
            int b = push_bin[k];
            if (b == i1 || b == i2 || b == i3)
                continue;
            kp_to_mp[push_idx[k]] = -1;
            nmatches--;
        }
    }
done:
    free(vIndices);
    free(push_idx);
    free(push_bin);
    return nmatches;
}

/* ------------------------------------------------------------------ M5a */
/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist), :1472-1599 */
int ora_search_by_projection_keyframe(const ora_frame_view *cur, const float *cur_Tcw, float fx, float fy, float cx,
                                      float cy, float log_scale_factor, const ora_keyframe_view *kf, float th,
                                      int orb_dist, int check_orientation, int32_t *kp_to_mp)
{
    int nmatches = 0;
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    int *push_idx = (int *)malloc(sizeof(int) * (size_t)(kf->n > 0 ? kf->n : 1));
    int *push_bin = (int *)malloc(sizeof(int) * (size_t)(kf->n > 0 ? kf->n : 1));
    int npush = 0;
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cur->n > 0 ? cur->n : 1));
    float Ow[3];
    minus_rt_t(cur_Tcw, Ow);
    for (int i = 0; i < kf->n; i++) {
        if (!kf->has_mp[i])
            continue;
        if ((kf->bad && kf->bad[i]) || (kf->already_found && kf->already_found[i]))
            continue;
        const float *Pw = kf->world_pos + 3 * (size_t)i;
        float xc3[3];
        rt_apply(cur_Tcw, Pw, xc3);
        const float xc = xc3[0], yc = xc3[1];
        const float invzc = (float)(1.0 / (double)xc3[2]);
        const float u = fx * xc * invzc + cx;
        const float v = fy * yc * invzc + cy;
        if (u < cur->min_x || u > cur->max_x)
            continue;
        if (v < cur->min_y || v > cur->max_y)
            continue;
        float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        const float maxDistance = kf->max_dist_inv[i];
        const float minDistance = kf->min_dist_inv[i];
        if (dist3D < minDistance || dist3D > maxDistance)
            continue;
        float ratio = kf->max_dist[i] / dist3D;
        int lvl = (int)ceilf(logf(ratio) / log_scale_factor); /* MapPoint::PredictScale */
        if (lvl < 0 || lvl >= cur->nlevels) {
            nmatches = -1;
            goto done;
        }
        const float radius = th * cur->scale_factors[lvl];
        int nc = ora_get_features_in_area(cur, u, v, radius, lvl - 1, lvl + 1, vIndices);
        if (nc == 0)
            continue;
        const uint8_t *dmp = kf->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = vIndices[c];
            if (kp_to_mp[i2] != -1)
                continue; /* :1540-1541 any association blocks */
            const int dist = ora_descriptor_distance(dmp, cur->desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= orb_dist) {
            kp_to_mp[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                int bin = rot_bin(kf->kp_angle[i], cur->kp_angle[bestIdx2]);
                push_idx[npush] = bestIdx2;
                push_bin[npush++] = bin;
                histo[bin]++;
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(histo, ORA_HISTO_LENGTH, &i1, &i2, &i3);
        for (int k = 0; k < npush; k++) {
            int b = push_bin[k];
            if (b == i1 || b == i2 || b == i3)
                continue;
            kp_to_mp[push_idx[k]] = -1;
            nmatches--;
        }
    }
done:
    free(vIndices);
    free(push_idx);
    free(push_bin);
    return nmatches;
}

/* ------------------------------------------------------------------ M5b */
/* Sim3 decomposition of ORBmatcher.cc:299-303: scw = sqrt(row0 . row0) (Mat::dot: double), Rcw = sRcw / scw and
 * tcw = t / scw (cv::Mat / scalar = convertTo with alpha = 1/scw, applied in float for CV_32F), Ow = -Rcw^t tcw.
 * T34 receives [Rcw | tcw] rows. */
void ora_sim3_decompose(const float *Scw, float *T34, float *Ow)
{
    const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)sqrt(d);
    const float alpha = (float)(1.0 / (double)scw);
    float T[16];
    memset(T, 0, sizeof(T));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            volatile float v = Scw[4 * r + c] * alpha;
            T[4 * r + c] = v;
            T34[4 * r + c] = v;
        }
    minus_rt_t(T, Ow);
}

/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), :290-403 (loop closing).
 * kp_to_mp in: vpMatched as -1 (NULL), -2 (a map point outside `pts`) or the row of `pts`; out: rows written
 * by this search.  Returns nmatches, -1 on a predicted level outside [0, nlevels). */
int ora_search_by_projection_sim3(const ora_frame_view *kf, const float *Scw, float fx, float fy, float cx, float cy,
                                  float log_scale_factor, const ora_points_view *pts, int th, int32_t *kp_to_mp)
{
    float T[16], Ow[3];
    memset(T, 0, sizeof(T));
    ora_sim3_decompose(Scw, T, Ow);
    int nmatches = 0;
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kf->n > 0 ? kf->n : 1));
    uint8_t *found = (uint8_t *)calloc((size_t)(pts->m > 0 ? pts->m : 1), 1); /* spAlreadyFound */
    for (int j = 0; j < kf->n; j++)
        if (kp_to_mp[j] >= 0 && kp_to_mp[j] < pts->m)
            found[kp_to_mp[j]] = 1;
    for (int i = 0; i < pts->m; i++) {
        if ((pts->bad && pts->bad[i]) || found[i])
            continue;
        const float *Pw = pts->world_pos + 3 * (size_t)i;
        float pc[3];
        rt_apply(T, Pw, pc);
        if (pc[2] < 0.0f)
            continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y)) /* KeyFrame::IsInImage */
            continue;
        const float maxDistance = 1.2f * pts->max_dist[i], minDistance = 0.8f * pts->min_dist[i];
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance)
            continue;
        const float *Pn = pts->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
        if (dot < 0.5 * dist)
            continue;
        const float ratio = pts->max_dist[i] / dist;
        const int lvl = (int)ceilf(logf(ratio) / log_scale_factor);
        if (lvl < 0 || lvl >= kf->nlevels) {
            nmatches = -1;
            goto done;
        }
        const float radius = th * kf->scale_factors[lvl];
        const int nc = ora_get_features_in_area(kf, u, v, radius, -1, -1, vIndices);
        if (nc == 0)
            continue;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = vIndices[c];
            if (kp_to_mp[idx] != -1)
                continue; /* vpMatched[idx] */
            const int kpLevel = kf->kp_octave[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl)
                continue;
            const int dd = ora_descriptor_distance(pts->desc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
            if (dd < bestDist) {
                bestDist = dd;
                bestIdx = idx;
            }
        }
        if (bestDist <= ORA_TH_LOW) {
            kp_to_mp[bestIdx] = i;
            nmatches++;
        }
    }
done:
    free(vIndices);
    free(found);
    return nmatches;
}

/* MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:242-307: index of the descriptor with the least median
 * distance to the set (float Distances[N][N], rows copied to int, std::sort, [0.5*(N-1)], first minimum). */
static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }
int ora_distinctive_descriptor(int n, const uint8_t *desc)
{
    if (n <= 0)
        return -1;
    int *row = (int *)malloc(sizeof(int) * (size_t)n);
    int best_median = INT32_MAX, best_idx = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++)
            row[j] = i == j ? 0 : ora_descriptor_distance(desc + (size_t)i * 32, desc + (size_t)j * 32);
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        const int median = row[(size_t)(0.5 * (n - 1))];
        if (median < best_median) {
            best_median = median;
            best_idx = i;
        }
    }
    free(row);
    return best_idx;
}

/* ------------------------------------------------------------------ M8 */
/* Frame::isInFrustum, Frame.cc:269-325; MapPoint::PredictScale, MapPoint.cc:385-394 */
int ora_is_in_frustum(const float *Tcw, float fx, float fy, float cx, float cy, float mbf, float min_x,
                      float max_x, float min_y, float max_y, const float *P, const float *normal, float min_dist,
                      float max_dist, float log_scale_factor, int nlevels_unused, float cos_limit, float *proj_x,
                      float *proj_y, float *proj_xr, int32_t *level, float *view_cos)
{
    (void)nlevels_unused;
    float Pc[3];
    rt_apply(Tcw, P, Pc);
    if (Pc[2] < 0.0f)
        return 0;
    const float invz = 1.0f / Pc[2];
    const float u = fx * Pc[0] * invz + cx;
    const float v = fy * Pc[1] * invz + cy;
    if (u < min_x || u > max_x)
        return 0;
    if (v < min_y || v > max_y)
        return 0;
    /* GetMax/MinDistanceInvariance (MapPoint.cc:373-383) */
    const float maxDistance = 1.2f * max_dist;
    const float minDistance = 0.8f * min_dist;
    float Ow[3];
    minus_rt_t(Tcw, Ow);
    float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    /* cv::norm (double accumulation) */
    const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance)
        return 0;
    /* Mat::dot (double accumulation) */
    double dot = (double)PO[0] * normal[0] + (double)PO[1] * normal[1] + (double)PO[2] * normal[2];
    const float viewCos = (float)(dot / (double)dist);
    if (viewCos < cos_limit)
        return 0;
    float ratio = max_dist / dist;
    const int nPredictedLevel = (int)ceilf(logf(ratio) / log_scale_factor);
    *proj_x = u;
    *proj_xr = u - mbf * invz;
    *proj_y = v;
    *level = nPredictedLevel;
    *view_cos = viewCos;
    return 1;
}

/* Tracking::SearchLocalPoints, Tracking.cc:1447-1497, for a fresh frame: every listed point that is not skipped (bad,
 * or already seen by the frame: the caller's `skip`) goes through Frame::isInFrustum(pMP, 0.5) (:1473), then
 * ORBmatcher(0.8).SearchByProjection(F, vpMapPoints, th) (:1488-1496) runs over the ones in view.  A predicted level
 * outside [0, nlevels) is counted in *n_level_out and the point left out (the reference reads past mvScaleFactors, H5).
 * Scratch arrays (m entries each) are the MapPoint members isInFrustum fills (MapPoint.h:91-96). */
int ora_search_local_points(const ora_frame_view *f, const float *Tcw, float fx, float fy, float cx, float cy, float mbf,
                            int m, const float *world_pos, const float *normal, const float *min_dist,
                            const float *max_dist, const uint8_t *skip, const uint8_t *obs_pos, const uint8_t *desc,
                            float log_scale_factor, float cos_limit, float th, float nnratio, uint8_t *in_view,
                            float *proj_x, float *proj_y, float *proj_xr, int32_t *level, float *view_cos,
                            int32_t *kp_to_mp, int *n_level_out)
{
    int nout = 0;
    for (int i = 0; i < m; i++) {
        in_view[i] = 0;
        proj_x[i] = proj_y[i] = proj_xr[i] = view_cos[i] = 0.f;
        level[i] = 0;
        if (skip && skip[i])
            continue;
        float px, py, pxr, vc;
        int32_t lv;
        if (!ora_is_in_frustum(Tcw, fx, fy, cx, cy, mbf, f->min_x, f->max_x, f->min_y, f->max_y, world_pos + 3 * i,
                               normal + 3 * i, min_dist[i], max_dist[i], log_scale_factor, f->nlevels, cos_limit, &px, &py,
                               &pxr, &lv, &vc))
            continue;
        if (lv < 0 || lv >= f->nlevels) {
            nout++;
            continue;
        }
        in_view[i] = 1, proj_x[i] = px, proj_y[i] = py, proj_xr[i] = pxr, level[i] = lv, view_cos[i] = vc;
    }
    if (n_level_out)
        *n_level_out = nout;
    ora_mappoint_view mp;
    mp.m = m, mp.in_view = in_view, mp.bad = skip, mp.obs_pos = obs_pos, mp.level = level, mp.view_cos = view_cos;
    mp.proj_x = proj_x, mp.proj_y = proj_y, mp.proj_xr = proj_xr, mp.desc = desc;
    return ora_search_by_projection(f, &mp, th, nnratio, kp_to_mp);
}

/* cv::undistortPoints(src, dst, K, distCoef, Mat(), K) as Frame::UndistortKeyPoints / ComputeImageBounds call it
 * (Frame.cc:404-468), OpenCV 2.4 cvUndistortPoints: everything in double, five fixed-point iterations, the
 * rational terms k4..k6 zero, R = identity, P = K.  dist = k1 k2 p1 p2 k3 (k3 = 0 for a 4-entry mDistCoef).
 * Restated from knowledge of the 2.4 source: unpinned like the other OpenCV pieces. */
void ora_undistort_points(int n, const float *xy_in, float fx_, float fy_, float cx_, float cy_, const float *dist,
                          float *xy_out)
{
    const double fx = fx_, fy = fy_, cx = cx_, cy = cy_;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k[8] = {dist[0], dist[1], dist[2], dist[3], dist[4], 0, 0, 0};
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        const double x0 = x = (x - cx) * ifx;
        const double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* RR = P * I = K: xx = fx*x + 0*y + cx, ww = 1/(0*x + 0*y + 1) */
        const double xx = fx * x + 0. * y + cx;
        const double yy = 0. * x + fy * y + cy;
        const double ww = 1. / (0. * x + 0. * y + 1.);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::ComputeStereoFromRGBD, Frame.cc:641-662 */
void ora_compute_stereo_from_rgbd(int n, const float *kp_x, const float *kp_y, const float *kpun_x,
                                  const float *depth, size_t depth_stride_elems, float mbf, float *u_right,
                                  float *kp_depth)
{
    for (int i = 0; i < n; i++) {
        u_right[i] = -1;
        kp_depth[i] = -1;
        /* imDepth.at<float>(v,u): float -> int conversion truncates */
        const int vi = (int)kp_y[i], ui = (int)kp_x[i];
        const float d = depth[(size_t)vi * depth_stride_elems + (size_t)ui];
        if (d > 0) {
            kp_depth[i] = d;
            u_right[i] = kpun_x[i] - mbf / d;
        }
    }
}

/* =====================================================================================
 * M6: the background-thread matchers (LocalMapping / LoopClosing).  Same conventions as above.
 * ===================================================================================== */

/* best key point of a window without claims: the inner loop shared by ORBmatcher::Fuse (:895-944), Fuse(Sim3)
 * (:1056-1078) and SearchBySim3 (:1193-1221, :1273-1301).  gate: Fuse's reprojection-error tests (:908-933). */
static int best_in_window(const ora_frame_view *kf, float u, float v, float ur, float radius, int lvl, int gate,
                          const float *inv_level_sigma2, const uint8_t *dMP, int32_t *vIndices, int *bestDistOut)
{
    const int nc = ora_get_features_in_area(kf, u, v, radius, -1, -1, vIndices); /* KeyFrame.cc:607-646 */
    int bestDist = 256, bestIdx = -1; /* (Fuse(Sim3) / SearchBySim3 start at INT_MAX: same result, a distance is <= 256) */
    for (int c = 0; c < nc; c++) {
        const int idx = vIndices[c];
        const int kpLevel = kf->kp_octave[idx];
        if (kpLevel < lvl - 1 || kpLevel > lvl)
            continue;
        if (gate) {
            const float kpx = kf->kp_x[idx], kpy = kf->kp_y[idx];
            const float ex = u - kpx, ey = v - kpy;
            if (kf->u_right[idx] >= 0) {
                const float er = ur - kf->u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8)
                    continue;
            } else {
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99)
                    continue;
            }
        }
        const int dist = ora_descriptor_distance(dMP, kf->desc + (size_t)idx * 32);
        if (dist < bestDist) {
            bestDist = dist;
            bestIdx = idx;
        }
    }
    *bestDistOut = bestDist;
    return bestIdx;
}

/* ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th), ORBmatcher.cc:825-975: the candidate
 * phase.  pts->bad[i] = !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF) at call time.  best_idx[i] = key point the
 * point fuses with (bestDist <= TH_LOW) or -1; the map edits (:948-968) are the caller's.  Returns the number of
 * candidates, -1 on a predicted level outside [0, nlevels) (the reference reads mvScaleFactors out of range). */
int ora_fuse(const ora_frame_view *kf, const float *Tcw, float fx, float fy, float cx, float cy, float bf,
             float log_scale_factor, const ora_points_view *pts, float th, const float *inv_level_sigma2,
             int32_t *best_idx)
{
    float Ow[3];
    minus_rt_t(Tcw, Ow); /* GetCameraCenter */
    int n = 0;
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kf->n > 0 ? kf->n : 1));
    for (int i = 0; i < pts->m; i++) {
        best_idx[i] = -1;
        if (pts->bad && pts->bad[i])
            continue;
        const float *Pw = pts->world_pos + 3 * (size_t)i;
        float pc[3];
        rt_apply(Tcw, Pw, pc);
        if (pc[2] < 0.0f)
            continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y))
            continue;
        const float ur = u - bf * invz;
        const float maxDistance = 1.2f * pts->max_dist[i], minDistance = 0.8f * pts->min_dist[i];
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist3D < minDistance || dist3D > maxDistance)
            continue;
        const float *Pn = pts->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
        if (dot < 0.5 * dist3D)
            continue;
        const float ratio = pts->max_dist[i] / dist3D;
        const int lvl = (int)ceilf(logf(ratio) / log_scale_factor);
        if (lvl < 0 || lvl >= kf->nlevels) {
            n = -1;
            break;
        }
        const float radius = th * kf->scale_factors[lvl];
        int bestDist;
        const int bestIdx = best_in_window(kf, u, v, ur, radius, lvl, 1, inv_level_sigma2,
                                           pts->desc + (size_t)i * 32, vIndices, &bestDist);
        if (bestIdx >= 0 && bestDist <= ORA_TH_LOW) {
            best_idx[i] = bestIdx;
            n++;
        }
    }
    free(vIndices);
    return n;
}

/* ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint), ORBmatcher.cc:977-1100: candidate
 * phase.  pts->bad[i] = isBad() || spAlreadyFound.count(pMP). */
int ora_fuse_sim3(const ora_frame_view *kf, const float *Scw, float fx, float fy, float cx, float cy,
                  float log_scale_factor, const ora_points_view *pts, float th, int32_t *best_idx)
{
    float T[16], Ow[3];
    memset(T, 0, sizeof(T));
    ora_sim3_decompose(Scw, T, Ow);
    int n = 0;
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kf->n > 0 ? kf->n : 1));
    for (int i = 0; i < pts->m; i++) {
        best_idx[i] = -1;
        if (pts->bad && pts->bad[i])
            continue;
        const float *Pw = pts->world_pos + 3 * (size_t)i;
        float pc[3];
        rt_apply(T, Pw, pc);
        if (pc[2] < 0.0f)
            continue;
        const float invz = (float)(1.0 / (double)pc[2]); /* :1017 `1.0/` */
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y))
            continue;
        const float maxDistance = 1.2f * pts->max_dist[i], minDistance = 0.8f * pts->min_dist[i];
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist3D < minDistance || dist3D > maxDistance)
            continue;
        const float *Pn = pts->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
        if (dot < 0.5 * dist3D)
            continue;
        const float ratio = pts->max_dist[i] / dist3D;
        const int lvl = (int)ceilf(logf(ratio) / log_scale_factor);
        if (lvl < 0 || lvl >= kf->nlevels) {
            n = -1;
            break;
        }
        const float radius = th * kf->scale_factors[lvl];
        int bestDist;
        const int bestIdx = best_in_window(kf, u, v, 0.f, radius, lvl, 0, NULL, pts->desc + (size_t)i * 32, vIndices,
                                           &bestDist);
        if (bestIdx >= 0 && bestDist <= ORA_TH_LOW) {
            best_idx[i] = bestIdx;
            n++;
        }
    }
    free(vIndices);
    return n;
}

/* one direction of SearchBySim3 (:1143-1227 / :1229-1307): points of key frame A (camera frame Taw) moved into key
 * frame B by the similarity (sR, t), best key point of B in the window, threshold TH_HIGH */
static int sim3_direction(const ora_frame_view *kfB, const float *Taw, const float sR[9], const float t[3], float fx,
                          float fy, float cx, float cy, float log_sfB, const ora_points_view *ptsA,
                          const uint8_t *skipA, float th, int32_t *match)
{
    int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kfB->n > 0 ? kfB->n : 1));
    int rc = 0;
    for (int i = 0; i < ptsA->m; i++) {
        match[i] = -1;
        if ((ptsA->bad && ptsA->bad[i]) || (skipA && skipA[i]))
            continue;
        float pa[3], pb[3];
        rt_apply(Taw, ptsA->world_pos + 3 * (size_t)i, pa);
        for (int r = 0; r < 3; r++) {
            const float t0 = sR[3 * r] * pa[0] + sR[3 * r + 1] * pa[1];
            const float t1 = t0 + sR[3 * r + 2] * pa[2];
            pb[r] = t1 + t[r];
        }
        if (pb[2] < 0.0)
            continue;
        const float invz = (float)(1.0 / (double)pb[2]);
        const float x = pb[0] * invz, y = pb[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= kfB->min_x && u < kfB->max_x && v >= kfB->min_y && v < kfB->max_y))
            continue;
        const float maxDistance = 1.2f * ptsA->max_dist[i], minDistance = 0.8f * ptsA->min_dist[i];
        const float dist3D = (float)sqrt((double)pb[0] * pb[0] + (double)pb[1] * pb[1] + (double)pb[2] * pb[2]);
        if (dist3D < minDistance || dist3D > maxDistance)
            continue;
        const float ratio = ptsA->max_dist[i] / dist3D;
        const int lvl = (int)ceilf(logf(ratio) / log_sfB);
        if (lvl < 0 || lvl >= kfB->nlevels) {
            rc = -1;
            break;
        }
        const float radius = th * kfB->scale_factors[lvl];
        int bestDist;
        const int bestIdx = best_in_window(kfB, u, v, 0.f, radius, lvl, 0, NULL, ptsA->desc + (size_t)i * 32, vIndices,
                                           &bestDist);
        if (bestIdx >= 0 && bestDist <= ORA_TH_HIGH)
            match[i] = bestIdx;
    }
    free(vIndices);
    return rc;
}

/* ORBmatcher::SearchBySim3, ORBmatcher.cc:1102-1326.  pts1 / pts2: row i = key point i of the key frame (bad[i] =
 * no map point or isBad()); already1 / already2 = vbAlreadyMatched1 / 2 (:1133-1144).  R12 row-major 3x3, t12.
 * match12[i1] = index in key frame 2 where both directions agree, else -1.  Returns nFound, -1 on a level error. */
int ora_search_by_sim3(const ora_frame_view *kf1, const ora_frame_view *kf2, const float *T1w, const float *T2w,
                       float s12, const float *R12, const float *t12, float fx, float fy, float cx, float cy,
                       float log_sf1, float log_sf2, const ora_points_view *pts1, const uint8_t *already1,
                       const ora_points_view *pts2, const uint8_t *already2, float th, int32_t *match12)
{
    float sR12[9], sR21[9], t21[3];
    const double inv_s = 1.0 / (double)s12;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            sR12[3 * r + c] = s12 * R12[3 * r + c];                           /* :1121 */
            sR21[3 * r + c] = (float)((double)R12[3 * c + r] * inv_s);        /* :1122 (1.0/s12)*R12.t() */
        }
    for (int r = 0; r < 3; r++) { /* :1123 t21 = -sR21*t12 (small-matrix product, alpha = -1) */
        const float t0 = sR21[3 * r] * t12[0] + sR21[3 * r + 1] * t12[1];
        const float t1 = t0 + sR21[3 * r + 2] * t12[2];
        t21[r] = -t1;
    }
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pts1->m > 0 ? pts1->m : 1));
    int32_t *m2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pts2->m > 0 ? pts2->m : 1));
    int rc = sim3_direction(kf2, T1w, sR21, t21, fx, fy, cx, cy, log_sf2, pts1, already1, th, m1);
    if (rc == 0)
        rc = sim3_direction(kf1, T2w, sR12, t12, fx, fy, cx, cy, log_sf1, pts2, already2, th, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < pts1->m; i1++) {
        match12[i1] = -1;
        if (rc != 0)
            continue;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { /* :1312-1322 */
            match12[i1] = idx2;
            nFound++;
        }
    }
    free(m1);
    free(m2);
    return rc != 0 ? -1 : nFound;
}

/* ORBmatcher::CheckDistEpipolarLine, ORBmatcher.cc:140-157 (F12 row-major 3x3) */
static int check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float *F12, float sigma2_kp2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0)
        return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2_kp2;
}

/* ORBmatcher::SearchForTriangulation, ORBmatcher.cc:657-823.  Key frames as frame views (kp_x/kp_y = mvKeysUn,
 * u_right = mvuRight), has_mp* = GetMapPoint(idx) != NULL, feature vectors as CSR.  (ex, ey) = the epipole in the
 * second image (:664-670, computed by the caller from the poses), level_sigma2 / scale factors of key frame 2.
 * vbMatched2 is never set in the reference (:676, 724): rows are independent.  match12[i1] = idx2 or -1. */
int ora_search_for_triangulation(const ora_frame_view *kf1, const uint8_t *has_mp1, int n_fv1, const int32_t *fv_nodes1,
                                 const int32_t *fv_start1, const int32_t *fv_items1, const ora_frame_view *kf2,
                                 const uint8_t *has_mp2, int n_fv2, const int32_t *fv_nodes2,
                                 const int32_t *fv_start2, const int32_t *fv_items2, const float *F12, float ex,
                                 float ey, const float *level_sigma2_2, int only_stereo, int check_orientation,
                                 int32_t *match12)
{
    int nmatches = 0;
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    int *bin_of = (int *)malloc(sizeof(int) * (size_t)(kf1->n > 0 ? kf1->n : 1));
    for (int i = 0; i < kf1->n; i++) {
        match12[i] = -1;
        bin_of[i] = -1;
    }
    int a = 0, b = 0;
    while (a < n_fv1 && b < n_fv2) {
        if (fv_nodes1[a] == fv_nodes2[b]) {
            for (int p1 = fv_start1[a]; p1 < fv_start1[a + 1]; p1++) {
                const int idx1 = fv_items1[p1];
                if (has_mp1[idx1])
                    continue;
                const int bStereo1 = kf1->u_right[idx1] >= 0;
                if (only_stereo && !bStereo1)
                    continue;
                int bestDist = ORA_TH_LOW, bestIdx2 = -1;
                for (int p2 = fv_start2[b]; p2 < fv_start2[b + 1]; p2++) {
                    const int idx2 = fv_items2[p2];
                    if (has_mp2[idx2])
                        continue;
                    const int bStereo2 = kf2->u_right[idx2] >= 0;
                    if (only_stereo && !bStereo2)
                        continue;
                    const int dist = ora_descriptor_distance(kf1->desc + (size_t)idx1 * 32, kf2->desc + (size_t)idx2 * 32);
                    if (dist > ORA_TH_LOW || dist > bestDist)
                        continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kf2->kp_x[idx2], distey = ey - kf2->kp_y[idx2];
                        if (distex * distex + distey * distey < 100 * kf2->scale_factors[kf2->kp_octave[idx2]])
                            continue;
                    }
                    if (check_dist_epipolar_line(kf1->kp_x[idx1], kf1->kp_y[idx1], kf2->kp_x[idx2], kf2->kp_y[idx2], F12,
                                                 level_sigma2_2[kf2->kp_octave[idx2]])) {
                        bestIdx2 = idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        const int bin = rot_bin(kf1->kp_angle[idx1], kf2->kp_angle[bestIdx2]);
                        bin_of[idx1] = bin;
                        histo[bin]++;
                    }
                }
            }
            a++;
            b++;
        } else if (fv_nodes1[a] < fv_nodes2[b]) {
            while (a < n_fv1 && fv_nodes1[a] < fv_nodes2[b])
                a++;
        } else {
            while (b < n_fv2 && fv_nodes2[b] < fv_nodes1[a])
                b++;
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(histo, ORA_HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < kf1->n; i++) {
            const int bn = bin_of[i];
            if (bn < 0 || bn == i1 || bn == i2 || bn == i3)
                continue;
            match12[i] = -1;
            nmatches--;
        }
    }
    free(bin_of);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12), ORBmatcher.cc:522-655.  valid1 / valid2:
 * the key point has a map point that is not bad.  match12[idx1] = idx2 (the index whose map point vpMatches12 gets)
 * or -1.  Note the strict `bestDist1 < TH_LOW` (:592), unlike the frame version. */
int ora_search_by_bow_kf(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1, int n_fv1,
                         const int32_t *fv_nodes1, const int32_t *fv_start1, const int32_t *fv_items1,
                         const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2, int n_fv2,
                         const int32_t *fv_nodes2, const int32_t *fv_start2, const int32_t *fv_items2, float nnratio,
                         int check_orientation, int32_t *match12)
{
    int nmatches = 0;
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    int *bin_of = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
    uint8_t *matched2 = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1);
    for (int i = 0; i < n1; i++) {
        match12[i] = -1;
        bin_of[i] = -1;
    }
    int a = 0, b = 0;
    while (a < n_fv1 && b < n_fv2) {
        if (fv_nodes1[a] == fv_nodes2[b]) {
            for (int p1 = fv_start1[a]; p1 < fv_start1[a + 1]; p1++) {
                const int idx1 = fv_items1[p1];
                if (valid1 && !valid1[idx1])
                    continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int p2 = fv_start2[b]; p2 < fv_start2[b + 1]; p2++) {
                    const int idx2 = fv_items2[p2];
                    if (matched2[idx2] || (valid2 && !valid2[idx2]))
                        continue;
                    const int dist = ora_descriptor_distance(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < ORA_TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (check_orientation) {
                            const int bin = rot_bin(angle1[idx1], angle2[bestIdx2]);
                            bin_of[idx1] = bin;
                            histo[bin]++;
                        }
                        nmatches++;
                    }
                }
            }
            a++;
            b++;
        } else if (fv_nodes1[a] < fv_nodes2[b]) {
            while (a < n_fv1 && fv_nodes1[a] < fv_nodes2[b])
                a++;
        } else {
            while (b < n_fv2 && fv_nodes2[b] < fv_nodes1[a])
                b++;
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(histo, ORA_HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < n1; i++) {
            const int bn = bin_of[i];
            if (bn < 0 || bn == i1 || bn == i2 || bn == i3)
                continue;
            match12[i] = -1;
            nmatches--;
        }
    }
    free(bin_of);
    free(matched2);
    return nmatches;
}

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), ORBmatcher.cc:405-520
 * (Tracking::MonocularInitialization, Tracking.cc:877).  f1: kp_octave / kp_angle / desc of the initial frame's
 * mvKeysUn; f2: the current frame (grid, key points, descriptors).  prev_matched [n1][2] in/out (:514-517);
 * matches12 [n1] out.  A later row steals a key point of F2 when its distance is strictly smaller (:441-442,
 * 459-470).  Returns nmatches. */
int ora_search_for_initialization(const ora_frame_view *f1, const ora_frame_view *f2, float *prev_matched,
                                  int window_size, float nnratio, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    const int n1 = f1->n, n2 = f2->n;
    int *vMatchedDistance = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int *vnMatches21 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int32_t *vIndices2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    int *hist_items = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1)); /* rotHist pushes: (bin, i1) in push order */
    int *hist_bin = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
    int npush = 0;
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    for (int i = 0; i < n1; i++)
        matches12[i] = -1;
    for (int j = 0; j < n2; j++) {
        vMatchedDistance[j] = 2147483647;
        vnMatches21[j] = -1;
    }
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = f1->kp_octave[i1];
        if (level1 > 0)
            continue;
        const int nc = ora_get_features_in_area(f2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size,
                                                level1, level1, vIndices2);
        if (nc == 0)
            continue;
        const uint8_t *d1 = f1->desc + (size_t)i1 * 32;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = vIndices2[c];
            const int dist = ora_descriptor_distance(d1, f2->desc + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist)
                continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= ORA_TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    matches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    const int bin = rot_bin(f1->kp_angle[i1], f2->kp_angle[bestIdx2]);
                    hist_items[npush] = i1;
                    hist_bin[npush] = bin;
                    npush++;
                    histo[bin]++;
                }
            }
        }
    }
    if (check_orientation) {
        int i1m, i2m, i3m;
        three_maxima(histo, ORA_HISTO_LENGTH, &i1m, &i2m, &i3m);
        for (int p = 0; p < npush; p++) {
            const int b = hist_bin[p];
            if (b == i1m || b == i2m || b == i3m)
                continue;
            const int idx1 = hist_items[p];
            if (matches12[idx1] >= 0) { /* :497-501: a stolen match is already gone */
                matches12[idx1] = -1;
                nmatches--;
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++) /* :514-517 */
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = f2->kp_x[matches12[i1]];
            prev_matched[2 * i1 + 1] = f2->kp_y[matches12[i1]];
        }
    free(vMatchedDistance);
    free(vnMatches21);
    free(vIndices2);
    free(hist_items);
    free(hist_bin);
    return nmatches;
}
