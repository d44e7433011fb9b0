/*
 * orb_oracle_cloud.c -- CPU ORACLE (test infrastructure, NOT product code; see orb_oracle.h).
 *
 * Restates the RGB-D dense-map arithmetic of ORB_SLAM2::PointCloudMapping (reference
 * src/PointCloudMap.cc) and the PCL 1.7 / g2o / Eigen primitives it calls.
 * Compile with -ffp-contract=off.  Parity unpinned at the PCL boundary.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* P1: convertToPointCloud, PointCloudMap.cc:112-138 (same loop as generatePointCloud :78-100) */
int ora_backproject(const float *depth, size_t depth_stride_elems, const uint8_t *rgb, size_t rgb_stride, int w,
                    int h, float fx, float fy, float cx, float cy, ora_point *out)
{
    int n = 0;
    for (int m = 0; m < h; m += 3) {
        for (int c = 0; c < w; c += 3) {
            float d = depth[(size_t)m * depth_stride_elems + (size_t)c];
            if ((double)d < 0.01 || d > 10) /* :121, 0.01 is a double literal */
                continue;
            ora_point p;
            p.z = d;
            p.x = ((float)c - cx) * p.z / fx;
            p.y = ((float)m - cy) * p.z / fy;
            const uint8_t *px = rgb + (size_t)m * rgb_stride + (size_t)c * 3;
            /* b,g,r <- image bytes 0,1,2 (:128-130); alpha stays 0 (PCL 1.7 PointXYZRGBA ctor) */
            p.rgba = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
            out[n++] = p;
        }
    }
    return n;
}

/* P2: Converter::toSE3Quat (Converter.cc:37-47) -> g2o::SE3Quat(R,t) (Eigen quaternion from
 * matrix + normalizeRotation) -> Eigen::Isometry3d -> .inverse() (PointCloudMap.cc:103-105). */
void ora_pose_inverse(const float *Tcw, double R[9], double t[3])
{
    double m[3][3], tt[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            m[i][j] = (double)Tcw[4 * i + j];
        tt[i] = (double)Tcw[4 * i + 3];
    }
    /* Eigen::Quaternion = Matrix3 (quaternionbase_assign_impl<Other,3,3>) */
    double q[4]; /* x y z w */
    double tr = m[0][0] + m[1][1] + m[2][2];
    if (tr > 0) {
        double s = sqrt(tr + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (m[2][1] - m[1][2]) * s;
        q[1] = (m[0][2] - m[2][0]) * s;
        q[2] = (m[1][0] - m[0][1]) * s;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0])
            i = 1;
        if (m[2][2] > m[i][i])
            i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (m[k][j] - m[j][k]) * s;
        q[j] = (m[j][i] + m[i][j]) * s;
        q[k] = (m[k][i] + m[i][k]) * s;
    }
    /* SE3Quat::normalizeRotation */
    if (q[3] < 0)
        for (int i = 0; i < 4; i++)
            q[i] = -q[i];
    double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++)
        q[i] /= nrm;
    /* Quaternion::toRotationMatrix */
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    double Rq[3][3];
    Rq[0][0] = 1 - (tyy + tzz);
    Rq[0][1] = txy - twz;
    Rq[0][2] = txz + twy;
    Rq[1][0] = txy + twz;
    Rq[1][1] = 1 - (txx + tzz);
    Rq[1][2] = tyz - twx;
    Rq[2][0] = txz - twy;
    Rq[2][1] = tyz + twx;
    Rq[2][2] = 1 - (txx + tyy);
    /* Isometry inverse: R^T, -R^T t */
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = Rq[j][i];
    }
    for (int i = 0; i < 3; i++)
        t[i] = -(R[3 * i + 0] * tt[0] + R[3 * i + 1] * tt[1] + R[3 * i + 2] * tt[2]);
}

/* P2: pcl::transformPointCloud(in, out, Matrix4d) (A8): double arithmetic, stored as float.
 * Non-finite points are copied unchanged (is_dense == false path). */
void ora_transform_points(const ora_point *in, int n, const double R[9], const double t[3], ora_point *out)
{
    for (int i = 0; i < n; i++) {
        ora_point p = in[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            double x = p.x, y = p.y, z = p.z;
            ora_point o = p;
            o.x = (float)(R[0] * x + R[1] * y + R[2] * z + t[0]);
            o.y = (float)(R[3] * x + R[4] * y + R[5] * z + t[1]);
            o.z = (float)(R[6] * x + R[7] * y + R[8] * z + t[2]);
            out[i] = o;
        } else {
            out[i] = p;
        }
    }
}

/* P3: pcl::VoxelGrid<PointXYZRGBA>::applyFilter (PCL 1.7, A7), downsample_all_data = true,
 * min_points_per_voxel = 0, no filter field.  Intra-voxel summation order: the reference's
 * std::sort on the voxel index alone leaves it unspecified; the oracle uses input order. */
typedef struct {
    uint32_t idx;
    int pt;
} vox_ref;

static int vox_cmp(const void *a, const void *b)
{
    const vox_ref *A = (const vox_ref *)a, *B = (const vox_ref *)b;
    if (A->idx != B->idx)
        return A->idx < B->idx ? -1 : 1;
    return A->pt < B->pt ? -1 : (A->pt > B->pt ? 1 : 0);
}

int ora_voxel_filter(const ora_point *in, int n, float leaf, ora_point *out, int *overflow)
{
    if (overflow)
        *overflow = 0;
    const float inv = 1.0f / leaf;
    /* getMinMax3D over finite points */
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    int nfinite = 0;
    for (int i = 0; i < n; i++) {
        const ora_point *p = &in[i];
        if (!isfinite(p->x) || !isfinite(p->y) || !isfinite(p->z))
            continue;
        nfinite++;
        if (p->x < mn[0]) mn[0] = p->x;
        if (p->y < mn[1]) mn[1] = p->y;
        if (p->z < mn[2]) mn[2] = p->z;
        if (p->x > mx[0]) mx[0] = p->x;
        if (p->y > mx[1]) mx[1] = p->y;
        if (p->z > mx[2]) mx[2] = p->z;
    }
    if (nfinite == 0)
        return 0;
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
    int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
    int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)2147483647) {
        if (overflow)
            *overflow = 1;
        memcpy(out, in, sizeof(ora_point) * (size_t)n);
        return n;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; a++) {
        min_b[a] = (int)floorf(mn[a] * inv);
        max_b[a] = (int)floorf(mx[a] * inv);
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};

    vox_ref *iv = (vox_ref *)malloc(sizeof(vox_ref) * (size_t)nfinite);
    int m = 0;
    for (int i = 0; i < n; i++) {
        const ora_point *p = &in[i];
        if (!isfinite(p->x) || !isfinite(p->y) || !isfinite(p->z))
            continue;
        int ijk0 = (int)(floorf(p->x * inv) - (float)min_b[0]);
        int ijk1 = (int)(floorf(p->y * inv) - (float)min_b[1]);
        int ijk2 = (int)(floorf(p->z * inv) - (float)min_b[2]);
        iv[m].idx = (uint32_t)(ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2]);
        iv[m].pt = i;
        m++;
    }
    qsort(iv, (size_t)m, sizeof(vox_ref), vox_cmp);

    int nout = 0;
    int i = 0;
    while (i < m) {
        int j = i;
        float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
        while (j < m && iv[j].idx == iv[i].idx) {
            const ora_point *p = &in[iv[j].pt];
            float r = (float)((p->rgba >> 16) & 255), g = (float)((p->rgba >> 8) & 255), b = (float)(p->rgba & 255);
            if (j == i) {
                sx = p->x; sy = p->y; sz = p->z; sr = r; sg = g; sb = b;
            } else {
                sx += p->x; sy += p->y; sz += p->z; sr += r; sg += g; sb += b;
            }
            j++;
        }
        const float cnt = (float)(j - i);
        ora_point o;
        o.x = sx / cnt;
        o.y = sy / cnt;
        o.z = sz / cnt;
        int ri = (int)(sr / cnt), gi = (int)(sg / cnt), bi = (int)(sb / cnt);
        o.rgba = (uint32_t)((ri << 16) | (gi << 8) | bi);
        out[nout++] = o;
        i = j;
    }
    free(iv);
    return nout;
}

/* ---- pcl::StatisticalOutlierRemoval<PointXYZRGBA>::filter as PointCloudMapping::viewer uses it at shutdown
 * (PointCloudMap.cc:46-47: setMeanK(50), setStddevMulThresh(1.0); :283-285: sor.filter of the concatenated key-frame
 * clouds).  Restated from PCL 1.7 filters/impl/statistical_outlier_removal.hpp (applyFilterIndices) -- PCL is not
 * vendored in the reference: third-party boundary, unpinned:
 *   - per finite point: nearestKSearch(point, mean_k + 1) on a kd-tree of the finite points, squared distances in
 *     float (FLANN L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz), ascending; entry 0 is the query point itself;
 *     dist_sum (double) += sqrt(nn_dists[k]) for k = 1 .. mean_k -- ::sqrt(double) of the promoted float, the
 *     overload a gnu++98 libstdc++ of PCL 1.7's time resolves the unqualified call to;
 *     distances[i] = (float)(dist_sum / mean_k);  non-finite points get 0 and do not count;
 *   - sum += distances[i], sq_sum += distances[i] * distances[i] (a float product) over all points, in index order;
 *     mean = sum / valid, variance = (sq_sum - sum * sum / valid) / (valid - 1), threshold = mean + mul * sqrt(variance);
 *   - a point is removed iff distances[i] > threshold; the others keep their order.
 * The exact kd-tree search is replaced by brute force (same multiset of the mean_k + 1 smallest distances).
 * Needs n_finite > mean_k (PCL reads past the neighbour list otherwise).  mean_dist (n floats) may be NULL.
 * Returns the number of points kept, -1 on bad arguments. */
static int sor_cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}
int ora_statistical_outlier_removal(const ora_point *in, int n, int mean_k, double stddev_mul, ora_point *out,
                                    float *mean_dist)
{
    if (n <= 0 || mean_k < 1)
        return -1;
    int nfinite = 0;
    for (int i = 0; i < n; i++)
        nfinite += isfinite(in[i].x) && isfinite(in[i].y) && isfinite(in[i].z);
    if (nfinite <= mean_k)
        return -1;
    float *dist = (float *)malloc(sizeof(float) * (size_t)n);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)nfinite);
    int valid = 0;
    for (int i = 0; i < n; i++) {
        const ora_point *p = &in[i];
        if (!(isfinite(p->x) && isfinite(p->y) && isfinite(p->z))) {
            dist[i] = 0.0f;
            continue;
        }
        int m = 0;
        for (int j = 0; j < n; j++) {
            const ora_point *q = &in[j];
            if (!(isfinite(q->x) && isfinite(q->y) && isfinite(q->z)))
                continue;
            const float dx = p->x - q->x, dy = p->y - q->y, dz = p->z - q->z;
            float r = 0.0f;
            r += dx * dx;
            r += dy * dy;
            r += dz * dz;
            d2[m++] = r;
        }
        qsort(d2, (size_t)m, sizeof(float), sor_cmp_float);
        double dist_sum = 0.0;
        for (int k = 1; k < mean_k + 1; k++)
            dist_sum += sqrt((double)d2[k]);
        dist[i] = (float)(dist_sum / mean_k);
        valid++;
    }
    double sum = 0, sq_sum = 0;
    for (int i = 0; i < n; i++) {
        sum += dist[i];
        sq_sum += dist[i] * dist[i];
    }
    const double mean = sum / (double)valid;
    const double variance = (sq_sum - sum * sum / (double)valid) / ((double)valid - 1);
    const double stddev = sqrt(variance);
    const double distance_threshold = mean + stddev_mul * stddev;
    int kept = 0;
    for (int i = 0; i < n; i++) {
        if (mean_dist)
            mean_dist[i] = dist[i];
        if (dist[i] > distance_threshold)
            continue;
        out[kept++] = in[i];
    }
    free(dist);
    free(d2);
    return kept;
}
