// On-disk formats of the reference's end-of-run artefacts, byte for byte:
//   Map::_WriteMapPoint / Map::_WriteKeyFrame   reference src/Map.cc:123-183   (records of Map::Save, :188-249)
//   Converter::toQuaternion                     reference src/Converter.cc:137-149
//   pcl::io::savePCDFileBinary(PointCloud<PointXYZRGBA>)   reference src/PointCloudMap.cc:287  (PCL 1.7 PCDWriter)
// Host code: these are serialisers of a few KB to a few MB per call; the key points / descriptors / map they take
// are what the device entry points produced.  The PCD header text is PCL 1.7's as recalled (SURVEY.md Appendix A:
// not verifiable in this image); the body is the map's 16-byte points, which is already PCL's packed field order.
#include "common.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace orbgpu {

// Eigen::Quaterniond(const Matrix3d&): trace-based conversion, no normalisation, no sign convention
static void eigen_quaternion(const double m[3][3], double q[4] /* x y z w */)
{
    const double tr = m[0][0] + m[1][1] + m[2][2];
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
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (m[k][j] - m[j][k]) * s;
        q[j] = (m[j][i] + m[i][j]) * s;
        q[k] = (m[k][i] + m[i][k]) * s;
    }
}

template <typename T> static inline void put(uint8_t *&p, const T &v)
{
    memcpy(p, &v, sizeof(T));
    p += sizeof(T);
}

} // namespace orbgpu

using namespace orbgpu;

extern "C" {

size_t orbgpu_mappoint_record_bytes(void) { return 8 + 3 * 4; }

size_t orbgpu_keyframe_record_bytes(int32_t n_features)
{
    return 8 + 8 + 3 * 4 + 4 * 4 + 4 + (size_t)(n_features > 0 ? n_features : 0) * (6 * 4 + 32 + 8);
}

int orbgpu_write_mappoint_record(uint64_t id, const float *world_pos, uint8_t *out, size_t cap, size_t *written)
{
    ORBGPU_REQUIRE(world_pos && out && cap >= orbgpu_mappoint_record_bytes(), "bad arguments");
    uint8_t *p = out;
    put(p, id);  // long unsigned int mnId (:125)
    put(p, world_pos[0]);
    put(p, world_pos[1]);
    put(p, world_pos[2]);
    if (written)
        *written = (size_t)(p - out);
    return ORBGPU_OK;
}

int orbgpu_write_keyframe_record(uint64_t id, double timestamp, const float *Tcw, int32_t n,
                                 const orbgpu_keypoint *keys, const uint8_t *desc, const uint64_t *mappoint_index,
                                 uint8_t *out, size_t cap, size_t *written)
{
    ORBGPU_REQUIRE(Tcw && out && n >= 0 && (n == 0 || (keys && desc && mappoint_index)), "bad arguments");
    const size_t need = orbgpu_keyframe_record_bytes(n);
    if (cap < need) {
        set_error("key-frame record needs %zu bytes, cap is %zu", need, cap);
        return ORBGPU_ECAPACITY;
    }
    uint8_t *p = out;
    put(p, id);         // :135 long unsigned int
    put(p, timestamp);  // :136 double
    put(p, Tcw[3]);     // :154-156 translation column of the pose
    put(p, Tcw[7]);
    put(p, Tcw[11]);
    double m[3][3], q[4];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            m[r][c] = (double)Tcw[4 * r + c];
    eigen_quaternion(m, q);
    for (int k = 0; k < 4; k++)  // :157-161 x y z w as floats
        put(p, (float)q[k]);
    put(p, n);  // :163 int
    for (int i = 0; i < n; i++) {
        const orbgpu_keypoint &kp = keys[i];
        put(p, kp.x);
        put(p, kp.y);
        put(p, kp.size);
        put(p, kp.angle);
        put(p, kp.response);
        put(p, kp.octave);
        memcpy(p, desc + (size_t)i * 32, 32);  // :175-176
        p += 32;
        put(p, mappoint_index[i]);  // :178-182 ULONG_MAX when the key point has no map point
    }
    if (written)
        *written = (size_t)(p - out);
    return ORBGPU_OK;
}

int orbgpu_pcd_binary_header(int64_t n_points, char *buf, size_t cap, size_t *len)
{
    ORBGPU_REQUIRE(buf && n_points >= 0, "bad arguments");
    // pcl::PCDWriter::generateHeader for PointXYZRGBA (padding fields are skipped; width = n, height = 1 after
    // PointCloud::operator+= / VoxelGrid; sensor origin 0, orientation identity)
    const int w = snprintf(buf, cap,
                           "# .PCD v0.7 - Point Cloud Data file format\n"
                           "VERSION 0.7\n"
                           "FIELDS x y z rgba\n"
                           "SIZE 4 4 4 4\n"
                           "TYPE F F F U\n"
                           "COUNT 1 1 1 1\n"
                           "WIDTH %lld\n"
                           "HEIGHT 1\n"
                           "VIEWPOINT 0 0 0 1 0 0 0\n"
                           "POINTS %lld\n"
                           "DATA binary\n",
                           (long long)n_points, (long long)n_points);
    if (w < 0 || (size_t)w >= cap) {
        set_error("PCD header needs %d bytes", w + 1);
        return ORBGPU_ECAPACITY;
    }
    if (len)
        *len = (size_t)w;
    return ORBGPU_OK;
}

int orbgpu_write_pcd_binary(const char *path, const orbgpu_point_xyzrgba *points, int64_t n)
{
    ORBGPU_REQUIRE(path && n >= 0 && (n == 0 || points), "bad arguments");
    char head[512];
    size_t len = 0;
    int rc = orbgpu_pcd_binary_header(n, head, sizeof(head), &len);
    if (rc != ORBGPU_OK)
        return rc;
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error("cannot open %s for writing", path);
        return ORBGPU_EINVAL;
    }
    bool ok = fwrite(head, 1, len, f) == len;
    if (ok && n > 0)
        ok = fwrite(points, sizeof(orbgpu_point_xyzrgba), (size_t)n, f) == (size_t)n;
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        set_error("short write to %s", path);
        return ORBGPU_EINVAL;
    }
    return ORBGPU_OK;
}

int orbgpu_cloud_save_pcd(orbgpu_cloud *h, const char *path)
{
    ORBGPU_REQUIRE(h && path, "null argument");
    int64_t n = 0;
    int rc = orbgpu_cloud_size(h, &n);
    if (rc != ORBGPU_OK)
        return rc;
    std::vector<orbgpu_point_xyzrgba> pts((size_t)std::max<int64_t>(n, 1));
    if ((rc = orbgpu_cloud_download(h, pts.data(), (int64_t)pts.size(), &n)) != ORBGPU_OK)
        return rc;
    return orbgpu_write_pcd_binary(path, pts.data(), n);
}

} // extern "C"
