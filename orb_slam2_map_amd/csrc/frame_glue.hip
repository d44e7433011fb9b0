// Frame glue between the extractor and the projection matchers, device resident:
//   Frame::ComputeStereoFromRGBD  (reference src/Frame.cc:641-662)
//   Frame::AssignFeaturesToGrid / PosInGrid  (src/Frame.cc:230-245, 382-392)
//   Frame::UndistortKeyPoints  (src/Frame.cc:404-434, cv::undistortPoints with R = I, P = K)
// for a batch of frames whose key points were just written by orbgpu_extract_batch_device, so that
// extract -> mvKeysUn -> mvuRight/mvDepth -> mGrid needs no host round trip.
#include "common.h"

namespace orbgpu {

// cv::undistortPoints for one point, OpenCV 2.4 cvUndistortPoints: double arithmetic, five fixed-point iterations
// of the Brown model (k1 k2 p1 p2 k3; the rational terms are zero), then reprojection with P = K.  Evaluated
// exactly in the order of the C source (no contraction: the library is built with -ffp-contract=off).
__device__ __forceinline__ void undistort_point(float xin, float yin, const orbgpu_camera &cam, float &xo, float &yo)
{
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = cam.dist[0], k1 = cam.dist[1], k2 = cam.dist[2], k3 = cam.dist[3], k4 = cam.dist[4];
    double x = xin, y = yin;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx;
    const double yy = 0. * x + fy * y + cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    xo = (float)(xx * ww);
    yo = (float)(yy * ww);
}

__global__ void k_undistort_points(int n, const float *__restrict__ in, orbgpu_camera cam, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float x, y;
    undistort_point(in[2 * i], in[2 * i + 1], cam, x, y);
    out[2 * i] = x;
    out[2 * i + 1] = y;
}

constexpr int FG_COLS = ORBGPU_GRID_COLS, FG_ROWS = ORBGPU_GRID_ROWS, FG_CELLS = FG_COLS * FG_ROWS;

// One workgroup per frame.
//  (1) mvuRight / mvDepth from the depth image at the (truncated) key-point position;
//  (2) counting sort of the key points into the 64x48 grid: LDS histogram, scan, scatter with atomics,
//      then every cell's short list is put back into insertion order (ascending key-point index), which is
//      the order Frame::AssignFeaturesToGrid's push_back produces and the matchers' tie-breaking depends on.
__global__ __launch_bounds__(1024) void k_frame_glue(const orbgpu_keypoint *__restrict__ kps,
                                                     const int *__restrict__ n_kp, int cap,
                                                     const float *__restrict__ depth, size_t depth_stride,
                                                     size_t depth_frame_stride, orbgpu_camera cam, float inv_w,
                                                     float inv_h, orbgpu_keypoint *__restrict__ kps_un,
                                                     float *__restrict__ u_right, float *__restrict__ kp_depth,
                                                     int *__restrict__ cell_start, int *__restrict__ cell_items)
{
    const float mbf = cam.mbf, min_x = cam.min_x, min_y = cam.min_y;
    const bool undistort = cam.dist[0] != 0.0f;  // Frame.cc:406
    __shared__ int cnt[FG_CELLS + 1];
    __shared__ int pos[FG_CELLS];
    __shared__ int s_w[16];
    const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n = min(max(n_kp[f], 0), cap);
    const orbgpu_keypoint *k = kps + (size_t)f * cap;
    for (int c = tid; c <= FG_CELLS; c += nt)
        cnt[c] = 0;
    if (kps_un) {  // mvKeysUn: a copy of mvKeys with the undistorted position (Frame.cc:425-433)
        orbgpu_keypoint *ku = kps_un + (size_t)f * cap;
        for (int i = tid; i < n; i += nt) {
            orbgpu_keypoint kp = k[i];
            if (undistort)
                undistort_point(kp.x, kp.y, cam, kp.x, kp.y);
            ku[i] = kp;
        }
    }
    __syncthreads();
    const orbgpu_keypoint *kun = kps_un ? kps_un + (size_t)f * cap : k;  // grid and uRight use mvKeysUn
    for (int i = tid; i < n; i += nt) {
        const float x = k[i].x, y = k[i].y;
        const float xu = kun[i].x, yu = kun[i].y;
        if (depth) {
            // imDepth.at<float>(v,u): float -> int truncation (Frame.cc:654)
            const float d = depth[(size_t)f * depth_frame_stride + (size_t)(int)y * depth_stride + (size_t)(int)x];
            float ur = -1.f, dz = -1.f;
            if (d > 0) {
                dz = d;
                ur = xu - mbf / d;
            }
            u_right[(size_t)f * cap + i] = ur;
            kp_depth[(size_t)f * cap + i] = dz;
        }
        const int px = (int)roundf((xu - min_x) * inv_w), py = (int)roundf((yu - min_y) * inv_h);
        if (px >= 0 && px < FG_COLS && py >= 0 && py < FG_ROWS)
            atomicAdd(&cnt[px * FG_ROWS + py], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0..FG_CELLS) (3072 = 3 per thread at 1024 threads)
    {
        const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
        const int per = (FG_CELLS + nt - 1) / nt;
        const int beg = min(tid * per, FG_CELLS), end = min(beg + per, FG_CELLS);
        int sum = 0;
        for (int c = beg; c < end; c++)
            sum += cnt[c];
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if (lane >= off)
                inc += t;
        }
        if (lane == 63)
            s_w[wave] = inc;
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < nw; w++) {
            if (w < wave)
                woff += s_w[w];
            total += s_w[w];
        }
        int run = woff + inc - sum;
        for (int c = beg; c < end; c++) {
            const int t = cnt[c];
            cnt[c] = run;
            pos[c] = run;
            run += t;
        }
        if (tid == 0)
            cnt[FG_CELLS] = total;
    }
    __syncthreads();
    int *cs = cell_start + (size_t)f * (FG_CELLS + 1);
    int *items = cell_items + (size_t)f * cap;
    for (int c = tid; c <= FG_CELLS; c += nt)
        cs[c] = cnt[c];
    for (int i = tid; i < n; i += nt) {
        const int px = (int)roundf((kun[i].x - min_x) * inv_w), py = (int)roundf((kun[i].y - min_y) * inv_h);
        if (px >= 0 && px < FG_COLS && py >= 0 && py < FG_ROWS)
            items[atomicAdd(&pos[px * FG_ROWS + py], 1)] = i;
    }
    __syncthreads();
    // restore insertion order inside every cell (lists are short: insertion sort by one thread per cell)
    for (int c = tid; c < FG_CELLS; c += nt) {
        const int b = cnt[c], e = cnt[c + 1];
        for (int a = b + 1; a < e; a++) {
            const int v = items[a];
            int j = a - 1;
            while (j >= b && items[j] > v) {
                items[j + 1] = items[j];
                j--;
            }
            items[j + 1] = v;
        }
    }
}

} // namespace orbgpu

using namespace orbgpu;

extern "C" int orbgpu_frame_glue_batch_device(int32_t device_id, int32_t batch, int32_t cap,
                                              const orbgpu_keypoint *d_kps, const int32_t *d_n,
                                              const float *d_depth, size_t depth_stride, size_t depth_frame_stride,
                                              const orbgpu_camera *cam, orbgpu_keypoint *d_kps_un, float *d_u_right,
                                              float *d_kp_depth, int32_t *d_cell_start, int32_t *d_cell_items,
                                              void *hip_stream)
{
    ORBGPU_REQUIRE(batch >= 1 && cap >= 1, "bad batch/cap");
    ORBGPU_REQUIRE(d_kps && d_n && d_cell_start && d_cell_items && cam, "null argument");
    ORBGPU_REQUIRE(!d_depth || (d_u_right && d_kp_depth), "depth given without stereo outputs");
    ORBGPU_REQUIRE(cam->max_x > cam->min_x && cam->max_y > cam->min_y, "empty image bounds");
    ORBGPU_REQUIRE(cam->dist[0] == 0.0f || d_kps_un, "non-zero distortion needs the mvKeysUn output");
    ORBGPU_REQUIRE(cam->dist[0] == 0.0f || (cam->fx != 0.0f && cam->fy != 0.0f), "bad camera matrix");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    // mfGridElementWidthInv / HeightInv, Frame.cc:155-156
    const float inv_w = (float)FG_COLS / (cam->max_x - cam->min_x), inv_h = (float)FG_ROWS / (cam->max_y - cam->min_y);
    hipLaunchKernelGGL(k_frame_glue, dim3(batch), dim3(1024), 0, (hipStream_t)hip_stream, d_kps, d_n, cap, d_depth,
                       depth_stride, depth_frame_stride, *cam, inv_w, inv_h, d_kps_un, d_u_right, d_kp_depth,
                       d_cell_start, d_cell_items);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

extern "C" int orbgpu_undistort_points(int32_t n, const float *xy_in, const orbgpu_camera *cam, float *xy_out,
                                       int32_t device_id)
{
    ORBGPU_REQUIRE(n >= 0 && cam && (n == 0 || (xy_in && xy_out)), "bad arguments");
    ORBGPU_REQUIRE(cam->fx != 0.0f && cam->fy != 0.0f, "bad camera matrix");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK || n == 0)
        return rc;
    struct Scoped : DevBuf {
        ~Scoped() { release(); }
    } in, out;
    if ((rc = in.reserve((size_t)n * 8)) != ORBGPU_OK || (rc = out.reserve((size_t)n * 8)) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpy(in.p, xy_in, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_undistort_points, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, in.as<float>(), *cam,
                       out.as<float>());
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipMemcpy(xy_out, out.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return ORBGPU_OK;
}
