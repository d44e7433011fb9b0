// Frame glue between the extractor and the projection matchers, device resident:
//   Frame::ComputeStereoFromRGBD  (reference src/Frame.cc:641-662)
//   Frame::AssignFeaturesToGrid / PosInGrid  (src/Frame.cc:230-245, 382-392)
// for a batch of frames whose key points were just written by orbgpu_extract_batch_device, so that
// extract -> mvuRight/mvDepth -> mGrid needs no host round trip.  mvKeysUn == mvKeys (zero distortion,
// Frame.cc:406-410); cv::undistortPoints is not restated here.
#include "common.h"

namespace orbgpu {

constexpr int FG_COLS = ORBGPU_GRID_COLS, FG_ROWS = ORBGPU_GRID_ROWS, FG_CELLS = FG_COLS * FG_ROWS;

// One workgroup per frame.
//  (1) mvuRight / mvDepth from the depth image at the (truncated) key-point position;
//  (2) counting sort of the key points into the 64x48 grid: LDS histogram, scan, scatter with atomics,
//      then every cell's short list is put back into insertion order (ascending key-point index), which is
//      the order Frame::AssignFeaturesToGrid's push_back produces and the matchers' tie-breaking depends on.
__global__ __launch_bounds__(1024) void k_frame_glue(const orbgpu_keypoint *__restrict__ kps,
                                                     const int *__restrict__ n_kp, int cap,
                                                     const float *__restrict__ depth, size_t depth_stride,
                                                     size_t depth_frame_stride, float mbf, float min_x, float min_y,
                                                     float inv_w, float inv_h, float *__restrict__ u_right,
                                                     float *__restrict__ kp_depth, int *__restrict__ cell_start,
                                                     int *__restrict__ cell_items)
{
    __shared__ int cnt[FG_CELLS + 1];
    __shared__ int pos[FG_CELLS];
    __shared__ int s_w[16];
    const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n = min(max(n_kp[f], 0), cap);
    const orbgpu_keypoint *k = kps + (size_t)f * cap;
    for (int c = tid; c <= FG_CELLS; c += nt)
        cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const float x = k[i].x, y = k[i].y;
        if (depth) {
            // imDepth.at<float>(v,u): float -> int truncation (Frame.cc:654)
            const float d = depth[(size_t)f * depth_frame_stride + (size_t)(int)y * depth_stride + (size_t)(int)x];
            float ur = -1.f, dz = -1.f;
            if (d > 0) {
                dz = d;
                ur = x - mbf / d;
            }
            u_right[(size_t)f * cap + i] = ur;
            kp_depth[(size_t)f * cap + i] = dz;
        }
        const int px = (int)roundf((x - min_x) * inv_w), py = (int)roundf((y - min_y) * inv_h);
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
        const int px = (int)roundf((k[i].x - min_x) * inv_w), py = (int)roundf((k[i].y - min_y) * inv_h);
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
                                              float mbf, float min_x, float max_x, float min_y, float max_y,
                                              float *d_u_right, float *d_kp_depth, int32_t *d_cell_start,
                                              int32_t *d_cell_items, void *hip_stream)
{
    ORBGPU_REQUIRE(batch >= 1 && cap >= 1, "bad batch/cap");
    ORBGPU_REQUIRE(d_kps && d_n && d_cell_start && d_cell_items, "null argument");
    ORBGPU_REQUIRE(!d_depth || (d_u_right && d_kp_depth), "depth given without stereo outputs");
    ORBGPU_REQUIRE(max_x > min_x && max_y > min_y, "empty image bounds");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    // mfGridElementWidthInv / HeightInv, Frame.cc:155-156
    const float inv_w = (float)FG_COLS / (max_x - min_x), inv_h = (float)FG_ROWS / (max_y - min_y);
    hipLaunchKernelGGL(k_frame_glue, dim3(batch), dim3(1024), 0, (hipStream_t)hip_stream, d_kps, d_n, cap, d_depth,
                       depth_stride, depth_frame_stride, mbf, min_x, min_y, inv_w, inv_h, d_u_right, d_kp_depth,
                       d_cell_start, d_cell_items);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}
