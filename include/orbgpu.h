/*
 * orbgpu.h -- C ABI of liborbgpu.so: MI355X (gfx950) implementation of the per-frame hot path
 * of carry4985/ORB_SLAM2_MAP (ORBextractor + ORBmatcher + PointCloudMapping arithmetic).
 *
 * The reference has no FFI/plugin layer: its boundary is three C++ classes compiled into
 * libORB_SLAM2.so (reference CMakeLists.txt:63-84).  Every entry point below names the
 * reference interface (file:line, relative to the reference tree) it replaces; the C++ shims
 * that re-create those classes on top of this ABI are in orb_slam2_map_amd/shim/ and the binding a
 * maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types; never throws across the ABI;
 *  - every function returns an orbgpu_status (0 = OK, <0 = error); orbgpu_last_error_string()
 *    gives a thread-local message;
 *  - "host" entry points take host pointers and synchronise before returning;
 *    "*_device" entry points take device pointers plus a hipStream_t (passed as void*), enqueue
 *    work on that stream and return without synchronising;
 *  - a handle must not be used from two threads at once; distinct handles are independent.
 *  - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *    ORBGPU_EHIP.
 */
#ifndef ORBGPU_H
#define ORBGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBGPU_ABI_VERSION 1
#define ORBGPU_MAX_LEVELS 16

typedef enum {
    ORBGPU_OK = 0,
    ORBGPU_EINVAL = -1,    /* bad argument */
    ORBGPU_ENOMEM = -2,    /* host or device allocation failed */
    ORBGPU_EHIP = -3,      /* HIP runtime error / no device */
    ORBGPU_ECAPACITY = -4, /* caller-provided output capacity too small */
    ORBGPU_ELEVEL = -5     /* predicted pyramid level outside [0,nlevels) (UB in the reference:
                              ORBmatcher.cc:69 indexes mvScaleFactors with an unclamped
                              MapPoint::PredictScale, MapPoint.cc:385-394) */
} orbgpu_status;

const char *orbgpu_last_error_string(void);
int orbgpu_abi_version(void);
/* Number of visible HIP devices (0 if none); never fails. */
int orbgpu_device_count(void);
/* Measurement aid: achieved HBM GB/s (bytes read + bytes written per second) of a plain 16-byte-per-lane
 * device-to-device copy kernel over `bytes` (>= 1 MiB; use a size well past the 256 MB Infinity Cache) --
 * the practical roofline bench.py reports next to the nominal 8 TB/s. */
int orbgpu_measure_copy_bandwidth(size_t bytes, int32_t reps, int32_t device_id, float *gbs);

/* ======================================================================================
 * ORBextractor  (reference include/ORBextractor.h:46-110, src/ORBextractor.cc)
 * ====================================================================================== */

/* cv::KeyPoint layout (28 B): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbgpu_keypoint;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.h:51-52; values come from the YAML keys read at Tracking.cc:193-197). */
typedef struct {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t device_id; /* HIP device ordinal */
    int32_t max_batch; /* frames per batched launch the handle pre-sizes for (>=1) */
} orbgpu_extractor_params;

typedef struct orbgpu_extractor orbgpu_extractor;

/* replaces `new ORBextractor(...)` (Tracking.cc:201-213) */
int orbgpu_extractor_create(const orbgpu_extractor_params *params, orbgpu_extractor **out);
int orbgpu_extractor_destroy(orbgpu_extractor *h);

/* computeOrbDescriptor's  a = (float)cos(angle), b = (float)sin(angle)  (ORBextractor.cc:112-113).  With
 * `using namespace std;` in force (:66-67) the calls resolve to std::cos(float) / std::sin(float) = libm's cosf / sinf,
 * which are not correctly rounded: their last bit is a property of the host's libm.  Process-wide, read by
 * orbgpu_extractor_create:
 *   ORBGPU_TRIG_HOST_LIBM (default)  the values THIS host's cosf / sinf return, bit for bit: the device evaluates a fixed
 *       IEEE-double sequence and corrects it from a table of the arguments where the host differs, built by scanning
 *       every float of [0, 2 pi] once per process at the first extractor creation (about 15 core-seconds, spread over
 *       the CPUs the process may use; 18 MB on the device);
 *   ORBGPU_TRIG_ROUNDED_DOUBLE       (float)cos((double)angle): the correctly rounded value, no table, no start-up scan
 *       (differs from glibc 2.35's cosf / sinf in 1.3e-3 of the arguments, from a descriptor bit far more rarely:
 *       DESIGN.md section 2). */
enum { ORBGPU_TRIG_HOST_LIBM = 0, ORBGPU_TRIG_ROUNDED_DOUBLE = 1 };
int orbgpu_set_trig_mode(int32_t mode);
int orbgpu_get_trig_mode(void);
/* entries of the exception table (-1: not built yet) and the time its scan took */
int orbgpu_trig_table_info(int64_t *entries, double *build_ms);
/* Test aid, needs no device: cos / sin of x as ORBGPU_TRIG_HOST_LIBM delivers them (builds the table if necessary);
 * *from_table = 1 if the table corrected the base value. */
int orbgpu_trig_host_eval(float x, float *c, float *s, int32_t *from_table);
/* Test aid: the DEVICE's values (the function the descriptor stage calls) for n host floats, current mode. */
int orbgpu_trig_eval(const float *x, int32_t n, float *c, float *s, int32_t device_id);

/* GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (ORBextractor.h:63-83). Arrays receive nlevels floats. */
int orbgpu_extractor_get_levels(const orbgpu_extractor *h, int32_t *nlevels);
int orbgpu_extractor_get_scale_factor(const orbgpu_extractor *h, float *scale_factor);
int orbgpu_extractor_get_scale_factors(const orbgpu_extractor *h, float *out);
int orbgpu_extractor_get_inv_scale_factors(const orbgpu_extractor *h, float *out);
int orbgpu_extractor_get_sigma2(const orbgpu_extractor *h, float *out);
int orbgpu_extractor_get_inv_sigma2(const orbgpu_extractor *h, float *out);
/* mnFeaturesPerLevel (ORBextractor.h:102) */
int orbgpu_extractor_get_quotas(const orbgpu_extractor *h, int32_t *out);
/* Upper bound of keypoints one frame can produce (SURVEY.md E3': a level may return up to
 * max(4*nIni, quota+2) keypoints) -- the `cap` every extract call needs. */
int orbgpu_extractor_max_keypoints(const orbgpu_extractor *h, int32_t width, int32_t height, int32_t *cap);

/* ORBextractor::operator()(image, mask, keypoints, descriptors) (ORBextractor.h:59-61,
 * ORBextractor.cc:1043-1105; called from Frame::ExtractORB, Frame.cc:247-253).  The mask is
 * ignored by the reference (ORBextractor.h:58) and has no parameter here.
 * gray: 8-bit single channel, `stride` bytes per row.  kps[cap], desc[cap*32].  An empty image
 * (width or height 0) returns OK with *n_out = 0 (ORBextractor.cc:1046). */
int orbgpu_extract(orbgpu_extractor *h, const uint8_t *gray, int32_t width, int32_t height, size_t stride,
                   orbgpu_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out);

/* `batch` independent frames of one size in one launch sequence (frames x levels as grid
 * dimensions).  gray: batch images, `frame_stride` bytes apart.  Outputs are [batch][cap]. */
int orbgpu_extract_batch(orbgpu_extractor *h, const uint8_t *gray, int32_t batch, int32_t width, int32_t height,
                         size_t stride, size_t frame_stride, orbgpu_keypoint *kps, uint8_t *desc, int32_t cap,
                         int32_t *n_out);

/* Device-resident variant: all pointers are device pointers, nothing is copied or synchronised;
 * d_n_out[batch] receives the per-frame keypoint counts.  If a frame would exceed `cap` its
 * count is reported as -1 - (required count) and its outputs are unspecified. */
int orbgpu_extract_batch_device(orbgpu_extractor *h, const uint8_t *d_gray, int32_t batch, int32_t width,
                                int32_t height, size_t stride, size_t frame_stride, orbgpu_keypoint *d_kps,
                                uint8_t *d_desc, int32_t cap, int32_t *d_n_out, void *hip_stream);

/* mvImagePyramid[level] (public member, ORBextractor.h:85; read by Frame::ComputeStereoMatches,
 * Frame.cc:471,561,578).  Copies level `level` of frame `frame` of the LAST call (without the
 * 19 px border) to host memory, dst_stride bytes per row.
 * Level 0: when the input rows of the last call were 4-byte aligned (base pointer, stride, frame stride) and the width a
 * multiple of 8, no stage needs ComputePyramid's padded copy of the image (ORBextractor.cc:1125-1129: FAST, the
 * orientation disc and the resize only read the image interior, the blur reflects its own rim) and the library does not
 * make one ("direct mode"); this getter then makes it on demand from the image of the last call.  For the host entry
 * points that is the handle's own staging copy; after orbgpu_extract_batch_device it is the CALLER's device buffer, which
 * must still hold the images when level 0 is asked for (the RGB-D path never asks: only stereo matching reads it). */
int orbgpu_extractor_get_pyramid_level(orbgpu_extractor *h, int32_t frame, int32_t level, uint8_t *dst,
                                       size_t dst_stride, int32_t *width, int32_t *height);

/* Stage introspection of the last call, for stage-by-stage parity tests (not part of the
 * drop-in surface).  what: */
enum {
    ORBGPU_DBG_PYRAMID_PADDED = 0, /* u8, (h+38) rows of `pitch` bytes; *n = bytes, aux = pitch   */
    ORBGPU_DBG_BLURRED_PADDED = 1, /* same geometry, only the w x h interior is defined           */
    ORBGPU_DBG_CANDIDATES = 2,     /* int32 triples (x,y,response), vToDistributeKeys order       */
    ORBGPU_DBG_SELECTED = 3        /* int32 triples (x,y,response), DistributeOctTree list order  */
};
int orbgpu_extractor_debug_read(orbgpu_extractor *h, int32_t what, int32_t frame, int32_t level, void *dst,
                                size_t dst_bytes, size_t *n, int32_t *aux);

/* Per-stage device time (ms), measured with HIP events recorded on the call's own stream at the stage
 * boundaries of every call made while profiling is enabled (up to 256 calls per window; enabling and
 * disabling keeps the window, so a caller may profile a sample of its calls).
 * Names are returned by orbgpu_extractor_stage_name(i); count by orbgpu_extractor_stage_count(). */
/* The host entry points (orbgpu_extract, orbgpu_extract_batch) replay their launch sequence as a hipGraph from the
 * third call of a configuration on: *state = 1 graph in use, 0 not recorded yet, -1 recording failed (plain launches).
 * The graph is built node by node (hipGraphAddKernelNode), never by stream capture: nothing process-wide is touched. */
int orbgpu_extractor_graph_state(const orbgpu_extractor *h, int32_t *state);
/* How DistributeOctTree (ORBextractor.cc:539-763) is launched for a call of `batch` frames at the configured image size:
 * keys of a level the workgroup keeps in LDS (0 = the batch variant, all keys in memory), threads per workgroup and the
 * dynamic LDS bytes.  Test aid: the environment hook ORBGPU_DEBUG_QT_KEYS (read by orbgpu_extractor_create) shrinks
 * the LDS share so that the mixed LDS / memory path runs; this getter tells a test that it really did. */
int orbgpu_extractor_debug_quadtree_config(const orbgpu_extractor *h, int32_t batch, int32_t *lds_keys, int32_t *threads,
                                           int32_t *lds_bytes);
int orbgpu_extractor_set_profiling(orbgpu_extractor *h, int32_t enable);
/* Throughput option for resident batches (no counterpart in the reference): the 7x7 blur (ORBextractor.cc:1085-1086;
 * HBM-bound) runs on a stream owned by the handle next to the FAST pass (VALU-bound) and the quadtree, forked after the
 * pyramid and joined in front of the descriptor stage.  Results are unchanged; with profiling on, the blur's stage time
 * is the one measured on its own stream (it overlaps the "fast" / "quadtree" stage times). */
int orbgpu_extractor_set_concurrent_blur(orbgpu_extractor *h, int32_t enable);
/* Option of the FAST stage (no counterpart in the reference, results unchanged): before the corner-score network of a
 * pixel pair runs, an exact bound from the four compass pixels of the ring (cv::FAST's own high-speed test, A4: an arc of
 * 9 contains two adjacent compass pixels) is evaluated; where it clears every pixel pair of a wavefront's row segment, the
 * network is skipped.  Pays on images with flat regions (walls, table tops); costs ~20 % of the FAST stage on images that
 * are textured everywhere, like the synthetic benchmark stream -- hence off by default.  Environment variable
 * ORBGPU_FAST_EARLY_OUT=1 turns it on for handles created afterwards. */
int orbgpu_extractor_set_fast_early_out(orbgpu_extractor *h, int32_t enable);

/* Pipelining aid for callers that run other work next to an extraction (bench.py starts the matcher of the previous
 * batch there): `hip_event` (a hipEvent_t, or NULL to clear) is recorded on the launch stream of every later
 * orbgpu_extract_batch_device call right after stage `stage` (index as in orbgpu_extractor_stage_name). */
int orbgpu_extractor_set_stage_signal(orbgpu_extractor *h, int32_t stage, void *hip_event);
int orbgpu_extractor_stage_count(void);
const char *orbgpu_extractor_stage_name(int32_t i);
/* Synchronises the recorded events, returns the AVERAGE ms per stage over the calls profiled since
 * the last stage_times, and starts a new window. */
int orbgpu_extractor_stage_times(orbgpu_extractor *h, float *ms_out);

/* Throughput mode of ORBextractor::operator() over a resident batch (no counterpart in the reference, which extracts one
 * frame per call, Frame.cc:247-253): the batch is cut into `parts` sub-batches, each with an extractor handle and a HIP
 * stream of its own; part k starts when part k-1 (part 0: the last part of the previous call) has passed its pyramid
 * stage, so the HBM-bound, VALU-bound and latency-bound stages of neighbouring parts overlap -- across calls too, because
 * a call only ENQUEUES.  params->max_batch = frames per call.  Results are bit-identical to orbgpu_extract_batch_device.
 *   wait_event (hipEvent_t or NULL): every part waits for it first (inputs uploaded, output buffers free);
 *   done_event (hipEvent_t or NULL): recorded when every part of THIS call has finished;
 *   orbgpu_pipeline_wait: makes `hip_stream` wait for everything enqueued so far (plain stream-ordered use);
 *   orbgpu_pipeline_part: the handle of part k, for set_profiling / stage_times / set_stage_signal. */
typedef struct orbgpu_pipeline orbgpu_pipeline;
int orbgpu_pipeline_create(const orbgpu_extractor_params *params, int32_t parts, orbgpu_pipeline **out);
int orbgpu_pipeline_destroy(orbgpu_pipeline *pl);
int orbgpu_pipeline_parts(const orbgpu_pipeline *pl, int32_t *parts);
int orbgpu_pipeline_part(orbgpu_pipeline *pl, int32_t k, orbgpu_extractor **part);
int orbgpu_pipeline_extract_device(orbgpu_pipeline *pl, const uint8_t *d_gray, int32_t batch, int32_t width,
                                   int32_t height, size_t stride, size_t frame_stride, orbgpu_keypoint *d_kps,
                                   uint8_t *d_desc, int32_t cap, int32_t *d_n_out, void *wait_event, void *done_event);
int orbgpu_pipeline_wait(orbgpu_pipeline *pl, void *hip_stream);

/* ======================================================================================
 * ORBmatcher  (reference include/ORBmatcher.h:41-106, src/ORBmatcher.cc)
 * ====================================================================================== */

#define ORBGPU_TH_HIGH 100     /* ORBmatcher::TH_HIGH  ORBmatcher.cc:37 */
#define ORBGPU_TH_LOW 50       /* ORBmatcher::TH_LOW   ORBmatcher.cc:38 */
#define ORBGPU_HISTO_LENGTH 30 /* ORBmatcher::HISTO_LENGTH ORBmatcher.cc:39 */
#define ORBGPU_GRID_COLS 64    /* FRAME_GRID_COLS Frame.h:38 */
#define ORBGPU_GRID_ROWS 48    /* FRAME_GRID_ROWS Frame.h:37 */

/* ORBmatcher::DescriptorDistance (ORBmatcher.h:44, ORBmatcher.cc:1647-1663) for n pairs:
 * out[i] = Hamming(a[i], b[i]) over 256 bits. Host pointers. */
int orbgpu_hamming256(const uint8_t *a, const uint8_t *b, int32_t n, int32_t *out, int32_t device_id);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:242-307), batched over `groups` map points: group g owns the
 * descriptors [offsets[g], offsets[g+1]) of desc (the rows of its non-bad observing key frames, in the iteration
 * order of mObservations).  best_idx[g] = index INSIDE the group of the descriptor with the least median Hamming
 * distance to the group (median = sorted row [(size_t)(0.5*(N-1))], first minimum), -1 for an empty group (the
 * reference leaves mDescriptor untouched).  Host pointers; at most 2048 descriptors per group. */
int orbgpu_distinctive_descriptors(int32_t groups, const int32_t *offsets, const uint8_t *desc, int32_t *best_idx,
                                   int32_t device_id);

/* Brute-force 256-bit Hamming matcher with the acceptance rule, greedy claim order and rotation
 * consistency of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:159-288) when all
 * features share one vocabulary node (the reference has no other brute-force matcher).
 * For each valid A row in index order: best/second-best over B rows not yet claimed; accept iff
 * best <= th_low and (float)best < nnratio*(float)second; claim.  match_b[j] = A index or -1.
 * valid_a may be NULL.  angle_* are cv::KeyPoint::angle (degrees), used iff check_orientation. */
int orbgpu_match_bf(const uint8_t *desc_a, const float *angle_a, const uint8_t *valid_a, int32_t na,
                    const uint8_t *desc_b, const float *angle_b, int32_t nb, int32_t th_low, float nnratio,
                    int32_t check_orientation, int32_t *match_b, int32_t *nmatches, int32_t device_id);

/* Batched device-resident variant: `pairs` independent (A,B) problems.  Row p of every array is
 * `cap` elements apart; na[p]/nb[p] are device int32 counts (e.g. d_n_out of the extractor).
 * d_desc: [pairs][cap][32]; d_angle_stride_bytes lets the angle be read in place from an
 * orbgpu_keypoint array (stride 28) or from a packed float array (stride 4).
 * d_match_b: [pairs][cap]; d_nmatches: [pairs].  d_valid_a may be NULL. */
typedef struct orbgpu_matcher orbgpu_matcher;
int orbgpu_matcher_create(int32_t device_id, int32_t max_pairs, int32_t cap, orbgpu_matcher **out);
int orbgpu_matcher_destroy(orbgpu_matcher *m);
int orbgpu_match_bf_batch_device(orbgpu_matcher *m, int32_t pairs, int32_t cap, const uint8_t *d_desc_a,
                                 const void *d_angle_a, const uint8_t *d_valid_a, const int32_t *d_na,
                                 const uint8_t *d_desc_b, const void *d_angle_b, const int32_t *d_nb,
                                 size_t angle_stride_bytes, int32_t th_low, float nnratio,
                                 int32_t check_orientation, int32_t *d_match_b, int32_t *d_nmatches,
                                 void *hip_stream);
/* Fixpoint sweeps the last batched call needed (diagnostic). */
int orbgpu_matcher_last_sweeps(orbgpu_matcher *m, int32_t *sweeps);

/* SoA view of the Frame members the projection matchers read (Frame.h:100-190):
 * mvKeysUn (pt, octave, angle), mvuRight, mDescriptors, image bounds, grid metrics, the
 * extractor's scale factors, and mGrid flattened to CSR in (ix*ROWS+iy) order with items in
 * insertion order (Frame::AssignFeaturesToGrid, Frame.cc:230-245). Host pointers. */
typedef struct {
    int32_t n;
    const float *kp_x, *kp_y;
    const int32_t *kp_octave;
    const float *kp_angle;
    const float *u_right;
    const uint8_t *desc;
    float min_x, max_x, min_y, max_y;
    float grid_inv_w, grid_inv_h;
    const float *scale_factors;
    int32_t nlevels;
    const int32_t *cell_start; /* COLS*ROWS+1 */
    const int32_t *cell_items; /* cell_start[COLS*ROWS] entries */
} orbgpu_frame_view;

/* Frame::AssignFeaturesToGrid + PosInGrid (Frame.cc:230-245, 382-392): fills the CSR arrays. */
int orbgpu_assign_features_to_grid(int32_t n, const float *kp_x, const float *kp_y, float min_x, float min_y,
                                   float grid_inv_w, float grid_inv_h, int32_t *cell_start, int32_t *cell_items);

/* Camera of a Frame: mK (Frame.h:107-111), mDistCoef (k1 k2 p1 p2 k3; k3 = 0 for the 4-entry form, Tracking.cc:96-106),
 * mbf, and the undistorted image bounds mnMinX..mnMaxY (Frame::ComputeImageBounds, Frame.cc:436-468). */
typedef struct orbgpu_camera {
    float fx, fy, cx, cy;
    float dist[5];
    float mbf;
    float min_x, max_x, min_y, max_y;
} orbgpu_camera;

/* cv::undistortPoints(src, dst, mK, mDistCoef, Mat(), mK) as Frame::UndistortKeyPoints (Frame.cc:404-434) and
 * Frame::ComputeImageBounds (:436-468) call it: n interleaved (x, y) pairs, host pointers, evaluated on the device
 * in double exactly as OpenCV 2.4 cvUndistortPoints does (5 iterations).  For the four image corners this gives
 * the bounds to put into orbgpu_camera. */
int orbgpu_undistort_points(int32_t n, const float *xy_in, const orbgpu_camera *cam, float *xy_out, int32_t device_id);

/* Device-resident Frame glue for a batch of frames straight out of orbgpu_extract_batch_device:
 * Frame::UndistortKeyPoints (Frame.cc:404-434; d_kps_un [batch][cap] receives mvKeysUn, required when
 * cam->dist[0] != 0, optional otherwise), Frame::ComputeStereoFromRGBD (Frame.cc:641-662: depth looked up at the
 * distorted key point, mvuRight = kpUn.x - mbf/d, mvDepth = d where d > 0, else -1) and
 * Frame::AssignFeaturesToGrid (Frame.cc:230-245, on mvKeysUn) as CSR per frame (cell_start[batch][COLS*ROWS+1],
 * cell_items[batch][cap], items in insertion order).  d_depth may be NULL (no stereo outputs); depth strides are
 * in floats.  All pointers except cam are device pointers; nothing is synchronised. */
int orbgpu_frame_glue_batch_device(int32_t device_id, int32_t batch, int32_t cap, const orbgpu_keypoint *d_kps,
                                   const int32_t *d_n, const float *d_depth, size_t depth_stride,
                                   size_t depth_frame_stride, const orbgpu_camera *cam, orbgpu_keypoint *d_kps_un,
                                   float *d_u_right, float *d_kp_depth, int32_t *d_cell_start,
                                   int32_t *d_cell_items, void *hip_stream);

/* ---- Device-resident Tracking::SearchLocalPoints (Tracking.cc:1447-1497) ----------------------------------
 * = Frame::isInFrustum (Frame.cc:269-325) + MapPoint::PredictScale (MapPoint.cc:385-394) over a MapPoint table
 * kept in HBM, followed by ORBmatcher::SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:45-137), for a
 * frame that never left the device (orbgpu_extract_batch_device -> orbgpu_frame_glue_batch_device).  Only the
 * pose crosses PCIe. */
typedef struct orbgpu_device_frame_view {
    int32_t cap;                  /* capacity of the per-frame arrays (upper bound of *n) */
    const int32_t *n;             /* device: N */
    const orbgpu_keypoint *kps;   /* device [cap]: mvKeysUn */
    const uint8_t *desc;          /* device [cap][32]: mDescriptors */
    const float *u_right;         /* device [cap]: mvuRight */
    const int32_t *cell_start;    /* device [COLS*ROWS+1]: mGrid as CSR */
    const int32_t *cell_items;    /* device [cap] */
    int32_t nlevels;
    const float *scale_factors;   /* HOST [nlevels]: mvScaleFactors */
    float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY */
} orbgpu_device_frame_view;

typedef struct orbgpu_device_mappoint_table { /* all device pointers, row i = mvpLocalMapPoints[i] */
    int32_t m;
    const float *world_pos;   /* [m][3] GetWorldPos() */
    const float *normal;      /* [m][3] GetNormal() */
    const float *min_dist;    /* [m] mfMinDistance (the 0.8 / 1.2 invariance factors are applied inside) */
    const float *max_dist;    /* [m] mfMaxDistance */
    const uint8_t *desc;      /* [m][32] GetDescriptor() */
    const uint8_t *skip;      /* [m] or NULL: isBad() || mnLastFrameSeen == F.mnId (Tracking.cc:1474-1477) */
    const uint8_t *obs_pos;   /* [m] or NULL (= all 1): Observations() > 0 (ORBmatcher.cc:87-89) */
} orbgpu_device_mappoint_table;

typedef struct orbgpu_track_scratch { /* optional device outputs: the MapPoint::mTrack* members isInFrustum fills */
    uint8_t *in_view;  /* [m] mbTrackInView (written for every row; drives IncreaseVisible, Tracking.cc:1481) */
    float *proj_x, *proj_y, *proj_xr, *view_cos; /* [m] written where in_view */
    int32_t *level;    /* [m] mnTrackScaleLevel */
} orbgpu_track_scratch;

/* d_kp_to_mp [cap] in/out (device): -1 free, -2 held by a point outside the table, >= 0 row of the table.
 * d_counts [2] (device): [0] = return value of SearchByProjection, [1] = map points whose predicted level fell
 * outside [0, nlevels) (the reference reads mvScaleFactors out of range there; they are left unmatched).
 * Tcw: HOST, 4x4 row-major float (mTcw).  Asynchronous on hip_stream; calls issued by one host thread share a
 * workspace and must be ordered on one stream. */
int orbgpu_search_local_points_device(const orbgpu_device_frame_view *f, const orbgpu_device_mappoint_table *mp,
                                      const float *Tcw, float fx, float fy, float cx, float cy, float mbf,
                                      float log_scale_factor, float cos_limit, float th, float nnratio,
                                      int32_t *d_kp_to_mp, int32_t *d_counts, const orbgpu_track_scratch *d_track,
                                      int32_t device_id, void *hip_stream);

/* The same for n independent problems (one per sequence: frames shard by sequence) in four launches:
 * the claim fixpoint of one frame is one workgroup by construction, so many sequences are what fills the device.
 * Every problem has its own frame, map-point table, pose and outputs; cos_limit / th / nnratio are shared. */
typedef struct orbgpu_local_points_problem {
    const orbgpu_device_frame_view *frame;
    const orbgpu_device_mappoint_table *table;
    const float *Tcw;                     /* HOST 4x4 row-major float */
    float fx, fy, cx, cy, mbf, log_scale_factor;
    int32_t *d_kp_to_mp;                  /* device [frame->cap] in/out */
    int32_t *d_counts;                    /* device [2] */
    const orbgpu_track_scratch *d_track;  /* optional */
} orbgpu_local_points_problem;
int orbgpu_search_local_points_batch_device(int32_t n, const orbgpu_local_points_problem *problems, float cos_limit,
                                            float th, float nnratio, int32_t device_id, void *hip_stream);

/* Device-resident ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
 * (ORBmatcher.cc:1328-1470; Tracking::TrackWithMotionModel, Tracking.cc:1169/1175): the per-frame matcher of
 * RGB-D tracking.  cur = the frame that never left the device; last = the previous frame's key points (octave and
 * angle are read from the records) plus, per key point, what Tracking keeps of its map point.  Both poses are HOST
 * 4x4 row-major floats; nothing else crosses PCIe.  The projection (:1360-1376), the level window by direction of
 * motion (:1339-1349, :1385-1390), the greedy claims, TH_HIGH and the rotation histogram are the reference's.
 * d_kp_to_mp [cur->cap] in/out: -1 free, -2 held by a map point outside `last`, >= 0 row of `last`.
 * d_counts [2]: [0] = nmatches, [1] = rows whose octave fell outside [0, nlevels) (left unmatched). */
typedef struct orbgpu_device_lastframe_view { /* device pointers unless noted */
    int32_t cap;                 /* rows (upper bound of *n) */
    const int32_t *n;            /* LastFrame.N */
    const orbgpu_keypoint *kps;  /* [cap] LastFrame.mvKeys (octave, angle) */
    const uint8_t *has_mp;       /* [cap] LastFrame.mvpMapPoints[i] != NULL */
    const uint8_t *outlier;      /* [cap] or NULL: LastFrame.mvbOutlier[i] */
    const uint8_t *obs_pos;      /* [cap] or NULL (= all 1): Observations() > 0 */
    const float *world_pos;      /* [cap][3] GetWorldPos() */
    const uint8_t *desc;         /* [cap][32] GetDescriptor() of the map point */
} orbgpu_device_lastframe_view;
int orbgpu_search_by_projection_last_device(const orbgpu_device_frame_view *cur, const float *cur_Tcw,
                                            const orbgpu_device_lastframe_view *last, const float *last_Tcw, float fx,
                                            float fy, float cx, float cy, float mbf, float mb, float th, int32_t mono,
                                            int32_t check_orientation, int32_t *d_kp_to_mp, int32_t *d_counts,
                                            int32_t device_id, void *hip_stream);

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
 * vector<MapPoint*>& vpMatched, int th) (ORBmatcher.cc:290-403; loop closing, LoopClosing.cc:397).
 * kf: the key frame as a frame view (mvKeysUn, mDescriptors, mGrid; u_right is not read).  Scw: HOST 4x4 row-major
 * float Sim3 [sR | st].  pts: the candidate points.  kp_to_mp [kf->n] in/out = vpMatched: -1 NULL, -2 a map point
 * outside pts, >= 0 row of pts (those rows form spAlreadyFound and are skipped); every accepted point (best
 * distance <= TH_LOW among the free key points of levels [l-1, l] in the window) is written there.
 * ORBGPU_ELEVEL if a predicted level falls outside [0, nlevels). */
typedef struct orbgpu_points_view {
    int32_t m;
    const uint8_t *bad;      /* [m] isBad(), may be NULL */
    const float *world_pos;  /* [m][3] GetWorldPos() */
    const float *normal;     /* [m][3] GetNormal() */
    const float *min_dist;   /* [m] mfMinDistance */
    const float *max_dist;   /* [m] mfMaxDistance */
    const uint8_t *desc;     /* [m][32] GetDescriptor() */
} orbgpu_points_view;
int orbgpu_search_by_projection_sim3(const orbgpu_frame_view *kf, const float *Scw, float fx, float fy, float cx,
                                     float cy, float log_scale_factor, const orbgpu_points_view *pts, int32_t th,
                                     int32_t *kp_to_mp, int32_t *nmatches, int32_t device_id);

/* Convergence diagnostics of the calling thread's most recent projection match (orbgpu_search_by_projection*,
 * orbgpu_search_local_points_device): claim sweeps run and rows that had to be re-walked with the claim filter.
 * Synchronises the device. */
int orbgpu_projection_last_sweeps(int32_t *sweeps, int32_t *rewalked_rows);

/* MapPoint tracking scratch filled by Frame::isInFrustum (MapPoint.h:91-96, Frame.cc:317-322)
 * plus the flags and descriptor the matcher reads per point (ORBmatcher.cc:53-63, 77, 88). */
typedef struct {
    int32_t m;
    const uint8_t *in_view;  /* mbTrackInView */
    const uint8_t *bad;      /* isBad(); may be NULL */
    const uint8_t *obs_pos;  /* Observations()>0; may be NULL (= all true) */
    const int32_t *level;    /* mnTrackScaleLevel */
    const float *view_cos;   /* mTrackViewCos */
    const float *proj_x, *proj_y, *proj_xr;
    const uint8_t *desc;     /* GetDescriptor(), m x 32 */
} orbgpu_mappoint_view;

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.h:48,
 * ORBmatcher.cc:45-129; called from Tracking::SearchLocalPoints, Tracking.cc:1495).
 * kp_to_mp[f->n] in/out: in  >=0 existing association (index into mp), -1 free, -2 held by a map
 * point outside `mp` with Observations()>0; out: as F.mvpMapPoints after the call. */
int orbgpu_search_by_projection(const orbgpu_frame_view *f, const orbgpu_mappoint_view *mp, float th,
                                float nnratio, int32_t *kp_to_mp, int32_t *nmatches, int32_t device_id);

/* ---- Device-resident MapPoint table ("Host-side gather cost") -------------------------------------
 * What the projection matchers read of a map point -- GetWorldPos(), GetNormal(), mfMinDistance / mfMaxDistance,
 * GetDescriptor() (a per-point mutex + 32-byte clone in the reference, MapPoint.cc:309-313), isBad(), Observations() > 0
 * (ORBmatcher.cc:53-63, 88) -- kept in HBM, keyed by MapPoint::mnId (MapPoint.h:84).  The table is edited where the
 * reference edits the object (INTEGRATION.md section 2c lists the one-line hooks):
 *   upsert            MapPoint::MapPoint (MapPoint.cc:32-71), SetWorldPos (:73-78), UpdateNormalAndDepth (:330-371),
 *                     ComputeDistinctiveDescriptors (:242-307); RGB-D temporal points (Tracking::UpdateLastFrame) with
 *                     n_obs = 0
 *   set_observations  AddObservation / EraseObservation (:98-149)
 *   set_bad           SetBadFlag (:151-175), Replace (:177-228)
 * Per frame the caller passes ids; id -> row lookup, the gather of the call's local map and the translation of the
 * frame's existing associations run on the device.  A table is used by ONE host thread at a time and the CALLER must
 * serialise: most of the reference's edit sites run WITHOUT Map::mMutexMapUpdate (`new MapPoint` at LocalMapping.cc:434,
 * SetBadFlag in MapPointCulling and at Tracking.cc:806 / 937 / 1416; only Tracking.cc:463, Optimizer.cc:746 / 989 and
 * the LoopClosing sites hold it), while Tracking searches the table concurrently -- an unserialised edit mutates the
 * host-side id hash under a running lookup.  The C++ wrapper MapPointTableT serialises its own calls and the matcher
 * overloads that search it through MapPointTableT::mutex().  Every call returns synchronised.
 * Limits: at most 2^26 ids per call, 2^27 rows per table. */
typedef struct orbgpu_mappoint_table orbgpu_mappoint_table;
int orbgpu_mappoint_table_create(int32_t device_id, int32_t initial_rows, orbgpu_mappoint_table **out);
int orbgpu_mappoint_table_destroy(orbgpu_mappoint_table *t);
/* Rows in use: distinct ids inserted since creation or since the last orbgpu_mappoint_table_retain. */
int orbgpu_mappoint_table_rows(const orbgpu_mappoint_table *t, int32_t *rows);
/* Inserts or updates n points (ids distinct within a call, >= 0; host arrays).  An attribute array may be NULL: known
 * ids keep that attribute, new ids get zeros (n_obs NULL: a new point counts as observed).  world_pos / normal
 * [n][3], desc [n][32], n_obs = Observations(). */
int orbgpu_mappoint_table_upsert(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, const float *world_pos,
                                 const float *normal, const float *min_dist, const float *max_dist, const uint8_t *desc,
                                 const int32_t *n_obs);
/* Ids the table does not know are ignored; *known (optional) = how many it knew. */
int orbgpu_mappoint_table_set_bad(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, int32_t *known);
int orbgpu_mappoint_table_set_observations(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, const int32_t *n_obs,
                                           int32_t *known);
/* Ids of the most recent orbgpu_search_local_points_table / orbgpu_search_by_projection_last_table call that the table
 * had never been told about: list rows (skipped) and key-point associations (treated as held).  Either may be NULL. */
int orbgpu_mappoint_table_last_unknown(const orbgpu_mappoint_table *t, int32_t *list_ids, int32_t *kp_ids);
/* Rows are not recycled by the edits above (a bad point keeps its row, flagged), like the reference, which never frees a
 * MapPoint (Map.cc:76-83, "This only erase the pointer").  This call bounds the table: only the listed ids stay -- typically
 * Map::GetAllMapPoints() plus whatever the current and the last Frame still reference --, rows are renumbered densely in the
 * order given, capacity shrinks to fit; *dropped (optional) = rows released.  Ids dropped here are "unknown" afterwards
 * (skipped as list entries, held as key-point associations: orbgpu_search_local_points_table). */
int orbgpu_mappoint_table_retain(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, int32_t *dropped);
/* One row back to the host (tests, debugging); any output may be NULL. */
int orbgpu_mappoint_table_read(orbgpu_mappoint_table *t, int64_t id, float *world_pos, float *normal, float *min_dist,
                               float *max_dist, uint8_t *desc, int32_t *has_observations, int32_t *bad);

/* A Frame's matcher-side members (orbgpu_frame_view: mvKeysUn, mvuRight, mDescriptors, mGrid, bounds, scale factors)
 * uploaded ONCE and kept on the device: the same frame is the current frame of SearchByProjection(Cur, Last) and of
 * SearchLocalPoints, and the last frame of the next call.  Uploads return synchronised. */
typedef struct orbgpu_frame orbgpu_frame;
int orbgpu_frame_create(int32_t device_id, orbgpu_frame **out);
int orbgpu_frame_destroy(orbgpu_frame *fr);
int orbgpu_frame_upload(orbgpu_frame *fr, const orbgpu_frame_view *host_view);
/* The device pointers of the uploaded frame (valid until the next upload / destroy), for the *_device entry points. */
int orbgpu_frame_device_view(const orbgpu_frame *fr, orbgpu_device_frame_view *view, int32_t *n);

/* Tracking::SearchLocalPoints (Tracking.cc:1447-1497) / ORBmatcher::SearchByProjection(F, vpMapPoints, th)
 * (ORBmatcher.cc:45-129) over the table.  ids [m] (host) = mvpLocalMapPoints[i]->mnId; skip [m] or NULL = the host-only
 * part of Tracking.cc:1474 (mnLastFrameSeen == F.mnId); isBad() comes from the table.
 *   scratch != NULL: drop-in at the ORBmatcher level -- the caller's Frame::isInFrustum has filled the mTrack* members;
 *                    in_view, level, view_cos, proj_x, proj_y, proj_xr [m] are uploaded (the struct's bad / obs_pos /
 *                    desc fields are ignored: they come from the table).  ORBGPU_ELEVEL as orbgpu_search_by_projection.
 *   scratch == NULL: isInFrustum + PredictScale run on the device from Tcw and the camera (as
 *                    orbgpu_search_local_points_device); track_out (optional, HOST arrays [m], any may be NULL) receives
 *                    mbTrackInView etc. so that the caller can IncreaseVisible() (Tracking.cc:1479-1483).
 * kp_ids [n] or NULL (host): mnId of F.mvpMapPoints[j], -1 = none.  kp_to_mp [n] (host, out): position in ids[] of the
 * point key point j holds after the call, -1 none, -2 a point outside the list that has observations (untouched).
 * An id the table does not know yet is NOT an error: LocalMapping publishes a new point to the key frames
 * (LocalMapping.cc:434-440, AddMapPoint) before ComputeDistinctiveDescriptors / UpdateNormalAndDepth have run, and
 * Tracking::UpdateLocalPoints may list it in that window; the reference tolerates the window, so does this call: an
 * unknown list id is a skipped row, an unknown key-point id counts as "held" (-2).  orbgpu_mappoint_table_last_unknown
 * reports how many of each the most recent search call over the table met.
 * With m == 0 (or an empty frame) nothing is searched; kp_to_mp is still translated through the table. */
int orbgpu_search_local_points_table(const orbgpu_frame *fr, orbgpu_mappoint_table *t, int32_t m, const int64_t *ids,
                                     const uint8_t *skip, const orbgpu_mappoint_view *scratch, const float *Tcw, float fx,
                                     float fy, float cx, float cy, float mbf, float log_scale_factor, float cos_limit,
                                     float th, float nnratio, const int64_t *kp_ids, int32_t *kp_to_mp, int32_t *nmatches,
                                     orbgpu_track_scratch *track_out);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1328-1470) over the table: both
 * frames are device-resident; last_ids [last n] (host) = mnId of LastFrame.mvpMapPoints[i] or -1, last_outlier [last n]
 * or NULL = LastFrame.mvbOutlier, cur_kp_ids [cur n] or NULL = mnId of CurrentFrame.mvpMapPoints[j] or -1.
 * kp_to_mp [cur n] (host, out): index i of the last-frame key point whose map point key point j now holds, -1, -2. */
int orbgpu_search_by_projection_last_table(const orbgpu_frame *cur, const float *cur_Tcw, const orbgpu_frame *last,
                                           const float *last_Tcw, orbgpu_mappoint_table *t, const int64_t *last_ids,
                                           const uint8_t *last_outlier, const int64_t *cur_kp_ids, float fx, float fy,
                                           float cx, float cy, float mbf, float mb, float th, int32_t mono,
                                           int32_t check_orientation, int32_t *kp_to_mp, int32_t *nmatches);


/* LastFrame members read by SearchByProjection(CurrentFrame, LastFrame, th, bMono). */
typedef struct {
    int32_t n;
    const uint8_t *has_mp;    /* mvpMapPoints[i] != NULL */
    const uint8_t *outlier;   /* mvbOutlier[i]; may be NULL */
    const uint8_t *obs_pos;   /* pMP->Observations()>0; may be NULL (= all true) */
    const float *world_pos;   /* n x 3 */
    const uint8_t *desc;      /* pMP->GetDescriptor(), n x 32 */
    const int32_t *kp_octave; /* LastFrame.mvKeys[i].octave */
    const float *kp_angle;    /* LastFrame.mvKeysUn[i].angle */
    const float *Tcw;         /* LastFrame.mTcw, 4x4 row-major */
} orbgpu_lastframe_view;

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
 * (ORBmatcher.h:52, ORBmatcher.cc:1328-1470; Tracking::TrackWithMotionModel, Tracking.cc:1169). */
int orbgpu_search_by_projection_last(const orbgpu_frame_view *cur, const float *cur_Tcw, float fx, float fy,
                                     float cx, float cy, float mbf, float mb, const orbgpu_lastframe_view *last,
                                     float th, int32_t mono, int32_t check_orientation, int32_t *kp_to_mp,
                                     int32_t *nmatches, int32_t device_id);

/* pKF->GetMapPointMatches() members read by the relocalisation matcher. */
typedef struct {
    int32_t n;
    const uint8_t *has_mp;        /* vpMPs[i] != NULL */
    const uint8_t *bad;           /* isBad(); may be NULL */
    const uint8_t *already_found; /* sAlreadyFound.count(pMP); may be NULL */
    const float *world_pos;       /* n x 3 */
    const float *min_dist_inv;    /* GetMinDistanceInvariance() (MapPoint.cc:373-377) */
    const float *max_dist_inv;    /* GetMaxDistanceInvariance() (MapPoint.cc:379-383) */
    const float *max_dist;        /* mfMaxDistance, the numerator of MapPoint::PredictScale (MapPoint.cc:385-394) */
    const uint8_t *desc;          /* GetDescriptor(), n x 32 */
    const float *kp_angle;        /* pKF->mvKeysUn[i].angle */
} orbgpu_keyframe_view;

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound,
 * th, ORBdist) (ORBmatcher.h:56, ORBmatcher.cc:1472-1599; Tracking::Relocalization, Tracking.cc:1756,1770).
 * log_scale_factor = CurrentFrame.mfLogScaleFactor.  kp_to_mp in: -1 = free, anything else = occupied
 * (any association hides the key point, :1540-1541); out: index of the key-frame map point. */
int orbgpu_search_by_projection_keyframe(const orbgpu_frame_view *cur, const float *cur_Tcw, float fx, float fy,
                                         float cx, float cy, float log_scale_factor,
                                         const orbgpu_keyframe_view *kf, float th, int32_t orb_dist,
                                         int32_t check_orientation, int32_t *kp_to_mp, int32_t *nmatches,
                                         int32_t device_id);

/* ======================================================================================
 * Vocabulary tree (DBoW2, vendored in the reference): Frame::ComputeBoW (src/Frame.cc:395-402) and the node-wise
 * ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:159-288).
 * ====================================================================================== */
typedef struct orbgpu_vocabulary orbgpu_vocabulary;

/* The tree as TemplatedVocabulary::loadFromTextFile builds it (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1348-1437):
 * node 0 is the root, node i > 0 has parent[i] < i (file order), children of a node are visited in ascending id,
 * leaves are numbered in node order (= word ids).  desc [n_nodes][32], weight [n_nodes] (Node::weight, doubles);
 * weighting 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring 0 L1_NORM .. 5 DOT_PRODUCT (BowVector.h:30-54).
 * The binary / text vocabulary FILE readers stay host code of the caller (the reference's own loader can fill
 * these arrays from m_nodes). */
int orbgpu_vocabulary_create(int32_t k, int32_t L, int32_t n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                             const uint8_t *desc, const double *weight, int32_t weighting, int32_t scoring,
                             int32_t device_id, orbgpu_vocabulary **out);
int orbgpu_vocabulary_destroy(orbgpu_vocabulary *v);
int orbgpu_vocabulary_size(const orbgpu_vocabulary *v, int32_t *n_words);

/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (:1140-1207): per feature the
 * word id, the word's weight and the id of the node `levelsup` levels above the word (:1231-1274; -1 if the word
 * is stopped, i.e. weight <= 0: such a feature enters neither vector); the BowVector as n_bow ascending
 * (bow_ids, bow_vals) pairs, weighted / normalised as the vocabulary's weighting and scoring types prescribe; the
 * FeatureVector as CSR: node fv_nodes[t] (ascending) holds features fv_items[fv_start[t] .. fv_start[t+1]) in
 * ascending feature index.  Capacities: n for every array, n + 1 for fv_start.  bow_* / fv_* may be NULL. */
int orbgpu_bow_transform(orbgpu_vocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, int32_t *word_id,
                         double *word_weight, int32_t *node_id, int32_t *bow_ids, double *bow_vals, int32_t *n_bow,
                         int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_items, int32_t *n_fv);
/* Device-resident: descriptors [batch][cap][32] straight out of the extractor, counts d_n [batch]; outputs
 * [batch][cap].  Asynchronous on hip_stream. */
int orbgpu_bow_transform_batch_device(orbgpu_vocabulary *v, const uint8_t *d_desc, int32_t batch, int32_t cap,
                                      const int32_t *d_n, int32_t levelsup, int32_t *d_word_id, double *d_word_weight,
                                      int32_t *d_node_id, void *hip_stream);

/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (ORBmatcher.cc:159-288):
 * key-frame feature i (valid_kf[i]: has a map point that is not bad) is compared with the frame features under the
 * same vocabulary node (node_kf / node_f: orbgpu_bow_transform's node_id, -1 = not in the feature vector), greedy
 * claims in the reference's visiting order, TH_LOW / ratio test, rotation histogram.  match_f[j] = key-frame
 * feature whose map point frame feature j received, or -1; *nmatches = the return value. */
int orbgpu_search_by_bow(const uint8_t *desc_kf, const float *angle_kf, const uint8_t *valid_kf,
                         const int32_t *node_kf, int32_t n_kf, const uint8_t *desc_f, const float *angle_f,
                         const int32_t *node_f, int32_t n_f, int32_t th_low, float nnratio, int32_t check_orientation,
                         int32_t *match_f, int32_t *nmatches, int32_t device_id);
/* The same for `pairs` independent (key frame, frame) pairs resident on the device; arguments as
 * orbgpu_match_bf_batch_device plus the node arrays [pairs][cap]. */
int orbgpu_search_by_bow_batch_device(orbgpu_matcher *m, int32_t pairs, int32_t cap, const uint8_t *d_desc_a,
                                      const void *d_angle_a, const uint8_t *d_valid_a, const int32_t *d_node_a,
                                      const int32_t *d_na, const uint8_t *d_desc_b, const void *d_angle_b,
                                      const int32_t *d_node_b, const int32_t *d_nb, size_t angle_stride, int32_t th_low,
                                      float nnratio, int32_t check_orientation, int32_t *d_match_b,
                                      int32_t *d_nmatches, void *hip_stream);

/* ---- the background-thread matchers (LocalMapping / LoopClosing), SURVEY.md M6 ----------------------------- */

/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (ORBmatcher.cc:522-655;
 * LoopClosing::ComputeSim3, LoopClosing.cc:266).  valid1 / valid2: the key point has a map point that is not bad.
 * match12[i1] = key point of key frame 2 whose map point vpMatches12[i1] receives, or -1. */
int orbgpu_search_by_bow_keyframes(const uint8_t *desc1, const float *angle1, const uint8_t *valid1,
                                   const int32_t *node1, int32_t n1, const uint8_t *desc2, const float *angle2,
                                   const uint8_t *valid2, const int32_t *node2, int32_t n2, float nnratio,
                                   int32_t check_orientation, int32_t *match12, int32_t *nmatches, int32_t device_id);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:405-520;
 * Tracking::MonocularInitialization, Tracking.cc:877 -- monocular bootstrap only).  f1: kp_octave, kp_angle, desc of
 * the initial frame (its grid is not read); f2: the current frame; prev_matched [f1->n][2] in/out.  The Hamming
 * distances of every window candidate are computed on the device; the steal-if-strictly-closer bookkeeping between
 * rows (:441-470) is sequential and runs on the host over those lists, statement by statement. */
int orbgpu_search_for_initialization(const orbgpu_frame_view *f1, const orbgpu_frame_view *f2, float *prev_matched,
                                     int32_t window_size, float nnratio, int32_t check_orientation,
                                     int32_t *matches12, int32_t *nmatches, int32_t device_id);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORBmatcher.cc:657-823, epipolar
 * test :140-157; LocalMapping::CreateNewMapPoints, LocalMapping.cc:268).  kf1 / kf2: the key frames as frame views
 * (kp_x / kp_y = mvKeysUn, u_right = mvuRight, kp_octave, kp_angle, desc; grids are not read); has_mp*: the key
 * point already has a map point; node*: orbgpu_bow_transform's node_id; F12 row-major 3x3; (ex, ey) the epipole
 * in image 2 (:664-670, from the poses); level_sigma2_2 = pKF2->mvLevelSigma2.  match12[i1] = i2 or -1; the pairs
 * vector of the reference is its non-negative entries in order. */
int orbgpu_search_for_triangulation(const orbgpu_frame_view *kf1, const uint8_t *has_mp1, const int32_t *node1,
                                    const orbgpu_frame_view *kf2, const uint8_t *has_mp2, const int32_t *node2,
                                    const float *F12, float ex, float ey, const float *level_sigma2_2,
                                    int32_t only_stereo, int32_t check_orientation, int32_t *match12,
                                    int32_t *nmatches, int32_t device_id);

/* ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) (ORBmatcher.cc:825-975;
 * LocalMapping::SearchInNeighbors, LocalMapping.cc:489, 514): the candidate phase.  pts->bad[i] = !pMP ||
 * pMP->isBad() || pMP->IsInKeyFrame(pKF).  best_idx[i] = the key point map point i fuses with (least distance in its
 * window after the level and reprojection-error gates, <= TH_LOW) or -1.  The map edits of :948-968 (Replace /
 * AddObservation / AddMapPoint) mutate the pointer graph and stay with the caller, applied in index order.
 * ORBGPU_ELEVEL if a predicted level falls outside [0, nlevels). */
int orbgpu_fuse(const orbgpu_frame_view *kf, const float *Tcw, float fx, float fy, float cx, float cy, float bf,
                float log_scale_factor, const orbgpu_points_view *pts, float th, const float *inv_level_sigma2,
                int32_t *best_idx, int32_t *n_candidates, int32_t device_id);
/* ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:977-1100;
 * LoopClosing::SearchAndFuse, LoopClosing.cc:597): candidate phase.  pts->bad[i] = isBad() || already in pKF. */
int orbgpu_fuse_sim3(const orbgpu_frame_view *kf, const float *Scw, float fx, float fy, float cx, float cy,
                     float log_scale_factor, const orbgpu_points_view *pts, float th, int32_t *best_idx,
                     int32_t *n_candidates, int32_t device_id);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1102-1326;
 * LoopClosing::ComputeSim3, LoopClosing.cc:324).  pts1 / pts2: one row per key point of the key frame (bad[i] = no
 * map point or isBad(); normal is not read); already1 / already2 = vbAlreadyMatched1 / 2 (:1133-1144) or NULL;
 * T1w / T2w: the key-frame poses, R12 row-major 3x3.  match12[i1] = i2 where both directions agree, else -1. */
int orbgpu_search_by_sim3(const orbgpu_frame_view *kf1, const orbgpu_frame_view *kf2, const float *T1w, const float *T2w,
                          float s12, const float *R12, const float *t12, float fx, float fy, float cx, float cy,
                          float log_sf1, float log_sf2, const orbgpu_points_view *pts1, const uint8_t *already1,
                          const orbgpu_points_view *pts2, const uint8_t *already2, float th, int32_t *match12,
                          int32_t *nfound, int32_t device_id);

/* ======================================================================================
 * PointCloudMapping  (reference include/PointCloudMap.h:41-88, src/PointCloudMap.cc)
 * ====================================================================================== */

/* pcl::PointXYZRGBA payload: xyz + packed colour (b | g<<8 | r<<16 | a<<24). 16 B. */
typedef struct {
    float x, y, z;
    uint32_t rgba;
} orbgpu_point_xyzrgba;

typedef struct orbgpu_cloud orbgpu_cloud;

/* PointCloudMapping::PointCloudMapping(resolution, loopCloser) (PointCloudMap.h:46,
 * PointCloudMap.cc:36-57): voxel leaf = (float)resolution on all axes. */
int orbgpu_cloud_create(double resolution, int32_t device_id, orbgpu_cloud **out);
int orbgpu_cloud_destroy(orbgpu_cloud *h);

/* One key-frame step of PointCloudMapping::viewer (PointCloudMap.cc:204-262, no-loop branch):
 * convertToPointCloud(kf) -> transformPointCloud(Twc) -> globalMap += -> voxel.filter(globalMap).
 * depth: float32 metres (after DepthMapFactor), rgb: 8UC3 in the byte order of KeyFrame::mImRGB,
 * Tcw: kf->GetPose() 4x4 row-major float.  depth_stride in floats, rgb_stride in bytes. */
int orbgpu_cloud_insert(orbgpu_cloud *h, const float *depth, size_t depth_stride, const uint8_t *rgb,
                        size_t rgb_stride, int32_t width, int32_t height, float fx, float fy, float cx, float cy,
                        const float *Tcw);
/* Same step with the key frame's images already resident in device memory (e.g. the frame the extractor just
 * processed): nothing but the pose crosses PCIe.  Runs on the handle's stream and returns when the map is
 * updated. */
int orbgpu_cloud_insert_device(orbgpu_cloud *h, const float *d_depth, size_t depth_stride, const uint8_t *d_rgb,
                               size_t rgb_stride, int32_t width, int32_t height, float fx, float fy, float cx,
                               float cy, const float *Tcw);
/* Loop-closure branch (PointCloudMap.cc:217-243): drop the map, re-generate every key-frame cloud
 * with its (new) pose, filter once. Arrays of n key-frames of one size. */
int orbgpu_cloud_rebuild(orbgpu_cloud *h, int32_t n, const float *const *depth, size_t depth_stride,
                         const uint8_t *const *rgb, size_t rgb_stride, int32_t width, int32_t height, float fx,
                         float fy, float cx, float cy, const float *const *Tcw);
/* The shutdown pass of PointCloudMapping::viewer (PointCloudMap.cc:270-282): clear the map, then for every key
 * frame generatePointCloud + voxel.filter of THAT cloud alone + `globalMap += `.  After it the map is a
 * concatenation of per-key-frame filtered clouds (what optimized_pointcloud.pcd holds before the outlier filter). */
int orbgpu_cloud_clear(orbgpu_cloud *h);
int orbgpu_cloud_append_filtered(orbgpu_cloud *h, const float *depth, size_t depth_stride, const uint8_t *rgb,
                                 size_t rgb_stride, int32_t width, int32_t height, float fx, float fy, float cx,
                                 float cy, const float *Tcw);
/* The outlier filter of the shutdown pass (PointCloudMap.cc:46-47: sor.setMeanK(50), sor.setStddevMulThresh(1.0);
 * :283-285: sor.setInputCloud(globalMap); sor.filter(*tmp); globalMap->swap(*tmp)), applied to the handle's map in
 * place; the points that stay keep their order.  pcl::StatisticalOutlierRemoval is restated from PCL 1.7
 * (filters/impl/statistical_outlier_removal.hpp -- PCL is not vendored by the reference: unpinned): mean distance of
 * every point to its mean_k nearest neighbours (exact search, float squared distances as FLANN's L2_Simple, summed
 * in double), a point goes when that exceeds mean + stddev_mul * stddev of all of them.  mean_k <= 63; needs more
 * than mean_k points (PCL reads past the neighbour list otherwise).  removed may be NULL. */
int orbgpu_cloud_remove_outliers(orbgpu_cloud *h, int32_t mean_k, double stddev_mul, int64_t *removed);
int orbgpu_cloud_size(orbgpu_cloud *h, int64_t *n);
/* Global map in ascending voxel-index order (pcl::VoxelGrid output order). */
int orbgpu_cloud_download(orbgpu_cloud *h, orbgpu_point_xyzrgba *out, int64_t cap, int64_t *n);
/* 1 if the last filter hit PCL's int32 voxel-index overflow and returned its input unfiltered. */
int orbgpu_cloud_last_overflow(orbgpu_cloud *h, int32_t *overflow);
/* Which implementation served the last insert / rebuild: 1 = merge of the new points into the sorted resident
 * map, 2 = general path (sort of everything), 3 = merge attempted, its precondition check failed, redone by the
 * general path.  Results are identical; this is for tests and measurements. */
int orbgpu_cloud_last_path(orbgpu_cloud *h, int32_t *path);
/* Measurement aid: HIP events on the handle's stream around the kernels of an insert (first launch to last,
 * uploads and the final state read-back excluded); last_insert_ms fails if no profiled insert happened. */
int orbgpu_cloud_set_profiling(orbgpu_cloud *h, int32_t enable);
int orbgpu_cloud_last_insert_ms(orbgpu_cloud *h, float *ms);

/* Stateless stages (host pointers), for parity tests and other callers:
 * convertToPointCloud (PointCloudMap.cc:112-138) with optional pose transform (Tcw != NULL:
 * generatePointCloud, :78-110).  out cap >= ceil(h/3)*ceil(w/3). */
int orbgpu_backproject(const float *depth, size_t depth_stride, const uint8_t *rgb, size_t rgb_stride,
                       int32_t width, int32_t height, float fx, float fy, float cx, float cy, const float *Tcw,
                       orbgpu_point_xyzrgba *out, int64_t cap, int64_t *n, int32_t device_id);
/* pcl::VoxelGrid<PointXYZRGBA>::filter with leaf = (float)resolution (PointCloudMap.cc:41, 240-243). */
int orbgpu_voxel_filter(const orbgpu_point_xyzrgba *in, int64_t n, double resolution, orbgpu_point_xyzrgba *out,
                        int64_t cap, int64_t *n_out, int32_t *overflow, int32_t device_id);

/* pcl::StatisticalOutlierRemoval<PointXYZRGBA>::filter on host points (see orbgpu_cloud_remove_outliers).  cap >= n;
 * mean_dist ([n], may be NULL) receives every point's mean neighbour distance (0 for non-finite points, which stay). */
int orbgpu_statistical_outlier_removal(const orbgpu_point_xyzrgba *in, int64_t n, int32_t mean_k, double stddev_mul,
                                       orbgpu_point_xyzrgba *out, int64_t cap, int64_t *n_out, float *mean_dist,
                                       int32_t device_id);

/* ======================================================================================
 * On-disk formats of the end-of-run artefacts: host serialisers, byte for byte.
 * ====================================================================================== */
/* Map::_WriteMapPoint (Map.cc:123-130): id (u64) + world position (3 x f32) = 20 bytes. */
size_t orbgpu_mappoint_record_bytes(void);
int orbgpu_write_mappoint_record(uint64_t id, const float *world_pos, uint8_t *out, size_t cap, size_t *written);
/* Map::_WriteKeyFrame (Map.cc:133-183): id (u64), time stamp (f64), translation (3 x f32), rotation as the Eigen
 * quaternion x y z w of Converter::toQuaternion (Converter.cc:137-149, 4 x f32), N (i32), then per key point
 * x, y, size, angle, response (f32), octave (i32), the 32 descriptor bytes and the index of its map point in the
 * file's map-point list (u64, ULONG_MAX = none): 64 bytes per key point.  keys = mvKeys (orbgpu_keypoint records
 * as the extractor writes them). */
size_t orbgpu_keyframe_record_bytes(int32_t n_features);
int orbgpu_write_keyframe_record(uint64_t id, double timestamp, const float *Tcw, int32_t n,
                                 const orbgpu_keypoint *keys, const uint8_t *desc, const uint64_t *mappoint_index,
                                 uint8_t *out, size_t cap, size_t *written);
/* pcl::io::savePCDFileBinary of a PointCloud<PointXYZRGBA> (PointCloudMap.cc:287; PCL 1.7 PCDWriter, header text
 * as recalled -- unpinned): header, then n packed (x, y, z, rgba) records of 16 bytes. */
int orbgpu_pcd_binary_header(int64_t n_points, char *buf, size_t cap, size_t *len);
int orbgpu_write_pcd_binary(const char *path, const orbgpu_point_xyzrgba *points, int64_t n);
/* pcl::io::savePCDFileBinary("optimized_pointcloud.pcd", *globalMap) (PointCloudMap.cc:287): the handle's map as it
 * is (call orbgpu_cloud_remove_outliers first for the reference's shutdown sequence). */
int orbgpu_cloud_save_pcd(orbgpu_cloud *h, const char *path);

#ifdef __cplusplus
}
#endif
#endif /* ORBGPU_H */
