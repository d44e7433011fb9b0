/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's per-frame hot path
 * (carry4985/ORB_SLAM2_MAP: src/ORBextractor.cc, src/ORBmatcher.cc, src/Frame.cc,
 * src/PointCloudMap.cc) together with the OpenCV-2.4 / PCL-1.7 arithmetic those
 * files delegate to.  Every function cites the reference
 * file:line it follows.
 *
 * PARITY STATUS: *parity unpinned* at the third-party boundary.  The reference ships no
 * tests, golden vectors or fixtures, and cannot be built here (OpenCV 2.4 / PCL 1.7 are
 * absent), so this oracle is pinned only by (a) constants the reference's sources imply
 * (pyramid sizes, per-level quotas, umax, blur taps, pattern checksum) and (b) independent
 * definitional cross-checks in tests/ (brute-force FAST, big-int Hamming, float64 blur).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (liborbgpu.so) never links or calls it.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_LEVELS 16
#define ORA_EDGE 19 /* EDGE_THRESHOLD, ORBextractor.cc:74 */

/* cv::KeyPoint field order (28 B), ORBextractor.h:59-61 output element. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} ora_keypoint;

/* FAST / quadtree candidate: integer level coordinates + integer response. */
typedef struct {
    int32_t x, y, response;
} ora_corner;

typedef struct ora_extractor ora_extractor;

/* CPU-baseline aid: wall-clock seconds the extractor spent per stage (accumulated over ora_extract calls). */
enum { ORA_STAGE_PYRAMID = 0, ORA_STAGE_FAST, ORA_STAGE_QUADTREE, ORA_STAGE_ORIENT, ORA_STAGE_BLUR,
       ORA_STAGE_DESCRIBE, ORA_STAGE_COUNT };
void ora_extractor_stage_seconds(ora_extractor *e, double out[ORA_STAGE_COUNT], int reset);

/* ---- E0: constructor tables, ORBextractor.cc:410-470 ---- */
ora_extractor *ora_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                                    int min_th_fast);
void ora_extractor_destroy(ora_extractor *e);
int ora_get_levels(const ora_extractor *e);
const float *ora_get_scale_factors(const ora_extractor *e);
const float *ora_get_inv_scale_factors(const ora_extractor *e);
const float *ora_get_sigma2(const ora_extractor *e);
const float *ora_get_inv_sigma2(const ora_extractor *e);
const int *ora_get_quotas(const ora_extractor *e);
const int *ora_get_umax(const ora_extractor *e); /* 16 entries */
const int8_t *ora_get_pattern(void);             /* 1024 entries */

/* ---- E8: operator(), ORBextractor.cc:1043-1105. Returns number of keypoints (<= cap) or
 * -1 if cap is too small.  desc is cap x 32 bytes. ---- */
int ora_extract(ora_extractor *e, const uint8_t *gray, int w, int h, size_t stride, ora_keypoint *kps,
                uint8_t *desc, int cap);

/* ---- stage introspection of the LAST ora_extract call (for stage-by-stage GPU parity) ---- */
/* E1/E9: padded pyramid level (border 19). *pitch = bytes per row; returned pointer is the
 * padded origin, image origin is at +19*pitch+19. */
const uint8_t *ora_pyramid_level(const ora_extractor *e, int level, int *w, int *h, int *pitch);
/* E6: blurred level (unpadded, pitch == w). NULL if the level had no keypoints (not blurred). */
const uint8_t *ora_blurred_level(const ora_extractor *e, int level, int *w, int *h);
/* E2: FAST candidates of a level in vToDistributeKeys order (coords relative to minBorder). */
int ora_level_candidates(const ora_extractor *e, int level, const ora_corner **out);
/* E3: distributed keypoints of a level in output (list) order (coords relative to minBorder). */
int ora_level_selected(const ora_extractor *e, int level, const ora_corner **out);

/* ---- stand-alone stage functions ---- */
/* A2: cv::resize(src,dst,INTER_LINEAR) for 8UC1 (OpenCV 2.4 fixed-point path). */
void ora_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw, int dh,
                          size_t dstride);
/* A6: copyMakeBorder(BORDER_REFLECT_101); dst is (w+2b) x (h+2b). */
void ora_border_reflect101_u8(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride,
                              int border);
/* A3: GaussianBlur(7x7, sigma 2, REFLECT_101) for 8UC1 (OpenCV 2.4 8-bit fixed point). */
void ora_gauss7_u8(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride);
/* A4: cv::FAST(img, kps, threshold, nonmaxSuppression=true) on a sub-image. Returns count
 * (writes at most cap). Output order row-major. */
int ora_fast9_16(const uint8_t *img, int w, int h, size_t stride, int threshold, ora_corner *out, int cap);
/* E3: DistributeOctTree, ORBextractor.cc:539-763. keys in vToDistributeKeys order. */
int ora_distribute_octtree(const ora_corner *keys, int nkeys, int minX, int maxX, int minY, int maxY, int N,
                           ora_corner *out, int cap);
/* E5: IC_Angle, ORBextractor.cc:77-104 (+ fastAtan2, A5). center points at the keypoint pixel. */
float ora_ic_angle(const uint8_t *center, int step, const int *umax);
float ora_fast_atan2(float y, float x);
/* E7: computeOrbDescriptor, ORBextractor.cc:108-147. */
void ora_orb_descriptor(const uint8_t *center, int step, float angle_deg, uint8_t desc[32]);
/* :112-113 resolve to std::cos(float) / std::sin(float) (`using namespace std`, :66-67) = this host's cosf / sinf:
 * ORA_TRIG_LIBM_FLOAT (default).  ORA_TRIG_ROUNDED_DOUBLE = (float)cos((double)angle), the correctly rounded value. */
enum { ORA_TRIG_LIBM_FLOAT = 0, ORA_TRIG_ROUNDED_DOUBLE = 1 };
void ora_set_trig_mode(int mode);
int ora_get_trig_mode(void);
void ora_descriptor_trig(float angle_rad, float *a, float *b);
void ora_descriptor_trig_array(const float *angle_rad, int n, float *a, float *b);

/* ---- M0: ORBmatcher::DescriptorDistance, ORBmatcher.cc:1647-1663 ---- */
int ora_descriptor_distance(const uint8_t *a, const uint8_t *b);

/* ---- M1 constants, ORBmatcher.cc:37-39 ---- */
#define ORA_TH_HIGH 100
#define ORA_TH_LOW 50
#define ORA_HISTO_LENGTH 30

/* ---- M4: brute-force matcher = ORBmatcher::SearchByBoW(KF,F) (ORBmatcher.cc:159-288) with
 * every feature in one vocabulary node.  For each valid A row (in index order): best/second
 * Hamming over not-yet-claimed B rows, accept iff best<=th_low && best < ratio*second, claim.
 * Optional rotation-histogram consistency (angles in degrees).  match_b[j] = index of the A
 * row matched to B row j, or -1.  Returns nmatches.  valid_a may be NULL (all valid). */
int ora_match_bf(const uint8_t *desc_a, const float *angle_a, const uint8_t *valid_a, int na,
                 const uint8_t *desc_b, const float *angle_b, int nb, int th_low, float nnratio,
                 int check_orientation, int32_t *match_b);

/* ---- M7: Frame grid (Frame.cc:230-245, 327-392; Frame.h:37-38) ---- */
#define ORA_GRID_COLS 64
#define ORA_GRID_ROWS 48

typedef struct {
    int n;                /* Frame::N */
    const float *kp_x;    /* mvKeysUn[i].pt.x */
    const float *kp_y;    /* mvKeysUn[i].pt.y */
    const int32_t *kp_octave;
    const float *kp_angle;
    const float *u_right; /* mvuRight */
    const uint8_t *desc;  /* n x 32 */
    float min_x, max_x, min_y, max_y; /* mnMinX.. */
    float grid_inv_w, grid_inv_h;     /* mfGridElementWidthInv / HeightInv */
    const float *scale_factors;       /* mvScaleFactors */
    int nlevels;
    /* CSR of mGrid[ix][iy] in (ix*ROWS+iy) order, items in insertion order */
    const int32_t *cell_start; /* COLS*ROWS+1 */
    const int32_t *cell_items;
} ora_frame_view;

/* AssignFeaturesToGrid: fills cell_start (COLS*ROWS+1) and cell_items (n). */
void ora_assign_features_to_grid(int n, const float *kp_x, const float *kp_y, float min_x, float min_y,
                                 float inv_w, float inv_h, int32_t *cell_start, int32_t *cell_items);
/* GetFeaturesInArea, Frame.cc:327-380. Returns count written to out (cap >= n). */
int ora_get_features_in_area(const ora_frame_view *f, float x, float y, float r, int min_level, int max_level,
                             int32_t *out);

/* ---- M2: SearchByProjection(Frame&, vector<MapPoint*>&, th), ORBmatcher.cc:45-129 ---- */
typedef struct {
    int m;
    const uint8_t *in_view;    /* mbTrackInView */
    const uint8_t *bad;        /* isBad() */
    const uint8_t *obs_pos;    /* Observations()>0 */
    const int32_t *level;      /* mnTrackScaleLevel */
    const float *view_cos;     /* mTrackViewCos */
    const float *proj_x, *proj_y, *proj_xr;
    const uint8_t *desc;       /* m x 32 */
} ora_mappoint_view;

/* kp_to_mp: in/out, size f->n.  In: >=0 index of an already-associated map point (its
 * obs_pos decides whether the keypoint is skipped), -1 = free, -2 = occupied by a map point
 * outside the view with Observations()>0.  Out: assignments written like F.mvpMapPoints.
 * Returns nmatches, or -1 if a predicted level is outside [0,nlevels) (H5). */
int ora_search_by_projection(const ora_frame_view *f, const ora_mappoint_view *mp, float th, float nnratio,
                             int32_t *kp_to_mp);

/* Tracking::SearchLocalPoints (Tracking.cc:1447-1497) for a fresh frame: isInFrustum (cos_limit 0.5) over the listed
 * points that are not skipped, then SearchByProjection(F, vpMapPoints, th) with nnratio.  The m-entry scratch arrays
 * receive the mTrack* members; *n_level_out counts predicted levels outside [0, nlevels) (left out of the view). */
int ora_search_local_points(const ora_frame_view *f, const float *Tcw, float fx, float fy, float cx, float cy, float mbf,
                            int m, const float *world_pos, const float *normal, const float *min_dist,
                            const float *max_dist, const uint8_t *skip, const uint8_t *obs_pos, const uint8_t *desc,
                            float log_scale_factor, float cos_limit, float th, float nnratio, uint8_t *in_view,
                            float *proj_x, float *proj_y, float *proj_xr, int32_t *level, float *view_cos,
                            int32_t *kp_to_mp, int *n_level_out);

/* ---- M3: SearchByProjection(CurrentFrame, LastFrame, th, bMono), ORBmatcher.cc:1328-1470 ---- */
typedef struct {
    int n;                      /* LastFrame.N */
    const uint8_t *has_mp;      /* mvpMapPoints[i] != NULL */
    const uint8_t *outlier;     /* mvbOutlier[i] */
    const uint8_t *obs_pos;     /* pMP->Observations()>0 */
    const float *world_pos;     /* n x 3, pMP->GetWorldPos() */
    const uint8_t *desc;        /* n x 32, pMP->GetDescriptor() */
    const int32_t *kp_octave;   /* LastFrame.mvKeys[i].octave */
    const float *kp_angle;      /* LastFrame.mvKeysUn[i].angle */
    const float *Tcw;           /* LastFrame.mTcw 4x4 row-major */
} ora_lastframe_view;

int ora_search_by_projection_last(const ora_frame_view *cur, const float *cur_Tcw, float fx, float fy, float cx,
                                  float cy, float mbf, float mb, const ora_lastframe_view *last, float th,
                                  int mono, int check_orientation, int32_t *kp_to_mp);

/* ---- M5a: SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound,
 * th, ORBdist), ORBmatcher.cc:1472-1599 (Tracking::Relocalization, Tracking.cc:1756,1770) ---- */
typedef struct {
    int n;                        /* pKF->GetMapPointMatches().size() */
    const uint8_t *has_mp;        /* vpMPs[i] != NULL */
    const uint8_t *bad;           /* isBad() */
    const uint8_t *already_found; /* sAlreadyFound.count(pMP) */
    const float *world_pos;       /* n x 3 */
    const float *min_dist_inv;    /* GetMinDistanceInvariance() = 0.8f * mfMinDistance */
    const float *max_dist_inv;    /* GetMaxDistanceInvariance() = 1.2f * mfMaxDistance */
    const float *max_dist;        /* mfMaxDistance, used by PredictScale */
    const uint8_t *desc;          /* n x 32 */
    const float *kp_angle;        /* pKF->mvKeysUn[i].angle */
} ora_keyframe_view;

/* kp_to_mp in: -1 = CurrentFrame.mvpMapPoints[j] is NULL, anything else = occupied; out: index of the
 * key-frame map point written there.  Returns nmatches, -1 on a level outside [0,nlevels). */
int ora_search_by_projection_keyframe(const ora_frame_view *cur, const float *cur_Tcw, float fx, float fy, float cx,
                                      float cy, float log_scale_factor, const ora_keyframe_view *kf, float th,
                                      int orb_dist, int check_orientation, int32_t *kp_to_mp);

/* ---- M5b: ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), ORBmatcher.cc:290-403 ---- */
typedef struct {
    int32_t m;
    const uint8_t *bad;      /* isBad() */
    const float *world_pos;  /* m x 3 */
    const float *normal;     /* m x 3 */
    const float *min_dist;   /* mfMinDistance */
    const float *max_dist;   /* mfMaxDistance */
    const uint8_t *desc;     /* m x 32 */
} ora_points_view;
void ora_sim3_decompose(const float *Scw, float *T34, float *Ow);
int ora_search_by_projection_sim3(const ora_frame_view *kf, const float *Scw, float fx, float fy, float cx, float cy,
                                  float log_scale_factor, const ora_points_view *pts, int th, int32_t *kp_to_mp);

/* ---- MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:242-307 ---- */
int ora_distinctive_descriptor(int n, const uint8_t *desc);

/* ---- M8: Frame::isInFrustum, Frame.cc:269-325 (+ MapPoint::PredictScale, MapPoint.cc:385-394) ---- */
int ora_is_in_frustum(const float *Tcw, float fx, float fy, float cx, float cy, float mbf, float min_x,
                      float max_x, float min_y, float max_y, const float *P, const float *normal, float min_dist,
                      float max_dist, float log_scale_factor, int nlevels_unused, float cos_limit, float *proj_x,
                      float *proj_y, float *proj_xr, int32_t *level, float *view_cos);

/* ---- cv::undistortPoints as used by Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cc:404-468) ---- */
void ora_undistort_points(int n, const float *xy_in, float fx, float fy, float cx, float cy, const float *dist,
                          float *xy_out);

/* ---- ComputeStereoFromRGBD, Frame.cc:641-662 ---- */
void ora_compute_stereo_from_rgbd(int n, const float *kp_x, const float *kp_y, const float *kpun_x,
                                  const float *depth, size_t depth_stride_elems, float mbf, float *u_right,
                                  float *kp_depth);

/* ---- P1..P3 point cloud ---- */
typedef struct {
    float x, y, z;
    uint32_t rgba; /* PCL PointXYZRGBA packing: b | g<<8 | r<<16 | a<<24 */
} ora_point;

/* P1: convertToPointCloud, PointCloudMap.cc:112-138. Returns point count (cap >= ceil(h/3)*ceil(w/3)). */
int ora_backproject(const float *depth, size_t depth_stride_elems, const uint8_t *rgb, size_t rgb_stride, int w,
                    int h, float fx, float fy, float cx, float cy, ora_point *out);
/* P2: Converter::toSE3Quat + Isometry3d::inverse (Converter.cc:37-47, PointCloudMap.cc:103-105):
 * Tcw (4x4 float row-major) -> Twc as R (row-major 3x3 double) and t (double). */
void ora_pose_inverse(const float *Tcw, double R[9], double t[3]);
/* P2: pcl::transformPointCloud with a double transform (A8). In place allowed. */
void ora_transform_points(const ora_point *in, int n, const double R[9], const double t[3], ora_point *out);
/* P3: pcl::VoxelGrid<PointXYZRGBA>::filter (A7), leaf = (float)resolution on all axes.
 * Output ascending voxel index. Returns output count; if the index space would overflow
 * int32 the input is returned unfiltered (PCL behaviour) and *overflow is set. */
int ora_voxel_filter(const ora_point *in, int n, float leaf, ora_point *out, int *overflow);
/* pcl::StatisticalOutlierRemoval::filter (PointCloudMap.cc:46-47, 283-285; PCL 1.7 applyFilterIndices, brute-force
 * neighbours).  Returns the number of points kept (-1: bad arguments or fewer than mean_k + 1 finite points);
 * mean_dist[n] (may be NULL) receives every point's mean neighbour distance. */
int ora_statistical_outlier_removal(const ora_point *in, int n, int mean_k, double stddev_mul, ora_point *out,
                                    float *mean_dist);

/* ---- M6: background-thread matchers (orb_oracle_match.c); see the function comments there ---- */
int ora_fuse(const ora_frame_view *kf, const float *Tcw, float fx, float fy, float cx, float cy, float bf,
             float log_scale_factor, const ora_points_view *pts, float th, const float *inv_level_sigma2,
             int32_t *best_idx);
int ora_fuse_sim3(const ora_frame_view *kf, const float *Scw, float fx, float fy, float cx, float cy,
                  float log_scale_factor, const ora_points_view *pts, float th, int32_t *best_idx);
int ora_search_by_sim3(const ora_frame_view *kf1, const ora_frame_view *kf2, const float *T1w, const float *T2w,
                       float s12, const float *R12, const float *t12, float fx, float fy, float cx, float cy,
                       float log_sf1, float log_sf2, const ora_points_view *pts1, const uint8_t *already1,
                       const ora_points_view *pts2, const uint8_t *already2, float th, int32_t *match12);
int ora_search_for_triangulation(const ora_frame_view *kf1, const uint8_t *has_mp1, int n_fv1, const int32_t *fv_nodes1,
                                 const int32_t *fv_start1, const int32_t *fv_items1, const ora_frame_view *kf2,
                                 const uint8_t *has_mp2, int n_fv2, const int32_t *fv_nodes2,
                                 const int32_t *fv_start2, const int32_t *fv_items2, const float *F12, float ex,
                                 float ey, const float *level_sigma2_2, int only_stereo, int check_orientation,
                                 int32_t *match12);
int ora_search_by_bow_kf(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1, int n_fv1,
                         const int32_t *fv_nodes1, const int32_t *fv_start1, const int32_t *fv_items1,
                         const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2, int n_fv2,
                         const int32_t *fv_nodes2, const int32_t *fv_start2, const int32_t *fv_items2, float nnratio,
                         int check_orientation, int32_t *match12);

int ora_search_for_initialization(const ora_frame_view *f1, const ora_frame_view *f2, float *prev_matched,
                                  int window_size, float nnratio, int check_orientation, int32_t *matches12);

/* ---- BoW: vocabulary tree transform + node-wise matcher (orb_oracle_bow.c) ----
 * Nodes in the order TemplatedVocabulary::loadFromTextFile creates them (node 0 = root, a node after its parent);
 * children of a node in ascending id; leaves numbered in node order = word ids.  weighting: 0 TF_IDF, 1 TF, 2 IDF,
 * 3 BINARY; scoring: 0 L1_NORM .. 5 DOT_PRODUCT (BowVector.h:30-54). */
typedef struct ora_vocabulary ora_vocabulary;
ora_vocabulary *ora_vocabulary_create(int k, int L, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                                      const uint8_t *desc, const double *weight, int weighting, int scoring);
void ora_vocabulary_destroy(ora_vocabulary *v);
int ora_vocabulary_words(const ora_vocabulary *v);
/* TemplatedVocabulary::transform(feature, id, weight, nid, levelsup), TemplatedVocabulary.h:1231-1274 */
void ora_vocabulary_transform_feature(const ora_vocabulary *v, const uint8_t *feature, int levelsup, int32_t *word_id,
                                      double *weight, int32_t *nid);
/* TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup), :1140-1207 (Frame::ComputeBoW,
 * Frame.cc:395-402 with levelsup 4).  bow_* / fv_items capacity n, fv_nodes n, fv_start n + 1. */
int ora_bow_transform(const ora_vocabulary *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                      double *word_weight, int32_t *node_id, int32_t *bow_ids, double *bow_vals, int32_t *n_bow,
                      int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_items, int32_t *n_fv);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches), ORBmatcher.cc:159-288 */
int ora_search_by_bow(const uint8_t *desc_kf, const float *angle_kf, const uint8_t *valid_kf, int n_fv_kf,
                      const int32_t *fv_nodes_kf, const int32_t *fv_start_kf, const int32_t *fv_items_kf,
                      const uint8_t *desc_f, const float *angle_f, int nf, int n_fv_f, const int32_t *fv_nodes_f,
                      const int32_t *fv_start_f, const int32_t *fv_items_f, int th_low, float nnratio,
                      int check_orientation, int32_t *match_f);

#ifdef __cplusplus
}
#endif
#endif
