// orbgpu_shim.hpp -- C++ host side above the C ABI: re-creates the reference's three hot-path
// interfaces (ORB_SLAM2::ORBextractor, ORBmatcher::SearchByProjection x2 + DescriptorDistance,
// PointCloudMapping) on top of liborbgpu.so.
//
// The reference's own classes depend on OpenCV 2.4 / PCL 1.7 (absent from the build image), so
// everything that touches Frame / MapPoint / KeyFrame / cv::Mat is a template on those types: inside
// the reference tree the templates bind to the real classes (INTEGRATION.md shows the three edits);
// in this repository tests/shim_test.cpp binds them to small stand-ins with the same member names.
// Field names, argument meaning and return values follow the reference (file:line cited per item).
#pragma once

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "orbgpu.h"

namespace orbgpu_shim {

inline void check(int status, const char *what)
{
    if (status != ORBGPU_OK)
        throw std::runtime_error(std::string(what) + ": " + orbgpu_last_error_string());
}

// --------------------------------------------------------------------------------------------
// ORBextractor (reference include/ORBextractor.h:46-110)
// KeyPointT must be layout-compatible with cv::KeyPoint (pt.x, pt.y, size, angle, response,
// octave, class_id = 28 bytes), which orbgpu_keypoint mirrors.
// --------------------------------------------------------------------------------------------
template <typename KeyPointT> class ORBextractorT {
  public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractorT(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device_id = 0)
    {
        static_assert(sizeof(KeyPointT) == sizeof(orbgpu_keypoint), "KeyPointT must match cv::KeyPoint");
        orbgpu_extractor_params p{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id, 1};
        check(orbgpu_extractor_create(&p, &h_), "ORBextractor");
        nlevels_ = nlevels;
        scaleFactor_ = scaleFactor;
        mvScaleFactor.resize(nlevels);
        mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        check(orbgpu_extractor_get_scale_factors(h_, mvScaleFactor.data()), "scale factors");
        check(orbgpu_extractor_get_inv_scale_factors(h_, mvInvScaleFactor.data()), "scale factors");
        check(orbgpu_extractor_get_sigma2(h_, mvLevelSigma2.data()), "scale factors");
        check(orbgpu_extractor_get_inv_sigma2(h_, mvInvLevelSigma2.data()), "scale factors");
    }
    ~ORBextractorT() { orbgpu_extractor_destroy(h_); }
    ORBextractorT(const ORBextractorT &) = delete;
    ORBextractorT &operator=(const ORBextractorT &) = delete;

    // operator()(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray descriptors)
    // (ORBextractor.h:59-61).  `image`: 8-bit gray rows x cols with `step` bytes per row; the mask is
    // ignored as in the reference; descriptors: N x 32 bytes, contiguous (ORBextractor.cc:1068).
    void operator()(const uint8_t *image, int rows, int cols, size_t step, std::vector<KeyPointT> &keypoints,
                    std::vector<uint8_t> &descriptors)
    {
        keypoints.clear();
        descriptors.clear();
        if (!image || rows <= 0 || cols <= 0)
            return;  // ORBextractor.cc:1046
        int32_t cap = 0;
        check(orbgpu_extractor_max_keypoints(h_, cols, rows, &cap), "ORBextractor::operator()");
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int32_t n = 0;
        check(orbgpu_extract(h_, image, cols, rows, step, reinterpret_cast<orbgpu_keypoint *>(keypoints.data()),
                             descriptors.data(), cap, &n),
              "ORBextractor::operator()");
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        last_rows_ = rows;
        last_cols_ = cols;
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid[level] (ORBextractor.h:85): lazily copied from the device; only stereo
    // matching reads it (Frame.cc:471,561,578).
    void GetPyramidLevel(int level, std::vector<uint8_t> &pixels, int &width, int &height)
    {
        // level size as ComputePyramid computes it (ORBextractor.cc:1112)
        int32_t w = (int32_t)lrintf((float)last_cols_ * mvInvScaleFactor[level]);
        int32_t h = (int32_t)lrintf((float)last_rows_ * mvInvScaleFactor[level]);
        pixels.resize((size_t)w * h);
        check(orbgpu_extractor_get_pyramid_level(h_, 0, level, pixels.data(), (size_t)w, &w, &h), "pyramid");
        width = w;
        height = h;
    }

    orbgpu_extractor *handle() { return h_; }

  protected:
    orbgpu_extractor *h_ = nullptr;
    int nlevels_ = 0, last_rows_ = 0, last_cols_ = 0;
    float scaleFactor_ = 0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// --------------------------------------------------------------------------------------------
// ORBVocabulary (reference include/ORBVocabulary.h: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>)
// The file readers (loadFromTextFile / loadFromBinaryFile) stay the reference's; after loading, hand the node arrays
// over once (node i > 0: parent, isLeaf, descriptor row, weight -- m_nodes in index order).
// --------------------------------------------------------------------------------------------
typedef std::map<unsigned int, double> BowVector;                      // DBoW2::BowVector
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;  // DBoW2::FeatureVector

class ORBVocabularyT {
  public:
    ORBVocabularyT(int k, int L, const std::vector<int32_t> &parent, const std::vector<uint8_t> &is_leaf,
                   const std::vector<uint8_t> &desc, const std::vector<double> &weight, int weighting = 0,
                   int scoring = 0, int device_id = 0)
    {
        check(orbgpu_vocabulary_create(k, L, (int32_t)parent.size(), parent.data(), is_leaf.data(), desc.data(),
                                       weight.data(), weighting, scoring, device_id, &h_),
              "ORBVocabulary");
    }
    ~ORBVocabularyT() { orbgpu_vocabulary_destroy(h_); }
    ORBVocabularyT(const ORBVocabularyT &) = delete;
    ORBVocabularyT &operator=(const ORBVocabularyT &) = delete;

    // void transform(const std::vector<TDescriptor>& features, BowVector &v, FeatureVector &fv, int levelsup) const
    // (TemplatedVocabulary.h:145-146; Frame::ComputeBoW, Frame.cc:395-402).  descriptors: n rows of 32 bytes.
    void transform(const uint8_t *descriptors, int n, BowVector &v, FeatureVector &fv, int levelsup) const
    {
        v.clear();
        fv.clear();
        const size_t m = (size_t)std::max(n, 1);
        std::vector<int32_t> wid(m), nid(m), bid(m), fvn(m), fvs(m + 1), fvi(m);
        std::vector<double> wgt(m), bval(m);
        int32_t nb = 0, nf = 0;
        check(orbgpu_bow_transform(h_, descriptors, n, levelsup, wid.data(), wgt.data(), nid.data(), bid.data(),
                                   bval.data(), &nb, fvn.data(), fvs.data(), fvi.data(), &nf),
              "ORBVocabulary::transform");
        for (int i = 0; i < nb; i++)
            v.emplace_hint(v.end(), (unsigned)bid[i], bval[i]);
        for (int t = 0; t < nf; t++)
            fv.emplace_hint(fv.end(), (unsigned)fvn[t],
                            std::vector<unsigned int>(fvi.begin() + fvs[t], fvi.begin() + fvs[t + 1]));
    }
    orbgpu_vocabulary *handle() const { return h_; }

  protected:
    orbgpu_vocabulary *h_ = nullptr;
};

// per-feature node ids (-1 = not in the vector) from a FeatureVector: what orbgpu_search_by_bow consumes
inline std::vector<int32_t> NodeIdsOf(const FeatureVector &fv, int n)
{
    std::vector<int32_t> node((size_t)std::max(n, 1), -1);
    for (const auto &kv : fv)
        for (unsigned int i : kv.second)
            if ((int)i < n)
                node[i] = (int32_t)kv.first;
    return node;
}

// --------------------------------------------------------------------------------------------
// ORBmatcher (reference include/ORBmatcher.h:41-106)
// --------------------------------------------------------------------------------------------
constexpr int FRAME_GRID_ROWS = ORBGPU_GRID_ROWS, FRAME_GRID_COLS = ORBGPU_GRID_COLS;

// Flattens the Frame members the matcher reads into the ABI's SoA view.  FrameT needs the
// reference's member names: N, mvKeysUn[i].pt/.octave/.angle, mvuRight, descriptor rows via
// desc_row(F,i), mGrid[COLS][ROWS], mnMinX.., mfGridElementWidthInv.., mvScaleFactors.
template <typename FrameT> struct FrameSoA {
    std::vector<float> x, y, angle, uright;
    std::vector<int32_t> octave, cell_start, cell_items;
    std::vector<uint8_t> desc;
    orbgpu_frame_view view{};
    template <typename DescRow> FrameSoA(const FrameT &F, DescRow desc_row)
    {
        const int n = F.N;
        x.resize(n), y.resize(n), angle.resize(n), uright.resize(n), octave.resize(n), desc.resize((size_t)n * 32);
        for (int i = 0; i < n; i++) {
            x[i] = F.mvKeysUn[i].pt.x;
            y[i] = F.mvKeysUn[i].pt.y;
            angle[i] = F.mvKeysUn[i].angle;
            octave[i] = F.mvKeysUn[i].octave;
            uright[i] = F.mvuRight[i];
            std::memcpy(&desc[(size_t)i * 32], desc_row(F, i), 32);
        }
        cell_start.assign(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, 0);
        for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
            for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
                const auto &cell = F.mGrid[ix][iy];
                cell_start[ix * FRAME_GRID_ROWS + iy + 1] = cell_start[ix * FRAME_GRID_ROWS + iy] + (int32_t)cell.size();
                for (size_t k = 0; k < cell.size(); k++)
                    cell_items.push_back((int32_t)cell[k]);
            }
        if (cell_items.empty())
            cell_items.push_back(0);
        view.n = n;
        view.kp_x = x.data(), view.kp_y = y.data(), view.kp_octave = octave.data(), view.kp_angle = angle.data();
        view.u_right = uright.data(), view.desc = desc.data();
        view.min_x = F.mnMinX, view.max_x = F.mnMaxX, view.min_y = F.mnMinY, view.max_y = F.mnMaxY;
        view.grid_inv_w = F.mfGridElementWidthInv, view.grid_inv_h = F.mfGridElementHeightInv;
        view.scale_factors = F.mvScaleFactors.data(), view.nlevels = (int32_t)F.mvScaleFactors.size();
        view.cell_start = cell_start.data(), view.cell_items = cell_items.data();
    }
};

// Positions of a few pointers inside a long pointer list in O(list + queries): the queries go into a small
// open-addressing set (cache resident), the list is walked once.  Replaces the per-key-point linear search over
// vpMapPoints (2 016 x 10 059 pointer compares per SearchByProjection call at C3's size).
template <typename T> class PtrIndex {
  public:
    // keys: the pointers to look for (nullptr entries are ignored)
    template <typename It> PtrIndex(It first, It last)
    {
        size_t n = 0;
        for (It it = first; it != last; ++it)
            n += *it != nullptr;
        size_t cap = 16;
        while (cap < 2 * n + 2)
            cap *= 2;
        mask_ = cap - 1;
        key_.assign(cap, nullptr);
        val_.assign(cap, -1);
        for (It it = first; it != last; ++it)
            if (*it != nullptr)
                slot(*it, true);
    }
    // first position of every key inside [list, list + m)
    void locate(T *const *list, int m)
    {
        for (int i = 0; i < m; i++) {
            const long s = slot(list[i], false);
            if (s >= 0 && val_[(size_t)s] < 0)
                val_[(size_t)s] = i;
        }
    }
    int position(T *p) const
    {
        if (!p)
            return -1;
        for (size_t s = hash(p) & mask_;; s = (s + 1) & mask_) {
            if (key_[s] == p)
                return val_[s];
            if (!key_[s])
                return -1;
        }
    }

  private:
    static size_t hash(const T *p) { return (size_t)(((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull >> 20); }
    long slot(T *p, bool insert)
    {
        if (!p)
            return -1;
        for (size_t s = hash(p) & mask_;; s = (s + 1) & mask_) {
            if (key_[s] == p)
                return (long)s;
            if (!key_[s]) {
                if (!insert)
                    return -1;
                key_[s] = p;
                return (long)s;
            }
        }
    }
    std::vector<T *> key_;
    std::vector<int> val_;
    size_t mask_ = 0;
};

// --------------------------------------------------------------------------------------------
// Device-resident MapPoint table (orbgpu_mappoint_table_*): what the matchers read of every map point, keyed by
// MapPoint::mnId (MapPoint.h:84), so that a matcher call hands over ids instead of locking and cloning 10 k objects
// (MapPoint::GetDescriptor, MapPoint.cc:309-313).  MapPointT needs mnId; the accessors are the same callables the
// matcher overloads take.  INTEGRATION.md section 2c lists where the reference calls these.
// --------------------------------------------------------------------------------------------
template <typename MapPointT> class MapPointTableT {
  public:
    explicit MapPointTableT(int device_id = 0, int initial_rows = 0)
    {
        check(orbgpu_mappoint_table_create(device_id, initial_rows, &h_), "MapPointTable");
    }
    ~MapPointTableT() { orbgpu_mappoint_table_destroy(h_); }
    MapPointTableT(const MapPointTableT &) = delete;
    MapPointTableT &operator=(const MapPointTableT &) = delete;
    orbgpu_mappoint_table *handle() const { return h_; }
    // One caller at a time (the C ABI's rule for a table).  In the reference LocalMapping creates and culls map points
    // while Tracking searches, and only some of those edits happen under Map::mMutexMapUpdate, so the wrapper serialises
    // its own calls and the matcher overloads that search the table take the same lock.
    std::mutex &mutex() const { return mu_; }
    int rows() const
    {
        std::lock_guard<std::mutex> g(mu_);
        int32_t r = 0;
        check(orbgpu_mappoint_table_rows(h_, &r), "MapPointTable::rows");
        return r;
    }

    // every attribute of a batch of points (new key frame: MapPoint constructors, UpdateNormalAndDepth,
    // ComputeDistinctiveDescriptors of the points it created or observed)
    template <typename WorldPos, typename Normal, typename MinDist, typename MaxDist, typename MpDesc>
    void Upsert(const std::vector<MapPointT *> &pts, WorldPos world_pos, Normal normal, MinDist min_dist, MaxDist max_dist,
                MpDesc mp_desc)
    {
        std::lock_guard<std::mutex> g(mu_);
        const size_t n = pts.size();
        ids_.resize(n), wp_.resize(3 * n), nr_.resize(3 * n), mn_.resize(n), mx_.resize(n), ds_.resize(32 * n), ob_.resize(n);
        for (size_t i = 0; i < n; i++) {
            MapPointT *p = pts[i];
            ids_[i] = (int64_t)p->mnId;
            const float *w = world_pos(p), *nn = normal(p);
            for (int c = 0; c < 3; c++)
                wp_[3 * i + c] = w[c], nr_[3 * i + c] = nn[c];
            mn_[i] = min_dist(p), mx_[i] = max_dist(p);
            std::memcpy(&ds_[32 * i], mp_desc(p), 32);
            ob_[i] = p->Observations();
        }
        check(orbgpu_mappoint_table_upsert(h_, (int32_t)n, ids_.data(), wp_.data(), nr_.data(), mn_.data(), mx_.data(),
                                           ds_.data(), ob_.data()),
              "MapPointTable::Upsert");
        for (size_t i = 0; i < n; i++)
            if (pts[i]->isBad())
                bad_.push_back(ids_[i]);
        if (!bad_.empty()) {
            check(orbgpu_mappoint_table_set_bad(h_, (int32_t)bad_.size(), bad_.data(), nullptr), "MapPointTable::SetBad");
            bad_.clear();
        }
    }
    // MapPoint::SetWorldPos (MapPoint.cc:73-78)
    void SetWorldPos(MapPointT *p, const float *w)
    {
        std::lock_guard<std::mutex> g(mu_);
        const int64_t id = (int64_t)p->mnId;
        check(orbgpu_mappoint_table_upsert(h_, 1, &id, w, nullptr, nullptr, nullptr, nullptr, nullptr), "SetWorldPos");
    }
    // MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:242-307): the chosen descriptor
    void SetDescriptor(MapPointT *p, const uint8_t *d)
    {
        std::lock_guard<std::mutex> g(mu_);
        const int64_t id = (int64_t)p->mnId;
        check(orbgpu_mappoint_table_upsert(h_, 1, &id, nullptr, nullptr, nullptr, nullptr, d, nullptr), "SetDescriptor");
    }
    // MapPoint::AddObservation / EraseObservation (MapPoint.cc:98-149)
    void SetObservations(MapPointT *p)
    {
        std::lock_guard<std::mutex> g(mu_);
        const int64_t id = (int64_t)p->mnId;
        const int32_t n = p->Observations();
        check(orbgpu_mappoint_table_set_observations(h_, 1, &id, &n, nullptr), "SetObservations");
    }
    // MapPoint::SetBadFlag (MapPoint.cc:151-175), MapPoint::Replace (:177-228)
    void SetBad(MapPointT *p)
    {
        std::lock_guard<std::mutex> g(mu_);
        const int64_t id = (int64_t)p->mnId;
        check(orbgpu_mappoint_table_set_bad(h_, 1, &id, nullptr), "SetBad");
    }
    // Batch forms: ONE staged copy, launch and synchronisation for the whole set, under one hold of the mutex Tracking's
    // searches take.  These are what the bulk writers call: the write-back loops of bundle adjustment
    // (Optimizer.cc:775 / 1040: SetWorldPos + UpdateNormalAndDepth per point, thousands per local BA), loop correction
    // (LoopClosing.cc:520-560) and MapPointCulling.  The one-id forms above cost a full host <-> device round trip each.
    template <typename WorldPos, typename Normal, typename MinDist, typename MaxDist>
    void SetWorldPos(const std::vector<MapPointT *> &pts, WorldPos world_pos, Normal normal, MinDist min_dist, MaxDist max_dist)
    {
        std::lock_guard<std::mutex> g(mu_);
        const size_t n = pts.size();
        ids_.resize(n), wp_.resize(3 * n), nr_.resize(3 * n), mn_.resize(n), mx_.resize(n);
        for (size_t i = 0; i < n; i++) {
            MapPointT *p = pts[i];
            ids_[i] = (int64_t)p->mnId;
            const float *w = world_pos(p), *nn = normal(p);
            for (int c = 0; c < 3; c++)
                wp_[3 * i + c] = w[c], nr_[3 * i + c] = nn[c];
            mn_[i] = min_dist(p), mx_[i] = max_dist(p);
        }
        check(orbgpu_mappoint_table_upsert(h_, (int32_t)n, ids_.data(), wp_.data(), nr_.data(), mn_.data(), mx_.data(), nullptr,
                                           nullptr),
              "SetWorldPos (batch)");
    }
    void SetObservations(const std::vector<MapPointT *> &pts)
    {
        std::lock_guard<std::mutex> g(mu_);
        const size_t n = pts.size();
        ids_.resize(n), ob_.resize(n);
        for (size_t i = 0; i < n; i++)
            ids_[i] = (int64_t)pts[i]->mnId, ob_[i] = pts[i]->Observations();
        check(orbgpu_mappoint_table_set_observations(h_, (int32_t)n, ids_.data(), ob_.data(), nullptr), "SetObservations (batch)");
    }
    // the count taken by the caller: for a hook that runs where MapPoint::mMutexFeatures is held (Observations() would
    // lock it again)
    void SetObservations(MapPointT *p, int nObs)
    {
        std::lock_guard<std::mutex> g(mu_);
        const int64_t id = (int64_t)p->mnId;
        const int32_t n = nObs;
        check(orbgpu_mappoint_table_set_observations(h_, 1, &id, &n, nullptr), "SetObservations");
    }
    void SetBad(const std::vector<MapPointT *> &pts)
    {
        std::lock_guard<std::mutex> g(mu_);
        ids_.resize(pts.size());
        for (size_t i = 0; i < pts.size(); i++)
            ids_[i] = (int64_t)pts[i]->mnId;
        check(orbgpu_mappoint_table_set_bad(h_, (int32_t)ids_.size(), ids_.data(), nullptr), "SetBad (batch)");
    }
    // Bounds the table: only the listed points stay (Map::GetAllMapPoints() + what the current / last Frame still hold);
    // returns the number of rows released.  The reference never frees a MapPoint; its table equivalent is this call.
    int Retain(const std::vector<MapPointT *> &pts)
    {
        std::lock_guard<std::mutex> g(mu_);
        ids_.resize(pts.size());
        for (size_t i = 0; i < pts.size(); i++)
            ids_[i] = (int64_t)pts[i]->mnId;
        int32_t dropped = 0;
        check(orbgpu_mappoint_table_retain(h_, (int32_t)ids_.size(), ids_.data(), &dropped), "Retain");
        return dropped;
    }
    // ids of the last table search that the table had not been told about yet (skipped rows, held key points)
    std::pair<int, int> LastUnknown() const
    {
        std::lock_guard<std::mutex> g(mu_);
        int32_t a = 0, b = 0;
        check(orbgpu_mappoint_table_last_unknown(h_, &a, &b), "LastUnknown");
        return {a, b};
    }

  private:
    mutable std::mutex mu_;
    orbgpu_mappoint_table *h_ = nullptr;
    std::vector<int64_t> ids_, bad_;
    std::vector<float> wp_, nr_, mn_, mx_;
    std::vector<uint8_t> ds_;
    std::vector<int32_t> ob_;
};

// A Frame's matcher-side members on the device (orbgpu_frame_*): uploaded once per frame, then current frame of
// SearchByProjection(Cur, Last) and of SearchLocalPoints, and last frame of the next call.
template <typename FrameT> class DeviceFrameT {
  public:
    explicit DeviceFrameT(int device_id = 0) { check(orbgpu_frame_create(device_id, &h_), "DeviceFrame"); }
    ~DeviceFrameT() { orbgpu_frame_destroy(h_); }
    DeviceFrameT(const DeviceFrameT &) = delete;
    DeviceFrameT &operator=(const DeviceFrameT &) = delete;
    template <typename DescRow> void Upload(const FrameT &F, DescRow desc_row)
    {
        FrameSoA<FrameT> soa(F, desc_row);
        check(orbgpu_frame_upload(h_, &soa.view), "DeviceFrame::Upload");
        n_ = F.N;
    }
    orbgpu_frame *handle() const { return h_; }
    int N() const { return n_; }

  private:
    orbgpu_frame *h_ = nullptr;
    int n_ = 0;
};

template <typename FrameT, typename MapPointT> class ORBmatcherT {
  public:
    static const int TH_LOW = ORBGPU_TH_LOW, TH_HIGH = ORBGPU_TH_HIGH, HISTO_LENGTH = ORBGPU_HISTO_LENGTH;

    ORBmatcherT(float nnratio = 0.6f, bool checkOri = true, int device_id = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device_id)
    {
    }

    // static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)  (ORBmatcher.h:44)
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b, int device_id = 0)
    {
        int32_t d = 0;
        check(orbgpu_hamming256(a, b, 1, &d, device_id), "DescriptorDistance");
        return d;
    }

    // int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th=3)
    // (ORBmatcher.h:48, ORBmatcher.cc:45-129).  desc_row(F,i) / mp_desc(pMP) return 32-byte rows.
    template <typename DescRow, typename MpDesc>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th, DescRow desc_row,
                           MpDesc mp_desc)
    {
        FrameSoA<FrameT> soa(F, desc_row);
        const int m = (int)vpMapPoints.size();
        std::vector<uint8_t> in_view(m), bad(m), obs(m), desc((size_t)std::max(m, 1) * 32);
        std::vector<int32_t> level(m);
        std::vector<float> vcos(m), px(m), py(m), pxr(m);
        for (int i = 0; i < m; i++) {
            MapPointT *p = vpMapPoints[i];
            in_view[i] = p->mbTrackInView;
            bad[i] = p->isBad();
            obs[i] = p->Observations() > 0;
            level[i] = p->mnTrackScaleLevel;
            vcos[i] = p->mTrackViewCos;
            px[i] = p->mTrackProjX, py[i] = p->mTrackProjY, pxr[i] = p->mTrackProjXR;
            std::memcpy(&desc[(size_t)i * 32], mp_desc(p), 32);
        }
        orbgpu_mappoint_view mv{m,         in_view.data(), bad.data(), obs.data(), level.data(),
                                vcos.data(), px.data(),      py.data(),  pxr.data(), desc.data()};
        // F.mvpMapPoints -> indices: points of the list by position, others by their Observations()
        std::vector<int32_t> k2m(F.N, -1);
        PtrIndex<MapPointT> held(F.mvpMapPoints.begin(), F.mvpMapPoints.end());
        held.locate(vpMapPoints.data(), m);
        for (int j = 0; j < F.N; j++) {
            MapPointT *p = F.mvpMapPoints[j];
            if (!p)
                continue;
            const int idx = held.position(p);
            k2m[j] = idx >= 0 ? idx : (p->Observations() > 0 ? -2 : -1);
        }
        std::vector<int32_t> before = k2m;
        int32_t nmatches = 0;
        check(orbgpu_search_by_projection(&soa.view, &mv, th, mfNNratio, k2m.data(), &nmatches, device_),
              "SearchByProjection");
        for (int j = 0; j < F.N; j++)
            if (k2m[j] != before[j] && k2m[j] >= 0)
                F.mvpMapPoints[j] = vpMapPoints[k2m[j]];  // ORBmatcher.cc:123
        return nmatches;
    }

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    // (ORBmatcher.h:52, ORBmatcher.cc:1328-1470).  Tcw(F) returns the 16 floats of F.mTcw (row-major);
    // world_pos(pMP) the 3 floats of GetWorldPos().
    template <typename DescRow, typename MpDesc, typename TcwOf, typename WorldPos>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono,
                           DescRow desc_row, MpDesc mp_desc, TcwOf Tcw, WorldPos world_pos)
    {
        FrameSoA<FrameT> cur(CurrentFrame, desc_row);
        const int n = LastFrame.N;
        std::vector<uint8_t> has(n), outl(n), obs(n), desc((size_t)std::max(n, 1) * 32);
        std::vector<float> wp((size_t)std::max(n, 1) * 3), ang(n);
        std::vector<int32_t> oct(n);
        for (int i = 0; i < n; i++) {
            MapPointT *p = LastFrame.mvpMapPoints[i];
            has[i] = p != nullptr;
            outl[i] = LastFrame.mvbOutlier[i];
            oct[i] = LastFrame.mvKeys[i].octave;
            ang[i] = LastFrame.mvKeysUn[i].angle;
            if (p) {
                obs[i] = p->Observations() > 0;
                const float *w = world_pos(p);
                wp[3 * i] = w[0], wp[3 * i + 1] = w[1], wp[3 * i + 2] = w[2];
                std::memcpy(&desc[(size_t)i * 32], mp_desc(p), 32);
            }
        }
        orbgpu_lastframe_view lv{n, has.data(), outl.data(), obs.data(), wp.data(), desc.data(), oct.data(), ang.data(),
                                 Tcw(LastFrame)};
        std::vector<int32_t> k2m(CurrentFrame.N, -1);
        for (int j = 0; j < CurrentFrame.N; j++)
            if (MapPointT *p = CurrentFrame.mvpMapPoints[j])
                k2m[j] = p->Observations() > 0 ? -2 : -1;
        std::vector<int32_t> before = k2m;
        int32_t nmatches = 0;
        check(orbgpu_search_by_projection_last(&cur.view, Tcw(CurrentFrame), CurrentFrame.fx, CurrentFrame.fy,
                                               CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf, CurrentFrame.mb, &lv,
                                               th, bMono, mbCheckOrientation, k2m.data(), &nmatches, device_),
              "SearchByProjection(last)");
        for (int j = 0; j < CurrentFrame.N; j++) {
            if (k2m[j] == before[j])
                continue;
            CurrentFrame.mvpMapPoints[j] = k2m[j] >= 0 ? LastFrame.mvpMapPoints[k2m[j]] : nullptr;  // :1428, :1461
        }
        return nmatches;
    }

    // ---- the same two matchers over the device-resident MapPoint table ------------------------------------------------
    // SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:45-129) as a drop-in at the ORBmatcher level: the mTrack*
    // members Frame::isInFrustum filled are uploaded (7 values per point); descriptors, isBad() and Observations() come
    // from the table by mnId.  dF holds F's matcher-side members on the device (DeviceFrameT::Upload, once per frame).
    int SearchByProjection(FrameT &F, const DeviceFrameT<FrameT> &dF, const std::vector<MapPointT *> &vpMapPoints,
                           const float th, MapPointTableT<MapPointT> &table)
    {
        const int m = (int)vpMapPoints.size();
        ids_.resize(m), b0_.resize(m), i0_.resize(m), f0_.resize(m), f1_.resize(m), f2_.resize(m), f3_.resize(m);
        for (int i = 0; i < m; i++) {
            const MapPointT *p = vpMapPoints[i];
            ids_[i] = (int64_t)p->mnId;
            b0_[i] = p->mbTrackInView;
            i0_[i] = p->mnTrackScaleLevel;
            f0_[i] = p->mTrackViewCos, f1_[i] = p->mTrackProjX, f2_[i] = p->mTrackProjY, f3_[i] = p->mTrackProjXR;
        }
        orbgpu_mappoint_view sc{m, b0_.data(), nullptr, nullptr, i0_.data(), f0_.data(), f1_.data(), f2_.data(), f3_.data(), nullptr};
        return run_local(F, dF, vpMapPoints, table, nullptr, &sc, nullptr, 0, 0, 0, 0, 0, 0, 0.5f, th, nullptr);
    }

    // Tracking::SearchLocalPoints (Tracking.cc:1468-1496) in one call: Frame::isInFrustum + MapPoint::PredictScale run on
    // the device over the table, then SearchByProjection(F, vpLocalMapPoints, th).  skip(pMP) is the host-only half of
    // :1474 (pMP->mnLastFrameSeen == F.mnId); isBad() comes from the table.  Returns the match count; *nToMatch
    // (optional) = points in the frustum; in_view(pMP, bool) is called for every listed point so that the caller can do
    // IncreaseVisible() and keep mbTrackInView current (:1479-1483).
    template <typename Skip, typename TcwOf, typename InView>
    int SearchLocalPoints(FrameT &F, const DeviceFrameT<FrameT> &dF, const std::vector<MapPointT *> &vpLocalMapPoints,
                          const float th, MapPointTableT<MapPointT> &table, Skip skip, TcwOf Tcw, InView in_view,
                          int *nToMatch = nullptr)
    {
        const int m = (int)vpLocalMapPoints.size();
        ids_.resize(m), b0_.resize(m), b1_.resize(m);
        for (int i = 0; i < m; i++) {
            ids_[i] = (int64_t)vpLocalMapPoints[i]->mnId;
            b0_[i] = skip(vpLocalMapPoints[i]) ? 1 : 0;
        }
        orbgpu_track_scratch trk{b1_.data(), nullptr, nullptr, nullptr, nullptr, nullptr};
        const int nm = run_local(F, dF, vpLocalMapPoints, table, b0_.data(), nullptr, Tcw(F), F.fx, F.fy, F.cx, F.cy, F.mbf,
                                 F.mfLogScaleFactor, 0.5f, th, &trk);
        int seen = 0;
        for (int i = 0; i < m; i++) {
            in_view(vpLocalMapPoints[i], b1_[i] != 0);
            seen += b1_[i] != 0;
        }
        if (nToMatch)
            *nToMatch = seen;
        return nm;
    }

    // SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1328-1470) with both frames on the device and
    // the last frame's map points (world position, descriptor, Observations()) looked up by mnId.
    template <typename TcwOf>
    int SearchByProjection(FrameT &CurrentFrame, const DeviceFrameT<FrameT> &dCur, const FrameT &LastFrame,
                           const DeviceFrameT<FrameT> &dLast, const float th, const bool bMono,
                           MapPointTableT<MapPointT> &table, TcwOf Tcw)
    {
        const int nl = LastFrame.N, n = CurrentFrame.N;
        ids_.resize(nl), b0_.resize(nl), kid_.resize(n), k2m_.resize(std::max(n, 1));
        for (int i = 0; i < nl; i++) {
            const MapPointT *p = LastFrame.mvpMapPoints[i];
            ids_[i] = p ? (int64_t)p->mnId : -1;
            b0_[i] = LastFrame.mvbOutlier[i];
        }
        for (int j = 0; j < n; j++)
            kid_[j] = CurrentFrame.mvpMapPoints[j] ? (int64_t)CurrentFrame.mvpMapPoints[j]->mnId : -1;
        int32_t nmatches = 0;
        std::lock_guard<std::mutex> table_lock(table.mutex());
        check(orbgpu_search_by_projection_last_table(dCur.handle(), Tcw(CurrentFrame), dLast.handle(), Tcw(LastFrame),
                                                     table.handle(), ids_.data(), b0_.data(), kid_.data(), CurrentFrame.fx,
                                                     CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf,
                                                     CurrentFrame.mb, th, bMono, mbCheckOrientation, k2m_.data(), &nmatches),
              "SearchByProjection(last, table)");
        for (int j = 0; j < n; j++)
            if (k2m_[j] >= 0)
                CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[k2m_[j]];  // :1428
        return nmatches;
    }

    // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint*> &sAlreadyFound,
    //                        const float th, const int ORBdist)   (ORBmatcher.h:56, ORBmatcher.cc:1472-1599)
    // KeyFrameT needs GetMapPointMatches() and mvKeysUn; found(pMP) tells whether pMP is in sAlreadyFound;
    // max_dist(pMP) returns mfMaxDistance (the numerator of MapPoint::PredictScale, a protected member:
    // add an accessor or a friend declaration to MapPoint.h).
    template <typename KeyFrameT, typename DescRow, typename MpDesc, typename TcwOf, typename WorldPos, typename Found,
              typename MaxDist>
    int SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, Found found, const float th, const int ORBdist,
                           DescRow desc_row, MpDesc mp_desc, TcwOf Tcw, WorldPos world_pos, MaxDist max_dist)
    {
        FrameSoA<FrameT> cur(CurrentFrame, desc_row);
        const std::vector<MapPointT *> vpMPs = pKF->GetMapPointMatches();
        const int n = (int)vpMPs.size();
        std::vector<uint8_t> has(n), bad(n), fnd(n), desc((size_t)std::max(n, 1) * 32);
        std::vector<float> wp((size_t)std::max(n, 1) * 3), mininv(n), maxinv(n), maxd(n), ang(n);
        for (int i = 0; i < n; i++) {
            MapPointT *p = vpMPs[i];
            has[i] = p != nullptr;
            ang[i] = pKF->mvKeysUn[i].angle;
            if (!p)
                continue;
            bad[i] = p->isBad();
            fnd[i] = found(p);
            mininv[i] = p->GetMinDistanceInvariance(), maxinv[i] = p->GetMaxDistanceInvariance();
            maxd[i] = max_dist(p);
            const float *w = world_pos(p);
            wp[3 * i] = w[0], wp[3 * i + 1] = w[1], wp[3 * i + 2] = w[2];
            std::memcpy(&desc[(size_t)i * 32], mp_desc(p), 32);
        }
        orbgpu_keyframe_view kv{n, has.data(), bad.data(), fnd.data(), wp.data(), mininv.data(), maxinv.data(), maxd.data(),
                                desc.data(), ang.data()};
        std::vector<int32_t> k2m(CurrentFrame.N, -1);
        for (int j = 0; j < CurrentFrame.N; j++)
            if (CurrentFrame.mvpMapPoints[j])
                k2m[j] = -2;
        int32_t nmatches = 0;
        check(orbgpu_search_by_projection_keyframe(&cur.view, Tcw(CurrentFrame), CurrentFrame.fx, CurrentFrame.fy,
                                                   CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mfLogScaleFactor, &kv,
                                                   th, ORBdist, mbCheckOrientation, k2m.data(), &nmatches, device_),
              "SearchByProjection(keyframe)");
        for (int j = 0; j < CurrentFrame.N; j++)
            if (k2m[j] >= 0)
                CurrentFrame.mvpMapPoints[j] = vpMPs[k2m[j]];  // :1556
        return nmatches;
    }

    // int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*> &vpPoints,
    //                        std::vector<MapPoint*> &vpMatched, int th)   (ORBmatcher.h:60, ORBmatcher.cc:290-403)
    // KeyFrameT needs the Frame member names FrameSoA reads (KeyFrame.h has them all) plus fx, fy, cx, cy and
    // mfLogScaleFactor; Scw is the 4x4 row-major float Sim3; normal / min_dist / max_dist return GetNormal() and the
    // protected mfMinDistance / mfMaxDistance of a map point.
    template <typename KeyFrameT, typename DescRow, typename MpDesc, typename WorldPos, typename Normal,
              typename MinDist, typename MaxDist>
    int SearchByProjection(KeyFrameT *pKF, const float *Scw, const std::vector<MapPointT *> &vpPoints,
                           std::vector<MapPointT *> &vpMatched, int th, DescRow desc_row, MpDesc mp_desc,
                           WorldPos world_pos, Normal normal, MinDist min_dist, MaxDist max_dist)
    {
        FrameSoA<KeyFrameT> kf(*pKF, desc_row);
        const int m = (int)vpPoints.size();
        std::vector<uint8_t> bad(std::max(m, 1)), desc((size_t)std::max(m, 1) * 32);
        std::vector<float> wp((size_t)std::max(m, 1) * 3), nr((size_t)std::max(m, 1) * 3), mind(std::max(m, 1)),
            maxd(std::max(m, 1));
        for (int i = 0; i < m; i++) {
            MapPointT *p = vpPoints[i];
            bad[i] = p->isBad();
            const float *w = world_pos(p), *nn = normal(p);
            wp[3 * i] = w[0], wp[3 * i + 1] = w[1], wp[3 * i + 2] = w[2];
            nr[3 * i] = nn[0], nr[3 * i + 1] = nn[1], nr[3 * i + 2] = nn[2];
            mind[i] = min_dist(p), maxd[i] = max_dist(p);
            std::memcpy(&desc[(size_t)i * 32], mp_desc(p), 32);
        }
        orbgpu_points_view pv{m, bad.data(), wp.data(), nr.data(), mind.data(), maxd.data(), desc.data()};
        std::vector<int32_t> k2m(pKF->N, -1);
        PtrIndex<MapPointT> row(vpMatched.begin(), vpMatched.end());  // first row of every matched point (vpPoints has no duplicates in the reference)
        row.locate(vpPoints.data(), m);
        for (int j = 0; j < pKF->N; j++)
            if (vpMatched[j]) {
                const int r = row.position(vpMatched[j]);
                k2m[j] = r < 0 ? -2 : r;
            }
        const std::vector<int32_t> before = k2m;
        int32_t nmatches = 0;
        check(orbgpu_search_by_projection_sim3(&kf.view, Scw, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mfLogScaleFactor,
                                               &pv, th, k2m.data(), &nmatches, device_),
              "SearchByProjection(Sim3)");
        for (int j = 0; j < pKF->N; j++)
            if (k2m[j] != before[j])
                vpMatched[j] = vpPoints[k2m[j]];  // :394
        return nmatches;
    }

    // int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)
    // (ORBmatcher.h:64, ORBmatcher.cc:159-288; Tracking::TrackReferenceKeyFrame, Tracking.cc:1051).
    // KeyFrameT: N, mvKeysUn, mFeatVec, GetMapPointMatches(); FrameT: N, mvKeys, mFeatVec.
    template <typename KeyFrameT, typename KfDescRow, typename DescRow>
    int SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches, KfDescRow kf_desc_row,
                    DescRow desc_row)
    {
        const std::vector<MapPointT *> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches.assign(F.N, nullptr);  // :163
        const int nk = (int)vpMapPointsKF.size(), nf = F.N;
        std::vector<uint8_t> dk((size_t)std::max(nk, 1) * 32), df((size_t)std::max(nf, 1) * 32), valid(std::max(nk, 1));
        std::vector<float> ak(std::max(nk, 1)), af(std::max(nf, 1));
        for (int i = 0; i < nk; i++) {
            std::memcpy(&dk[(size_t)i * 32], kf_desc_row(*pKF, i), 32);
            ak[i] = pKF->mvKeysUn[i].angle;
            valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :195-199
        }
        for (int j = 0; j < nf; j++) {
            std::memcpy(&df[(size_t)j * 32], desc_row(F, j), 32);
            af[j] = F.mvKeys[j].angle;  // :238
        }
        const std::vector<int32_t> nodek = NodeIdsOf(pKF->mFeatVec, nk), nodef = NodeIdsOf(F.mFeatVec, nf);
        std::vector<int32_t> match(std::max(nf, 1), -1);
        int32_t nmatches = 0;
        check(orbgpu_search_by_bow(dk.data(), ak.data(), valid.data(), nodek.data(), nk, df.data(), af.data(), nodef.data(),
                                   nf, TH_LOW, mfNNratio, mbCheckOrientation, match.data(), &nmatches, device_),
              "SearchByBoW");
        for (int j = 0; j < nf; j++)
            if (match[j] >= 0)
                vpMapPointMatches[j] = vpMapPointsKF[match[j]];  // :232
        return nmatches;
    }

    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,
    //                            std::vector<pair<size_t,size_t>> &vMatchedPairs, const bool bOnlyStereo)
    // (ORBmatcher.h:79, ORBmatcher.cc:657-823; LocalMapping::CreateNewMapPoints, LocalMapping.cc:268).
    // KeyFrameT: the members FrameSoA reads (N, mvKeysUn, mvuRight, mGrid, mnMinX.., mvScaleFactors), mFeatVec,
    // GetMapPoint(idx), fx, fy, cx, cy, mvLevelSigma2.  F12: the 9 floats of the fundamental matrix, row-major;
    // camera_center(kf, float[3]) / rotation(kf, float[9]) / translation(kf, float[3]) read GetCameraCenter(),
    // GetRotation(), GetTranslation().
    template <typename KeyFrameT, typename DescRow, typename Vec3Of, typename Mat3Of, typename Vec3Of2>
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, const float *F12,
                               std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo,
                               DescRow desc_row, Vec3Of camera_center, Mat3Of rotation, Vec3Of2 translation)
    {
        // epipole in the second image (:664-670): C2 = R2w * Cw + t2w as cv::gemm computes a 3x3 * 3x1 float product
        // (products summed left to right, then the C term)
        float Cw[3], R2w[9], t2w[3], C2[3];
        camera_center(pKF1, Cw);
        rotation(pKF2, R2w);
        translation(pKF2, t2w);
        for (int r = 0; r < 3; r++) {
            volatile float a = R2w[3 * r] * Cw[0], b = R2w[3 * r + 1] * Cw[1], c = R2w[3 * r + 2] * Cw[2];
            volatile float ab = a + b;
            volatile float abc = ab + c;
            C2[r] = abc + t2w[r];
        }
        const float invz = 1.0f / C2[2];
        const float ex = pKF2->fx * C2[0] * invz + pKF2->cx;
        const float ey = pKF2->fy * C2[1] * invz + pKF2->cy;
        FrameSoA<KeyFrameT> s1(*pKF1, desc_row), s2(*pKF2, desc_row);
        std::vector<uint8_t> has1(std::max(pKF1->N, 1)), has2(std::max(pKF2->N, 1));
        for (int i = 0; i < pKF1->N; i++)
            has1[i] = pKF1->GetMapPoint(i) != nullptr;  // :702-705
        for (int i = 0; i < pKF2->N; i++)
            has2[i] = pKF2->GetMapPoint(i) != nullptr;  // :726-729
        const std::vector<int32_t> node1 = NodeIdsOf(pKF1->mFeatVec, pKF1->N), node2 = NodeIdsOf(pKF2->mFeatVec, pKF2->N);
        std::vector<int32_t> match12(std::max(pKF1->N, 1), -1);
        int32_t nmatches = 0;
        check(orbgpu_search_for_triangulation(&s1.view, has1.data(), node1.data(), &s2.view, has2.data(), node2.data(), F12, ex,
                                              ey, pKF2->mvLevelSigma2.data(), bOnlyStereo, mbCheckOrientation, match12.data(),
                                              &nmatches, device_),
              "SearchForTriangulation");
        vMatchedPairs.clear();  // :812-820
        vMatchedPairs.reserve((size_t)nmatches);
        for (int i = 0; i < pKF1->N; i++)
            if (match12[i] >= 0)
                vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match12[i]));
        return nmatches;
    }

    // int Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th=3.0)
    // (ORBmatcher.h:88, ORBmatcher.cc:825-975; LocalMapping::SearchInNeighbors, LocalMapping.cc:489, 514).
    // The candidate of every map point (projection, level and chi-square gates, best descriptor within TH_LOW) comes
    // from the device in one call; the edits of :946-969 -- Replace / AddObservation / AddMapPoint, which mutate the
    // pointer graph -- are applied here in index order exactly as the reference's loop does, re-checking isBad() and
    // IsInKeyFrame() since an earlier edit can change them.  pose(kf, float[16]) reads GetPose() (row-major Tcw);
    // world_pos / normal(pMP) return 3 floats, min_dist / max_dist the protected mfMinDistance / mfMaxDistance,
    // mp_desc(pMP) the 32 descriptor bytes (as for the Sim3 overload above).
    template <typename KeyFrameT, typename DescRow, typename PoseOf, typename MpDesc, typename WorldPos, typename Normal,
              typename MinDist, typename MaxDist>
    int Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th, DescRow desc_row, PoseOf pose,
             MpDesc mp_desc, WorldPos world_pos, Normal normal, MinDist min_dist, MaxDist max_dist)
    {
        FrameSoA<KeyFrameT> soa(*pKF, desc_row);
        const int m = (int)vpMapPoints.size();
        std::vector<uint8_t> bad(std::max(m, 1)), desc((size_t)std::max(m, 1) * 32);
        std::vector<float> wp((size_t)std::max(m, 1) * 3), nrm((size_t)std::max(m, 1) * 3), dmin(std::max(m, 1)), dmax(std::max(m, 1));
        for (int i = 0; i < m; i++) {
            MapPointT *p = vpMapPoints[i];
            bad[i] = !p || p->isBad() || p->IsInKeyFrame(pKF);  // :843-848
            if (!p)
                continue;
            const float *w = world_pos(p), *nn = normal(p);
            wp[3 * i] = w[0], wp[3 * i + 1] = w[1], wp[3 * i + 2] = w[2];
            nrm[3 * i] = nn[0], nrm[3 * i + 1] = nn[1], nrm[3 * i + 2] = nn[2];
            dmin[i] = min_dist(p), dmax[i] = max_dist(p);
            std::memcpy(&desc[(size_t)i * 32], mp_desc(p), 32);
        }
        orbgpu_points_view pv{m, bad.data(), wp.data(), nrm.data(), dmin.data(), dmax.data(), desc.data()};
        float Tcw[16];
        pose(pKF, Tcw);
        std::vector<int32_t> best(std::max(m, 1), -1);
        int32_t ncand = 0;
        check(orbgpu_fuse(&soa.view, Tcw, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf, pKF->mfLogScaleFactor, &pv, th,
                          pKF->mvInvLevelSigma2.data(), best.data(), &ncand, device_),
              "Fuse");
        int nFused = 0;
        for (int i = 0; i < m; i++) {
            MapPointT *pMP = vpMapPoints[i];
            if (best[i] < 0 || !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF))
                continue;
            MapPointT *pMPinKF = pKF->GetMapPoint(best[i]);  // :948
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations())
                        pMP->Replace(pMPinKF);
                    else
                        pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, best[i]);
                pKF->AddMapPoint(pMP, best[i]);
            }
            nFused++;
        }
        return nFused;
    }

    // void MapPoint::ComputeDistinctiveDescriptors()  (MapPoint.cc:242-307) for a batch of map points: groups[g] holds
    // the descriptor rows of the non-bad observing key frames of point g (in mObservations order); returns the index
    // of the chosen row per point (-1: no observation, mDescriptor stays).
    static std::vector<int32_t> ComputeDistinctiveDescriptors(const std::vector<std::vector<const uint8_t *>> &groups,
                                                              int device_id = 0)
    {
        std::vector<int32_t> off(groups.size() + 1, 0), best(std::max<size_t>(groups.size(), 1), -1);
        for (size_t g = 0; g < groups.size(); g++)
            off[g + 1] = off[g] + (int32_t)groups[g].size();
        std::vector<uint8_t> flat((size_t)std::max(off.back(), 1) * 32);
        for (size_t g = 0; g < groups.size(); g++)
            for (size_t i = 0; i < groups[g].size(); i++)
                std::memcpy(&flat[((size_t)off[g] + i) * 32], groups[g][i], 32);
        check(orbgpu_distinctive_descriptors((int32_t)groups.size(), off.data(), flat.data(), best.data(), device_id),
              "ComputeDistinctiveDescriptors");
        best.resize(groups.size());
        return best;
    }

  protected:
    // the common tail of the table flavours of SearchByProjection(F, vpMapPoints, th): ids in, list positions out
    int run_local(FrameT &F, const DeviceFrameT<FrameT> &dF, const std::vector<MapPointT *> &vp,
                  MapPointTableT<MapPointT> &table, const uint8_t *skip, const orbgpu_mappoint_view *scratch,
                  const float *Tcw, float fx, float fy, float cx, float cy, float mbf, float log_sf, float cos_limit, float th,
                  orbgpu_track_scratch *trk)
    {
        const int n = F.N, m = (int)vp.size();
        kid_.resize(n), k2m_.resize(std::max(n, 1));
        for (int j = 0; j < n; j++)
            kid_[j] = F.mvpMapPoints[j] ? (int64_t)F.mvpMapPoints[j]->mnId : -1;
        int32_t nmatches = 0;
        std::lock_guard<std::mutex> table_lock(table.mutex());
        check(orbgpu_search_local_points_table(dF.handle(), table.handle(), m, ids_.data(), skip, scratch, Tcw, fx, fy, cx, cy,
                                               mbf, log_sf, cos_limit, th, mfNNratio, kid_.data(), k2m_.data(), &nmatches, trk),
              "SearchByProjection(table)");
        for (int j = 0; j < n; j++)
            if (k2m_[j] >= 0 && F.mvpMapPoints[j] != vp[k2m_[j]])
                F.mvpMapPoints[j] = vp[k2m_[j]];  // ORBmatcher.cc:123
        return nmatches;
    }

    float mfNNratio;
    bool mbCheckOrientation;
    int device_;
    // per-call staging, kept between calls (a matcher that lives on the stack per call site, as in the reference, pays
    // these allocations every time; Tracking can keep one matcher per thread instead)
    std::vector<int64_t> ids_, kid_;
    std::vector<int32_t> k2m_, i0_;
    std::vector<uint8_t> b0_, b1_;
    std::vector<float> f0_, f1_, f2_, f3_;
};

// --------------------------------------------------------------------------------------------
// PointCloudMapping (reference include/PointCloudMap.h:41-88, src/PointCloudMap.cc)
// Keeps the reference's thread / condition-variable protocol; the per-key-frame arithmetic runs on
// the GPU.  KeyFrameT needs mImDep (float depth), mImRGB (8UC3), fx, fy, cx, cy and GetPose();
// the Adapter adapts cv::Mat (or a stand-in): depth(kf), rgb(kf) -> ImageView, pose(kf, float[16]),
// fx/fy/cx/cy(kf), and -- only for the loop-closure branch -- id(kf) (mnId) and isBad(kf).
// Visualisation stays out; the outlier filter of the shutdown pass is orbgpu_cloud_remove_outliers, the PCD writer
// orbgpu_cloud_save_pcd.
//
// viewer() follows PointCloudMap.cc:182-289 branch by branch:
//   * wait for key frames (:195-198) -- with a predicate: the reference's bare wait() can lose a wake-up and, woken
//     by shutdown() with no new key frame, indexes keyframes[N] (:246);
//   * loop closure (:217-243): when the hook reports LoopClosing::loop_detected (and clears it, :219), the map is
//     rebuilt from all non-bad key frames of the Map sorted by id;
//   * otherwise (:244-267) the new key frames are inserted.  The reference transforms only the LAST new cloud
//     (keyFrameCloud.back(), :247) with the pose of the FIRST new key frame (keyframes[lastKeyframeSize], :246) and,
//     in the loop branch, does not advance lastKeyframeSize (so the next insert takes the pose of the first key frame
//     that arrived with the loop closure).  That IS the default here -- the drop-in's map equals the reference's for the
//     same grouping of key frames into wake-ups.  setReferenceQuirks(false) is the corrected opt-in: every new key
//     frame is inserted with its own pose and the loop branch consumes its key frames.  After a loop branch the
//     reference only runs again on the next notify, i.e. the next insertKeyFrame: the wait predicate therefore
//     compares against the number of key frames SEEN (seenSize), not against the stale lastKeyframeSize;
//   * after the loop (:270-288): clear, per key frame generatePointCloud + voxel.filter + `+=`, then sor.filter
//     (meanK 50, stddev factor 1.0: the constructor's settings, :46-47) and, if an output path is set,
//     savePCDFileBinary.  A map of meanK points or fewer is left unfiltered (PCL reads past its neighbour list there).
// --------------------------------------------------------------------------------------------
struct ImageView {
    const void *data;
    int rows, cols;
    size_t step;  // bytes
};

template <typename KeyFrameT, typename Adapter> class PointCloudMappingT {
  public:
    struct LoopHooks {
        // returns true once per detected loop: `if (loopCloser->loop_detected) { loopCloser->loop_detected = false; ...`
        std::function<bool()> take_loop_detected;
        // loopCloser->getMap()->GetAllKeyFrames()
        std::function<std::vector<KeyFrameT *>()> all_keyframes;
    };

    PointCloudMappingT(double resolution_, Adapter adapter = Adapter(), int device_id = 0, LoopHooks hooks_ = LoopHooks())
        : resolution(resolution_), adapt(adapter), hooks(hooks_)
    {
        check(orbgpu_cloud_create(resolution, device_id, &cloud_), "PointCloudMapping");
        viewerThread = std::make_shared<std::thread>(&PointCloudMappingT::viewer, this);  // PointCloudMap.cc:53
    }
    ~PointCloudMappingT()
    {
        if (viewerThread && viewerThread->joinable())
            shutdown();
        orbgpu_cloud_destroy(cloud_);
    }

    // true (default): PointCloudMap.cc:217-262 statement by statement; false: the corrected insert (own pose per key
    // frame, loop branch advances lastKeyframeSize).  Set before the first insertKeyFrame.
    void setReferenceQuirks(bool on)
    {
        std::unique_lock<std::mutex> lck(keyframeMutex);
        quirks = on;
    }
    // "optimized_pointcloud.pcd" in the reference (:287); empty (default) = do not write a file
    void setOutputPath(const std::string &path) { outputPath = path; }
    // the reference fixes these in its constructor (:46-47); meanK <= 0 skips the filter
    void setOutlierFilter(int meanK, double stddevMul) { sorMeanK = meanK, sorStddevMul = stddevMul; }

    void insertKeyFrame(KeyFrameT *kf)  // PointCloudMap.cc:69-76
    {
        std::unique_lock<std::mutex> lck(keyframeMutex);
        keyframes.push_back(kf);
        keyFrameUpdated.notify_one();
    }

    // blocks until the viewer has consumed every key frame / loop notification handed over so far (not in the
    // reference: there the only way to know is the console output)
    void waitProcessed()
    {
        std::unique_lock<std::mutex> lck(keyframeMutex);
        processed.wait(lck, [&] { return finished || (!busy && !loopPending && keyframes.size() <= seenSize); });
    }

    // a detected loop must wake the viewer too (the reference relies on the next key frame's notify)
    void notifyLoop()
    {
        std::unique_lock<std::mutex> lck(keyframeMutex);
        loopPending = true;
        keyFrameUpdated.notify_one();
    }

    void shutdown()  // PointCloudMap.cc:59-67
    {
        {
            // The flag is published under keyframeMutex, the mutex the condition variable waits on: a notify
            // between the viewer's predicate evaluation and its block cannot be lost (the reference sets it under
            // shutDownMutex only and can hang in join()).
            std::unique_lock<std::mutex> lck(keyframeMutex);
            {
                std::unique_lock<std::mutex> lck2(shutDownMutex);
                shutDownFlag = true;
            }
            keyFrameUpdated.notify_one();
        }
        if (viewerThread->joinable())
            viewerThread->join();
    }

    void viewer()  // PointCloudMap.cc:182-289
    {
        while (true) {
            if (shutDownFlagLocked())  // :186-192
                break;
            size_t N, first;
            bool ref;
            {
                std::unique_lock<std::mutex> lck(keyframeMutex);  // :195-198
                keyFrameUpdated.wait(lck, [&] { return shutDownFlagLocked() || loopPending || keyframes.size() > seenSize; });
                loopPending = false;
                busy = true;
                N = keyframes.size();  // :202-205
                first = lastKeyframeSize;
                seenSize = N;
                ref = quirks;
            }
            size_t done = first;
            if (hooks.take_loop_detected && hooks.all_keyframes && hooks.take_loop_detected()) {  // :217-243
                std::vector<KeyFrameT *> all = hooks.all_keyframes();
                std::sort(all.begin(), all.end(), [&](KeyFrameT *a, KeyFrameT *b) { return adapt.id(a) < adapt.id(b); });
                std::vector<KeyFrameT *> good;
                for (KeyFrameT *kf : all)
                    if (!adapt.isBad(kf))
                        good.push_back(kf);
                rebuild(good);
                if (!ref)
                    done = N;  // the reference leaves it (:243 falls out of the branch without `lastKeyframeSize = N`)
            } else if (N > first) {  // :244-267
                if (ref)
                    insertOne(keyframeAt(N - 1), keyframeAt(first));  // keyFrameCloud.back() with keyframes[lastKeyframeSize]'s pose (:246-247)
                else
                    for (size_t i = first; i < N; i++)
                        insertOne(keyframeAt(i), keyframeAt(i));
                done = N;
            }
            {
                std::unique_lock<std::mutex> lck(keyframeMutex);
                lastKeyframeSize = done;
                busy = false;
                processed.notify_all();
            }
        }
        {
            std::unique_lock<std::mutex> lck(keyframeMutex);
            busy = false;
            finished = true;
            processed.notify_all();
        }
        // :270-288: the down-sampled clouds of all key frames, one by one
        std::vector<KeyFrameT *> kfs;
        {
            std::unique_lock<std::mutex> lck(keyframeMutex);
            kfs = keyframes;
        }
        {
            std::unique_lock<std::mutex> lck(cloudMutex);
            check(orbgpu_cloud_clear(cloud_), "PointCloudMapping::shutdown");
            for (KeyFrameT *kf : kfs) {
                ImageView d = adapt.depth(kf), c = adapt.rgb(kf);
                float Tcw[16];
                adapt.pose(kf, Tcw);
                check(orbgpu_cloud_append_filtered(cloud_, (const float *)d.data, d.step / sizeof(float),
                                                   (const uint8_t *)c.data, c.step, d.cols, d.rows, adapt.fx(kf),
                                                   adapt.fy(kf), adapt.cx(kf), adapt.cy(kf), Tcw),
                      "PointCloudMapping::shutdown");
            }
            int64_t npts = 0;
            check(orbgpu_cloud_size(cloud_, &npts), "PointCloudMapping::shutdown");
            if (sorMeanK > 0 && npts > sorMeanK)  // :283-285
                check(orbgpu_cloud_remove_outliers(cloud_, sorMeanK, sorStddevMul, nullptr), "PointCloudMapping::shutdown");
            if (!outputPath.empty())
                check(orbgpu_cloud_save_pcd(cloud_, outputPath.c_str()), "PointCloudMapping::save");
        }
    }

    // loop-closure branch (PointCloudMap.cc:217-243): regenerate every key frame with its new pose
    void rebuild(const std::vector<KeyFrameT *> &kfs)
    {
        std::vector<const float *> depth, Tcw;
        std::vector<const uint8_t *> rgb;
        std::vector<std::vector<float>> poses;
        if (kfs.empty())
            return;
        ImageView d0 = adapt.depth(kfs[0]), c0 = adapt.rgb(kfs[0]);
        for (KeyFrameT *kf : kfs) {
            depth.push_back((const float *)adapt.depth(kf).data);
            rgb.push_back((const uint8_t *)adapt.rgb(kf).data);
            poses.emplace_back(16);
            adapt.pose(kf, poses.back().data());
        }
        for (auto &p : poses)
            Tcw.push_back(p.data());
        std::unique_lock<std::mutex> lck(cloudMutex);
        check(orbgpu_cloud_rebuild(cloud_, (int)kfs.size(), depth.data(), d0.step / sizeof(float), rgb.data(), c0.step,
                                   d0.cols, d0.rows, adapt.fx(kfs[0]), adapt.fy(kfs[0]), adapt.cx(kfs[0]),
                                   adapt.cy(kfs[0]), Tcw.data()),
              "PointCloudMapping::rebuild");
    }

    // globalMapRGBD (PointCloudMap.h:55)
    std::vector<orbgpu_point_xyzrgba> globalMap()
    {
        std::unique_lock<std::mutex> lck(cloudMutex);
        int64_t n = 0;
        check(orbgpu_cloud_size(cloud_, &n), "size");
        std::vector<orbgpu_point_xyzrgba> out((size_t)std::max<int64_t>(n, 1));
        check(orbgpu_cloud_download(cloud_, out.data(), (int64_t)out.size(), &n), "download");
        out.resize((size_t)n);
        return out;
    }

  protected:
    bool shutDownFlagLocked()
    {
        std::unique_lock<std::mutex> lck(shutDownMutex);
        return shutDownFlag;
    }
    KeyFrameT *keyframeAt(size_t i)
    {
        std::unique_lock<std::mutex> lck(keyframeMutex);
        return keyframes[i];
    }
    void insertOne(KeyFrameT *image_kf, KeyFrameT *pose_kf)
    {
        ImageView d = adapt.depth(image_kf), c = adapt.rgb(image_kf);
        float Tcw[16];
        adapt.pose(pose_kf, Tcw);
        std::unique_lock<std::mutex> lck(cloudMutex);
        check(orbgpu_cloud_insert(cloud_, (const float *)d.data, d.step / sizeof(float), (const uint8_t *)c.data, c.step,
                                  d.cols, d.rows, adapt.fx(image_kf), adapt.fy(image_kf), adapt.cx(image_kf),
                                  adapt.cy(image_kf), Tcw),
              "PointCloudMapping::insertKeyFrame");
    }

    orbgpu_cloud *cloud_ = nullptr;
    std::shared_ptr<std::thread> viewerThread;
    bool shutDownFlag = false, loopPending = false, quirks = true, busy = false, finished = false;
    std::mutex shutDownMutex, keyframeMutex, cloudMutex;
    std::condition_variable keyFrameUpdated, processed;
    std::vector<KeyFrameT *> keyframes;
    size_t lastKeyframeSize = 0;  // the reference's member (:207, :265)
    size_t seenSize = 0;          // keyframes.size() at the last wake-up (what a bare wait() + notify per insert amounts to)
    double resolution = 0.01;
    int sorMeanK = 50;          // sor.setMeanK(50), PointCloudMap.cc:46
    double sorStddevMul = 1.0;  // sor.setStddevMulThresh(1.0), :47
    Adapter adapt;
    LoopHooks hooks;
    std::string outputPath;
};

} // namespace orbgpu_shim
