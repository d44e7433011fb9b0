// Exercises the C++ shim (orb_slam2_map_amd/shim/orbgpu_shim.hpp) with stand-ins that carry the
// reference's member names (Frame.h / MapPoint.h / KeyFrame.h).  Reads a scenario file written by
// tests/test_shim.py, runs extractor -> SearchByProjection x2 -> PointCloudMapping through the shim
// and writes the results for comparison with the oracle.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "orbgpu_shim.hpp"

struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };  // cv::KeyPoint layout

struct MapPoint {  // members read by the matcher: MapPoint.h:91-96 + accessors
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool bad = false;
    int nObs = 1;
    float world[3] = {0, 0, 0}, normal[3] = {0, 0, 1}, minDist = 0, maxDist = 0;
    uint8_t desc[32];
    int id = -1;
    long unsigned int mnId = 0;  // MapPoint.h:84
    bool isBad() const { return bad; }
    int Observations() const { return nObs; }
    // the part of the pointer graph ORBmatcher::Fuse edits (MapPoint.h: AddObservation, Replace, IsInKeyFrame), reduced
    // to one key frame: which key point of it observes this point, and who replaced it
    int idxInKF = -1;
    MapPoint *replacedBy = nullptr;
    template <typename K> bool IsInKeyFrame(K *) const { return idxInKF >= 0; }
    template <typename K> void AddObservation(K *, size_t idx)
    {
        idxInKF = (int)idx;
        nObs++;
    }
    void Replace(MapPoint *pMP);  // defined after MatchKeyFrame
};

struct Frame {  // Frame.h:100-190
    int N = 0;
    std::vector<KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    std::vector<uint8_t> mDescriptors;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<size_t> mGrid[orbgpu_shim::FRAME_GRID_COLS][orbgpu_shim::FRAME_GRID_ROWS];
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    std::vector<float> mvScaleFactors;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0, mfLogScaleFactor = 0;
    float mTcw[16];
    orbgpu_shim::FeatureVector mFeatVec;
    orbgpu_shim::BowVector mBowVec;
};

struct MatchKeyFrame : Frame {  // the KeyFrame members the background matchers read (KeyFrame.h)
    std::vector<float> mvLevelSigma2, mvInvLevelSigma2;
    MapPoint *GetMapPoint(size_t idx) const { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *p, size_t idx) { mvpMapPoints[idx] = p; }
};
static MatchKeyFrame *g_fuse_kf = nullptr;  // the one key frame of the stand-in graph
inline void MapPoint::Replace(MapPoint *pMP)  // MapPoint.cc:171-212, for the one-key-frame graph
{
    if (pMP == this)
        return;
    if (idxInKF >= 0 && g_fuse_kf) {
        if (!pMP->IsInKeyFrame(g_fuse_kf)) {
            g_fuse_kf->mvpMapPoints[idxInKF] = pMP;  // ReplaceMapPointMatch + AddObservation
            pMP->idxInKF = idxInKF;
            pMP->nObs++;
        } else
            g_fuse_kf->mvpMapPoints[idxInKF] = nullptr;  // EraseMapPointMatch
    }
    idxInKF = -1;
    nObs = 0;
    bad = true;
    replacedBy = pMP;
}

struct KeyFrame {
    std::vector<float> mImDep;
    std::vector<uint8_t> mImRGB;
    int rows = 0, cols = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    float pose[16];
    long unsigned int mnId = 0;
    bool bad = false;
};

struct BowKeyFrame {  // the KeyFrame members SearchByBoW reads (KeyFrame.h)
    int N = 0;
    std::vector<KeyPoint> mvKeysUn;
    std::vector<uint8_t> mDescriptors;
    std::vector<MapPoint *> mvpMapPoints;
    orbgpu_shim::FeatureVector mFeatVec;
    orbgpu_shim::BowVector mBowVec;
    std::vector<MapPoint *> GetMapPointMatches() const { return mvpMapPoints; }
};
struct KFAdapter {
    orbgpu_shim::ImageView depth(KeyFrame *k) const { return {k->mImDep.data(), k->rows, k->cols, (size_t)k->cols * 4}; }
    orbgpu_shim::ImageView rgb(KeyFrame *k) const { return {k->mImRGB.data(), k->rows, k->cols, (size_t)k->cols * 3}; }
    void pose(KeyFrame *k, float *T) const { std::memcpy(T, k->pose, 64); }
    float fx(KeyFrame *k) const { return k->fx; }
    float fy(KeyFrame *k) const { return k->fy; }
    float cx(KeyFrame *k) const { return k->cx; }
    float cy(KeyFrame *k) const { return k->cy; }
    long unsigned int id(KeyFrame *k) const { return k->mnId; }
    bool isBad(KeyFrame *k) const { return k->bad; }
};

template <typename T> static void rd(std::ifstream &f, T *p, size_t n) { f.read(reinterpret_cast<char *>(p), sizeof(T) * n); }
template <typename T> static void wr(std::ofstream &f, const T *p, size_t n) { f.write(reinterpret_cast<const char *>(p), sizeof(T) * n); }

// Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:230-245, 382-392) for the stand-in
static void assign_grid(Frame &F)
{
    for (int i = 0; i < F.N; i++) {
        const int px = (int)roundf((F.mvKeysUn[i].pt.x - F.mnMinX) * F.mfGridElementWidthInv);
        const int py = (int)roundf((F.mvKeysUn[i].pt.y - F.mnMinY) * F.mfGridElementHeightInv);
        if (px < 0 || px >= orbgpu_shim::FRAME_GRID_COLS || py < 0 || py >= orbgpu_shim::FRAME_GRID_ROWS)
            continue;
        F.mGrid[px][py].push_back(i);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: shim_test <scenario.bin> <out.bin> [--compile-only]\n");
        return 2;
    }
    try {
        std::ifstream in(argv[1], std::ios::binary);
        std::ofstream out(argv[2], std::ios::binary);
        int32_t hdr[4];  // w, h, nfeatures, n_map
        rd(in, hdr, 4);
        const int w = hdr[0], h = hdr[1], nfeat = hdr[2], m = hdr[3];
        float cam[7];  // fx fy cx cy bf th, mfLogScaleFactor as the scenario's isInFrustum used it
        rd(in, cam, 7);
        std::vector<uint8_t> gray((size_t)w * h), rgb((size_t)w * h * 3);
        std::vector<float> depth((size_t)w * h);
        rd(in, gray.data(), gray.size());
        rd(in, rgb.data(), rgb.size());
        rd(in, depth.data(), depth.size());
        float Tcw[16];
        rd(in, Tcw, 16);

        // ---- extractor through the reference's interface
        orbgpu_shim::ORBextractorT<KeyPoint> extractor(nfeat, 1.2f, 8, 20, 7);
        Frame F;
        extractor(gray.data(), h, w, (size_t)w, F.mvKeys, F.mDescriptors);
        F.N = (int)F.mvKeys.size();
        F.mvKeysUn = F.mvKeys;  // zero distortion (Frame.cc:406-410)
        F.mvScaleFactors = extractor.GetScaleFactors();
        F.mnMinX = 0, F.mnMaxX = (float)w, F.mnMinY = 0, F.mnMaxY = (float)h;
        F.mfGridElementWidthInv = (float)orbgpu_shim::FRAME_GRID_COLS / (F.mnMaxX - F.mnMinX);
        F.mfGridElementHeightInv = (float)orbgpu_shim::FRAME_GRID_ROWS / (F.mnMaxY - F.mnMinY);
        F.fx = cam[0], F.fy = cam[1], F.cx = cam[2], F.cy = cam[3], F.mbf = cam[4], F.mb = cam[4] / cam[0];
        std::memcpy(F.mTcw, Tcw, 64);
        F.mvuRight.assign(F.N, -1.f);
        for (int i = 0; i < F.N; i++) {  // Frame::ComputeStereoFromRGBD (Frame.cc:641-662)
            const float d = depth[(size_t)(int)F.mvKeys[i].pt.y * w + (int)F.mvKeys[i].pt.x];
            if (d > 0)
                F.mvuRight[i] = F.mvKeysUn[i].pt.x - F.mbf / d;
        }
        F.mvpMapPoints.assign(F.N, nullptr);
        F.mvbOutlier.assign(F.N, false);
        assign_grid(F);
        int32_t n = F.N;
        wr(out, &n, 1);
        wr(out, F.mvKeys.data(), F.mvKeys.size());
        wr(out, F.mDescriptors.data(), F.mDescriptors.size());

        // ---- local map points with pre-filled tracking scratch
        std::vector<MapPoint> mps(m);
        std::vector<MapPoint *> vp(m);
        for (int i = 0; i < m; i++) {
            MapPoint &p = mps[i];
            uint8_t flags[3];
            int32_t lvl;
            float f4[4];
            rd(in, flags, 3);
            rd(in, &lvl, 1);
            rd(in, f4, 4);
            rd(in, p.world, 3);
            rd(in, p.normal, 3);
            rd(in, &p.minDist, 1);
            rd(in, &p.maxDist, 1);
            rd(in, p.desc, 32);
            p.mbTrackInView = flags[0], p.bad = flags[1], p.nObs = flags[2] ? 1 : 0;
            p.mnTrackScaleLevel = lvl, p.mTrackViewCos = f4[0], p.mTrackProjX = f4[1], p.mTrackProjY = f4[2],
            p.mTrackProjXR = f4[3];
            p.id = i;
            p.mnId = 1000003ul * (unsigned long)((i * 7919) % m) + 17ul;  // sparse, shuffled ids (7919 and m are coprime or not: see below)
            vp[i] = &p;
        }
        {  // ids must be distinct: fall back to a plain affine map when 7919 shares a factor with m
            std::vector<unsigned long> seen;
            for (int i = 0; i < m; i++)
                seen.push_back(mps[i].mnId);
            std::sort(seen.begin(), seen.end());
            if (std::adjacent_find(seen.begin(), seen.end()) != seen.end())
                for (int i = 0; i < m; i++)
                    mps[i].mnId = 1000003ul * (unsigned long)i + 17ul;
        }
        auto desc_row = [](const Frame &fr, int i) { return &fr.mDescriptors[(size_t)i * 32]; };
        auto mp_desc = [](MapPoint *p) { return p->desc; };
        orbgpu_shim::ORBmatcherT<Frame, MapPoint> matcher(0.8f, true);
        const int nm = matcher.SearchByProjection(F, vp, cam[5], desc_row, mp_desc);
        int32_t nm32 = nm;
        wr(out, &nm32, 1);
        for (int j = 0; j < F.N; j++) {
            int32_t id = F.mvpMapPoints[j] ? F.mvpMapPoints[j]->id : -1;
            wr(out, &id, 1);
        }
        const int dist = orbgpu_shim::ORBmatcherT<Frame, MapPoint>::DescriptorDistance(mps[0].desc, mps[m - 1].desc);
        int32_t d32 = dist;
        wr(out, &d32, 1);

        // ---- the same search over the device-resident MapPoint table (ids instead of objects), both flavours, and
        //      SearchByProjection(Cur, Last) with both frames on the device; host time per call of every flavour
        {
            // (a map point seen at exactly its creation distance predicts level l or l + 1 depending on the last bit of
            //  this value: the scenario's own figure keeps the stand-in and the oracle on the same side)
            F.mfLogScaleFactor = cam[6];
            using clk = std::chrono::steady_clock;
            auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            orbgpu_shim::MapPointTableT<MapPoint> table(0, 256);
            auto wp_of = [](MapPoint *p) { return p->world; };
            auto nr_of = [](MapPoint *p) { return p->normal; };
            auto mn_of = [](MapPoint *p) { return p->minDist; };
            auto mx_of = [](MapPoint *p) { return p->maxDist; };
            const auto t_up0 = clk::now();
            for (int a = 0; a < m; a += 1500) {  // key frame by key frame, as the map grows
                std::vector<MapPoint *> part(vp.begin() + a, vp.begin() + std::min(a + 1500, m));
                table.Upsert(part, wp_of, nr_of, mn_of, mx_of, mp_desc);
            }
            const double up_us = us(t_up0, clk::now());
            if (table.rows() != m)
                throw std::runtime_error("table rows != map points");
            Frame F2 = F;
            orbgpu_shim::DeviceFrameT<Frame> dF;
            dF.Upload(F2, desc_row);  // (the first call allocates the pinned and the device block: 0.1 ms once)
            const auto t_f0 = clk::now();
            dF.Upload(F2, desc_row);
            const double frame_us = us(t_f0, clk::now());
            const int reps = 20;
            double t_host = 0, t_tab = 0, t_dev = 0;
            int nm_tab = 0, nm_dev = 0, nseen = 0;
            std::vector<int32_t> ids_tab(F.N), ids_dev(F.N);
            std::vector<uint8_t> seen(m);
            for (int rep = 0; rep < reps; rep++) {
                F2.mvpMapPoints.assign(F2.N, nullptr);
                auto t0 = clk::now();
                const int nh = matcher.SearchByProjection(F2, vp, cam[5], desc_row, mp_desc);
                t_host += us(t0, clk::now());
                if (nh != nm)
                    throw std::runtime_error("host-pointer SearchByProjection is not repeatable");
                F2.mvpMapPoints.assign(F2.N, nullptr);
                t0 = clk::now();
                nm_tab = matcher.SearchByProjection(F2, dF, vp, cam[5], table);
                t_tab += us(t0, clk::now());
                for (int j = 0; j < F2.N; j++)
                    ids_tab[j] = F2.mvpMapPoints[j] ? F2.mvpMapPoints[j]->id : -1;
                F2.mvpMapPoints.assign(F2.N, nullptr);
                t0 = clk::now();
                nm_dev = matcher.SearchLocalPoints(
                    F2, dF, vp, cam[5], table, [](MapPoint *) { return false; }, [](const Frame &fr) { return fr.mTcw; },
                    [&](MapPoint *p, bool v) { seen[p->id] = v; }, &nseen);
                t_dev += us(t0, clk::now());
                for (int j = 0; j < F2.N; j++)
                    ids_dev[j] = F2.mvpMapPoints[j] ? F2.mvpMapPoints[j]->id : -1;
            }
            int32_t v32 = nm_tab;
            wr(out, &v32, 1);
            wr(out, ids_tab.data(), ids_tab.size());
            v32 = nm_dev;
            wr(out, &v32, 1);
            wr(out, ids_dev.data(), ids_dev.size());
            v32 = nseen;
            wr(out, &v32, 1);
            wr(out, seen.data(), seen.size());
            std::printf("shim timing (host wall time per call, %d map points, %d key points): table upload %.0f us (once), "
                        "frame upload %.0f us (per frame), SearchByProjection host-pointer %.0f us, over the table %.0f us, "
                        "SearchLocalPoints on the device %.0f us\n",
                        m, F.N, up_us, frame_us, t_host / reps, t_tab / reps, t_dev / reps);

            // last frame = the key points the first n_last map points were made from
            int32_t n_last = 0;
            rd(in, &n_last, 1);
            Frame L;
            L.N = n_last;
            L.mvKeys.resize(n_last), L.mvKeysUn.resize(n_last), L.mvuRight.assign(n_last, -1.f);
            L.mDescriptors.resize((size_t)n_last * 32);
            L.mvpMapPoints.assign(n_last, nullptr), L.mvbOutlier.assign(n_last, false);
            for (int i = 0; i < n_last; i++) {
                float xy[2], ang;
                int32_t oct;
                rd(in, xy, 2), rd(in, &oct, 1), rd(in, &ang, 1);
                L.mvKeys[i] = KeyPoint{{xy[0], xy[1]}, 31.f, ang, 0.f, oct, -1};
                L.mvKeysUn[i] = L.mvKeys[i];
                std::memcpy(&L.mDescriptors[(size_t)i * 32], mps[i].desc, 32);
                L.mvpMapPoints[i] = (i % 5 == 4) ? nullptr : &mps[i];
                L.mvbOutlier[i] = i % 17 == 0;
            }
            L.mvScaleFactors = F.mvScaleFactors;
            L.mnMinX = F.mnMinX, L.mnMaxX = F.mnMaxX, L.mnMinY = F.mnMinY, L.mnMaxY = F.mnMaxY;
            L.mfGridElementWidthInv = F.mfGridElementWidthInv, L.mfGridElementHeightInv = F.mfGridElementHeightInv;
            L.fx = F.fx, L.fy = F.fy, L.cx = F.cx, L.cy = F.cy, L.mbf = F.mbf, L.mb = F.mb;
            std::memcpy(L.mTcw, Tcw, 64);
            assign_grid(L);
            orbgpu_shim::DeviceFrameT<Frame> dL;
            dL.Upload(L, desc_row);
            auto tcw_of = [](const Frame &fr) { return fr.mTcw; };
            double t_lh = 0, t_lt = 0;
            int nl_host = 0, nl_tab = 0;
            std::vector<int32_t> idl_host(F.N), idl_tab(F.N);
            for (int rep = 0; rep < reps; rep++) {
                F2.mvpMapPoints.assign(F2.N, nullptr);  // Tracking.cc:1166
                auto t0 = clk::now();
                nl_host = matcher.SearchByProjection(F2, L, 15.f, false, desc_row, mp_desc, tcw_of, wp_of);
                t_lh += us(t0, clk::now());
                for (int j = 0; j < F2.N; j++)
                    idl_host[j] = F2.mvpMapPoints[j] ? F2.mvpMapPoints[j]->id : -1;
                F2.mvpMapPoints.assign(F2.N, nullptr);
                t0 = clk::now();
                nl_tab = matcher.SearchByProjection(F2, dF, L, dL, 15.f, false, table, tcw_of);
                t_lt += us(t0, clk::now());
                for (int j = 0; j < F2.N; j++)
                    idl_tab[j] = F2.mvpMapPoints[j] ? F2.mvpMapPoints[j]->id : -1;
            }
            v32 = nl_host;
            wr(out, &v32, 1);
            wr(out, idl_host.data(), idl_host.size());
            v32 = nl_tab;
            wr(out, &v32, 1);
            wr(out, idl_tab.data(), idl_tab.size());
            std::printf("shim timing: SearchByProjection(Cur, Last) host-pointer %.0f us, over the table %.0f us (%d last-frame key points)\n",
                        t_lh / reps, t_lt / reps, n_last);
            // LocalMapping edits the map while Tracking searches: a second thread inserts points (the table grows under the
            // searches), changes their observation counts and flags some bad -- none of them is in the list, so every search
            // must give the same matches; the wrapper's lock makes the two threads take turns
            {
                std::vector<MapPoint> extra(6000);
                for (size_t i = 0; i < extra.size(); i++) {
                    extra[i].mnId = 4000000007ul + 13ul * i;
                    extra[i].world[2] = 1.f + (float)i;
                    std::memset(extra[i].desc, (int)(i & 255), 32);
                }
                std::atomic<bool> stop{false};
                std::atomic<int> edits{0};
                std::string edit_error;
                std::thread editor([&] {
                    try {
                        size_t next = 0;
                        while (!stop.load()) {
                            std::vector<MapPoint *> part;
                            for (int k = 0; k < 500 && next < extra.size(); k++)
                                part.push_back(&extra[next++]);
                            if (!part.empty())
                                table.Upsert(part, wp_of, nr_of, mn_of, mx_of, mp_desc);
                            MapPoint &p = extra[(size_t)edits.load() % extra.size()];
                            p.nObs = edits.load() % 3;
                            if (p.mnId < 4000000007ul + 13ul * next) {  // only points the table has been told about
                                table.SetObservations(&p);
                                if (edits.load() % 7 == 0)
                                    table.SetBad(&p);
                            }
                            edits++;
                        }
                    } catch (const std::exception &e) {
                        edit_error = e.what();
                    }
                });
                for (int rep = 0; rep < 40 || (edits.load() < 3 && rep < 4000); rep++) {  // (at least 40; until the editor had turns)
                    F2.mvpMapPoints.assign(F2.N, nullptr);
                    const int nmt = matcher.SearchLocalPoints(
                        F2, dF, vp, cam[5], table, [](MapPoint *) { return false; }, tcw_of, [](MapPoint *, bool) {});
                    if (nmt != nm_dev) {
                        stop = true;
                        editor.join();
                        throw std::runtime_error("a search gave other matches while another thread edited the table");
                    }
                }
                stop = true;
                editor.join();
                if (!edit_error.empty())
                    throw std::runtime_error("editor thread: " + edit_error);
                if (edits.load() < 1 || table.rows() <= m)
                    throw std::runtime_error("the editor thread never got a turn in 4000 searches");
                std::printf("table edited %d times by a second thread during the searches (%d rows)\n", edits.load(), table.rows());
            }
            // LocalMapping publishes a new point to the key frames (LocalMapping.cc:434-440) before it reaches the table:
            // Tracking::UpdateLocalPoints may list it in that window.  The search must go through (the row is skipped)
            // and say how many ids it did not know; after the upsert the count is back to zero.
            {
                std::vector<MapPoint> late(30);
                std::vector<MapPoint *> vp2 = vp, latep;
                for (size_t i = 0; i < late.size(); i++) {
                    late[i].mnId = 9000000001ul + 3ul * i;
                    late[i].world[2] = -5.f - (float)i;  // behind the camera: never in view once the table knows them
                    std::memset(late[i].desc, 0x5a, 32);
                    vp2.insert(vp2.begin() + (long)(7 * i), &late[i]);
                    latep.push_back(&late[i]);
                }
                F2.mvpMapPoints.assign(F2.N, nullptr);
                int nmt = matcher.SearchLocalPoints(F2, dF, vp2, cam[5], table, [](MapPoint *) { return false; }, tcw_of,
                                                    [](MapPoint *, bool) {});
                if (nmt != nm_dev || table.LastUnknown() != std::make_pair(30, 0))
                    throw std::runtime_error("ids published before their upsert must be skipped rows, counted");
                for (int j = 0; j < F2.N; j++)
                    if ((F2.mvpMapPoints[j] ? F2.mvpMapPoints[j]->id : -1) != ids_dev[j])
                        throw std::runtime_error("unknown ids in the list changed the matches");
                table.Upsert(latep, wp_of, nr_of, mn_of, mx_of, mp_desc);
                F2.mvpMapPoints.assign(F2.N, nullptr);
                nmt = matcher.SearchLocalPoints(F2, dF, vp2, cam[5], table, [](MapPoint *) { return false; }, tcw_of,
                                                [](MapPoint *, bool) {});
                if (nmt != nm_dev || table.LastUnknown() != std::make_pair(0, 0))
                    throw std::runtime_error("after the upsert no id is unknown");
                // the batch setters (bundle-adjustment write-back, culling) against the one-id forms: same table contents,
                // and what a one-id call costs on the host
                auto t0 = clk::now();
                for (MapPoint *p : latep)
                    table.SetWorldPos(p, p->world);
                const double one_us = us(t0, clk::now()) / (double)latep.size();
                for (MapPoint *p : latep)
                    p->world[0] += 1.f, p->nObs = 0;
                t0 = clk::now();
                table.SetWorldPos(latep, wp_of, nr_of, mn_of, mx_of);
                table.SetObservations(latep);
                table.SetBad(latep);
                const double batch_us = us(t0, clk::now());
                float wpos[3];
                int32_t hobs = -1, hbad = -1;
                orbgpu_shim::check(orbgpu_mappoint_table_read(table.handle(), (int64_t)late[7].mnId, wpos, nullptr, nullptr, nullptr, nullptr,
                                                 &hobs, &hbad), "read");
                if (wpos[0] != late[7].world[0] || hobs != 0 || hbad != 1)
                    throw std::runtime_error("batch setters did not reach the table");
                std::printf("table edits: %.0f us per one-id SetWorldPos call, %.0f us for three batch calls over %zu points\n",
                            one_us, batch_us, latep.size());
            }
            // edits reach the table: a point flagged bad is skipped by the next search
            if (nm_dev > 0) {
                int victim = -1;
                for (int j = 0; j < F.N && victim < 0; j++)
                    victim = ids_dev[j];
                mps[victim].bad = true;
                table.SetBad(&mps[victim]);
                F2.mvpMapPoints.assign(F2.N, nullptr);
                (void)matcher.SearchLocalPoints(
                    F2, dF, vp, cam[5], table, [](MapPoint *) { return false; }, tcw_of, [](MapPoint *, bool) {});
                for (int j = 0; j < F2.N; j++)
                    if (F2.mvpMapPoints[j] == &mps[victim])
                        throw std::runtime_error("a map point flagged bad in the table was matched");
                mps[victim].bad = false;  // the stand-ins go on to the other sections unchanged
            }
        }
        if (argc > 3 && std::string(argv[3]) == "--projection-only") {
            std::printf("shim ok (projection only): %d key points, %d projection matches\n", n, nm);
            return 0;
        }

        // ---- loop-closing projection (the stand-in Frame plays the key frame) and distinctive descriptors
        {
            // (the test driver computes the same value: double log rounded to float, so both sides agree to the bit)
            F.mfLogScaleFactor = (float)std::log((double)F.mvScaleFactors[1]);
            std::vector<MapPoint *> vpMatched(F.N, nullptr);
            float Scw[16];
            std::memcpy(Scw, Tcw, 64);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++)
                    Scw[4 * r + c] *= 1.5f;
            const int ns = matcher.SearchByProjection(
                &F, Scw, vp, vpMatched, 10, desc_row, mp_desc, [](MapPoint *p) { return p->world; },
                [](MapPoint *p) { return p->normal; }, [](MapPoint *p) { return p->minDist; },
                [](MapPoint *p) { return p->maxDist; });
            int32_t ns32 = ns;
            wr(out, &ns32, 1);
            for (int j = 0; j < F.N; j++) {
                int32_t id = vpMatched[j] ? vpMatched[j]->id : -1;
                wr(out, &id, 1);
            }
            std::vector<std::vector<const uint8_t *>> groups(3);
            for (int i = 0; i < m; i++)
                groups[i % 2].push_back(mps[i].desc);  // group 2 stays empty
            const std::vector<int32_t> best = orbgpu_shim::ORBmatcherT<Frame, MapPoint>::ComputeDistinctiveDescriptors(groups);
            wr(out, best.data(), best.size());
        }

        // ---- vocabulary + SearchByBoW (TrackReferenceKeyFrame): the scenario's vocabulary, the first n_kf map points
        //      play the key frame's features (every key point has its map point), the extracted frame is F
        {
            int32_t vk = 0, vL = 0, vn = 0, n_kf = 0;
            rd(in, &vk, 1), rd(in, &vL, 1), rd(in, &vn, 1), rd(in, &n_kf, 1);
            std::vector<int32_t> parent(vn);
            std::vector<uint8_t> leaf(vn), vdesc((size_t)vn * 32);
            std::vector<double> weight(vn);
            rd(in, parent.data(), vn), rd(in, leaf.data(), vn), rd(in, vdesc.data(), vdesc.size()), rd(in, weight.data(), vn);
            orbgpu_shim::ORBVocabularyT voc(vk, vL, parent, leaf, vdesc, weight);
            BowKeyFrame kfb;
            kfb.N = n_kf;
            kfb.mvKeysUn.resize(n_kf);
            kfb.mDescriptors.resize((size_t)n_kf * 32);
            kfb.mvpMapPoints.resize(n_kf);
            for (int i = 0; i < n_kf; i++) {
                kfb.mvKeysUn[i] = KeyPoint{{0, 0}, 31, 0, 0, 0, -1};
                std::memcpy(&kfb.mDescriptors[(size_t)i * 32], mps[i].desc, 32);
                kfb.mvpMapPoints[i] = &mps[i];
            }
            voc.transform(kfb.mDescriptors.data(), kfb.N, kfb.mBowVec, kfb.mFeatVec, 2);  // KeyFrame::ComputeBoW
            voc.transform(F.mDescriptors.data(), F.N, F.mBowVec, F.mFeatVec, 2);          // Frame::ComputeBoW
            orbgpu_shim::ORBmatcherT<Frame, MapPoint> bm(0.7f, false);
            std::vector<MapPoint *> vpMatches;
            const int nb = bm.SearchByBoW(
                &kfb, F, vpMatches, [](const BowKeyFrame &k, int i) { return &k.mDescriptors[(size_t)i * 32]; }, desc_row);
            int32_t nb32 = nb, nbow = (int32_t)F.mBowVec.size();
            wr(out, &nb32, 1);
            for (int j = 0; j < F.N; j++) {
                int32_t id = vpMatches[j] ? vpMatches[j]->id : -1;
                wr(out, &id, 1);
            }
            wr(out, &nbow, 1);
            for (const auto &kv : F.mBowVec) {
                int32_t w = (int32_t)kv.first;
                wr(out, &w, 1);
                wr(out, &kv.second, 1);
            }
        }

        // ---- background matchers through the shim: SearchForTriangulation of the frame against itself (two key frames
        //      with identical features: every key point's epipolar line under a skew F12 passes through it) and Fuse
        //      of the local map into the frame as a key frame (the associations SearchByProjection made stay in place)
        {
            MatchKeyFrame K1, K2;
            static_cast<Frame &>(K1) = F;
            static_cast<Frame &>(K2) = F;
            K1.mvLevelSigma2 = K2.mvLevelSigma2 = extractor.GetScaleSigmaSquares();
            K1.mvInvLevelSigma2 = K2.mvInvLevelSigma2 = extractor.GetInverseScaleSigmaSquares();
            for (int i = 0; i < F.N; i++) {  // every third key point of each already has a map point (:702-705, :726-729)
                K1.mvpMapPoints[i] = (i % 3 == 0) ? &mps[0] : nullptr;
                K2.mvpMapPoints[i] = (i % 3 == 1) ? &mps[0] : nullptr;
            }
            const float F12[9] = {0.f, -0.f, 0.6f, 0.f, 0.f, -0.8f, -0.6f, 0.8f, 0.f};  // skew of (0.8, 0.6, 0)
            const float Cw[3] = {0.5f, -0.25f, 0.125f}, R2w[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t2w[3] = {0.25f, 0.5f, 2.0f};
            orbgpu_shim::ORBmatcherT<Frame, MapPoint> tm(0.6f, true);
            std::vector<std::pair<size_t, size_t>> pairs;
            const int nt = tm.SearchForTriangulation(
                &K1, &K2, F12, pairs, false, [](const MatchKeyFrame &k, int i) { return &k.mDescriptors[(size_t)i * 32]; },
                [&](MatchKeyFrame *, float *o) { std::memcpy(o, Cw, 12); }, [&](MatchKeyFrame *, float *o) { std::memcpy(o, R2w, 36); },
                [&](MatchKeyFrame *, float *o) { std::memcpy(o, t2w, 12); });
            int32_t nt32 = nt, np32 = (int32_t)pairs.size();
            wr(out, &nt32, 1);
            wr(out, &np32, 1);
            for (const auto &pr : pairs) {
                int32_t ab[2] = {(int32_t)pr.first, (int32_t)pr.second};
                wr(out, ab, 2);
            }

            MatchKeyFrame KF;
            static_cast<Frame &>(KF) = F;  // mvpMapPoints = what SearchByProjection associated above
            KF.mvLevelSigma2 = K1.mvLevelSigma2, KF.mvInvLevelSigma2 = K1.mvInvLevelSigma2;
            g_fuse_kf = &KF;
            for (int j = 0; j < KF.N; j++)
                if (KF.mvpMapPoints[j])
                    KF.mvpMapPoints[j]->idxInKF = j;
            orbgpu_shim::ORBmatcherT<Frame, MapPoint> fm(0.6f, true);
            const int nfused = fm.Fuse(
                &KF, vp, cam[5], [](const MatchKeyFrame &k, int i) { return &k.mDescriptors[(size_t)i * 32]; },
                [&](MatchKeyFrame *, float *T) { std::memcpy(T, Tcw, 64); }, mp_desc, [](MapPoint *p) { return p->world; },
                [](MapPoint *p) { return p->normal; }, [](MapPoint *p) { return p->minDist; },
                [](MapPoint *p) { return p->maxDist; });
            g_fuse_kf = nullptr;
            int32_t nf32 = nfused;
            wr(out, &nf32, 1);
            for (int j = 0; j < KF.N; j++) {
                int32_t id = KF.mvpMapPoints[j] ? KF.mvpMapPoints[j]->id : -1;
                wr(out, &id, 1);
            }
            for (int i = 0; i < m; i++) {
                int32_t st[3] = {mps[i].bad ? 1 : 0, mps[i].nObs, mps[i].replacedBy ? mps[i].replacedBy->id : -1};
                wr(out, st, 3);
            }
        }

        // ---- point cloud thread protocol: three key frames (same images, shifted poses), then a loop closure that
        //      moves every pose, then the shutdown pass
        std::vector<KeyFrame> kfs(3);
        for (int i = 0; i < 3; i++) {
            KeyFrame &kf = kfs[i];
            kf.mImDep = depth, kf.mImRGB = rgb, kf.rows = h, kf.cols = w;
            kf.fx = cam[0], kf.fy = cam[1], kf.cx = cam[2], kf.cy = cam[3];
            std::memcpy(kf.pose, Tcw, 64);
            kf.pose[3] += 0.25f * (float)i;
            kf.mnId = 10 - i;  // ids descending: the loop branch sorts by id
        }
        // shutdown() right after construction, repeatedly: the wake-up must never be lost (join would hang)
        for (int rep = 0; rep < 40; rep++) {
            orbgpu_shim::PointCloudMappingT<KeyFrame, KFAdapter> idle(0.05);
            idle.shutdown();
        }
        bool loop_flag = false;
        orbgpu_shim::PointCloudMappingT<KeyFrame, KFAdapter>::LoopHooks hooks;
        hooks.take_loop_detected = [&] { const bool v = loop_flag; loop_flag = false; return v; };
        hooks.all_keyframes = [&] { return std::vector<KeyFrame *>{&kfs[0], &kfs[1], &kfs[2]}; };
        auto dump = [&](const std::vector<orbgpu_point_xyzrgba> &c) {
            int64_t nc = (int64_t)c.size();
            wr(out, &nc, 1);
            wr(out, c.data(), c.size());
        };
        int64_t nc = 0;
        {
            orbgpu_shim::PointCloudMappingT<KeyFrame, KFAdapter> mapping(0.05, KFAdapter(), 0, hooks);
            for (int i = 0; i < 3; i++) {
                mapping.insertKeyFrame(&kfs[i]);
                mapping.waitProcessed();  // one key frame per pass: the reference's common case
            }
            dump(mapping.globalMap());
            for (int i = 0; i < 3; i++)
                kfs[i].pose[7] -= 0.125f;  // the loop closure corrected the poses
            kfs[1].bad = true;             // a culled key frame is skipped (:228-229)
            loop_flag = true;
            mapping.notifyLoop();
            mapping.waitProcessed();
            dump(mapping.globalMap());
            mapping.shutdown();
            const std::vector<orbgpu_point_xyzrgba> fin = mapping.globalMap();
            nc = (int64_t)fin.size();
            dump(fin);
        }
        // ---- the reference's insert semantics (PointCloudMap.cc:244-262, the shim's default) and the corrected opt-in:
        //      five key frames with DIFFERENT depth images and poses; (1) one alone, (2) two in ONE wake-up (the
        //      reference inserts the last cloud with the first pose), (3) a loop closure detected at a wake-up that
        //      brings a key frame (rebuild; the reference leaves lastKeyframeSize stale), (4) one more insert (the
        //      reference takes the pose of the key frame that arrived with the loop closure)
        struct TwoAtOnce : orbgpu_shim::PointCloudMappingT<KeyFrame, KFAdapter> {
            using PointCloudMappingT::PointCloudMappingT;
            void insertTwo(KeyFrame *a, KeyFrame *b)  // both visible to the viewer's next look at keyframes.size()
            {
                std::unique_lock<std::mutex> lck(keyframeMutex);
                keyframes.push_back(a);
                keyframes.push_back(b);
                keyFrameUpdated.notify_one();
            }
        };
        for (int mode = 0; mode < 2; mode++) {
            std::vector<KeyFrame> q(5);
            for (int i = 0; i < 5; i++) {
                KeyFrame &kf = q[i];
                kf.mImDep = depth, kf.mImRGB = rgb, kf.rows = h, kf.cols = w;
                for (float &d : kf.mImDep)
                    d += 0.125f * (float)i;
                kf.fx = cam[0], kf.fy = cam[1], kf.cx = cam[2], kf.cy = cam[3];
                std::memcpy(kf.pose, Tcw, 64);
                kf.pose[3] += 0.25f * (float)i;
                kf.pose[11] -= 0.0625f * (float)i;
                kf.mnId = 20 + i;
            }
            bool flag2 = false;
            int nmap = 0;
            orbgpu_shim::PointCloudMappingT<KeyFrame, KFAdapter>::LoopHooks hooks2;
            hooks2.take_loop_detected = [&] { const bool v = flag2; flag2 = false; return v; };
            hooks2.all_keyframes = [&] {
                std::vector<KeyFrame *> all;
                for (int i = 0; i < nmap; i++)
                    all.push_back(&q[i]);
                return all;
            };
            TwoAtOnce mapping(0.05, KFAdapter(), 0, hooks2);
            if (mode == 1)
                mapping.setReferenceQuirks(false);
            mapping.setOutlierFilter(0, 1.0);  // the shutdown pass is covered above
            mapping.insertKeyFrame(&q[0]);
            mapping.waitProcessed();
            mapping.insertTwo(&q[1], &q[2]);
            mapping.waitProcessed();
            dump(mapping.globalMap());
            for (int i = 0; i < 5; i++)
                q[i].pose[7] -= 0.125f;  // the loop closure corrected the poses
            nmap = 4;
            flag2 = true;
            mapping.insertKeyFrame(&q[3]);
            mapping.waitProcessed();
            dump(mapping.globalMap());
            mapping.insertKeyFrame(&q[4]);
            mapping.waitProcessed();
            dump(mapping.globalMap());
            mapping.shutdown();
        }
        std::printf("shim ok: %d key points, %d projection matches, %lld map points\n", n, nm, (long long)nc);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "shim_test failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
