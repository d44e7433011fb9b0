// The declarations of the reference's own headers that INTEGRATION.md's replacement bodies are written against
// (include/ORBextractor.h, Frame.h, MapPoint.h, KeyFrame.h, ORBmatcher.h, LoopClosing.h, Map.h), reduced to the members
// those bodies and the shim templates touch, with the reference's names and types.  Test scaffolding: declarations only.
#pragma once
#include <cassert>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "cv_standin.hpp"
#include "orbgpu_shim.hpp"

namespace ORB_SLAM2 {
using std::set;
using std::vector;

class KeyFrame;
class Map;

class ORBextractor {  // include/ORBextractor.h:46-110, plus the two lines INTEGRATION.md section 1 adds
  public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);
    std::vector<cv::Mat> mvImagePyramid;

  protected:
    int nfeatures;
    double scaleFactor;
    int nlevels, iniThFAST, minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    struct Impl;
    std::unique_ptr<Impl> impl;
};

class MapPoint {  // include/MapPoint.h
  public:
    long unsigned int mnId = 0;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    long unsigned int mnLastFrameSeen = 0;
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    bool isBad() { return mbBad; }
    int Observations() { return nObs; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    float GetMaxDistance() { return mfMaxDistance; }  // the accessor INTEGRATION.md section 2 asks for (MapPoint.h:141 is protected)
    float GetMinDistance() { return mfMinDistance; }
    void IncreaseVisible(int n = 1) { mnVisible += n; }

  protected:
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    bool mbBad = false;
    int nObs = 0, mnVisible = 0;
    float mfMinDistance = 0, mfMaxDistance = 0;
};

class Frame {  // include/Frame.h:100-190
  public:
    long unsigned int mnId = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors, mTcw;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<std::size_t> mGrid[orbgpu_shim::FRAME_GRID_COLS][orbgpu_shim::FRAME_GRID_ROWS];
    std::vector<float> mvScaleFactors;
    orbgpu_shim::FeatureVector mFeatVec;  // DBoW2::FeatureVector derives from this map type
    orbgpu_shim::BowVector mBowVec;
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    float mbf = 0, mb = 0, mfLogScaleFactor = 0;
};

class KeyFrame {  // include/KeyFrame.h
  public:
    long unsigned int mnId = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat mDescriptors, mImRGB, mImDep;
    orbgpu_shim::FeatureVector mFeatVec;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    cv::Mat GetPose() { return Tcw.clone(); }
    bool isBad() { return mbBad; }
    static bool lId(KeyFrame *a, KeyFrame *b) { return a->mnId < b->mnId; }

  protected:
    std::vector<MapPoint *> mvpMapPoints;
    cv::Mat Tcw;
    bool mbBad = false;
};

class Map {
  public:
    std::vector<KeyFrame *> GetAllKeyFrames() { return std::vector<KeyFrame *>(); }
};
class LoopClosing {  // include/LoopClosing.h: the two members PointCloudMapping::viewer reads (PointCloudMap.cc:217-221)
  public:
    bool loop_detected = false;
    Map *getMap() { return &map_; }

  private:
    Map map_;
};
struct System { enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2 }; };

class ORBmatcher {  // include/ORBmatcher.h:41-106
  public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                           const int ORBdist);
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);

  protected:
    float mfNNratio;
    bool mbCheckOrientation;
};
} // namespace ORB_SLAM2
