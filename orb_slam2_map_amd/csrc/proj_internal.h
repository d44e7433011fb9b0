// Internal interface between the projection matchers (matcher_proj.hip) and the MapPoint table entry points
// (map_table.hip): the device-resident SearchLocalPoints with its rows either computed on the device
// (Frame::isInFrustum) or taken from mTrack* members the caller uploaded.
#pragma once
#include "common.h"

namespace orbgpu {

struct ScratchDev {  // device arrays [m]: MapPoint.h:91-96 as filled by Frame::isInFrustum (Frame.cc:317-322)
    const uint8_t *in_view;
    const int32_t *level;
    const float *view_cos, *proj_x, *proj_y, *proj_xr;
};

// a projection-matcher workspace owned by the caller instead of the calling thread (see matcher_proj.hip)
struct ProjWorkspace;
ProjWorkspace *proj_workspace_new();
void proj_workspace_delete(ProjWorkspace *ws);
class ProjWorkspaceScope {
  public:
    explicit ProjWorkspaceScope(ProjWorkspace *ws);
    ~ProjWorkspaceScope();
    ProjWorkspaceScope(const ProjWorkspaceScope &) = delete;
    ProjWorkspaceScope &operator=(const ProjWorkspaceScope &) = delete;

  private:
    ProjWorkspace *prev_;
};

int validate_frame(const orbgpu_frame_view *f);  // consistency of a host frame view (counts, grid CSR)

int search_local_points_device_impl(const orbgpu_device_frame_view *f, const orbgpu_device_mappoint_table *mp,
                                    const ScratchDev *scratch, const float *Tcw, float fx, float fy, float cx, float cy,
                                    float mbf, float log_scale_factor, float cos_limit, float th, float nnratio,
                                    int32_t *d_kp_to_mp, int32_t *d_counts, const orbgpu_track_scratch *d_track,
                                    int32_t device_id, void *hip_stream);

} // namespace orbgpu
