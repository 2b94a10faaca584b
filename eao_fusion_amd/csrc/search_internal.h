// search_internal.h -- the cores of the list-based guided searches (csrc/search.hip), shared with the keyframe-handle entry points (csrc/keyframe.hip); not
// part of the C-ABI.  `res` (may be NULL): the searched frame's arrays are resident on the device -- only the queries travel, the view's host arrays serve the
// host replay as before.
#pragma once
#include "match_internal.h"

namespace eao {
namespace search {
eao_status projection_sim3(const eao_frame_view* KF, const match::Resident* res, const float* Scw, float fx, float fy, float cx, float cy,
                           const eao_map_points* pts, int32_t th, int32_t* kp_match, int32_t* nmatches);
eao_status projection_kf(const eao_frame_view* Cur, const match::Resident* res, const float* Tcw, float fx, float fy, float cx, float cy, const eao_map_points* pts,
                         const float* kf_angle, float th, int32_t orb_dist, int32_t check_orientation, int32_t* cur_match, int32_t* nmatches);
eao_status initialization(int32_t n1, const int32_t* octave1, const float* angle1, const uint8_t* desc1, const eao_frame_view* F2, const match::Resident* res,
                          float* prev_matched, int32_t window_size, float nnratio, int32_t check_orientation, int32_t* match12, int32_t* nmatches);
eao_status by_sim3(const eao_frame_view* K1, const match::Resident* res1, const float* T1w, const eao_map_points* pts1, const eao_frame_view* K2,
                   const match::Resident* res2, const float* T2w, const eao_map_points* pts2, float fx, float fy, float cx, float cy, float s12, const float* R12,
                   const float* t12, float th, int32_t* match12, int32_t* nfound);
// argument checks shared with the handle entry points
bool feature_vector_ok(const eao_feature_vector* f, int n);
}  // namespace search
}  // namespace eao
