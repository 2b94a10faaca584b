// oracle/frame_cpu.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into or called by the product).
//
// Sequential restatement, over plain arrays, of the Frame glue either side of the matcher:
//   Frame::isInFrustum(MapPoint*, viewingCosLimit)      src/Frame.cc:638-695   (driven by Tracking::SearchLocalPoints,
//                                                        src/Tracking.cc:2587-2641)
//   Frame::AssignFeaturesToGrid / PosInGrid             src/Frame.cc:597-614, 751-761
//   Frame::ComputeStereoFromRGBD                        src/Frame.cc:1016-1037
//   MapPoint::PredictScale(dist, logScaleFactor)        src/MapPoint.cc:385-394
// PARITY UNPINNED (no upstream tests / fixtures for these functions; Frame.cc needs OpenCV, PCL and the whole SLAM
// object model to compile).  Documented choices for the cv::Mat expressions of float matrices, the same as in
// search_cpu.cpp: a product A*x + b accumulates in double and rounds once to float (cv::gemm); cv::norm / Mat::dot of
// floats accumulate in double; PredictScale evaluates log / division / ceil on float operands in float.
#include <cmath>
#include <cstdint>
#include <vector>

extern "C" {

// One call = the isInFrustum loop of SearchLocalPoints over n map points.  Rcw (9, row-major), tcw (3), Ow (3) are the
// frame's float matrices.  in_view[i] = return value; the other outputs are written only where it is 1 (upstream
// leaves the MapPoint members untouched otherwise).
int orc_is_in_frustum(int32_t n, const float* Xw, const float* normal, const float* min_dist, const float* max_dist,
                      const float* max_dist_num, const float* Rcw, const float* tcw, const float* Ow, float fx, float fy,
                      float cx, float cy, float mbf, float min_x, float max_x, float min_y, float max_y,
                      float log_scale_factor, float viewing_cos_limit, uint8_t* in_view, float* proj_x, float* proj_y,
                      float* proj_xr, float* view_cos, int32_t* pred_level) {
    for (int i = 0; i < n; i++) {
        in_view[i] = 0;                                                        // :640
        const float* P = Xw + 3 * i;
        float Pc[3];
        for (int r = 0; r < 3; r++) {                                          // Pc = mRcw*P + mtcw, :646
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (double)Rcw[3 * r + k] * (double)P[k];
            Pc[r] = (float)(acc + (double)tcw[r]);
        }
        const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
        if (PcZ < 0.0f) continue;                                              // :652
        const float invz = 1.0f / PcZ;                                         // :656
        const float u = fx * PcX * invz + cx;
        const float v = fy * PcY * invz + cy;
        if (u < min_x || u > max_x) continue;                                  // :660-663
        if (v < min_y || v > max_y) continue;
        const float maxDistance = max_dist[i], minDistance = min_dist[i];      // :666-667
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};         // :668
        const float dist = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);   // cv::norm, :669
        if (dist < minDistance || dist > maxDistance) continue;                // :671
        const float* Pn = normal + 3 * i;
        const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];   // Mat::dot returns double
        const float viewCos = (float)(dot / (double)dist);                     // :677
        if (viewCos < viewing_cos_limit) continue;                             // :679
        const float ratio = max_dist_num[i] / dist;                            // PredictScale, src/MapPoint.cc:390-393
        const int level = (int)std::ceil(std::log(ratio) / log_scale_factor);
        in_view[i] = 1;                                                        // :686-691
        proj_x[i] = u;
        proj_xr[i] = u - mbf * invz;
        proj_y[i] = v;
        pred_level[i] = level;
        view_cos[i] = viewCos;
    }
    return 0;
}

// mGrid as a CSR: cell c = ix * rows + iy (mGrid[ix][iy]); cell_start has cols*rows + 1 entries; items in push_back order.
int orc_assign_features_to_grid(int32_t n, const float* kp_x, const float* kp_y, float min_x, float min_y, float inv_w,
                                float inv_h, int32_t cols, int32_t rows, int32_t* cell_start, int32_t* items) {
    std::vector<std::vector<int>> grid((size_t)cols * rows);
    for (int i = 0; i < n; i++) {                                              // :604-611
        const int px = (int)std::round((kp_x[i] - min_x) * inv_w);              // PosInGrid, :753-754
        const int py = (int)std::round((kp_y[i] - min_y) * inv_h);
        if (px < 0 || px >= cols || py < 0 || py >= rows) continue;            // :757
        grid[(size_t)px * rows + py].push_back(i);
    }
    int o = 0;
    for (size_t c = 0; c < grid.size(); c++) {
        cell_start[c] = o;
        for (int i : grid[c]) items[o++] = i;
    }
    cell_start[grid.size()] = o;
    return 0;
}

// depth: CV_32F image, `pitch` floats per row.  kp_x / kp_y: mvKeys (distorted), kpu_x: mvKeysUn.  at<float>(v, u)
// converts its float arguments to int by truncation.
int orc_stereo_from_rgbd(int32_t n, const float* kp_x, const float* kp_y, const float* kpu_x, const float* depth, int32_t pitch,
                         float mbf, float* u_right, float* out_depth) {
    for (int i = 0; i < n; i++) {
        u_right[i] = -1; out_depth[i] = -1;                                    // :1018-1019
        const float d = depth[(size_t)(int)kp_y[i] * pitch + (int)kp_x[i]];     // :1029
        if (d > 0) {                                                           // :1031-1035
            out_depth[i] = d;
            u_right[i] = kpu_x[i] - mbf / d;
        }
    }
    return 0;
}

}  // extern "C"
