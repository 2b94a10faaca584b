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

// Frame::UndistortKeyPoints, src/Frame.cc:773-806.  cv::undistortPoints(src, dst, mK, mDistCoef, cv::Mat(), mK) is third-party code (OpenCV; the reference
// asks for "OpenCV 3.0" without pinning a release, ros_test/CMakeLists.txt:24): restated from the published algorithm of OpenCV 3.x's cvUndistortPoints
// (modules/imgproc/src/undistort.cpp) -- K and the coefficients promoted to double, x = (u - cx) / fx as a product with 1 / fx, FIVE fixed-point iterations
//     r2 = x^2 + y^2;  icdist = (1 + ((k6 r2 + k5) r2 + k4) r2) / (1 + ((k3 r2 + k2) r2 + k1) r2)
//     dX = 2 p1 x y + p2 (r2 + 2 x^2) + s1 r2 + s2 r2^2;   dY = p1 (r2 + 2 y^2) + 2 p2 x y + s3 r2 + s4 r2^2
//     x = (x0 - dX) icdist;  y = (y0 - dY) icdist
// with k4 = k5 = k6 = s1..s4 = 0 for the reference's four or five coefficients, then xx = fx x + 0 y + cx (R = I, P = K: RR = K), divided by w = 1, to float.
int orc_undistort_keypoints(int32_t n, const float* kp_x, const float* kp_y, float fxf, float fyf, float cxf, float cyf, const float* dist, int32_t n_coef,
                            float* out_x, float* out_y) {
    if (n_coef < 1 || dist[0] == 0.0f) {                                       // :775-779
        for (int i = 0; i < n; i++) { out_x[i] = kp_x[i]; out_y[i] = kp_y[i]; }
        return 0;
    }
    double k[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_coef && i < 5; i++) k[i] = dist[i];                  // k1 k2 p1 p2 k3
    const double fx = fxf, fy = fyf, cx = cxf, cy = cyf, ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        double x = kp_x[i], y = kp_y[i];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        out_x[i] = (float)(xx * ww); out_y[i] = (float)(yy * ww);
    }
    return 0;
}

// Frame::ComputeImageBounds, src/Frame.cc:808-842.  bounds = mnMinX, mnMaxX, mnMinY, mnMaxY
int orc_compute_image_bounds(int32_t cols, int32_t rows, float fx, float fy, float cx, float cy, const float* dist, int32_t n_coef, float* bounds) {
    if (n_coef < 1 || dist[0] == 0.0f) { bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows; return 0; }
    const float x[4] = {0.0f, (float)cols, 0.0f, (float)cols}, y[4] = {0.0f, 0.0f, (float)rows, (float)rows};
    float ux[4], uy[4];
    orc_undistort_keypoints(4, x, y, fx, fy, cx, cy, dist, n_coef, ux, uy);
    bounds[0] = std::fmin(ux[0], ux[2]); bounds[1] = std::fmax(ux[1], ux[3]);
    bounds[2] = std::fmin(uy[0], uy[1]); bounds[3] = std::fmax(uy[2], uy[3]);
    bounds[0] = std::fmax(bounds[0], 0.0f); bounds[1] = std::fmin(bounds[1], (float)cols);
    bounds[2] = std::fmax(bounds[2], 0.0f); bounds[3] = std::fmin(bounds[3], (float)rows);
    return 0;
}

}  // extern "C"
