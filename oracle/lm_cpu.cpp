// oracle/lm_cpu.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into or called by the product).
//
// fp64 restatement of the arithmetic the reference delegates to its vendored g2o for
//   Optimizer::PoseOptimization        reference src/Optimizer.cc:325-673
//   Optimizer::LocalBundleAdjustment   reference src/Optimizer.cc:675-1138
// following (all under reference Thirdparty/g2o/g2o/):
//   core/optimization_algorithm_levenberg.cpp:61-189   LM driver: lambda init, rho, nu, "3 bad iterations" stop
//   core/sparse_optimizer.cpp:61-114,166-267,354-435   active sets, robust chi2, optimize(), update()
//   core/block_solver.hpp:354-486,502-604              buildSystem, setLambda/restoreDiagonal, Schur solve
//   core/base_binary_edge.hpp:55-120, core/base_unary_edge.hpp:43-72   quadratic forms (rho' weighting)
//   core/base_edge.h:58-102, core/robust_kernel_impl.cpp:78-91         chi2, Huber
//   types/types_six_dof_expmap.{h,cpp}, types/types_sba.h:40-57, types/se3quat.h, types/se3_ops.hpp
//   solvers/linear_solver_dense.h:55-112 (pose-only 6x6), solvers/linear_solver_eigen.h:62-112 (Schur system)
//   src/Converter.cc:28-104 (float32 <-> double boundary)
//
// PARITY UNPINNED: the reference holds no test, fixture or known answer for this path, and g2o does not
// compile here (Eigen is an un-vendored dependency: Thirdparty/g2o/CMakeLists.txt:74 requires >= 3.1.0,
// README.md:47 names 3.2.1).  Eigen's pieces are restated from their published algorithms: 3x3 cofactor
// inverse, Quaternion<->matrix, pivoted dense LDLT (6x6) and -- deviation -- an UNPERMUTED dense LDLT in
// place of SimplicialLDLT+AMD for the Schur system (same factorisation up to rounding order).  Pinned by the
// known-answer tests in tests/test_oracle_lm.py: Jacobians vs central differences, Schur solve vs full
// dense solve, zero-noise convergence to ground truth, Huber values.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "ref_constants.inc"   // GENERATED from the reference text (tools/gen_ref_constants.py): namespace refc

namespace {

// ------------------------------------------------------------------ small fixed-size helpers
struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };

Quat quat_from_matrix(const double m[9]) {  // row-major 3x3
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t;
        q.y = (m[2] - m[6]) * t;
        q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
void quat_normalize_pos(Quat& q) {  // SE3Quat::normalizeRotation
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
Quat quat_mul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
void quat_rotate(const Quat& q, const double v[3], double out[3]) {
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
void quat_to_matrix(const Quat& q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
SE3 se3_from_Tcw_f32(const float* T) {  // Converter::toSE3Quat
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    SE3 s;
    s.r = quat_from_matrix(R);
    s.t[0] = T[3]; s.t[1] = T[7]; s.t[2] = T[11];
    quat_normalize_pos(s.r);
    return s;
}
void se3_to_Tcw_f32(const SE3& s, float* T) {  // Converter::toCvMat(SE3Quat)
    double R[9];
    quat_to_matrix(s.r, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)s.t[i];
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}
void se3_map(const SE3& s, const double p[3], double out[3]) {
    quat_rotate(s.r, p, out);
    out[0] += s.t[0]; out[1] += s.t[1]; out[2] += s.t[2];
}
void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
SE3 se3_exp(const double u[6]) {  // SE3Quat::exp: u = (omega, upsilon)
    const double w[3] = {u[0], u[1], u[2]}, v[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double Om2[9], R[9], V[9];
    mat3_mul(Om, Om, Om2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = I[i] + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; i++) { R[i] = I[i] + a * Om[i] + b * Om2[i]; V[i] = I[i] + b * Om[i] + c * Om2[i]; }
    }
    SE3 s;
    s.r = quat_from_matrix(R);
    for (int i = 0; i < 3; i++) s.t[i] = V[i * 3] * v[0] + V[i * 3 + 1] * v[1] + V[i * 3 + 2] * v[2];
    quat_normalize_pos(s.r);
    return s;
}
SE3 se3_mul(const SE3& a, const SE3& b) {  // SE3Quat::operator*
    SE3 r;
    double rt[3];
    quat_rotate(a.r, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
    r.r = quat_mul(a.r, b.r);
    quat_normalize_pos(r.r);
    return r;
}
bool inv3_cofactor(const double A[9], double out[9]) {  // Eigen fixed-size 3x3 inverse
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const double id = 1.0 / det;
    out[0] = c00 * id; out[1] = (A[2] * A[7] - A[1] * A[8]) * id; out[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    out[3] = c01 * id; out[4] = (A[0] * A[8] - A[2] * A[6]) * id; out[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    out[6] = c02 * id; out[7] = (A[1] * A[6] - A[0] * A[7]) * id; out[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}

// Huber (robust_kernel_impl.cpp:78-91): rho0 = rho(e), rho1 = rho'(e)
inline void huber(double e, double delta, double dsqr, double& rho0, double& rho1) {
    if (e <= dsqr) { rho0 = e; rho1 = 1.; }
    else { double s = std::sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// pivoted dense LDLT (largest |diagonal| pivot), n <= 6; returns "positive" like Eigen::LDLT::isPositive
bool ldlt_pivot_solve(int n, const double* Ain, const double* b, double* x) {
    double A[36], d[6];
    int perm[6];
    std::memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) perm[i] = i;
    bool positive = true;
    for (int k = 0; k < n; k++) {
        int p = k; double big = std::fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (std::fabs(A[i * n + i]) > big) { big = std::fabs(A[i * n + i]); p = i; }
        if (p != k) {  // symmetric row/column swap
            for (int j = 0; j < n; j++) std::swap(A[k * n + j], A[p * n + j]);
            for (int i = 0; i < n; i++) std::swap(A[i * n + k], A[i * n + p]);
            std::swap(perm[k], perm[p]);
        }
        double dk = A[k * n + k];
        d[k] = dk;
        if (!(dk > 0)) positive = false;
        if (dk == 0) continue;
        for (int i = k + 1; i < n; i++) {
            double l = A[i * n + k] / dk;
            for (int j = k + 1; j <= i; j++) { A[i * n + j] -= l * A[k * n + j]; A[j * n + i] = A[i * n + j]; }
            A[i * n + k] = l;
        }
    }
    if (!positive) return false;
    double y[6];
    for (int i = 0; i < n; i++) y[i] = b[perm[i]];
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= A[i * n + j] * y[j];
    double dmax = 0;
    for (int i = 0; i < n; i++) dmax = std::max(dmax, std::fabs(d[i]));
    const double tol = std::max(dmax * DBL_EPSILON, 1.0 / DBL_MAX);
    for (int i = 0; i < n; i++) y[i] = (std::fabs(d[i]) > tol) ? y[i] / d[i] : 0.0;
    for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= A[j * n + i] * y[j];
    for (int i = 0; i < n; i++) x[perm[i]] = y[i];
    return true;
}

// unpermuted dense LDLT on the upper triangle (stand-in for SimplicialLDLT<Upper>; fails on a zero pivot)
bool ldlt_dense_solve(int n, std::vector<double>& A, const double* b, double* x) {
    // A row-major symmetric (upper valid).  In-place: L below the diagonal (unit), D on the diagonal.
    for (int j = 0; j < n; j++) {
        double dj = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) dj -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
        if (dj == 0.0 || !std::isfinite(dj)) return false;
        A[(size_t)j * n + j] = dj;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)j * n + i];  // upper element (j,i) == (i,j)
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
            A[(size_t)i * n + j] = s / dj;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * x[k]; x[i] = s; }
    for (int i = 0; i < n; i++) x[i] /= A[(size_t)i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k]; x[i] = s; }
    return true;
}

// ------------------------------------------------------------------ LM driver shared by both problems
struct LMTrace { std::vector<double> lambda, chi2; std::vector<int> trials; };

struct Problem {
    virtual ~Problem() {}
    virtual bool hasActive() const = 0;          // _ivMap.size() != 0
    virtual void computeActiveErrors() = 0;
    virtual double activeRobustChi2() const = 0;
    virtual void buildSystem() = 0;
    virtual double maxDiagonal() const = 0;
    virtual bool solveWithLambda(double lambda) = 0;   // setLambda + solve + restoreDiagonal; fills x
    virtual double computeScale(double lambda) const = 0;  // sum x_j (lambda x_j + b_j)
    virtual void push() = 0;
    virtual void pop() = 0;
    virtual void discardTop() = 0;
    virtual void applyUpdate() = 0;
};

struct LM {
    double lambda = -1, ni = 2;
    int nBad = 0;
    enum Result { OK, Terminate };
    Result solve(Problem& P, int iteration, const volatile bool* stop, LMTrace* tr) {
        P.computeActiveErrors();
        double currentChi = P.activeRobustChi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        P.buildSystem();
        if (iteration == 0) { lambda = refc::LM_TAU * P.maxDiagonal(); ni = refc::LM_NI; nBad = 0; }
        double rho = 0;
        int qmax = 0;
        do {
            P.push();
            bool ok2 = P.solveWithLambda(lambda);
            P.applyUpdate();
            P.computeActiveErrors();
            tempChi = P.activeRobustChi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = P.computeScale(lambda);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = refc::LM_NI;
                currentChi = tempChi;
                P.discardTop();
            } else {
                lambda *= ni;
                ni *= 2;
                P.pop();
            }
            qmax++;
        } while (rho < 0 && qmax < refc::LM_MAX_TRIALS && !(stop && *stop));
        if (tr) { tr->lambda.push_back(lambda); tr->chi2.push_back(currentChi); tr->trials.push_back(qmax); }
        if (qmax == refc::LM_MAX_TRIALS || rho == 0) return Terminate;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) return Terminate;
        return OK;
    }
    // SparseOptimizer::optimize
    int optimize(Problem& P, int iterations, const volatile bool* stop, LMTrace* tr) {
        if (!P.hasActive()) return -1;
        int done = 0;
        bool ok = true;
        for (int i = 0; i < iterations && !(stop && *stop) && ok; i++) {
            ok = (solve(P, i, stop, tr) == OK);
            ++done;
        }
        return done;
    }
};

// ------------------------------------------------------------------ PoseOptimization (unary edges)
// ------------------------------------------------------------------ planes (src/g2oAddition/Plane3D.h, EdgePlane.h)
// Plane3D keeps (n, -d) normalised with a non-negative fourth coefficient; ominus returns (azimuth, elevation, distance)
// of the measured plane in the frame that rotates this plane's normal onto +x.
void plane_normalize(double c[4]) {                       // Plane3D::normalize, Plane3D.h:125-130
    const double n = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double s = 1. / n;
    for (int k = 0; k < 4; k++) c[k] = c[k] * s;
    if (c[3] < 0.0) for (int k = 0; k < 4; k++) c[k] = -c[k];
}
void plane_from_f32(const float* v, double c[4]) {         // Converter::toPlane3D, src/Converter.cc:215-225
    for (int k = 0; k < 4; k++) c[k] = v[k];
    if (v[3] < 0.0) for (int k = 0; k < 4; k++) c[k] = -c[k];
    plane_normalize(c);
}
void plane_rotation(const double v[3], double R[9]) {      // Plane3D::rotation, Plane3D.h:65-71: AngleAxis(az, Z) * AngleAxis(-el, Y)
    const double az = std::atan2(v[1], v[0]);
    const double el = std::atan2(v[2], std::sqrt(v[0] * v[0] + v[1] * v[1]));
    const double ha = 0.5 * az, hb = 0.5 * (-el);
    const Quat a{0, 0, std::sin(ha) * 1.0, std::cos(ha)}, b{0, std::sin(hb) * 1.0, 0, std::cos(hb)};
    quat_to_matrix(quat_mul(a, b), R);
}
void plane_error(const SE3& T, const double world[4], const double meas[4], double err[3]) {   // EdgePlane::computeError
    double R[9];
    quat_to_matrix(T.r, R);
    double v2[4];
    for (int r = 0; r < 3; r++) v2[r] = R[r * 3] * world[0] + R[r * 3 + 1] * world[1] + R[r * 3 + 2] * world[2];
    v2[3] = world[3] - (T.t[0] * v2[0] + T.t[1] * v2[1] + T.t[2] * v2[2]);
    if (v2[3] < 0.0) for (int k = 0; k < 4; k++) v2[k] = -v2[k];
    plane_normalize(v2);                                   // Plane3D(v2)
    double Rn[9];
    plane_rotation(v2, Rn);
    double n[3];
    for (int r = 0; r < 3; r++) n[r] = Rn[r] * meas[0] + Rn[3 + r] * meas[1] + Rn[6 + r] * meas[2];   // rotation^T * normal
    err[0] = std::atan2(n[1], n[0]);
    err[1] = std::atan2(n[2], std::sqrt(n[0] * n[0] + n[1] * n[1]));
    err[2] = (-v2[3]) - (-meas[3]);
}
void plane_oplus(const double c[4], const double v[3], double out[4]) {   // Plane3D::oplus, Plane3D.h:73-89 (VertexPlane::oplusImpl)
    const double az = v[0], el = v[1];
    const double sn = std::sin(el), cs = std::cos(el);
    const double n[3] = {cs * std::cos(az), cs * std::sin(az), sn};
    double R[9];
    plane_rotation(c, R);
    const double d = (-c[3]) + v[2];
    for (int r = 0; r < 3; r++) out[r] = R[r * 3] * n[0] + R[r * 3 + 1] * n[1] + R[r * 3 + 2] * n[2];
    out[3] = -d;
    plane_normalize(out);
}
struct PlaneEdge {
    double world[4], meas[4], infoA, infoD;
    int level = 0;
    bool robust = true;
    double err[3] = {0, 0, 0};
};

struct PoseEdge {
    double Xw[3], obs[3], info;
    bool stereo;
    int level = 0;
    bool robust = true;
    double err[3] = {0, 0, 0};
};

struct PoseProblem : Problem {
    SE3 est, backup;
    std::vector<PoseEdge> edges;
    double fx, fy, cx, cy, bf;
    double deltaMono, deltaStereo;  // float sqrt() results promoted (Optimizer.cc:361-362)
    std::vector<PlaneEdge> planes;  // EdgePlane against fixed plane vertices (src/Optimizer.cc:456-535), after the point edges
    double deltaPlane = 0;
    double H[36], b[6], x[6];

    static double chi2(const PlaneEdge& e) {
        return e.err[0] * (e.infoA * e.err[0]) + e.err[1] * (e.infoA * e.err[1]) + e.err[2] * (e.infoD * e.err[2]);
    }
    void edgeError(PoseEdge& e) const {
        double p[3];
        se3_map(est, e.Xw, p);
        if (!e.stereo) {
            e.err[0] = e.obs[0] - (p[0] / p[2] * fx + cx);
            e.err[1] = e.obs[1] - (p[1] / p[2] * fy + cy);
            e.err[2] = 0;
        } else {
            const float invz = (float)(1.0 / p[2]);  // "const float invz = 1.0f/trans_xyz[2]" (:336)
            double r0 = p[0] * invz * fx + cx;
            double r1 = p[1] * invz * fy + cy;
            double r2 = r0 - bf * invz;               // bf is a double member here (OnlyPose edge)
            e.err[0] = e.obs[0] - r0; e.err[1] = e.obs[1] - r1; e.err[2] = e.obs[2] - r2;
        }
    }
    static double chi2(const PoseEdge& e) {
        // _error.dot(information * _error), information = invSigma2 * I
        double s = e.err[0] * (e.info * e.err[0]) + e.err[1] * (e.info * e.err[1]);
        if (e.stereo) s += e.err[2] * (e.info * e.err[2]);
        return s;
    }
    bool hasActive() const override {
        for (auto& e : edges) if (e.level == 0) return true;
        for (auto& e : planes) if (e.level == 0) return true;
        return false;
    }
    void computeActiveErrors() override {
        for (auto& e : edges) if (e.level == 0) edgeError(e);
        for (auto& e : planes) if (e.level == 0) plane_error(est, e.world, e.meas, e.err);
    }
    double activeRobustChi2() const override {
        double chi = 0;
        for (auto& e : edges) {
            if (e.level != 0) continue;
            double c = chi2(e);
            if (e.robust) { double d = e.stereo ? deltaStereo : deltaMono, r0, r1; huber(c, d, d * d, r0, r1); chi += r0; }
            else chi += c;
        }
        for (auto& e : planes) {
            if (e.level != 0) continue;
            double c = chi2(e);
            if (e.robust) { double r0, r1; huber(c, deltaPlane, deltaPlane * deltaPlane, r0, r1); chi += r0; }
            else chi += c;
        }
        return chi;
    }
    void buildSystem() override {
        std::memset(H, 0, sizeof(H)); std::memset(b, 0, sizeof(b));
        for (auto& e : edges) {
            if (e.level != 0) continue;
            double p[3];
            se3_map(est, e.Xw, p);
            const double X = p[0], Y = p[1], invz = 1.0 / p[2], invz2 = invz * invz;
            double J[3][6];
            J[0][0] = X * Y * invz2 * fx; J[0][1] = -(1 + (X * X * invz2)) * fx; J[0][2] = Y * invz * fx;
            J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = X * invz2 * fx;
            J[1][0] = (1 + Y * Y * invz2) * fy; J[1][1] = -X * Y * invz2 * fy; J[1][2] = -X * invz * fy;
            J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = Y * invz2 * fy;
            int D = 2;
            if (e.stereo) {
                D = 3;
                J[2][0] = J[0][0] - bf * Y * invz2; J[2][1] = J[0][1] + bf * X * invz2; J[2][2] = J[0][2];
                J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - bf * invz2;
            }
            double w = 1.0;
            if (e.robust) { double d = e.stereo ? deltaStereo : deltaMono, r0; huber(chi2(e), d, d * d, r0, w); }
            // b -= rho' * J^T (Omega e) ; H += J^T (rho' Omega) J
            const double wi = w * e.info;
            for (int i = 0; i < 6; i++) {
                double s = 0;
                for (int k = 0; k < D; k++) s += J[k][i] * (e.info * e.err[k]);
                b[i] -= w * s;
                for (int j = 0; j < 6; j++) {
                    double h = 0;
                    for (int k = 0; k < D; k++) h += J[k][i] * wi * J[k][j];
                    H[i * 6 + j] += h;
                }
            }
        }
        // plane edges: g2o's numeric Jacobian with respect to the pose (central differences, delta = 1e-9,
        // core/base_binary_edge.hpp:131-205); the plane vertex is fixed
        for (auto& e : planes) {
            if (e.level != 0) continue;
            const double delta = 1e-9, scalar = 1.0 / (2 * delta);
            double J[3][6];
            for (int d = 0; d < 6; d++) {
                double add[6] = {0, 0, 0, 0, 0, 0}, ep[3], em[3];
                add[d] = delta;
                plane_error(se3_mul(se3_exp(add), est), e.world, e.meas, ep);
                add[d] = -delta;
                plane_error(se3_mul(se3_exp(add), est), e.world, e.meas, em);
                for (int k = 0; k < 3; k++) J[k][d] = scalar * (ep[k] - em[k]);
            }
            double w = 1.0;
            if (e.robust) { double r0; huber(chi2(e), deltaPlane, deltaPlane * deltaPlane, r0, w); }
            const double info[3] = {e.infoA, e.infoA, e.infoD};
            for (int i = 0; i < 6; i++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += J[k][i] * (info[k] * e.err[k]);
                b[i] -= w * s;
                for (int j = 0; j < 6; j++) {
                    double h = 0;
                    for (int k = 0; k < 3; k++) h += J[k][i] * (w * info[k]) * J[k][j];
                    H[i * 6 + j] += h;
                }
            }
        }
    }
    double maxDiagonal() const override { double m = 0; for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 6 + j]), m); return m; }
    bool solveWithLambda(double lambda) override {
        double A[36];
        std::memcpy(A, H, sizeof(A));
        for (int j = 0; j < 6; j++) A[j * 6 + j] += lambda;
        // only the upper triangle is copied into the dense matrix, then mirrored (linear_solver_dense.h:86-96)
        for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) A[i * 6 + j] = A[j * 6 + i];
        bool ok = ldlt_pivot_solve(6, A, b, x);
        return ok;
    }
    double computeScale(double lambda) const override { double s = 0; for (int j = 0; j < 6; j++) s += x[j] * (lambda * x[j] + b[j]); return s; }
    void push() override { backup = est; }
    void pop() override { est = backup; }
    void discardTop() override {}
    void applyUpdate() override { est = se3_mul(se3_exp(x), est); }
};

// ------------------------------------------------------------------ LocalBundleAdjustment (binary edges + Schur)
struct BAEdge {
    int cam, pt;
    double obs[3], info;
    bool stereo;
    int level = 0;
    bool robust = true;
    double err[3] = {0, 0, 0};
    // EdgePlane (src/g2oAddition/EdgePlane.h) between a VertexPlane and a keyframe: `pt` is then a landmark id beyond the
    // points (point count + plane index), `meas` the measured plane, `info3` the diagonal of its information matrix
    bool plane = false;
    double meas[4] = {0, 0, 0, 0}, info3[3] = {0, 0, 0};
};

struct BAProblem : Problem {
    std::vector<SE3> cams, camsBackup;
    std::vector<uint8_t> camFixed;
    std::vector<double> pts, ptsBackup;  // 3 per point
    std::vector<double> planes, planesBackup;   // 4 per MapPlane vertex (marginalised 3-dof landmarks after the points)
    double deltaPlane = 0;
    std::vector<BAEdge> edges;
    double fx, fy, cx, cy, bf;
    float bf_f;
    double deltaMono, deltaStereo;
    // active structure (rebuilt by initialize(level))
    std::vector<int> camIdx, ptIdx;     // hessian block index or -1 (fixed / inactive)
    std::vector<int> activeCams, activePts, activeEdges;
    int nP = 0, nL = 0;
    std::vector<double> Hpp, Hll, bvec, x;   // Hpp: nP 6x6 diagonal blocks; Hll: nL 3x3; bvec/x: 6nP + 3nL
    std::vector<double> Hpl;                   // per active edge: 6x3 block (pose row, landmark col) or unused
    std::vector<std::vector<int>> ptEdges;     // per landmark block: active edge ids with a free camera, by ascending camera block

    // SparseOptimizer::initializeOptimization(level) + buildIndexMapping
    void initialize() {
        const int nc = (int)cams.size(), np = (int)pts.size() / 3 + (int)planes.size() / 4;
        std::vector<int> camCnt(nc, 0), ptCnt(np, 0);
        activeEdges.clear();
        for (int k = 0; k < (int)edges.size(); k++) {
            if (edges[k].level != 0) continue;
            // !allVerticesFixed(): the point vertex is never fixed
            activeEdges.push_back(k);
            camCnt[edges[k].cam]++; ptCnt[edges[k].pt]++;
        }
        camIdx.assign(nc, -1); ptIdx.assign(np, -1);
        activeCams.clear(); activePts.clear();
        nP = 0;
        for (int c = 0; c < nc; c++) if (camCnt[c] && !camFixed[c]) { camIdx[c] = nP++; activeCams.push_back(c); }
        nL = 0;
        for (int p = 0; p < np; p++) if (ptCnt[p]) { ptIdx[p] = nL++; activePts.push_back(p); }
        Hpp.assign((size_t)nP * 36, 0); Hll.assign((size_t)nL * 9, 0);
        bvec.assign((size_t)nP * 6 + (size_t)nL * 3, 0); x.assign(bvec.size(), 0);
        Hpl.assign(edges.size() * 18, 0);
        ptEdges.assign(nL, {});
        for (int k : activeEdges) if (camIdx[edges[k].cam] >= 0) ptEdges[ptIdx[edges[k].pt]].push_back(k);
        for (auto& v : ptEdges)
            std::stable_sort(v.begin(), v.end(), [&](int a, int b2) { return camIdx[edges[a].cam] < camIdx[edges[b2].cam]; });
    }
    bool hasActive() const override { return nP + nL > 0; }

    void project(const BAEdge& e, const double p[3], double r[3]) const {
        if (!e.stereo) {
            r[0] = p[0] / p[2] * fx + cx; r[1] = p[1] / p[2] * fy + cy; r[2] = 0;
        } else {
            const float invz = (float)(1.0 / p[2]);   // types_six_dof_expmap.cpp:150-156
            r[0] = p[0] * invz * fx + cx;
            r[1] = p[1] * invz * fy + cy;
            r[2] = r[0] - (double)(bf_f * invz);        // "const float &bf": float * float product
        }
    }
    const double* planeOf(const BAEdge& e) const { return &planes[4 * (size_t)(e.pt - (int)pts.size() / 3)]; }
    void edgeError(BAEdge& e) const {
        if (e.plane) { plane_error(cams[e.cam], planeOf(e), e.meas, e.err); return; }
        double p[3], r[3];
        se3_map(cams[e.cam], &pts[3 * e.pt], p);
        project(e, p, r);
        e.err[0] = e.obs[0] - r[0]; e.err[1] = e.obs[1] - r[1]; e.err[2] = e.stereo ? e.obs[2] - r[2] : 0;
    }
    bool depthPositive(const BAEdge& e) const {
        double p[3];
        se3_map(cams[e.cam], &pts[3 * e.pt], p);
        return p[2] > 0.0;
    }
    static double chi2(const BAEdge& e) {
        if (e.plane) return e.err[0] * (e.info3[0] * e.err[0]) + e.err[1] * (e.info3[1] * e.err[1]) + e.err[2] * (e.info3[2] * e.err[2]);
        double s = e.err[0] * (e.info * e.err[0]) + e.err[1] * (e.info * e.err[1]);
        if (e.stereo) s += e.err[2] * (e.info * e.err[2]);
        return s;
    }
    void computeActiveErrors() override { for (int k : activeEdges) edgeError(edges[k]); }
    double activeRobustChi2() const override {
        double chi = 0;
        for (int k : activeEdges) {
            const BAEdge& e = edges[k];
            double c = chi2(e);
            if (e.robust) { double d = e.plane ? deltaPlane : e.stereo ? deltaStereo : deltaMono, r0, r1; huber(c, d, d * d, r0, r1); chi += r0; }
            else chi += c;
        }
        return chi;
    }
    // Jacobians: A (D x 3, point) and B (D x 6, pose)
    void jacobians(const BAEdge& e, double A[3][3], double B[3][6]) const {
        if (e.plane) {
            // g2o's numeric Jacobians (central differences, delta = 1e-9, core/base_binary_edge.hpp:131-205): vertex 0 =
            // the plane (VertexPlane::oplusImpl), vertex 1 = the keyframe (exp(delta) * T); a fixed keyframe is skipped
            const double delta = 1e-9, scalar = 1.0 / (2 * delta);
            const double* pl = planeOf(e);
            for (int d = 0; d < 3; d++) {
                double add[3] = {0, 0, 0}, q[4], ep[3], em[3];
                add[d] = delta;  plane_oplus(pl, add, q); plane_error(cams[e.cam], q, e.meas, ep);
                add[d] = -delta; plane_oplus(pl, add, q); plane_error(cams[e.cam], q, e.meas, em);
                for (int k = 0; k < 3; k++) A[k][d] = scalar * (ep[k] - em[k]);
            }
            for (int d = 0; d < 6; d++) {
                double add[6] = {0, 0, 0, 0, 0, 0}, ep[3] = {0, 0, 0}, em[3] = {0, 0, 0};
                if (!camFixed[e.cam]) {
                    add[d] = delta;  plane_error(se3_mul(se3_exp(add), cams[e.cam]), pl, e.meas, ep);
                    add[d] = -delta; plane_error(se3_mul(se3_exp(add), cams[e.cam]), pl, e.meas, em);
                }
                for (int k = 0; k < 3; k++) B[k][d] = scalar * (ep[k] - em[k]);
            }
            return;
        }
        double p[3], R[9];
        se3_map(cams[e.cam], &pts[3 * e.pt], p);
        quat_to_matrix(cams[e.cam].r, R);
        const double X = p[0], Y = p[1], Z = p[2], z2 = Z * Z;
        if (!e.stereo) {
            // -1/z * [fx 0 -x/z fx; 0 fy -y/z fy] * R
            double tmp[2][3] = {{fx, 0, -X / Z * fx}, {0, fy, -Y / Z * fy}};
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 3; j++) {
                    double s = (-1. / Z * tmp[i][0]) * R[j] + (-1. / Z * tmp[i][1]) * R[3 + j] + (-1. / Z * tmp[i][2]) * R[6 + j];
                    A[i][j] = s;
                }
        } else {
            for (int j = 0; j < 3; j++) {
                A[0][j] = -fx * R[j] / Z + fx * X * R[6 + j] / z2;
                A[1][j] = -fy * R[3 + j] / Z + fy * Y * R[6 + j] / z2;
                A[2][j] = A[0][j] - bf * R[6 + j] / z2;
            }
        }
        B[0][0] = X * Y / z2 * fx; B[0][1] = -(1 + (X * X / z2)) * fx; B[0][2] = Y / Z * fx;
        B[0][3] = -1. / Z * fx; B[0][4] = 0; B[0][5] = X / z2 * fx;
        B[1][0] = (1 + Y * Y / z2) * fy; B[1][1] = -X * Y / z2 * fy; B[1][2] = -X / Z * fy;
        B[1][3] = 0; B[1][4] = -1. / Z * fy; B[1][5] = Y / z2 * fy;
        if (e.stereo) {
            B[2][0] = B[0][0] - bf * Y / z2; B[2][1] = B[0][1] + bf * X / z2; B[2][2] = B[0][2];
            B[2][3] = B[0][3]; B[2][4] = 0; B[2][5] = B[0][5] - bf / z2;
        }
    }
    void buildSystem() override {
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(bvec.begin(), bvec.end(), 0.0);
        for (int k : activeEdges) {
            BAEdge& e = edges[k];
            double A[3][3], B[3][6];
            jacobians(e, A, B);
            const int D = (e.stereo || e.plane) ? 3 : 2;
            double w = 1.0;
            if (e.robust) { double d = e.plane ? deltaPlane : e.stereo ? deltaStereo : deltaMono, r0; huber(chi2(e), d, d * d, r0, w); }
            const double inf[3] = {e.plane ? e.info3[0] : e.info, e.plane ? e.info3[1] : e.info, e.plane ? e.info3[2] : e.info};
            const double wi[3] = {w * inf[0], w * inf[1], w * inf[2]};
            double omr[3];  // rho' * (-Omega e)
            for (int r = 0; r < D; r++) omr[r] = w * (-(inf[r] * e.err[r]));
            const int li = ptIdx[e.pt], pi = camIdx[e.cam];
            double* bl = &bvec[(size_t)nP * 6 + (size_t)li * 3];
            double* Hl = &Hll[(size_t)li * 9];
            for (int i = 0; i < 3; i++) {
                double s = 0; for (int r = 0; r < D; r++) s += A[r][i] * omr[r];
                bl[i] += s;
                for (int j = 0; j < 3; j++) { double h = 0; for (int r = 0; r < D; r++) h += A[r][i] * wi[r] * A[r][j]; Hl[i * 3 + j] += h; }
            }
            if (pi >= 0) {
                double* bp = &bvec[(size_t)pi * 6];
                double* Hp = &Hpp[(size_t)pi * 36];
                double* Hx = &Hpl[(size_t)k * 18];
                for (int i = 0; i < 6; i++) {
                    double s = 0; for (int r = 0; r < D; r++) s += B[r][i] * omr[r];
                    bp[i] += s;
                    for (int j = 0; j < 6; j++) { double h = 0; for (int r = 0; r < D; r++) h += B[r][i] * wi[r] * B[r][j]; Hp[i * 6 + j] += h; }
                    for (int j = 0; j < 3; j++) { double h = 0; for (int r = 0; r < D; r++) h += B[r][i] * wi[r] * A[r][j]; Hx[i * 3 + j] = h; }
                }
            }
        }
    }
    double maxDiagonal() const override {
        double m = 0;
        for (int p = 0; p < nP; p++) for (int j = 0; j < 6; j++) m = std::max(std::fabs(Hpp[(size_t)p * 36 + j * 7]), m);
        for (int l = 0; l < nL; l++) for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll[(size_t)l * 9 + j * 4]), m);
        return m;
    }
    bool solveWithLambda(double lambda) override {
        const int n = nP * 6;
        if (nL == 0) {  // no marginalised vertex: plain pose system
            std::vector<double> S((size_t)n * n, 0.0);
            for (int p = 0; p < nP; p++) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++)
                S[(size_t)(p * 6 + i) * n + p * 6 + j] = Hpp[(size_t)p * 36 + i * 6 + j] + (i == j ? lambda : 0.0);
            return ldlt_dense_solve(n, S, bvec.data(), x.data());
        }
        std::vector<double> S((size_t)n * n, 0.0), coeff(n, 0.0), Dinv((size_t)nL * 9);
        for (int p = 0; p < nP; p++) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++)
            S[(size_t)(p * 6 + i) * n + p * 6 + j] = Hpp[(size_t)p * 36 + i * 6 + j] + (i == j ? lambda : 0.0);
        const double* bl = &bvec[(size_t)nP * 6];
        for (int l = 0; l < nL; l++) {
            double Dm[9];
            for (int i = 0; i < 9; i++) Dm[i] = Hll[(size_t)l * 9 + i];
            Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
            double* Di = &Dinv[(size_t)l * 9];
            inv3_cofactor(Dm, Di);
            double db[3];
            for (int i = 0; i < 3; i++) db[i] = Di[i * 3] * bl[l * 3] + Di[i * 3 + 1] * bl[l * 3 + 1] + Di[i * 3 + 2] * bl[l * 3 + 2];
            const std::vector<int>& col = ptEdges[l];
            for (size_t a = 0; a < col.size(); a++) {
                const int i1 = camIdx[edges[col[a]].cam];
                const double* Bi = &Hpl[(size_t)col[a] * 18];
                double BDinv[18];
                for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++)
                    BDinv[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
                for (int r = 0; r < 6; r++) coeff[i1 * 6 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
                for (size_t b2 = a; b2 < col.size(); b2++) {
                    const int i2 = camIdx[edges[col[b2]].cam];
                    const double* Bj = &Hpl[(size_t)col[b2] * 18];
                    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++)
                        S[(size_t)(i1 * 6 + r) * n + i2 * 6 + c] -= BDinv[r * 3] * Bj[c * 3] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
                }
            }
        }
        std::vector<double> bs(n);
        for (int i = 0; i < n; i++) bs[i] = bvec[i] - coeff[i];
        if (n > 0 && !ldlt_dense_solve(n, S, bs.data(), x.data())) return false;
        // landmarks: xl = Dinv (bl - Hpl^T xp)
        for (int l = 0; l < nL; l++) {
            double cl[3] = {bl[l * 3], bl[l * 3 + 1], bl[l * 3 + 2]};
            for (int k : ptEdges[l]) {
                const int i1 = camIdx[edges[k].cam];
                const double* Bi = &Hpl[(size_t)k * 18];
                for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 6; r++) s += Bi[r * 3 + c] * (-x[i1 * 6 + r]); cl[c] += s; }
            }
            const double* Di = &Dinv[(size_t)l * 9];
            for (int i = 0; i < 3; i++) x[(size_t)n + l * 3 + i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
        }
        return true;
    }
    double computeScale(double lambda) const override { double s = 0; for (size_t j = 0; j < x.size(); j++) s += x[j] * (lambda * x[j] + bvec[j]); return s; }
    void push() override { camsBackup = cams; ptsBackup = pts; planesBackup = planes; }
    void pop() override { cams = camsBackup; pts = ptsBackup; planes = planesBackup; }
    void discardTop() override {}
    void applyUpdate() override {
        for (int p = 0; p < nP; p++) { int c = activeCams[p]; cams[c] = se3_mul(se3_exp(&x[(size_t)p * 6]), cams[c]); }
        const int npo = (int)pts.size() / 3;
        for (int l = 0; l < nL; l++) {
            const int q = activePts[l];
            if (q < npo) { for (int i = 0; i < 3; i++) pts[3 * q + i] += x[(size_t)nP * 6 + l * 3 + i]; }
            else { double out[4]; plane_oplus(&planes[4 * (size_t)(q - npo)], &x[(size_t)nP * 6 + l * 3], out); for (int i = 0; i < 4; i++) planes[4 * (size_t)(q - npo) + i] = out[i]; }
        }
    }
};

}  // namespace

// ================================================================================ C surface (ctypes)
extern "C" {

struct orc_pose_problem {
    int32_t n;
    const float* Tcw;         // 16, row-major
    const float* Xw;          // n*3
    const float* obs;         // n*3 (u, v, ur); ur < 0 => monocular
    const float* inv_sigma2;  // n
    float fx, fy, cx, cy, bf;
};
struct orc_trace { int32_t n; double lambda[64]; double chi2[64]; int32_t trials[64]; };

// returns nInitialCorrespondences - nBad; Tcw_out float[16]; outlier[n]; pose_d = final (qx,qy,qz,qw,tx,ty,tz) in fp64.
// Planes (src/Optimizer.cc:456-535, 626-658): plane_world / plane_obs 4 floats each (MapPlane::GetWorldPos(),
// Frame::mvPlaneCoefficients[i]), plane_seen[i] = MapPlane::mbSeen, plane_outlier[i] = Frame::mvbPlaneOutlier[i] on return.
int orc_pose_optimization_planes(const orc_pose_problem* P, int n_planes, const float* plane_world, const float* plane_obs,
                                 const uint8_t* plane_seen, float* Tcw_out, uint8_t* outlier, uint8_t* plane_outlier, double* pose_d,
                                 orc_trace* trace) {
    const int n = P->n;
    PoseProblem pb;
    pb.fx = P->fx; pb.fy = P->fy; pb.cx = P->cx; pb.cy = P->cy; pb.bf = P->bf;
    pb.deltaMono = (float)std::sqrt(refc::POSE_HUBER2_MONO); pb.deltaStereo = (float)std::sqrt(refc::POSE_HUBER2_STEREO);
    pb.edges.resize(n);
    for (int i = 0; i < n; i++) {
        PoseEdge& e = pb.edges[i];
        for (int k = 0; k < 3; k++) { e.Xw[k] = P->Xw[3 * i + k]; e.obs[k] = P->obs[3 * i + k]; }
        e.stereo = !(P->obs[3 * i + 2] < 0);
        e.info = P->inv_sigma2[i];
        outlier[i] = 0;
    }
    if (trace) trace->n = 0;
    if (n < 3) { std::memcpy(Tcw_out, P->Tcw, 16 * sizeof(float)); return 0; }   // checked BEFORE the planes are added (:453-454)
    const double angleInfo = refc::PLANE_ANGLE_INFO / (1.0 * 1.0), disInfo = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT, planeChi = refc::PLANE_CHI2;
    pb.deltaPlane = (float)std::sqrt(planeChi);
    pb.planes.resize(n_planes);
    for (int i = 0; i < n_planes; i++) {
        PlaneEdge& e = pb.planes[i];
        plane_from_f32(plane_world + 4 * i, e.world);
        plane_from_f32(plane_obs + 4 * i, e.meas);
        const double f = plane_seen[i] ? 1.0 : 2.0;
        e.infoA = f * angleInfo; e.infoD = f * disInfo;
        plane_outlier[i] = 0;
    }
    const int nInitial = n + n_planes;
    const float chi2Mono = refc::POSE_CHI2_MONO, chi2Stereo = refc::POSE_CHI2_STEREO;
    int nBad = 0;
    pb.est = se3_from_Tcw_f32(P->Tcw);
    for (int it = 0; it < refc::POSE_ROUNDS; it++) {
        pb.est = se3_from_Tcw_f32(P->Tcw);
        LM lm;
        LMTrace tr;
        lm.optimize(pb, refc::POSE_ITS, nullptr, &tr);
        if (trace) for (size_t k = 0; k < tr.lambda.size() && trace->n < 64; k++) {
            trace->lambda[trace->n] = tr.lambda[k]; trace->chi2[trace->n] = tr.chi2[k]; trace->trials[trace->n] = tr.trials[k]; trace->n++;
        }
        nBad = 0;
        for (int i = 0; i < n; i++) {
            PoseEdge& e = pb.edges[i];
            if (outlier[i]) pb.edgeError(e);
            const float c = (float)PoseProblem::chi2(e);
            if (c > (e.stereo ? chi2Stereo : chi2Mono)) { outlier[i] = 1; e.level = 1; nBad++; }
            else { outlier[i] = 0; e.level = 0; }
            if (!e.stereo) e.robust = false;             // mono: kernel dropped after every round (:584-586)
            else if (it == refc::POSE_UNROBUST_ROUND) e.robust = false;          // stereo: at it == 2 (:620-621)
        }
        for (int i = 0; i < n_planes; i++) {             // :626-658
            PlaneEdge& e = pb.planes[i];
            if (plane_outlier[i]) plane_error(pb.est, e.world, e.meas, e.err);
            const float c = (float)PoseProblem::chi2(e);
            if (c > planeChi) { plane_outlier[i] = 1; e.level = 1; nBad++; }
            else { e.level = 0; plane_outlier[i] = 0; }
            if (it == refc::POSE_UNROBUST_ROUND) e.robust = false;
        }
        if (nInitial < 10) break;
    }
    se3_to_Tcw_f32(pb.est, Tcw_out);
    if (pose_d) { pose_d[0] = pb.est.r.x; pose_d[1] = pb.est.r.y; pose_d[2] = pb.est.r.z; pose_d[3] = pb.est.r.w;
                  pose_d[4] = pb.est.t[0]; pose_d[5] = pb.est.t[1]; pose_d[6] = pb.est.t[2]; }
    return nInitial - nBad;
}
int orc_pose_optimization(const orc_pose_problem* P, float* Tcw_out, uint8_t* outlier, double* pose_d, orc_trace* trace) {
    return orc_pose_optimization_planes(P, 0, nullptr, nullptr, nullptr, Tcw_out, outlier, nullptr, pose_d, trace);
}

struct orc_ba_problem {
    int32_t n_cams, n_points, n_edges;
    const float* cam_Tcw;          // n_cams*16
    const uint8_t* cam_fixed;      // n_cams
    const float* points;           // n_points*3
    const int32_t* edge_cam;       // n_edges
    const int32_t* edge_point;     // n_edges
    const float* edge_obs;         // n_edges*3
    const float* edge_inv_sigma2;  // n_edges
    float fx, fy, cx, cy, bf;
    int32_t its_first, its_second;
};

// out: cam_Tcw_out (n_cams*16 f32), points_out (n_points*3 f32), edge_outlier (n_edges), cams_d (n_cams*7 fp64),
// points_d (n_points*3 fp64), iters[2], trace (both passes concatenated).  Returns 0, or 1 if aborted before optimising.
int orc_local_ba(const orc_ba_problem* P, const volatile bool* stop, float* cam_Tcw_out, float* points_out,
                 uint8_t* edge_outlier, double* cams_d, double* points_d, int32_t* iters, orc_trace* trace) {
    BAProblem pb;
    pb.fx = P->fx; pb.fy = P->fy; pb.cx = P->cx; pb.cy = P->cy; pb.bf = P->bf; pb.bf_f = P->bf;
    pb.deltaMono = (float)std::sqrt(refc::LBA_HUBER2_MONO); pb.deltaStereo = (float)std::sqrt(refc::LBA_HUBER2_STEREO);
    pb.cams.resize(P->n_cams); pb.camFixed.assign(P->cam_fixed, P->cam_fixed + P->n_cams);
    for (int c = 0; c < P->n_cams; c++) pb.cams[c] = se3_from_Tcw_f32(P->cam_Tcw + 16 * c);
    pb.pts.resize((size_t)P->n_points * 3);
    for (size_t i = 0; i < pb.pts.size(); i++) pb.pts[i] = P->points[i];
    pb.edges.resize(P->n_edges);
    for (int k = 0; k < P->n_edges; k++) {
        BAEdge& e = pb.edges[k];
        e.cam = P->edge_cam[k]; e.pt = P->edge_point[k];
        for (int j = 0; j < 3; j++) e.obs[j] = P->edge_obs[3 * k + j];
        e.stereo = !(P->edge_obs[3 * k + 2] < 0);
        e.info = P->edge_inv_sigma2[k];
    }
    if (trace) trace->n = 0;
    if (iters) { iters[0] = iters[1] = 0; }
    auto write_out = [&]() {
        for (int c = 0; c < P->n_cams; c++) {
            se3_to_Tcw_f32(pb.cams[c], cam_Tcw_out + 16 * c);
            if (cams_d) { double* d = cams_d + 7 * c; d[0] = pb.cams[c].r.x; d[1] = pb.cams[c].r.y; d[2] = pb.cams[c].r.z; d[3] = pb.cams[c].r.w;
                          d[4] = pb.cams[c].t[0]; d[5] = pb.cams[c].t[1]; d[6] = pb.cams[c].t[2]; }
        }
        for (size_t i = 0; i < pb.pts.size(); i++) { points_out[i] = (float)pb.pts[i]; if (points_d) points_d[i] = pb.pts[i]; }
    };
    std::memset(edge_outlier, 0, P->n_edges);
    if (stop && *stop) { write_out(); return 1; }
    auto add_trace = [&](const LMTrace& tr) {
        if (trace) for (size_t k = 0; k < tr.lambda.size() && trace->n < 64; k++) {
            trace->lambda[trace->n] = tr.lambda[k]; trace->chi2[trace->n] = tr.chi2[k]; trace->trials[trace->n] = tr.trials[k]; trace->n++;
        }
    };
    {
        pb.initialize();
        LM lm; LMTrace tr;
        int it = lm.optimize(pb, P->its_first, stop, &tr);
        if (iters) iters[0] = it;
        add_trace(tr);
    }
    bool doMore = !(stop && *stop);
    if (doMore) {
        for (BAEdge& e : pb.edges) {
            const double th = e.stereo ? refc::LBA_CHI2_STEREO : refc::LBA_CHI2_MONO;
            if (BAProblem::chi2(e) > th || !pb.depthPositive(e)) e.level = 1;
            e.robust = false;
        }
        pb.initialize();
        LM lm; LMTrace tr;
        int it = lm.optimize(pb, P->its_second, stop, &tr);
        if (iters) iters[1] = it;
        add_trace(tr);
    }
    for (int k = 0; k < P->n_edges; k++) {
        const BAEdge& e = pb.edges[k];
        const double th = e.stereo ? refc::LBA_CHI2_STEREO : refc::LBA_CHI2_MONO;
        edge_outlier[k] = (BAProblem::chi2(e) > th || !pb.depthPositive(e)) ? 1 : 0;
    }
    write_out();
    return 0;
}

// Optimizer::BundleAdjustment over keyframes, map points and map planes (src/Optimizer.cc:55-323): every point edge robust iff bRobust with delta sqrt(5.99) / sqrt(7.815) (:94-95), ONE initializeOptimization() +
// optimize(iterations) (:254-255), estimates written back (:258-300).  A map point without an edge is removed from the graph
// (:193-201) and keeps its position.  Same outputs as orc_local_ba (iters[1] = 0).
// Planes (:203-252): n_planes MapPlane vertices (plane_world 4 floats each, MapPlane::GetWorldPos()), marginalised 3-dof landmarks
// with ids beyond the points; n_pedges EdgePlane edges (plane index, camera, measured coefficients KeyFrame::mvPlaneCoefficients[.]),
// information diag(3282.8, 3282.8, 1e4), ALWAYS a Huber kernel with delta sqrt(300), numeric Jacobians on both vertices.  An edge
// whose keyframe is not in the graph is skipped by the caller (:234-235).  planes_out: 4 floats per plane (Converter::toCvMat(Plane3D)).
struct orc_ba_planes { int32_t n_planes; const float* plane_world; int32_t n_pedges; const int32_t* pedge_plane; const int32_t* pedge_cam; const float* pedge_obs; };
int orc_bundle_adjustment_planes(const orc_ba_problem* P, const orc_ba_planes* PL, int32_t iterations, int32_t robust, const volatile bool* stop,
                                 float* cam_Tcw_out, float* points_out, float* planes_out, double* cams_d, double* points_d, int32_t* iters, orc_trace* trace) {
    BAProblem pb;
    pb.fx = P->fx; pb.fy = P->fy; pb.cx = P->cx; pb.cy = P->cy; pb.bf = P->bf; pb.bf_f = P->bf;
    pb.deltaMono = (float)std::sqrt(refc::GBA_HUBER2_MONO); pb.deltaStereo = (float)std::sqrt(refc::GBA_HUBER2_STEREO);
    pb.cams.resize(P->n_cams); pb.camFixed.assign(P->cam_fixed, P->cam_fixed + P->n_cams);
    for (int c = 0; c < P->n_cams; c++) pb.cams[c] = se3_from_Tcw_f32(P->cam_Tcw + 16 * c);
    pb.pts.resize((size_t)P->n_points * 3);
    for (size_t i = 0; i < pb.pts.size(); i++) pb.pts[i] = P->points[i];
    pb.edges.resize(P->n_edges);
    for (int k = 0; k < P->n_edges; k++) {
        BAEdge& e = pb.edges[k];
        e.cam = P->edge_cam[k]; e.pt = P->edge_point[k];
        for (int j = 0; j < 3; j++) e.obs[j] = P->edge_obs[3 * k + j];
        e.stereo = !(P->edge_obs[3 * k + 2] < 0);
        e.info = P->edge_inv_sigma2[k];
        e.robust = robust != 0;
    }
    if (PL && PL->n_planes > 0) {
        pb.deltaPlane = (float)std::sqrt(refc::PLANE_CHI2);
        pb.planes.resize((size_t)PL->n_planes * 4);
        for (int i = 0; i < PL->n_planes; i++) plane_from_f32(PL->plane_world + 4 * i, &pb.planes[4 * (size_t)i]);
        const double angleInfo = refc::PLANE_ANGLE_INFO / (1.0 * 1.0), disInfo = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT;
        for (int k = 0; k < PL->n_pedges; k++) {
            BAEdge e;
            e.cam = PL->pedge_cam[k]; e.pt = P->n_points + PL->pedge_plane[k];
            e.plane = true; e.stereo = false; e.info = 0; e.robust = true;
            plane_from_f32(PL->pedge_obs + 4 * k, e.meas);
            e.info3[0] = angleInfo; e.info3[1] = angleInfo; e.info3[2] = disInfo;
            pb.edges.push_back(e);
        }
    }
    if (trace) trace->n = 0;
    if (iters) { iters[0] = iters[1] = 0; }
    pb.initialize();
    LM lm; LMTrace tr;
    const int it = lm.optimize(pb, iterations, stop, &tr);
    if (iters) iters[0] = it;
    if (trace) for (size_t k = 0; k < tr.lambda.size() && trace->n < 64; k++) {
        trace->lambda[trace->n] = tr.lambda[k]; trace->chi2[trace->n] = tr.chi2[k]; trace->trials[trace->n] = tr.trials[k]; trace->n++;
    }
    for (int c = 0; c < P->n_cams; c++) {
        se3_to_Tcw_f32(pb.cams[c], cam_Tcw_out + 16 * c);
        if (cams_d) { double* d = cams_d + 7 * c; d[0] = pb.cams[c].r.x; d[1] = pb.cams[c].r.y; d[2] = pb.cams[c].r.z; d[3] = pb.cams[c].r.w;
                      d[4] = pb.cams[c].t[0]; d[5] = pb.cams[c].t[1]; d[6] = pb.cams[c].t[2]; }
    }
    for (size_t i = 0; i < pb.pts.size(); i++) { points_out[i] = (float)pb.pts[i]; if (points_d) points_d[i] = pb.pts[i]; }
    if (planes_out) for (size_t i = 0; i < pb.planes.size(); i++) planes_out[i] = (float)pb.planes[i];
    return 0;
}
int orc_bundle_adjustment(const orc_ba_problem* P, int32_t iterations, int32_t robust, const volatile bool* stop, float* cam_Tcw_out,
                          float* points_out, double* cams_d, double* points_d, int32_t* iters, orc_trace* trace) {
    return orc_bundle_adjustment_planes(P, nullptr, iterations, robust, stop, cam_Tcw_out, points_out, nullptr, cams_d, points_d, iters, trace);
}

// ---- known-answer hooks for tests -------------------------------------------------------------
// residual of one binary edge at (Tcw given as quaternion+translation fp64, point fp64)
void orc_ba_edge_eval(const double* cam7, const double* pt3, const double* obs3, int stereo,
                      double fx, double fy, double cx, double cy, double bf, double* err3, double* A9, double* B18) {
    BAProblem pb;
    pb.fx = fx; pb.fy = fy; pb.cx = cx; pb.cy = cy; pb.bf = bf; pb.bf_f = (float)bf;
    SE3 s; s.r = {cam7[0], cam7[1], cam7[2], cam7[3]}; s.t[0] = cam7[4]; s.t[1] = cam7[5]; s.t[2] = cam7[6];
    pb.cams.push_back(s); pb.pts.assign(pt3, pt3 + 3);
    BAEdge e; e.cam = 0; e.pt = 0; e.stereo = stereo != 0; e.info = 1;
    for (int i = 0; i < 3; i++) e.obs[i] = obs3[i];
    pb.edgeError(e);
    for (int i = 0; i < 3; i++) err3[i] = e.err[i];
    double A[3][3] = {{0}}, B[3][6] = {{0}};
    pb.jacobians(e, A, B);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A9[i * 3 + j] = A[i][j]; for (int j = 0; j < 6; j++) B18[i * 6 + j] = B[i][j]; }
}
// exp(update) * T, quaternion+translation in/out (tests the manifold update used for numeric Jacobians)
void orc_se3_oplus(const double* cam7, const double* upd6, double* out7) {
    SE3 s; s.r = {cam7[0], cam7[1], cam7[2], cam7[3]}; s.t[0] = cam7[4]; s.t[1] = cam7[5]; s.t[2] = cam7[6];
    SE3 r = se3_mul(se3_exp(upd6), s);
    out7[0] = r.r.x; out7[1] = r.r.y; out7[2] = r.r.z; out7[3] = r.r.w; out7[4] = r.t[0]; out7[5] = r.t[1]; out7[6] = r.t[2];
}
void orc_huber(double e, double delta, double* rho2) { huber(e, delta, delta * delta, rho2[0], rho2[1]); }
// Plane3D::oplus / EdgePlane::computeError on raw coefficients (cam7 = qx qy qz qw tx ty tz)
void orc_plane_oplus(const double* c4, const double* v3, double* out4) { plane_oplus(c4, v3, out4); }
void orc_plane_error(const double* cam7, const double* world4, const double* meas4, double* err3) {
    SE3 T; T.r = Quat{cam7[0], cam7[1], cam7[2], cam7[3]}; T.t[0] = cam7[4]; T.t[1] = cam7[5]; T.t[2] = cam7[6];
    plane_error(T, world4, meas4, err3);
}
void orc_Tcw_to_cam7(const float* T, double* out7) {
    SE3 s = se3_from_Tcw_f32(T);
    out7[0] = s.r.x; out7[1] = s.r.y; out7[2] = s.r.z; out7[3] = s.r.w; out7[4] = s.t[0]; out7[5] = s.t[1]; out7[6] = s.t[2];
}

}  // extern "C"
