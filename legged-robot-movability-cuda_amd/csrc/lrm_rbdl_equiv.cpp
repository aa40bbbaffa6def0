// lrm_rbdl_equiv.cpp -- "RBDL-equivalent" CPU baseline: what rbdl_benchmark.cpp:18-111 (apply_RBDL) asks RBDL to
// do, restated for the one chain it builds.  PARITY UNPINNED: RBDL is an external dependency of the reference
// (find_package(RBDL REQUIRED), CMakeLists.txt:26; no version pinned, not vendored, absent from this image), no
// fixture of the reference holds its outputs, and the reference never compares them with the analytic mask (the
// RBDL model has no joint limits and no coxa pitch).  This is a TIMING baseline and nothing else.
//
// What apply_RBDL does (rbdl_benchmark.cpp):
//   :29-54   chain root -(Xtrans(body,0,0)/400, revolute Z)- a -(Xtrans(coxa_length)/400, revolute Y)- b
//            -(Xtrans(femur_length)/400, revolute Y)- c -(Xtrans(tibia_length)/400, fixed)- d; the tip is d's origin;
//   :81      Cs.max_steps = 10;  (lambda = 1e-9, step_tol = 1e-12: RBDL's InverseKinematicsConstraintSet defaults)
//   :85-103  per target: one point constraint (tip -> target / 400); InverseKinematics from q = 0; on failure
//            up to 4 more starts from Vector3d::Random() * 3.14 * 2 (uniform in [-2 pi, 2 pi]); out[i] = converged.
// RBDL's InverseKinematics(model, Qinit, CS, Qres) is the Levenberg-Marquardt iteration published with the library
// (Kinematics.cc, "task space / joint space" weights after the Puppeteer tool): per step
//   e = target - tip(q);  stop with success when |e| < step_tol;
//   q += J^T (J J^T + Ek)^-1 e,  Ek = diag(e_i^2 / 2 + lambda);  success when |dq| < step_tol.
// Here: the same iteration in double precision with the closed-form forward kinematics and 3x3 Jacobian of the
// chain (RBDL evaluates them with its generic spatial-algebra recursion and dynamically sized Eigen matrices, which
// is where most of its 14.6 us per point on an i5-12600K goes, bdata/pc/rbdl.csv: this restatement is an order of
// magnitude leaner per iteration -- a FASTER baseline than the real library, never a slower one).
#include <chrono>
#include <cmath>
#include <cstdint>
#include "../../include/lrm.h"

namespace {

struct Chain {
    double body, coxa, femur, tibia; // already divided by 400
};

// tip(q) = (body,0,0) + Rz(q0) [ (coxa,0,0) + Ry(q1) [ (femur,0,0) + Ry(q2) (tibia,0,0) ] ]
inline void fk_jac(const Chain& c, const double q[3], double p[3], double J[3][3]) {
    const double c0 = std::cos(q[0]), s0 = std::sin(q[0]);
    const double c1 = std::cos(q[1]), s1 = std::sin(q[1]);
    const double c12 = std::cos(q[1] + q[2]), s12 = std::sin(q[1] + q[2]);
    // in the coxa frame (after the yaw): Ry(a)(L,0,0) = (L cos a, 0, -L sin a)
    const double rx = c.coxa + c.femur * c1 + c.tibia * c12;
    const double rz = -(c.femur * s1 + c.tibia * s12);
    p[0] = c.body + c0 * rx;
    p[1] = s0 * rx;
    p[2] = rz;
    const double drx1 = -(c.femur * s1 + c.tibia * s12), drx2 = -c.tibia * s12;
    const double drz1 = -(c.femur * c1 + c.tibia * c12), drz2 = -c.tibia * c12;
    J[0][0] = -s0 * rx; J[0][1] = c0 * drx1; J[0][2] = c0 * drx2;
    J[1][0] = c0 * rx;  J[1][1] = s0 * drx1; J[1][2] = s0 * drx2;
    J[2][0] = 0.0;      J[2][1] = drz1;      J[2][2] = drz2;
}

// 3x3 solve with partial pivoting (RBDL uses a rank-revealing QR; the systems here are 3x3 and regularised)
inline bool solve3(double A[3][3], double b[3], double x[3]) {
    int piv[3] = {0, 1, 2};
    for (int k = 0; k < 3; k++) {
        int best = k;
        for (int r = k + 1; r < 3; r++)
            if (std::fabs(A[piv[r]][k]) > std::fabs(A[piv[best]][k])) best = r;
        const int t = piv[k]; piv[k] = piv[best]; piv[best] = t;
        const double d = A[piv[k]][k];
        if (d == 0.0) return false;
        for (int r = k + 1; r < 3; r++) {
            const double f = A[piv[r]][k] / d;
            for (int cc = k; cc < 3; cc++) A[piv[r]][cc] -= f * A[piv[k]][cc];
            b[piv[r]] -= f * b[piv[k]];
        }
    }
    for (int k = 2; k >= 0; k--) {
        double s = b[piv[k]];
        for (int cc = k + 1; cc < 3; cc++) s -= A[piv[k]][cc] * x[cc];
        x[k] = s / A[piv[k]][k];
    }
    return true;
}

inline bool ik_lm(const Chain& c, const double target[3], double q[3], int max_steps, double lambda, double step_tol) {
    for (int step = 0; step < max_steps; step++) {
        double p[3], J[3][3], e[3];
        fk_jac(c, q, p, J);
        for (int i = 0; i < 3; i++) e[i] = target[i] - p[i];
        if (std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) < step_tol) return true;
        // dq = J^T (J J^T + Ek)^-1 e,  Ek = diag(e_i^2 / 2 + lambda): damping that vanishes with the error
        double A[3][3], b[3] = {e[0], e[1], e[2]}, y[3], dq[3];
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++)
                A[r][k] = J[r][0] * J[k][0] + J[r][1] * J[k][1] + J[r][2] * J[k][2] + (r == k ? 0.5 * e[r] * e[r] + lambda : 0.0);
        if (!solve3(A, b, y)) return false;
        for (int k = 0; k < 3; k++) dq[k] = J[0][k] * y[0] + J[1][k] * y[1] + J[2][k] * y[2];
        for (int i = 0; i < 3; i++) q[i] += dq[i];
        if (std::sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < step_tol) return true;
    }
    return false;
}

} // namespace

extern "C" int lrm_rbdl_equiv_cpu(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, uint8_t* mask_out,
                                  double* ms) {
    if (!leg || (n && (!xyz_aos || !mask_out))) return LRM_EINVAL;
    constexpr double fact = 400.0; // rbdl_benchmark.cpp:30
    const Chain c{leg->body / fact, leg->coxa_length / fact, leg->femur_length / fact, leg->tibia_length / fact};
    uint64_t rng = 0x9e3779b97f4a7c15ull; // Eigen's Random() is std::rand-based: any uniform stream will do
    auto uniform_pm1 = [&]() {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(rng >> 11) / (double)(1ull << 53) * 2.0 - 1.0;
    };
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (size_t i = 0; i < n; i++) {
        const double target[3] = {xyz_aos[3 * i] / fact, xyz_aos[3 * i + 1] / fact, xyz_aos[3 * i + 2] / fact};
        double q[3] = {0.0, 0.0, 0.0};
        bool valid = false;
        for (int s = 0; s < 5; s++) { // substep = 5, rbdl_benchmark.cpp:83
            valid = ik_lm(c, target, q, /*max_steps=*/10, /*lambda=*/1e-9, /*step_tol=*/1e-12);
            if (valid) break;
            for (int k = 0; k < 3; k++) q[k] = uniform_pm1() * 3.14 * 2;
        }
        mask_out[i] = valid;
    }
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (ms) *ms = std::chrono::duration<double>(t1 - t0).count() * 1000.0;
    return LRM_OK;
}
