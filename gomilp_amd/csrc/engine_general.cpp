// General initial basis for standard forms whose trailing columns are not distinct unit
// vectors, i.e. LPs with equality rows (api.go `EqualTo`, /root/reference/ilp_test.go cases without slacks).
//
// findLinearlyIndependent (simplex.go:611-637) walks the columns n-1 -> 0 and accepts a column when
// mat.Cond(columns[:, :k+1], 1) <= 1e12.  The reference evaluates that through Householder QR + the Hager/Higham
// estimate of kappa_1(R) (tall case) or LU + estimate (square case).  This host routine makes the same decisions from
// the EXACT 1-norm condition numbers: kappa_1(R) = |R|_1 |R^-1|_1 with R from a Householder QR (tall), and
// kappa_1(A) = |A|_1 |A^-1|_1 (square).  The estimator is a lower bound within a small factor of these values, so the
// two can only disagree for matrices whose condition number is within that factor of 1e12.
// Small problems only (the work is O(m^4) like the reference's); the device pipelines then start from B^-1 computed
// here instead of a permutation.
#include "engine_work.hpp"

namespace gomilp {

namespace {

double norm1(const std::vector<double> &M, int rows, int cols, int ld) {
    double best = 0;
    for (int j = 0; j < cols; j++) {
        double s = 0;
        for (int i = 0; i < rows; i++) s += fabs(M[(size_t)i * ld + j]);
        if (s != s) return s;
        best = std::max(best, s);
    }
    return best;
}

// Gauss-Jordan inverse with partial pivoting; false when a pivot is exactly zero / not finite
bool invert(const std::vector<double> &A, int n, std::vector<double> &inv) {
    std::vector<double> W(A);
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = fabs(W[(size_t)k * n + k]);
        for (int i = k + 1; i < n; i++) {
            const double v = fabs(W[(size_t)i * n + k]);
            if (v > best) { best = v; p = i; }
        }
        if (!(best > 0) || !std::isfinite(best)) return false;
        if (p != k)
            for (int j = 0; j < n; j++) { std::swap(W[(size_t)k * n + j], W[(size_t)p * n + j]); std::swap(inv[(size_t)k * n + j], inv[(size_t)p * n + j]); }
        const double d = W[(size_t)k * n + k];
        for (int j = 0; j < n; j++) { W[(size_t)k * n + j] /= d; inv[(size_t)k * n + j] /= d; }
        for (int i = 0; i < n; i++) {
            if (i == k) continue;
            const double f = W[(size_t)i * n + k];
            if (f == 0) continue;
            for (int j = 0; j < n; j++) { W[(size_t)i * n + j] -= f * W[(size_t)k * n + j]; inv[(size_t)i * n + j] -= f * inv[(size_t)k * n + j]; }
        }
    }
    return true;
}

// exact kappa_1 of the m x k (k <= m) matrix C (row-major, ld = m, columns 0..k-1)
double cond1_exact(const std::vector<double> &C, int m, int k) {
    if (k == m) {
        std::vector<double> A((size_t)m * m), inv;
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) A[(size_t)i * m + j] = C[(size_t)i * m + j];
        if (!invert(A, m, inv)) return std::numeric_limits<double>::infinity();
        return norm1(A, m, m, m) * norm1(inv, m, m, m);
    }
    // Householder QR of the m x k block -> R (k x k upper)
    std::vector<double> Q((size_t)m * k);
    for (int i = 0; i < m; i++) for (int j = 0; j < k; j++) Q[(size_t)i * k + j] = C[(size_t)i * m + j];
    for (int j = 0; j < k; j++) {
        double nrm = 0;
        for (int i = j; i < m; i++) nrm = hypot(nrm, Q[(size_t)i * k + j]);
        if (nrm == 0) continue;
        const double alpha = Q[(size_t)j * k + j];
        const double beta = alpha >= 0 ? -nrm : nrm;
        std::vector<double> v(m - j);
        v[0] = alpha - beta;
        for (int i = j + 1; i < m; i++) v[i - j] = Q[(size_t)i * k + j];
        double vv = 0;
        for (double x : v) vv += x * x;
        if (vv == 0) continue;
        for (int c = j; c < k; c++) {
            double dot = 0;
            for (int i = j; i < m; i++) dot += v[i - j] * Q[(size_t)i * k + c];
            const double f = 2 * dot / vv;
            for (int i = j; i < m; i++) Q[(size_t)i * k + c] -= f * v[i - j];
        }
    }
    std::vector<double> R((size_t)k * k, 0.0), Rinv;
    for (int i = 0; i < k; i++) for (int j = i; j < k; j++) R[(size_t)i * k + j] = Q[(size_t)i * k + j];
    if (!invert(R, k, Rinv)) return std::numeric_limits<double>::infinity();
    return norm1(R, k, k, k) * norm1(Rinv, k, k, k);
}

}  // namespace

// findLinearlyIndependent on the host copy of A.  Returns the accepted columns in scan order (basis position order).
int general_find_linearly_independent(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs) {
    idxs.clear();
    std::vector<double> columns((size_t)m * m, 0.0);
    for (int i = n - 1; i >= 0; i--) {
        if ((int)idxs.size() == m) break;
        const int k = (int)idxs.size();
        for (int r = 0; r < m; r++) columns[(size_t)r * m + k] = A[(size_t)r * n + i];
        if (k == 0) { idxs.push_back(i); continue; }  // simplex.go:624-629
        if (cond1_exact(columns, m, k + 1) > 1e12) continue;  // :630 not linearly independent
        idxs.push_back(i);
    }
    return (int)idxs.size() == m ? GOMILP_OK : GOMILP_ERR_SINGULAR;  // :495-497
}

// B^-1 of the basis made of the columns `basic` of A (row-major, m x n)
bool general_basis_inverse(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, int ncols_with_art,
                           const std::vector<double> &art, std::vector<double> &binv) {
    std::vector<double> B((size_t)m * m);
    for (int i = 0; i < m; i++)
        for (int p = 0; p < m; p++) {
            const int j = basic[p];
            B[(size_t)i * m + p] = (j < n) ? A[(size_t)i * n + j] : art[i];
        }
    (void)ncols_with_art;
    return invert(B, m, binv);
}

// ---- condition guards of gonum's LU.Solve (mat/lu.go:301,321) for SMALL bases, by replaying the pivot sequence on the host
// The reference factorises the basis three times per pivot and leaves its loop with the current point when a solve reports
// mat.Condition (cond > 1e16 or Det() == 0: simplex.go:236-239, :289-292) or lp.ErrLinSolve (computeMove, :316-318).  The
// device loop never forms those factorizations; for bases of up to 64 rows the host re-walks the recorded pivots with the EXACT
// condition numbers (kappa_1 for the solve with ab^T, kappa_inf for the two solves with ab — the Hager / Higham estimate the
// reference uses is a lower bound within a small factor of these) and reports where the reference would have
// stopped.  Returns the number of pivots the reference would have performed (== npiv: no guard fired) and, when a guard
// fired, the status and the basis whose x_B the reference returns.
static double norm_inf(const std::vector<double> &M, int n) {
    double best = 0;
    for (int i = 0; i < n; i++) {
        double s2 = 0;
        for (int j = 0; j < n; j++) s2 += fabs(M[(size_t)i * n + j]);
        if (s2 != s2) return s2;
        best = std::max(best, s2);
    }
    return best;
}
static void basis_matrix(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, std::vector<double> &B) {
    B.resize((size_t)m * m);
    for (int i = 0; i < m; i++) for (int p = 0; p < m; p++) B[(size_t)i * m + p] = A[(size_t)i * n + basic[p]];
}
int general_condition_replay(const std::vector<double> &A, int m, int n, std::vector<int32_t> &basic, const std::vector<std::pair<int, int>> &pivots,
                             bool ended_in_compute_move, int *status_out, int64_t *evaluations) {
    *status_out = GOMILP_OK;
    std::vector<double> B, inv;
    const double lim = 1e16;   // mat.ConditionTolerance (mat/errors.go:33)
    auto conds = [&](double &k1, double &kinf) {
        basis_matrix(A, m, n, basic, B);
        (*evaluations)++;
        if (!invert(B, m, inv)) { k1 = kinf = std::numeric_limits<double>::infinity(); return; }
        k1 = norm1(B, m, m, m) * norm1(inv, m, m, m);
        kinf = norm_inf(B, m) * norm_inf(inv, m);
    };
    double k1, kinf;
    conds(k1, kinf);
    for (size_t t = 0;; t++) {
        if (k1 > lim || k1 != k1) { *status_out = GOMILP_ERR_CONDITION; return (int)t; }          // duals: SolveVec(ab^T, cb), :236-239
        if (t == pivots.size() && !ended_in_compute_move) return (int)t;                          // (the optimality test ended the loop)
        if (kinf > lim || kinf != kinf) { *status_out = GOMILP_ERR_LINSOLVE; return (int)t; }     // computeMove, :316-318 (before its unbounded test)
        if (t == pivots.size()) return (int)t;
        basic[pivots[t].first] = pivots[t].second;                                                // the swap of :280
        conds(k1, kinf);
        if (kinf > lim || kinf != kinf) { *status_out = GOMILP_ERR_CONDITION; return (int)t + 1; } // x_B of the new basis, :289-292
    }
}

// exact kappa_inf of a square matrix (m == n path: any Condition of the single solve becomes lp.ErrSingular, simplex.go:109-112)
double general_cond_inf(const std::vector<double> &A, int n) {
    std::vector<double> inv;
    if (!invert(A, n, inv)) return std::numeric_limits<double>::infinity();
    return norm_inf(A, n) * norm_inf(inv, n);
}

// plain LU solve B x = b for the point the reference returns with a mid-loop error (small bases)
bool general_solve_basis(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, const std::vector<double> &b, std::vector<double> &x) {
    std::vector<double> B, inv;
    basis_matrix(A, m, n, basic, B);
    if (!invert(B, m, inv)) return false;
    x.assign(m, 0.0);
    for (int i = 0; i < m; i++) { double s2 = 0; for (int j = 0; j < m; j++) s2 += inv[(size_t)i * m + j] * b[j]; x[i] = s2; }
    return true;
}

}  // namespace gomilp
