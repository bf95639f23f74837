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

#include <atomic>
#include <thread>

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
//
// One Householder QR is carried along instead of a fresh factorisation per candidate (the reference's cost: sum_k m k^2 ~ m^4/3
// flop, hours at m = 2048): a candidate column a is reduced with the k reflectors of the accepted columns (w = Q^T a), which
// yields the new column of R (w[0..k)) and its diagonal entry (+-|w[k..m)|); R^-1 grows by one column in O(k^2), so the EXACT
// kappa_1(R') = |R'|_1 |R'^-1|_1 of the decision `mat.Cond(columns, 1) <= 1e12` (simplex.go:630; R of a full-rank matrix is unique
// up to row signs, which no norm sees) costs O(m k + k^2) per candidate, O(m^2 n) in all.  The last column makes the matrix
// square, where the reference measures kappa_1 of the matrix itself through its LU: |A|_1 |R^-1 Q^T|_1 here.
// dot product with eight independent partial sums: a single chain of dependent additions (the compiler may not reassociate)
// made the search latency-bound — 125 ms for 600 rows, 580 ms for 1000; these are the engine's own quantities (the decision
// is a threshold on a condition number), not values the reference defines bit by bit
static bool invert_threaded(const std::vector<double> &A, int n, std::vector<double> &inv);

static inline double dot8(const double *x, const double *y, int n) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        s0 += x[i] * y[i]; s1 += x[i + 1] * y[i + 1]; s2 += x[i + 2] * y[i + 2]; s3 += x[i + 3] * y[i + 3];
        s4 += x[i + 4] * y[i + 4]; s5 += x[i + 5] * y[i + 5]; s6 += x[i + 6] * y[i + 6]; s7 += x[i + 7] * y[i + 7];
    }
    for (; i < n; i++) s0 += x[i] * y[i];
    return ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}

int general_find_linearly_independent(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs, std::vector<double> *binv_out) {
    idxs.clear();
    if (binv_out) binv_out->clear();
    // reflector k in ROW k of V (entries k..m-1; contiguous: unit-stride dot / axpy), H = I - 2 v v^T / (v^T v)
    std::vector<double> V((size_t)m * m, 0.0);
    std::vector<double> vnorm2(m, 0.0);
    std::vector<double> R((size_t)m * m, 0.0), Rinv((size_t)m * m, 0.0);   // upper triangular, row-major m x m (leading k x k used)
    std::vector<double> colsum_R(m, 0.0), colsum_Rinv(m, 0.0);              // 1-norm bookkeeping: absolute column sums
    std::vector<double> w(m), t(m);
    std::vector<double> Acols((size_t)m * m, 0.0);   // the accepted columns (for the square-case norm)
    double nR = 0, nRinv = 0;
    for (int i = n - 1; i >= 0; i--) {
        const int k = (int)idxs.size();
        if (k == m) break;
        for (int r = 0; r < m; r++) w[r] = A[(size_t)r * n + i];
        if (k == 0) {   // simplex.go:624-629: the first column is always accepted
            double nrm = 0;
            for (int r = 0; r < m; r++) nrm = hypot(nrm, w[r]);
            const double alpha = w[0], beta = alpha >= 0 ? -nrm : nrm;
            for (int r = 0; r < m; r++) V[r] = w[r];
            V[0] = alpha - beta;
            vnorm2[0] = dot8(V.data(), V.data(), m);
            R[0] = beta; Rinv[0] = beta != 0 ? 1 / beta : std::numeric_limits<double>::infinity();
            colsum_R[0] = fabs(beta); colsum_Rinv[0] = fabs(Rinv[0]);
            nR = colsum_R[0]; nRinv = colsum_Rinv[0];
            for (int r = 0; r < m; r++) Acols[(size_t)r * m + 0] = A[(size_t)r * n + i];
            idxs.push_back(i);
            continue;
        }
        // w = H_{k-1} ... H_0 a
        for (int j = 0; j < k; j++) {
            if (vnorm2[j] == 0) continue;
            const double *vj = V.data() + (size_t)j * m;
            const double f = 2 * dot8(vj + j, w.data() + j, m - j) / vnorm2[j];
            for (int r = j; r < m; r++) w[r] -= f * vj[r];
        }
        double nrm = 0;
        for (int r = k; r < m; r++) nrm = hypot(nrm, w[r]);
        const double alpha = w[k], beta = alpha >= 0 ? -nrm : nrm;   // diagonal entry of the new R column
        // candidate norms: |R'|_1, |R'^-1|_1 with R'^-1 = [[Rinv, -Rinv w_top / beta], [0, 1 / beta]]
        double cs = fabs(beta);
        for (int r = 0; r < k; r++) cs += fabs(w[r]);
        double csi = beta != 0 ? fabs(1 / beta) : std::numeric_limits<double>::infinity();
        for (int r = 0; r < k; r++) {
            const double acc = dot8(Rinv.data() + (size_t)r * m + r, w.data() + r, k - r);
            t[r] = beta != 0 ? -acc / beta : std::numeric_limits<double>::infinity();
            csi += fabs(t[r]);
        }
        double cond;
        if (beta == 0 || !std::isfinite(nRinv) || !std::isfinite(csi)) {
            cond = std::numeric_limits<double>::infinity();   // a zero on the diagonal of R: exactly rank deficient (Dtrcon / Dgecon report rcond = 0)
        } else if (k + 1 < m) {
            cond = std::max(nR, cs) * std::max(nRinv, csi);
        } else {
            // square: kappa_1 of the matrix itself, |A|_1 |A^-1|_1 with A^-1 = R'^-1 Q^T
            for (int r = 0; r < m; r++) Acols[(size_t)r * m + k] = A[(size_t)r * n + i];
            if (!(nrm > 0) && !(fabs(alpha) > 0)) cond = std::numeric_limits<double>::infinity();
            else {
                // (threaded LU above 256 rows: this one inversion used to cost more than the whole incremental search; the caller
                // reuses it as B^-1 of the starting basis — the columns of C are the basis in position order)
                std::vector<double> inv;
                cond = invert_threaded(Acols, m, inv) ? norm1(Acols, m, m, m) * norm1(inv, m, m, m) : std::numeric_limits<double>::infinity();
                if (!(cond > 1e12) && binv_out) binv_out->swap(inv);
            }
        }
        if (cond > 1e12) continue;   // :630 not linearly independent (a NaN passes, as in the reference)
        // accept: reflector k, new columns of R and R^-1
        for (int r = 0; r < k; r++) { R[(size_t)r * m + k] = w[r]; Rinv[(size_t)r * m + k] = t[r]; }
        R[(size_t)k * m + k] = beta; Rinv[(size_t)k * m + k] = 1 / beta;
        colsum_R[k] = cs; colsum_Rinv[k] = csi;
        nR = std::max(nR, cs); nRinv = std::max(nRinv, csi);
        double *vk = V.data() + (size_t)k * m;
        for (int r = k; r < m; r++) vk[r] = (r == k) ? alpha - beta : w[r];
        vnorm2[k] = dot8(vk + k, vk + k, m - k);
        if (k + 1 < m) for (int r = 0; r < m; r++) Acols[(size_t)r * m + k] = A[(size_t)r * n + i];
        idxs.push_back(i);
    }
    return (int)idxs.size() == m ? GOMILP_OK : GOMILP_ERR_SINGULAR;  // :495-497
}

// The last step of the search when the device form (general_kernels.hip) has accepted m - 1 columns: the candidate makes the
// matrix square, where the reference measures kappa_1 of the matrix itself through its LU (mat.Cond of a square matrix) —
// |C|_1 |C^-1|_1 with the inversion the caller needs as B^-1 anyway.  Candidates from `start_col` downwards.
int general_finish_last_column(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs, int start_col, std::vector<double> *binv_out) {
    if ((int)idxs.size() != m - 1) return GOMILP_ERR_SINGULAR;
    std::vector<double> C((size_t)m * m);
    for (int k = 0; k < m - 1; k++)
        for (int r = 0; r < m; r++) C[(size_t)r * m + k] = A[(size_t)r * n + idxs[k]];
    for (int i = start_col; i >= 0; i--) {
        for (int r = 0; r < m; r++) C[(size_t)r * m + (m - 1)] = A[(size_t)r * n + i];
        std::vector<double> inv;
        const double cond = invert_threaded(C, m, inv) ? norm1(C, m, m, m) * norm1(inv, m, m, m) : std::numeric_limits<double>::infinity();
        if (cond > 1e12) continue;   // simplex.go:630
        if (binv_out) binv_out->swap(inv);
        idxs.push_back(i);
        return GOMILP_OK;
    }
    return GOMILP_ERR_SINGULAR;   // :495-497
}

// reference form of the same decisions: a fresh exact condition number per candidate (O(m^4); kept for the tests that pin
// the incremental version against it)
int general_find_linearly_independent_slow(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs) {
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

// B^-1 for large bases: LU with partial pivoting (rows of the trailing update split over host threads), then the m unit
// right-hand sides solved independently, a slice per thread.  (Gauss-Jordan above is O(2 m^3) on one core: 20 s at m = 2048.)
static bool invert_threaded(const std::vector<double> &A, int n, std::vector<double> &inv) {
    const int nt = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
    if (n < 256 || nt == 1) return invert(A, n, inv);
    std::vector<double> LU(A);
    std::vector<int> piv(n);
    std::atomic<int> arrived(0), gen(0);
    std::atomic<bool> ok(true);
    auto barrier = [&](int &my_gen) {   // sense-reversing spin barrier of the team
        my_gen++;
        if (arrived.fetch_add(1) + 1 == nt) { arrived.store(0); gen.store(my_gen); }
        else while (gen.load() < my_gen) std::this_thread::yield();
    };
    auto work = [&](int tid) {
        int my_gen = 0;
        for (int k = 0; k < n; k++) {
            if (tid == 0) {
                int p = k;
                double best = fabs(LU[(size_t)k * n + k]);
                for (int i = k + 1; i < n; i++) { const double v = fabs(LU[(size_t)i * n + k]); if (v > best) { best = v; p = i; } }
                if (!(best > 0) || !std::isfinite(best)) ok.store(false);
                piv[k] = p;
                if (p != k) for (int j = 0; j < n; j++) std::swap(LU[(size_t)k * n + j], LU[(size_t)p * n + j]);
            }
            barrier(my_gen);
            if (!ok.load()) return;
            const double d = LU[(size_t)k * n + k];
            const double *rk = &LU[(size_t)k * n];
            for (int i = k + 1 + tid; i < n; i += nt) {
                double *ri = &LU[(size_t)i * n];
                const double f = ri[k] / d;
                ri[k] = f;
                if (f != 0) for (int j = k + 1; j < n; j++) ri[j] -= f * rk[j];
            }
            barrier(my_gen);
        }
        // columns of the inverse: P A = L U  ->  A^-1 e_c = U^-1 L^-1 P e_c
        std::vector<double> x(n);
        for (int c = tid; c < n; c += nt) {
            for (int i = 0; i < n; i++) x[i] = 0;
            x[c] = 1;
            for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(x[k], x[piv[k]]);
            for (int i = 0; i < n; i++) { const double *ri = &LU[(size_t)i * n]; double s2 = x[i]; for (int j = 0; j < i; j++) s2 -= ri[j] * x[j]; x[i] = s2; }
            for (int i = n - 1; i >= 0; i--) { const double *ri = &LU[(size_t)i * n]; double s2 = x[i]; for (int j = i + 1; j < n; j++) s2 -= ri[j] * x[j]; x[i] = s2 / ri[i]; }
            for (int i = 0; i < n; i++) inv[(size_t)i * n + c] = x[i];
        }
    };
    inv.assign((size_t)n * n, 0.0);
    std::vector<std::thread> th;
    for (int t2 = 1; t2 < nt; t2++) th.emplace_back(work, t2);
    work(0);
    for (auto &t2 : th) t2.join();
    return ok.load();
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
    return invert_threaded(B, m, binv);
}

bool general_invert(const std::vector<double> &B, int m, std::vector<double> &inv) { return invert_threaded(B, m, inv); }

// exact kappa_1 of a basis (columns `basic` of A, the artificial column `art` standing for index n): the trial bases of the Bland
// rule (simplex.go:374-379: mat.Cond(abTmp, 1) < 1e16; the reference has Dgecon's estimate of the same number)
double general_basis_cond1(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, const std::vector<double> &art) {
    std::vector<double> B((size_t)m * m);
    for (int i = 0; i < m; i++)
        for (int p = 0; p < m; p++) {
            const int j = basic[p];
            B[(size_t)i * m + p] = (j < n) ? A[(size_t)i * n + j] : art[i];
        }
    return cond1_exact(B, m, m);
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
        if (m <= kGonumCondMax) {
            // the reference's own quantities: the cond mat.LU estimated when it factorized ab^T (duals) resp. ab (x_B, computeMove), and its
            // Det() == 0 test (gonum_cond.cpp) — an ESTIMATE on rounded factors, which near 1e16 can fall on either side of the exact value
            bool dz = false;
            gonum_lu_cond(B.data(), m, m, true, &k1, &dz);
            if (dz) k1 = std::numeric_limits<double>::infinity();
            gonum_lu_cond(B.data(), m, m, false, &kinf, &dz);
            if (dz) kinf = std::numeric_limits<double>::infinity();
            return;
        }
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

// Hager / Higham estimate of the 1-norm of M (transposed = false) or of M^T (true) for an explicitly given n x n matrix M — the
// iteration LAPACK's dlacn2 drives (published algorithm: Higham, "FORTRAN codes for estimating the one-norm of a real or complex
// matrix", ACM TOMS 14, 1988; gonum: lapack/gonum/dlacn2.go): start from the uniform vector, follow the sign vector's gradient
// through at most five products with M and M^T, then compare with the alternating-sign probe.  Engine::cond_check hands it
// B^-1, so the value is what gonum's Dgecon reports for the basis up to the rounding of the products.
double inverse_norm1_estimate(const std::vector<double> &M, int n, bool transposed) {
    if (n <= 0) return 0.0;
    std::vector<double> x(n), y(n), z(n);
    std::vector<int> sgn(n);
    auto mul = [&](const std::vector<double> &in, std::vector<double> &out, bool tr) {   // out = M in (tr: M^T in), `transposed` swaps the two
        const bool t = tr != transposed;
        std::fill(out.begin(), out.end(), 0.0);
        if (!t) { for (int i = 0; i < n; i++) { double s2 = 0; const double *row = &M[(size_t)i * n]; for (int j = 0; j < n; j++) s2 += row[j] * in[j]; out[i] = s2; } }
        else { for (int i = 0; i < n; i++) { const double xi = in[i]; if (xi == 0) continue; const double *row = &M[(size_t)i * n]; for (int j = 0; j < n; j++) out[j] += row[j] * xi; } }
    };
    auto asum = [&](const std::vector<double> &v) { double s2 = 0; for (double e : v) s2 += fabs(e); return s2; };
    auto amax = [&](const std::vector<double> &v) { int best = 0; for (int i = 1; i < n; i++) if (fabs(v[i]) > fabs(v[best])) best = i; return best; };
    for (int i = 0; i < n; i++) x[i] = 1.0 / n;
    mul(x, y, false);
    if (n == 1) return fabs(y[0]);
    double est = asum(y);
    for (int i = 0; i < n; i++) { sgn[i] = std::signbit(y[i]) ? -1 : 1; x[i] = sgn[i]; }
    mul(x, z, true);
    int j = amax(z);
    for (int iter = 2;; iter++) {
        std::fill(x.begin(), x.end(), 0.0);
        x[j] = 1.0;
        mul(x, y, false);
        const double estold = est;
        est = asum(y);
        bool same = true;
        for (int i = 0; i < n; i++) if ((std::signbit(y[i]) ? -1 : 1) != sgn[i]) { same = false; break; }
        if (same || est <= estold) break;
        for (int i = 0; i < n; i++) { sgn[i] = std::signbit(y[i]) ? -1 : 1; x[i] = sgn[i]; }
        mul(x, z, true);
        const int jlast = j;
        j = amax(z);
        if (!(z[jlast] != fabs(z[j]) && iter < 5)) break;
    }
    double alt = 1.0;
    for (int i = 0; i < n; i++) { x[i] = alt * (1.0 + (double)i / (double)(n - 1)); alt = -alt; }
    mul(x, y, false);
    const double probe = 2.0 * asum(y) / (double)(3 * n);
    return probe > est ? probe : est;
}

// exact kappa_inf of a square matrix (m == n path: any Condition of the single solve becomes lp.ErrSingular, simplex.go:109-112)
double general_cond_inf(const std::vector<double> &A, int n) {
    if (n <= kGonumCondMax) {   // the reference's estimate itself (gonum_cond.cpp)
        double c = 0;
        bool dz = false;
        gonum_lu_cond(A.data(), n, n, false, &c, &dz);
        return dz ? std::numeric_limits<double>::infinity() : c;
    }
    std::vector<double> inv;
    if (!invert(A, n, inv)) return std::numeric_limits<double>::infinity();
    return norm_inf(A, n) * norm_inf(inv, n);
}

// plain LU solve B x = b for the point the reference returns with a mid-loop error (small bases)
bool general_solve_basis(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, const std::vector<double> &b, std::vector<double> &x) {
    std::vector<double> B, inv;
    basis_matrix(A, m, n, basic, B);
    if (m <= kGonumCondMax) {   // LU.SolveVec's own result (gonum_cond.cpp): the point the reference returns with a mat.Condition error, bit for bit
        x.assign(m, 0.0);
        return gonum_lu_solve(B.data(), m, m, b.data(), x.data());
    }
    if (!invert(B, m, inv)) return false;
    x.assign(m, 0.0);
    for (int i = 0; i < m; i++) { double s2 = 0; for (int j = 0; j < m; j++) s2 += inv[(size_t)i * m + j] * b[j]; x[i] = s2; }
    return true;
}

}  // namespace gomilp
