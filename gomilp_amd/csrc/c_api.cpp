// extern "C" surface declared in include/gomilp_lp.h.  Each entry point names the reference
// interface it replaces; see INTEGRATION.md for the cgo binding.
#include <math.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "engine.hpp"

using gomilp::Engine;

struct gomilp_ctx {
    std::unique_ptr<Engine> eng;
};

extern "C" {

const char *gomilp_version(void) { return "gomilp_amd 0.1 (gfx950)"; }
int gomilp_device_count(void) { return gomilp::device_count(); }
const char *gomilp_compiled_arch(void) { return gomilp::compiled_arch(); }

gomilp_ctx *gomilp_ctx_create(int device, int *status) {
    int n = gomilp::device_count();
    if (n <= 0) { if (status) *status = GOMILP_ERR_DEVICE; return nullptr; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) { if (status) *status = GOMILP_ERR_DEVICE; return nullptr; }
    gomilp_ctx *c = new gomilp_ctx;
    c->eng.reset(new Engine(device));
    if (status) *status = GOMILP_OK;
    return c;
}
void gomilp_ctx_destroy(gomilp_ctx *ctx) { delete ctx; }
int gomilp_ctx_device(const gomilp_ctx *ctx) { return ctx ? ctx->eng->device() : -1; }
int gomilp_ctx_set(gomilp_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->set(key, value);
}
int64_t gomilp_lp_upload(gomilp_ctx *ctx, const double *c, const double *A, int64_t lda, const double *b, int64_t m,
                         int64_t n) {
    if (!ctx) return -GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->upload(c, A, lda, b, m, n);
}
int gomilp_lp_free(gomilp_ctx *ctx, int64_t problem) {
    if (!ctx) return GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->free_problem(problem);
}
int gomilp_lp_solve_resident(gomilp_ctx *ctx, int64_t problem, double tol, const int64_t *initial_basic, double *opt_f,
                             double *opt_x, int32_t *has_x, int64_t *basis_out, gomilp_lp_stats *stats) {
    if (!ctx) return GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->solve(problem, tol, initial_basic, opt_f, opt_x, has_x, basis_out, stats);
}
int64_t gomilp_lp_last_trace(gomilp_ctx *ctx, gomilp_pivot *out, int64_t cap) {
    if (!ctx) return -1;
    return ctx->eng->last_trace(out, cap);
}

// lp.Simplex drop-in (simplex.go:88): one shared context per device, created on first use; the call
// uploads, solves and frees, so nothing of the caller's memory is retained (cgo rules).
int gomilp_lp_simplex(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n, double tol,
                      const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x, int64_t *basis_out,
                      gomilp_lp_stats *stats) {
    static std::mutex mu;
    static std::map<int, gomilp_ctx *> ctxs;
    if (has_x) *has_x = 0;
    if (opt_f) *opt_f = NAN;
    if (!c || !A || !b || !opt_f || !opt_x || !has_x || m <= 0 || n <= 0 || lda < n) return GOMILP_ERR_BAD_SHAPE;
    int dev = 0;
    if (gomilp::device_count() <= 0 || hipGetDevice(&dev) != hipSuccess) return GOMILP_ERR_DEVICE;
    gomilp_ctx *ctx;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ctxs.find(dev);
        if (it == ctxs.end()) {
            int st = 0;
            ctx = gomilp_ctx_create(dev, &st);
            if (!ctx) return st;
            ctxs[dev] = ctx;
        } else {
            ctx = it->second;
        }
    }
    int64_t id = gomilp_lp_upload(ctx, c, A, lda, b, m, n);
    if (id < 0) return (int)-id;
    int rc = gomilp_lp_solve_resident(ctx, id, tol, initial_basic, opt_f, opt_x, has_x, basis_out, stats);
    gomilp_lp_free(ctx, id);
    return rc;
}

}  // extern "C"
