// zf_decide.h - the line-search / termination decision, shared by the decide
// kernel (device) and zf_decide_host() (host, for GPU-less tests).
//
// Follows zfista/proximal_gradient.py term by term:
//   model value   :149-155   fun = <grad,dx> + g(x+) + |dx|^2/2/lr (+ f(y) - F_old)
//   F(x+)         :295
//   acceptance    :298-305   decay_rate == 1 -> accept; deprecated test; default test
//   failure       :306-307   max_backtrack_iter trials without acceptance
//   termination   :510,:525  err = max|x+ - y| < tol (strict), tested before momentum
//   max_iter      :539-543
// The arithmetic is plain IEEE double with contraction disabled (the library
// is built with -ffp-contract=off) so the branch decisions are the ones NumPy
// scalars would take on the same reduced sums.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/zfista_hip.h"

#if defined(__HIPCC__)
#define ZF_HD __host__ __device__
#else
#define ZF_HD
#endif

// pack layout (per rank): sums are added over ranks in rank order, the max is
// maxed.  [0] f(y)  [1] <grad f(y), x+ - y>  [2] |x+ - y|^2  [3] g(x+)
//         [4] f(x+) [5] max|x+ - y|          [6],[7] spare
enum { ZF_PK_FY = 0, ZF_PK_DOT = 1, ZF_PK_SS = 2, ZF_PK_GX = 3, ZF_PK_FX = 4, ZF_PK_ERR = 5 };
// trace row: [0] err [1] F(x+) [2] lr [3] model value [4] trials [5] f(x+) [6] g(x+) [7] f(y)
enum { ZF_TR_ERR = 0, ZF_TR_F = 1, ZF_TR_LR = 2, ZF_TR_FUN = 3, ZF_TR_TRIALS = 4,
       ZF_TR_FX = 5, ZF_TR_GX = 6, ZF_TR_FY = 7 };

ZF_HD inline void zf_reduce_packs(const double* packs, int world, double* out) {
    for (int k = 0; k < ZF_PACK_LEN; ++k) out[k] = packs[k];
    for (int r = 1; r < world; ++r) {
        const double* p = packs + (int64_t)r * ZF_PACK_LEN;
        for (int k = 0; k < ZF_PACK_LEN; ++k) {
            if (k == ZF_PK_ERR) out[k] = (p[k] > out[k] || p[k] != p[k]) ? p[k] : out[k];
            else out[k] = out[k] + p[k];
        }
    }
}

// beta_ring (may be NULL on the host): momentum ring indexed by the accepted-iteration count;
// on acceptance the factor of the NEXT trial is copied into the control block.
ZF_HD inline void zf_decide_step(zf_control* c, const double* packs, double* trace,
                                 const double* beta_ring = nullptr) {
    if (c->status != ZF_RUNNING) return;
    double pk[ZF_PACK_LEN];
    zf_reduce_packs(packs, c->world, pk);
    const double f_y = pk[ZF_PK_FY], dot = pk[ZF_PK_DOT], ss = pk[ZF_PK_SS];
    const double g_x = pk[ZF_PK_GX], f_x = pk[ZF_PK_FX], err = pk[ZF_PK_ERR];
    const double lr = c->lr;
    const double F_old = c->F_old;

    const double nrm = sqrt(ss);                       // np.linalg.norm(x+ - y)
    double fun = (dot + g_x) + nrm * nrm / 2 / lr;     // :150-152
    if (!c->deprecated) fun = fun + (f_y - F_old);     // :155
    const double F_x = f_x + g_x;                      // :295

    bool accept;
    if (c->decay_rate == 1.0) accept = true;                                  // :298
    else if (c->deprecated) accept = (f_x - f_y <= fun + c->tol_internal);    // :301
    else accept = (F_x - F_old <= fun + c->tol_internal);                     // :303

    c->trial += 1;
    c->total_trials += 1;
    c->f_y = f_y;
    if (!accept) {
        c->lr = lr * c->decay_rate;                                           // :305
        c->need_grad = 0;   // y_k unchanged: grad f(y_k), f(y_k) stay valid
        if (c->trial >= c->max_backtrack) c->status = ZF_BACKTRACK_FAILED;    // :306-307
        return;
    }
    const int64_t nit = c->nit + 1;
    double* row = trace + ((nit - 1) % ZF_RING) * ZF_TRACE_COLS;
    row[ZF_TR_ERR] = err;
    row[ZF_TR_F] = F_x;
    row[ZF_TR_LR] = lr;
    row[ZF_TR_FUN] = fun;
    row[ZF_TR_TRIALS] = (double)c->trial;
    row[ZF_TR_FX] = f_x;
    row[ZF_TR_GX] = g_x;
    row[ZF_TR_FY] = f_y;
    c->nit = nit;
    c->F_old = F_x;
    c->f_x = f_x;
    c->g_x = g_x;
    c->err = err;
    c->fun = fun;
    c->trial = 0;
    c->cur = (c->cur + 1) % 3;   // x+ becomes x_k; old x_k becomes x_{k-1} (:538)
    c->need_grad = 1;
    if (beta_ring) c->beta_next = beta_ring[nit % ZF_RING];   // y_{k+1} = x_k + beta (x_k - x_{k-1})  :533-534
    if (err < c->tol) c->status = ZF_CONVERGED;               // :525
    else if (nit >= c->max_iter) c->status = ZF_MAXITER;      // :539
}
