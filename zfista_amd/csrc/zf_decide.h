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
//
// Temporal blocking (separable f): one pass of the trial kernel chains up to S FRESH trials in
// registers, each assuming the previous one was accepted, and stores only the last two iterates
// of the chain.  zf_decide_pass() examines the fresh packs in order.  If the whole chain holds,
// the stored pair becomes (x_k, x_{k-1}).  If the chain breaks after `a` acceptances (a
// rejection, or a termination before the last fresh trial), the a iterations are accepted all
// the same - counters, F_old, lr, trace rows advance exactly as in the one-trial-per-pass loop -
// but their iterates were not stored: ctl->lag += a and ctl->lag_lr[] records the step size of
// each.  The next pass REPLAYS the lagging iterations element-wise (no reductions, their
// decisions are known; 11 flops per element instead of 20) and chains its fresh trials behind
// them, so a broken chain costs some arithmetic in the next pass, not a pass over the data.
// A final status reached while iterates lag is held back in ctl->pend_status for one
// replay-only pass that materialises x_k.  lag + fresh <= 2 S - 1 bounds every chain.
// Every decision is taken on exactly the sums a one-trial pass would have produced: results do
// not depend on S.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zfista_hip.h"

#if defined(__HIPCC__)
#define ZF_HD __host__ __device__
#else
#define ZF_HD
#endif

// pack layout (per rank): sums are added over ranks in rank order, the max is
// maxed.  [0] f(y)  [1] <grad f(y), x+ - y>  [2] |x+ - y|^2  [3] g(x+)
//         [4] f(x+) [5] max|x+ - y|          [6] stamp of the block the pass read (sharded solves; zf_pack_stamp)
//         [7] f(x+) - f(y), accumulated element by element (ZF_ACCEPT_RESOLVED only, else 0)
enum { ZF_PK_FY = 0, ZF_PK_DOT = 1, ZF_PK_SS = 2, ZF_PK_GX = 3, ZF_PK_FX = 4, ZF_PK_ERR = 5, ZF_PK_DF = 7 };
// trace row: [0] err [1] F(x+) [2] lr [3] model value [4] trials [5] f(x+) [6] g(x+) [7] f(y)
enum { ZF_TR_ERR = 0, ZF_TR_F = 1, ZF_TR_LR = 2, ZF_TR_FUN = 3, ZF_TR_TRIALS = 4,
       ZF_TR_FX = 5, ZF_TR_GX = 6, ZF_TR_FY = 7 };

// `stride` = doubles between the packs of consecutive ranks (ZF_PACK_LEN x trials per pass)
ZF_HD inline void zf_reduce_packs(const double* packs, int world, int stride, double* out) {
    for (int k = 0; k < ZF_PACK_LEN; ++k) out[k] = packs[k];
    for (int r = 1; r < world; ++r) {
        const double* p = packs + (int64_t)r * stride;
        for (int k = 0; k < ZF_PACK_LEN; ++k) {
            if (k == ZF_PK_ERR) out[k] = (p[k] > out[k] || p[k] != p[k]) ? p[k] : out[k];
            else out[k] = out[k] + p[k];
        }
    }
}

// The x ring: x_k in buffer `cur`, x_{k-1} in `prev`; a pass writes into the lowest-numbered
// buffers that hold neither.  One trial: x+ -> *first.  A chain of n >= 2 trials:
// x_{k+n-1} -> *first, x_{k+n} -> *second (ring_size 4).
ZF_HD inline void zf_free_bufs(int cur, int prev, int ring, int* first, int* second) {
    int f[2] = {-1, -1}, m = 0;
    if (ring >= 5) {
        // run-ahead passes (6 buffers): the buffers BEHIND x_k in ring order, so that the pass after a pass never
        // writes what that pass reads - it may start before that pass has been decided, and a chain that breaks
        // is replayed from its inputs (cur = 0, prev = 5: -> 1, 2;  then cur = 2, prev = 1: -> 3, 4;  then -> 5, 0)
        for (int k = 1; k < ring && m < 2; ++k) {
            const int i = (cur + k) % ring;
            if (i != prev) f[m++] = i;
        }
    } else {
        for (int i = 0; i < ring && m < 2; ++i)
            if (i != cur && i != prev) f[m++] = i;
    }
    *first = f[0];
    *second = (m > 1) ? f[1] : f[0];
}

// FRESH trials of the next pass: S when a full chain fits and leaves either nothing or at least another
// full chain's worth behind it; the last S < left < 2 S iterations before max_iter (:539) are SHARED by
// two passes of about left / 2 (a pass of 10 + one of 10 costs 2 x 0.85 ms at n = 1e8, a pass of 16 + one of
// 4 costs 1.2 + 0.85: a short chain is bound by its 48 B per element, not by its trials); after a broken
// chain up to S (S / 2 for S = 16) behind the lagging iterations, bounded by the chain capacity 2 S - 1; 0
// for a materialise-only pass.  (Which trials share a pass never changes a result: every decision is taken on
// the sums of its own trial.)
ZF_HD inline int zf_fresh_len(const zf_control* c) {
    if (c->pend_status != 0) return 0;
    const int sub = c->sub_iters > 0 ? c->sub_iters : 1;
    const int64_t left = c->max_iter - c->nit;
    int64_t n = sub >= 16 ? sub / 2 : sub;   // behind lagging iterations: a short chain (the pass already carries the replays)
    if (c->lag == 0) {
        if (left >= 2 * (int64_t)sub || left == sub) return sub;
        n = left > sub ? (left + 1) / 2 : left;
    }
    const int64_t cap = (int64_t)(2 * sub - 1) - c->lag;
    if (n > cap) n = cap;
    if (n > left) n = left;
    if (n < 1) n = 1;
    return (int)n;
}

// Stamp of the control-block state a pass was computed from, carried in spare slot 6 of its packs (never 0, the
// value every other producer of packs writes there).  Every decide step changes it: fresh trials raise
// total_trials, a materialise-only step clears lag and pend_status.  The control block is identical on all ranks,
// so the decide step of a sharded solve can tell packs of THIS step from packs an earlier step left behind when the
// one kernel launched for the predicted shape of the pass was not the one the pass needed (zf_predict_parts).
ZF_HD inline double zf_pack_stamp(const zf_control* c) {
    return (double)(((c->total_trials + 1) << 6) | ((int64_t)(c->lag & 31) << 1) | (c->pend_status != 0 ? 1 : 0));
}

// The floating-point part of one trial: model value, F(x+) and the acceptance test for a given
// F_old and step size (:149-155, :295, :298-303).  Pure: trials of a chain can be evaluated
// independently (lane-parallel on the device) once F_old of trial j is taken as F(x+) of trial
// j - 1, which is what it is if the chain holds up to there.
typedef struct zf_trial_eval {
    double fun, F_x, f_x, g_x, f_y, err;
    double lr_used, F_old_used;
    int accept;
} zf_trial_eval;

ZF_HD inline void zf_eval_trial(const zf_control* c, double F_old, double lr, const double* pk /* reduced */,
                                zf_trial_eval* e) {
    const double f_y = pk[ZF_PK_FY], dot = pk[ZF_PK_DOT], ss = pk[ZF_PK_SS];
    const double g_x = pk[ZF_PK_GX], f_x = pk[ZF_PK_FX];
    const double nrm = sqrt(ss);                       // np.linalg.norm(x+ - y)
    double fun = (dot + g_x) + nrm * nrm / 2 / lr;     // :150-152
    if (!c->deprecated) fun = fun + (f_y - F_old);     // :155
    const double F_x = f_x + g_x;                      // :295
    bool accept;
    if (c->decay_rate == 1.0) accept = true;                                  // :298
    else if (c->accept_mode == ZF_ACCEPT_RESOLVED) {
        // The same inequalities with f(x+) - f(y) taken from the element-wise accumulation (pack slot 7) instead of the
        // difference of two O(|F|) sums.  :303 reads, F_old and g(x+) cancelled (they do, exactly, in exact arithmetic):
        //   [f(x+) - f(y)] - <grad f(y), x+ - y> - |x+ - y|^2 / 2 / lr <= tol_internal
        // - every term of the size of the step.  :301 (deprecated) keeps g(x+) on its right-hand side: f(x+) - f(y) <= fun.
        const double df = pk[ZF_PK_DF];
        if (c->deprecated) accept = (df <= fun + c->tol_internal);
        else accept = ((df - dot) - nrm * nrm / 2 / lr <= c->tol_internal);
    }
    else if (c->deprecated) accept = (f_x - f_y <= fun + c->tol_internal);    // :301
    else accept = (F_x - F_old <= fun + c->tol_internal);                     // :303
    e->fun = fun;
    e->F_x = F_x;
    e->f_x = f_x;
    e->g_x = g_x;
    e->f_y = f_y;
    e->err = pk[ZF_PK_ERR];
    e->lr_used = lr;
    e->F_old_used = F_old;
    e->accept = accept ? 1 : 0;
}

// The bookkeeping part of one trial (:305-307, :510, :525, :533-543) for an evaluation made with
// the block's current lr and F_old.  Returns true when the trial was accepted and the loop goes on.
ZF_HD inline bool zf_apply_trial(zf_control* c, const zf_trial_eval* e, double* trace, const double* beta_ring) {
    const double lr = c->lr;
    c->trial += 1;
    c->total_trials += 1;
    c->f_y = e->f_y;
    if (!e->accept) {
        c->lr = lr * c->decay_rate;                                           // :305
        c->need_grad = 0;   // y_k unchanged: grad f(y_k), f(y_k) stay valid
        if (c->trial >= c->max_backtrack) c->status = ZF_BACKTRACK_FAILED;    // :306-307
        return false;
    }
    const int64_t nit = c->nit + 1;
    double* row = trace + ((nit - 1) % ZF_RING) * ZF_TRACE_COLS;
    row[ZF_TR_ERR] = e->err;
    row[ZF_TR_F] = e->F_x;
    row[ZF_TR_LR] = lr;
    row[ZF_TR_FUN] = e->fun;
    row[ZF_TR_TRIALS] = (double)c->trial;
    row[ZF_TR_FX] = e->f_x;
    row[ZF_TR_GX] = e->g_x;
    row[ZF_TR_FY] = e->f_y;
    c->nit = nit;
    c->F_old = e->F_x;
    c->f_x = e->f_x;
    c->g_x = e->g_x;
    c->err = e->err;
    c->fun = e->fun;
    c->trial = 0;
    c->need_grad = 1;
    // (beta of the next trial: resolved once per pass by the caller - zf_resolve_beta - because each
    //  load of the momentum ring is a dependent global-memory round trip)
    (void)beta_ring;
    if (e->err < c->tol) c->status = ZF_CONVERGED;            // :525
    else if (nit >= c->max_iter) c->status = ZF_MAXITER;      // :539
    return c->status == ZF_RUNNING;
}

// y_{k+1} = x_k + beta (x_k - x_{k-1})  (:533-534): the factor of the trial that follows the
// accepted-iteration count now in the block
// (with lagging iterations the first trial of the next pass is the replay of iteration nit - lag + 1)
ZF_HD inline void zf_resolve_beta(zf_control* c, const double* beta_ring) {
    if (beta_ring) c->beta_next = beta_ring[(c->nit - c->lag) % ZF_RING];
}

// One trial against the control block *c (buffer indices are not touched: the caller commits
// them).  beta_ring (may be NULL on the host): momentum ring indexed by the accepted-iteration
// count; on acceptance the factor of the NEXT trial is copied into the control block.
// `pre` (optional): this trial evaluated in advance; used when it was made with the block's
// current lr and F_old, recomputed otherwise.
ZF_HD inline bool zf_decide_step(zf_control* c, const double* packs, double* trace,
                                 const double* beta_ring = nullptr, int pack_stride = ZF_PACK_LEN,
                                 const zf_trial_eval* pre = nullptr) {
    if (c->status != ZF_RUNNING) return false;
    if (pre && pre->lr_used == c->lr && pre->F_old_used == c->F_old) return zf_apply_trial(c, pre, trace, beta_ring);
    double pk[ZF_PACK_LEN];
    zf_reduce_packs(packs, c->world, pack_stride, pk);
    zf_trial_eval e;
    zf_eval_trial(c, c->F_old, c->lr, pk, &e);
    return zf_apply_trial(c, &e, trace, beta_ring);
}

// single-trial form with the momentum factor resolved (host tests, experiments)
ZF_HD inline bool zf_decide_one(zf_control* c, const double* packs, double* trace, const double* beta_ring = nullptr) {
    const int64_t before = c->nit;
    const bool go = zf_decide_step(c, packs, trace, beta_ring);
    if (c->nit > before) zf_resolve_beta(c, beta_ring);
    return go;
}

// write back a control block examined on a copy: everything but lag_lr[], whose entries are
// written in place (indexing the copy dynamically would push the whole block to scratch memory)
ZF_HD inline void zf_store_head(zf_control* dst, const zf_control* src) {
    __builtin_memcpy(dst, src, offsetof(zf_control, lag_lr));
}

// the stored pair of a chain of `ntr` trials run from (cur0, prev0) becomes (x_k, x_{k-1}) (:538)
ZF_HD inline void zf_commit_chain(zf_control* c, int cur0, int prev0, int ring, int ntr) {
    int first, second;
    zf_free_bufs(cur0, prev0, ring, &first, &second);
    if (ntr == 1) {
        c->prev = cur0;
        c->cur = first;
    } else {
        c->prev = first;
        c->cur = second;
    }
}

// All trials of one pass.  packs: world x sub_iters x ZF_PACK_LEN (rank-major), pack j of a
// rank = FRESH trial j of the chain (the replayed trials in front of them emit nothing).
// `pre` (optional): sub_iters evaluations made in advance, one per fresh trial.
ZF_HD inline void zf_decide_pass(zf_control* ctl, const double* packs, double* trace,
                                 const double* beta_ring = nullptr, const zf_trial_eval* pre = nullptr) {
    if (ctl->status != ZF_RUNNING) return;
    const int sub = ctl->sub_iters > 0 ? ctl->sub_iters : 1;
    const int ring = ctl->ring_size > 0 ? ctl->ring_size : 3;
    const int stride = sub * ZF_PACK_LEN;
    const int lag = ctl->lag;
    zf_control c = *ctl;   // examined on a copy, written back once
    if (ctl->pend_status != 0) {
        // the pass only materialised the lagging iterates: they are x_k, x_{k-1} now
        if (lag > 0) zf_commit_chain(&c, ctl->cur, ctl->prev, ring, lag);
        c.lag = 0;
        if (c.pend_status > 0) c.status = c.pend_status;
        c.pend_status = 0;
        zf_resolve_beta(&c, beta_ring);
        zf_store_head(ctl, &c);
        return;
    }
    const int nf = zf_fresh_len(ctl);
    int accepted = 0;
    for (int j = 0; j < nf; ++j) {
        const int64_t before = c.nit;
        const double lr_used = c.lr;
        const bool go = zf_decide_step(&c, packs + j * ZF_PACK_LEN, trace, beta_ring, stride, pre ? pre + j : nullptr);
        if (c.nit > before) {
            if (lag + accepted < ZF_MAX_LAG) ctl->lag_lr[lag + accepted] = lr_used;
            accepted += 1;
        }
        if (!go) break;
    }
    if (accepted == nf) {
        // the whole chain holds: its last two iterates are what the pass stored
        zf_commit_chain(&c, ctl->cur, ctl->prev, ring, lag + nf);
        c.lag = 0;
    } else {
        // broken after `accepted` fresh trials: accepted, not stored
        c.lag = lag + accepted;
        if (c.status != ZF_RUNNING && c.lag > 0) {   // final status: first materialise x_k
            c.pend_status = c.status;
            c.status = ZF_RUNNING;
        }
    }
    zf_resolve_beta(&c, beta_ring);
    zf_store_head(ctl, &c);
}

#if defined(__HIPCC__)
// Device: the decide pass of one wave.  Fresh trial j of the chain is evaluated by lane j * lstr
// (sqrt, divisions and the acceptance test cost a few thousand cycles on a single lane; the S of
// them run side by side) and left in LDS; lane 0 then walks the chain with the evaluations at hand.
// `pk`: the pack of this lane's trial, already reduced over ranks (other lanes: ignored);
// `packs`: all packs of the pass, readable by lane 0 (used if an evaluation does not apply);
// `lds_pre`: ZF_MAX_SUB_ITERS entries of LDS.  Must be called by all 64 lanes of ONE wave.
__device__ __forceinline__ void zf_decide_pass_wave(zf_control* ctl, const double* packs,
                                                    const double (&pk)[ZF_PACK_LEN], double* trace,
                                                    const double* beta_ring, int lane, int lstr,
                                                    zf_trial_eval* lds_pre) {
    if (ctl->status != ZF_RUNNING) return;
    const int trial = lane / lstr;
    const double lr = ctl->lr;   // every fresh trial of a chain assumes the current step size
    const double F_x_mine = pk[ZF_PK_FX] + pk[ZF_PK_GX];               // :295
    const double F_x_prev = __shfl_up(F_x_mine, lstr, 64);
    const double F_old = (trial == 0) ? ctl->F_old : F_x_prev;
    zf_trial_eval e;
    zf_eval_trial(ctl, F_old, lr, pk, &e);
    if (lane % lstr == 0 && trial < ZF_MAX_SUB_ITERS) lds_pre[trial] = e;
    if (lane == 0) zf_decide_pass(ctl, packs, trace, beta_ring, lds_pre);
}
#endif
