/* zfista_hip.h - C ABI of libzfista_hip.so, the MI355X (gfx950) engine behind
 * zfista_amd.minimize_proximal_gradient().
 *
 * The reference (zalgo3/zfista) has no FFI: the path it replaces is the Python
 * keyword API zfista/proximal_gradient.py:311-331.  This header is therefore the
 * interface the drop-in's own Python host binds through ctypes (INTEGRATION.md
 * shows the binding a zfista maintainer would add).  Every entry point names the
 * reference lines whose arithmetic it carries.
 *
 * Conventions: plain C, no exceptions across the boundary; every function
 * returns ZF_OK (0) or a negative ZF_ERR_* and leaves a message retrievable with
 * zf_last_error() (thread-local); sizes are int64_t, reals are double (the
 * reference is float64 throughout: zfista/problems.py:22); "dev" pointers are
 * device (HBM) addresses owned by the caller unless stated; `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  One solver object
 * per host thread/process and GPU; stream-ordered; thread-compatible, not
 * thread-safe.
 *
 * Caller-owned output memory is SIZED (ABI 4): every entry point that writes a struct or a
 * library-defined amount of data into caller memory takes the capacity of that buffer and
 * returns ZF_ERR_ARG - writing nothing - when it is too small (a host compiled against an
 * older, smaller zf_control is refused instead of overrun).  Arrays whose length is an
 * argument of the same call (n, m) are the caller's to size.
 */
#ifndef ZFISTA_HIP_H
#define ZFISTA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZF_ABI_VERSION 6

/* ---- status codes ------------------------------------------------------ */
#define ZF_OK 0
#define ZF_ERR_HIP (-1)      /* a HIP runtime call failed (message has details) */
#define ZF_ERR_ARG (-2)      /* invalid argument */
#define ZF_ERR_STATE (-3)    /* call sequence violated */
#define ZF_ERR_NODEVICE (-4) /* no usable GPU */

/* ---- zf_control.status: where the device-resident loop stands ---------- */
#define ZF_RUNNING 0
#define ZF_CONVERGED 1          /* err < tol            proximal_gradient.py:525-529 */
#define ZF_MAXITER 2            /* loop exhausted       proximal_gradient.py:539-543 */
#define ZF_BACKTRACK_FAILED 3   /* RuntimeError         proximal_gradient.py:306-307 */

/* ---- problem kinds (single objective, recognised descriptors) ---------- */
#define ZF_PROBLEM_DIAG_QUAD_L1 1     /* f = 1/2 sum d_i (x_i-c_i)^2, g = lam |x|_1 (+box) */
#define ZF_PROBLEM_LEAST_SQUARES_L1 2 /* f = scale |Ax-b|^2,          g = lam |x|_1 (+box) */
#define ZF_PROBLEM_BLUR_HAAR_L1 3     /* (ABI 5) f = scale |B W^-1 x - b|^2 with B an op_k x op_k correlation with symmetric
                                         boundary and W one orthonormal Haar level: the operator-form LASSO of the
                                         reference's examples/cameraman.ipynb:219-272; g = lam |x|_1 (+box)        */

#define ZF_PACK_LEN 8    /* doubles in one per-trial scalar pack */
#define ZF_MAX_SUB_ITERS 16 /* packs per pass: a rank's pack buffer holds sub_iters x ZF_PACK_LEN doubles */
#define ZF_DEFAULT_SUB_ITERS 16
#define ZF_MAX_LAG (2 * ZF_MAX_SUB_ITERS - 2) /* accepted iterations a solve may run ahead of its stored iterates */
#define ZF_PEND_FLUSH (-1) /* zf_control.pend_status: materialise the lagging iterates, then keep running */
#define ZF_TRACE_COLS 8  /* doubles per accepted iteration in the trace ring */
#define ZF_RING 1024     /* capacity (iterations) of the trace and momentum rings */

/* Device-resident control block of the line-search / termination logic
 * (proximal_gradient.py:279-307,510,525-538).  Lives in HBM; the decide kernel
 * is its only writer while a chunk of trials is in flight; zf_solver_poll()
 * copies it to the host.  Field order is ABI: the Python host mirrors it as a
 * numpy structured dtype and checks zf_sizeof_control(). */
typedef struct zf_control {
    double lr;            /* current learning rate; only ever shrinks (:305)          */
    double F_old;         /* F(x_k) = f + g at the latest accepted iterate (:279)    */
    double f_x, g_x;      /* its two parts                                            */
    double err;           /* max|x_k - y_k| of the latest accepted iteration (:510)  */
    double fun;           /* model value of the latest accepted trial (:149-155)     */
    double tol;           /* outer tolerance (:525)                                   */
    double tol_internal;  /* slack of the sufficient-decrease test (:303, tol=...)   */
    double decay_rate;    /* (:305); == 1 accepts unconditionally (:298)             */
    double f_y;           /* f(y_k) of the current line search                        */
    int64_t nit;          /* accepted outer iterations so far                         */
    int64_t max_iter;
    int64_t trial;        /* trials spent in the current line search                  */
    int64_t max_backtrack;
    int64_t total_trials; /* over the whole run                                       */
    int32_t status;       /* ZF_RUNNING ...                                           */
    int32_t cur;          /* which x buffer holds x_k                                 */
    int32_t nesterov;
    int32_t deprecated;   /* deprecated acceptance test (:300-302)                    */
    int32_t need_grad;    /* least squares: gradient at y_k must be (re)computed      */
    int32_t world;        /* ranks whose packs are summed by the decide step          */
    int32_t accept_mode;  /* (ABI 6) ZF_ACCEPT_*: how the sufficient-decrease test (:303) is evaluated */
    int32_t reserved0;
    double beta_next;     /* momentum factor of the next trial (:533), resolved from the
                             momentum ring by the decide step so that a trial kernel needs
                             ONE dependent scalar load (this block) before its first
                             vector load                                               */
    int32_t ring_size;    /* x buffers: 3 (one iteration per pass), 4 (chains), 6 (run-ahead) */
    int32_t sub_iters;    /* S: trials one pass chains in registers (temporal blocking) */
    int32_t prev;         /* which x buffer holds x_{k-1}                             */
    /* Deferred materialisation (temporal blocking, csrc/zf_decide.h).  A pass stores only the last two
     * iterates of its chain.  When a chain of fresh trials breaks after `a` acceptances (a rejection
     * or a termination in the middle), those a iterations ARE accepted - counters, F_old, trace rows
     * move on - but their iterates exist in no buffer: `lag` counts them, lag_lr[] keeps the step
     * size each was accepted with, and the buffers `cur` / `prev` hold x_{nit-lag}, x_{nit-lag-1}.
     * The next pass recomputes the lagging iterates element-wise in registers (no reductions: their
     * decisions are known) and chains its fresh trials behind them.  pend_status != 0: the next
     * pass only materialises (no fresh trials); afterwards status = pend_status (a final status
     * reached while iterates were lagging) or, for ZF_PEND_FLUSH, the solve simply continues. */
    int32_t lag;
    int32_t pend_status;
    int32_t pass_seq;     /* number of the step whose pass was decided inside its trial launch (0: none yet) */
    double lag_lr[ZF_MAX_LAG];
} zf_control;

/* zf_options.accept_mode / zf_control.accept_mode (ABI 6).  The reference accepts a trial when
 *     F(x+) - F(x_k) <= fun + tol_internal,  fun = <grad f(y), x+ - y> + g(x+) + |x+ - y|^2 / 2 / lr + (f(y) - F(x_k))   (:149-155, :303)
 * - two differences of O(|F|) numbers.  In exact arithmetic F(x_k) and g(x+) cancel and the test reads
 *     [f(x+) - f(y)] - <grad f(y), x+ - y> - |x+ - y|^2 / 2 / lr <= tol_internal.
 * ZF_ACCEPT_REFERENCE evaluates the reference's expression as it stands (default: same decisions as the reference, and
 * below double-precision resolution once |x+ - y|^2 << ulp(F): at n = 1e8 trials are rejected by rounding noise from
 * iteration ~90 on).  ZF_ACCEPT_RESOLVED (opt-in; separable problems) evaluates the second form with f(x+) - f(y)
 * accumulated element by element (pack slot 7), so the test resolves ~1e-16 of the step, not of F; f(x+) is then reported
 * as f(y) + [f(x+) - f(y)].  Iterates of accepted trials are the same arithmetic either way.  One difference by design:
 * with F(x_k) = inf (x_k outside the box) the reference's expression accepts every trial (-inf <= -inf); the resolved form
 * still tests the smooth part. */
#define ZF_ACCEPT_REFERENCE 0
#define ZF_ACCEPT_RESOLVED 1

typedef struct zf_problem_desc {
    int32_t kind;      /* ZF_PROBLEM_*                                                */
    int32_t world;     /* ranks sharing the decision vector (1 = unsharded)           */
    int32_t rank;
    int32_t row_sharded; /* LEAST_SQUARES_L1, world > 1: 0 = the COLUMNS of A and x are split over the
                            ranks (b replicated; the m-vector A_p x_p is exchanged once per trial);
                            1 = the ROWS of A and b are split, x is replicated on every rank and the
                            n-vector A_p^T r_p is exchanged ("all-reduce of A^T r")          */
    int64_t n;         /* length of x held by this rank (its shard; the whole x if row_sharded) */
    int64_t m_rows;    /* least squares: rows of A held by this rank (0 otherwise)   */
    const double* d;   /* dev, n   - DIAG_QUAD_L1                                     */
    const double* c;   /* dev, n   - DIAG_QUAD_L1                                     */
    const double* A;   /* dev, m_rows x n row-major - LEAST_SQUARES_L1                */
    const double* b;   /* dev, m_rows                                                 */
    double scale;      /* LEAST_SQUARES_L1: f = scale |Ax-b|^2                        */
    double lam;        /* l1 weight                                                   */
    double box_lo;     /* -inf / +inf when there is no box                            */
    double box_hi;
    /* (ABI 5) ZF_PROBLEM_BLUR_HAAR_L1: x = the Haar coefficients [cA, cH, cV, cD] (each op_h/2 x op_w/2, row-major) of an
     * op_h x op_w image (both even; n = m_rows = op_h * op_w), b = the observed image (dev), op_taps = the correlation
     * kernel, op_k x op_k row-major (dev; op_k odd, <= 15).  Zero for the other kinds. */
    int64_t op_h, op_w;
    const double* op_taps;
    int32_t op_k;
    int32_t op_reserved;
} zf_problem_desc;

typedef struct zf_options {  /* keyword arguments of proximal_gradient.py:317-330 */
    double lr;
    double tol;
    double tol_internal;
    double decay_rate;
    int64_t max_iter;
    int64_t max_backtrack_iter;
    int32_t nesterov;
    int32_t deprecated;
    int32_t sub_iters;   /* S: trials one pass over the data chains in registers, for separable
                            problems (temporal blocking): 0 = library default (ZF_SUB_ITERS in the
                            environment, else ZF_DEFAULT_SUB_ITERS), 1 / 2 / 4 / 8 / 16 explicit.  Results do not depend on it;
                            lam >= 0, lr > 0, decay_rate > 0 are required by the fused kernels.     */
    int32_t accept_mode; /* (ABI 6; was reserved = 0) ZF_ACCEPT_REFERENCE or ZF_ACCEPT_RESOLVED (ZF_PROBLEM_DIAG_QUAD_L1 only) */
} zf_options;

typedef struct zf_solver zf_solver; /* opaque; owns x ring, partials, control, rings */

/* ---- library, device, memory ------------------------------------------- */
int zf_abi_version(void);
const char* zf_last_error(void);
int64_t zf_sizeof_control(void);
int zf_device_count(int* count);
int zf_set_device(int device);
int zf_malloc(void** dev_ptr, int64_t bytes);
int zf_free(void* dev_ptr);
int zf_memcpy_h2d(void* dst_dev, const void* src_host, int64_t bytes, void* stream);
int zf_memcpy_d2h(void* dst_host, const void* src_dev, int64_t bytes, void* stream);
int zf_memcpy_d2d(void* dst_dev, const void* src_dev, int64_t bytes, void* stream);
int zf_stream_synchronize(void* stream);
/* release the library's staging workspaces (one per host thread and device, used by zf_host_* / zf_dev_*);
 * call when none of those is in flight - a later call allocates again */
int zf_shutdown(void);

/* ---- RCCL communicator (x sharded over the GPUs of one node) -----
 * One rank per GPU/process.  Rank 0 calls zf_comm_unique_id and ships the 128 bytes to the other ranks
 * over any host channel; then every rank calls zf_comm_create (collective) with its GPU current.
 * librccl is loaded with dlopen on first use - single-GPU users never need it. */
typedef struct zf_comm zf_comm;
int zf_comm_unique_id(void* id128);
int zf_comm_create(zf_comm** out, int32_t rank, int32_t world, const void* id128);
int zf_comm_destroy(zf_comm* c);
/* in-process stand-in (tests on one GPU, where RCCL cannot place two ranks): `world` communicators for
 * `world` host threads with a stream each; all-gathers go through a shared staging buffer, a host
 * barrier and stream events.  cap_doubles bounds the per-rank count of an all-gather. */
int zf_comm_create_local_group(zf_comm** out_world, int32_t world, int64_t cap_doubles);
int zf_comm_info(zf_comm* c, int32_t* rank, int32_t* world);
/* What the communicator says about itself (ABI 5): nccl_* come from RCCL's own queries of the ncclComm_t
 * (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 for the in-process stand-in), not from what the caller
 * passed to zf_comm_create; `library` is the file the RCCL symbols were resolved from.  out_bytes = capacity of *out
 * (a short buffer is refused before anything is written). */
typedef struct zf_comm_desc {
    int32_t rank, world;      /* as given at creation */
    int32_t kind;             /* 0: RCCL communicator, 1: in-process thread-rank group */
    int32_t nccl_count;       /* ranks RCCL reports for this communicator */
    int32_t nccl_user_rank;
    int32_t nccl_device;
    int32_t rccl_version;     /* ncclGetVersion (0: not available) */
    int32_t reserved;
    int64_t all_gathers;      /* zf_comm_all_gather calls issued on this communicator so far */
    char library[256];
} zf_comm_desc;
int zf_comm_describe(zf_comm* c, zf_comm_desc* out, int64_t out_bytes);
/* recv (world x count doubles, rank-major) <- send (count doubles) of every rank; stream-ordered */
int zf_comm_all_gather(zf_comm* c, const double* send_dev, double* recv_dev, int64_t count, void* stream);

/* f(x) and (grad_out_host != NULL) jac_f(x) of ZF_PROBLEM_BLUR_HAAR_L1 for a host vector x (n = h * w doubles): the
 * problem's callbacks outside the device-resident loop (ABI 5) */
int zf_op_eval(const double* taps_dev, int32_t k, const double* b_dev, int64_t h, int64_t w, double scale,
               const double* x_host, double* f_out, double* grad_out_host);

/* ---- the decision step on the host (no GPU needed) ----------------------
 * Same inline function the decide kernel runs (csrc/zf_decide.h); exported so
 * the control logic of proximal_gradient.py:279-307,510,525-543 can be tested
 * on a machine without a GPU.  `packs` holds `world` packs of ZF_PACK_LEN
 * doubles; `trace` is a ZF_RING x ZF_TRACE_COLS ring. */
int zf_decide_host(zf_control* ctl, int64_t ctl_bytes /* == zf_sizeof_control() */, const double* packs,
                   double* trace);

/* ---- device-resident single-objective solver -----------------------------
 * Replaces the body of the outer loop, proximal_gradient.py:474-538, for the
 * recognised problem kinds.  One "step" = one pass over the data = a chain of up to sub_iters
 * line-search trials, each assuming the one before was accepted (least squares: one trial):
 *   trial kernel  : y = x_k + beta (x_k - x_{k-1})            (:534)
 *                   x+ = prox_{lr g}(y - lr grad f(y))         (:148)
 *                   per-block partials of f(y), <grad f,x+-y>, |x+-y|^2,
 *                   g(x+), f(x+), max|x+-y|                    (:140,:150-152,:295,:510)
 *   finalize      : fixed-order reduction of the partials -> one pack per trial
 *   decide        : per trial, in order: model value, acceptance, lr decay, termination,
 *                   trace row (:149-155,:298-305,:525); a chain that holds is committed
 *                   (buffer hand-over); one that breaks after >= 1 accepted trials leaves those
 *                   iterations accepted but not stored (zf_control.lag; csrc/zf_decide.h) and
 *                   the next pass recomputes them in registers ahead of its fresh trials
 */
int zf_solver_create(zf_solver** out, const zf_problem_desc* desc, const zf_options* opt,
                     void* stream);
int zf_solver_destroy(zf_solver* s);
/* copy x0 (dev, n) into the ring and evaluate f(x0), g(x0) -> local init pack (:463-466) */
int zf_solver_enqueue_init(zf_solver* s, const double* x0_dev);
/* consume the (gathered) init packs: F_old = sum over ranks; world == 1 may skip the gather */
int zf_solver_enqueue_init_commit(zf_solver* s);
/* momentum factors beta_j for accepted-iteration indices first..first+count-1 (:531-533) */
int zf_solver_set_beta(zf_solver* s, int64_t first, const double* beta_host, int64_t count);
/* world == 1: enqueue `steps` complete steps (trial+finalize+decide), no host sync; a step accepts
 * between 0 and sub_iters iterations, so poll before the device can be ZF_RING iterations ahead.
 * CONTRACT: a step runs the pass the host EXPECTS from the control block of its last zf_solver_poll (advanced on
 * the assumption that every trial since was accepted).  When the device takes another turn - a trial rejected, a
 * termination inside a chain - the steps already enqueued behind it may be no-ops (no kernel finds its shape, the
 * control block stays as it is) until the next zf_solver_poll: N steps advance the solve by AT MOST N passes, and
 * only the first step after a poll is guaranteed to run.  Behind a chunk that saw rejections the library launches a
 * fallback kernel with every step, so such chunks lose no passes.  ZF_SPECULATE=0 launches every shape always. */
int zf_solver_enqueue_steps(zf_solver* s, int64_t steps);
/* after initialisation: fix the launch geometry (interleaved tiles per workgroup) of the trial
 * kernel.  It is a function of n only - never of a timing measurement, the chain length or the
 * device - because it decides the rounding of the reduced sums: the same problem takes the same
 * decisions in every process.  ZF_TILES_PER_WG=<n> overrides it (experiments). */
int zf_solver_autotune(zf_solver* s, int32_t* chosen_tiles);
/* make the next step materialise iterates that lag behind the accepted count (zf_control.lag),
 * so that buffers cur / prev hold x_k, x_{k-1} afterwards; stream-ordered, no-op without lag */
int zf_solver_flush(zf_solver* s);
/* world > 1 with a communicator attached: the solver issues the exchanges of a sharded step itself
 * (packed all-gather of the scalar packs; for column-sharded least squares also of A_p x_p) on its
 * stream - zf_solver_enqueue_steps then works for world > 1 and a pass needs no host code.
 * zf_solver_enqueue_init_all = init + exchanges + commit in one call (world == 1 too). */
int zf_solver_set_comm(zf_solver* s, zf_comm* comm);
int zf_solver_enqueue_init_all(zf_solver* s, const double* x0_dev);
/* world > 1 without one: the two halves of a step; the caller gathers pack_local -> pack_all between them */
int zf_solver_enqueue_trial(zf_solver* s);
int zf_solver_enqueue_decide(zf_solver* s);
/* device addresses of this rank's packs (sub_iters x ZF_PACK_LEN doubles) and of the gathered
 * packs (world x sub_iters x ZF_PACK_LEN doubles, rank-major); zf_solver_sub_iters() tells S */
int zf_solver_sub_iters(zf_solver* s, int32_t* sub_iters);
/* streaming return_all (proximal_gradient.py:521-524): from now on every trial also stores its
 * iterate x_{k+1} into slot (k + 1) % cap_slots of the caller-owned ring `hist_dev` (slots `stride`
 * doubles apart, stride >= n and a multiple of 64) - 8 more bytes per element and iteration, no host
 * transfer, chains stay 8 long.  Slot 0 (x0) is the caller's.  Chain lengths 1 and 8. */
int zf_solver_set_history(zf_solver* s, double* hist_dev, int64_t cap_slots, int64_t stride);
/* change max_iter of a live solve (stream-ordered); a solve stopped by ZF_MAXITER resumes when
 * the new bound is above nit - how a caller continues `res.nit < max_iter` runs (:539) */
int zf_solver_set_max_iter(zf_solver* s, int64_t max_iter);
int zf_solver_pack_ptrs(zf_solver* s, double** pack_local_dev, double** pack_all_dev);
/* make the solver write / read caller-owned pack buffers instead (e.g. torch
 * tensors the collective runs on); sizes as above; must outlive the solver */
int zf_solver_set_pack_buffers(zf_solver* s, double* pack_local_dev, double* pack_all_dev);
/* sharded least squares.  Column blocks (x split): the m-vector A_p x_p of every rank
 * is exchanged once per trial.  Row blocks (desc.row_sharded, x replicated): the n-vector
 * 2 scale A_p^T r_p is exchanged instead (s_part / s_all then hold n and world x n doubles) and the
 * init needs no exchange of its own.  Between
 * zf_solver_enqueue_trial() and zf_solver_enqueue_trial_finish() (and between
 * zf_solver_enqueue_init() and zf_solver_enqueue_init_finish()) the caller gathers
 * s_part (m doubles) of all ranks into s_all (world x m, rank-major); the finish call
 * adds the parts in rank order.  Both finish calls are no-ops for other problems. */
int zf_solver_svec_ptrs(zf_solver* s, double** s_part_dev, double** s_all_dev);
int zf_solver_set_svec_buffers(zf_solver* s, double* s_part_dev, double* s_all_dev);
int zf_solver_enqueue_trial_finish(zf_solver* s);
int zf_solver_enqueue_init_finish(zf_solver* s);
/* synchronise the stream, copy out the control block and (trace_host != NULL) the trace ring.
 * ctl_bytes / trace_bytes: capacity of the two host buffers; ZF_ERR_ARG, nothing written, unless
 * ctl_bytes >= zf_sizeof_control() and trace_bytes >= ZF_RING * ZF_TRACE_COLS * 8 */
int zf_solver_poll(zf_solver* s, zf_control* ctl_host, int64_t ctl_bytes, double* trace_host, int64_t trace_bytes);
/* device address / host copy of the latest accepted iterate x_k (local shard) */
int zf_solver_x_dev(zf_solver* s, const double** x_dev);
int zf_solver_get_x(zf_solver* s, double* x_host, int64_t count /* capacity in doubles, >= n */);
/* checkpoint / resume: zf_solver_poll + zf_solver_get_x + zf_solver_get_x_prev are the state of a
 * solve (x_k, x_{k-1}, control block); zf_solver_restore puts it into a freshly created solver
 * instead of zf_solver_enqueue_init(+_commit), after which the host re-uploads the momentum
 * factors from accepted count `nit` on.  The resumed solve continues bit for bit. */
int zf_solver_get_x_prev(zf_solver* s, double* x_host, int64_t count /* >= n */);
int zf_solver_restore(zf_solver* s, const double* xk_dev, const double* xprev_dev, const zf_control* saved,
                      int64_t saved_bytes /* == zf_sizeof_control(): a block of another layout is refused */);
/* average duration (ms) of the trial kernel over the launches since the last
 * call, measured with HIP events on the solver's stream; resets the window */
int zf_solver_trial_kernel_ms(zf_solver* s, double* avg_ms, int64_t* launches);
/* since creation: out[0] = trial steps issued, out[1] = trial kernels launched for them; with count >= 4 also
 * out[2] = out[3] = 0 (ABI 5: the counts of a multi-pass kernel that round 5 withdrew).  A chained pass has
 * one of several shapes (full chain, short, general, a mid chain of 9 .. 15 trials) and needs one kernel; the host
 * launches the one it expects once a poll has shown it the control block - unsharded, and sharded through the
 * library's communicator; behind a chunk that saw rejections the general body rides along as the complement of the
 * expected kernel (on small grids: alone); with nothing known, the three kernels that between them run every shape.
 * With count >= 6 also out[4] = run-ahead passes launched, out[5] = those of them launched while their predecessor
 * was still in flight: on grids the device holds at once consecutive full chains go alternately to the solver's stream and a
 * second one of its own, pass p + 1 running while pass p is finalised (zf_runahead_kernel; ZF_RUNAHEAD=0: off; six iterate
 * buffers instead of four).  zf_solver_enqueue_steps returns with the solver's stream made to wait for the second.
 * (ABI 6) With count >= 8, as of the last zf_solver_poll: out[6] = waits of run-ahead passes that GAVE UP (a workgroup's
 * for its predecessor workgroup, a deciding wave's for the decision before it: ZF_RUNAHEAD_SPIN_LIMIT polls, ~15 ms -
 * the device did not hold two passes of this solver at once, i.e. it is shared), out[7] = run-ahead passes that turned
 * out VOID (a prediction failed, or a wait gave up).  After the first wait that gave up the solver launches no further
 * run-ahead passes (count >= 11: out[10] = 1) and runs one launch per pass - same results, no 15 ms stalls.
 * With count >= 10: out[8] = passes launched AHEAD at kernel granularity, out[9] = those of them void: sharded solves through
 * the library's communicator (and, ZF_AHEAD_UNSHARDED=1, unsharded grids the run-ahead kernel does not take) run the trial
 * kernels of consecutive exactly predicted full / mid chains back to back on the solver's stream, each on the head the host
 * expects, while finalisation, all-gather and decide of the pass before run on the second stream (ZF_AHEAD=0: off). */
int zf_solver_launch_counts(zf_solver* s, int64_t* out, int64_t count /* >= 2 */);
/* the same window split by the shape of the pass, which the kernel logs itself: out[0], out[1] = mean
 * ms and count of full chains (sub_iters fresh trials, nothing replayed); out[2], out[3] = every other
 * pass (shorter chains, replays, materialise-only).  Resets the window. */
int zf_solver_pass_stats(zf_solver* s, double* out, int64_t count /* >= 4 */);
/* the same window plus out[4] = fresh trials and out[5] = replayed iterations the other passes carried */
int zf_solver_pass_stats_ex(zf_solver* s, double* out, int64_t count /* >= 6 */);
/* sharded solves, timing on (ABI 5): out[0] = mean ms, out[1] = count of the per-pass pack exchanges since the last
 * call, each timed on this rank's stream from "my packs are ready" to "the gathered packs are here" */
int zf_solver_exchange_stats(zf_solver* s, double* out, int64_t count /* >= 2 */);
/* timing on (ABI 5): one (shape, milliseconds) pair per launch that ran a pass since the last call, oldest first; shape =
 * fresh trials | lagging iterations << 5; launches whose pass turned out void (it ran ahead on a head that did not come true)
 * are left out; *count = pairs written (<= cap_pairs) */
int zf_solver_pass_records(zf_solver* s, double* out, int64_t cap_pairs, int64_t* count);
int zf_solver_set_timing(zf_solver* s, int32_t enabled);

/* ---- vector kernels for opaque (Python) callbacks ------------------------
 * The solver's own O(n) arithmetic when f/g/jac_f/prox are arbitrary host
 * callables: v = y - lr*jac (:148), {<jac,x-y>, |x-y|^2, max|x-y|} (:150-152,
 * :510) and y = x + beta (x - x_old) (:534).  Host pointers in, host pointers
 * out; the library stages through its own device workspace. */
int zf_host_grad_step(double* v_host, const double* y_host, const double* jac_host, double lr,
                      int64_t n);
int zf_host_model_terms(const double* jac_host, const double* x_host, const double* y_host,
                        int64_t n, double out3[3]);
int zf_host_momentum(double* y_out_host, const double* x_host, const double* x_old_host,
                     double beta, int64_t n);
/* the same three expressions on device vectors (callbacks written against device tensors:
 * iterates stay in HBM; zf_dev_model_terms synchronises `stream` to return its three scalars) */
int zf_dev_grad_step(double* v_dev, const double* y_dev, const double* jac_dev, double lr, int64_t n, void* stream);
int zf_dev_model_terms(const double* jac_dev, const double* x_dev, const double* y_dev, int64_t n,
                       double out3_host[3], void* stream);
int zf_dev_model_terms_async(const double* jac_dev, const double* x_dev, const double* y_dev, int64_t n,
                             double* out3_dev, void* stream);   /* no synchronisation; scalars stay on the device */
int zf_dev_momentum(double* y_out_dev, const double* x_dev, const double* x_old_dev, double beta, int64_t n,
                    void* stream);
/* Multi-objective trial with callbacks on device tensors (m >= 2): the solver's own expressions of
 * _dual_minimized_fun_jac (:162-173) around the caller's prox_wsum_g and g.
 *   zf_dev_mo_combine:    v = y - lr (w @ J) (:164), ss_dev[0] = |w @ J|^2 (:171); J m x n row-major, w on the host
 *   zf_dev_mo_post_terms: out_dev[0..m) = J_i . (p - y) (:173), out_dev[m] = |p - v|^2 (:168)
 * stream-ordered, nothing synchronised (the caller fetches the scalars with its own in one transfer) */
int zf_dev_mo_combine(double* v_dev, const double* y_dev, const double* J_dev, const double* w_host, double lr,
                      int32_t m, int64_t n, double* ss_dev, void* stream);
int zf_dev_mo_post_terms(const double* J_dev, const double* y_dev, const double* p_dev, const double* v_dev,
                         int32_t m, int64_t n, double* out_dev /* m + 1 */, void* stream);

/* ---- operator evaluations at a host point ---------------------------------
 * The callback contract of proximal_gradient.py rows f / g / jac_f /
 * prox_wsum_g (:140-148) for the recognised operators, evaluated on the GPU at
 * a point given in host memory (problem data stays in HBM).  These back the
 * NumPy-callable methods of zfista_amd.problems.* . */
/* out = sign(x) max(|x| - tau, 0), then clip to [lo, hi]   (problems.py:128-137) */
int zf_host_prox_l1_box(double* out_host, const double* x_host, double tau, double lo, double hi,
                        int64_t n);
/* out = sum |x_i|   (the l1 value of g, tests/test_proximal_gradient.py:53) */
int zf_host_asum(const double* x_host, int64_t n, double* out);
/* out = d * (x - c) */
int zf_host_diag_grad(double* out_host, const double* x_host, const double* d_dev,
                      const double* c_dev, int64_t n);
/* f = scale |A x - b|^2 and (grad_out_host != NULL) grad = 2 scale A^T (A x - b)
 * (tests/test_proximal_gradient.py:49-57) */
int zf_ls_eval(const double* A_dev, const double* b_dev, int64_t m_rows, int64_t n, double scale,
               const double* x_host, double* f_out, double* grad_out_host);

/* ---- multi-objective trial (m >= 2), device side ---------------------------
 * The dual of the scalarised subproblem is minimised on the host by SciPy exactly
 * as the reference does (proximal_gradient.py:179-205); every O(n) expression runs
 * on the GPU with x_k, x_{k-1}, y, x+ and J (m x n) resident in HBM.  g / prox are
 * the shifted-l1 + box family of zfista/problems.py:101-138. */
#define ZF_MO_GENERIC 0 /* f, jac_f are host callbacks; J is uploaded per trial     */
#define ZF_MO_JOS1 1    /* zfista/problems.py:193-205  (m = 2)                      */
#define ZF_MO_FDS 2     /* zfista/problems.py:309-328  (m = 3)                      */
typedef struct zf_mo zf_mo;
int zf_mo_create(zf_mo** out, int32_t kind, int32_t m, int64_t n, const double* l1_ratios_host,
                 const double* l1_shifts_host, double box_lo, double box_hi, void* stream);
/* x sharded over ranks in contiguous blocks: this engine holds [offset, offset+n)
 * of n_global features.  `fn` is called synchronously inside every call that reduces (eval_F,
 * prepare, dual_eval, recover, post_terms) with this rank's raw totals and must replace them by
 * the combination over all ranks - sums added in rank order, entry max_index (>= 0) a maximum -
 * so that every rank continues with identical scalars; returns nonzero on failure. */
typedef int (*zf_mo_exchange_fn)(void* ctx, double* vals, int32_t count, int32_t max_index);
int zf_mo_set_shard(zf_mo* s, int64_t n_global, int64_t offset, zf_mo_exchange_fn fn, void* ctx);
/* the same sharding over a communicator of the library instead of a callback: the totals of every reduction are
 * all-gathered on the stream and added in rank order on the device, and zf_mo_solve_dual exchanges once per BATCH
 * of its search (<= m + 1 points: ~3 collectives per trial) instead of once per dual evaluation.  The persistent-kernel
 * search (zf_mo_solve_dual_device / zf_mo_trial_launch) stays single-rank; zf_mo_solve_dual_stream is the device-driven
 * search of a sharded x.  zf_mo_exchange_count: collectives so far. */
int zf_mo_set_comm(zf_mo* s, zf_comm* comm, int64_t n_global, int64_t offset);
int zf_mo_exchange_count(zf_mo* s, int64_t* count);
/* per-coordinate box bounds (host arrays of n) instead of the scalar pair given at creation */
int zf_mo_set_bounds(zf_mo* s, const double* lo_host, const double* hi_host);
int zf_mo_destroy(zf_mo* s);
int zf_mo_set_x0(zf_mo* s, const double* x0_host);              /* :463-465 */
/* point selector `which`: 0 = x_k, 1 = y, 2 = x+ (trial point), 3 = x_{k-1} */
int zf_mo_eval_F(zf_mo* s, int32_t which, double* f_out /* m or NULL */, double* g_out /* m */); /* :279,:295 */
int zf_mo_prepare(zf_mo* s, double* f_y_out /* m */);           /* J = jac_f(y), f(y)   :140,:142 */
/* the same, stream-ordered and without a host round trip (built-in problems, unsharded x): f(y) is
 * formed and kept on the device for zf_mo_solve_dual_device, which hands it back with its result;
 * zf_mo_get_f_y fetches it otherwise (synchronises) */
int zf_mo_prepare_async(zf_mo* s);
/* Fused outer iteration (JOS1 / FDS, unsharded x), on / off: zf_mo_commit and
 * zf_mo_prepare_async only record what is due and the next zf_mo_solve_dual_device forms y = x_k +
 * beta (x_k - x_{k-1}) (:534), f(y) (:140) and J = jac_f(y) (:142) inside its one kernel: one launch and one
 * read-back per trial.  Every other entry point first brings the buffers up to date, so results do not change. */
int zf_mo_set_fused(zf_mo* s, int32_t on);
/* Trials launched AHEAD of their predecessor's result (fused mode; the outer loop :474-538 without the host in its
 * critical path).  zf_mo_trial_launch enqueues one trial (after zf_mo_prepare_async) and returns a ticket;
 * zf_mo_trial_wait returns its result and the acceptance decision (:298-303) the kernel took on F(x+) formed on
 * the device.  A trial launched with gated = 1 - after zf_mo_commit + zf_mo_prepare_async, before the result of
 * the trial before it is known - runs only if that trial was accepted (else it exits at once having touched
 * nothing: *skipped_out = 1; call zf_mo_uncommit, then retry with a smaller step) and takes F(x_k) from that
 * trial's F(x+) on the device (F_old may then be NULL); gated = 2: it also starts its search from that trial's
 * weights (warm_start, :286-288), likewise taken on the device.  At most one trial may be in flight ahead of the
 * one waited for.  *ticket_out = -1: not a device trial (sharded x, m > 8, the grid not co-resident) - use the host loop. */
int zf_mo_trial_launch(zf_mo* s, double lr, const double* F_old /* m or NULL */, int32_t deprecated,
                       const double* w0 /* m or NULL */, double tol, int64_t max_iter, double accept_tol,
                       int32_t decay_is_one, int32_t gated, int32_t* ticket_out);
int zf_mo_trial_wait(zf_mo* s, int32_t ticket, double* w_out, double* fun_out, int64_t* nit_out, int32_t* ok_out,
                     int64_t* evals_out, double* err_out, double* f_x_out, double* g_x_out, double* f_y_out,
                     int32_t* accepted_out, int32_t* skipped_out);
int zf_mo_uncommit(zf_mo* s);   /* undo of zf_mo_commit (+ zf_mo_prepare_async) after a skipped gated trial */
/* The device trial keeps its whole grid spinning on grid-wide hand-overs, so every workgroup must be resident at
 * once.  What the device can hold of the kernel is checked at launch (hipOccupancyMaxActiveBlocksPerMultiprocessor;
 * if it cannot: *ticket_out = -1 / *ok_out = 0, nothing launched).  If other work occupies CUs at run time a wait
 * gives up after ZF_MO_SPIN_LIMIT polls (default 2^24, seconds): the reducer publishes no totals, every workgroup
 * leaves, *ok_out = -1.  Then: zf_mo_trial_wait the trial launched ahead (it was skipped), zf_mo_uncommit,
 * zf_mo_invalidate_prepare (y, f(y), J of that trial are due again) and continue with zf_mo_solve_dual +
 * zf_mo_recover. */
int zf_mo_invalidate_prepare(zf_mo* s, int32_t ticket /* of the trial that gave up; -1: the most recent launch */);
int zf_mo_debug_force_timeout(zf_mo* s, int32_t launches);   /* test hook: the next `launches` device trials give up */
int zf_mo_get_f_y(zf_mo* s, double* f_y_out /* m */);
int zf_mo_set_jac(zf_mo* s, const double* J_host);              /* generic kind         :142 */
/* out[0..m) = g_i(p), out[m] = |p-v|^2, out[m+1] = |w@J|^2, out[m+2..2m+2) = J_i.(p-y)   :162-173 */
int zf_mo_dual_eval(zf_mo* s, double lr, const double* w_host, double* out);
/* H_out (m x m): the generalised Hessian of the dual (:161-177) at w, d jac_i / d w_k on the quadratic piece w sits
 * in - from the derivative of prox_wsum_g through its composed soft-thresholds and the clip (problems.py:126-138);
 * what the device-side search builds its Newton model from instead of finite-difference probes.  m <= 5. */
int zf_mo_dual_hessian(zf_mo* s, double lr, const double* w_host, double* H_out);
/* the whole dual search of one trial inside the library (opt-in replacement of the two SciPy
 * calls :179-205): m = 2 bracketing root finder on the monotone derivative, m >= 3 projected
 * Newton on the simplex; every evaluation is one zf_mo_dual_eval.  *ok_out = 0: not attempted
 * (non-finite start), fall back to the reference's calls. */
int zf_mo_solve_dual(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated,
                     const double* w0, double tol, int64_t max_iter, double* w_out, double* fun_out,
                     int64_t* nit_out, int32_t* ok_out, int64_t* evals_out);
/* (ABI 6) the same search for an x SHARDED over a library communicator (zf_mo_set_comm) with the state machine in DEVICE
 * memory: a batch = evaluation kernel -> reduce -> ONE zf_comm_all_gather -> a one-wave kernel that adds the totals in rank
 * order, composes D(w), grad D(w) (:165-177) and advances the machine; batches are enqueued back to back and the host looks
 * once per six of them (zf_mo_solve_dual synchronises with the host once per batch).  What dual_solver="device" runs when x
 * is sharded.  Arguments and results as zf_mo_solve_dual; zf_mo_recover follows. */
int zf_mo_solve_dual_stream(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated,
                            const double* w0, double tol, int64_t max_iter, double* w_out, double* fun_out,
                            int64_t* nit_out, int32_t* ok_out, int64_t* evals_out);
/* the same search and the primal recovery (:206, :510) inside ONE persistent kernel: the dual
 * evaluations run on register-resident (J, y), their sums are combined by a last-arriver reduction,
 * the solver's state machine advances on the device; f(x+), g(x+) (:295) come out of the same pass
 * (f_x_out[0] = NaN when f is a host callback); one launch and one read-back per trial.
 * *ok_out = 0: not attempted (non-finite start, x sharded over ranks, m > 8, grid not co-resident) - fall back to
 * zf_mo_solve_dual / the reference's calls + zf_mo_recover.  *ok_out = -1: a grid-wide wait gave up (see
 * zf_mo_invalidate_prepare); same fall-back. */
int zf_mo_solve_dual_device(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated,
                            const double* w0, double tol, int64_t max_iter, double* w_out, double* fun_out,
                            int64_t* nit_out, int32_t* ok_out, int64_t* evals_out, double* err_out,
                            double* f_x_out /* m or NULL */, double* g_x_out /* m or NULL */,
                            double* f_y_out /* m or NULL: the f(y) used; f_y may be NULL after zf_mo_prepare_async */);
/* diagnostics of the last device solve: [0] batches (grid-wide hand-overs), [1] dual evaluations,
 * [2..5] shader-clock cycles of workgroup 0: whole kernel / evaluation loops / hand-overs / solver steps */
int zf_mo_solve_stats(zf_mo* s, int64_t* out, int64_t count /* >= 6 */);
int zf_mo_recover(zf_mo* s, double lr, const double* w_host, double* err_out);   /* :206, :510 */
int zf_mo_commit(zf_mo* s, double beta, int32_t nesterov);      /* :530-538 */
int zf_mo_get(zf_mo* s, int32_t which, double* host, int64_t count /* >= n */);
int zf_mo_put(zf_mo* s, int32_t which, const double* host);
int zf_mo_get_jac(zf_mo* s, double* J_host, int64_t count /* >= m * n */);
int zf_mo_prox_host(zf_mo* s, const double* weight_host, const double* x_host, double* out_host); /* problems.py:119-138 */
/* generic kind, after the user's prox callback produced p (host):
 * out[0..m) = J_i.(p - y), out[m] = |p - (y - lr w@J)|^2          proximal_gradient.py:168,173 */
int zf_mo_post_terms(zf_mo* s, double lr, const double* w_host, const double* p_host, double* out);

/* ---- device vector kernels (device pointers) ----------------------------- */
int zf_eval_diag_l1(const double* x_dev, const double* d_dev, const double* c_dev, double lam,
                    int64_t n, double out2_host[2], void* stream); /* f(x), g(x) */

#ifdef __cplusplus
}
#endif
#endif /* ZFISTA_HIP_H */
