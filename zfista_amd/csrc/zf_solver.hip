// zf_solver.hip - device-resident single-objective proximal-gradient solver
// and the C ABI of include/zfista_hip.h (gfx950 only).
//
// Replaces the outer loop body of zfista/proximal_gradient.py:474-538 for the
// recognised problem kinds.  The host enqueues "steps" (one line-search trial
// each) back to back on one HIP stream; acceptance, lr decay, termination and
// buffer rotation happen in a one-workgroup decide kernel, so there is no host
// round trip inside a chunk of steps.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <tuple>
#include <vector>

#include "zf_common.h"
#include "zf_decide.h"
#include "zf_kernels_gemv.h"
#include "zf_kernels_ls_small.h"
#include "zf_kernels_op.h"
#include "zf_kernels_step.h"
#include "zf_trial_launch.h"

thread_local char zf_errbuf[512] = "";

// ---------------------------------------------------------------------------
// finalize / decide kernels
// ---------------------------------------------------------------------------
// s_out = sum over ranks (rank order) of the gathered parts A_p x_p; slot as in zf_gemv_rows_kernel
__global__ __launch_bounds__(ZF_BLOCK) void zf_sum_parts_kernel(const zf_control* ctl, const double* __restrict__ s_all,
                                                                int world, int64_t m, zf_ring3 sr, int slot) {
    int idx = 0;
    if (slot >= 0) {
        if (ctl->status != ZF_RUNNING) return;
        idx = (ctl->cur + slot) % 3;
    }
    double* __restrict__ out = sr.p[idx];
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < m; i += stride) {
        double t = s_all[i];
        for (int r = 1; r < world; ++r) t += s_all[(int64_t)r * m + i];
        out[i] = t;
    }
}

// sharded x: packs = world x sub x ZF_PACK_LEN (rank-major).  The sub-iterations of a pass are
// examined in order; the first rejection / termination discards the speculative rest.
// stamped != 0 (separable problems): every rank's packs carry zf_pack_stamp of the control block their pass read.
// A step whose predicted shape kernel did not match (zf_predict_parts) produced none: the gathered packs are those of
// an earlier step, and this decide step must leave the control block alone - the host finds it unchanged at its next
// poll and launches every shape again.  All ranks hold the same control block and see the same gathered packs, so
// all of them skip together.
__global__ __launch_bounds__(64) void zf_decide_kernel(zf_control* ctl, const double* packs, double* trace,
                                                       const double* beta_ring, int sub, int stamped) {
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    if (stamped != 0) {
        const double want = zf_pack_stamp(ctl);
        const int world = ctl->world;
        for (int r = 0; r < world; ++r)
            if (packs[(int64_t)r * sub * ZF_PACK_LEN + 6] != want) return;
    }
    double pk[ZF_PACK_LEN] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (lane < sub) zf_reduce_packs(packs + lane * ZF_PACK_LEN, ctl->world, sub * ZF_PACK_LEN, pk);   // rank order
    zf_decide_pass_wave(ctl, packs, pk, trace, beta_ring, lane, 1, s_pre);
}

// The decide step of a SHARDED pass that ran ahead of its predecessor's decision (zf_decide_ahead, zf_kernels_step.h):
// as zf_decide_kernel, but the control block is first checked against the head the pass's trial kernel ran on - a
// pass on a head that did not come true is void - and the done / good word is published for the trial kernel two
// passes on.  All ranks hold the same block and the same gathered packs: all of them decide, or void, together.
__global__ __launch_bounds__(64) void zf_decide_ahead_kernel(zf_control* ctl, const double* packs, double* trace, const double* beta_ring,
                                                             int sub, zf_ahead_check H, unsigned long long* ra_word, unsigned* ra_stats,
                                                             int* pass_log, int pass_slot, int pass_tag) {
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    bool packs_ok = true;
    {
        const double want = zf_pack_stamp(ctl);
        const int world = ctl->world;
        for (int r = 0; r < world; ++r)
            if (packs[(int64_t)r * sub * ZF_PACK_LEN + 6] != want) packs_ok = false;
    }
    double pk[ZF_PACK_LEN] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (lane < sub) zf_reduce_packs(packs + lane * ZF_PACK_LEN, ctl->world, sub * ZF_PACK_LEN, pk);   // rank order
    zf_decide_ahead(ctl, packs, pk, trace, beta_ring, lane, 1, s_pre, H, packs_ok, ra_word, ra_stats, pass_log, pass_slot, pass_tag);
}

__global__ void zf_set_max_iter_kernel(zf_control* ctl, int64_t max_iter) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ctl->max_iter = max_iter;
    if (ctl->status == ZF_MAXITER && ctl->nit < max_iter) ctl->status = ZF_RUNNING;
    if (ctl->pend_status == ZF_MAXITER && ctl->nit < max_iter) ctl->pend_status = 0;   // keep going instead
}

// ask the next pass to materialise lagging iterates (x_k, x_{k-1} into buffers) and nothing else
__global__ void zf_flush_kernel(zf_control* ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ctl->status == ZF_RUNNING && ctl->lag > 0 && ctl->pend_status == 0) ctl->pend_status = ZF_PEND_FLUSH;
}

// after the host refilled the momentum ring: re-resolve the factor of the pending trial
__global__ void zf_refresh_beta_kernel(zf_control* ctl, const double* beta_ring) {
    if (threadIdx.x == 0 && blockIdx.x == 0) zf_resolve_beta(ctl, beta_ring);
}

// initial F(x0): partials [f raw, |x|_1, violations] -> init pack [f, g, 0...]
struct zf_init_args {
    const double* partials;
    int nblocks;
    double f_scale, lam;
    const double* f_ext;
    double* pack;
    int contribute_g;   // row-sharded least squares: x is replicated, only rank 0 contributes g(x0)
};
__global__ __launch_bounds__(ZF_FIN_BLOCK) void zf_init_finalize_kernel(zf_init_args I) {
    __shared__ double lds[(ZF_FIN_BLOCK / 64) * 8];
    __shared__ double totals[8];
    zf_finalize_partials(I.partials, I.nblocks, 3, -1, lds, totals);
    if (threadIdx.x == 0) {
        const double f = I.f_ext ? *I.f_ext : I.f_scale * totals[0];
        double g = I.lam * totals[1];
        if (totals[2] > 0.0) g = INFINITY;   // zfista/problems.py:104-106
        I.pack[0] = f;
        I.pack[1] = I.contribute_g ? g : 0.0;
        for (int k = 2; k < ZF_PACK_LEN; ++k) I.pack[k] = 0.0;
        I.pack[2] = I.contribute_g ? totals[2] : 0.0;
    }
}
__global__ void zf_init_commit_kernel(zf_control* ctl, const double* packs, int f_replicated, int stride) {
    if (threadIdx.x || blockIdx.x) return;
    double f = packs[0], g = packs[1], viol = packs[2];
    for (int r = 1; r < ctl->world; ++r) {
        if (!f_replicated) f = f + packs[r * stride + 0];
        g = g + packs[r * stride + 1];
        viol = viol + packs[r * stride + 2];
    }
    if (viol > 0.0) g = INFINITY;
    ctl->f_x = f;
    ctl->g_x = g;
    ctl->F_old = f + g;   // proximal_gradient.py:279 at the first line search
}

// ---------------------------------------------------------------------------
// solver object
// ---------------------------------------------------------------------------
struct zf_comm;
extern "C" int zf_comm_all_gather(zf_comm* c, const double* send_dev, double* recv_dev, int64_t count, void* stream);
extern "C" int zf_comm_info(zf_comm* c, int32_t* rank, int32_t* world);

struct zf_solver {
    zf_problem_desc desc;
    zf_options opt;
    hipStream_t stream;
    int grid;                 // trial kernel grid
    bool box;
    bool res = false;         // ZF_ACCEPT_RESOLVED: the kernels accumulate f(x+) - f(y) element by element (zf_elem_diag<..., RES>)
    // device memory owned by the solver
    double* xbuf = nullptr;   // ring * n_pad
    double* xb[ZF_MAX_RING] = {};
    int sub = 1;              // FISTA iterations computed per pass (temporal blocking; separable f only)
    int ring = 3;             // iterate buffers (zf_decide.h: zf_free_bufs)
    double* partials = nullptr;   // init-time evaluation partials
    double* blk_part = nullptr;   // ZF_NPART x max_grid per-workgroup partials of the trial kernel
    double* slice_part = nullptr; // ZF_NPART x ZF_FIN_WGS
    unsigned* fin_cnt = nullptr;  // arrival counters: of the finalize workgroups / of the in-kernel finalisation (zf_pass_tail)
    double* grp_part = nullptr;   // ZF_NPART x S x ZF_FIN_GROUPS group rows of the in-kernel finalisation
    bool nt = true;               // nontemporal policy for once-touched streams
    char* ctl_trace = nullptr;    // one allocation: the control block (ZF_CTL_SLOT bytes) and the trace ring behind it
    char* mail = nullptr;         // pinned host mirror of ctl_trace: what a poll copies into
    zf_control* ctl = nullptr;    // = ctl_trace
    double* trace = nullptr;      // = ctl_trace + ZF_CTL_SLOT: ZF_RING * ZF_TRACE_COLS
    double* beta_ring = nullptr;  // ZF_RING
    double* pack_local = nullptr; // ZF_PACK_LEN
    double* pack_all = nullptr;   // world * ZF_PACK_LEN
    // least squares
    double* grad = nullptr;       // n
    double* sbuf = nullptr;       // 3 * m_pad
    zf_ring3 sring = {{nullptr, nullptr, nullptr}};
    double* resid = nullptr;      // m_rows
    double* slab = nullptr;       // slices * n
    double* ls_scal = nullptr;    // [0] f(y) [1] f(x+)
    double* s_part = nullptr;     // sharded: this rank's A_p x_p (m)
    double* s_all = nullptr;      // sharded: gathered parts (world x m, rank-major)
    bool own_svec = true;
    int slices = 1;
    int64_t rows_per_slice = 0;
    bool initialised = false;
    bool own_packs = true;
    bool gemv_mfma = false;       // A^T r on v_mfma_f64_16x16x4 (n % 32 == 0; ZF_GEMV_MFMA=0 disables)
    bool ls_small = false;        // cache-resident A: two fused launches per trial (zf_kernels_ls_small.h)
    zf_op_plan op_plan = {};      // operator problem: which instantiation of the correlation kernels runs it (zf_op_make_plan)
    double* op_buf = nullptr;     // its taps as launched (zero-padded to K x K) and, behind them, the rank-1 factors u, v when the kernel is separable
    const double* op_taps = nullptr, *op_sep = nullptr;
    double* row_part = nullptr;   // ls_small: workgroup sums of the row kernel
    unsigned* ls_cnt = nullptr;
    int64_t ntiles = 1;           // 16 KiB tiles of the trial kernel
    int tiles = 1;                // interleaved tiles per workgroup (zf_solver_autotune picks it)
    int max_grid = 1;
    // trial-kernel timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double ms_total = 0.0;
    int64_t ms_count = 0;
    std::vector<std::pair<int, float>> records;   // (shape, ms) of every timed launch since the last zf_solver_pass_records
    // the per-pass exchange of a sharded solve (zf_gather_packs), bracketed the same way when timing is on
    std::vector<std::pair<hipEvent_t, hipEvent_t>> xev_pool;
    size_t xev_used = 0;
    double xms_total = 0.0;
    int64_t xms_count = 0;
    // the shape of every timed pass, written by the kernel itself (the host cannot know it at launch)
    int* pass_log = nullptr;              // ZF_PASS_LOG slots (device; inside ctl_trace, behind the trace ring)
    int64_t launches = 0;                 // timed launches so far (slot = launches % ZF_PASS_LOG)
    int64_t first_uncollected = 0;
    double ms_full = 0.0, ms_part = 0.0;  // S-trial chains without replay / everything else
    int64_t n_full = 0, n_part = 0;
    int64_t fresh_part = 0, lag_part = 0;   // fresh trials / replayed iterations carried by the other passes
    zf_comm* comm = nullptr;              // RCCL communicator (zf_solver_set_comm): the solver gathers itself
    // Which shape-specific kernel a pass needs is decided on the device; the host PREDICTS it from the control
    // block of its last poll (zf_predict_parts) and launches only that one.  A wrong prediction costs passes that
    // do nothing (no kernel finds its shape, the control block stays as it is), never a wrong result.
    int part_mask = 7;                    // ZF_K_* bits (ZF_K_ALL until a prediction says otherwise)
    int mid_len = 0;                      // ZF_K_MID: trials of the mid chain
    int fb_part = -1, fb_len = 0;         // ZF_K_FALLBACK: the general body runs what this kernel does not (-1: everything)
    bool mid_chains = true;               // ZF_MID_CHAINS=0 at creation: tails of 9 .. 15 trials through the general body (A/B)
    bool short_general = false;           // ZF_SHORT_VIA_GENERAL=1 at creation: the shapes of PART 1 through the general body (A/B)
    bool speculate = true;                // ZF_SPECULATE=0 at creation: always launch every shape
    int64_t steps_issued = 0, kernels_issued = 0;   // trial steps and the shape kernels launched for them (zf_solver_launch_counts)
    int pass_seq = 0;                     // step counter (zf_step_args.pass_seq)
    zf_control shadow;                    // the control block as the host expects it after the passes enqueued so far
    bool shadow_valid = false;            // false until the next poll (after init / restore / flush / set_max_iter ...)
    bool careful = false;                 // the last chunk saw rejections: launch every kernel a pass may need
    int64_t steps_since_poll = 0;         // trial steps issued since the shadow was read
    int64_t polled_rejections = 0;
    // Run-ahead passes (zf_runahead_kernel): consecutive full chains of a grid the device holds at once go alternately
    // to `stream` and `stream2`; stream k has its own rows / group rows / counters / packs (two passes are in flight)
    bool ra = false;                      // eligible (separable f, chains of 16, one rank, a one-round grid) and not switched off (ZF_RUNAHEAD=0)
    int ra_cap = -1;                      // co-resident workgroups of the run-ahead full chain (-1: not asked yet)
    int ra_cap_mid[ZF_MAX_SUB_ITERS] = {};   // ... of the run-ahead mid chain of that length (0: no such kernel; asked at the first use: -1)
    unsigned ra_spin = 1u << 13;          // polls before a wait gives up (ZF_RUNAHEAD_SPIN_LIMIT): ~15 ms, some tens of passes' worth; 0: every pass behind a pass in flight gives up at once (tests)
    hipStream_t stream2 = nullptr;
    hipStream_t stream3 = nullptr;        // sharded run-ahead passes: all-gather + decide of the pass before, beside the two trial streams
    hipEvent_t ra_join3 = nullptr;        // stream3 -> stream at the end of such a run
    bool ra_c_pending = false;            // stream3 holds work `stream` has not been made to wait for
    bool chunk_last = false;              // the step being enqueued is the last of its zf_solver_enqueue_steps call
    bool ra_sharded = true;               // ZF_RUNAHEAD_SHARDED: one-round grids behind a library communicator take run-ahead passes (else passes ahead)
    int64_t ras_passes = 0;               // sharded run-ahead passes launched
    hipEvent_t ra_join = nullptr;         // stream2 -> stream at the end of a run of run-ahead passes
    hipEvent_t ra_fork = nullptr;         // stream -> stream2 in front of a run
    bool ra_b_pending = false;            // stream2 holds work `stream` has not been made to wait for
    bool ra_fork_due = false;             // ra_fork was recorded in front of the current run and stream2 has not waited for it yet
    int ra_last = 0, ra_last2 = 0;        // pass_seq of the last / last but one run-ahead pass of the current run (0: none)
    int ra_last_idx = 0;                  // stream of the last one
    zf_pass_head ra_last_head = {};       // its head (whose buffers the next pass must not write)
    unsigned long long* ra_word = nullptr;
    unsigned* ra_flags = nullptr;         // max_grid + 32 words
    double* blk_part2 = nullptr;
    double* grp_part2 = nullptr;
    unsigned* fin_cnt2 = nullptr;
    double* pack2 = nullptr;
    int64_t ra_passes = 0, ra_ahead = 0;  // run-ahead kernels launched / of them behind a pass still in flight (zf_solver_launch_counts)
    // What the device reports about passes that ran ahead: three counters in the slack of the control block's slot,
    // copied with every poll - [0] waits that gave up (a run-ahead workgroup's, or a deciding wave's), [1] void run-ahead
    // passes, [2] void passes AHEAD (below).  After the first wait that gave up the solver launches no further run-ahead
    // passes (ra_off): the device is being shared, and every such wait costs its whole limit.
    unsigned* ra_stats = nullptr;         // = ctl_trace + ZF_STATS_OFF
    int64_t ra_timeouts = 0, ra_voids = 0, ah_voids = 0;   // as of the last poll
    bool ra_off = false;
    // Passes AHEAD at kernel granularity (zf_trial_kernel<..., AHEAD>, zf_tail_kernel, zf_decide_ahead_kernel): sharded
    // solves through the library's communicator, and (ZF_AHEAD_UNSHARDED) unsharded grids of more than one round.  The
    // trial kernels of consecutive exactly predicted full / mid chains go back to back to `stream`, each on the head the
    // host expects; rows -> packs (-> all-gather) -> decide of pass p run on `stream2` behind an event, beside the
    // trial kernel of pass p + 1; the trial kernel of pass p + 2 is launched behind an event of decide p.
    bool ah = false;                      // eligible (separable f, chains of 16, nontemporal policy, six buffers) and not switched off (ZF_AHEAD=0)
    bool ah_unsharded = false;            // ... also without a communicator
    hipEvent_t ah_evT[4] = {}, ah_evD[4] = {};   // trial kernel done (stream) / decided (stream2), by pass number of the run % 4
    int ah_run = 0;                       // passes of the current run so far
    int ah_last_nf = 0;                   // fresh trials of the last one
    int run_mode = 0;                     // what the current run of passes in flight consists of: 0 none, 1 run-ahead passes (workgroup granularity), 2 passes ahead (kernel granularity) - a run is one or the other
    int64_t ah_passes = 0;                // passes launched ahead since creation
    // streaming return_all: caller-owned ring of iterates in HBM (zf_solver_set_history)
    double* hist = nullptr;
    int64_t hist_cap = 0, hist_stride = 0;
};
constexpr int ZF_PASS_LOG = 4096;
constexpr size_t ZF_CTL_SLOT = (sizeof(zf_control) + 255) / 256 * 256;   // bytes in front of the trace ring (zf_solver::ctl_trace)
constexpr size_t ZF_STATS_OFF = ZF_CTL_SLOT - 32;                         // the run-ahead counters (zf_solver::ra_stats): 8 words in the slot's slack
static_assert(sizeof(zf_control) <= ZF_STATS_OFF, "the run-ahead counters live behind the control block, inside its slot");
constexpr size_t ZF_TRACE_BYTES = sizeof(double) * ZF_RING * ZF_TRACE_COLS;
constexpr size_t ZF_MAIL_BYTES = ZF_CTL_SLOT + ZF_TRACE_BYTES + sizeof(int) * ZF_PASS_LOG;

static int zf_tiles_for(int64_t ntiles);
static bool zf_fin_kernel_mode();
// passes ahead at kernel granularity for UNSHARDED grids of more than one round (ZF_AHEAD_UNSHARDED overrides)
constexpr bool ZF_AHEAD_UNSHARDED_DEFAULT = false;

static int zf_solver_free_all(zf_solver* s) {
    if (s->stream2) (void)hipStreamDestroy(s->stream2);
    if (s->stream3) (void)hipStreamDestroy(s->stream3);
    if (s->ra_join3) (void)hipEventDestroy(s->ra_join3);
    for (int k = 0; k < 4; ++k) {
        if (s->ah_evT[k]) (void)hipEventDestroy(s->ah_evT[k]);
        if (s->ah_evD[k]) (void)hipEventDestroy(s->ah_evD[k]);
    }
    if (s->ra_join) (void)hipEventDestroy(s->ra_join);
    if (s->ra_fork) (void)hipEventDestroy(s->ra_fork);
    void* ptrs[] = {s->op_buf, s->row_part, s->ls_cnt, s->blk_part, s->slice_part, s->fin_cnt, s->grp_part, s->xbuf, s->partials, s->ctl_trace, s->beta_ring,
                    s->ra_word, s->ra_flags, s->blk_part2, s->grp_part2, s->fin_cnt2, s->pack2,
                    s->own_packs ? s->pack_local : nullptr, s->own_packs ? s->pack_all : nullptr,
                    s->grad, s->sbuf, s->resid, s->slab, s->ls_scal,
                    s->own_svec ? s->s_part : nullptr, s->own_svec ? s->s_all : nullptr};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->mail) (void)hipHostFree(s->mail);
    for (auto& e : s->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (auto& e : s->xev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    return ZF_OK;
}

static bool zf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// least squares with an explicit matrix, or with the blur o inverse-Haar operator (zf_kernels_op.h): everything
// around the two applications of A / its adjoint is shared
static bool zf_is_ls(int kind) { return kind == ZF_PROBLEM_LEAST_SQUARES_L1 || kind == ZF_PROBLEM_BLUR_HAAR_L1; }
static zf_op_args zf_op_of(const zf_problem_desc& d, const zf_control* ctl, const zf_op_plan& pl, const double* taps, const double* sep) {
    zf_op_args P;
    P.ctl = ctl;
    P.H = (int)d.op_h;
    P.W = (int)d.op_w;
    P.K = pl.K;
    P.taps = taps;
    P.sep = pl.sep ? sep : nullptr;
    static const int bands = [] { const char* e = getenv("ZF_OP_XCD_BANDS"); return e ? atoi(e) : 1; }();
    P.xcd_bands = bands;
    P.tiles = pl.grid;
    return P;
}
static zf_op_args zf_op_of(const zf_solver* s, const zf_control* ctl) { return zf_op_of(s->desc, ctl, s->op_plan, s->op_taps, s->op_sep); }
static zf_op_fuse zf_op_no_fuse() {
    zf_op_fuse F;
    memset(&F, 0, sizeof(F));
    return F;
}
// The taps as the kernels take them: K x K with K the launched size (1 x 1 is zero-padded to 3 x 3), and - when the
// kernel has rank 1 and ZF_OP_SEPARABLE is not 0 - its factors u, v behind them.  *buf: one device allocation
// (caller frees); `st`: the stream the upload is ordered on.
static int zf_op_prepare(const double* taps_dev, int k, int64_t h, int64_t w, hipStream_t st, zf_op_plan* pl, double** buf,
                         const double** taps_out, const double** sep_out) {
    double host[ZF_OP_MAXK * ZF_OP_MAXK];
    ZF_HIP(hipMemcpyAsync(host, taps_dev, sizeof(double) * k * k, hipMemcpyDeviceToHost, st));
    ZF_HIP(hipStreamSynchronize(st));
    const int K = k < 3 ? 3 : k;
    double up[ZF_OP_MAXK * ZF_OP_MAXK + 2 * (ZF_OP_MAXK + 1)];
    memset(up, 0, sizeof(up));
    const int pad = (K - k) / 2;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) up[(i + pad) * K + j + pad] = host[i * k + j];
    double* u = up + ZF_OP_MAXK * ZF_OP_MAXK;
    double* v = u + ZF_OP_MAXK + 1;
    const char* e = getenv("ZF_OP_SEPARABLE");
    const bool sep = !(e && atoi(e) == 0) && zf_op_factor_rank1(up, K, u, v);
    *pl = zf_op_make_plan(h, w, k, sep);
    ZF_HIP(hipMalloc(buf, sizeof(up)));
    ZF_HIP(hipMemcpyAsync(*buf, up, sizeof(up), hipMemcpyHostToDevice, st));
    ZF_HIP(hipStreamSynchronize(st));   // `up` is a stack object
    *taps_out = *buf;
    *sep_out = *buf + ZF_OP_MAXK * ZF_OP_MAXK;
    return ZF_OK;
}
static int zf_op_grid(const zf_problem_desc& d) { return zf_op_make_plan(d.op_h, d.op_w, d.op_k, false).grid; }

// s[(cur + slot) % 3] = A x[(cur + slot) % 3] inside the loop (ctl given), or sout.p[0] = A xr.p[0] outside it (ctl NULL)
static void zf_launch_apply_A(zf_solver* s, const zf_control* ctl, zf_ring3 xr, zf_ring3 sout, int slot) {
    const zf_problem_desc& d = s->desc;
    const int64_t n = d.n, m = d.m_rows;
    if (d.kind == ZF_PROBLEM_BLUR_HAAR_L1) {
        zf_launch_op_apply(s->op_plan, s->stream, zf_op_of(s, ctl), xr.p[0], xr.p[1], xr.p[2], sout.p[0], sout.p[1], sout.p[2], slot,
                           zf_op_no_fuse());
        return;
    }
    const int V = (n % 2 == 0) ? 2 : 1;
    int gr = (int)((m + GEMV_ROWS - 1) / GEMV_ROWS);
    if (gr > 8 * ZF_MAX_GRID) gr = 8 * ZF_MAX_GRID;
    if (V == 2)
        hipLaunchKernelGGL(zf_gemv_rows_kernel<2>, dim3(gr), dim3(ZF_BLOCK), 0, s->stream, ctl, d.A, xr, sout, slot, m, n);
    else
        hipLaunchKernelGGL(zf_gemv_rows_kernel<1>, dim3(gr), dim3(ZF_BLOCK), 0, s->stream, ctl, d.A, xr, sout, slot, m, n);
}

// What two passes in flight need: the second stream (non-blocking, at the highest priority: what runs there - a waiting
// pass, or the small kernels that finalise and decide a pass beside the next pass's trial kernel - must not queue behind
// that kernel's workgroups), the events, the done / good word and the per-workgroup flags, a second set of rows / group
// rows / counters / packs.  Idempotent.
static hipError_t zf_second_stream(zf_solver* s) {
    if (s->stream2) return hipSuccess;
    hipError_t e;
#define ZF_E(expr)                  \
    do {                            \
        e = (expr);                 \
        if (e != hipSuccess) return e; \
    } while (0)
    int lo = 0, hi = 0;
    ZF_E(hipDeviceGetStreamPriorityRange(&lo, &hi));
    ZF_E(hipStreamCreateWithPriority(&s->stream2, hipStreamNonBlocking, hi));
    ZF_E(hipStreamCreateWithPriority(&s->stream3, hipStreamNonBlocking, hi));
    ZF_E(hipEventCreateWithFlags(&s->ra_join3, hipEventDisableTiming));
    ZF_E(hipEventCreateWithFlags(&s->ra_join, hipEventDisableTiming));
    ZF_E(hipEventCreateWithFlags(&s->ra_fork, hipEventDisableTiming));
    for (int k = 0; k < 4; ++k) {
        ZF_E(hipEventCreateWithFlags(&s->ah_evT[k], hipEventDisableTiming));
        ZF_E(hipEventCreateWithFlags(&s->ah_evD[k], hipEventDisableTiming));
    }
    ZF_E(hipMalloc(&s->ra_word, 128));
    ZF_E(hipMemsetAsync(s->ra_word, 0, 128, s->stream));
    ZF_E(hipMalloc(&s->ra_flags, sizeof(unsigned) * (s->max_grid + 32)));
    ZF_E(hipMemsetAsync(s->ra_flags, 0, sizeof(unsigned) * (s->max_grid + 32), s->stream));
    ZF_E(hipMalloc(&s->blk_part2, sizeof(double) * ZF_NPART * s->sub * s->max_grid));
    ZF_E(hipMalloc(&s->grp_part2, sizeof(double) * ZF_NPART * s->sub * ZF_FIN_GROUPS));
    ZF_E(hipMalloc(&s->fin_cnt2, sizeof(unsigned) * ZF_FIN_CNT_STRIDE * (ZF_FIN_GROUPS + 2)));
    ZF_E(hipMemsetAsync(s->fin_cnt2, 0, sizeof(unsigned) * ZF_FIN_CNT_STRIDE * (ZF_FIN_GROUPS + 2), s->stream));
    ZF_E(hipMalloc(&s->pack2, sizeof(double) * ZF_PACK_LEN * s->sub));
#undef ZF_E
    return hipSuccess;
}

extern "C" int zf_solver_create(zf_solver** out, const zf_problem_desc* desc, const zf_options* opt,
                                void* stream) {
    ZF_REQUIRE(out && desc && opt, "zf_solver_create: null argument");
    ZF_REQUIRE(desc->n >= 1, "zf_solver_create: n must be >= 1");
    ZF_REQUIRE(desc->world >= 1 && desc->rank >= 0 && desc->rank < desc->world,
               "zf_solver_create: bad world/rank");
    ZF_REQUIRE(opt->max_iter >= 1, "zf_solver_create: max_iter must be >= 1");
    ZF_REQUIRE(opt->max_backtrack_iter >= 0, "zf_solver_create: max_backtrack_iter must be >= 0");
    // the fused kernels threshold with tau = lam * lr >= 0 (zf_soft_threshold_nn)
    ZF_REQUIRE(desc->lam >= 0.0 && opt->lr > 0.0 && opt->decay_rate > 0.0,
               "zf_solver_create: needs lam >= 0, lr > 0, decay_rate > 0 (use the callback path otherwise)");
    if (desc->kind == ZF_PROBLEM_DIAG_QUAD_L1) {
        ZF_REQUIRE(desc->d && desc->c, "zf_solver_create: d and c are required");
        ZF_REQUIRE(zf_aligned16(desc->d) && zf_aligned16(desc->c),
                   "zf_solver_create: d and c must be 16-byte aligned");
    } else if (desc->kind == ZF_PROBLEM_LEAST_SQUARES_L1) {
        ZF_REQUIRE(desc->A && desc->b && desc->m_rows >= 1, "zf_solver_create: A, b, m_rows required");
        ZF_REQUIRE(zf_aligned16(desc->A), "zf_solver_create: A must be 16-byte aligned");
    } else if (desc->kind == ZF_PROBLEM_BLUR_HAAR_L1) {
        ZF_REQUIRE(desc->b && desc->op_taps, "zf_solver_create: b (the observed image) and op_taps are required");
        ZF_REQUIRE(desc->op_h >= 2 && desc->op_w >= 2 && desc->op_h % 2 == 0 && desc->op_w % 2 == 0 &&
                       desc->op_h * desc->op_w == desc->n && desc->m_rows == desc->n,
                   "zf_solver_create: the image must be op_h x op_w (both even) with n = m_rows = op_h * op_w");
        ZF_REQUIRE(desc->op_k >= 1 && desc->op_k % 2 == 1 && desc->op_k <= ZF_OP_MAXK && desc->op_k / 2 < desc->op_h &&
                       desc->op_k / 2 < desc->op_w,
                   "zf_solver_create: op_k must be odd, at most 15 and smaller than twice the image");
        ZF_REQUIRE(desc->world == 1, "zf_solver_create: the operator problem is not sharded");
    } else {
        return zf_fail(ZF_ERR_ARG, "zf_solver_create: unknown problem kind");
    }
    zf_solver* s = new (std::nothrow) zf_solver();
    if (!s) return zf_fail(ZF_ERR_ARG, "zf_solver_create: out of host memory");
    s->desc = *desc;
    s->opt = *opt;
    if (const char* e = getenv("ZF_SPECULATE")) s->speculate = atoi(e) != 0;
    if (const char* e = getenv("ZF_MID_CHAINS")) s->mid_chains = atoi(e) != 0;
    if (const char* e = getenv("ZF_SHORT_VIA_GENERAL")) s->short_general = atoi(e) != 0;
    if (const char* e = getenv("ZF_PASS_SEQ_START"))   // (tests: the step counter wraps at 0x7ffffff0)
        s->pass_seq = (int)std::min<long long>(std::max<long long>(atoll(e), 0), 0x7ffffff0LL);
    s->stream = reinterpret_cast<hipStream_t>(stream);
    s->box = !(desc->box_lo == -INFINITY && desc->box_hi == INFINITY);
    if (opt->accept_mode != ZF_ACCEPT_REFERENCE) {
        const bool ok = opt->accept_mode == ZF_ACCEPT_RESOLVED && desc->kind == ZF_PROBLEM_DIAG_QUAD_L1;
        if (!ok) {
            delete s;
            return zf_fail(ZF_ERR_ARG, "zf_solver_create: accept_mode must be ZF_ACCEPT_REFERENCE, or ZF_ACCEPT_RESOLVED for "
                                       "ZF_PROBLEM_DIAG_QUAD_L1 (the element-wise difference f(x+) - f(y) exists for the separable problem only)%s");
        }
        s->res = true;
    }
    const int64_t n = desc->n;
    const int64_t n_pad = (n + 63) & ~int64_t(63);   // keep every ring buffer 512-B aligned
    {   // one workgroup per tile of ZF_TILE_UNITS 16-byte units (zf_kernels_step.h)
        int64_t ntiles = (n / 2 + ZF_TILE_UNITS - 1) / ZF_TILE_UNITS;
        if (ntiles < 1) ntiles = 1;
        if (ntiles > 0x7fffffff) {
            delete s;
            return zf_fail(ZF_ERR_ARG, "zf_solver_create: n too large for one rank");
        }
        s->ntiles = ntiles;
        s->grid = (int)ntiles;            // tiles_per_wg = 1: the largest grid; buffers are sized for it
        s->max_grid = s->grid;
    }
    const char* nt_env = getenv("ZF_NT");
    if (nt_env) s->nt = atoi(nt_env) != 0;
#define ZF_TRY(expr)                                                                    \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            zf_solver_free_all(s);                                                      \
            delete s;                                                                   \
            return zf_fail(ZF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));         \
        }                                                                               \
    } while (0)
    if (desc->kind == ZF_PROBLEM_DIAG_QUAD_L1) {
        // every term of the recursion is elementwise for this f, so one pass may run `sub`
        // consecutive iterations in registers (zf_kernels_step.h); 0 = library default
        int sub = opt->sub_iters;
        const char* se = getenv("ZF_SUB_ITERS");
        if (sub <= 0 && se) sub = atoi(se);
        if (sub <= 0) sub = ZF_DEFAULT_SUB_ITERS;
        if (sub > ZF_MAX_SUB) sub = ZF_MAX_SUB;
        s->sub = sub >= 16 ? 16 : sub >= 8 ? 8 : sub >= 4 ? 4 : sub >= 2 ? 2 : 1;
        // (the kernels of ZF_ACCEPT_RESOLVED exist for chains of 16 and single trials, nontemporal policy)
        if (s->res && s->sub != 16) s->sub = 1;
        if (s->res) s->nt = true;
    }
    s->ring = s->sub > 1 ? 4 : 3;   // x_k, x_{k-1} + the one or two iterates a pass stores
    {   // passes that run ahead of their predecessor's decision (chains of 16 of the separable problem), two ways:
        // * run-ahead passes at WORKGROUP granularity (zf_runahead_kernel): one rank, a grid the device holds at once
        //   (the geometry is a function of n; what the device holds of the kernel is asked of the runtime): full chains of
        //   every variant, mid chains where the per-pass mid chains exist (no box);
        // * passes AHEAD at KERNEL granularity: every grid, boxes too; for sharded solves through the library's
        //   communicator (zf_solver_set_comm) and - ZF_AHEAD_UNSHARDED - unsharded grids the other scheme does not take.
        const char* e = getenv("ZF_RUNAHEAD");
        const bool on = e ? atoi(e) != 0 : true;
        const char* te = getenv("ZF_TILES_PER_WG");
        const int t = te ? std::max(1, atoi(te)) : zf_tiles_for(s->ntiles);
        const int64_t grid = (s->ntiles + t - 1) / t;
        if (const char* l = getenv("ZF_RUNAHEAD_SPIN_LIMIT")) s->ra_spin = (unsigned)strtoul(l, nullptr, 10);
        const bool chains16 = desc->kind == ZF_PROBLEM_DIAG_QUAD_L1 && s->sub >= 16 && !zf_fin_kernel_mode();
        // (world > 1, or one rank behind a communicator: the SHARDED run-ahead passes - same kernels, packs instead of a decision)
        s->ra = on && chains16;
        if (const char* se = getenv("ZF_RUNAHEAD_SHARDED")) s->ra_sharded = atoi(se) != 0;
        if (s->ra) {
            const zf_trial_sel v = {s->opt.nesterov != 0, s->box, s->nt, s->res};
            s->ra_cap = zf_runahead_capacity(v, s->sub);
            s->ra = grid <= s->ra_cap;
            for (int l = 0; l < ZF_MAX_SUB_ITERS; ++l) s->ra_cap_mid[l] = -1;
        }
        const char* ae = getenv("ZF_AHEAD");
        s->ah = (ae ? atoi(ae) != 0 : true) && chains16 && s->nt;
        const char* ue = getenv("ZF_AHEAD_UNSHARDED");
        s->ah_unsharded = s->ah && desc->world == 1 && (ue ? atoi(ue) != 0 : ZF_AHEAD_UNSHARDED_DEFAULT);
    }
    if (s->ra || (s->ah && (desc->world > 1 || s->ah_unsharded))) {
        s->ring = 6;   // a pass never writes what its predecessor reads (zf_free_bufs)
        // (no second stream - a runtime without stream priorities, an allocation that failed: every pass then runs on the
        //  caller's stream, one launch at a time; nothing else depends on it)
        if (zf_second_stream(s) != hipSuccess) {
            (void)hipGetLastError();
            s->ra = false;
            s->ah = false;
        }
    }
    ZF_TRY(hipMalloc(&s->xbuf, sizeof(double) * s->ring * n_pad));
    for (int k = 0; k < s->ring; ++k) s->xb[k] = s->xbuf + k * n_pad;
    ZF_TRY(hipMalloc(&s->partials, sizeof(double) * ZF_NPART * ZF_MAX_GRID));
    {   // (the small least-squares step kernel has n / 32 workgroups: more than tiles)
        const int64_t parts = std::max<int64_t>(s->max_grid, n / LS_SMALL_COLS + 1);
        ZF_TRY(hipMalloc(&s->blk_part, sizeof(double) * ZF_NPART * s->sub * parts));
    }
    ZF_TRY(hipMalloc(&s->slice_part, sizeof(double) * ZF_NPART * s->sub * ZF_FIN_WGS));
    ZF_TRY(hipMalloc(&s->fin_cnt, sizeof(unsigned) * ZF_FIN_CNT_STRIDE * (ZF_FIN_GROUPS + 2)));
    ZF_TRY(hipMemsetAsync(s->fin_cnt, 0, sizeof(unsigned) * ZF_FIN_CNT_STRIDE * (ZF_FIN_GROUPS + 2), s->stream));
    ZF_TRY(hipMalloc(&s->grp_part, sizeof(double) * ZF_NPART * s->sub * ZF_FIN_GROUPS));
    // the control block and the trace ring side by side, and a pinned host mirror of both: a poll is ONE DMA into
    // pinned memory (two copies into the caller's pageable arrays cost 34 us on an idle stream, this costs a third)
    //  - and, behind them, the log of pass shapes the kernels keep when timing is on)
    ZF_TRY(hipMalloc(&s->ctl_trace, ZF_MAIL_BYTES));
    s->ctl = reinterpret_cast<zf_control*>(s->ctl_trace);
    s->trace = reinterpret_cast<double*>(s->ctl_trace + ZF_CTL_SLOT);
    s->pass_log = reinterpret_cast<int*>(s->ctl_trace + ZF_CTL_SLOT + ZF_TRACE_BYTES);
    s->ra_stats = reinterpret_cast<unsigned*>(s->ctl_trace + ZF_STATS_OFF);
    ZF_TRY(hipMemsetAsync(s->ctl_trace, 0, ZF_CTL_SLOT, s->stream));
    ZF_TRY(hipMemsetAsync(s->pass_log, 0xff, sizeof(int) * ZF_PASS_LOG, s->stream));   // (no launch carries tag 0x7fff + negative sign)
    ZF_TRY(hipHostMalloc(&s->mail, ZF_MAIL_BYTES, hipHostMallocDefault));
    ZF_TRY(hipMalloc(&s->beta_ring, sizeof(double) * ZF_RING));
    ZF_TRY(hipMalloc(&s->pack_local, sizeof(double) * ZF_PACK_LEN * s->sub));
    ZF_TRY(hipMalloc(&s->pack_all, sizeof(double) * ZF_PACK_LEN * s->sub * desc->world));
    ZF_TRY(hipMemsetAsync(s->trace, 0, sizeof(double) * ZF_RING * ZF_TRACE_COLS, s->stream));
    ZF_TRY(hipMemsetAsync(s->beta_ring, 0, sizeof(double) * ZF_RING, s->stream));
    ZF_TRY(hipMemsetAsync(s->pack_all, 0, sizeof(double) * ZF_PACK_LEN * s->sub * desc->world, s->stream));
    if (desc->kind == ZF_PROBLEM_BLUR_HAAR_L1) {
        const int64_t m_pad = n_pad;
        ZF_TRY(hipMalloc(&s->grad, sizeof(double) * n_pad));
        ZF_TRY(hipMalloc(&s->sbuf, sizeof(double) * 3 * m_pad));
        for (int k = 0; k < 3; ++k) s->sring.p[k] = s->sbuf + k * m_pad;
        ZF_TRY(hipMalloc(&s->resid, sizeof(double) * m_pad));
        ZF_TRY(hipMalloc(&s->ls_scal, sizeof(double) * 8));
        if (zf_op_prepare(desc->op_taps, (int)desc->op_k, desc->op_h, desc->op_w, s->stream, &s->op_plan, &s->op_buf, &s->op_taps, &s->op_sep) != ZF_OK) {
            zf_solver_free_all(s);
            delete s;
            return ZF_ERR_HIP;   // (message set by zf_op_prepare)
        }
        ZF_TRY(hipMalloc(&s->row_part, sizeof(double) * 2 * s->op_plan.grid));   // shares of |r(y)|^2 and of |s+ - b|^2
        ZF_TRY(hipMemsetAsync(s->row_part, 0, sizeof(double) * 2 * s->op_plan.grid, s->stream));
        ZF_TRY(hipMalloc(&s->ls_cnt, 64));
        ZF_TRY(hipMemsetAsync(s->ls_cnt, 0, 64, s->stream));
    }
    if (desc->kind == ZF_PROBLEM_LEAST_SQUARES_L1) {
        const int64_t m = desc->m_rows;
        const int64_t m_pad = (m + 63) & ~int64_t(63);
        // enough row slices that panels x slices fills the chip (>= ~2048 workgroups)
        const int V = (n % 2 == 0) ? 2 : 1;
        const int64_t panels = (n / V + ZF_BLOCK - 1) / ZF_BLOCK;
        int64_t slices = (ZF_MAX_GRID + panels - 1) / panels;
        if (slices > (m + 7) / 8) slices = (m + 7) / 8;
        if (slices < 1) slices = 1;
        if (slices > 64) slices = 64;
        s->rows_per_slice = (m + slices - 1) / slices;
        s->slices = (int)((m + s->rows_per_slice - 1) / s->rows_per_slice);
        const char* mf = getenv("ZF_GEMV_MFMA");
        s->gemv_mfma = (n % 32 == 0) && !(mf && atoi(mf) == 0);
        ZF_TRY(hipMalloc(&s->grad, sizeof(double) * n_pad));
        ZF_TRY(hipMalloc(&s->sbuf, sizeof(double) * 3 * m_pad));
        for (int k = 0; k < 3; ++k) s->sring.p[k] = s->sbuf + k * m_pad;
        ZF_TRY(hipMalloc(&s->resid, sizeof(double) * m_pad));
        ZF_TRY(hipMalloc(&s->slab, sizeof(double) * s->slices * n));
        ZF_TRY(hipMalloc(&s->ls_scal, sizeof(double) * 8));
        if (desc->world > 1) {   // exchanged vector: A_p x_p (m) for column blocks, A_p^T r_p (n) for row blocks
            const int64_t len = desc->row_sharded ? n : m;
            ZF_TRY(hipMalloc(&s->s_part, sizeof(double) * ((len + 63) & ~int64_t(63))));
            ZF_TRY(hipMalloc(&s->s_all, sizeof(double) * len * desc->world));
        }
        // launch-bound sizes (BASELINE cfg1): two fused launches per trial; ZF_LS_SMALL=0 keeps the general path
        const char* sm = getenv("ZF_LS_SMALL");
        s->ls_small = desc->world == 1 && n % LS_SMALL_COLS == 0 && m <= LS_SMALL_MAX_M &&
                      m * n <= LS_SMALL_MAX_ELEMS && zf_aligned16(desc->A) && !(sm && atoi(sm) == 0);
        if (s->ls_small) {
            ZF_TRY(hipMalloc(&s->row_part, sizeof(double) * ((m + ZF_WAVES - 1) / ZF_WAVES)));
            ZF_TRY(hipMalloc(&s->ls_cnt, 64));
            ZF_TRY(hipMemsetAsync(s->ls_cnt, 0, 64, s->stream));
        }
    }
#undef ZF_TRY
    *out = s;
    return ZF_OK;
}

extern "C" int zf_solver_destroy(zf_solver* s) {
    if (!s) return ZF_OK;
    (void)hipStreamSynchronize(s->stream);
    zf_solver_free_all(s);
    delete s;
    return ZF_OK;
}

// ---- launches ---------------------------------------------------------------
// Bits of zf_solver::part_mask: which shape-specific kernels a step launches (zf_trial_kernel, PART).  Each exits at
// once unless the pass has its shape (zf_pass_claims).
enum {
    ZF_K_FULL = 1,       // PART 0: the full chain
    ZF_K_SHORT = 2,      // PART 1: up to S / 2 fresh trials behind the lagging iterations, materialise-only passes
    ZF_K_GENERAL = 4,    // PART 2 (S = 16): the general body on the shapes of its own (more than S / 2 fresh trials)
    ZF_K_MID = 8,        // PART 3 (S = 16): the branch-free mid chain of s->mid_len trials
    ZF_K_FALLBACK = 16,  // PART 2 as the fallback: every shape the kernel named by s->fb_part / fb_len does NOT run (-1: every shape)
    ZF_K_ALL = ZF_K_FULL | ZF_K_SHORT | ZF_K_GENERAL,   // nothing is known: between them these three run every shape
};

static void zf_launch_trial_kernels(zf_solver* s, const zf_step_args& a, bool grad_inline) {
    const zf_trial_sel v = {s->opt.nesterov != 0, s->box, s->nt, s->res};
    const int mask = s->part_mask;
    const int grid = s->grid;
    hipStream_t st = s->stream;
    if (!grad_inline) {   // least squares: one trial per pass
        if (s->hist) zf_launch_hist(v, false, 1, 0, grid, st, a);
        else zf_launch_vec(v, grid, st, a);
        return;
    }
    if (s->hist) {   // (zf_solver_set_history allowed only chain lengths 1 and 8)
        const int S = s->sub == 8 ? 8 : 1;
        if (mask & ZF_K_FULL) zf_launch_hist(v, true, S, 0, grid, st, a);
        if (S > 1 && (mask & ZF_K_SHORT)) zf_launch_hist(v, true, S, 1, grid, st, a);
        return;
    }
    if (s->sub >= 16) {
        if (mask & ZF_K_FULL) zf_launch_s16_full(v, grid, st, a);
        if (mask & ZF_K_SHORT) zf_launch_s16_short(v, grid, st, a);
        if (mask & ZF_K_MID) zf_launch_s16_mid(v, s->mid_len, grid, st, a);   // (zf_predict_parts checked that it exists)
        if (mask & ZF_K_GENERAL) zf_launch_s16_general(v, grid, st, a);
        if (mask & ZF_K_FALLBACK) {
            zf_step_args f = a;
            f.fb_on = 1;
            f.fb_part = s->fb_part;
            f.fb_len = s->fb_len;
            zf_launch_s16_general(v, grid, st, f);
        }
        return;
    }
    if (mask & ZF_K_FULL) zf_launch_chain(v, s->sub, 0, grid, st, a);
    if (s->sub > 1 && (mask & ZF_K_SHORT)) zf_launch_chain(v, s->sub, 1, grid, st, a);
}

// second launch of a step: partials -> pack (+ decide when x is unsharded)
static void zf_launch_finalize(zf_solver* s, bool decide) {
    const zf_problem_desc& d = s->desc;
    zf_finalize_args F;
    F.blk_part = s->blk_part;
    F.nblocks = s->grid;
    F.sub_iters = s->sub;
    F.slice_part = s->slice_part;
    F.cnt = s->fin_cnt;
    for (int k = 0; k < ZF_NPART; ++k) F.scale[k] = 1.0;
    F.scale[3] = d.lam;               // g = lam * sum|x|
    F.f_y_ext = nullptr;
    F.f_x_ext = nullptr;
    F.contribute_f = 1;
    F.contribute_x = 1;
    if (d.kind == ZF_PROBLEM_DIAG_QUAD_L1) {
        F.scale[0] = 0.5;             // f = 0.5 * sum(d (x-c)^2)
        F.scale[4] = 0.5;
    } else {
        F.f_y_ext = s->ls_scal + 0;   // f(y), f(x+) come from the GEMV side and are replicated
        F.f_x_ext = s->ls_scal + 1;
        // column blocks: f is replicated, the x sums are partial; row blocks: the other way round
        F.contribute_f = (d.world == 1 || d.rank == 0 || d.row_sharded) ? 1 : 0;
        F.contribute_x = (d.world == 1 || d.rank == 0 || !d.row_sharded) ? 1 : 0;
    }
    F.pack = s->pack_local;
    F.ctl = s->ctl;
    F.decide = decide ? 1 : 0;
    F.trace = s->trace;
    F.beta_ring = s->beta_ring;
    int wgs = (s->grid + ZF_FIN_THREADS - 1) / ZF_FIN_THREADS;   // no more workgroups than slices of work
    // up to 1024 rows of partials one workgroup (four rows per thread) does it alone: publishing slice sums,
    // the ticket and the last arriver's gather are three dependent round trips to memory (~7 us of a ~20 us kernel)
    if (s->grid <= 1024) wgs = 1;
    if (wgs > ZF_FIN_WGS) wgs = ZF_FIN_WGS;
    if (wgs < 1) wgs = 1;
    if (s->sub == 16) hipLaunchKernelGGL(zf_finalize_kernel<16>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, s->stream, F);
    else if (s->sub == 8) hipLaunchKernelGGL(zf_finalize_kernel<8>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, s->stream, F);
    else if (s->sub == 4) hipLaunchKernelGGL(zf_finalize_kernel<4>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, s->stream, F);
    else if (s->sub == 2) hipLaunchKernelGGL(zf_finalize_kernel<2>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, s->stream, F);
    else hipLaunchKernelGGL(zf_finalize_kernel<1>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, s->stream, F);
}

// the arguments every trial launch shares; the in-kernel finalisation off (fin_mode 0)
static void zf_fill_step_args(const zf_solver* s, zf_step_args& a) {
    const zf_problem_desc& d = s->desc;
    memset(&a, 0, sizeof(a));
    a.ctl = s->ctl;
    a.beta_ring = s->beta_ring;
    for (int k = 0; k < ZF_MAX_RING; ++k) a.xb[k] = s->xb[k < s->ring ? k : 0];
    a.lam = d.lam;
    a.lo = d.box_lo;
    a.hi = d.box_hi;
    a.n = d.n;
    a.tiles_per_wg = s->tiles;
    a.blk_part = s->blk_part;
    a.pass_log = nullptr;
    a.pass_tag = 0;
    a.pass_slot = 0;
    a.hist = s->hist;
    a.hist_cap = s->hist_cap > 0 ? s->hist_cap : 1;
    a.hist_stride = s->hist_stride;
    a.fin_mode = 0;
    a.accept_mode = s->res ? ZF_ACCEPT_RESOLVED : ZF_ACCEPT_REFERENCE;
}

// Groups of the in-kernel finalisation: up to ZF_FIN_GROUPS workgroups reduce as ONE level (every workgroup its
// own "group"); beyond that ceil(grid / ZF_FIN_GROUPS) consecutive workgroups form a group.  A function of the
// grid only.
static void zf_fin_groups(int grid, int* gsz, int* ng) {
    if (grid <= ZF_FIN_GROUPS) {
        *gsz = 1;
        *ng = grid;
        return;
    }
    *gsz = (grid + ZF_FIN_GROUPS - 1) / ZF_FIN_GROUPS;
    *ng = (grid + *gsz - 1) / *gsz;
}

static bool zf_fin_kernel_mode() {
    static const bool on = [] { const char* e = getenv("ZF_FIN_KERNEL"); return e && atoi(e) != 0; }();
    return on;
}

// grids of at most this many workgroups are latency-bound: a second (idle) launch per pass costs more than a slower body
constexpr int ZF_SMALL_GRID = 64;
constexpr int64_t ZF_OP_FUSE_MAX_PIXELS = int64_t(5) << 20;   // operator problem: the prox step rides in the adjoint kernel up to this size
// tiles per workgroup (a function of n: zf_tiles_for) between which a run of run-ahead passes may START with a mid chain
constexpr int ZF_RA_MID_START_MIN_TILES = 4;    // n >= ~4e6
constexpr int ZF_RA_MID_START_MAX_TILES = 12;   // n <= ~1.2e7

// move the shadow control block past a pass on the assumption that every fresh trial is accepted and nothing
// terminates but max_iter - true for whole chunks in the regime a line search settles in
static void zf_shadow_advance(zf_control& c) {
    const int ring = c.ring_size > 0 ? c.ring_size : 3;
    if (c.pend_status != 0) {
        if (c.lag > 0) zf_commit_chain(&c, c.cur, c.prev, ring, c.lag);
        c.lag = 0;
        if (c.pend_status > 0) c.status = c.pend_status;
        c.pend_status = 0;
        return;
    }
    const int nf = zf_fresh_len(&c);
    c.nit += nf;
    c.total_trials += nf;   // (every trial accepted: zf_pack_stamp of a pass ahead is computed from the shadow)
    if (c.lag + nf > 0) zf_commit_chain(&c, c.cur, c.prev, ring, c.lag + nf);   // (the buffers too: a run-ahead pass is told where to read)
    c.lag = 0;
    if (c.nit >= c.max_iter) c.status = ZF_MAXITER;
}

// The kernels of the next step (ZF_K_* bits; s->mid_len, s->fb_part, s->fb_len beside them), and the shadow control
// block moved past that pass.  The shape rule is the kernel's own (zf_pass_part, zf_fresh_len).
// * nothing known (no poll yet, a host-driven exchange, ZF_SPECULATE=0): the three kernels that between them run every shape;
// * the shadow is what the device holds, or what it holds if no chain broke since the poll: the ONE kernel of that shape;
// * the last chunk saw rejections ("careful"): the first step after the poll is still exact; behind it the optimistic
//   prediction plus the general body as its complement (whatever shape the pass has, exactly one of the two runs it)
//   - on small grids, where an idle launch costs more than the general body loses, the general body alone;
// * the shadow says the solve is over (and the device may not be: a chain broke): every shape again.
// A wrong prediction costs passes that do nothing (no kernel finds its shape, the control block stays as it is) until
// the next poll - never a wrong result.
static int zf_predict_parts(zf_solver* s) {
    const bool off = !s->speculate;   // (ZF_SPECULATE=0 when the solver was created)
    // (x sharded: only with the library's communicator, whose decide step checks that every rank's packs are those
    //  of this step - all ranks predict from identical control blocks; a host-driven exchange launches everything)
    if (off || s->sub <= 1 || !s->shadow_valid || (s->desc.world != 1 && !s->comm) ||
        s->desc.kind != ZF_PROBLEM_DIAG_QUAD_L1 || zf_fin_kernel_mode())
        return ZF_K_ALL;
    zf_control& c = s->shadow;
    const int S = s->sub;
    const bool s16 = S >= 16;
    if (c.status != ZF_RUNNING) {
        // The shadow has reached the end of the solve.  If the device has not - chains broke; in a solve that crosses
        // the resolution limit of the acceptance test MOST steps are issued in this state, the optimistic shadow being
        // done after max_iter / S of them - any shape may be due: the three kernels that between them run every shape,
        // each on the shapes it runs best (the general body alone took these solves from 5 880 to 4 900 it/s at n = 1e8:
        // it runs the replaying shapes at 1.65 ms against 1.28).  Grids of a few workgroups: the general body alone.
        if (!s16) return ZF_K_ALL;
        if (s->grid > ZF_SMALL_GRID) {   // PART 1 on its shapes, the general body on every other one: two launches, one of them runs
            s->fb_part = 1;
            s->fb_len = 0;
            return ZF_K_SHORT | ZF_K_FALLBACK;
        }
        s->fb_part = -1;
        return ZF_K_FALLBACK;
    }
    const int lag = c.lag;
    const int nf = zf_fresh_len(&c);
    int part = zf_pass_part(S, lag, nf);
    const zf_trial_sel v = {s->opt.nesterov != 0, s->box, s->nt, s->res};
    if (part == 3 && (!s->mid_chains || !zf_have_s16_mid(v, nf))) part = 2;
    int mask = part == 3 ? ZF_K_MID : (1 << part);
    s->mid_len = nf;
    if (part == 1 && s16 && s->short_general) {
        s->fb_part = -1;
        mask = ZF_K_FALLBACK;
    }
    if (s->careful && s->steps_since_poll > 0) {
        // Chains have been breaking: behind the first step the shape is no longer known.  A pass behind a broken chain
        // replays and has at most S / 2 fresh trials, or only materialises (PART 1); a pass with nothing lagging far
        // from max_iter is a full chain (PART 0); near max_iter it may be any length (PART 2 as well).  The general
        // body could run all of them alone - one launch, never idle - but it runs the replaying shapes at 1.65 ms
        // where PART 1 needs 1.28 (n = 1e8, tools/r4_noise.sh): only grids of a few workgroups, where a launch costs
        // more than a body, take it.
        const int64_t left = c.max_iter - c.nit;
        if (!s16) {
            mask = ZF_K_ALL;
        } else if (s->grid <= ZF_SMALL_GRID) {
            s->fb_part = -1;
            mask = ZF_K_FALLBACK;
        } else {
            if (left >= 2 * (int64_t)S) {
                mask = ZF_K_FULL | ZF_K_SHORT;
            } else {   // near max_iter any length may be due: PART 1 on its shapes, the general body on all others
                s->fb_part = 1;
                s->fb_len = 0;
                mask = ZF_K_SHORT | ZF_K_FALLBACK;
            }
        }
    }
    zf_shadow_advance(c);
    return mask;
}

// `stream` waits for what stream2 holds; the next run-ahead pass starts a new run (it reads the real control block)
static int zf_ra_join(zf_solver* s) {
    if (s->ra_b_pending) {
        ZF_HIP(hipEventRecord(s->ra_join, s->stream2));
        ZF_HIP(hipStreamWaitEvent(s->stream, s->ra_join, 0));
        s->ra_b_pending = false;
    }
    if (s->ra_c_pending) {
        ZF_HIP(hipEventRecord(s->ra_join3, s->stream3));
        ZF_HIP(hipStreamWaitEvent(s->stream, s->ra_join3, 0));
        s->ra_c_pending = false;
    }
    s->ra_last = s->ra_last2 = 0;
    s->run_mode = 0;
    return ZF_OK;
}

// A full chain the shadow predicts exactly, as a run-ahead pass (zf_runahead_kernel): behind another one of the same
// run it goes to the other stream and starts while that one is still finalising.  `before`: the shadow control block
// in front of this pass.
static int zf_gather_packs(zf_solver* s, int64_t packs, hipStream_t st, const double* src = nullptr);

// SHARDED (s->comm; round 5): the same kernels, the same flags between workgroups of consecutive passes - but the pass ends
// with its packs (zf_pass_tail: decide == 0), and the all-gather + zf_decide_ahead_kernel of pass p run on a THIRD stream behind
// an event, beside the workgroups of pass p + 1:
//   stream  : T(0)            | [wait decided(0)] T(2) ...
//   stream2 :    T(1) (workgroup j behind workgroup j of T(0))  | [wait decided(1)] T(3) ...
//   stream3 : [wait T(0)] gather, decide(0) | [wait T(1)] gather, decide(1) | ...
// The decide step checks the block against the head the pass ran on (a pass on a head that did not come true is void) and
// publishes the done / good word the kernels read.  What a sharded one-round grid gets out of it is what the unsharded one
// gets: the finalisation, the exchange and the decision of a pass no longer lie between two passes.
static int zf_launch_runahead(zf_solver* s, zf_step_args a, const zf_control& before, int nf, hipEvent_t e0, hipEvent_t e1) {
    const zf_trial_sel v = {s->opt.nesterov != 0, s->box, s->nt, s->res};
    const bool sharded = s->comm != nullptr;
    const int mode = sharded ? 3 : 1;
    zf_pass_head h;
    h.cur = before.cur;
    h.prev = before.prev;
    h.ring = before.ring_size;
    h.lr = before.lr;
    h.beta_next = 0.0;
    h.nit = before.nit;
    bool chain = s->run_mode == mode && s->ra_last != 0 && a.pass_seq > s->ra_last;
    if (chain) {   // what this pass writes, the pass in flight must not be reading (six buffers in ring order: it never is)
        int f0, f1;
        zf_free_bufs(h.cur, h.prev, h.ring, &f0, &f1);
        const zf_pass_head& q = s->ra_last_head;
        if (f0 == q.cur || f0 == q.prev || f1 == q.cur || f1 == q.prev || h.nit != q.nit + s->ah_last_nf) chain = false;
    }
    int idx = 0;
    if (!chain) {
        int rc = zf_ra_join(s);
        if (rc) return rc;
        // The second stream may go on once the solver's stream has reached this point - NOT earlier: were a per-pass
        // kernel still running in front of this pass, the pass behind it would fill the CUs with waiting workgroups
        // before this one is dispatched, and this one would trickle through the slots they leave (not measured: ruled
        // out by construction).  Recorded before the launch, so the two passes still start together.
        ZF_HIP(hipEventRecord(s->ra_fork, s->stream));
        s->ra_fork_due = true;   // (waited for when - if - a pass of this run goes to the second stream)
        s->ah_run = 0;
    } else {
        idx = 1 - s->ra_last_idx;
        if (idx == 1 && s->ra_fork_due) {
            ZF_HIP(hipStreamWaitEvent(s->stream2, s->ra_fork, 0));
            s->ra_fork_due = false;
        }
    }
    const int k = s->ah_run;   // (number of the pass within its run: the events of a sharded run go by k % 4)
    a.ra_word = s->ra_word;
    a.ra_flags = s->ra_flags;
    a.ra_stats = s->ra_stats;
    a.ra_wait = chain ? s->ra_last : 0;
    a.ra_need = chain ? s->ra_last2 : 0;
    a.ra_spin = s->ra_spin;
    a.ra_head = h;
    if (idx == 1) {
        a.blk_part = s->blk_part2;
        a.grp_part = s->grp_part2;
        a.fin_cnt = s->fin_cnt2;
        a.pack = s->pack2;
    }
    hipStream_t st = idx == 1 ? s->stream2 : s->stream;
    if (sharded) {
        a.decide = 0;
        a.head_nf = nf;
        a.head_stamp = zf_pack_stamp(&before);
        // pass k writes what pass k - 2 (same stream) read, and reuses its rows and packs: that pass must have been DECIDED
        // (third stream) - and as expected, which the kernel checks (ra_need)
        if (k >= 2) ZF_HIP(hipStreamWaitEvent(st, s->ah_evD[(k - 2) & 3], 0));
    }
    if (e0) ZF_HIP(hipEventRecord(e0, st));
    if (!zf_launch_s16_runahead(v, nf, s->grid, st, a)) return zf_fail(ZF_ERR_STATE, "zf_launch_runahead: no run-ahead kernel of that length%s");
    if (e1) ZF_HIP(hipEventRecord(e1, st));
    if (sharded) {
        // exchange and decide step: on the third stream behind an event - but for the LAST pass of a chunk, which nothing
        // runs beside: on the pass's own stream (one hop between queues, ~20 us, less in front of the poll; the decide steps stay
        // in order through the event of the one before, long signalled by then)
        static const bool last_inline = [] { const char* e = getenv("ZF_RAS_LAST_INLINE"); return e ? atoi(e) != 0 : true; }();
        hipStream_t ds = s->stream3;
        if (s->chunk_last && last_inline) {
            ds = st;
            if (k >= 1) ZF_HIP(hipStreamWaitEvent(ds, s->ah_evD[(k - 1) & 3], 0));
        } else if (e1) {
            ZF_HIP(hipStreamWaitEvent(ds, e1, 0));
        } else {
            ZF_HIP(hipEventRecord(s->ah_evT[k & 3], st));
            ZF_HIP(hipStreamWaitEvent(ds, s->ah_evT[k & 3], 0));
        }
        int rc = zf_gather_packs(s, s->sub, ds, a.pack);
        if (rc) return rc;
        zf_ahead_check H;
        H.head = h;
        H.nf = nf;
        H.seq = a.pass_seq;
        hipLaunchKernelGGL(zf_decide_ahead_kernel, dim3(1), dim3(64), 0, ds, s->ctl, s->pack_all, s->trace, s->beta_ring, s->sub, H,
                           s->ra_word, s->ra_stats, a.pass_log, a.pass_slot, a.pass_tag);
        ZF_HIP(hipEventRecord(s->ah_evD[k & 3], ds));
        if (ds == s->stream3) s->ra_c_pending = true;
        s->ras_passes += 1;
    }
    s->run_mode = mode;
    s->ah_run = k + 1;
    s->ah_last_nf = nf;
    s->ra_last2 = chain ? s->ra_last : 0;
    s->ra_last = a.pass_seq;
    s->ra_last_idx = idx;
    s->ra_last_head = h;
    if (idx == 1) s->ra_b_pending = true;
    s->ra_passes += 1;
    if (chain) s->ra_ahead += 1;
    return ZF_OK;
}


// A full or mid chain the shadow predicts exactly, as a pass AHEAD of its predecessor's decision at kernel granularity:
//   stream  : ... T(p) | T(p+1) | [wait decided(p)] T(p+2) ...        trial kernels back to back, each on the head the
//                                                                     host expects (zf_trial_kernel<..., AHEAD>)
//   stream2 : [wait T(p) done] rows -> packs (-> all-gather) -> decide(p) | ...   beside T(p+1)
// T(p+2) writes the buffers T(p) read; it is launched behind the event of decide(p) and leaves at once unless pass p went
// as expected (ra_need).  decide(p) checks the control block against the head T(p) ran on: a pass on a head that did not
// come true is VOID (zf_decide_ahead).  Nothing waits inside a kernel: every grid size, every sharing of the device.
// `before`: the shadow control block in front of this pass; nf: its fresh trials; part: 0 full chain, 3 mid chain.
static int zf_launch_ahead(zf_solver* s, zf_step_args a, const zf_control& before, int part, int nf, hipEvent_t e0, hipEvent_t e1) {
    const zf_trial_sel v = {s->opt.nesterov != 0, s->box, s->nt, s->res};
    zf_pass_head h;
    h.cur = before.cur;
    h.prev = before.prev;
    h.ring = before.ring_size;
    h.lr = before.lr;
    h.beta_next = 0.0;
    h.nit = before.nit;
    bool chain = s->run_mode == 2 && s->ra_last != 0 && a.pass_seq > s->ra_last;
    if (chain) {   // what this pass writes, the pass before it must not be reading (six buffers in ring order: it never is)
        int f0, f1;
        zf_free_bufs(h.cur, h.prev, h.ring, &f0, &f1);
        const zf_pass_head& q = s->ra_last_head;
        if (f0 == q.cur || f0 == q.prev || f1 == q.cur || f1 == q.prev || h.nit != q.nit + s->ah_last_nf) chain = false;
    }
    if (!chain) {
        int rc = zf_ra_join(s);   // (the solver's stream now comes behind every decide step so far: the block is what this head was read from)
        if (rc) return rc;
        s->ah_run = 0;
    }
    const int k = s->ah_run;
    // the pass two before this one has been decided (its inputs are this pass's outputs): kernel-granularity dependency
    if (k >= 2) ZF_HIP(hipStreamWaitEvent(s->stream, s->ah_evD[(k - 2) & 3], 0));
    a.ra_word = s->ra_word;
    a.ra_stats = s->ra_stats;
    a.ra_need = chain ? s->ra_last2 : 0;
    a.ra_wait = chain ? s->ra_last : 0;   // (0: the first pass of a run - its kernel checks the block against the head)
    a.ra_head = h;
    a.head_nf = nf;
    a.head_stamp = zf_pack_stamp(&before);
    a.fin_mode = 2;
    a.fin_grid = s->grid;
    a.blk_part = (k & 1) ? s->blk_part2 : s->blk_part;   // rows of pass p are read on stream2 while T(p+1) writes its own
    if (e0) ZF_HIP(hipEventRecord(e0, s->stream));
    if (part == 0) zf_launch_s16_ahead_full(v, s->grid, s->stream, a);
    else if (!zf_launch_s16_ahead_mid(v, nf, s->grid, s->stream, a)) return zf_fail(ZF_ERR_STATE, "zf_launch_ahead: no mid chain of that length%s");
    // "trial kernel p is done": the timing mode's closing event when there is one (a wait captures what the event holds NOW) -
    // every event packet between two trial kernels costs the chain 3-6 us (profiles/r05_event_probe.txt)
    if (e1) {
        ZF_HIP(hipEventRecord(e1, s->stream));
        ZF_HIP(hipStreamWaitEvent(s->stream2, e1, 0));
    } else {
        ZF_HIP(hipEventRecord(s->ah_evT[k & 3], s->stream));
        ZF_HIP(hipStreamWaitEvent(s->stream2, s->ah_evT[k & 3], 0));
    }
    // rows -> packs; unsharded: the deciding wave of the same launch decides
    a.decide = s->comm ? 0 : 1;
    zf_launch_s16_tail(s->stream2, a);
    if (s->comm) {
        int rc = zf_gather_packs(s, s->sub, s->stream2);
        if (rc) return rc;
        zf_ahead_check H;
        H.head = h;
        H.nf = nf;
        H.seq = a.pass_seq;
        hipLaunchKernelGGL(zf_decide_ahead_kernel, dim3(1), dim3(64), 0, s->stream2, s->ctl, s->pack_all, s->trace, s->beta_ring, s->sub, H,
                           s->ra_word, s->ra_stats, a.pass_log, a.pass_slot, a.pass_tag);
    }
    ZF_HIP(hipEventRecord(s->ah_evD[k & 3], s->stream2));
    s->ra_b_pending = true;
    s->run_mode = 2;
    s->ra_last2 = chain ? s->ra_last : 0;
    s->ra_last = a.pass_seq;
    s->ra_last_head = h;
    s->ah_last_nf = nf;
    s->ah_run = k + 1;
    s->ah_passes += 1;
    return ZF_OK;
}

// returns in *done_ahead (if given) whether the step was a pass ahead: tail, exchange and decide are then enqueued too
static int zf_launch_trial(zf_solver* s, bool decide_in_launch, bool dry = false, bool* done_ahead = nullptr) {
    const zf_problem_desc& d = s->desc;
    zf_step_args a;
    zf_fill_step_args(s, a);
    if (done_ahead) *done_ahead = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->timing && !dry) {
        a.pass_log = s->pass_log;
        a.pass_slot = (int)(s->launches % ZF_PASS_LOG);
        a.pass_tag = (int)(s->launches & 0x7fff) << 16;
        s->launches += 1;
    }
    if (s->timing) {
        if (s->ev_used == s->ev_pool.size()) {
            hipEvent_t x, y;
            ZF_HIP(hipEventCreate(&x));
            ZF_HIP(hipEventCreate(&y));
            s->ev_pool.emplace_back(x, y);
        }
        e0 = s->ev_pool[s->ev_used].first;
        e1 = s->ev_pool[s->ev_used].second;
        s->ev_used++;
    }
    if (d.kind == ZF_PROBLEM_DIAG_QUAD_L1) {
        a.p0 = d.d;
        a.p1 = d.c;
        // (ZF_FIN_KERNEL=1: the round-2 sequence - plain rows, a separate zf_finalize_kernel launch - for A/B
        //  measurements on one box; its sums are added in another order, so knife-edge decisions may differ)
        const bool fin_kernel = zf_fin_kernel_mode();
        if (!dry && !fin_kernel) {
            // the pass finalises itself (zf_pass_tail): packs, and - unsharded - the decide pass, in the same launch
            a.fin_mode = 1;
            zf_fin_groups(s->grid, &a.fin_gsz, &a.fin_ng);
            a.grp_part = s->grp_part;
            a.fin_cnt = s->fin_cnt;
            a.fin_scale_f = 0.5;   // f = 0.5 * sum(d (x-c)^2)
            a.fin_scale_g = d.lam; // g = lam * sum|x|
            a.pack = s->pack_local;
            a.ctl_rw = s->ctl;
            a.decide = (d.world == 1 && decide_in_launch) ? 1 : 0;
            a.trace = s->trace;
            s->pass_seq = s->pass_seq >= 0x7ffffff0 ? 1 : s->pass_seq + 1;
            a.pass_seq = s->pass_seq;
        }
        zf_control before;
        // passes that run ahead of their predecessor's decision: at workgroup granularity (unsharded one-round grids), or
        // at kernel granularity (through the library's communicator; ZF_AHEAD_UNSHARDED: other unsharded grids)
        const bool two_streams = s->stream2 != nullptr && !dry && !fin_kernel && !s->hist && s->shadow_valid;
        // (behind a communicator only from zf_solver_enqueue_steps, which hands over done_ahead: a caller that drives trial /
        //  exchange / decide itself - zf_solver_enqueue_trial - gets one pass at a time, whatever is attached)
        const bool ra_can = two_streams && s->ra && !s->ra_off &&
                            (s->comm ? (!decide_in_launch && done_ahead != nullptr && s->ra_sharded && s->stream3 != nullptr) : decide_in_launch);
        // (unsharded: what the run-ahead kernel does not take - grids of several rounds, clipped problems, and on its own
        //  grids the mid chains of a shared tail)
        const bool ah_can = two_streams && s->ah && s->ring >= 6 &&
                            (s->comm ? (!decide_in_launch && done_ahead != nullptr) : (decide_in_launch && s->ah_unsharded && d.world == 1));
        const bool have_before = ra_can || ah_can;
        if (have_before) before = s->shadow;
        s->part_mask = (dry || !(decide_in_launch || s->comm)) ? ZF_K_ALL : zf_predict_parts(s);
        // the ONE kernel the shadow predicts (not a pair behind a chunk that saw rejections), nothing lagging
        const bool exact = have_before && before.status == ZF_RUNNING && before.lag == 0 && before.pend_status == 0 &&
                           before.ring_size == s->ring;
        const int nf_before = exact ? zf_fresh_len(&before) : 0;
        bool ra_ok = ra_can && exact && s->part_mask == ZF_K_FULL && nf_before == s->sub && s->grid <= s->ra_cap;
        // A mid chain (the passes of a tail shared by two passes, the tail itself): its own kernel, its own capacity.  Behind a
        // pass of the same run it always runs ahead (the alternative joins the two streams first).  A run that STARTS with
        // one - blocks of K = 20: 10 + 10 - pays only in the middle of the one-round sizes (profiles/r05_runahead_mid_k20_by_size.jsonl:
        // n = 4e6 .. 1e7 +2-3 %, with bench.py's events +4-7 %): below, two launches on one stream cost less than the fork and
        // join of two; above, the chain of <= 10 trials is HBM-bound and its per-pass kernel loads through registers, 4 % faster
        // than the DMA pipeline the coherent loads need.
        const bool mid_run = (s->run_mode == 1 || s->run_mode == 3) && s->ra_last != 0;
        if (ra_can && exact && !ra_ok && s->part_mask == ZF_K_MID && nf_before == s->mid_len && nf_before < ZF_MAX_SUB_ITERS &&
            (mid_run || (s->tiles >= ZF_RA_MID_START_MIN_TILES && s->tiles <= ZF_RA_MID_START_MAX_TILES))) {
            int& cap = s->ra_cap_mid[nf_before];
            if (cap < 0) cap = zf_runahead_capacity(zf_trial_sel{s->opt.nesterov != 0, s->box, s->nt, s->res}, nf_before);
            ra_ok = cap > 0 && s->grid <= cap;
        }
        const bool ah_ok = ah_can && !ra_ok && exact && ((s->part_mask == ZF_K_FULL && nf_before == s->sub) ||
                                                          (s->part_mask == ZF_K_MID && nf_before == s->mid_len));
        if (s->stream2 && a.pass_seq == 1 && !dry) {   // (the step counter started or wrapped: sequence numbers are compared)
            int rc = zf_ra_join(s);
            if (rc) return rc;
            ZF_HIP(hipMemsetAsync(s->ra_word, 0, 128, s->stream));
            ZF_HIP(hipMemsetAsync(s->ra_flags, 0, sizeof(unsigned) * (s->max_grid + 32), s->stream));
            // (the second stream must not find the flags of the numbers before the wrap: they satisfy every wait)
            ZF_HIP(hipEventRecord(s->ra_fork, s->stream));
            ZF_HIP(hipStreamWaitEvent(s->stream2, s->ra_fork, 0));
        }
        if (!dry) {
            s->steps_since_poll += 1;
            const int shapes = s->part_mask & (s->sub >= 16 ? 31 : s->sub > 1 ? 3 : 1);
            s->steps_issued += 1;
            s->kernels_issued += __builtin_popcount(shapes);
        }
        if (ra_ok) {
            int rc = zf_launch_runahead(s, a, before, nf_before, e0, e1);
            if (rc) return rc;
            if (s->comm && done_ahead) *done_ahead = true;   // (exchange and decide step are enqueued: third stream)
        } else if (ah_ok) {
            int rc = zf_launch_ahead(s, a, before, s->part_mask == ZF_K_FULL ? 0 : 3, nf_before, e0, e1);
            if (rc) return rc;
            if (done_ahead) *done_ahead = true;
        } else {
            if (s->stream2) {
                int rc = zf_ra_join(s);
                if (rc) return rc;
            }
            if (e0) ZF_HIP(hipEventRecord(e0, s->stream));
            zf_launch_trial_kernels(s, a, true);
            if (e1) ZF_HIP(hipEventRecord(e1, s->stream));
        }
        s->part_mask = ZF_K_ALL;
        if (!dry && fin_kernel) zf_launch_finalize(s, d.world == 1 && decide_in_launch);
    } else if (s->ls_small && !dry && decide_in_launch) {
        // cache-resident A: the whole trial in two launches (zf_kernels_ls_small.h)
        zf_ls_small_args P;
        P.ctl = s->ctl;
        P.beta_ring = s->beta_ring;
        for (int k = 0; k < 3; ++k) P.xb[k] = s->xb[k];
        P.sring = s->sring;
        P.A = d.A;
        P.b = d.b;
        P.m = d.m_rows;
        P.n = d.n;
        P.scale = d.scale;
        P.lam = d.lam;
        P.lo = d.box_lo;
        P.hi = d.box_hi;
        P.ls_scal = s->ls_scal;
        P.blk_part = s->blk_part;
        P.grid_step = (int)(d.n / LS_SMALL_COLS);
        P.row_part = s->row_part;
        P.cnt = s->ls_cnt;
        P.pack = s->pack_local;
        P.trace = s->trace;
        P.hist = s->hist;
        P.hist_cap = s->hist_cap > 0 ? s->hist_cap : 1;
        P.hist_stride = s->hist_stride;
        P.pass_log = a.pass_log;
        P.pass_slot = a.pass_slot;
        P.pass_tag = a.pass_tag;
        const bool nest = s->opt.nesterov != 0;
        dim3 gs(P.grid_step), gr((unsigned)((d.m_rows + ZF_WAVES - 1) / ZF_WAVES)), b(ZF_BLOCK);
        if (e0) ZF_HIP(hipEventRecord(e0, s->stream));
        if (nest && s->box) hipLaunchKernelGGL((zf_ls_small_step_kernel<true, true>), gs, b, 0, s->stream, P);
        else if (nest) hipLaunchKernelGGL((zf_ls_small_step_kernel<true, false>), gs, b, 0, s->stream, P);
        else if (s->box) hipLaunchKernelGGL((zf_ls_small_step_kernel<false, true>), gs, b, 0, s->stream, P);
        else hipLaunchKernelGGL((zf_ls_small_step_kernel<false, false>), gs, b, 0, s->stream, P);
        if (e1) ZF_HIP(hipEventRecord(e1, s->stream));
        hipLaunchKernelGGL(zf_ls_small_rows_kernel, gr, b, 0, s->stream, P);
    } else {
        const int64_t n = d.n, m = d.m_rows;
        const int V = (n % 2 == 0) ? 2 : 1;
        // (1) r = A y - b by linearity, f(y); grad = 2 scale A^T r   [only when y changed]
        if (d.kind != ZF_PROBLEM_BLUR_HAAR_L1)   // (the operator problem forms r inside its adjoint kernel)
            hipLaunchKernelGGL(zf_resid_y_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, s->ctl,
                               s->beta_ring, s->sring, d.b, s->resid, d.scale, m, s->ls_scal + 0,
                               (int)s->opt.nesterov);
        const int64_t nv = n / V;
        dim3 gT((unsigned)((nv + ZF_BLOCK - 1) / ZF_BLOCK), (unsigned)s->slices);
        if (d.kind == ZF_PROBLEM_BLUR_HAAR_L1) {
            // TWO launches per trial (unsharded, deciding in the launch): the adjoint kernel - the residual at y formed in its
            // tile load, the PROX STEP in its epilogue (round 5; ZF_OP_FUSE_PROX=0 or a history ring: a launch of its own
            // in between) - and s+ = B W^-1 x+ whose last workgroup sums f(y), f(x+) and the prox step's partials and decides
            // (zf_op_fuse) - instead of resid_y / adjoint / prox / apply / resid_x / finalize
            zf_op_fuse F;
            memset(&F, 0, sizeof(F));
            F.on = 1;
            F.b = d.b;
            for (int k = 0; k < 3; ++k) F.sk[k] = s->sring.p[k];
            F.scale = d.scale;
            F.lam = d.lam;
            F.nesterov = s->opt.nesterov;
            F.part_y = s->row_part;
            F.part_x = s->row_part + s->op_plan.grid;
            F.cnt = s->ls_cnt;
            F.blk_part = s->blk_part;
            F.grid_step = s->grid;
            F.ls_scal = s->ls_scal;
            F.pack = s->pack_local;
            F.ctl_rw = s->ctl;
            F.trace = s->trace;
            F.beta_ring = s->beta_ring;
            static const bool fuse_env = [] {
                const char* e = getenv("ZF_OP_FUSE_PROX");
                return e ? atoi(e) != 0 : true;
            }();
            // (images whose six arrays - x_k, x_{k-1}, x+, the two cached B W^-1 x, b - no longer fit the 256 MB of memory-side
            //  cache gain nothing: 3072^2 and 4096^2 -0.3 ... -0.7 %, the epilogue's 256-byte pieces of four coefficient
            //  quadrants go to HBM at 0.56 of peak where the separate step streams at 0.74; up to 2048^2: +6 ... +16 %,
            //  profiles/r05_operator_fuse_prox_ab.txt)
            const bool fuse_prox = fuse_env && decide_in_launch && !s->hist && d.n <= ZF_OP_FUSE_MAX_PIXELS;
            if (fuse_prox) {
                F.prox = 1;
                F.box = s->box ? 1 : 0;
                F.lo = d.box_lo;
                F.hi = d.box_hi;
                for (int k = 0; k < 3; ++k) F.xb[k] = s->xb[k];
                F.step_part = s->blk_part;
                F.grid_step = s->op_plan.grid;
                F.pass_log = a.pass_log;
                F.pass_slot = a.pass_slot;
                F.pass_tag = a.pass_tag;
                if (e0) ZF_HIP(hipEventRecord(e0, s->stream));
                zf_launch_op_adjoint(s->op_plan, s->stream, zf_op_of(s, s->ctl), nullptr, s->grad, 2 * d.scale, F);
                if (e1) ZF_HIP(hipEventRecord(e1, s->stream));
            } else {
                zf_launch_op_adjoint(s->op_plan, s->stream, zf_op_of(s, s->ctl), nullptr, s->grad, 2 * d.scale, F);
                a.p0 = s->grad;
                a.p1 = nullptr;
                if (e0) ZF_HIP(hipEventRecord(e0, s->stream));
                zf_launch_trial_kernels(s, a, false);
                if (e1) ZF_HIP(hipEventRecord(e1, s->stream));
            }
            if (decide_in_launch) {
                zf_launch_op_apply(s->op_plan, s->stream, zf_op_of(s, s->ctl), s->xb[0], s->xb[1], s->xb[2], s->sring.p[0], s->sring.p[1],
                                   s->sring.p[2], 1, F);
            } else {   // (a host-driven step sequence: s+, f(x+) and the finalize launch as for a matrix)
                zf_ring3 xr3 = {{s->xb[0], s->xb[1], s->xb[2]}};
                zf_launch_apply_A(s, s->ctl, xr3, s->sring, 1);
                hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, s->ctl, s->sring, 1, d.b,
                                   d.scale, m, s->ls_scal + 1);
                zf_launch_finalize(s, false);
            }
            ZF_HIP(hipGetLastError());
            return ZF_OK;
        }
        if (s->gemv_mfma) {
            dim3 gM((unsigned)((n + GEMVT_MFMA_COLS - 1) / GEMVT_MFMA_COLS), (unsigned)s->slices);
            hipLaunchKernelGGL(zf_gemvT_partial_mfma_kernel, gM, dim3(ZF_BLOCK), 0, s->stream, s->ctl, d.A,
                               s->resid, s->slab, m, n, s->rows_per_slice);
        } else if (nv > 0) {
            if (V == 2)
                hipLaunchKernelGGL(zf_gemvT_partial_kernel<2>, gT, dim3(ZF_BLOCK), 0, s->stream, s->ctl,
                                   d.A, s->resid, s->slab, m, n, s->rows_per_slice);
            else
                hipLaunchKernelGGL(zf_gemvT_partial_kernel<1>, gT, dim3(ZF_BLOCK), 0, s->stream, s->ctl,
                                   d.A, s->resid, s->slab, m, n, s->rows_per_slice);
        }
        const bool rows = d.world > 1 && d.row_sharded;
        // (row blocks: this rank's part 2 scale A_p^T r_p goes to s_part; the caller gathers the parts
        //  and zf_solver_enqueue_trial_finish() adds them and runs the rest of the trial)
        hipLaunchKernelGGL(zf_gemvT_combine_kernel, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, s->stream,
                           rows ? nullptr : s->ctl, s->slab, rows ? s->s_part : s->grad, 2 * d.scale, n, s->slices);
        if (rows) {
            // (the prox step of a row-sharded trial runs in zf_solver_enqueue_trial_finish, after the exchange: the
            //  event pair and the log slot taken above belong to a launch that does not happen here - give them back,
            //  or zf_collect_timing would read events that were never recorded)
            if (s->timing) {
                s->ev_used -= 1;
                if (a.pass_log) s->launches -= 1;
            }
            ZF_HIP(hipGetLastError());
            return ZF_OK;
        }
        // (2) fused prox step with the gradient vector in HBM
        a.p0 = s->grad;
        a.p1 = nullptr;
        if (e0) ZF_HIP(hipEventRecord(e0, s->stream));
        zf_launch_trial_kernels(s, a, false);
        if (e1) ZF_HIP(hipEventRecord(e1, s->stream));
        // (3) s+ = A x+ ; f(x+).  Sharded x (column blocks): this rank's A_p x_p+ goes to
        // s_part; the caller gathers the parts and zf_solver_enqueue_trial_finish() adds them.
        zf_ring3 xr = {{s->xb[0], s->xb[1], s->xb[2]}};
        zf_ring3 sout = s->sring;
        if (d.world > 1) sout = {{s->s_part, s->s_part, s->s_part}};   // (column blocks; row blocks returned above)
        int gr = (int)((m + GEMV_ROWS - 1) / GEMV_ROWS);
        if (gr > 8 * ZF_MAX_GRID) gr = 8 * ZF_MAX_GRID;
        if (V == 2)
            hipLaunchKernelGGL(zf_gemv_rows_kernel<2>, dim3(gr), dim3(ZF_BLOCK), 0, s->stream, s->ctl, d.A,
                               xr, sout, 1, m, n);
        else
            hipLaunchKernelGGL(zf_gemv_rows_kernel<1>, dim3(gr), dim3(ZF_BLOCK), 0, s->stream, s->ctl, d.A,
                               xr, sout, 1, m, n);
        if (d.world == 1) {
            hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, s->ctl, s->sring, 1,
                               d.b, d.scale, m, s->ls_scal + 1);
            zf_launch_finalize(s, decide_in_launch);
        }
    }
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// sharded least squares, second half of a trial: s+ = sum over ranks of A_p x_p+ (rank
// order), f(x+), local pack (f values contributed by rank 0 only: they are replicated)
extern "C" int zf_solver_enqueue_trial_finish(zf_solver* s) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_enqueue_trial_finish: solver not initialised");
    const zf_problem_desc& d = s->desc;
    if (d.kind != ZF_PROBLEM_LEAST_SQUARES_L1 || d.world == 1) return ZF_OK;
    const int64_t m = d.m_rows;
    if (d.row_sharded) {
        // row blocks: grad = sum over ranks (rank order) of the gathered parts, then - on the full,
        // replicated x, identically on every rank - the prox step, this rank's rows of A x+ and its
        // part of f(x+); the pack carries the replicated x sums from rank 0 only
        const int64_t n = d.n;
        zf_ring3 g3 = {{s->grad, s->grad, s->grad}};
        hipLaunchKernelGGL(zf_sum_parts_kernel, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, s->stream, nullptr, s->s_all,
                           (int)d.world, n, g3, -1);
        zf_step_args a;
        zf_fill_step_args(s, a);
        a.p0 = s->grad;
        a.p1 = nullptr;
        zf_launch_trial_kernels(s, a, false);
        const int V = (n % 2 == 0) ? 2 : 1;
        zf_ring3 xr = {{s->xb[0], s->xb[1], s->xb[2]}};
        int gr = (int)((m + GEMV_ROWS - 1) / GEMV_ROWS);
        if (gr > 8 * ZF_MAX_GRID) gr = 8 * ZF_MAX_GRID;
        if (V == 2)
            hipLaunchKernelGGL(zf_gemv_rows_kernel<2>, dim3(gr), dim3(ZF_BLOCK), 0, s->stream, s->ctl, d.A, xr, s->sring,
                               1, m, n);
        else
            hipLaunchKernelGGL(zf_gemv_rows_kernel<1>, dim3(gr), dim3(ZF_BLOCK), 0, s->stream, s->ctl, d.A, xr, s->sring,
                               1, m, n);
        hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, s->ctl, s->sring, 1, d.b,
                           d.scale, m, s->ls_scal + 1);
        zf_launch_finalize(s, false);
        ZF_HIP(hipGetLastError());
        return ZF_OK;
    }
    hipLaunchKernelGGL(zf_sum_parts_kernel, dim3(zf_grid_for(m)), dim3(ZF_BLOCK), 0, s->stream, s->ctl, s->s_all,
                       (int)d.world, m, s->sring, 1);
    hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, s->ctl, s->sring, 1, d.b,
                       d.scale, m, s->ls_scal + 1);
    zf_launch_finalize(s, false);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// least squares: with s0 = A x0 in sring[0], finish F(x0): A x_{-1} = A x0, f(x0), g(x0), init pack
static int zf_init_ls_tail(zf_solver* s) {
    const zf_problem_desc& d = s->desc;
    const int64_t n = d.n, m = d.m_rows;
    zf_ring3 s0 = {{s->sring.p[0], s->sring.p[0], s->sring.p[0]}};
    ZF_HIP(hipMemcpyAsync(s->sring.p[2], s->sring.p[0], sizeof(double) * m, hipMemcpyDeviceToDevice, s->stream));
    if (m > (int64_t)1 << 18) {   // long residuals (the operator problem at image sizes beyond 512 x 512): two launches, many workgroups
        const int wgs = zf_grid_for(m / 8);
        hipLaunchKernelGGL(zf_resid_x_wide_kernel, dim3(wgs), dim3(ZF_BLOCK), 0, s->stream, s0.p[0], d.b, m, s->partials);
        hipLaunchKernelGGL(zf_resid_x_finish_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, s->partials, wgs, d.scale, s->ls_scal + 1);
    } else {
        hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, s->stream, nullptr, s0, -1, d.b, d.scale,
                           m, s->ls_scal + 1);
    }
    const int g = zf_grid_for(n);
    if (s->box)
        hipLaunchKernelGGL((zf_eval_kernel<false, true>), dim3(g), dim3(ZF_BLOCK), 0, s->stream, s->xb[0], nullptr,
                           nullptr, d.box_lo, d.box_hi, n, s->partials);
    else
        hipLaunchKernelGGL((zf_eval_kernel<false, false>), dim3(g), dim3(ZF_BLOCK), 0, s->stream, s->xb[0], nullptr,
                           nullptr, d.box_lo, d.box_hi, n, s->partials);
    zf_init_args I;
    I.partials = s->partials;
    I.nblocks = g;
    I.f_scale = 0.0;
    I.lam = d.lam;
    I.f_ext = s->ls_scal + 1;
    I.pack = s->pack_local;
    I.contribute_g = (d.world == 1 || !d.row_sharded || d.rank == 0) ? 1 : 0;
    hipLaunchKernelGGL(zf_init_finalize_kernel, dim3(1), dim3(ZF_FIN_BLOCK), 0, s->stream, I);
    ZF_HIP(hipGetLastError());
    if (d.world == 1)
        ZF_HIP(hipMemcpyAsync(s->pack_all, s->pack_local, sizeof(double) * ZF_PACK_LEN, hipMemcpyDeviceToDevice,
                              s->stream));
    return ZF_OK;
}

extern "C" int zf_solver_svec_ptrs(zf_solver* s, double** s_part_dev, double** s_all_dev) {
    ZF_REQUIRE(s && s_part_dev && s_all_dev, "zf_solver_svec_ptrs: null argument");
    *s_part_dev = s->s_part;
    *s_all_dev = s->s_all;
    return ZF_OK;
}

extern "C" int zf_solver_set_svec_buffers(zf_solver* s, double* s_part_dev, double* s_all_dev) {
    ZF_REQUIRE(s && s_part_dev && s_all_dev, "zf_solver_set_svec_buffers: null argument");
    ZF_REQUIRE(!s->initialised, "zf_solver_set_svec_buffers: call before zf_solver_enqueue_init");
    ZF_REQUIRE(s->desc.kind == ZF_PROBLEM_LEAST_SQUARES_L1 && s->desc.world > 1,
               "zf_solver_set_svec_buffers: only for sharded least squares (m doubles per rank for column blocks, n for row blocks)");
    if (s->own_svec) {
        if (s->s_part) (void)hipFree(s->s_part);
        if (s->s_all) (void)hipFree(s->s_all);
    }
    s->own_svec = false;
    s->s_part = s_part_dev;
    s->s_all = s_all_dev;
    return ZF_OK;
}

extern "C" int zf_solver_enqueue_init(zf_solver* s, const double* x0_dev) {
    ZF_REQUIRE(s && x0_dev, "zf_solver_enqueue_init: null argument");
    s->shadow_valid = false;   // (the control block changes behind the host's back: predict again after the next poll)
    const zf_problem_desc& d = s->desc;
    const int64_t n = d.n;
    // x_k = x_{k-1} = y = x0  (proximal_gradient.py:463-465)
    ZF_HIP(hipMemcpyAsync(s->xb[0], x0_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s->stream));
    ZF_HIP(hipMemcpyAsync(s->xb[s->ring - 1], x0_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s->stream));
    zf_control c;
    memset(&c, 0, sizeof(c));
    c.lr = s->opt.lr;
    c.tol = s->opt.tol;
    c.tol_internal = s->opt.tol_internal;
    c.decay_rate = s->opt.decay_rate;
    c.max_iter = s->opt.max_iter;
    c.max_backtrack = s->opt.max_backtrack_iter;
    c.status = (s->opt.max_backtrack_iter == 0) ? ZF_BACKTRACK_FAILED : ZF_RUNNING;  // :280,:306
    c.cur = 0;
    c.nesterov = s->opt.nesterov;
    c.deprecated = s->opt.deprecated;
    c.accept_mode = s->res ? ZF_ACCEPT_RESOLVED : ZF_ACCEPT_REFERENCE;
    c.need_grad = 1;
    c.world = d.world;
    c.ring_size = s->ring;
    c.sub_iters = s->sub;
    c.prev = s->ring - 1;   // x_{-1} = x_0 (:463-465)
    c.lag = 0;
    c.pend_status = 0;
    ZF_HIP(hipMemcpyAsync(s->ctl, &c, sizeof(c), hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));   // `c` is a stack object
    zf_init_args I;
    I.partials = s->partials;
    I.lam = d.lam;
    I.pack = s->pack_local;
    I.f_ext = nullptr;
    I.contribute_g = 1;
    const int g = zf_grid_for(n);
    I.nblocks = g;
    if (d.kind == ZF_PROBLEM_DIAG_QUAD_L1) {
        I.f_scale = 0.5;
        if (s->box)
            hipLaunchKernelGGL((zf_eval_kernel<true, true>), dim3(g), dim3(ZF_BLOCK), 0, s->stream, s->xb[0],
                               d.d, d.c, d.box_lo, d.box_hi, n, s->partials);
        else
            hipLaunchKernelGGL((zf_eval_kernel<true, false>), dim3(g), dim3(ZF_BLOCK), 0, s->stream, s->xb[0],
                               d.d, d.c, d.box_lo, d.box_hi, n, s->partials);
    } else {
        // s0 = A x0 (sharded: this rank's A_p x0_p into s_part; zf_solver_enqueue_init_finish()
        // continues after the caller gathered the parts)
        zf_ring3 xr = {{s->xb[0], s->xb[0], s->xb[0]}};
        double* dst = (d.world > 1 && !d.row_sharded) ? s->s_part : s->sring.p[0];   // row blocks: A_p x0 is local
        zf_ring3 s0 = {{dst, dst, dst}};
        zf_launch_apply_A(s, nullptr, xr, s0, -1);
        ZF_HIP(hipGetLastError());
        if (d.world > 1 && !d.row_sharded) return ZF_OK;
        return zf_init_ls_tail(s);
    }
    hipLaunchKernelGGL(zf_init_finalize_kernel, dim3(1), dim3(ZF_FIN_BLOCK), 0, s->stream, I);
    ZF_HIP(hipGetLastError());
    if (d.world == 1)
        ZF_HIP(hipMemcpyAsync(s->pack_all, s->pack_local, sizeof(double) * ZF_PACK_LEN,
                              hipMemcpyDeviceToDevice, s->stream));
    return ZF_OK;
}

// Resume a solve from a saved state: the iterates in buffers `cur`, `prev`
// (device, n each; x_k, x_{k-1} once zf_solver_flush() has run) and the control block
// zf_solver_poll() returned when the state was taken.  Counters, lr, F(x_k), status continue from
// the saved values; buffer indices and the chain geometry are this solver's.  The momentum ring is NOT part of the state: the
// host re-uploads the factors from accepted count `nit` on (zf_solver_set_beta), which also
// resolves beta_next.  Least squares: A x_k and A x_{k-1} are recomputed by the same kernel that
// produced them (unsharded only).
extern "C" int zf_solver_restore(zf_solver* s, const double* xk_dev, const double* xprev_dev,
                                 const zf_control* saved, int64_t saved_bytes) {
    ZF_REQUIRE(s && xk_dev && xprev_dev && saved, "zf_solver_restore: null argument");
    // (nothing is read or written through `s` before the caller's sizes have been checked)
    ZF_REQUIRE(saved_bytes == (int64_t)sizeof(zf_control),
               "zf_solver_restore: saved_bytes differs from zf_sizeof_control() (a control block of another ABI version)");
    const zf_problem_desc& d = s->desc;
    ZF_REQUIRE(d.kind == ZF_PROBLEM_DIAG_QUAD_L1 || d.world == 1,
               "zf_solver_restore: sharded least squares is not supported");
    ZF_REQUIRE(saved->nit >= 0 && saved->lr > 0.0, "zf_solver_restore: implausible control block");
    const int64_t n = d.n;
    s->shadow_valid = false;   // (the control block changes behind the host's back: predict again after the next poll)
    ZF_HIP(hipMemcpyAsync(s->xb[0], xk_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s->stream));
    ZF_HIP(hipMemcpyAsync(s->xb[s->ring - 1], xprev_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s->stream));
    zf_control c = *saved;
    c.tol = s->opt.tol;
    c.tol_internal = s->opt.tol_internal;
    c.decay_rate = s->opt.decay_rate;
    c.max_iter = s->opt.max_iter;
    c.max_backtrack = s->opt.max_backtrack_iter;
    c.nesterov = s->opt.nesterov;
    c.deprecated = s->opt.deprecated;
    c.accept_mode = s->res ? ZF_ACCEPT_RESOLVED : ZF_ACCEPT_REFERENCE;   // (this solver's: the test a resumed solve runs is the resuming caller's choice)
    c.cur = 0;
    c.prev = s->ring - 1;
    c.ring_size = s->ring;
    c.world = d.world;
    c.need_grad = 1;
    // lagging iterations (accepted, iterates not stored) travel with the state when the chain
    // geometry can replay them; zf_solver_flush() before the snapshot removes them
    ZF_REQUIRE(saved->lag >= 0 && saved->lag <= 2 * s->sub - 2,
               "zf_solver_restore: the saved state lags more iterations than this chain length can replay "
               "(zf_solver_flush before taking the snapshot)");
    c.sub_iters = s->sub;
    c.pass_seq = 0;   // (step numbers are this solver's own)
    if (c.status == ZF_MAXITER && c.nit < c.max_iter) c.status = ZF_RUNNING;   // a larger max_iter continues (:539)
    // (packs of passes from before the restore must not pass for packs of the restored state: zf_pack_stamp)
    if (s->pack_all)
        ZF_HIP(hipMemsetAsync(s->pack_all, 0, sizeof(double) * ZF_PACK_LEN * s->sub * d.world, s->stream));
    ZF_HIP(hipMemcpyAsync(s->ctl, &c, sizeof(c), hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));   // `c` is a stack object
    if (zf_is_ls(d.kind)) {
        const int at[2] = {0, 2};   // A x_k -> sring[cur], A x_{k-1} -> sring[(cur + 2) % 3]
        for (int k = 0; k < 2; ++k) {
            double* xsrc = s->xb[at[k]];
            double* dst = s->sring.p[at[k]];
            zf_ring3 xr = {{xsrc, xsrc, xsrc}};
            zf_ring3 so = {{dst, dst, dst}};
            zf_launch_apply_A(s, nullptr, xr, so, -1);
        }
        ZF_HIP(hipGetLastError());
    }
    s->initialised = true;
    return ZF_OK;
}

extern "C" int zf_solver_get_x_prev(zf_solver* s, double* x_host, int64_t count) {
    ZF_REQUIRE(s && x_host, "zf_solver_get_x_prev: null argument");
    ZF_REQUIRE(count >= s->desc.n, "zf_solver_get_x_prev: the host buffer holds fewer than n doubles");
    zf_control c;
    ZF_HIP(hipMemcpyAsync(&c, s->ctl, sizeof(c), hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    ZF_HIP(hipMemcpyAsync(x_host, s->xb[c.prev], sizeof(double) * s->desc.n, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

// sharded least squares, second half of the initialisation: s0 = sum of the gathered parts
extern "C" int zf_solver_enqueue_init_finish(zf_solver* s) {
    ZF_REQUIRE(s, "zf_solver_enqueue_init_finish: null solver");
    const zf_problem_desc& d = s->desc;
    if (d.kind != ZF_PROBLEM_LEAST_SQUARES_L1 || d.world == 1 || d.row_sharded) return ZF_OK;
    zf_ring3 s0 = {{s->sring.p[0], s->sring.p[0], s->sring.p[0]}};
    hipLaunchKernelGGL(zf_sum_parts_kernel, dim3(zf_grid_for(d.m_rows)), dim3(ZF_BLOCK), 0, s->stream, nullptr,
                       s->s_all, (int)d.world, d.m_rows, s0, -1);
    ZF_HIP(hipGetLastError());
    return zf_init_ls_tail(s);
}

extern "C" int zf_solver_enqueue_init_commit(zf_solver* s) {
    ZF_REQUIRE(s, "zf_solver_enqueue_init_commit: null solver");
    // column-sharded least squares: f(x0) is replicated (taken once); row blocks and P-diag: partial sums
    const int f_repl = (s->desc.kind == ZF_PROBLEM_LEAST_SQUARES_L1 && !s->desc.row_sharded) ? 1 : 0;
    hipLaunchKernelGGL(zf_init_commit_kernel, dim3(1), dim3(64), 0, s->stream, s->ctl, s->pack_all, f_repl,
                       s->sub * ZF_PACK_LEN);
    ZF_HIP(hipGetLastError());
    s->initialised = true;
    return ZF_OK;
}

extern "C" int zf_solver_set_beta(zf_solver* s, int64_t first, const double* beta_host, int64_t count) {
    ZF_REQUIRE(s && beta_host, "zf_solver_set_beta: null argument");
    ZF_REQUIRE(first >= 0 && count >= 0 && count <= ZF_RING, "zf_solver_set_beta: bad range");
    // ring index = accepted-iteration count % ZF_RING; at most two contiguous pieces
    int64_t done = 0;
    while (done < count) {
        const int64_t pos = (first + done) % ZF_RING;
        int64_t len = count - done;
        if (len > ZF_RING - pos) len = ZF_RING - pos;
        ZF_HIP(hipMemcpyAsync(s->beta_ring + pos, beta_host + done, sizeof(double) * len,
                              hipMemcpyHostToDevice, s->stream));
        done += len;
    }
    hipLaunchKernelGGL(zf_refresh_beta_kernel, dim3(1), dim3(64), 0, s->stream, s->ctl, s->beta_ring);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipStreamSynchronize(s->stream));   // beta_host may be reused by the caller
    return ZF_OK;
}

extern "C" int zf_solver_enqueue_trial(zf_solver* s) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_enqueue_trial: solver not initialised");
    return zf_launch_trial(s, false);
}

extern "C" int zf_solver_enqueue_decide(zf_solver* s) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_enqueue_decide: solver not initialised");
    // (separable problems finalise in the trial launch and stamp their packs: zf_pack_stamp)
    const int stamped = (s->desc.kind == ZF_PROBLEM_DIAG_QUAD_L1 && !zf_fin_kernel_mode()) ? 1 : 0;
    hipLaunchKernelGGL(zf_decide_kernel, dim3(1), dim3(64), 0, s->stream, s->ctl, s->pack_all, s->trace,
                       s->beta_ring, s->sub, stamped);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// Streaming return_all (proximal_gradient.py:521-524 keeps every iterate): `hist_dev` is a caller-owned
// ring of `cap_slots` iterates, `stride` doubles apart (>= n, a multiple of 64: slots stay 512-B
// aligned).  From now on every trial stores its iterate x_{k+1} into slot (k + 1) % cap_slots as it
// computes it - the iterates inside a chain of 8 exist nowhere else - so recording costs 8 more bytes
// per element and iteration and no host transfer; x0 (slot 0) is the caller's.  The caller reads slots
// of accepted iterations after a poll and must not let the solve run cap_slots iterations ahead of
// what it still needs.  Chain lengths 1 and 8 only (2 and 4 fall back to 1 at creation).
extern "C" int zf_solver_set_history(zf_solver* s, double* hist_dev, int64_t cap_slots, int64_t stride) {
    ZF_REQUIRE(s && hist_dev && cap_slots >= 2, "zf_solver_set_history: bad argument");
    ZF_REQUIRE(stride >= s->desc.n && stride % 64 == 0 && zf_aligned16(hist_dev),
               "zf_solver_set_history: stride must be >= n and a multiple of 64 doubles, the ring 16-byte aligned");
    ZF_REQUIRE(s->sub == 1 || s->sub == 8, "zf_solver_set_history: chain length must be 1 or 8");
    ZF_REQUIRE(!s->res || s->sub == 1, "zf_solver_set_history: with ZF_ACCEPT_RESOLVED the recording kernels exist for single trials only (sub_iters 1)");
    ZF_REQUIRE(cap_slots > 2 * s->sub, "zf_solver_set_history: the ring must hold more than two chains");
    s->hist = hist_dev;
    s->hist_cap = cap_slots;
    s->hist_stride = stride;
    return ZF_OK;
}

extern "C" int zf_solver_launch_counts(zf_solver* s, int64_t* out, int64_t count) {
    ZF_REQUIRE(s && out, "zf_solver_launch_counts: null argument");
    ZF_REQUIRE(count >= 2, "zf_solver_launch_counts: the output holds fewer than 2 values");
    out[0] = s->steps_issued;
    out[1] = s->kernels_issued;
    if (count >= 4) out[2] = out[3] = 0;   // (ABI 5: the withdrawn multi-pass kernel's counts)
    if (count >= 6) {   // run-ahead passes launched, and those of them launched behind a pass still in flight
        out[4] = s->ra_passes;
        out[5] = s->ra_ahead;
    }
    if (count >= 8) {   // (ABI 6) as of the last poll: waits of run-ahead passes that gave up; void run-ahead passes
        out[6] = s->ra_timeouts;
        out[7] = s->ra_voids;
    }
    if (count >= 10) {  // (ABI 6) passes launched ahead at kernel granularity; of them void, as of the last poll
        out[8] = s->ah_passes;
        out[9] = s->ah_voids;
    }
    if (count >= 11) out[10] = s->ra_off ? 1 : 0;   // run-ahead passes were switched off for this solver after a wait gave up
    return ZF_OK;
}

extern "C" int zf_solver_sub_iters(zf_solver* s, int32_t* sub_iters) {
    ZF_REQUIRE(s && sub_iters, "zf_solver_sub_iters: null argument");
    *sub_iters = s->sub;
    return ZF_OK;
}

extern "C" int zf_solver_set_max_iter(zf_solver* s, int64_t max_iter) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_set_max_iter: solver not initialised");
    ZF_REQUIRE(max_iter >= 1, "zf_solver_set_max_iter: max_iter must be >= 1");
    if (s->shadow_valid) {   // (mirror of zf_set_max_iter_kernel on the host's expectation)
        zf_control& c = s->shadow;
        c.max_iter = max_iter;
        if (c.status == ZF_MAXITER && c.nit < max_iter) c.status = ZF_RUNNING;
        if (c.pend_status == ZF_MAXITER && c.nit < max_iter) c.pend_status = 0;
    }
    s->opt.max_iter = max_iter;
    hipLaunchKernelGGL(zf_set_max_iter_kernel, dim3(1), dim3(64), 0, s->stream, s->ctl, max_iter);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// world == 1: make the next step materialise lagging iterates (no-op when there are none), so
// that x_k, x_{k-1} are in buffers `cur`, `prev` at the next poll - snapshots, history taps
extern "C" int zf_solver_flush(zf_solver* s) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_flush: solver not initialised");
    if (s->shadow_valid) {   // (mirror of zf_flush_kernel)
        zf_control& c = s->shadow;
        if (c.status == ZF_RUNNING && c.lag > 0 && c.pend_status == 0) c.pend_status = ZF_PEND_FLUSH;
    }
    hipLaunchKernelGGL(zf_flush_kernel, dim3(1), dim3(64), 0, s->stream, s->ctl);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// Attach an RCCL communicator (zf_comm_create; its rank / world must be the descriptor's): from now on
// zf_solver_enqueue_init_all and zf_solver_enqueue_steps issue the exchanges of a sharded step - the
// packed all-gather of the scalar packs (C1), and for column-sharded least squares the all-gather of
// A_p x_p (C2) - themselves, on the solver's stream.  The communicator must outlive the solver.
extern "C" int zf_solver_set_comm(zf_solver* s, zf_comm* comm) {
    ZF_REQUIRE(s && comm, "zf_solver_set_comm: null argument");
    int32_t rank = 0, world = 1;
    int rc = zf_comm_info(comm, &rank, &world);
    if (rc) return rc;
    ZF_REQUIRE(rank == s->desc.rank && world == s->desc.world, "zf_solver_set_comm: rank / world differ from the problem descriptor");
    s->comm = comm;
    {   // Thread ranks (the in-process group: several ranks of ONE device) do not hold two passes of every rank at once - sharded
        // run-ahead passes would wait for workgroups that have no slot (correct, counted, switched off at the first poll, but
        // 15 ms per wait): those groups keep to passes ahead, which never wait inside a kernel.  ZF_RUNAHEAD_SHARDED=1 insists.
        zf_comm_desc cd;
        if (world > 1 && !getenv("ZF_RUNAHEAD_SHARDED") && zf_comm_describe(comm, &cd, (int64_t)sizeof(cd)) == ZF_OK && cd.kind == 1)
            s->ra_sharded = false;
    }
    // passes ahead of their predecessor's decision (zf_launch_ahead) need six iterate buffers and the second stream: a
    // one-rank solver was created without knowing that a communicator would follow (before the initialisation only)
    if (s->ah && !s->initialised && (s->ring < 6 || !s->stream2)) {
        ZF_HIP(hipStreamSynchronize(s->stream));
        if (s->ring < 6) {
            const int64_t n_pad = (s->desc.n + 63) & ~int64_t(63);
            double* nb = nullptr;
            if (hipMalloc(&nb, sizeof(double) * 6 * n_pad) == hipSuccess) {
                (void)hipFree(s->xbuf);
                s->xbuf = nb;
                s->ring = 6;
                for (int k = 0; k < s->ring; ++k) s->xb[k] = s->xbuf + k * n_pad;
            } else {
                (void)hipGetLastError();   // (no room for two more iterates: the solve runs one pass at a time)
            }
        }
        if (s->ring >= 6 && zf_second_stream(s) != hipSuccess) {
            (void)hipGetLastError();
            s->ah = false;   // (the sequence trial -> all-gather -> decide on the caller's stream, as without passes ahead)
        }
    }
    return ZF_OK;
}

static int zf_gather_packs(zf_solver* s, int64_t packs, hipStream_t st, const double* src) {
    if (!st) st = s->stream;
    if (!src) src = s->pack_local;
    if (!s->timing) return zf_comm_all_gather(s->comm, src, s->pack_all, packs * ZF_PACK_LEN, st);
    // timed: from "this rank's packs are ready" to "the gathered packs are here" on this rank's stream - the
    // collective itself plus the wait for the slowest rank (zf_solver_exchange_stats)
    if (s->xev_used == s->xev_pool.size()) {
        hipEvent_t x, y;
        ZF_HIP(hipEventCreate(&x));
        ZF_HIP(hipEventCreate(&y));
        s->xev_pool.emplace_back(x, y);
    }
    const auto& ev = s->xev_pool[s->xev_used++];
    ZF_HIP(hipEventRecord(ev.first, st));
    const int rc = zf_comm_all_gather(s->comm, src, s->pack_all, packs * ZF_PACK_LEN, st);
    ZF_HIP(hipEventRecord(ev.second, st));
    return rc;
}
static int zf_gather_svec(zf_solver* s, bool at_init = false) {
    if (s->desc.kind != ZF_PROBLEM_LEAST_SQUARES_L1 || s->desc.world == 1) return ZF_OK;
    if (s->desc.row_sharded) {   // the n-vector A_p^T r_p of a trial; the init has nothing to exchange
        if (at_init) return ZF_OK;
        return zf_comm_all_gather(s->comm, s->s_part, s->s_all, s->desc.n, s->stream);
    }
    return zf_comm_all_gather(s->comm, s->s_part, s->s_all, s->desc.m_rows, s->stream);
}

// the whole initialisation of a solver with a communicator: init, exchanges, commit (world == 1 too)
extern "C" int zf_solver_enqueue_init_all(zf_solver* s, const double* x0_dev) {
    ZF_REQUIRE(s && x0_dev, "zf_solver_enqueue_init_all: null argument");
    int rc = zf_solver_enqueue_init(s, x0_dev);
    if (rc) return rc;
    if (s->comm) {
        if ((rc = zf_gather_svec(s, true))) return rc;
        if ((rc = zf_solver_enqueue_init_finish(s))) return rc;
        // the init pack sits at the head of this rank's pack buffer; ranks are sub x ZF_PACK_LEN apart in
        // pack_all (zf_init_commit_kernel's stride), so the whole buffer is gathered
        if ((rc = zf_gather_packs(s, s->sub, nullptr))) return rc;
    } else {
        ZF_REQUIRE(s->desc.world == 1, "zf_solver_enqueue_init_all: world > 1 needs zf_solver_set_comm");
    }
    return zf_solver_enqueue_init_commit(s);
}

extern "C" int zf_solver_enqueue_steps(zf_solver* s, int64_t steps) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_enqueue_steps: solver not initialised");
    ZF_REQUIRE(s->desc.world == 1 || s->comm, "zf_solver_enqueue_steps: world > 1 needs zf_solver_set_comm "
                                              "(or the caller's own trial / gather / decide sequence)");
    ZF_REQUIRE(steps >= 0 && steps <= ZF_RING, "zf_solver_enqueue_steps: steps must be in [0, ZF_RING]");
    for (int64_t k = 0; k < steps; ++k) {
        int rc;
        if (!s->comm) {
            if ((rc = zf_launch_trial(s, true))) return rc;
            continue;
        }
        // sharded step: trial -> (C2) -> finish -> C1 -> decide; every rank runs the same decide on the
        // same gathered packs, so nothing else is exchanged.  A pass AHEAD (zf_launch_ahead) has enqueued all of it
        // already: finalisation, exchange and decide on the second stream, beside the next pass's trial kernel.
        bool ahead = false;
        s->chunk_last = k + 1 == steps;
        rc = zf_launch_trial(s, false, false, &ahead);
        s->chunk_last = false;
        if (rc) return rc;
        if (ahead) continue;
        if ((rc = zf_gather_svec(s))) return rc;
        if ((rc = zf_solver_enqueue_trial_finish(s))) return rc;
        if ((rc = zf_gather_packs(s, s->sub, nullptr))) return rc;
        if ((rc = zf_solver_enqueue_decide(s))) return rc;
    }
    // everything enqueued on the solver's stream after this call (a poll's copy, a snapshot, another solver's work on
    // the caller's stream) comes behind the passes on the second stream as well
    if (s->stream2) return zf_ra_join(s);
    return ZF_OK;
}

static void zf_set_tiles(zf_solver* s, int tiles) {
    if (tiles < 1) tiles = 1;
    if (tiles > ZF_MAX_TILES_PER_WG) tiles = ZF_MAX_TILES_PER_WG;
    s->tiles = tiles;
    s->grid = (int)((s->ntiles + tiles - 1) / tiles);
}

// Launch geometry of the trial kernel: T interleaved tiles per workgroup.  T decides which
// elements a thread accumulates, hence the rounding of the six sums and, at a knife edge, an
// accept / reject decision - so it is a FUNCTION OF n ONLY (never of a timing measurement, the
// chain length or the device found at run time): the same problem takes the same decisions in every
// process, on every rank layout with equal shard sizes, for every S and after every restore.
// Up to 512 tiles: T = 1.  From there on the grid is a
// round or more deep and T follows the rounds: the chained kernels keep two workgroups per CU resident - 512
// on the 256 CUs of an MI355X - and a workgroup costs its T tiles plus ~0.3 of a tile for its start and
// epilogue (6 S wave reductions), so a pass takes  rounds(T) x (T + 0.3)  tile times with
// rounds = ceil(ceil(tiles / T) / 512); T is the smallest minimiser in 8 .. 24 (n = 1e8: 4 rounds of 24
// tiles instead of 12 of 8: 1.285 -> 1.26 ms per 16-chain pass; the sweep T = 2 .. 96 is in DESIGN.md 4.1;
// n = 1e7: ONE round of 489 workgroups of 10 tiles instead of 611 of 8 - a second round 19 % full: 0.149 ->
// 0.128 ms per 16-chain pass, round 3; round 2 had measured "no faster" through a 22 us finalize launch).
// (Round 1 picked T by timing; tools/tune_trial.hip keeps that experiment.)  ZF_TILES_PER_WG=<n>
// overrides it for experiments and changes the rounding of the sums with it.
static int zf_tiles_for(int64_t ntiles) {
    constexpr int64_t SLOTS = 512;
    if (ntiles <= SLOTS) return 1;
    // (round 4) between one round of single-tile workgroups and 4096 tiles the same cost rule over T = 1 .. 8: whole
    // rounds here too - n = 2e6 runs ONE round of 489 workgroups of 2 tiles instead of two rounds of single tiles - and
    // a grid the device holds at once, which is what lets passes run ahead of each other (zf_runahead_kernel)
    const int t_lo = ntiles < 4096 ? 1 : 8, t_hi = ntiles < 4096 ? 8 : ZF_MAX_TILES_PER_WG;
    int best_t = t_lo;
    double best_cost = 0.0;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int64_t wgs = (ntiles + t - 1) / t;
        const int64_t rounds = (wgs + SLOTS - 1) / SLOTS;
        const double cost = (double)rounds * ((double)t + 0.3);
        if (t == t_lo || cost < best_cost - 1e-9) {
            best_cost = cost;
            best_t = t;
        }
    }
    return best_t;
}

extern "C" int zf_solver_autotune(zf_solver* s, int32_t* chosen_tiles) {
    ZF_REQUIRE(s && s->initialised, "zf_solver_autotune: solver not initialised");
    const char* env = getenv("ZF_TILES_PER_WG");
    if (env) zf_set_tiles(s, atoi(env));
    else if (s->desc.kind == ZF_PROBLEM_DIAG_QUAD_L1) zf_set_tiles(s, zf_tiles_for(s->ntiles));
    else zf_set_tiles(s, 1);
    if (chosen_tiles) *chosen_tiles = s->tiles;
    return ZF_OK;
}

extern "C" int zf_solver_pack_ptrs(zf_solver* s, double** pack_local_dev, double** pack_all_dev) {
    ZF_REQUIRE(s && pack_local_dev && pack_all_dev, "zf_solver_pack_ptrs: null argument");
    *pack_local_dev = s->pack_local;
    *pack_all_dev = s->pack_all;
    return ZF_OK;
}

extern "C" int zf_solver_set_pack_buffers(zf_solver* s, double* pack_local_dev, double* pack_all_dev) {
    ZF_REQUIRE(s && pack_local_dev && pack_all_dev, "zf_solver_set_pack_buffers: null argument");
    ZF_REQUIRE(!s->initialised, "zf_solver_set_pack_buffers: call before zf_solver_enqueue_init");
    if (s->pack_local && s->own_packs) (void)hipFree(s->pack_local);
    if (s->pack_all && s->own_packs) (void)hipFree(s->pack_all);
    s->own_packs = false;
    s->pack_local = pack_local_dev;
    s->pack_all = pack_all_dev;
    return ZF_OK;
}

// log_in_mail: the caller (a poll) has just copied the log into the pinned mirror, with the control block and the trace
static int zf_collect_timing(zf_solver* s, bool log_in_mail = false) {
    if (!log_in_mail && s->ev_used > 0) {
        const size_t off = ZF_CTL_SLOT + ZF_TRACE_BYTES;
        ZF_HIP(hipMemcpyAsync(s->mail + off, s->ctl_trace + off, sizeof(int) * ZF_PASS_LOG, hipMemcpyDeviceToHost, s->stream));
        ZF_HIP(hipStreamSynchronize(s->stream));
    }
    const bool have_log = s->ev_used > 0 && s->ev_used <= (size_t)ZF_PASS_LOG &&
                          s->launches - s->first_uncollected == (int64_t)s->ev_used;
    const int* log = reinterpret_cast<const int*>(s->mail + ZF_CTL_SLOT + ZF_TRACE_BYTES);
    for (size_t k = 0; k < s->ev_used; ++k) {
        float ms = 0.f;
        ZF_HIP(hipEventElapsedTime(&ms, s->ev_pool[k].first, s->ev_pool[k].second));
        s->ms_total += ms;
        s->ms_count += 1;
        if (have_log) {
            const int64_t launch = s->first_uncollected + (int64_t)k;
            const int entry = log[launch % ZF_PASS_LOG];
            // the launch found the solve finished and exited: the slot still holds what an earlier launch wrote
            if (entry < 0 || (entry >> 16) != (int)(launch & 0x7fff)) continue;
            const int shape = entry & 0xffff;   // zf_log_shape: fresh trials | lagging iterations << 5 | passes << 10
            if (shape == 0) continue;           // a pass that ran ahead and turned out VOID: the launch ran, its pass does not count
            if (s->records.size() < 65536) s->records.emplace_back(shape, ms);
            const int nf = shape & 31, lag = (shape >> 5) & 31, cnt = (shape >> 10) & 63;
            if (lag == 0 && nf == s->sub) {
                // (a persistent launch ran cnt full-chain passes back to back: its duration counts for all of them)
                s->ms_full += ms;
                s->n_full += cnt > 0 ? cnt : 1;
            } else {
                s->ms_part += ms;
                s->n_part += 1;
                s->fresh_part += nf;
                s->lag_part += lag;
            }
        }
    }
    s->first_uncollected = s->launches;
    s->ev_used = 0;
    for (size_t k = 0; k < s->xev_used; ++k) {
        float ms = 0.f;
        ZF_HIP(hipEventElapsedTime(&ms, s->xev_pool[k].first, s->xev_pool[k].second));
        s->xms_total += ms;
        s->xms_count += 1;
    }
    s->xev_used = 0;
    return ZF_OK;
}

// Timing on: the launches that ran a pass since the last call, one (shape, milliseconds) pair each - shape = fresh trials |
// lagging iterations << 5 | passes of a persistent launch << 10 (zf_log_shape) - oldest first; *count receives how
// many pairs were written (at most cap_pairs; the rest is dropped).  For per-shape kernel statistics (tools/long_run.py).
extern "C" int zf_solver_pass_records(zf_solver* s, double* out, int64_t cap_pairs, int64_t* count) {
    ZF_REQUIRE(out && count && cap_pairs >= 0, "zf_solver_pass_records: bad argument");
    ZF_REQUIRE(s, "zf_solver_pass_records: null solver");
    ZF_HIP(hipStreamSynchronize(s->stream));
    int rc = zf_collect_timing(s);
    if (rc) return rc;
    int64_t k = 0;
    for (; k < (int64_t)s->records.size() && k < cap_pairs; ++k) {
        out[2 * k] = (double)s->records[k].first;
        out[2 * k + 1] = (double)s->records[k].second;
    }
    *count = k;
    s->records.clear();
    return ZF_OK;
}

// Sharded solves with timing on: out[0] = mean ms, out[1] = count of the per-pass pack exchanges since the last
// call - each measured on this rank's stream from "my packs are ready" to "the gathered packs are here" (the
// collective and the wait for the slowest rank).  Resets its window.
extern "C" int zf_solver_exchange_stats(zf_solver* s, double* out, int64_t count) {
    ZF_REQUIRE(out && count >= 2, "zf_solver_exchange_stats: needs a buffer of >= 2 doubles");
    ZF_REQUIRE(s, "zf_solver_exchange_stats: null solver");
    ZF_HIP(hipStreamSynchronize(s->stream));
    int rc = zf_collect_timing(s);
    if (rc) return rc;
    out[0] = s->xms_count ? s->xms_total / (double)s->xms_count : 0.0;
    out[1] = (double)s->xms_count;
    s->xms_total = 0.0;
    s->xms_count = 0;
    return ZF_OK;
}

extern "C" int zf_solver_poll(zf_solver* s, zf_control* ctl_host, int64_t ctl_bytes, double* trace_host,
                              int64_t trace_bytes) {
    ZF_REQUIRE(s && ctl_host, "zf_solver_poll: null argument");
    // (a host built against an older, smaller zf_control is refused - not overrun)
    ZF_REQUIRE(ctl_bytes >= (int64_t)sizeof(zf_control),
               "zf_solver_poll: ctl_bytes is smaller than zf_sizeof_control() (host built against another ABI version?)");
    ZF_REQUIRE(!trace_host || trace_bytes >= (int64_t)(sizeof(double) * ZF_RING * ZF_TRACE_COLS),
               "zf_solver_poll: trace_bytes is smaller than ZF_RING * ZF_TRACE_COLS doubles");
    const size_t trace_len = ZF_TRACE_BYTES;
    const bool with_log = s->ev_used > 0;   // (events were recorded: the kernels logged their pass shapes)
    const size_t want = with_log ? ZF_MAIL_BYTES : trace_host ? ZF_CTL_SLOT + trace_len : ZF_CTL_SLOT;
    ZF_HIP(hipMemcpyAsync(s->mail, s->ctl_trace, want, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    memcpy(ctl_host, s->mail, sizeof(zf_control));
    if (trace_host) memcpy(trace_host, s->mail + ZF_CTL_SLOT, trace_len);
    {   // what the host now knows about the device: the basis of the next chunk's predictions
        const int64_t rej = ctl_host->total_trials - ctl_host->nit;
        s->careful = s->shadow_valid ? (rej != s->polled_rejections) : (rej != 0);
        s->polled_rejections = rej;
        s->shadow = *ctl_host;
        s->shadow_valid = true;
        s->steps_since_poll = 0;
        // what the passes that ran ahead report (zf_solver::ra_stats).  A wait that gave up means the device does not hold
        // two passes of this solver at once - it is being shared - and every further one would cost its whole limit again:
        // no more run-ahead passes for this solver (passes ahead at kernel granularity never wait and go on)
        const unsigned* st = reinterpret_cast<const unsigned*>(s->mail + ZF_STATS_OFF);
        s->ra_timeouts = st[0];
        s->ra_voids = st[1];
        s->ah_voids = st[2];
        if (s->ra_timeouts > 0) s->ra_off = true;
    }
    return zf_collect_timing(s, with_log);
}

extern "C" int zf_solver_x_dev(zf_solver* s, const double** x_dev) {
    ZF_REQUIRE(s && x_dev, "zf_solver_x_dev: null argument");
    zf_control c;
    ZF_HIP(hipMemcpyAsync(&c, s->ctl, sizeof(c), hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    *x_dev = s->xb[c.cur];
    return ZF_OK;
}

extern "C" int zf_solver_get_x(zf_solver* s, double* x_host, int64_t count) {
    ZF_REQUIRE(s && x_host, "zf_solver_get_x: null argument");
    ZF_REQUIRE(count >= s->desc.n, "zf_solver_get_x: the host buffer holds fewer than n doubles");
    const double* xd = nullptr;
    int rc = zf_solver_x_dev(s, &xd);
    if (rc) return rc;
    ZF_HIP(hipMemcpyAsync(x_host, xd, sizeof(double) * s->desc.n, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

extern "C" int zf_solver_set_timing(zf_solver* s, int32_t enabled) {
    ZF_REQUIRE(s, "zf_solver_set_timing: null solver");
    s->timing = enabled != 0;
    return ZF_OK;
}

// Trial-kernel durations since the last call, split by the shape of the pass (the kernel logs it):
// out[0], out[1] = mean ms and count of full chains (S fresh trials, nothing replayed);
// out[2], out[3] = mean ms and count of every other pass (shorter chains, replays, materialise-only).
// Launches that found the solve finished are not counted.  Resets the window; call it INSTEAD of
// zf_solver_trial_kernel_ms (which resets the same window).
extern "C" int zf_solver_pass_stats(zf_solver* s, double* out, int64_t count) {
    ZF_REQUIRE(s && out && count >= 4, "zf_solver_pass_stats: needs a buffer of >= 4 doubles");
    ZF_HIP(hipStreamSynchronize(s->stream));
    int rc = zf_collect_timing(s);
    if (rc) return rc;
    out[0] = s->n_full ? s->ms_full / (double)s->n_full : 0.0;
    out[1] = (double)s->n_full;
    out[2] = s->n_part ? s->ms_part / (double)s->n_part : 0.0;
    out[3] = (double)s->n_part;
    s->ms_full = s->ms_part = 0.0;
    s->n_full = s->n_part = 0;
    s->fresh_part = s->lag_part = 0;
    s->ms_total = 0.0;
    s->ms_count = 0;
    return ZF_OK;
}

// The same window with the work of the other passes: out[4] = fresh trials, out[5] = replayed iterations
// they carried in total (a pass of the shared tail before max_iter runs about left / 2 fresh trials, a pass
// after a broken chain replays the lagging iterations first).
extern "C" int zf_solver_pass_stats_ex(zf_solver* s, double* out, int64_t count) {
    ZF_REQUIRE(s && out && count >= 6, "zf_solver_pass_stats_ex: needs a buffer of >= 6 doubles");
    ZF_HIP(hipStreamSynchronize(s->stream));
    int rc = zf_collect_timing(s);
    if (rc) return rc;
    out[4] = (double)s->fresh_part;
    out[5] = (double)s->lag_part;
    return zf_solver_pass_stats(s, out, 4);
}

extern "C" int zf_solver_trial_kernel_ms(zf_solver* s, double* avg_ms, int64_t* launches) {
    ZF_REQUIRE(s && avg_ms && launches, "zf_solver_trial_kernel_ms: null argument");
    ZF_HIP(hipStreamSynchronize(s->stream));
    int rc = zf_collect_timing(s);
    if (rc) return rc;
    *launches = s->ms_count;
    *avg_ms = s->ms_count ? s->ms_total / (double)s->ms_count : 0.0;
    s->ms_total = 0.0;
    s->ms_count = 0;
    s->ms_full = s->ms_part = 0.0;
    s->n_full = s->n_part = 0;
    s->fresh_part = s->lag_part = 0;
    return ZF_OK;
}

// ---------------------------------------------------------------------------
// library / device / memory
// ---------------------------------------------------------------------------
extern "C" int zf_abi_version(void) { return ZF_ABI_VERSION; }
extern "C" const char* zf_last_error(void) { return zf_errbuf; }
extern "C" int64_t zf_sizeof_control(void) { return (int64_t)sizeof(zf_control); }

extern "C" int zf_device_count(int* count) {
    ZF_REQUIRE(count, "zf_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return zf_fail(ZF_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return ZF_OK;
}
extern "C" int zf_set_device(int device) {
    ZF_HIP(hipSetDevice(device));
    return ZF_OK;
}
extern "C" int zf_malloc(void** dev_ptr, int64_t bytes) {
    ZF_REQUIRE(dev_ptr && bytes >= 0, "zf_malloc: bad argument");
    ZF_HIP(hipMalloc(dev_ptr, (size_t)(bytes > 0 ? bytes : 16)));
    return ZF_OK;
}
extern "C" int zf_free(void* dev_ptr) {
    if (dev_ptr) ZF_HIP(hipFree(dev_ptr));
    return ZF_OK;
}
extern "C" int zf_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream) {
    ZF_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    ZF_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZF_OK;
}
extern "C" int zf_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream) {
    ZF_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    ZF_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZF_OK;
}
extern "C" int zf_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream) {
    ZF_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ZF_OK;
}
extern "C" int zf_stream_synchronize(void* stream) {
    ZF_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZF_OK;
}

extern "C" int zf_decide_host(zf_control* ctl, int64_t ctl_bytes, const double* packs, double* trace) {
    ZF_REQUIRE(ctl && packs && trace, "zf_decide_host: null argument");
    ZF_REQUIRE(ctl_bytes == (int64_t)sizeof(zf_control), "zf_decide_host: ctl_bytes differs from zf_sizeof_control()");
    zf_decide_pass(ctl, packs, trace);
    return ZF_OK;
}

// ---------------------------------------------------------------------------
// least-squares operator at a host point (callback contract, not the hot loop)
// ---------------------------------------------------------------------------
extern "C" int zf_ls_eval(const double* A_dev, const double* b_dev, int64_t m_rows, int64_t n, double scale,
                          const double* x_host, double* f_out, double* grad_out_host) {
    ZF_REQUIRE(A_dev && b_dev && x_host && f_out && m_rows >= 1 && n >= 1, "zf_ls_eval: bad argument");
    ZF_REQUIRE(zf_aligned16(A_dev), "zf_ls_eval: A must be 16-byte aligned");
    const int V = (n % 2 == 0) ? 2 : 1;
    const int64_t nv = n / V;
    const int64_t panels = (nv + ZF_BLOCK - 1) / ZF_BLOCK;
    int64_t slices = (ZF_MAX_GRID + panels - 1) / panels;
    if (slices > (m_rows + 7) / 8) slices = (m_rows + 7) / 8;
    if (slices < 1) slices = 1;
    if (slices > 64) slices = 64;
    const int64_t rps = (m_rows + slices - 1) / slices;
    slices = (m_rows + rps - 1) / rps;
    double *x = nullptr, *s = nullptr, *slab = nullptr, *grad = nullptr, *fdev = nullptr;
    int rc = ZF_OK;
#define ZF_LS(expr)                                                               \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess && rc == ZF_OK)                                      \
            rc = zf_fail(ZF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));     \
    } while (0)
    ZF_LS(hipMalloc(&x, sizeof(double) * (n + 2)));
    ZF_LS(hipMalloc(&s, sizeof(double) * (m_rows + 2)));
    ZF_LS(hipMalloc(&fdev, sizeof(double) * 2));
    if (grad_out_host) {
        ZF_LS(hipMalloc(&slab, sizeof(double) * slices * n));
        ZF_LS(hipMalloc(&grad, sizeof(double) * n));
    }
    if (rc == ZF_OK) {
        ZF_LS(hipMemcpyAsync(x, x_host, sizeof(double) * n, hipMemcpyHostToDevice, nullptr));
        zf_ring3 xr = {{x, x, x}}, sr = {{s, s, s}};
        int gr = (int)((m_rows + GEMV_ROWS - 1) / GEMV_ROWS);
        if (gr > 8 * ZF_MAX_GRID) gr = 8 * ZF_MAX_GRID;
        if (V == 2)
            hipLaunchKernelGGL(zf_gemv_rows_kernel<2>, dim3(gr), dim3(ZF_BLOCK), 0, nullptr, nullptr, A_dev, xr, sr,
                               -1, m_rows, n);
        else
            hipLaunchKernelGGL(zf_gemv_rows_kernel<1>, dim3(gr), dim3(ZF_BLOCK), 0, nullptr, nullptr, A_dev, xr, sr,
                               -1, m_rows, n);
        hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, nullptr, nullptr, sr, -1, b_dev, scale,
                           m_rows, fdev);
        if (grad_out_host) {
            // r = s - b in place, then column sums
            hipLaunchKernelGGL(zf_axmb_kernel, dim3(zf_grid_for(m_rows)), dim3(ZF_BLOCK), 0, nullptr, s, b_dev,
                               m_rows);
            dim3 gT((unsigned)panels, (unsigned)slices);
            if (V == 2)
                hipLaunchKernelGGL(zf_gemvT_partial_kernel<2>, gT, dim3(ZF_BLOCK), 0, nullptr, nullptr, A_dev, s,
                                   slab, m_rows, n, rps);
            else
                hipLaunchKernelGGL(zf_gemvT_partial_kernel<1>, gT, dim3(ZF_BLOCK), 0, nullptr, nullptr, A_dev, s,
                                   slab, m_rows, n, rps);
            hipLaunchKernelGGL(zf_gemvT_combine_kernel, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, nullptr, nullptr,
                               slab, grad, 2 * scale, n, (int)slices);
            ZF_LS(hipMemcpyAsync(grad_out_host, grad, sizeof(double) * n, hipMemcpyDeviceToHost, nullptr));
        }
        ZF_LS(hipGetLastError());
        ZF_LS(hipMemcpyAsync(f_out, fdev, sizeof(double), hipMemcpyDeviceToHost, nullptr));
        ZF_LS(hipStreamSynchronize(nullptr));
    }
#undef ZF_LS
    for (void* p : {(void*)x, (void*)s, (void*)slab, (void*)grad, (void*)fdev})
        if (p) (void)hipFree(p);
    return rc;
}

// f(x) = scale |B W^-1 x - b|^2 and (optionally) jac_f(x) = 2 scale W B (B W^-1 x - b) of the operator problem
// (zf_kernels_op.h; reference: examples/cameraman.ipynb cell 8) for a host vector: the callbacks of
// zfista_amd.problems.BlurHaarL1 outside the device-resident loop.  taps_dev: K x K, b_dev: H x W (device).
extern "C" int zf_op_eval(const double* taps_dev, int32_t k, const double* b_dev, int64_t h, int64_t w, double scale,
                          const double* x_host, double* f_out, double* grad_out_host) {
    ZF_REQUIRE(taps_dev && b_dev && x_host && f_out, "zf_op_eval: null argument");
    ZF_REQUIRE(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && k >= 1 && k % 2 == 1 && k <= ZF_OP_MAXK && k / 2 < h && k / 2 < w,
               "zf_op_eval: the image must be even-sized and the kernel odd, at most 15 wide");
    const int64_t n = h * w;
    zf_problem_desc d;
    memset(&d, 0, sizeof(d));
    d.op_h = h;
    d.op_w = w;
    d.op_k = k;
    d.op_taps = taps_dev;
    double *x = nullptr, *sv = nullptr, *grad = nullptr, *fdev = nullptr, *opbuf = nullptr;
    zf_op_plan pl;
    const double *taps = nullptr, *sep = nullptr;
    int rc = zf_op_prepare(taps_dev, (int)k, h, w, nullptr, &pl, &opbuf, &taps, &sep);
    if (rc) return rc;
#define ZF_OP(expr)                                                               \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess && rc == ZF_OK)                                      \
            rc = zf_fail(ZF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));     \
    } while (0)
    ZF_OP(hipMalloc(&x, sizeof(double) * n));
    ZF_OP(hipMalloc(&sv, sizeof(double) * n));
    ZF_OP(hipMalloc(&fdev, sizeof(double) * 2));
    if (grad_out_host) ZF_OP(hipMalloc(&grad, sizeof(double) * n));
    if (rc == ZF_OK) {
        ZF_OP(hipMemcpyAsync(x, x_host, sizeof(double) * n, hipMemcpyHostToDevice, nullptr));
        zf_ring3 sr = {{sv, sv, sv}};
        zf_launch_op_apply(pl, nullptr, zf_op_of(d, nullptr, pl, taps, sep), x, x, x, sv, sv, sv, -1, zf_op_no_fuse());
        hipLaunchKernelGGL(zf_resid_x_kernel, dim3(1), dim3(RESID_BLOCK), 0, nullptr, nullptr, sr, -1, b_dev, scale, n, fdev);
        if (grad_out_host) {
            hipLaunchKernelGGL(zf_axmb_kernel, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, nullptr, sv, b_dev, n);
            zf_launch_op_adjoint(pl, nullptr, zf_op_of(d, nullptr, pl, taps, sep), sv, grad, 2 * scale, zf_op_no_fuse());
            ZF_OP(hipMemcpyAsync(grad_out_host, grad, sizeof(double) * n, hipMemcpyDeviceToHost, nullptr));
        }
        ZF_OP(hipGetLastError());
        ZF_OP(hipMemcpyAsync(f_out, fdev, sizeof(double), hipMemcpyDeviceToHost, nullptr));
        ZF_OP(hipStreamSynchronize(nullptr));
    }
#undef ZF_OP
    for (void* p : {(void*)x, (void*)sv, (void*)grad, (void*)fdev, (void*)opbuf})
        if (p) (void)hipFree(p);
    return rc;
}
