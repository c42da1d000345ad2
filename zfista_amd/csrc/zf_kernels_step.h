// zf_kernels_step.h - the fused trial kernels of the single-objective path.
//
// One HBM pass per line-search trial - or, for the separable problem (P-diag), per CHAIN of up
// to S trials (temporal blocking, see zf_trial_kernel).  A one-trial pass reads x_k, x_{k-1},
// d, c and writes x+ : 40 B per element; y_k, grad f(y_k) and all six
// reductions live in registers.  Reference sites carried by the element body:
//   y  = x_k + beta (x_k - x_{k-1})          proximal_gradient.py:534
//   v  = y - lr * grad f(y)                  proximal_gradient.py:148
//   x+ = prox_{lr g}(v)                      proximal_gradient.py:148 (callback :13)
//   <grad f(y), x+ - y>, |x+ - y|^2, g(x+)   proximal_gradient.py:150-152
//   f(y), f(x+)                              proximal_gradient.py:140,295
//   max |x+ - y|                             proximal_gradient.py:510
// Element arithmetic is written in NumPy's evaluation order and compiled with
// -ffp-contract=off, so every x+ is bit-identical to the NumPy expression; only
// the summation order of the reductions differs.
#pragma once
#include <type_traits>

#include "zf_common.h"
#include "zf_decide.h"

// number of per-block partial quantities a trial kernel emits
// [0] f(y) raw  [1] dot  [2] ss  [3] |x+|_1  [4] f(x+) raw  [5] max
constexpr int ZF_NPART = 6;

struct zf_elem_acc {
    double fy, dot, ss, l1, fx, mx;
};

// --- element bodies --------------------------------------------------------
// The ITERATE arithmetic (y, grad, v, x+) is NumPy's, operation by operation, never contracted:
// x+ is bit-identical to the reference expression.  The six REDUCTIONS feed only the scalar
// decisions; their terms are accumulated with fused multiply-adds (one rounding instead of two
// per term - at least as accurate as the NumPy sums, whose order they do not share anyway):
//   f(y)  = 1/2 sum (d r) r   with grad = d r already at hand      (NumPy: d * (r * r))
//   f(x+) = 1/2 sum (d rn) rn  - the same formula, so f(y_{k+1}) == f(x_{k+1}) bit for bit
//           whenever y_{k+1} == x_{k+1} (no momentum), as with one f callback
// 20 fp64 operations per element and trial (25 unfused).
// RES (ZF_ACCEPT_RESOLVED, include/zfista_hip.h): the fifth sum is not f(x+) but the DIFFERENCE f(x+) - f(y), element by
// element as a difference of squares - 1/2 d (rn^2 - r^2) = 1/2 (d dx)(rn + r) with dx = x+ - y = rn - r - so that the
// acceptance test (:303) is evaluated on numbers of the size of the step, not on the difference of two sums of the size
// of F; f(x+) is reported as f(y) + that.  One more addition per element and trial; the iterate is the same arithmetic.
template <bool NESTEROV, bool BOX, bool RES = false>
__device__ __forceinline__ double zf_elem_diag(double xk, double xo, double d, double c, double beta,
                                               double lr, double tau, double lo, double hi,
                                               zf_elem_acc& a) {
    double y = xk;
    if (NESTEROV) y = xk + beta * (xk - xo);
    const double r = y - c;
    const double grad = d * r;          // jac_f = d * (y - c)
    a.fy = __builtin_fma(grad, r, a.fy);
    const double v = y - lr * grad;
    double xn = zf_soft_threshold_nn(v, tau);
    if (BOX) xn = zf_clip(xn, lo, hi);
    const double dx = xn - y;
    a.dot = __builtin_fma(grad, dx, a.dot);
    a.ss = __builtin_fma(dx, dx, a.ss);
    a.l1 += fabs(xn);
    const double rn = xn - c;
    if (RES) a.fx = __builtin_fma(d * dx, rn + r, a.fx);
    else a.fx = __builtin_fma(d * rn, rn, a.fx);
    a.mx = zf_max_abs(a.mx, dx);
    return xn;
}

// the same iterate arithmetic without the reductions: an iteration whose decision is already
// known is recomputed ("replayed") in registers - 11 fp64 operations
template <bool NESTEROV, bool BOX>
__device__ __forceinline__ double zf_elem_diag_replay(double xk, double xo, double d, double c, double beta,
                                                      double lr, double tau, double lo, double hi) {
    double y = xk;
    if (NESTEROV) y = xk + beta * (xk - xo);
    const double grad = d * (y - c);
    const double v = y - lr * grad;
    double xn = zf_soft_threshold_nn(v, tau);
    if (BOX) xn = zf_clip(xn, lo, hi);
    return xn;
}

template <bool NESTEROV, bool BOX>
__device__ __forceinline__ double zf_elem_vec(double xk, double xo, double grad, double beta, double lr,
                                              double tau, double lo, double hi, zf_elem_acc& a) {
    double y = xk;
    if (NESTEROV) y = xk + beta * (xk - xo);
    const double v = y - lr * grad;
    double xn = zf_soft_threshold_nn(v, tau);
    if (BOX) xn = zf_clip(xn, lo, hi);
    const double dx = xn - y;
    a.dot = __builtin_fma(grad, dx, a.dot);
    a.ss = __builtin_fma(dx, dx, a.ss);
    a.l1 += fabs(xn);
    a.mx = zf_max_abs(a.mx, dx);
    return xn;
}

// the same step with the variant chosen at run time (the operator problem's adjoint kernel runs it in its epilogue: one
// kernel per blur size there, not per variant of the step)
__device__ __forceinline__ double zf_elem_vec_rt(bool nesterov, bool box, double xk, double xo, double grad, double beta, double lr,
                                                 double tau, double lo, double hi, zf_elem_acc& a) {
    if (nesterov) return box ? zf_elem_vec<true, true>(xk, xo, grad, beta, lr, tau, lo, hi, a) : zf_elem_vec<true, false>(xk, xo, grad, beta, lr, tau, lo, hi, a);
    return box ? zf_elem_vec<false, true>(xk, xo, grad, beta, lr, tau, lo, hi, a) : zf_elem_vec<false, false>(xk, xo, grad, beta, lr, tau, lo, hi, a);
}

// ---------------------------------------------------------------------------
// launch geometry of the trial kernel (measured on MI355X, tools/tune_trial.hip,
// n = 1e8: 0.66 ms = 6.05 TB/s for the streaming part; a capped grid-stride loop
// ran 0.87 ms, persistent tile walks 0.67-0.71 ms depending on the device):
//   * one STREAMING workgroup per tile of ZF_TILE_UNITS = 4 x 256 consecutive
//     16-byte units (16 KiB of every stream); the dispatcher hands out tiles in
//     order, so the resident workgroups sweep each array as one contiguous
//     window (DRAM locality);
//   * all 16 loads of a thread are issued before the first use;
//   * d, c are read once per trial and x+ is written once: nontemporal (nt)
//     loads / stores keep them from displacing x_k in the caches (+8 %).
// Reduction: every workgroup stores its 6 S partials with plain stores and retires
// (no per-workgroup hand-off); a second, tiny launch - zf_finalize_kernel, up to ZF_FIN_WGS
// workgroups of 256 threads, every load independent - adds them in index order, its
// last-arriving workgroup (one ticket per finalize workgroup, guide Guideline 16 R1
// counter form) builds the scalar packs and runs the decide pass.  Measured per step (S = 1) in
// loops of 20 launches (tools/tune_trial.hip, one box): streaming + finalize 0.675 ms
// (n = 1e8) / 0.069 ms (1e7); a fused single launch with per-workgroup tickets and a
// two-level in-launch reduction 0.686 / 0.078 ms at one tile per workgroup, and
// device-dependent (0.66 - 0.71) with several tiles per workgroup; the first version's
// single-workgroup finalize with dependent load rounds 0.849 / 0.098 ms.
// ---------------------------------------------------------------------------
constexpr int ZF_TILE_U = 4;
#ifndef ZF_X_NT
#define ZF_X_NT 1   // nontemporal loads of x_k, x_{k-1} in chained passes: read once per pass (tools/tune_trial.hip: -1 %)
#endif
#ifndef ZF_S8_UB
#define ZF_S8_UB 2   // units per load batch of the 8-trial chain: half a tile, software-pipelined (4 = whole tile, no pipeline)
#endif
constexpr int ZF_TILE_UNITS = ZF_TILE_U * ZF_BLOCK;   // 16-byte units per tile
constexpr int ZF_MAX_TILES_PER_WG = 24;               // upper bound of zf_step_args.tiles_per_wg
constexpr int ZF_FIN_WGS = 48;                        // workgroups of the finalize kernel
constexpr int ZF_FIN_GROUPS = 64;                     // groups of the in-kernel finalisation (zf_pass_tail)
constexpr int ZF_FIN_CNT_STRIDE = 32;                 // unsigned words between its arrival counters (one 128-B line each)
constexpr int ZF_FIN_THREADS = 256;

typedef double zf_d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ zf_d2 zf_ld2(const zf_d2* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT> __device__ __forceinline__ void zf_st2(zf_d2* p, zf_d2 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// AGENT-COHERENT 16-byte accesses (sc1): written through / read past the L2 of the XCD, which is not coherent with the
// other seven - what a workgroup of ANOTHER kernel running at the same time stored this way is seen by a load of
// this kind as soon as the store has been acknowledged (s_waitcnt vmcnt).  The compiler does not count an asm store
// in its s_waitcnt bookkeeping; vector-memory operations retire in issue order, so its own waits only get stricter.
__device__ __forceinline__ void zf_st2_coh(zf_d2* p, zf_d2 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// LDS-DMA: one 16-byte global load per lane straight into LDS (no VGPR destination).  The wave's 64
// pieces land lane-linear at the wave-uniform byte address `lds_dst` (M0) + lane * 16.  hipcc does not
// count an asm load in its s_waitcnt bookkeeping: the caller waits with zf_wait_vm<N>() before the
// LDS is read.  M0 is compiler-reserved: saved and restored inside the statement (guide, 5 / asm notes).
template <bool NT, bool COH = false> __device__ __forceinline__ void zf_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (COH)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding (they retire in issue order)
template <int N> __device__ __forceinline__ void zf_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
typedef __attribute__((address_space(3))) void* zf_lds_ptr;
// The full 16-trial chain holds 192 VGPRs of running sums (248 in all, two waves per SIMD): no registers
// are left for a second load batch, and with one batch the fp64 pipe idles while a wave waits for HBM
// (measured: 84 % of the issue slots at the running clock).  Its loads therefore go through LDS by DMA:
// the four 16-byte pieces of the NEXT unit are in flight while the chain of the current one computes.
// Each wave reads back only what it loaded itself: no barrier.
#ifndef ZF_S16_GLDS
#define ZF_S16_GLDS 1
#endif
// Two experiments of round 3 (profiles/r03_variants_ab.json, r03_stages_ab.json; same-box A/B, n = 1e8 and 1e7):
// * a THIRD DMA stage - two units in flight ahead of the chain.  The full chain is bound by the vector pipe and does
//   not care (1.181 vs 1.191 ms at n = 1e8, 0.131 vs 0.128 at 1e7); the HBM-bound 10-trial passes of the driver's
//   K = 20 blocks do: 1.06 - 1.08 ms against 1.13 - 1.19 with two stages, +5 % on that line, twice on one box.  A
//   fourth stage: no further gain.  ON.
// * the DMA path for the 8-trial bodies that serve the short passes of a 16-chain solver: passes after a rejection
//   1.22 - 1.30 ms vs 1.16 ms with plain register loads at 3 waves per SIMD.  OFF.
#ifndef ZF_S8_GLDS
#define ZF_S8_GLDS 0   // 1: the 8-trial bodies of PART 1 load by DMA too
#endif
#ifndef ZF_GENERAL_SINGLE_LOOP
#define ZF_GENERAL_SINGLE_LOOP 1   // the general (non-full-chain) DMA bodies as one loop with run-time issue / wait selection
#endif
#ifndef ZF_GLDS_STAGES
#define ZF_GLDS_STAGES 3   // LDS stages of the DMA pipeline: units in flight ahead of the one being computed + 1
#endif
#ifndef ZF_MID_REG_MAX
#define ZF_MID_REG_MAX 10   // mid chains (PART 3) of up to this many trials load into registers (software-pipelined), longer ones by DMA
#endif
// SP = packs per pass of the solver (S <= SP): the 8-trial bodies use the DMA path only inside a 16-chain solver
template <int S, int MODE, bool HIST, bool GRAD_INLINE, int SP = S> constexpr bool zf_uses_glds() {
    if (HIST || !GRAD_INLINE || ZF_S16_GLDS == 0) return false;
    if (S >= 16) return true;
    if (MODE == 0 && SP >= 16 && S > ZF_MID_REG_MAX) return true;   // the longer mid chains (zf_pass_part: PART 3)
    if (ZF_S8_GLDS == 2) return S == 8 && SP >= 16 && MODE == 1;   // (experiment: only the replay + 8 body)
    return ZF_S8_GLDS != 0 && S == 8 && SP >= 16;
}
constexpr int ZF_GLDS_STREAMS = 4;                                   // x_k, x_{k-1}, d, c
constexpr int ZF_GLDS_STAGE_UNITS = ZF_GLDS_STREAMS * ZF_BLOCK;      // 16-byte units per stage (16 KiB)
constexpr int ZF_GLDS_NST = ZF_GLDS_STAGES;
static_assert(ZF_GLDS_NST >= 2 && ZF_GLDS_NST <= 4, "2 .. 4 stages");

constexpr int ZF_MAX_SUB = ZF_MAX_SUB_ITERS;   // trials chained per pass (temporal blocking), upper bound
// levels of the transposing wave butterfly for a chain of S = 2^h trials: slot q < 5 of lane j * (64 >> h)
// ends up with quantity q of trial j
// (S a power of two: log2 S; otherwise the largest h <= 4 with 2^h | S - the butterfly then leaves S / 2^h trials per lane group)
constexpr int zf_chain_h(int S) {
    int h = 0;
    while (h < 4 && S % (2 << h) == 0) ++h;
    return h;
}
// Branch-free MID chains (S = 16 solvers): a pass of L fresh trials with nothing lagging, ZF_MID_MIN <= L <= ZF_MID_MAX - the
// lengths a tail shared by two passes takes before max_iter (zf_fresh_len: S < left < 2 S iterations -> two passes of
// about left / 2), and the tail of fewer than S iterations itself.  One kernel per length (PART 3, template
// parameter L): the registers of a chain are 12 L running sums, so each length gets the load pipeline it has room for -
// software-pipelined register loads up to ZF_MID_REG_MAX trials, loads through LDS by DMA (no VGPR destination) above.
// Round 3 had this body for L = 10 only - the shape of the driver's K = 20 blocks - and sent 9, 11 .. 15 through the
// general body (a wave-uniform branch per trial: 23.8 instead of 20.7 VALU instructions per element and trial).
constexpr int ZF_MID_MIN = 9;
constexpr int ZF_MID_MAX = 15;
// Which shape-specific kernel the HOST launches for a pass of `nf` fresh trials behind `lag` lagging iterations (S =
// chain length of the solver) when it knows the shape: 0 the full chain; 3 (S = 16) a mid chain of exactly nf trials,
// nothing replayed; 2 (S = 16) other chains of more than S / 2 fresh trials; 1 everything else.
ZF_HD inline int zf_pass_part(int S, int lag, int nf) {
    if (lag == 0 && nf == S) return 0;
#ifndef ZF_MID_CHAIN_OFF   // (A/B builds: the general body takes these passes)
    if (S >= 16 && lag == 0 && nf >= ZF_MID_MIN && nf <= ZF_MID_MAX) return 3;
#endif
    if (S >= 16 && nf > S / 2) return 2;
    return 1;
}
// Whether the kernel PART (mid chains: of length L) runs a pass of that shape.  The general body (PART 2) also runs
// what a mid chain could: when the host does not know the shape it launches parts 0, 1, 2 and no mid chain; when it
// does, exactly one kernel - a pass is never claimed by two kernels of one step.
ZF_HD inline bool zf_pass_claims(int PART, int L, int S, int lag, int nf) {
    if (PART == 0) return lag == 0 && nf == S;
    if (PART == 3) return lag == 0 && nf == L;
    const bool longer = S >= 16 && nf > S / 2 && !(lag == 0 && nf == S);
    return PART == 2 ? longer : !longer && !(lag == 0 && nf == S);
}
#ifndef ZF_S16_UB
#define ZF_S16_UB 1   // units per load batch of the 16-trial chain (192 VGPRs of running sums leave room for one)
#endif
constexpr int ZF_MAX_RING = 6;   // 3: one trial per pass; 4: chains; 6: chains whose passes run ahead of each other's decision

// what a pass reads of the control block before its first vector load
struct zf_pass_head {
    int cur, prev, ring;
    double lr, beta_next;
    int64_t nit;
};


struct zf_step_args {
    const zf_control* ctl;
    const double* beta_ring;  // momentum ring: trial j > 0 of a chain uses beta_ring[(nit + j) % ZF_RING]
    double* xb[ZF_MAX_RING];  // x ring of ctl->ring_size buffers
    const double* p0;         // diag: d        vec: grad
    const double* p1;         // diag: c        vec: unused
    double lam, lo, hi;
    int64_t n;
    int tiles_per_wg;         // interleaved tiles per workgroup (1 .. ZF_MAX_TILES_PER_WG)
    double* blk_part;         // (S * ZF_NPART) x gridDim.x per-workgroup partials, quantity-major
    int* pass_log;            // timing only (else NULL): slot pass_slot receives pass_tag | (lag << 8) | fresh trials
    int pass_slot;
    int pass_tag;             // (launch number & 0x7fff) << 16: tells this launch's entry from one an earlier launch left in the slot
    double* hist;             // HIST kernels: ring of hist_cap iterates (n doubles each); the iterate of
    int64_t hist_cap;         //   iteration k goes to slot k % hist_cap, written by the trial that computes it
    int64_t hist_stride;      // doubles between slots (>= n, 512-B aligned)
    // In-kernel finalisation (separable problems, zf_pass_tail): the workgroup rows are reduced by last arrivers
    // inside the SAME launch - groups of fin_gsz workgroups first, then the fin_ng group rows - and the last of
    // them builds the packs and (decide) runs the decide pass: a pass is ONE launch.
    int fin_mode;             // 0: rows stored plainly, a zf_finalize_kernel launch follows (least squares); 1: in-kernel
    int fin_gsz, fin_ng;      // workgroups per group and groups: functions of the grid - hence of n - only
    double* grp_part;         // [S * ZF_NPART][fin_ng] group rows
    unsigned* fin_cnt;        // [0] arrival counter of the groups, [1 + g] of group g's workgroups (zero between launches)
    double fin_scale_f, fin_scale_g;   // f = fin_scale_f x sum, g = fin_scale_g x sum |x|
    double* pack;             // local packs out (S x ZF_PACK_LEN)
    zf_control* ctl_rw;       // the control block, writable: the decide pass of the last arriver
    int decide;               // unsharded x: decide here; sharded: the packs are gathered first (zf_decide_kernel)
    double* trace;
    int pass_seq;             // number of this step (> 0): written to ctl->pass_seq by the pass that decides in-kernel
    // the general body (PART 2) as a FALLBACK: it runs every shape that the kernel PART fb_part (mid chains: of fb_len
    // trials) launched beside it does not run; fb_part < 0: every shape.  (zf_predict_parts)
    int fb_on, fb_part, fb_len;
    // run-ahead passes (zf_runahead_kernel): a pass launched on the OTHER of two streams while its predecessor
    // is still finalising - workgroup j waits for workgroup j of the predecessor only (ra_flags), runs on the control
    // block the host EXPECTS (ra_head), and its deciding wave accepts the pass only if every pass since the last
    // one that was verified against the real block turned out as expected (ra_word)
    unsigned long long* ra_word;   // done_seq << 32 | good_seq: the last pass decided (or voided) / the last that matched its prediction
    unsigned* ra_flags;            // [grid] pass_seq of the last pass whose workgroup j has stored its iterates; [grid + (pass_seq & 3)]: poison (a wait of that pass timed out)
    int ra_wait;                   // pass_seq of the predecessor in flight on the other stream (0: nothing in flight - the block is read)
    int ra_need;                   // entry: good_seq must have reached this pass_seq (the pass whose inputs this one overwrites); 0: no condition
    unsigned ra_spin;              // polls before a wait gives up (the pass is then void)
    zf_pass_head ra_head;          // the head of this pass by the host's account (beta_next: unused - taken from the momentum ring)
    unsigned* ra_stats;            // [0] waits of run-ahead workgroups that gave up, [1] void run-ahead passes, [2] void passes AHEAD (below); polled with the control block
    // Passes AHEAD of their predecessor's decision at KERNEL granularity (zf_trial_kernel<..., AHEAD>; sharded solves and
    // grids of more than one round): the trial kernel takes its head from ra_head - the control block may be being
    // decided on the other stream - stores its row plainly (row-major, fin_mode 2) and retires; zf_tail_kernel on the
    // second stream adds the rows in zf_pass_tail's order and (unsharded) decides; sharded: all-gather, zf_decide_ahead_kernel.
    double head_stamp;             // zf_pack_stamp of the control block the host expects this pass to find
    int head_nf;                   // its fresh trials (lag is 0): the shape the deciding kernel checks the block against
    int fin_grid;                  // zf_tail_kernel: workgroups of the trial kernel whose rows it adds
    int accept_mode;               // ZF_ACCEPT_RESOLVED: the fifth row value of a trial is f(x+) - f(y) (zf_elem_diag<..., RES>): packs carry it in slot 7
};

// pass_log entry (low 16 bits; the high 16 are the launch tag): fresh trials | lagging iterations << 5 | passes << 10
// (passes: always 0 since round 5 - the field counted the passes of one launch of the withdrawn multi-pass kernel)
ZF_HD inline int zf_log_shape(int lag, int nf, int passes) { return (nf & 31) | ((lag & 31) << 5) | ((passes & 63) << 10); }

struct zf_finalize_args {
    const double* blk_part;   // (S * ZF_NPART) x nblocks (written by the trial kernel)
    int nblocks;
    int sub_iters;            // S
    double* slice_part;       // (S * ZF_NPART) x ZF_FIN_WGS
    unsigned* cnt;            // arrival counter of the finalize workgroups (zero between launches)
    double scale[ZF_NPART];   // pack[k] = scale[k] * total[k]
    const double* f_y_ext;    // least squares: f(y), f(x+) come from the GEMV side (else NULL)
    const double* f_x_ext;
    int contribute_f;         // column-sharded least squares: only rank 0 contributes the replicated f values
    int contribute_x;         // row-sharded least squares: x is replicated - only rank 0 contributes its sums
    double* pack;             // local packs out (S x ZF_PACK_LEN)
    zf_control* ctl;          // read for the early exit; written when `decide`
    int decide;               // unsharded x: run the decide pass here
    double* trace;
    const double* beta_ring;
};

// Second launch of a step.  The trial kernel left S x 6 partial rows per workgroup (S = trials
// of the chain).  Every finalize workgroup adds its slice of them (thread t takes workgroups
// t, t+256, ... of the slice: index order; the loads of one index are independent), publishes
// the slice totals write-through and takes a ticket; the last arriver's wave 0 adds the <= 48
// slices lane-parallel in slice order, builds the S packs and (decide) runs zf_decide_pass:
// model value, acceptance, lr decay, failure, termination, buffer hand-over, trace rows
// (proximal_gradient.py:149-155,:298-307,:510,:525,:539) for the trials of the chain in order.
// Deterministic: no float atomics, sums in index order.
template <int S>
__global__ __launch_bounds__(ZF_FIN_THREADS) void zf_finalize_kernel(zf_finalize_args F) {
    constexpr int NQ = S * ZF_NPART;
    constexpr int NW = ZF_FIN_THREADS / 64;
    __shared__ double lds[NW * NQ];
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    __shared__ double s_pack[ZF_MAX_SUB_ITERS * ZF_PACK_LEN];
    __shared__ double s_single[NQ];
    __shared__ int s_last;
    if (F.ctl->status != ZF_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (F.nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = blockIdx.x * per;
    int b1 = b0 + per;
    if (b1 > F.nblocks) b1 = F.nblocks;
    // GT trials (6 GT quantities) per round: bounded registers; their wave reductions run as one
    // transposing butterfly (zf_wave_reduce_multi - the same pairing as a butterfly per quantity)
    constexpr int GT = S < 8 ? S : 8;   // (registers allow 48 loads of an index in flight: 8 trials per round)
    constexpr int GH = zf_chain_h(GT);
#pragma unroll
    for (int j0 = 0; j0 < S; j0 += GT) {
        double sums[5 * GT], maxs[GT];
#pragma unroll
        for (int k = 0; k < 5 * GT; ++k) sums[k] = 0.0;
#pragma unroll
        for (int k = 0; k < GT; ++k) maxs[k] = 0.0;
        for (int b = b0 + threadIdx.x; b < b1; b += ZF_FIN_THREADS) {
            double p[ZF_NPART * GT];
#pragma unroll
            for (int k = 0; k < ZF_NPART * GT; ++k) p[k] = F.blk_part[(int64_t)(j0 * ZF_NPART + k) * F.nblocks + b];
#pragma unroll
            for (int j = 0; j < GT; ++j) {
#pragma unroll
                for (int k = 0; k < 5; ++k) sums[j * 5 + k] = sums[j * 5 + k] + p[j * ZF_NPART + k];
                maxs[j] = fmax(maxs[j], p[j * ZF_NPART + 5]);
            }
        }
        zf_wave_reduce_multi<5 * GT, GH, false>(sums, lane);
        zf_wave_reduce_multi<GT, GH, true>(maxs, lane);
        if ((lane & ((64 >> GH) - 1)) == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int idx = zf_wave_reduce_multi_index<5 * GT, GH>(q, lane);
                lds[wave * NQ + (j0 + idx / 5) * ZF_NPART + idx % 5] = sums[q];
            }
            lds[wave * NQ + (j0 + zf_wave_reduce_multi_index<GT, GH>(0, lane)) * ZF_NPART + 5] = maxs[0];
        }
    }
    __syncthreads();
    const bool single = (gridDim.x == 1);   // small problems: one finalize workgroup, no hand-over needed
    if (threadIdx.x < NQ) {   // (NQ = 6 S <= 96 quantities: threads of the first two waves)
        const int k = threadIdx.x;
        double r = lds[k];
        for (int w = 1; w < NW; ++w)
            r = (k % ZF_NPART == ZF_NPART - 1) ? fmax(r, lds[w * NQ + k]) : r + lds[w * NQ + k];
        if (single) s_single[k] = r;
        else zf_publish(F.slice_part + (int64_t)k * ZF_FIN_WGS + blockIdx.x, r);
    }
    if (single) {
        if (threadIdx.x == 0) s_last = 1;
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NQ > 64) __syncthreads();   // publishers in two waves: all published before the ticket
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(F.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == (unsigned)(gridDim.x - 1));
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(F.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = last;
        }
    }
    __syncthreads();
    if (!s_last || wave != 0) return;
    // last arriver, wave 0: lane q holds slice q (independent sc1 loads), the fixed shuffle tree
    // adds the slices in slice order, lane j keeps pack j and lane 0 writes all of them
    const int nsl = (int)gridDim.x;
    double tot[NQ];   // all 6 S loads in flight at once: one round trip
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        tot[k] = single ? (lane == 0 ? s_single[k] : 0.0)
                        : (lane < nsl) ? zf_consume(F.slice_part + (int64_t)k * ZF_FIN_WGS + lane) : 0.0;
    // one transposing butterfly for all 6 S totals (same pairing as a shuffle tree per quantity):
    // the totals of trial j end in lane j * (64 / S), which builds, keeps and stores pack j
    constexpr int SH = zf_chain_h(S);   // (all S trials at once here; the gather above went 8 trials per round)
    constexpr int LSTR = 64 >> SH;
    double sums[5 * S], maxs[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
#pragma unroll
        for (int k = 0; k < 5; ++k) sums[j * 5 + k] = tot[j * ZF_NPART + k];
        maxs[j] = tot[j * ZF_NPART + 5];
    }
    zf_wave_reduce_multi<5 * S, SH, false>(sums, lane);
    zf_wave_reduce_multi<S, SH, true>(maxs, lane);
    const int trial = lane / LSTR;   // slot q of this lane = quantity q of trial `trial`
    double pk[ZF_PACK_LEN];
    pk[ZF_PK_FY] = F.f_y_ext ? (F.contribute_f ? *F.f_y_ext : 0.0) : F.scale[0] * sums[0];
    pk[ZF_PK_DOT] = F.contribute_x ? sums[1] : 0.0;
    pk[ZF_PK_SS] = F.contribute_x ? sums[2] : 0.0;
    pk[ZF_PK_GX] = F.contribute_x ? F.scale[3] * sums[3] : 0.0;
    pk[ZF_PK_FX] = F.f_x_ext ? (F.contribute_f ? *F.f_x_ext : 0.0) : F.scale[4] * sums[4];
    pk[ZF_PK_ERR] = maxs[0];
    pk[6] = 0.0;
    pk[7] = 0.0;
    if (lane % LSTR == 0) {
#pragma unroll
        for (int k = 0; k < ZF_PACK_LEN; ++k) {
            F.pack[trial * ZF_PACK_LEN + k] = pk[k];
            s_pack[trial * ZF_PACK_LEN + k] = pk[k];
        }
    }
    // unsharded x: the decide pass right here, trial j evaluated by its lane
    if (F.decide) zf_decide_pass_wave(F.ctl, s_pack, pk, F.trace, F.beta_ring, lane, LSTR, s_pre);
}

// Finalisation INSIDE the trial launch (fin_mode 1).  Thread t < NQ of every workgroup holds row value t of its
// workgroup (quantity t % 6 of trial t / 6).  The rows are published write-through (sc1: the L2s of the XCDs are not
// coherent with each other), the workgroup takes a ticket in its GROUP of fin_gsz consecutive workgroups; the last
// arriver of a group adds the group's rows in INDEX order (thread t: quantity t; the loads are independent, the
// additions sequential), publishes the group row and takes a ticket among the fin_ng groups; the last of those adds
// the group rows in group order, builds the S packs and (decide) runs zf_decide_pass: model value, acceptance, lr
// decay, failure, termination, buffer hand-over, trace rows (proximal_gradient.py:149-155,:298-307,:510,:525,:539).
// Deterministic - sums in index order, no float atomics - and a function of the grid only, hence of n only: every
// chain length S, every process and every rank layout with equal shards reduces the same elements in the same
// order.  Most groups finish while others still compute; on the critical path are the last group's fin_gsz rows
// and the fin_ng <= 64 group rows: 2 - 3 us, against a separate finalize launch of 15 - 26 us behind a kernel
// boundary (round 2).  Every other workgroup has taken its ticket - has read the control block for the last time -
// before the one that writes it gets there.
// RA (run-ahead passes, zf_runahead_kernel): behind its row every workgroup also publishes
// that its iterates are stored (ra_flags - what workgroup j of the NEXT pass waits for); the deciding wave waits until
// the pass before this one has been decided, accepts this pass only if that one - and every pass since the last that
// was checked against the real block - went as the host expected, decides on a copy of the block fetched past the
// caches, writes it back through them and publishes done_seq / good_seq (ra_word) last.  A pass that is not accepted
// is VOID: the block stays as it is, its iterates lie in buffers nobody reads.
// rows[0 .. count) of one quantity (rows are NQ doubles apart), added in index order with a compensated sum; a maximum
// for the sixth quantity of a trial.  ONE definition for the in-kernel finalisation (zf_pass_tail) and the stand-alone
// one (zf_tail_kernel): both add the same values in the same order, bit for bit.
template <int NQ>
__device__ __forceinline__ double zf_sum_rows(const double* rows, int count, bool is_max) {
    double acc = 0.0, comp = 0.0;
    auto add = [&](double p) {
        if (is_max) {
            acc = fmax(acc, p);
        } else {
            const double tsum = acc + p;
            comp += (fabs(acc) >= fabs(p)) ? (acc - tsum) + p : (p - tsum) + acc;
            acc = tsum;
        }
    };
    if (count > 32 && count <= 64) {
        // the <= 64 group rows of the last level: ALL in flight at once - one round trip to memory (~2 us on the
        // critical path of every pass) instead of two; same additions in the same order
        double p[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) p[u] = (u < count) ? zf_consume(rows + (int64_t)u * NQ) : 0.0;
#pragma unroll
        for (int u = 0; u < 64; ++u) add(p[u]);
        return is_max ? acc : acc + comp;
    }
    for (int k0 = 0; k0 < count; k0 += 32) {
        double p[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) p[u] = (k0 + u < count) ? zf_consume(rows + (int64_t)(k0 + u) * NQ) : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) add(p[u]);
    }
    return is_max ? acc : acc + comp;
}

template <int SP, bool RA = false, int LEN = SP>   // LEN: fresh trials of a run-ahead pass (a full or a mid chain)
__device__ __forceinline__ void zf_pass_tail(const zf_step_args& A, const double v) {
    constexpr int NQ = SP * ZF_NPART;
    __shared__ int s_role;
    __shared__ double s_tot[NQ];
    __shared__ double s_pack[ZF_MAX_SUB_ITERS * ZF_PACK_LEN];
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    const int t = (int)threadIdx.x, G = (int)gridDim.x, b = (int)blockIdx.x;
    const int gsz = A.fin_gsz, ng = A.fin_ng;
    const bool grouped = gsz > 1;
    const bool is_max = (t % ZF_NPART == ZF_NPART - 1);
    // (rows are ROW-major here - the NQ values of a workgroup side by side: whole cache lines written through,
    //  not 96 masked 8-byte pieces of lines that sixteen workgroups share - and every counter has a 128-byte
    //  line of its own: 2000 tickets on counters packed into two lines serialise in the memory system while
    //  the workgroups that wait for them hold their CU slots)
    if (t < NQ) zf_publish(A.blk_part + (int64_t)b * NQ + t, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (NQ up to 96: the publishers sit in two waves)
    if (t == 0) {
        // (every wave has waited for its own stores above - the iterates of its units among them - and passed the barrier)
        if constexpr (RA) __hip_atomic_store(A.ra_flags + b, (unsigned)A.pass_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int g = grouped ? b / gsz : 0;
        unsigned* cnt = grouped ? A.fin_cnt + (1 + g) * ZF_FIN_CNT_STRIDE : A.fin_cnt;
        int members = G;
        if (grouped) members = (g + 1) * gsz <= G ? gsz : G - g * gsz;
        const unsigned tk = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int role = 0;
        if (tk == (unsigned)(members - 1)) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            role = grouped ? 1 : 2;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_role = role;
    }
    __syncthreads();
    if (s_role == 0) return;
    // rows[0 .. count) of quantity t, added in index order with a COMPENSATED sum (Neumaier): the additions at this
    // level run at the full magnitude of the total, where every plain addition costs half an ulp of F - and the
    // acceptance test (:303) subtracts two such totals.  At the resolution limit of that test rounding
    // noise in the sums is what rejects trials: with plain index-order sums of the rows a 400-iteration solve at
    // n = 1e7 took 127 rejections and ended in "Backtracking failed", with the two-level tree of round 2 about 20,
    // the reference's NumPy sums 5 in 110 iterations.  Up to 32 loads are in flight at once (one round trip to memory per
    // 32 rows: the rows are read write-through / sc1, every load misses the caches by design).
    // (the same statements as zf_sum_rows above, as a lambda: through the function template every kernel of the family
    //  was allocated 232 VGPRs - the 8-trial bodies 176 before - although the text is the same)
    auto sum_rows = [&](const double* rows, int count) -> double {   // rows: quantity t of row 0; rows are NQ apart
        double acc = 0.0, comp = 0.0;
        auto add = [&](double p) {
            if (is_max) {
                acc = fmax(acc, p);
            } else {
                const double tsum = acc + p;
                comp += (fabs(acc) >= fabs(p)) ? (acc - tsum) + p : (p - tsum) + acc;
                acc = tsum;
            }
        };
        if (count > 32 && count <= 64) {
            // the <= 64 group rows of the last level: ALL in flight at once - one round trip to memory (~2 us on the
            // critical path of every pass) instead of two; same additions in the same order
            double p[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) p[u] = (u < count) ? zf_consume(rows + (int64_t)u * NQ) : 0.0;
#pragma unroll
            for (int u = 0; u < 64; ++u) add(p[u]);
            return is_max ? acc : acc + comp;
        }
        for (int k0 = 0; k0 < count; k0 += 32) {
            double p[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) p[u] = (k0 + u < count) ? zf_consume(rows + (int64_t)(k0 + u) * NQ) : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) add(p[u]);
        }
        return is_max ? acc : acc + comp;
    };
    if (s_role == 1) {
        const int g = b / gsz, b0 = g * gsz;
        const int b1 = b0 + gsz <= G ? b0 + gsz : G;
        if (t < NQ) zf_publish(A.grp_part + (int64_t)g * NQ + t, sum_rows(A.blk_part + (int64_t)b0 * NQ + t, b1 - b0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const unsigned tk = __hip_atomic_fetch_add(A.fin_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (tk == (unsigned)(ng - 1));
            if (last) __hip_atomic_store(A.fin_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_role = last ? 2 : 0;
        }
        __syncthreads();
        if (s_role != 2) return;
    }
    if (t < NQ) s_tot[t] = grouped ? sum_rows(A.grp_part + t, ng) : sum_rows(A.blk_part + t, G);
    __syncthreads();
    if (t >= 64) return;
    // wave 0: lane j * LSTR builds, keeps and stores pack j; then the decide pass, trial j evaluated by its lane
    constexpr int SH = zf_chain_h(SP);
    constexpr int LSTR = 64 >> SH;
    const int trial = t / LSTR;
    const double* q = s_tot + trial * ZF_NPART;
    double pk[ZF_PACK_LEN];
    pk[ZF_PK_FY] = A.fin_scale_f * q[0];
    pk[ZF_PK_DOT] = q[1];
    pk[ZF_PK_SS] = q[2];
    pk[ZF_PK_GX] = A.fin_scale_g * q[3];
    pk[ZF_PK_FX] = A.fin_scale_f * q[4];
    pk[ZF_PK_ERR] = q[5];
    pk[7] = 0.0;
    if (A.accept_mode != 0) {   // (ZF_ACCEPT_RESOLVED) f(x+) - f(y) travels in slot 7; f(x+) = f(y) + that
        pk[7] = pk[ZF_PK_FX];
        pk[ZF_PK_FX] = pk[ZF_PK_FY] + pk[7];
    }
    pk[6] = A.decide ? 0.0 : zf_pack_stamp(A.ctl_rw);   // sharded x: zf_decide_kernel checks whose packs it was given
    if constexpr (RA) {
        // A SHARDED run-ahead pass (round 5): the packs go to the all-gather and zf_decide_ahead_kernel on a third stream.  They
        // carry the stamp of the block the HOST expected in front of this pass (the real block may be a decision behind) - or
        // none, when a workgroup of this pass gave up waiting for its predecessor (rows of zeros): the decide step then
        // finds packs that are not those of the block and voids the pass.
        if (!A.decide) {
            const bool poisoned = __hip_atomic_load(A.ra_flags + G + (A.pass_seq & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)A.pass_seq;
            pk[6] = poisoned ? 0.0 : A.head_stamp;
        }
    }
    if (t % LSTR == 0) {
#pragma unroll
        for (int k = 0; k < ZF_PACK_LEN; ++k) {
            A.pack[trial * ZF_PACK_LEN + k] = pk[k];
            s_pack[trial * ZF_PACK_LEN + k] = pk[k];
        }
    }
    if constexpr (RA) {
        if (!A.decide) {   // (the shape of the pass: zf_decide_ahead overwrites it if the pass turns out void)
            if (t == 0 && A.pass_log) A.pass_log[A.pass_slot] = A.pass_tag | zf_log_shape(0, LEN, 0);
            return;
        }
        constexpr int CW = (int)(sizeof(zf_control) / 8);
        static_assert(sizeof(zf_control) % 8 == 0 && CW <= 64, "one lane per word of the control block");
        unsigned long long* gw = reinterpret_cast<unsigned long long*>(const_cast<zf_control*>(A.ctl));
        unsigned long long* lw = reinterpret_cast<unsigned long long*>(A.ctl_rw);
        // the pass before this one (on the other stream) must have been decided; wave-uniform
        unsigned long long W = __hip_atomic_load(A.ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (A.ra_wait != 0) {
            for (unsigned k = 0; (int)(W >> 32) < A.ra_wait && k < A.ra_spin; ++k) {
                __builtin_amdgcn_s_sleep(2);
                W = __hip_atomic_load(A.ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ok = ((int)(W >> 32) >= A.ra_wait && (int)(unsigned)W >= A.ra_wait) ? 1 : 0;
            // (the predecessor was not decided within the limit: counted like a workgroup's wait that gave up)
            if (t == 0 && (int)(W >> 32) < A.ra_wait) __hip_atomic_fetch_add(A.ra_stats + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // a workgroup of this pass gave up waiting for its predecessor (it contributed a row of zeros)
        if (__hip_atomic_load(A.ra_flags + G + (A.pass_seq & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)A.pass_seq) ok = 0;
        zf_control* c = A.ctl_rw;
        if (ok) {
            if (t < CW) lw[t] = __hip_atomic_load(gw + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_wave_barrier();
            // belt and braces: the block IS what the host expected (by induction it is, if ok)
            ok = (c->status == ZF_RUNNING && c->pend_status == 0 && c->lag == 0 && c->nit == A.ra_head.nit &&
                  c->lr == A.ra_head.lr && c->cur == A.ra_head.cur && c->prev == A.ra_head.prev && zf_fresh_len(c) == LEN) ? 1 : 0;
        }
        ok = __builtin_amdgcn_readfirstlane(ok);
        unsigned good = (unsigned)W;
        if (ok) {
            zf_decide_pass_wave(c, s_pack, pk, A.trace, A.beta_ring, t, LSTR, s_pre);
            if (t == 0) c->pass_seq = A.pass_seq;
            __builtin_amdgcn_wave_barrier();
            if (t < CW) __hip_atomic_store(gw + t, lw[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // as expected: every trial accepted, nothing terminated - the next pass's head is the one the host predicted
            if (c->status == ZF_RUNNING && c->pend_status == 0 && c->lag == 0 && c->nit == A.ra_head.nit + LEN && c->lr == A.ra_head.lr)
                good = (unsigned)A.pass_seq;
        }
        if (t == 0) {
            // the shape of the pass is logged HERE, by the one wave that knows whether the pass counted: a void pass
            // (shape 0) is no full chain in the host's statistics; a void pass is counted for the poll as well
            if (A.pass_log) A.pass_log[A.pass_slot] = A.pass_tag | (ok ? zf_log_shape(0, LEN, 0) : 0);
            if (!ok) __hip_atomic_fetch_add(A.ra_stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0)
            __hip_atomic_store(A.ra_word, ((unsigned long long)(unsigned)A.pass_seq << 32) | good, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (A.decide) {
        zf_decide_pass_wave(A.ctl_rw, s_pack, pk, A.trace, A.beta_ring, t, LSTR, s_pre);
        // this step has had its pass: the other shape kernels of the same step find the control block already
        // decided - describing the NEXT pass - and must not run it (zf_trial_kernel)
        if (t == 0) A.ctl_rw->pass_seq = A.pass_seq;
    }
}

// The decide step of a pass that ran AHEAD of its predecessor's decision at kernel granularity (zf_trial_kernel<..., AHEAD>),
// by ONE wave: in the stand-alone finalisation of an unsharded solve (zf_tail_kernel) and behind the all-gather of a
// sharded one (zf_decide_ahead_kernel).  The trial kernel ran on the head the host expected; here - every decide step
// runs on the second stream, in pass order - the block is what the decisions so far made of it.  It matches that head:
// the pass counts and is decided as every pass is (zf_decide_pass_wave).  It does not (a chain before it broke, a
// termination, and so this pass's kernel ran on iterates that are not x_k, or left at once on ra_need): the pass is
// VOID - the block stays as it is, its iterates lie in buffers nobody reads (six buffers in ring order), and every
// later pass of the chunk is void as well; the host finds the block unchanged at its next poll and goes on from there.
// `packs_ok`: the gathered packs of ALL ranks carry the stamp of the block (sharded; unsharded: true).
// ra_word = done_seq << 32 | good_seq is what the trial kernel two passes on reads (ra_need).
struct zf_ahead_check {
    zf_pass_head head;   // what the pass's kernel ran on
    int nf;              // its fresh trials
    int seq;             // its step number
};
__device__ __forceinline__ void zf_decide_ahead(zf_control* ctl, const double* packs, const double (&pk)[ZF_PACK_LEN], double* trace,
                                                const double* beta_ring, int lane, int lstr, zf_trial_eval* lds_pre,
                                                const zf_ahead_check& H, bool packs_ok, unsigned long long* ra_word,
                                                unsigned* ra_stats, int* pass_log, int pass_slot, int pass_tag) {
    const unsigned long long W = __hip_atomic_load(ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned good = (unsigned)W;
    const bool ok = packs_ok && ctl->status == ZF_RUNNING && ctl->pend_status == 0 && ctl->lag == 0 && ctl->nit == H.head.nit &&
                    ctl->lr == H.head.lr && ctl->cur == H.head.cur && ctl->prev == H.head.prev && zf_fresh_len(ctl) == H.nf;
    if (ok) {
        zf_decide_pass_wave(ctl, packs, pk, trace, beta_ring, lane, lstr, lds_pre);
        if (lane == 0) ctl->pass_seq = H.seq;
        __builtin_amdgcn_wave_barrier();
        // as expected: every trial accepted, nothing terminated - the head of the next pass is the one the host predicted
        if (ctl->status == ZF_RUNNING && ctl->pend_status == 0 && ctl->lag == 0 && ctl->nit == H.head.nit + H.nf && ctl->lr == H.head.lr)
            good = (unsigned)H.seq;
    } else if (lane == 0) {
        __hip_atomic_fetch_add(ra_stats + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pass_log) pass_log[pass_slot] = pass_tag;   // (shape 0: the launch ran, its pass does not count - zf_collect_timing)
    }
    if (lane == 0)
        __hip_atomic_store(ra_word, ((unsigned long long)(unsigned)H.seq << 32) | good, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Stand-alone finalisation of a pass whose trial kernel stored its rows plainly (fin_mode 2: passes ahead): the SAME
// sums in the SAME order as zf_pass_tail - workgroup g of this launch is "the last arriver of group g" (the group's rows
// in index order, compensated), the last of them adds the group rows, builds the packs (stamped with the state the host
// expected) and - unsharded x - decides (zf_decide_ahead).  Grids of up to ZF_FIN_GROUPS workgroups: one workgroup adds
// all rows, as the in-kernel finalisation does.  Launched with fin_ng workgroups (1 when not grouped) of 128 threads.
template <int SP>
__global__ __launch_bounds__(128) void zf_tail_kernel(zf_step_args A) {
    constexpr int NQ = SP * ZF_NPART;
    static_assert(NQ <= 128, "one thread per row value");
    __shared__ int s_last;
    __shared__ double s_tot[NQ];
    __shared__ double s_pack[ZF_MAX_SUB_ITERS * ZF_PACK_LEN];
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    const int t = (int)threadIdx.x, G = A.fin_grid, g = (int)blockIdx.x;
    const int gsz = A.fin_gsz, ng = A.fin_ng;
    const bool grouped = gsz > 1;
    const bool is_max = (t % ZF_NPART == ZF_NPART - 1);
    if (grouped) {
        const int b0 = g * gsz;
        const int b1 = b0 + gsz <= G ? b0 + gsz : G;
        if (t < NQ) zf_publish(A.grp_part + (int64_t)g * NQ + t, zf_sum_rows<NQ>(A.blk_part + (int64_t)b0 * NQ + t, b1 - b0, is_max));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const unsigned tk = __hip_atomic_fetch_add(A.fin_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (tk == (unsigned)(ng - 1));
            if (last) __hip_atomic_store(A.fin_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
    }
    if (t < NQ) s_tot[t] = grouped ? zf_sum_rows<NQ>(A.grp_part + t, ng, is_max) : zf_sum_rows<NQ>(A.blk_part + t, G, is_max);
    __syncthreads();
    if (t >= 64) return;
    constexpr int SH = zf_chain_h(SP);
    constexpr int LSTR = 64 >> SH;
    const int trial = t / LSTR;
    const double* q = s_tot + trial * ZF_NPART;
    double pk[ZF_PACK_LEN];
    pk[ZF_PK_FY] = A.fin_scale_f * q[0];
    pk[ZF_PK_DOT] = q[1];
    pk[ZF_PK_SS] = q[2];
    pk[ZF_PK_GX] = A.fin_scale_g * q[3];
    pk[ZF_PK_FX] = A.fin_scale_f * q[4];
    pk[ZF_PK_ERR] = q[5];
    pk[7] = 0.0;
    if (A.accept_mode != 0) {   // (ZF_ACCEPT_RESOLVED) f(x+) - f(y) travels in slot 7; f(x+) = f(y) + that
        pk[7] = pk[ZF_PK_FX];
        pk[ZF_PK_FX] = pk[ZF_PK_FY] + pk[7];
    }
    pk[6] = A.decide ? 0.0 : A.head_stamp;   // sharded x: zf_decide_ahead_kernel checks whose packs it was given
    if (t % LSTR == 0) {
#pragma unroll
        for (int k = 0; k < ZF_PACK_LEN; ++k) {
            A.pack[trial * ZF_PACK_LEN + k] = pk[k];
            s_pack[trial * ZF_PACK_LEN + k] = pk[k];
        }
    }
    if (A.decide) {
        zf_ahead_check H;
        H.head = A.ra_head;
        H.nf = A.head_nf;
        H.seq = A.pass_seq;
        zf_decide_ahead(A.ctl_rw, s_pack, pk, A.trace, A.beta_ring, t, LSTR, s_pre, H, true, A.ra_word, A.ra_stats, A.pass_log,
                        A.pass_slot, A.pass_tag);
    }
}

// What a pass reads of the control block before its first vector load.  The per-pass kernels fill it with plain
// (scalar) loads - the block was written by an earlier launch; run-ahead passes take it from the host's prediction.
// a wave-uniform 64-bit value into scalar registers
__device__ __forceinline__ unsigned long long zf_uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double zf_uniform_f64(double v) {
    return __longlong_as_double((long long)zf_uniform_u64((unsigned long long)__double_as_longlong(v)));
}
// lane `i` (wave-uniform) of a double
__device__ __forceinline__ double zf_readlane_f64(double v, int i) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, i);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), i);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ zf_pass_head zf_head_of(const zf_control* c) {
    zf_pass_head h;
    h.cur = c->cur;
    h.prev = c->prev;
    h.ring = c->ring_size;
    h.lr = c->lr;
    h.beta_next = c->beta_next;
    h.nit = c->nit;
    return h;
}

// GRAD_INLINE: true  -> separable quadratic, gradient computed from d, c
//              false -> gradient vector read from HBM (least squares; S = 1 only)
// NT: nontemporal policy for the once-touched streams (p0, p1 loads, x+ stores)
// S:  FRESH trials chained per pass (temporal blocking).  For a separable f the whole recursion
//     x_{k+1} = prox(y_k - lr grad f(y_k)), y_{k+1} = x_{k+1} + beta_{k+1} (x_{k+1} - x_k)
//     is elementwise, so one pass over x_k, x_{k-1}, d, c can run the next S iterations of an
//     element in registers (each assuming the one before was accepted), emit the six reductions
//     of EVERY trial and store only the last two iterates of the chain: 48 bytes per element
//     for S iterations instead of 40 S.  zf_decide_pass (zf_decide.h) then examines the S
//     acceptance / termination tests in order on exactly the sums a one-trial pass would have
//     produced.  Iterations accepted by an earlier pass whose iterates were not stored (a chain
//     that broke in the middle: ctl->lag of them, step sizes ctl->lag_lr[]) are REPLAYED first,
//     without reductions, and the fresh trials follow at the current step size; iterates, traces
//     and decisions are bit-identical to S = 1 (tests/test_gpu_temporal.py).
// FULL: no lagging iterations and S fresh trials (known at compile time: no per-trial branch);
// otherwise `lag` replayed + `nf` < = S fresh trials, trial by trial
// MODE 0: FULL;  1: lagging iterations replayed, then S fresh trials (no per-trial branch either:
// the usual pass after a broken chain);  2: anything else (shorter fresh chains, materialise-only)
// HIST: every fresh trial also stores its iterate into the history ring (streaming return_all: the
// iterates of a chain never exist anywhere else).  A trial that turns out rejected leaves garbage in a
// slot that the retry overwrites; replayed trials wrote theirs when they were fresh.
// SP: packs per pass of the solver (rows of partials written, S <= SP: a 16-chain solver runs its short
// passes through the 8-trial bodies and leaves the packs of trials 8 .. 15 zero)
// Returns the workgroup's row: thread t < 6 S holds quantity t % 6 of fresh trial t / 6 (0 for the other threads).
// COH: the iterates are loaded and stored agent-coherently (sc1, zf_st2_coh) - a run-ahead pass reads what a kernel
// still running on another XCD has just stored (full chains through the DMA pipeline only; +0.7-1 % per pass, measured)
template <bool GRAD_INLINE, bool NESTEROV, bool BOX, bool NT, int S, int MODE, bool HIST, int SP = S, bool COH = false, bool RES = false>
__device__ __forceinline__ double zf_trial_body(const zf_step_args& A, double* lds, const zf_pass_head& HD, const int lag,
                                                const int nf, zf_d2* stage = nullptr) {
    constexpr bool FULL = (MODE == 0);          // nothing replayed, S fresh trials
    static_assert(!RES || GRAD_INLINE, "resolved differences f(x+) - f(y): the separable problem");
    static_assert(!COH || (FULL && !HIST && GRAD_INLINE && SP >= 16 && ZF_S16_GLDS != 0),
                  "coherent iterate traffic: branch-free chains of a 16-chain solver through the DMA pipeline");
    // (a run-ahead mid chain of <= ZF_MID_REG_MAX trials takes the DMA pipeline too: the coherent 16-byte load exists as DMA only)
    constexpr bool GLDS = COH || zf_uses_glds<S, MODE, HIST, GRAD_INLINE, SP>();
    const unsigned tidx = threadIdx.x;
    constexpr bool FRESH_FULL = (MODE <= 1);    // S fresh trials
    const int cur = HD.cur;
    const int prev = HD.prev;
    const int ring = HD.ring;
    const double lr = HD.lr;
    const int ntr = FULL ? S : (FRESH_FULL ? lag + S : lag + nf);   // chain length: iterates x_{b+1} .. x_{b+ntr} from (x_b, x_{b-1})
    double beta[S];
    const int64_t nit = HD.nit;
    const int64_t base = nit - lag;        // iteration count the stored iterates belong to
    // momentum factor of the trial that produces iteration i + 1: ring[i % ZF_RING]; that of the
    // first trial of the pass was resolved into the control block by the previous decide step
    beta[0] = NESTEROV ? ((FULL || lag == 0) ? HD.beta_next : A.beta_ring[nit % ZF_RING]) : 0.0;
    // (COH - a run-ahead pass: behind the atomics of its entry logic these wave-uniform loads are no longer provably
    //  unclobbered, so they become VECTOR loads and the S factors live in 2 S vector registers - the clipped full chain then
    //  needs 289 of 256; moved into scalar registers by hand they cost what they cost the per-pass kernels: none)
#pragma unroll
    for (int j = 1; j < S; ++j)
        beta[j] = NESTEROV ? (COH ? zf_uniform_f64(A.beta_ring[(nit + j) % ZF_RING]) : A.beta_ring[(nit + j) % ZF_RING]) : 0.0;
    const double tau = A.lam * lr;   // oracle: soft_threshold(x, lam * weight)
    // The lagging iterations' parameters, one iteration per LANE (lag <= 31 < 64): replay step i takes its three
    // doubles with v_readlane - no memory access inside the replay loop.  As wave-uniform scalar loads (one per step
    // and load batch) they were a dependent K$ round trip per step, amortised over the 8 element recursions of a
    // whole-tile batch and not at all over the 2 of a DMA unit (replay + 8 bodies by DMA: 2.2 ms per pass).
    double rp_lr = 0.0, rp_beta = 0.0, rp_tau = 0.0;
    if constexpr (!FULL && GRAD_INLINE) {
        const int rl = (int)(tidx & 63);
        if (rl < lag) {
            rp_lr = A.ctl->lag_lr[rl];
            rp_beta = NESTEROV ? A.beta_ring[(base + rl) % ZF_RING] : 0.0;
            rp_tau = A.lam * rp_lr;
        }
    }
    auto replay_param = [&](int i, double& b_i, double& lr_i, double& tau_i) {
        b_i = zf_readlane_f64(rp_beta, i);
        lr_i = zf_readlane_f64(rp_lr, i);
        tau_i = zf_readlane_f64(rp_tau, i);
    };
    int first, second;
    zf_free_bufs(cur, prev, ring, &first, &second);
    const double* __restrict__ xk = A.xb[cur];
    const double* __restrict__ xo = A.xb[prev];
    double* __restrict__ out_last = A.xb[ntr == 1 ? first : second];   // x_{b+ntr}
    double* __restrict__ out_prev = A.xb[first];                       // x_{b+ntr-1} when ntr >= 2
    const double* __restrict__ p0 = A.p0;
    const double* __restrict__ p1 = A.p1;
    const int64_t n = A.n;

    zf_elem_acc acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t n2 = n >> 1;  // 16-byte units
    const zf_d2* __restrict__ xk2 = reinterpret_cast<const zf_d2*>(xk);
    const zf_d2* __restrict__ xo2 = reinterpret_cast<const zf_d2*>(xo);
    const zf_d2* __restrict__ p02 = reinterpret_cast<const zf_d2*>(p0);
    const zf_d2* __restrict__ p12 = reinterpret_cast<const zf_d2*>(p1);

    // one 16-byte unit through the fresh trials (a = x_nit, o = x_{nit-1} on entry)
    auto advance = [&](zf_d2 a, zf_d2 o, zf_d2 q, zf_d2 cc, int64_t i) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (FRESH_FULL || j < nf) {
                zf_d2 r;
                if (GRAD_INLINE) {
                    r.x = zf_elem_diag<NESTEROV, BOX, RES>(a.x, o.x, q.x, cc.x, beta[j], lr, tau, A.lo, A.hi, acc[j]);
                    r.y = zf_elem_diag<NESTEROV, BOX, RES>(a.y, o.y, q.y, cc.y, beta[j], lr, tau, A.lo, A.hi, acc[j]);
                } else {
                    r.x = zf_elem_vec<NESTEROV, BOX>(a.x, o.x, q.x, beta[j], lr, tau, A.lo, A.hi, acc[j]);
                    r.y = zf_elem_vec<NESTEROV, BOX>(a.y, o.y, q.y, beta[j], lr, tau, A.lo, A.hi, acc[j]);
                }
                o = a;
                a = r;
                if (HIST)   // x_{nit+j+1} -> its slot (16-byte stores: slots are 512-B aligned)
                    zf_st2<NT>(reinterpret_cast<zf_d2*>(A.hist + ((nit + j + 1) % A.hist_cap) * A.hist_stride) + i, a);
            }
        }
        if constexpr (COH) {
            zf_st2_coh(reinterpret_cast<zf_d2*>(out_last) + i, a);
            zf_st2_coh(reinterpret_cast<zf_d2*>(out_prev) + i, o);
        } else {
            zf_st2<NT>(reinterpret_cast<zf_d2*>(out_last) + i, a);
            if (S > 1 && (FRESH_FULL || ntr >= 2)) zf_st2<NT>(reinterpret_cast<zf_d2*>(out_prev) + i, o);
        }
        // long chains: finish one unit before the next (interleaving four 8-trial chains costs
        // ~100 more VGPRs and halves the occupancy)
        if (S >= 8) __builtin_amdgcn_sched_barrier(0);
    };

    // Workgroup b owns tiles b, b + G, b + 2G, ... (G = gridDim.x, A.tiles_per_wg of them):
    // interleaved, so that at any time the resident workgroups still cover one contiguous window
    // of every stream (consecutive tiles per workgroup measured 4-8 % slower).
    const int64_t full_tiles = n2 / ZF_TILE_UNITS;
    const int64_t G = gridDim.x;
    // (tile counts fit 32 bits - n < 2^42 - and gfx9 has no scalar 64-bit ordered compare: a 64-bit tile test
    //  is a VALU compare whose scalar operands are first copied into vector registers)
    const int full_tiles32 = (int)full_tiles, G32 = (int)gridDim.x, b32 = (int)blockIdx.x;
    // UB units are loaded, then computed, at a time
#ifndef ZF_M1_UB
#define ZF_M1_UB ZF_TILE_U   // units per load batch of the replay + 8 fresh trials body: the whole tile
#endif
    // (the replay + 8 body is not software-pipelined - pipelined it spills - so it waits for every batch it loads:
    //  a whole tile per batch, 8 independent recursions per replay step, halves the waits: 1.36 -> 1.30 ms per
    //  such pass at n = 1e8, same box; the same for the body of shorter fresh chains costs clean tails 2.7 %)
    constexpr int UB = (S >= 16) ? ZF_S16_UB : (S >= 8) ? ((MODE == 1 && S == 8 && !HIST) ? ZF_M1_UB : ZF_S8_UB) : ZF_TILE_U;
    auto load_batch = [&](int64_t first_unit, zf_d2 (&a)[UB], zf_d2 (&o)[UB], zf_d2 (&q)[UB], zf_d2 (&cc)[UB]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int64_t i = first_unit + u * ZF_BLOCK;
            a[u] = zf_ld2<NT && (ZF_X_NT != 0) && (S > 1)>(xk2 + i);
            o[u] = a[u];
            if (NESTEROV) o[u] = zf_ld2<NT && (ZF_X_NT != 0) && (S > 1)>(xo2 + i);
            q[u] = zf_ld2<NT>(p02 + i);
            cc[u] = q[u];
            if (GRAD_INLINE) cc[u] = zf_ld2<NT>(p12 + i);
        }
        // keep all loads of the batch in flight: without this fence the scheduler sinks the
        // last ones below the first arithmetic to save registers
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute_batch = [&](int64_t first_unit, zf_d2 (&a)[UB], zf_d2 (&o)[UB], zf_d2 (&q)[UB], zf_d2 (&cc)[UB]) {
        if constexpr (!FULL && GRAD_INLINE) {
            // replay of the lagging iterations: trial-outer, unit-inner - 2 UB independent
            // element recursions per step, parameters are wave-uniform scalar loads
            for (int i = 0; i < lag; ++i) {
                double b_i, lr_i, tau_i;
                replay_param(i, b_i, lr_i, tau_i);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    zf_d2 r;
                    r.x = zf_elem_diag_replay<NESTEROV, BOX>(a[u].x, o[u].x, q[u].x, cc[u].x, b_i, lr_i, tau_i, A.lo, A.hi);
                    r.y = zf_elem_diag_replay<NESTEROV, BOX>(a[u].y, o[u].y, q[u].y, cc[u].y, b_i, lr_i, tau_i, A.lo, A.hi);
                    o[u] = a[u];
                    a[u] = r;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) advance(a[u], o[u], q[u], cc[u], first_unit + u * ZF_BLOCK);
    };
    if constexpr (GLDS) {
        // LDS-DMA pipeline over this workgroup's units (tile-major, ZF_TILE_U units per tile): stage k % NST holds
        // unit k and NST - 1 units are in flight ahead of the one being computed (bytes in flight per CU = waves x
        // 4 KiB x (NST - 1): one unit ahead left the pipe latency-bound at n = 1e7, where a chain takes ~1 us).
        // Per unit: read the stage into registers, start the DMA of unit k + NST - 1 into the stage read one unit
        // ago, run the chain, store the iterate(s), then wait until the DMA of unit k + 1 has landed - everything
        // issued after it may stay in flight (vector-memory operations retire in issue order: a counted vmcnt).
        constexpr int NST = ZF_GLDS_NST;
        constexpr int NL = NESTEROV ? 4 : 3;   // DMA instructions per unit
        int my_tiles = 0;
        for (int t = 0; t < A.tiles_per_wg; ++t)
            if (t * G32 + b32 < full_tiles32) my_tiles = t + 1;
        const int total = my_tiles * ZF_TILE_U;
        const int wave = tidx >> 6;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(zf_lds_ptr)stage + wave * 1024u);
        auto unit_of = [&](int k) -> int64_t {
            return ((int64_t)(k / ZF_TILE_U) * G + blockIdx.x) * ZF_TILE_UNITS + (k % ZF_TILE_U) * ZF_BLOCK + tidx;
        };
        auto issue = [&](int k) {
            const int64_t i = unit_of(k);
            const unsigned base = lds0 + (unsigned)(k % NST) * (ZF_GLDS_STAGE_UNITS * 16u);
            zf_glds16<NT && (ZF_X_NT != 0), COH>(xk2 + i, base);
            if (NESTEROV) zf_glds16<NT && (ZF_X_NT != 0), COH>(xo2 + i, base + ZF_BLOCK * 16u);
            zf_glds16<NT>(p02 + i, base + 2 * ZF_BLOCK * 16u);
            zf_glds16<NT>(p12 + i, base + 3 * ZF_BLOCK * 16u);
        };
        if (total > 0) {
            // NST - 1 units in flight before the first chain; ALL of them are waited for (one more latency per
            // workgroup, ~1 % of its time) so that the counted wait below is the same constant in every trip:
            // a wait whose count varied per trip (a switch over immediates) made the compiler restructure the
            // loop and spill the full chain's 192 running sums
#pragma unroll
            for (int k = 0; k < NST - 1; ++k)
                if (k < total) issue(k);
            zf_wait_vm<0>();
            int st = 0;
            // One trip = one unit.  The DMA of unit k + 1 must have landed before the next trip reads its stage.
            // Younger than it (vector-memory operations retire in issue order): the stores of the NST - 1 units
            // computed since it was issued and the DMAs of the NST - 2 units behind it - or, in the last NST - 1
            // trips, which have nothing left to issue, the stores alone.  (The first NST - 2 trips have fewer
            // stores behind them: their unit k + 1 landed with the prologue's wait.)  Each of the two loops has
            // ONE wait with ONE immediate; a branch between two waits inside one loop cost the full chain its
            // register allocation just like the switch.  The general bodies store one or two iterates per unit
            // (wave-uniform, fixed for the pass): they choose between two immediates - counting one store where two
            // were issued made every trip wait for a STORE to retire, a memory round trip per unit (+15-25 % on the
            // 10-trial passes of the driver's K = 20 blocks, measured).
            const bool two_stores = FULL || FRESH_FULL || ntr >= 2;   // iterate stores per unit: 2 or 1
            // AHEAD: DMAs still in flight behind the one this trip has to wait for (NST - 2 while there is something
            // to issue; in the last NST - 1 trips one fewer per trip)
            auto trip = [&](int k, auto more_c, auto ahead_c) {
                constexpr bool MORE = decltype(more_c)::value;
                constexpr int AHEAD = decltype(ahead_c)::value;
                const zf_d2* sp = stage + st * ZF_GLDS_STAGE_UNITS + tidx;
                st = (st + 1 == NST) ? 0 : st + 1;
                const zf_d2 a = sp[0];
                const zf_d2 o = NESTEROV ? sp[ZF_BLOCK] : a;
                const zf_d2 q = sp[2 * ZF_BLOCK];
                const zf_d2 cc = sp[3 * ZF_BLOCK];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MORE) issue(k + NST - 1);
                if constexpr (FULL) {
                    advance(a, o, q, cc, unit_of(k));   // the chain + the two iterate stores
                } else {
                    // the general shape: replay of the lagging iterations (parameters are wave-uniform scalar
                    // loads), then the fresh trials; one or two iterates are stored behind the DMA
                    zf_d2 a1 = a, o1 = o;
                    for (int i = 0; i < lag; ++i) {
                        double b_i, lr_i, tau_i;
                        replay_param(i, b_i, lr_i, tau_i);
                        zf_d2 r;
                        r.x = zf_elem_diag_replay<NESTEROV, BOX>(a1.x, o1.x, q.x, cc.x, b_i, lr_i, tau_i, A.lo, A.hi);
                        r.y = zf_elem_diag_replay<NESTEROV, BOX>(a1.y, o1.y, q.y, cc.y, b_i, lr_i, tau_i, A.lo, A.hi);
                        o1 = a1;
                        a1 = r;
                    }
                    advance(a1, o1, q, cc, unit_of(k));
                }
                if constexpr (FULL) {
                    zf_wait_vm<NL * AHEAD + 2 * (NST - 1)>();
                } else {
                    if (two_stores) zf_wait_vm<NL * AHEAD + 2 * (NST - 1)>();
                    else zf_wait_vm<NL * AHEAD + (NST - 1)>();
                }
            };
            const int steady = total - (NST - 1);   // trips that still have a unit to issue
            int k = 0;
#if ZF_GENERAL_SINGLE_LOOP
            if constexpr (MODE == 2) {   // (MODE 1 - replay, then S fresh trials, always two stores - takes the steady loop + peeled trips below)
                // the general bodies: ONE loop, the issue and the wait chosen at run time (wave-uniform branches; these
                // bodies branch per trial anyway).  Three instances of the 16-trial general body - a steady loop and
                // two peeled end trips - ran the driver's 10-trial passes at 1.00-1.03 ms against 0.90 with one
                // (same-box A/B against round 2, profiles/r03_vs_r02_same_box.txt).
                static_assert(NST <= 3, "the run-time waits are written for 2 or 3 stages");
#pragma unroll 1
                for (; k < total; ++k) {
                    const zf_d2* sp = stage + st * ZF_GLDS_STAGE_UNITS + tidx;
                    st = (st + 1 == NST) ? 0 : st + 1;
                    const zf_d2 a = sp[0];
                    const zf_d2 o = NESTEROV ? sp[ZF_BLOCK] : a;
                    const zf_d2 q = sp[2 * ZF_BLOCK];
                    const zf_d2 cc = sp[3 * ZF_BLOCK];
                    __builtin_amdgcn_sched_barrier(0);
                    const bool more = k < steady;
                    if (more) issue(k + NST - 1);
                    zf_d2 a1 = a, o1 = o;
                    for (int i = 0; i < lag; ++i) {
                        double b_i, lr_i, tau_i;
                        replay_param(i, b_i, lr_i, tau_i);
                        zf_d2 r;
                        r.x = zf_elem_diag_replay<NESTEROV, BOX>(a1.x, o1.x, q.x, cc.x, b_i, lr_i, tau_i, A.lo, A.hi);
                        r.y = zf_elem_diag_replay<NESTEROV, BOX>(a1.y, o1.y, q.y, cc.y, b_i, lr_i, tau_i, A.lo, A.hi);
                        o1 = a1;
                        a1 = r;
                    }
                    advance(a1, o1, q, cc, unit_of(k));
                    // (with three stages the trip before the last has no DMA behind its own either: `more` is false)
                    if (more) {
                        if (two_stores) zf_wait_vm<NL * (NST - 2) + 2 * (NST - 1)>();
                        else zf_wait_vm<NL * (NST - 2) + (NST - 1)>();
                    } else {
                        if (two_stores) zf_wait_vm<2 * (NST - 1)>();
                        else zf_wait_vm<NST - 1>();
                    }
                }
            }
#endif
#pragma unroll 1
            for (; k < steady; ++k) trip(k, std::true_type{}, std::integral_constant<int, NST - 2>{});
            // the last NST - 1 trips, one statement each: trip `left` units before the end waits with left - 2 DMAs behind its own
            if constexpr (NST >= 4)
                if (k == total - 3 && k < total) trip(k++, std::false_type{}, std::integral_constant<int, 1>{});
            if constexpr (NST >= 3)
                if (k == total - 2 && k < total) trip(k++, std::false_type{}, std::integral_constant<int, 0>{});
            if (k < total) trip(k++, std::false_type{}, std::integral_constant<int, 0>{});
        }
    } else if constexpr (UB == ZF_TILE_U || MODE == 1 || HIST || S >= 16) {
        // short chains (<= 148 VGPRs, three or more waves per SIMD): the other waves of the SIMD
        // cover a wave's load latency; all loads of a batch in flight, then its arithmetic.
        // (Also the replay + S fresh trials body of long chains and the history-recording kernels:
        //  pipelined, their registers spill.)
        for (int t = 0; t < A.tiles_per_wg; ++t) {
            const int64_t tile = (int64_t)t * G + blockIdx.x;
            if (tile >= full_tiles) break;
#pragma unroll 1
            for (int u0 = 0; u0 < ZF_TILE_U; u0 += UB) {
                const int64_t base_u = tile * ZF_TILE_UNITS + tidx + (int64_t)u0 * ZF_BLOCK;
                zf_d2 a[UB], o[UB], q[UB], cc[UB];
                load_batch(base_u, a, o, q, cc);
                compute_batch(base_u, a, o, q, cc);
            }
        }
    } else {
        // long chains run two waves per SIMD (200 VGPRs) and compute ~1400 instructions per batch:
        // software pipeline - the loads of the NEXT batch (half a tile, 8 x 16 B per thread) are
        // issued before the arithmetic of the current one, across tile boundaries too, so every
        // wave keeps memory requests in flight while it computes
        static_assert(ZF_TILE_U == 2 * UB, "two batches per tile");
        int my_tiles = 0;
        for (int t = 0; t < A.tiles_per_wg; ++t)
            if (t * G32 + b32 < full_tiles32) my_tiles = t + 1;
        if (my_tiles > 0) {
            zf_d2 a0[UB], o0[UB], q0[UB], c0[UB], a1[UB], o1[UB], q1[UB], c1[UB];
            int64_t base_u = (int64_t)blockIdx.x * ZF_TILE_UNITS + tidx;
            load_batch(base_u, a0, o0, q0, c0);
            // (the last tile is peeled: a conditional prefetch inside the loop made the register
            //  allocator keep both buffers of both paths alive - 512 VGPRs and scratch)
            for (int t = 0; t + 1 < my_tiles; ++t) {
                load_batch(base_u + UB * ZF_BLOCK, a1, o1, q1, c1);
                compute_batch(base_u, a0, o0, q0, c0);
                const int64_t next_u = ((int64_t)(t + 1) * G + blockIdx.x) * ZF_TILE_UNITS + tidx;
                load_batch(next_u, a0, o0, q0, c0);
                compute_batch(base_u + UB * ZF_BLOCK, a1, o1, q1, c1);
                base_u = next_u;
            }
            load_batch(base_u + UB * ZF_BLOCK, a1, o1, q1, c1);
            compute_batch(base_u, a0, o0, q0, c0);
            compute_batch(base_u + UB * ZF_BLOCK, a1, o1, q1, c1);
        }
    }
    // remainder of the vector (less than one tile, including an odd last element): element by
    // element, after its own tiles, by the workgroup next in the round-robin
    const int64_t rem0 = full_tiles * ZF_TILE_UNITS * 2;
    if (rem0 != n && b32 == full_tiles32 % G32) {
        for (int64_t e = rem0 + tidx; e < n; e += ZF_BLOCK) {
            double a, o;
            if constexpr (COH) {
                a = __hip_atomic_load(xk + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                o = NESTEROV ? __hip_atomic_load(xo + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a;
            } else {
                a = xk[e];
                o = NESTEROV ? xo[e] : a;
            }
            const double q = p0[e], cc = GRAD_INLINE ? p1[e] : q;
            if constexpr (!FULL && GRAD_INLINE) {
                for (int i = 0; i < lag; ++i) {
                    double b_i, lr_i, tau_i;
                    replay_param(i, b_i, lr_i, tau_i);
                    const double r = zf_elem_diag_replay<NESTEROV, BOX>(a, o, q, cc, b_i, lr_i, tau_i, A.lo, A.hi);
                    o = a;
                    a = r;
                }
            }
#pragma unroll
            for (int j = 0; j < S; ++j) {
                if (FRESH_FULL || j < nf) {
                    double r;
                    if (GRAD_INLINE)
                        r = zf_elem_diag<NESTEROV, BOX, RES>(a, o, q, cc, beta[j], lr, tau, A.lo, A.hi, acc[j]);
                    else r = zf_elem_vec<NESTEROV, BOX>(a, o, q, beta[j], lr, tau, A.lo, A.hi, acc[j]);
                    o = a;
                    a = r;
                    if (HIST) A.hist[((nit + j + 1) % A.hist_cap) * A.hist_stride + e] = a;
                }
            }
            if constexpr (COH) {
                __hip_atomic_store(out_last + e, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out_prev + e, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                out_last[e] = a;
                if (S > 1 && (FRESH_FULL || ntr >= 2)) out_prev[e] = o;
            }
        }
    }

    // workgroup partials of every fresh trial of the chain: rows j * ZF_NPART + k.  All 6 S wave
    // reductions run as ONE transposing butterfly (zf_wave_reduce_multi: same pairing, hence the
    // same bits, as a butterfly per quantity), then the four wave totals are added in wave order.
    // (an odd chain length is reduced as the next even one with a trial of zeros: without a factor of two the
    //  butterfly transposes nothing and keeps all 6 S values alive through six levels - 316 VGPRs for S = 13)
    constexpr int RS = S + (S > 1 ? (S & 1) : 0);
    constexpr int H = zf_chain_h(RS);
    constexpr int NQ = S * ZF_NPART;
    const int lane = tidx & 63, wave = tidx >> 6;
    double sums[5 * RS], maxs[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) {
        if (j < S) {
            sums[j * 5 + 0] = acc[j].fy;
            sums[j * 5 + 1] = acc[j].dot;
            sums[j * 5 + 2] = acc[j].ss;
            sums[j * 5 + 3] = acc[j].l1;
            sums[j * 5 + 4] = acc[j].fx;
            maxs[j] = acc[j].mx;
        } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) sums[j * 5 + k] = 0.0;
            maxs[j] = 0.0;
        }
    }
    zf_wave_reduce_multi<5 * RS, H, false>(sums, lane);
    zf_wave_reduce_multi<RS, H, true>(maxs, lane);
    if ((lane & ((64 >> H) - 1)) == 0) {   // (RS >> H trials per lane group: one when RS is a power of two)
#pragma unroll
        for (int q = 0; q < ((5 * RS) >> H); ++q) {
            const int idx = zf_wave_reduce_multi_index<5 * RS, H>(q, lane);
            if (idx / 5 < S) lds[wave * NQ + (idx / 5) * ZF_NPART + idx % 5] = sums[q];
        }
#pragma unroll
        for (int q = 0; q < (RS >> H); ++q) {
            const int tr = zf_wave_reduce_multi_index<RS, H>(q, lane);
            if (tr < S) lds[wave * NQ + tr * ZF_NPART + 5] = maxs[q];
        }
    }
    __syncthreads();
    double v = 0.0;   // (rows of trials S .. SP - 1: no such trial in this pass)
    if (tidx < NQ) {
        const int t = tidx;
        v = lds[t];
#pragma unroll
        for (int w = 1; w < ZF_WAVES; ++w) v = (t % ZF_NPART == 5) ? fmax(v, lds[w * NQ + t]) : v + lds[w * NQ + t];
    }
    return v;
}

// what a per-pass kernel does with its row: stored plainly for the zf_finalize_kernel launch that follows (least
// squares), or finalised - and decided - inside this launch (zf_pass_tail)
template <int SP>
__device__ __forceinline__ void zf_pass_finish(const zf_step_args& A, const double v) {
    if (A.fin_mode == 0) {
        if (threadIdx.x < SP * ZF_NPART) A.blk_part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = v;
        return;
    }
    zf_pass_tail<SP>(A, v);
}

// PART (chains only, S > 1): the bodies of a pass live in separate kernels, each of which exits at once unless the
// pass has its shape (zf_pass_claims) -
//   0: the full chain (nothing replayed, S fresh trials: the hot, branch-free body);
//   1: every other shape (replays, shorter chains, materialise-only) - for S = 16: of up to 8 fresh
//      trials, through the 8-trial bodies (pack rows of trials 8 .. 15 written as zeros);
//   2: S = 16 only: 9 .. 15 fresh trials behind any number of lagging iterations through the general 16-trial body - a
//      wave-uniform branch per trial, LDS-DMA loads;
//   3: S = 16 only: exactly L fresh trials and nothing replayed - a branch-free mid chain (ZF_MID_MIN .. ZF_MID_MAX).
// One kernel holding all bodies needs the registers of the largest plus what the compiler hoists across
// the branches (S = 16: 274 VGPRs for parts 0 + 1, ~400 for parts 1 + 2 - one wave per SIMD instead of two;
// S = 8: 207 instead of 190); every further launch costs a kernel boundary (~1.5-4 us) per pass - so the host
// launches the ONE kernel it predicts (zf_predict_parts) and all of 0, 1, 2 only when it cannot know.
// AHEAD (PART 0 and 3 of chains of 16): the pass runs AHEAD of its predecessor's decision, at kernel granularity.  Its
// head is the one the host expects (A.ra_head; nothing lagging, exactly the kernel's chain length) - the control block
// is not read at all: the decide step of the pass before may be writing it on the other stream right now.  It leaves at
// once when the pass two before it - whose inputs it would overwrite - did not go as expected (ra_need against the good
// word), stores its row plainly (row-major) and retires: zf_tail_kernel on the second stream adds the rows, and the
// deciding wave there checks the block against this head before it decides (a pass on a wrong head is VOID).
// RES: the kernels of a solver with ZF_ACCEPT_RESOLVED (zf_elem_diag<..., RES>).
template <bool GRAD_INLINE, bool NESTEROV, bool BOX, bool NT, int S, bool HIST = false, int PART = 0, int L = 0, bool AHEAD = false,
          bool RES = false>
__global__ __launch_bounds__(ZF_BLOCK) void zf_trial_kernel(zf_step_args A) {
    static_assert(GRAD_INLINE || S == 1, "temporal blocking needs an elementwise gradient");
    static_assert(PART <= 1 || S >= 16, "the third and fourth kernels exist for chains of 16 only");
    static_assert(PART != 3 || (L >= ZF_MID_MIN && L <= ZF_MID_MAX), "mid chains: ZF_MID_MIN .. ZF_MID_MAX trials");
    static_assert(!AHEAD || (S >= 16 && !HIST && (PART == 0 || PART == 3)), "passes ahead: full and mid chains of a 16-chain solver");
    __shared__ double lds[ZF_WAVES * S * ZF_NPART];
    constexpr bool GLDS = PART == 3 ? zf_uses_glds<(PART == 3 ? L : S), 0, HIST, GRAD_INLINE, S>()
                          : PART != 1 ? zf_uses_glds<S, PART == 0 ? 0 : 2, HIST, GRAD_INLINE>()
                                      : (S >= 16 && zf_uses_glds<S / 2, 1, HIST, GRAD_INLINE, S>());
    __shared__ zf_d2 stage[GLDS ? ZF_GLDS_NST * ZF_GLDS_STAGE_UNITS : 1];   // the stages of the LDS-DMA pipeline (16 KiB each)
    if constexpr (AHEAD) {
        if (A.ra_need != 0) {
            const unsigned long long W = __hip_atomic_load(A.ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(unsigned)W < A.ra_need) return;   // (decided long ago: the launch came behind an event of that decide step)
        } else if (A.ra_wait == 0) {
            // The FIRST pass of a run: nothing is in flight (the host joined its streams), so the block may be read - and must
            // be: the host's head is a prediction, and a run that starts in the middle of a chunk (behind passes of the other
            // scheme, behind a chain that broke unseen) would otherwise write its iterates over what may be the real x_k.
            // On a head that did not come true the pass leaves; its decide step finds the mismatch and voids it.
            const zf_control* c = A.ctl;
            constexpr int LEN0 = PART == 3 ? L : S;
            const zf_pass_head& q = A.ra_head;
            if (!(c->status == ZF_RUNNING && c->pend_status == 0 && c->lag == 0 && c->nit == q.nit && c->lr == q.lr && c->cur == q.cur &&
                  c->prev == q.prev && zf_fresh_len(c) == LEN0))
                return;
        }
        zf_pass_head HA = A.ra_head;
        HA.beta_next = NESTEROV ? A.beta_ring[HA.nit % ZF_RING] : 0.0;   // (zf_resolve_beta with nothing lagging)
        constexpr int LEN = PART == 3 ? L : S;
        if (A.pass_log && blockIdx.x == 0 && threadIdx.x == 0) A.pass_log[A.pass_slot] = A.pass_tag | zf_log_shape(0, LEN, 0);
        const double va = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, LEN, 0, HIST, S, false, RES>(A, lds, HA, 0, LEN, stage);
        if (threadIdx.x < S * ZF_NPART) A.blk_part[(int64_t)blockIdx.x * (S * ZF_NPART) + threadIdx.x] = va;
        return;
    }
    // wave-uniform control reads (scalar loads); written by the previous step's decide
    if (A.ctl->status != ZF_RUNNING) return;
    // an earlier kernel of THIS step ran the pass and decided it in its own launch: the control block now describes
    // the next pass, which belongs to the next step (one step = at most one pass: the trace / momentum / history
    // rings are sized by that)
    if (A.fin_mode != 0 && A.decide && A.ctl->pass_seq == A.pass_seq) return;
    const zf_pass_head HD = zf_head_of(A.ctl);
    if constexpr (S == 1) {
        if (A.pass_log && blockIdx.x == 0 && threadIdx.x == 0) A.pass_log[A.pass_slot] = A.pass_tag | zf_log_shape(0, 1, 0);
        zf_pass_finish<S>(A, zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, S, 0, HIST, S, false, RES>(A, lds, HD, 0, 1));
    } else {
        const int lag = A.ctl->lag;
        const int nf = zf_fresh_len(A.ctl);   // fresh trials of this chain (0: materialise only)
        if constexpr (PART == 2) {
            if (A.fb_on ? (A.fb_part >= 0 && zf_pass_claims(A.fb_part, A.fb_len, S, lag, nf)) : !zf_pass_claims(2, 0, S, lag, nf)) return;
        } else {
            if (!zf_pass_claims(PART, L, S, lag, nf)) return;
        }
        if (A.pass_log && blockIdx.x == 0 && threadIdx.x == 0) A.pass_log[A.pass_slot] = A.pass_tag | zf_log_shape(lag, nf, 0);
        double v;
        if constexpr (PART == 0) {
            v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, S, 0, HIST, S, false, RES>(A, lds, HD, 0, S, stage);
        } else if constexpr (PART == 3) {
            // a branch-free chain of L trials: the passes of a tail shared by two passes (the driver's K = 20 blocks:
            // 10 + 10) and the tail itself
            v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, L, 0, HIST, S, false, RES>(A, lds, HD, 0, L, stage);
        } else if constexpr (PART == 2) {
            // (nf is laundered through readfirstlane: knowing nf > S / 2 the compiler made the first
            //  trials unconditional, scheduled across them and needed 379 VGPRs instead of 227; an empty
            //  asm as the barrier gave 260)
            const int nf_opaque = __builtin_amdgcn_readfirstlane(nf);
            v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, S, 2, HIST, S, false, RES>(A, lds, HD, lag, nf_opaque, stage);
        } else if constexpr (S >= 16) {
            constexpr int SS = S / 2;
            if (lag == 0 && nf == SS) v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, SS, 0, HIST, S, false, RES>(A, lds, HD, 0, SS, stage);
            else if (nf == SS) v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, SS, 1, HIST, S, false, RES>(A, lds, HD, lag, SS, stage);
            else v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, SS, 2, HIST, S, false, RES>(A, lds, HD, lag, nf, stage);
        } else {
            if (nf == S) v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, S, 1, HIST, S, false, RES>(A, lds, HD, lag, S);
            else v = zf_trial_body<GRAD_INLINE, NESTEROV, BOX, NT, S, 2, HIST, S, false, RES>(A, lds, HD, lag, nf);
        }
        zf_pass_finish<S>(A, v);
    }
}

// A RUN-AHEAD full chain (zf_runahead_kernel).  Consecutive full-chain passes of a grid the device holds at once are launched
// alternately on two streams: pass p + 1 starts while pass p still runs - workgroup j as soon as workgroup j of pass p
// has stored its iterates (the recursion is elementwise: that is its only data dependency) - on the control block the
// host expects pass p to leave (every trial accepted, which is what happens for whole chunks once a line search has
// settled).  The finalisation of pass p (~18 us of dependent trips to memory by one workgroup), the kernel boundary and
// the ramp of pass p + 1 no longer lie between two passes.  Pass p + 1 becomes VOID - decides nothing, its iterates lie
// in buffers nobody reads (six iterate buffers: a pass never writes what its predecessor reads) - when pass p did not
// go as expected; passes p and p + 2 share a stream, so pass p + 2 starts after pass p was decided and exits at once
// (ra_need) if that went wrong.  Results are those of one launch per pass, bit for bit (tests/test_gpu_runahead.py).
// (Two waves per SIMD: the full chains by their registers, the mid chains by the 48 KiB of DMA stages + 6 KiB of sums and
//  blocks in LDS - three workgroups would need 54.3 KiB each.  Every variant, the clipped ones included, allocates what the
//  per-pass kernel of the same chain does, now that the momentum factors are moved into scalar registers by hand: see
//  zf_trial_body on COH.)
// L: 0 - the full chain; ZF_MID_MIN .. ZF_MID_MAX - a branch-free mid chain of L trials (the passes of a tail shared by two
// passes, the tail itself), through the same DMA pipeline.
template <bool NESTEROV, bool BOX, bool NT, bool RES = false, int L = 0>
__global__ __launch_bounds__(ZF_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void zf_runahead_kernel(zf_step_args A) {
    constexpr int S = ZF_MAX_SUB;
    static_assert(S == 16, "run-ahead passes are chains of a 16-chain solver");
    static_assert(L == 0 || (L >= ZF_MID_MIN && L <= ZF_MID_MAX), "a full chain, or a mid chain of ZF_MID_MIN .. ZF_MID_MAX trials");
    constexpr int LEN = L == 0 ? S : L;
    __shared__ double lds[ZF_WAVES * S * ZF_NPART];
    __shared__ zf_d2 stage[ZF_GLDS_NST * ZF_GLDS_STAGE_UNITS];
    __shared__ zf_control s_ctl;   // the deciding workgroup's copy of the control block
    __shared__ int s_go;           // 1: run; 0: leave (the whole pass does); 2: no body - a row of zeros (the predecessor is void / a wait timed out)
    zf_pass_head HD = A.ra_head;
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    if (A.ra_wait == 0) {
        // nothing is in flight: the block is what an earlier launch left - checked as every per-pass kernel checks it
        const zf_control* c = A.ctl;
        const bool same = c->status == ZF_RUNNING && c->pend_status == 0 && c->lag == 0 && c->nit == HD.nit && c->lr == HD.lr &&
                          c->cur == HD.cur && c->prev == HD.prev && zf_fresh_len(c) == LEN;
        if (!same) {   // (every workgroup finds the same) - decided: void; a pass launched behind this one must not wait for it
            if (b == 0 && threadIdx.x == 0) {
                const unsigned long long W = __hip_atomic_load(A.ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(A.ra_word, ((unsigned long long)(unsigned)A.pass_seq << 32) | (unsigned)W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(A.ra_stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    } else {
        if (threadIdx.x == 0) {
            int go = 1;
            unsigned long long W = __hip_atomic_load(A.ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (A.ra_need != 0 && (int)(unsigned)W < A.ra_need) {
                go = 0;   // the pass whose inputs this one would overwrite broke its chain (it is decided: same stream)
            } else {
                unsigned k = 0;
                for (;;) {
                    // (a limit of 0 - ZF_RUNAHEAD_SPIN_LIMIT=0, the tests' hook - gives up without looking: whether a predecessor
                    //  happens to be finished at the first look depends on the box, the counters a test reads must not)
                    if (A.ra_spin != 0 && (int)__hip_atomic_load(A.ra_flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.ra_wait) break;
                    // (the word every waiting workgroup would poll - one line, one memory channel, the one the deciding
                    //  wave of the predecessor writes to - only now and then: it matters when the predecessor is void)
                    if ((k & 15) == 15) {
                        W = __hip_atomic_load(A.ra_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((int)(W >> 32) >= A.ra_wait && (int)(unsigned)W < A.ra_wait) {   // the predecessor is void
                            go = 2;
                            break;
                        }
                    }
                    if (++k > A.ra_spin) {   // gave up: this pass is void (its deciding wave reads the poison word)
                        // (a word per pass in flight - this one, the one behind it: the slot of pass_seq & 3)
                        __hip_atomic_store(A.ra_flags + G + (A.pass_seq & 3), (unsigned)A.pass_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(A.ra_stats + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the host reads it with its next poll and stops launching run-ahead passes)
                        go = 2;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            s_go = go;
        }
        __syncthreads();
    }
    // (wave-uniform, in a scalar register: as a value out of LDS it made the chain's control flow divergent)
    const int go = A.ra_wait == 0 ? 1 : __builtin_amdgcn_readfirstlane(s_go);
    if (go == 0) {   // the whole pass leaves: void
        if (b == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(A.ra_stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const bool run = go == 1;
    HD.beta_next = NESTEROV ? zf_uniform_f64(A.beta_ring[HD.nit % ZF_RING]) : 0.0;   // (zf_resolve_beta with nothing lagging)
    double v = 0.0;
    if (run) v = zf_trial_body<true, NESTEROV, BOX, NT, LEN, 0, false, S, true, RES>(A, lds, HD, 0, LEN, stage);
    zf_step_args T = A;   // decided on the deciding workgroup's own copy of the block
    T.ctl = A.ctl_rw;
    T.ctl_rw = &s_ctl;
    zf_pass_tail<S, true, LEN>(T, v);
}

// --- f(x), g(x) at a point (initial F(x0), proximal_gradient.py:466,472) -------
// partials: [0] f raw sum  [1] |x|_1  [2] box violations
template <bool GRAD_INLINE, bool BOX>
__global__ __launch_bounds__(ZF_BLOCK) void zf_eval_kernel(const double* __restrict__ x,
                                                           const double* __restrict__ d,
                                                           const double* __restrict__ c, double lo,
                                                           double hi, int64_t n, double* partials) {
    __shared__ double lds[ZF_WAVES * 3];
    double f = 0.0, l1 = 0.0, viol = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride) {
        const double xv = x[i];
        if (GRAD_INLINE) {
            const double r = xv - c[i];
            f += d[i] * (r * r);
        }
        l1 += fabs(xv);
        if (BOX) viol += (xv < lo || xv > hi) ? 1.0 : 0.0;
    }
    const double sums[3] = {f, l1, viol};
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<3, 0, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x < 3) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// --- fixed-order reduction of per-block partials (one block) ---------------
// out[k] = scale[k] * reduce_k(partials[k][0..nblocks))   k < nq; quantity
// `max_index` (or -1) is a max, the rest are sums.
constexpr int ZF_FIN_BLOCK = 256;
__device__ __forceinline__ void zf_finalize_partials(const double* __restrict__ partials, int nblocks,
                                                     int nq, int max_index, double* lds /* waves*8 */,
                                                     double* totals /* lds, 8 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int W = ZF_FIN_BLOCK / 64;
    for (int k = 0; k < nq; ++k) {
        const bool is_max = (k == max_index);
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += ZF_FIN_BLOCK) {
            const double p = partials[(int64_t)k * nblocks + b];
            v = is_max ? fmax(v, p) : v + p;
        }
        v = is_max ? zf_wave_max(v) : zf_wave_sum(v);
        if (lane == 0) lds[wave * 8 + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < nq) {
        const int k = threadIdx.x;
        double v = lds[k];
        for (int w = 1; w < W; ++w) v = (k == max_index) ? fmax(v, lds[w * 8 + k]) : v + lds[w * 8 + k];
        totals[k] = v;
    }
    __syncthreads();
}
