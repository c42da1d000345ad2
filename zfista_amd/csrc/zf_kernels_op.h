// zf_kernels_op.h - operator-form least squares: A = B W^-1 with B a K x K correlation with symmetric
// (edge-including mirror) boundary and W one orthonormal Haar level - the image-deblurring LASSO of the
// reference's examples/cameraman.ipynb (cells 8-11):
//     f(x)     = scale |B W^-1 x - b|^2                 (notebook: scale = 1)
//     jac_f(x) = 2 scale W B (B W^-1 x - b)             (the notebook applies B itself as the adjoint)
//     g(x)     = lam |x|_1,  prox = soft threshold
// x holds the Haar coefficients [cA, cH, cV, cD] (pywt.dwt2 layout, each (H/2) x (W/2), row-major), b the
// observed H x W image.  The operator replaces the two GEMV sweeps of the dense least-squares path
// (zf_kernels_gemv.h); everything else of a trial - the residual at y by linearity from the cached A x_k,
// A x_{k-1}, the fused prox step with the gradient vector in HBM, the decide pass - is that path's.
//
// Round 5 rewrite.  Round 4's kernels computed one output pixel per thread with 2 K^2 scalar LDS reads (tile value AND
// tap) - 162 ds_read_b64 per 16 bytes of HBM traffic at 9 x 9: at 4096 x 4096 an iteration took 2.6 ms where its bytes
// need 0.2 (profiles/r05_operator_*).  Now:
//   * a workgroup owns a 64 x TY output tile (TY = 32; 8 for images of few tiles) staged in LDS with its halo; the
//     inverse Haar level is folded into the tile load of the apply kernel - one 2 x 2 block per thread: four coefficient
//     loads, four pixels - and the forward level into the epilogue of the adjoint kernel: the image never exists in memory;
//   * a thread computes R = TY / 4 vertically adjacent outputs of one column from a sliding window in REGISTERS:
//     lanes run along x, so every LDS read is conflict-free, and each value read serves up to R outputs;
//   * SEPARABLE kernels (rank 1: k = u v^T - the notebook's Gaussian window is one; detected when the solver is
//     created): a horizontal pass (8 outputs per thread from a window of 8 + K - 1 values) into a second LDS array,
//     then the vertical pass: 2 K multiply-adds per pixel instead of K^2;
//   * general kernels: K (R + K - 1) LDS reads for R K^2 multiply-adds per thread, the taps come through the scalar
//     cache (uniform addresses: s_load), not from LDS;
//   * K is a template parameter (3 .. 15, odd; other sizes are zero-padded by the host): every window index is a constant.
// Arithmetic: the Haar sums in NumPy's left-to-right order ((a + b) + c) + d, then / 2; the correlation accumulates with
// fused multiply-adds in this kernel's own order (separable: rows of the window first) - SciPy's summation order is not
// specified either; parity is to the solver's 1e-10 (measured: 1e-15), not bit for bit.
#pragma once
#include "zf_common.h"
#include "zf_decide.h"
#include "zf_kernels_step.h"   // (zf_elem_vec: the prox step of a trial, fused into the adjoint kernel's epilogue)

constexpr int ZF_OP_TX = 64;                    // output tile width of a workgroup (lanes run along x)
constexpr int ZF_OP_MAXK = 15;                  // largest supported kernel size (odd)

struct zf_op_args {
    const zf_control* ctl;   // NULL: no early exit, slot ignored (evaluation outside the solver loop)
    int H, W, K;             // image size (even), kernel size as launched (odd, 3 .. ZF_OP_MAXK; zero-padded from the caller's)
    const double* taps;      // K x K, row-major
    const double* sep;       // NULL: general kernel; else u[ZF_OP_MAXK + 1] (rows) then v[ZF_OP_MAXK + 1] (columns): taps[i][j] = u[i] v[j]
    int tiles;               // tiles of the image (0: one per workgroup of the launch)
    int xcd_bands;           // != 0: workgroups that share an XCD (blockIdx % 8: the dispatcher deals workgroups round-robin) take a
                             // contiguous band of tiles, so that the halo a tile shares with its neighbours is found in THAT L2
};

// tile of workgroup `b` of `nwg` (row-major tile order).  Banded: the bijective XCD remap of the programming guide (5.5 T1) -
// a speed choice only: any bijection is correct, and nothing relies on which XCD a workgroup really runs on.
__device__ __forceinline__ int zf_op_tile(int b, int nwg, int banded) {
    if (!banded || nwg < 16) return b;
    const int q = nwg / 8, r = nwg % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

// The fused forms (inside the solver loop: two launches per trial instead of six; F.on != 0).
// adjoint kernel: the residual at y is formed in the tile load - r = (s_k + beta (s_k - s_{k-1})) - b by linearity from
// the cached B W^-1 x_k, B W^-1 x_{k-1} - and the workgroup's share of |r|^2 (its own tile, no halo) goes to part_y.
// ... and (F.prox, round 5) the PROX STEP of the trial runs in its epilogue: the gradient of a 2 x 2 Haar block is in
// registers there - x+ = prox(y - lr grad) is formed at its four coefficients from x_k, x_{k-1} (zf_elem_vec: the same
// expression, the same bits as the separate launch), x+ is stored instead of the gradient, and the workgroup's shares of
// <grad, x+ - y>, |x+ - y|^2, |x+|_1, max|x+ - y| go to step_part.  The gradient never exists in memory: a trial is
// 48 + 40 bytes per pixel in two launches instead of 32 + 40 + 40 in three.  (A retry after a rejected trial runs the
// adjoint again - need_grad is not consulted: rejections are rare, a third kernel in every trial is not.)
// apply kernel: the workgroup's share of |s+ - b|^2 goes to part_x, and the LAST workgroup to arrive adds both in
// workgroup order, adds the partials of the prox step, builds the pack and runs the decide pass (model value,
// acceptance, lr decay, termination, buffer hand-over, trace row: proximal_gradient.py:149-155,:298-307,:510,:525,:539)
// - the tail of zf_ls_small_rows_kernel for this operator.
struct zf_op_fuse {
    int on;                   // 0: plain operator application (F's other fields unused)
    const double* b;          // observed image
    const double* sk[3];      // ring of B W^-1 x (zf_solver::sring)
    double scale, lam;
    int nesterov;
    double* part_y;           // [workgroups] shares of |r(y)|^2   (written when need_grad)
    double* part_x;           // [workgroups] shares of |s+ - b|^2
    unsigned* cnt;            // arrival counter of the fused apply kernel (zero between launches)
    const double* blk_part;   // the prox step's partials, quantity-major [6][grid_step]
    int grid_step;
    // the prox step inside the adjoint kernel
    int prox;                 // != 0: fused; blk_part then holds what THAT kernel wrote (step_part), grid_step = its grid
    int box;
    double lo, hi;
    double* xb[3];            // the iterate ring (zf_solver::xb)
    double* step_part;        // [6][workgroups of the adjoint kernel], quantities 1, 2, 3, 5 written
    int* pass_log;            // timing mode: the trial's shape entry (zf_collect_timing), else NULL
    int pass_slot, pass_tag;
    double* ls_scal;          // [0] f(y) [1] f(x+)
    double* pack;
    zf_control* ctl_rw;
    double* trace;
    const double* beta_ring;
};

// scipy.signal.correlate2d(..., boundary="symm"): the image mirrored about its edges, edge sample included.  Indices
// further out than one mirror image (tile rows below / right of a partial tile: they feed outputs outside the image
// only) are clamped into range.
__device__ __forceinline__ int zf_op_reflect(int i, int n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// geometry of a K x K kernel on 64 x TY tiles
template <int K, int TY>
struct zf_op_geo {
    static constexpr int HALF = K / 2;
    static constexpr int HP = (HALF + 1) & ~1;            // halo rounded up to even: the apply kernel loads whole 2 x 2 Haar blocks
    static constexpr int LW = ZF_OP_TX + 2 * HP, LH = TY + 2 * HP;
    static constexpr int PITCH = LW + 2;                  // doubles between tile rows
    static constexpr int R = TY / 4;                      // outputs per thread (a column strip): 256 threads = 64 columns x 4 strips
    static constexpr int TROWS = TY + 2 * HALF;           // rows the correlation reads
    static constexpr int TMP_PITCH = ZF_OP_TX + 2;
    static constexpr int TILE_DOUBLES = LH * PITCH;
    static constexpr int TMP_DOUBLES = TROWS * TMP_PITCH;
    // the next tile's loads in flight during the correlation of this one (registers: 24 in the apply kernel, up to 72 in the
    // adjoint kernel).  Blur sizes above 9 x 9 have no room for them: with the prefetch their kernels fall to one wave per
    // SIMD and lose up to half their rate (K = 15, general path, 4096 x 4096: 810 against 1 484 it/s).
    // WALK: a workgroup walks several tiles (the host launches what the device holds at once).  Blur sizes above 9 x 9 keep a
    // workgroup per tile: the loop costs their kernels a wave per SIMD (K = 13, general path, 4096 x 4096: 1 403 against 1 684 it/s).
    static constexpr bool WALK = K <= 9;
    static constexpr bool PREFETCH = WALK;
#ifndef ZF_OP_ADJ_PREFETCH
#define ZF_OP_ADJ_PREFETCH 0
#endif
    static constexpr bool PREFETCH_ADJ = PREFETCH && (ZF_OP_ADJ_PREFETCH != 0);
};

// The correlation of the staged tile: thread (c, rg) produces out[o] = (B tile)(row rg R + o, column c), o < R.
// tile[row][col] holds image pixel (oy0 - HP + row, ox0 - HP + col).  SEP: tmp is the second LDS array; the function
// contains a workgroup barrier.
template <int K, int TY, bool SEP>
__device__ __forceinline__ void zf_op_correlate(const zf_op_args& P, const double* tile, double* tmp, double (&out)[zf_op_geo<K, TY>::R]) {
    using G = zf_op_geo<K, TY>;
    constexpr int R = G::R, OFF = G::HP - G::HALF;
    const int t = (int)threadIdx.x, c = t & (ZF_OP_TX - 1), rg = t >> 6;
#pragma unroll
    for (int o = 0; o < R; ++o) out[o] = 0.0;
    if constexpr (SEP) {
        double u[K], v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u[k] = P.sep[k];
            v[k] = P.sep[ZF_OP_MAXK + 1 + k];
        }
        // horizontal pass: task (row, segment of 8 columns): a window of 8 + K - 1 values in registers
        constexpr int SEGS = ZF_OP_TX / 8, TASKS = G::TROWS * SEGS;
        for (int task = t; task < TASKS; task += ZF_BLOCK) {
            const int row = task / SEGS, seg = task % SEGS;
            const double* src = tile + (OFF + row) * G::PITCH + OFF + seg * 8;
            double win[8 + K - 1];
#pragma unroll
            for (int q = 0; q < 8 + K - 1; ++q) win[q] = src[q];
            double* dst = tmp + row * G::TMP_PITCH + seg * 8;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) acc = __builtin_fma(win[o + j], v[j], acc);
                dst[o] = acc;
            }
        }
        __syncthreads();
        // vertical pass: the column strip of this thread
        double col[R + K - 1];
#pragma unroll
        for (int q = 0; q < R + K - 1; ++q) col[q] = tmp[(rg * R + q) * G::TMP_PITCH + c];
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int i = 0; i < K; ++i) out[o] = __builtin_fma(col[o + i], u[i], out[o]);
    } else {
        const double* __restrict__ taps = P.taps;   // uniform addresses: scalar loads
        // (one tap column per trip, not unrolled: unrolled, the scheduler hoists the K (R + K - 1) window loads of all
        //  columns in front of the arithmetic - 292 VGPRs at K = 9, 512 + scratch at K = 13)
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            double col[R + K - 1];
#pragma unroll
            for (int q = 0; q < R + K - 1; ++q) col[q] = tile[(OFF + rg * R + q) * G::PITCH + OFF + c + j];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const double w = taps[i * K + j];
#pragma unroll
                for (int o = 0; o < R; ++o) out[o] = __builtin_fma(col[o + i], w, out[o]);
            }
        }
    }
}

// block total of one value per thread, in wave order (fixed shuffle tree inside a wave)
__device__ __forceinline__ double zf_op_block_sum(double v, double* s_w) {
    v = zf_wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = s_w[0];
    for (int w = 1; w < ZF_WAVES; ++w) t += s_w[w];
    return t;
}

// s = B W^-1 x.   Inside the loop (ctl given): x = ring buffer (cur + slot) % 3 and s likewise (the trial's x+: slot 1);
// else x / s as given.  F.on: see zf_op_fuse.
template <int K, int TY, bool SEP>
__global__ __launch_bounds__(ZF_BLOCK) void zf_op_apply_kernel(zf_op_args P, const double* __restrict__ x0,
                                                               const double* __restrict__ x1,
                                                               const double* __restrict__ x2, double* s0, double* s1,
                                                               double* s2, int slot, zf_op_fuse F) {
    using G = zf_op_geo<K, TY>;
    __shared__ double tile[G::TILE_DOUBLES];
    __shared__ double tmp[SEP ? G::TMP_DOUBLES : 1];
    __shared__ double s_w[ZF_WAVES];
    __shared__ double s_pack[ZF_PACK_LEN];
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    __shared__ int s_last;
    int idx = 0;
    if (P.ctl) {
        if (P.ctl->status != ZF_RUNNING) return;
        idx = (P.ctl->cur + slot) % 3;
    }
    const double* __restrict__ x = idx == 0 ? x0 : idx == 1 ? x1 : x2;
    double* __restrict__ s = idx == 0 ? s0 : idx == 1 ? s1 : s2;
    // Tiles: a workgroup takes tiles v = blockIdx.x, blockIdx.x + gridDim.x, ... of the P.tiles the image has (round 5,
    // second half: the host launches what the device holds at once).  A workgroup is three phases between barriers - tile
    // load, correlation, stores - and then, fused, two dependent trips to memory for its share and its ticket: with a
    // workgroup per tile every tile paid all of that as latency (2048 x 2048: 16 us in the life of a workgroup whose bytes
    // need 2, 0.30 of HBM at 42.6 us; 35.9 us now: profiles/r05_operator_pmc_2048x2048.json).  Now the coefficients of the
    // NEXT tile are fetched into registers before the correlation of this one, shares are stored plainly, and the
    // ticket is taken once.  Shares are indexed by v as before: the sums of the last arriver are the same sums.
    const int tiles_x = (P.W + ZF_OP_TX - 1) / ZF_OP_TX;
    const int NT = P.tiles > 0 ? P.tiles : (int)gridDim.x;
    const int hh = P.H / 2, hw = P.W / 2;
    const int64_t q = (int64_t)hh * hw;
    constexpr int BW = G::LW / 2, BH = G::LH / 2, ROUNDS = (BW * BH + ZF_BLOCK - 1) / ZF_BLOCK;
    double cf[ROUNDS][4];
    int par[ROUNDS];
    // one 2 x 2 block of W^-1 x per thread and round (tile origin and halo are even: blocks are whole); the four coefficient
    // loads of ALL rounds of a thread are issued before the first is used
    auto fetch = [&](int v) {
        const int tile_id = zf_op_tile(v, NT, P.xcd_bands);
        const int oy0 = (tile_id / tiles_x) * TY, ox0 = (tile_id % tiles_x) * ZF_OP_TX;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int k = (int)threadIdx.x + rd * ZF_BLOCK;
            const int kk = k < BW * BH ? k : 0;
            const int by = kk / BW, bx = kk % BW;
            const int r0 = zf_op_reflect(oy0 - G::HP + 2 * by, P.H), r1 = zf_op_reflect(oy0 - G::HP + 2 * by + 1, P.H);
            const int c0 = zf_op_reflect(ox0 - G::HP + 2 * bx, P.W), c1 = zf_op_reflect(ox0 - G::HP + 2 * bx + 1, P.W);
            // (mirrored blocks are whole blocks with their parities swapped; clamped ones - far outside - feed no output inside)
            const int64_t at = (int64_t)(r0 >> 1) * hw + (c0 >> 1);
            cf[rd][0] = x[at];
            cf[rd][1] = x[q + at];
            cf[rd][2] = x[2 * q + at];
            cf[rd][3] = x[3 * q + at];
            par[rd] = (r0 & 1) | ((r1 & 1) << 1) | ((c0 & 1) << 2) | ((c1 & 1) << 3);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int k = (int)threadIdx.x + rd * ZF_BLOCK;
            if (k < BW * BH) {
                const int by = k / BW, bx = k % BW;
                const double cA = cf[rd][0], cH = cf[rd][1], cV = cf[rd][2], cD = cf[rd][3];
                // pixel (r, c) of the block: (((cA +- cH) +- cV) +- cD) / 2 with the signs of its row / column parity - written
                // with factors of +-1 (exact), not as a choice among four values: the compiler turned both an indexed array
                // and nested selects into a table in scratch memory
                auto pixel = [&](int rbit, int cbit) {
                    const double sh = rbit ? -1.0 : 1.0, sv = cbit ? -1.0 : 1.0;
                    return (((cA + sh * cH) + sv * cV) + (sh * sv) * cD) / 2;
                };
                const int pr0 = par[rd] & 1, pr1 = (par[rd] >> 1) & 1, pc0 = (par[rd] >> 2) & 1, pc1 = (par[rd] >> 3) & 1;
                double* dst = tile + (2 * by) * G::PITCH + 2 * bx;
                dst[0] = pixel(pr0, pc0);
                dst[1] = pixel(pr0, pc1);
                dst[G::PITCH] = pixel(pr1, pc0);
                dst[G::PITCH + 1] = pixel(pr1, pc1);
            }
        }
    };
    const int c = threadIdx.x & (ZF_OP_TX - 1), rg = threadIdx.x >> 6;
    int v = (int)blockIdx.x;
    if (G::PREFETCH && v < NT) fetch(v);
    for (; v < NT; v += (int)gridDim.x) {
        const int tile_id = zf_op_tile(v, NT, P.xcd_bands);
        const int oy0 = (tile_id / tiles_x) * TY, ox0 = (tile_id % tiles_x) * ZF_OP_TX;
        if (!G::PREFETCH) fetch(v);
        stage();
        __syncthreads();
        if (G::PREFETCH && v + (int)gridDim.x < NT) fetch(v + (int)gridDim.x);   // in flight during the correlation
        // (the observed pixels this thread's outputs are compared with: fetched in front of the correlation too - behind it
        //  every tile waited for them with nothing left to do)
        const int ox = ox0 + c;
        double bv[G::R];
        if (F.on) {
#pragma unroll
            for (int o = 0; o < G::R; ++o) {
                const int oy = oy0 + rg * G::R + o;
                bv[o] = (oy < P.H && ox < P.W) ? F.b[(int64_t)oy * P.W + ox] : 0.0;
            }
        }
        double out[G::R];
        zf_op_correlate<K, TY, SEP>(P, tile, tmp, out);
        double sq = 0.0;
#pragma unroll
        for (int o = 0; o < G::R; ++o) {
            const int oy = oy0 + rg * G::R + o;
            if (oy < P.H && ox < P.W) {
                s[(int64_t)oy * P.W + ox] = out[o];
                if (F.on) {
                    const double rv = out[o] - bv[o];
                    sq = __builtin_fma(rv, rv, sq);
                }
            }
        }
        if (F.on) {
            const double t = zf_op_block_sum(sq, s_w);
            if (threadIdx.x == 0) zf_publish(F.part_x + v, t);
        }
        __syncthreads();   // the tile, the second array and s_w are free for the next tile
        if constexpr (!G::WALK) break;
    }
    if (!F.on) return;
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the shares of this workgroup's tiles are out)
            const unsigned tk = __hip_atomic_fetch_add(F.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (tk == (unsigned)(gridDim.x - 1));
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(F.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        // last arriver: f(y), f(x+) from the workgroup shares and the prox step's partials.  ALL 256 threads take part:
        // thread t adds shares t, t + 256, ... (independent loads, several in flight), then the fixed block tree - in
        // round 4 one wave walked the shares in chunks of 64, a dependent trip to memory per chunk: 128 trips at
        // 4096 x 4096 (8192 workgroups) were 380 of the kernel's 566 us (profiles/r05_operator_*).  Deterministic: the
        // order depends on the grid only.
        __shared__ double s_red[6][ZF_WAVES];
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // |s+ - b|^2, |r(y)|^2, dot, ss, l1, max
        // (16 shares per thread and trip, all loads of a trip issued before the first addition: 8192 workgroups are two trips,
        //  not the 24 dependent round trips of four-at-a-time - this workgroup's reduction is the tail of the whole launch)
        const int NG = NT;
        for (int g0 = tid; g0 < NG; g0 += 16 * ZF_BLOCK) {
            double px[16], py[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int g = g0 + u * ZF_BLOCK;
                px[u] = g < NG ? zf_consume(F.part_x + g) : 0.0;
                py[u] = g < NG ? F.part_y[g] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[0] += px[u];
                acc[1] += py[u];
            }
        }
        const int64_t GS = F.grid_step;
        for (int64_t g0 = tid; g0 < GS; g0 += 8 * ZF_BLOCK) {
            double q1[8], q2[8], q3[8], q5[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t g = g0 + u * ZF_BLOCK;
                const bool in = g < GS;
                q1[u] = in ? F.blk_part[1 * GS + g] : 0.0;
                q2[u] = in ? F.blk_part[2 * GS + g] : 0.0;
                q3[u] = in ? F.blk_part[3 * GS + g] : 0.0;
                q5[u] = in ? F.blk_part[5 * GS + g] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[2] += q1[u];
                acc[3] += q2[u];
                acc[4] += q3[u];
                acc[5] = fmax(acc[5], q5[u]);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double v = k == 5 ? zf_wave_max(acc[k]) : zf_wave_sum(acc[k]);
            if (lane == 0) s_red[k][wave] = v;
        }
        __syncthreads();
        if (wave != 0) return;
        double tot[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            tot[k] = s_red[k][0];
            for (int w = 1; w < ZF_WAVES; ++w) tot[k] = k == 5 ? fmax(tot[k], s_red[k][w]) : tot[k] + s_red[k][w];
        }
        const double nx = sqrt(tot[0]), ny = sqrt(tot[1]);
        const double f_x = F.scale * (nx * nx), f_y = F.scale * (ny * ny);     // np.linalg.norm(.) ** 2
        const double dot = tot[2], ss = tot[3], l1 = tot[4], mx = tot[5];
        double pk[ZF_PACK_LEN];
        // (a rejected trial leaves y as it is: the shares of |r(y)|^2 - and f(y) - are those of the trial before)
        pk[ZF_PK_FY] = f_y;
        pk[ZF_PK_DOT] = dot;
        pk[ZF_PK_SS] = ss;
        pk[ZF_PK_GX] = F.lam * l1;
        pk[ZF_PK_FX] = f_x;
        pk[ZF_PK_ERR] = mx;
        pk[6] = 0.0;
        pk[7] = 0.0;
        if (lane == 0) {
            F.ls_scal[0] = pk[ZF_PK_FY];
            F.ls_scal[1] = pk[ZF_PK_FX];
#pragma unroll
            for (int k = 0; k < ZF_PACK_LEN; ++k) {
                F.pack[k] = pk[k];
                s_pack[k] = pk[k];
            }
        }
        zf_decide_pass_wave(F.ctl_rw, s_pack, pk, F.trace, F.beta_ring, lane, 64, s_pre);
    }
}

// grad = 2 scale W (B r):  r an H x W image (the residual at y); skipped unless ctl->need_grad.  F.on: r is formed in
// the tile load from the ring of B W^-1 x (zf_op_fuse), `r` is not read.  Tiles: as in the apply kernel - a workgroup walks
// tiles v = blockIdx.x, + gridDim.x, ...; the three arrays of the NEXT tile are fetched into registers before the correlation of
// this one.
template <int K, int TY, bool SEP>
__global__ __launch_bounds__(ZF_BLOCK) void zf_op_adjoint_kernel(zf_op_args P, const double* __restrict__ r,
                                                                 double* __restrict__ grad, double two_scale, zf_op_fuse F) {
    using G = zf_op_geo<K, TY>;
    __shared__ double tile[G::TILE_DOUBLES > TY * ZF_OP_TX ? G::TILE_DOUBLES : TY * ZF_OP_TX];   // later: the blurred tile
    __shared__ double tmp[SEP ? G::TMP_DOUBLES : 1];
    __shared__ double s_w5[5][ZF_WAVES];   // the wave totals of a tile's five shares (|r|^2, <grad, dx>, |dx|^2, |x+|_1, max|dx|)
    const bool prox = F.on != 0 && F.prox != 0;
    if (P.ctl && (P.ctl->status != ZF_RUNNING || (!prox && !P.ctl->need_grad))) return;
    const double* __restrict__ sk = nullptr;
    const double* __restrict__ so = nullptr;
    double beta = 0.0;
    if (F.on) {
        const int cur = P.ctl->cur;
        sk = F.sk[cur];
        so = F.sk[(cur + 2) % 3];
        beta = F.nesterov ? P.ctl->beta_next : 0.0;
    }
    const int tiles_x = (P.W + ZF_OP_TX - 1) / ZF_OP_TX;
    const int NT = P.tiles > 0 ? P.tiles : (int)gridDim.x;
    const bool fused = F.on != 0, nest = fused && F.nesterov != 0;
    // The fused prox step (epilogue): a thread owns the 2 x 2 Haar blocks k = threadIdx.x + rd * 256 of the tile.  Their
    // coefficients of x_k, x_{k-1} are fetched at the TOP of a tile, in front of its correlation: the loads retire behind it
    // instead of standing, unhidden, at the end (fetched in the epilogue the fused kernel took as long as the two it replaced).
    constexpr int BW = ZF_OP_TX / 2, BH = TY / 2, PR = (BW * BH + ZF_BLOCK - 1) / ZF_BLOCK;
    const double* __restrict__ xk = nullptr;
    const double* __restrict__ xo = nullptr;
    double* __restrict__ xn = nullptr;
    double lr = 0.0, tau = 0.0;
    if (prox) {   // the trial's head, as the separate prox launch reads it (zf_head_of; x+ goes to the first free buffer)
        const int cur = P.ctl->cur, prev = P.ctl->prev;
        int first, second;
        zf_free_bufs(cur, prev, P.ctl->ring_size, &first, &second);
        xk = F.xb[cur];
        xo = F.xb[prev];
        xn = F.xb[first];
        lr = P.ctl->lr;
        tau = F.lam * lr;
        if (F.pass_log && blockIdx.x == 0 && threadIdx.x == 0) F.pass_log[F.pass_slot] = F.pass_tag | zf_log_shape(0, 1, 0);
    }
    const int64_t w2 = P.W / 2, q = (int64_t)(P.H / 2) * w2;
    // the rows and columns the correlation reads (lanes along x: coalesced but for the mirrored edges): NB pixels - up to 3 NB
    // loads - per thread, all of a tile in flight at once
    constexpr int OFF = G::HP - G::HALF, CW = ZF_OP_TX + 2 * G::HALF, TOTAL = G::TROWS * CW, NB = (TOTAL + ZF_BLOCK - 1) / ZF_BLOCK;
    double a0[G::PREFETCH_ADJ ? NB : 1], a1[G::PREFETCH_ADJ ? NB : 1], a2[G::PREFETCH_ADJ ? NB : 1];
    auto fetch = [&](int v) {
        const int tile_id = zf_op_tile(v, NT, P.xcd_bands);
        const int oy0 = (tile_id / tiles_x) * TY, ox0 = (tile_id % tiles_x) * ZF_OP_TX;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k = (int)threadIdx.x + u * ZF_BLOCK;
            const int kk = k < TOTAL ? k : 0;
            const int ly = kk / CW, lx = kk % CW;
            const int iy = zf_op_reflect(oy0 - G::HALF + ly, P.H), ix = zf_op_reflect(ox0 - G::HALF + lx, P.W);
            const int64_t at = (int64_t)iy * P.W + ix;
            a0[u] = fused ? sk[at] : r[at];
            a1[u] = nest ? so[at] : 0.0;
            a2[u] = fused ? F.b[at] : 0.0;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k = (int)threadIdx.x + u * ZF_BLOCK;
            if (k < TOTAL) {
                const int ly = k / CW, lx = k % CW;
                double rv = a0[u];
                if (fused) {
                    if (nest) rv = rv + beta * (rv - a1[u]);   // B W^-1 y by linearity (zf_resid_y_kernel's expression)
                    rv = rv - a2[u];
                }
                tile[(OFF + ly) * G::PITCH + OFF + lx] = rv;
            }
        }
    };
    const int c = threadIdx.x & (ZF_OP_TX - 1), rg = threadIdx.x >> 6;
    const int64_t NG = NT;
    int v = (int)blockIdx.x;
    if constexpr (G::PREFETCH_ADJ)
        if (v < NT) fetch(v);
    for (; v < NT; v += (int)gridDim.x) {
        const int tile_id = zf_op_tile(v, NT, P.xcd_bands);
        const int oy0 = (tile_id / tiles_x) * TY, ox0 = (tile_id % tiles_x) * ZF_OP_TX;
        double kv[PR][4], ov[PR][4];
        if (prox) {
#pragma unroll
            for (int rd = 0; rd < PR; ++rd) {
                const int k = (int)threadIdx.x + rd * ZF_BLOCK;
                const int by = k / BW, bx = k % BW;
                const int py = oy0 + 2 * by, px = ox0 + 2 * bx;
                const bool in = k < BW * BH && py < P.H && px < P.W;
                const int64_t at = in ? (int64_t)(py >> 1) * w2 + (px >> 1) : 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    kv[rd][u] = xk[u * q + at];
                    ov[rd][u] = nest ? xo[u * q + at] : kv[rd][u];
                }
            }
        }
        if constexpr (G::PREFETCH_ADJ) {
            stage();
        } else {   // four pixels - up to twelve loads - per thread in flight at a time, staged as they arrive
            constexpr int BATCH = 4;
            for (int k0 = threadIdx.x; k0 < TOTAL; k0 += BATCH * ZF_BLOCK) {
                double b0[BATCH], b1[BATCH], b2[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int k = k0 + u * ZF_BLOCK;
                    const int kk = k < TOTAL ? k : 0;
                    const int ly = kk / CW, lx = kk % CW;
                    const int iy = zf_op_reflect(oy0 - G::HALF + ly, P.H), ix = zf_op_reflect(ox0 - G::HALF + lx, P.W);
                    const int64_t at = (int64_t)iy * P.W + ix;
                    b0[u] = fused ? sk[at] : r[at];
                    b1[u] = nest ? so[at] : 0.0;
                    b2[u] = fused ? F.b[at] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int k = k0 + u * ZF_BLOCK;
                    if (k < TOTAL) {
                        const int ly = k / CW, lx = k % CW;
                        double rv = b0[u];
                        if (fused) {
                            if (nest) rv = rv + beta * (rv - b1[u]);
                            rv = rv - b2[u];
                        }
                        tile[(OFF + ly) * G::PITCH + OFF + lx] = rv;
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (G::PREFETCH_ADJ)
            if (v + (int)gridDim.x < NT) fetch(v + (int)gridDim.x);   // in flight during the correlation
        double sq = 0.0;
        if (F.on) {   // this tile's share of |r|^2: its own pixels (the halo belongs to the neighbours); added up with the step's shares below
#pragma unroll
            for (int o = 0; o < G::R; ++o) {
                if (oy0 + rg * G::R + o < P.H && ox0 + c < P.W) {
                    const double rv = tile[(G::HP + rg * G::R + o) * G::PITCH + G::HP + c];
                    sq = __builtin_fma(rv, rv, sq);
                }
            }
        }
        double out[G::R];
        zf_op_correlate<K, TY, SEP>(P, tile, tmp, out);
        __syncthreads();   // every read of the tile is done: it becomes the blurred tile
        double* blurred = tile;
#pragma unroll
        for (int o = 0; o < G::R; ++o)
            blurred[(rg * G::R + o) * ZF_OP_TX + c] = (oy0 + rg * G::R + o < P.H && ox0 + c < P.W) ? out[o] : 0.0;
        __syncthreads();
        // one Haar level of the tile: a thread owns 2 x 2 blocks (by, bx), bx along the lanes
        zf_elem_acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int rd = 0; rd < PR; ++rd) {
            const int k = (int)threadIdx.x + rd * ZF_BLOCK;
            const int by = k / BW, bx = k % BW;
            const int py = oy0 + 2 * by, px = ox0 + 2 * bx;
            if (k < BW * BH && py < P.H && px < P.W) {
                const double a = blurred[(2 * by) * ZF_OP_TX + 2 * bx], b = blurred[(2 * by) * ZF_OP_TX + 2 * bx + 1];
                const double cc = blurred[(2 * by + 1) * ZF_OP_TX + 2 * bx], d = blurred[(2 * by + 1) * ZF_OP_TX + 2 * bx + 1];
                const int64_t at = (int64_t)(py >> 1) * w2 + (px >> 1);
                double g[4];
                g[0] = two_scale * ((((a + b) + cc) + d) / 2);
                g[1] = two_scale * ((((a + b) - cc) - d) / 2);
                g[2] = two_scale * ((((a - b) + cc) - d) / 2);
                g[3] = two_scale * ((((a - b) - cc) + d) / 2);
                if (prox) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        xn[u * q + at] = zf_elem_vec_rt(nest, F.box != 0, kv[rd][u], ov[rd][u], g[u], beta, lr, tau, F.lo, F.hi, acc);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) grad[u * q + at] = g[u];
                }
            }
        }
        // this tile's shares: ONE reduction for all five (a barrier pair, not five: a tile is a dozen phases between barriers and
        // every one of them waits for the slowest wave) - wave totals by the fixed shuffle tree, then in wave order, as before
        if (F.on) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const double w0 = zf_wave_sum(sq), w1 = zf_wave_sum(acc.dot), w2 = zf_wave_sum(acc.ss), w3 = zf_wave_sum(acc.l1), w4 = zf_wave_max(acc.mx);
            if (lane == 0) {
                s_w5[0][wave] = w0;
                s_w5[1][wave] = w1;
                s_w5[2][wave] = w2;
                s_w5[3][wave] = w3;
                s_w5[4][wave] = w4;
            }
        }
        __syncthreads();   // (also: the blurred tile was read)
        if (F.on && threadIdx.x == 0) {
            double tot[5];
#pragma unroll
            for (int qq = 0; qq < 5; ++qq) {
                tot[qq] = s_w5[qq][0];
                for (int wv = 1; wv < ZF_WAVES; ++wv) tot[qq] = qq == 4 ? fmax(tot[qq], s_w5[qq][wv]) : tot[qq] + s_w5[qq][wv];
            }
            F.part_y[v] = tot[0];
            if (prox) {   // rows of the quantity-major table the apply kernel adds up
                F.step_part[1 * NG + v] = tot[1];
                F.step_part[2 * NG + v] = tot[2];
                F.step_part[3 * NG + v] = tot[3];
                F.step_part[5 * NG + v] = tot[4];
            }
        }
        // (s_w5 is written again behind the barriers of the next tile's staging and correlation)
        if constexpr (!G::WALK) break;
    }
}

// ---- host side: which instantiation runs a problem (zf_op_apply.hip, zf_op_adjoint.hip) ----------------------------------------------------
// K as launched: the caller's odd size (3 .. 15: each has its kernels; 1 is zero-padded to 3); ty = 32 (64 x 32 tiles)
// when that still makes 256 tiles or more, else 8 (images of few tiles: more workgroups, shorter strips)
struct zf_op_plan {
    int K;          // 3 .. 15, odd
    int ty;         // 32 or 8
    bool sep;
    int grid;
};
zf_op_plan zf_op_make_plan(int64_t h, int64_t w, int k, bool separable);
void zf_launch_op_apply(const zf_op_plan& pl, hipStream_t st, const zf_op_args& P, const double* x0, const double* x1, const double* x2,
                        double* s0, double* s1, double* s2, int slot, const zf_op_fuse& F);
void zf_launch_op_adjoint(const zf_op_plan& pl, hipStream_t st, const zf_op_args& P, const double* r, double* grad, double two_scale,
                          const zf_op_fuse& F);
bool zf_op_persist();
int zf_op_resident(const void* kernel, int* cache);
// rank-1 test of a K x K kernel (host arrays): on success u[K], v[K] with |k[i][j] - u[i] v[j]| <= 1e-14 max |k|
bool zf_op_factor_rank1(const double* taps, int k, double* u, double* v);
