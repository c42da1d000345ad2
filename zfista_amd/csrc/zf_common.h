// zf_common.h - error plumbing and wave64/LDS reduction helpers (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/zfista_hip.h"

// ---- error plumbing ------------------------------------------------------
extern thread_local char zf_errbuf[512];
static inline int zf_fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(zf_errbuf, sizeof(zf_errbuf), fmt, a, b);
    return code;
}
#define ZF_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) return zf_fail(ZF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define ZF_REQUIRE(cond, msg)                                  \
    do {                                                       \
        if (!(cond)) return zf_fail(ZF_ERR_ARG, "%s%s", msg);  \
    } while (0)

// ---- launch geometry -----------------------------------------------------
// Streaming kernels: 256-thread blocks (4 waves, one per SIMD), at most 2048
// blocks (256 CUs x 8) and grid-stride beyond that (guide: Guideline 11).
constexpr int ZF_BLOCK = 256;
constexpr int ZF_MAX_GRID = 2048;
constexpr int ZF_WAVES = ZF_BLOCK / 64;

static inline int zf_grid_for(int64_t items_per_thread_units) {
    int64_t blocks = (items_per_thread_units + ZF_BLOCK - 1) / ZF_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > ZF_MAX_GRID) blocks = ZF_MAX_GRID;
    return (int)blocks;
}

// ---- hand-over of doubles between workgroups of one launch (agent scope) ----
__device__ __forceinline__ void zf_publish(double* p, double v) {   // 8-byte write-through store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double zf_consume(const double* p) {     // sc1 load, bypasses this CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- wave64 + LDS block reductions ----------------------------------------
// Fixed shuffle tree (offsets 32..1) inside the wave, lane 0 of each wave
// stages its value in LDS, thread k adds the four wave values in wave order:
// the result depends only on the launch geometry, never on timing.
__device__ __forceinline__ double zf_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double zf_wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// Reduce NS sums and NM maxima held per thread; thread k < NS+NM of the block
// returns the block total of quantity k in `out` (valid only for those threads).
template <int NS, int NM, int WAVES>
__device__ __forceinline__ void zf_block_reduce(const double (&sums)[NS], const double (&maxs)[NM > 0 ? NM : 1],
                                                double* lds /* WAVES*(NS+NM) */, double& out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double v = zf_wave_sum(sums[k]);
        if (lane == 0) lds[wave * (NS + NM) + k] = v;
    }
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        double v = zf_wave_max(maxs[k]);
        if (lane == 0) lds[wave * (NS + NM) + NS + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS + NM) {
        const int k = threadIdx.x;
        double v = lds[k];
        if (k < NS) {
            for (int w = 1; w < WAVES; ++w) v += lds[w * (NS + NM) + k];
        } else {
            for (int w = 1; w < WAVES; ++w) v = fmax(v, lds[w * (NS + NM) + k]);
        }
        out = v;
    }
}

// Wave reduction of N values per lane at once, for N = C * 2^H.  The pairing is the xor
// butterfly 32, 16, ... 1 - the tree zf_wave_sum builds for lane 0 (a + b == b + a exactly) -
// but during the first H levels each partner keeps only HALF of the values and hands the other
// half over, so a level costs N/2, N/4, ... exchanges instead of N.  On return slot q < C of
// lane L holds the wave total of the value with index
//     q + sum over levels l < H of ((L & (32 >> l)) ? N >> (l + 1) : 0);
// all lanes that agree in their top H bits hold the same totals.
template <int N, int H, bool IS_MAX, int LVL>
__device__ __forceinline__ void zf_wave_reduce_multi_level(double (&v)[N], int lane) {
    // (one template instance per level: every array index below is a compile-time constant, so
    // the values stay in registers - a rolled level loop sent them through scratch memory)
    if constexpr (LVL < 6) {
        constexpr int off = 32 >> LVL;
        if constexpr (LVL < H) {
            constexpr int half = N >> (LVL + 1);
            const bool hi = (lane & off) != 0;
#pragma unroll
            for (int q = 0; q < half; ++q) {
                const double send = hi ? v[q] : v[q + half];
                const double keep = hi ? v[q + half] : v[q];
                const double recv = __shfl_xor(send, off, 64);
                v[q] = IS_MAX ? fmax(keep, recv) : keep + recv;
            }
        } else {
            constexpr int cnt = N >> H;
#pragma unroll
            for (int q = 0; q < cnt; ++q) {
                const double recv = __shfl_xor(v[q], off, 64);
                v[q] = IS_MAX ? fmax(v[q], recv) : v[q] + recv;
            }
        }
        zf_wave_reduce_multi_level<N, H, IS_MAX, LVL + 1>(v, lane);
    }
}
template <int N, int H, bool IS_MAX>
__device__ __forceinline__ void zf_wave_reduce_multi(double (&v)[N], int lane) {
    static_assert(N % (1 << H) == 0, "N must be a multiple of 2^H");
    zf_wave_reduce_multi_level<N, H, IS_MAX, 0>(v, lane);
}
// index of the value whose total ends in slot q of lane `lane` (see above)
template <int N, int H>
__device__ __forceinline__ int zf_wave_reduce_multi_index(int q, int lane) {
    int idx = q;
#pragma unroll
    for (int lvl = 0; lvl < H; ++lvl)
        if (lane & (32 >> lvl)) idx += N >> (lvl + 1);
    return idx;
}

// soft-threshold, prox of tau*|.|:  sign(u) * max(|u| - tau, 0)
// (jaxopt.prox.prox_lasso as used at zfista/problems.py:128-135 and
// tests/test_proximal_gradient.py:61).  NaN propagates as in NumPy.
__device__ __forceinline__ double zf_soft_threshold(double u, double tau) {
    double a = fabs(u) - tau;
    a = (a < 0.0) ? 0.0 : a;
    return copysign(a, u);
}
// The same for tau >= 0 in four instructions: u - clamp(u, -tau, tau), sign of u copied onto the
// result.  |u| > tau: u -/+ tau is the one rounding of sign(u) * (|u| - tau); |u| <= tau: u - u = 0,
// signed like u, as sign(u) * 0 is; NaN and infinities propagate the same way (max/min drop a
// NaN operand, u - t restores it).  The solver checks tau >= 0 (lam >= 0, lr > 0, decay_rate > 0)
// at creation; hosts route other inputs through the general form above.
__device__ __forceinline__ double zf_soft_threshold_nn(double u, double tau) {
    const double t = fmin(fmax(u, -tau), tau);
    // copysign(u - t, u) as ONE v_bfi on the high word, in place.  (The library lowering - and the same
    // bit operations written in C - built the result in a fresh register pair, because the compiler
    // keeps m alive for the |x+| sum that follows: one v_bfi + one v_mov per element and trial.)
    const double m = u - t;
    int mhi = __double2hiint(m);
    asm("v_bfi_b32 %0, %2, %0, %1" : "+v"(mhi) : "v"(__double2hiint(u)), "s"(0x7fffffff));
    return __hiloint2double(mhi, __double2loint(m));
}
// running maximum of |x|: v_max_f64 acc, acc, |x| with no canonicalisation of the accumulator (the
// compiler re-quiets a loop-carried fmax operand once per trip: one more v_max per element pair and
// trial).  Drops a NaN x like fmax does; the accumulator is never a signalling NaN.
__device__ __forceinline__ double zf_max_abs(double acc, double x) {
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(acc), "v"(x));
    return r;
}
// np.clip(u, lo, hi) = minimum(maximum(u, lo), hi)  (jaxopt projection_box, problems.py:137)
__device__ __forceinline__ double zf_clip(double u, double lo, double hi) {
    double t = (u < lo) ? lo : u;
    return (t > hi) ? hi : t;
}
