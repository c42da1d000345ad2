// tune_trial.hip - sweep launch geometry / unroll / cache policy of the fused
// P-diag trial kernel at n = 1e8 (development tool, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I include tools/tune_trial.hip -o tools/tune_trial
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../zfista_amd/csrc/zf_kernels_step.h"

thread_local char zf_errbuf[512] = "";

// argument block of the experiment kernels below (the library's zf_step_args only carries
// what the shipped kernel needs)
constexpr int ZF_GROUP = 64;   // (experiments) workgroups per ticket group
struct tune_ws { double* blk_part; unsigned* grp_cnt; double* totals; };
struct tune_args {
    const zf_control* ctl;
    double* xb[3];
    const double* p0;
    const double* p1;
    double lam;
    int64_t n;
    tune_ws ws;
};
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ double2 ld2(const double2* p) {
    if (NT) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st2(double2* p, double2 v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}

// U = 16-byte units per thread per trip; BS = block size
template <int U, int BS, bool NTL, bool NTS>
__global__ __launch_bounds__(BS) void trial_v(const double* __restrict__ xk, const double* __restrict__ xo,
                                             const double* __restrict__ d, const double* __restrict__ c,
                                             double* __restrict__ xn, double beta, double lr, double lam,
                                             int64_t n, double* partials) {
    __shared__ double lds[(BS / 64) * ZF_NPART];
    const double tau = lam * lr;
    zf_elem_acc acc = {0, 0, 0, 0, 0, 0};
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * BS;
    const double2* xk2 = (const double2*)xk; const double2* xo2 = (const double2*)xo;
    const double2* d2 = (const double2*)d; const double2* c2 = (const double2*)c;
    double2* xn2 = (double2*)xn;
    int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 a[U], o[U], q[U], cc[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = ld2<false>(xk2 + i + u * stride); o[u] = ld2<false>(xo2 + i + u * stride);
            q[u] = ld2<NTL>(d2 + i + u * stride); cc[u] = ld2<NTL>(c2 + i + u * stride); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u].x = zf_elem_diag<true, false>(a[u].x, o[u].x, q[u].x, cc[u].x, beta, lr, tau, 0, 0, acc);
            r[u].y = zf_elem_diag<true, false>(a[u].y, o[u].y, q[u].y, cc[u].y, beta, lr, tau, 0, 0, acc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st2<NTS>(xn2 + i + u * stride, r[u]);
    }
    for (; i < n2; i += stride) {
        double2 a = xk2[i], o = xo2[i], q = d2[i], cc = c2[i], r;
        r.x = zf_elem_diag<true, false>(a.x, o.x, q.x, cc.x, beta, lr, tau, 0, 0, acc);
        r.y = zf_elem_diag<true, false>(a.y, o.y, q.y, cc.y, beta, lr, tau, 0, 0, acc);
        xn2[i] = r;
    }
    const double sums[5] = {acc.fy, acc.dot, acc.ss, acc.l1, acc.fx};
    const double maxs[1] = {acc.mx};
    double out = 0.0;
    zf_block_reduce<5, 1, BS / 64>(sums, maxs, lds, out);
    if (threadIdx.x < ZF_NPART) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// block-contiguous: a block owns U consecutive tiles of BS 16-B units; MASK bits: 1 = d,c nt; 2 = xo nt; 4 = xk nt; 8 = store nt
template <int U, int BS, int MASK>
__global__ __launch_bounds__(BS) void trial_c(const double* __restrict__ xk, const double* __restrict__ xo,
                                             const double* __restrict__ d, const double* __restrict__ c,
                                             double* __restrict__ xn, double beta, double lr, double lam,
                                             int64_t n, double* partials) {
    __shared__ double lds[(BS / 64) * ZF_NPART];
    const double tau = lam * lr;
    zf_elem_acc acc = {0, 0, 0, 0, 0, 0};
    const int64_t n2 = n >> 1;
    const double2* xk2 = (const double2*)xk; const double2* xo2 = (const double2*)xo;
    const double2* d2 = (const double2*)d; const double2* c2 = (const double2*)c;
    double2* xn2 = (double2*)xn;
    const int64_t base = (int64_t)blockIdx.x * (U * BS) + threadIdx.x;
    if (base + (U - 1) * BS < n2) {
        double2 a[U], o[U], q[U], cc[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = ld2<(MASK & 4) != 0>(xk2 + base + u * BS); o[u] = ld2<(MASK & 2) != 0>(xo2 + base + u * BS);
            q[u] = ld2<(MASK & 1) != 0>(d2 + base + u * BS); cc[u] = ld2<(MASK & 1) != 0>(c2 + base + u * BS); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u].x = zf_elem_diag<true, false>(a[u].x, o[u].x, q[u].x, cc[u].x, beta, lr, tau, 0, 0, acc);
            r[u].y = zf_elem_diag<true, false>(a[u].y, o[u].y, q[u].y, cc[u].y, beta, lr, tau, 0, 0, acc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st2<(MASK & 8) != 0>(xn2 + base + u * BS, r[u]);
    } else {
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * BS;
            if (i < n2) {
                double2 a = xk2[i], o = xo2[i], q = d2[i], cc = c2[i], r;
                r.x = zf_elem_diag<true, false>(a.x, o.x, q.x, cc.x, beta, lr, tau, 0, 0, acc);
                r.y = zf_elem_diag<true, false>(a.y, o.y, q.y, cc.y, beta, lr, tau, 0, 0, acc);
                xn2[i] = r;
            }
        }
    }
    const double sums[5] = {acc.fy, acc.dot, acc.ss, acc.l1, acc.fx};
    const double maxs[1] = {acc.mx};
    double out = 0.0;
    zf_block_reduce<5, 1, BS / 64>(sums, maxs, lds, out);
    if (threadIdx.x < ZF_NPART) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// persistent: workgroup b walks tiles b, b+G, b+2G, ... (tile = U*BS consecutive units); PF = prefetch next tile
template <int U, int BS, int MASK, bool PF>
__global__ __launch_bounds__(BS) void trial_p(const double* __restrict__ xk, const double* __restrict__ xo,
                                             const double* __restrict__ d, const double* __restrict__ c,
                                             double* __restrict__ xn, double beta, double lr, double lam,
                                             int64_t n, double* partials) {
    __shared__ double lds[(BS / 64) * ZF_NPART];
    const double tau = lam * lr;
    zf_elem_acc acc = {0, 0, 0, 0, 0, 0};
    const int64_t n2 = n >> 1;
    const int64_t ntiles = n2 / (U * BS);   // ragged tail ignored in this tool
    const double2* xk2 = (const double2*)xk; const double2* xo2 = (const double2*)xo;
    const double2* d2 = (const double2*)d; const double2* c2 = (const double2*)c;
    double2* xn2 = (double2*)xn;
    double2 a[U], o[U], q[U], cc[U], r[U];
    double2 a1[U], o1[U], q1[U], c1[U];
    int64_t t = blockIdx.x;
    if (PF && t < ntiles) {
        const int64_t base = t * (U * BS) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) { a1[u] = ld2<(MASK & 4) != 0>(xk2 + base + u * BS); o1[u] = ld2<(MASK & 2) != 0>(xo2 + base + u * BS);
            q1[u] = ld2<(MASK & 1) != 0>(d2 + base + u * BS); c1[u] = ld2<(MASK & 1) != 0>(c2 + base + u * BS); }
    }
    for (; t < ntiles; t += gridDim.x) {
        const int64_t base = t * (U * BS) + threadIdx.x;
        if (PF) {
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = a1[u]; o[u] = o1[u]; q[u] = q1[u]; cc[u] = c1[u]; }
            const int64_t tn = t + gridDim.x;
            if (tn < ntiles) {
                const int64_t bn = tn * (U * BS) + threadIdx.x;
#pragma unroll
                for (int u = 0; u < U; ++u) { a1[u] = ld2<(MASK & 4) != 0>(xk2 + bn + u * BS); o1[u] = ld2<(MASK & 2) != 0>(xo2 + bn + u * BS);
                    q1[u] = ld2<(MASK & 1) != 0>(d2 + bn + u * BS); c1[u] = ld2<(MASK & 1) != 0>(c2 + bn + u * BS); }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = ld2<(MASK & 4) != 0>(xk2 + base + u * BS); o[u] = ld2<(MASK & 2) != 0>(xo2 + base + u * BS);
                q[u] = ld2<(MASK & 1) != 0>(d2 + base + u * BS); cc[u] = ld2<(MASK & 1) != 0>(c2 + base + u * BS); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u].x = zf_elem_diag<true, false>(a[u].x, o[u].x, q[u].x, cc[u].x, beta, lr, tau, 0, 0, acc);
            r[u].y = zf_elem_diag<true, false>(a[u].y, o[u].y, q[u].y, cc[u].y, beta, lr, tau, 0, 0, acc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st2<(MASK & 8) != 0>(xn2 + base + u * BS, r[u]);
    }
    const double sums[5] = {acc.fy, acc.dot, acc.ss, acc.l1, acc.fx};
    const double maxs[1] = {acc.mx};
    double out = 0.0;
    zf_block_reduce<5, 1, BS / 64>(sums, maxs, lds, out);
    if (threadIdx.x < ZF_NPART) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}
// calibration: pure streams (R reads, 1 write), 16 B per lane
template <int R, int U, int BS>
__global__ __launch_bounds__(BS) void stream_v(const double2* __restrict__ a, const double2* __restrict__ b,
                                              const double2* __restrict__ c, const double2* __restrict__ d,
                                              double2* __restrict__ o, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * BS;
    int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = a[i + u * stride];
            if (R > 1) { double2 t = b[i + u * stride]; v[u].x += t.x; v[u].y += t.y; }
            if (R > 2) { double2 t = c[i + u * stride]; v[u].x += t.x; v[u].y += t.y; }
            if (R > 3) { double2 t = d[i + u * stride]; v[u].x += t.x; v[u].y += t.y; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) o[i + u * stride] = v[u];
    }
    for (; i < n2; i += stride) { double2 v = a[i]; o[i] = v; }
}

struct Bufs { double *xk, *xo, *d, *c, *xn, *partials; int64_t n; };

template <typename F> double time_ms(F launch, int reps = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int U, int BS, bool NTL, bool NTS> void run_trial(const Bufs& B, int grid, const char* tag) {
    double ms = time_ms([&] { hipLaunchKernelGGL((trial_v<U, BS, NTL, NTS>), dim3(grid), dim3(BS), 0, 0, B.xk, B.xo, B.d, B.c,
                                                 B.xn, 0.3, 0.45, 0.1, B.n, B.partials); });
    printf("trial U=%d BS=%4d ntl=%d nts=%d grid=%6d %-8s : %7.3f ms  %7.1f GB/s (40 B/elem)\n", U, BS, (int)NTL, (int)NTS, grid,
           tag, ms, 40.0 * B.n / ms / 1e6);
}
template <int U, int BS, int MASK> void run_c(const Bufs& B) {
    const int64_t n2 = B.n / 2;
    int grid = (int)((n2 + (int64_t)U * BS - 1) / ((int64_t)U * BS));
    double ms = time_ms([&] { hipLaunchKernelGGL((trial_c<U, BS, MASK>), dim3(grid), dim3(BS), 0, 0, B.xk, B.xo, B.d, B.c,
                                                 B.xn, 0.3, 0.45, 0.1, B.n, B.partials); });
    printf("contig U=%d BS=%4d mask=%2d grid=%6d : %7.3f ms  %7.1f GB/s\n", U, BS, MASK, grid, ms, 40.0 * B.n / ms / 1e6);
}
// staged copies of the library kernel's per-workgroup overheads (where do the 4-6 % go?)
//   LEVEL 0: control-block scalar loads + ring selection, plain partial stores
//   LEVEL 1: + wave 0 publishes sc1 and drains vmcnt before storing its x+ tile
//   LEVEL 2: + returning ticket on the group counter, LDS flag, workgroup barrier
template <int LEVEL>
__global__ __launch_bounds__(ZF_BLOCK) void trial_dbg(tune_args A) {
    __shared__ double lds[ZF_WAVES * 8 + 8];
    __shared__ int s_flag;
    if (A.ctl->status != ZF_RUNNING) return;
    const int cur = A.ctl->cur;
    const double lr = A.ctl->lr, beta = A.ctl->beta_next, tau = A.lam * lr;
    const zf_d2* xk2 = (const zf_d2*)A.xb[cur]; const zf_d2* xo2 = (const zf_d2*)A.xb[(cur + 2) % 3];
    zf_d2* xn2 = (zf_d2*)A.xb[(cur + 1) % 3];
    const zf_d2* p02 = (const zf_d2*)A.p0; const zf_d2* p12 = (const zf_d2*)A.p1;
    zf_elem_acc acc = {0, 0, 0, 0, 0, 0};
    const int64_t base = (int64_t)blockIdx.x * ZF_TILE_UNITS + threadIdx.x;
    zf_d2 a[4], o[4], q[4], cc[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = base + u * ZF_BLOCK; a[u] = xk2[i]; o[u] = xo2[i];
        q[u] = zf_ld2<true>(p02 + i); cc[u] = zf_ld2<true>(p12 + i); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        r[u].x = zf_elem_diag<true, false>(a[u].x, o[u].x, q[u].x, cc[u].x, beta, lr, tau, 0, 0, acc);
        r[u].y = zf_elem_diag<true, false>(a[u].y, o[u].y, q[u].y, cc[u].y, beta, lr, tau, 0, 0, acc);
    }
    const double sums[5] = {acc.fy, acc.dot, acc.ss, acc.l1, acc.fx};
    const double maxs[1] = {acc.mx};
    double mine = 0.0;
    zf_block_reduce<5, 1, ZF_WAVES>(sums, maxs, lds, mine);
    if (LEVEL == 0) {
        if (threadIdx.x < ZF_NPART) A.ws.blk_part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = mine;
    } else if (threadIdx.x < 64) {
        if (threadIdx.x < ZF_NPART) zf_publish(A.ws.blk_part + (int64_t)threadIdx.x * gridDim.x + blockIdx.x, mine);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (LEVEL >= 2 && threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(A.ws.grp_cnt + blockIdx.x / ZF_GROUP, 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
            s_flag = (t == 0xffffffffu);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) zf_st2<true>(xn2 + base + u * ZF_BLOCK, r[u]);
    if (LEVEL >= 2) {
        __syncthreads();
        if (s_flag) A.ws.totals[0] = 1.0;
    }
}
// LEVEL 2 structure, but a workgroup owns T tiles (consecutive when INTERLEAVE = 0, else b, b+G, ...)
// and takes ONE ticket after the last tile
template <int T, int INTERLEAVE>
__global__ __launch_bounds__(ZF_BLOCK) void trial_multi(tune_args A, int64_t ntiles) {
    __shared__ double lds[ZF_WAVES * 8 + 8];
    __shared__ int s_flag;
    if (A.ctl->status != ZF_RUNNING) return;
    const int cur = A.ctl->cur;
    const double lr = A.ctl->lr, beta = A.ctl->beta_next, tau = A.lam * lr;
    const zf_d2* xk2 = (const zf_d2*)A.xb[cur]; const zf_d2* xo2 = (const zf_d2*)A.xb[(cur + 2) % 3];
    zf_d2* xn2 = (zf_d2*)A.xb[(cur + 1) % 3];
    const zf_d2* p02 = (const zf_d2*)A.p0; const zf_d2* p12 = (const zf_d2*)A.p1;
    zf_elem_acc acc = {0, 0, 0, 0, 0, 0};
    zf_d2 r[4];
    int64_t last_base = -1;
    for (int t = 0; t < T; ++t) {
        const int64_t tile = INTERLEAVE ? ((int64_t)t * gridDim.x + blockIdx.x) : ((int64_t)blockIdx.x * T + t);
        if (tile >= ntiles) break;
        if (last_base >= 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) zf_st2<true>(xn2 + last_base + u * ZF_BLOCK, r[u]);
        }
        const int64_t base = tile * ZF_TILE_UNITS + threadIdx.x;
        zf_d2 a[4], o[4], q[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t i = base + u * ZF_BLOCK; a[u] = xk2[i]; o[u] = xo2[i];
            q[u] = zf_ld2<true>(p02 + i); cc[u] = zf_ld2<true>(p12 + i); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            r[u].x = zf_elem_diag<true, false>(a[u].x, o[u].x, q[u].x, cc[u].x, beta, lr, tau, 0, 0, acc);
            r[u].y = zf_elem_diag<true, false>(a[u].y, o[u].y, q[u].y, cc[u].y, beta, lr, tau, 0, 0, acc);
        }
        last_base = base;
    }
    const double sums[5] = {acc.fy, acc.dot, acc.ss, acc.l1, acc.fx};
    const double maxs[1] = {acc.mx};
    double mine = 0.0;
    zf_block_reduce<5, 1, ZF_WAVES>(sums, maxs, lds, mine);
    if (threadIdx.x < 64) {
        if (threadIdx.x < ZF_NPART) zf_publish(A.ws.blk_part + (int64_t)threadIdx.x * gridDim.x + blockIdx.x, mine);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(A.ws.grp_cnt + blockIdx.x / ZF_GROUP, 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
            s_flag = (t == 0xffffffffu);
        }
    }
    if (last_base >= 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) zf_st2<true>(xn2 + last_base + u * ZF_BLOCK, r[u]);
    }
    __syncthreads();
    if (s_flag) A.ws.totals[0] = 1.0;
}
template <int T, int INTERLEAVE> void run_multi(const Bufs& B) {
    const int64_t ntiles = (B.n / 2) / ZF_TILE_UNITS;
    int grid = (int)((ntiles + T - 1) / T);
    int ngroups = (grid + ZF_GROUP - 1) / ZF_GROUP;
    zf_control h; memset(&h, 0, sizeof(h)); h.lr = 0.45; h.status = ZF_RUNNING; h.beta_next = 0.3;
    zf_control* ctl; CK(hipMalloc(&ctl, sizeof(h))); CK(hipMemcpy(ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    unsigned* cnt; tune_args A;
    CK(hipMalloc(&cnt, 4 * (ngroups + 16))); CK(hipMemset(cnt, 0, 4 * (ngroups + 16)));
    CK(hipMalloc(&A.ws.blk_part, 8 * ZF_NPART * grid)); CK(hipMalloc(&A.ws.totals, 64)); A.ws.grp_cnt = cnt;
    A.ctl = ctl; A.xb[0] = B.xk; A.xb[1] = B.xn; A.xb[2] = B.xo; A.p0 = B.d; A.p1 = B.c; A.lam = 0.1; A.n = B.n;
    double ms = time_ms([&] { hipLaunchKernelGGL((trial_multi<T, INTERLEAVE>), dim3(grid), dim3(ZF_BLOCK), 0, 0, A, ntiles); });
    printf("multi-tile T=%d interleave=%d grid=%6d : %7.3f ms  %7.1f GB/s\n", T, INTERLEAVE, grid, ms, 40.0 * B.n / ms / 1e6);
}
// ---- alternative: plain partial stores + a separate fast finalize/decide kernel --------------------
// FB workgroups of 1024 threads each reduce a slice of the per-workgroup partials with all loads
// independent; the last arriver (one ticket per finalize workgroup) adds the slices and decides.
template <int FB>
__global__ __launch_bounds__(1024) void finalize_fast(const double* __restrict__ partials, int nblocks, double* slice_part,
                                                      unsigned* cnt, zf_control* ctl, double* trace, double* pack) {
    __shared__ double lds[16 * 8];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (nblocks + FB - 1) / FB;
    const int b0 = blockIdx.x * per;
    int b1 = b0 + per; if (b1 > nblocks) b1 = nblocks;
    double v[ZF_NPART];
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k) v[k] = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += 1024) {
        double p[ZF_NPART];
#pragma unroll
        for (int k = 0; k < ZF_NPART; ++k) p[k] = partials[(int64_t)k * nblocks + b];
#pragma unroll
        for (int k = 0; k < ZF_NPART - 1; ++k) v[k] += p[k];
        v[ZF_NPART - 1] = fmax(v[ZF_NPART - 1], p[ZF_NPART - 1]);
    }
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k) {
        const double r = (k == ZF_NPART - 1) ? zf_wave_max(v[k]) : zf_wave_sum(v[k]);
        if (lane == 0) lds[wave * 8 + k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x < ZF_NPART) {
            const int k = threadIdx.x;
            double r = lds[k];
            for (int w = 1; w < 16; ++w) r = (k == ZF_NPART - 1) ? fmax(r, lds[w * 8 + k]) : r + lds[w * 8 + k];
            if (FB == 1) lds[k] = r; else zf_publish(slice_part + (int64_t)k * FB + blockIdx.x, r);
        }
        if (FB > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (t == FB - 1);
                if (s_last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
    if (FB > 1 && !s_last) return;
    if (threadIdx.x == 0) {
        double tot[ZF_NPART];
        for (int k = 0; k < ZF_NPART; ++k) {
            if (FB == 1) tot[k] = lds[k];
            else {
                double r = zf_consume(slice_part + (int64_t)k * FB);
                for (int q = 1; q < FB; ++q) { const double p = zf_consume(slice_part + (int64_t)k * FB + q); r = (k == ZF_NPART - 1) ? fmax(r, p) : r + p; }
                tot[k] = r;
            }
        }
        pack[0] = 0.5 * tot[0]; pack[1] = tot[1]; pack[2] = tot[2]; pack[3] = 0.1 * tot[3]; pack[4] = 0.5 * tot[4]; pack[5] = tot[5];
        pack[6] = pack[7] = 0.0;
        zf_decide_step(ctl, pack, trace, nullptr);
    }
}

template <int FB> void run_split(const Bufs& B) {
    const int64_t n2 = B.n / 2;
    int grid = (int)(n2 / ZF_TILE_UNITS);
    zf_control h; memset(&h, 0, sizeof(h)); h.lr = 0.45; h.status = ZF_RUNNING; h.world = 1; h.max_iter = 1 << 30;
    h.max_backtrack = 100; h.decay_rate = 0.5; h.tol_internal = 1e300; h.F_old = 1e300; h.nesterov = 1; h.beta_next = 0.3;
    zf_control* ctl; CK(hipMalloc(&ctl, sizeof(h))); CK(hipMemcpy(ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    double *trace, *pack, *slice; unsigned* cnt; tune_args A;
    CK(hipMalloc(&trace, 8 * ZF_RING * 8)); CK(hipMalloc(&pack, 64)); CK(hipMalloc(&slice, 8 * ZF_NPART * 64));
    CK(hipMalloc(&cnt, 64)); CK(hipMemset(cnt, 0, 64));
    CK(hipMalloc(&A.ws.blk_part, 8 * ZF_NPART * grid)); CK(hipMalloc(&A.ws.totals, 64));
    A.ctl = ctl; A.xb[0] = B.xk; A.xb[1] = B.xn; A.xb[2] = B.xo; A.p0 = B.d; A.p1 = B.c; A.lam = 0.1; A.n = B.n; 
    const int K = 20;
    double ms = time_ms([&] { for (int k = 0; k < K; ++k) {
        hipLaunchKernelGGL((trial_dbg<0>), dim3(grid), dim3(ZF_BLOCK), 0, 0, A);
        hipLaunchKernelGGL((finalize_fast<FB>), dim3(FB), dim3(1024), 0, 0, A.ws.blk_part, grid, slice, cnt, ctl, trace, pack); } }, 5);
    printf("SPLIT  streaming + finalize_fast<%2d> : %7.3f ms per step (loop of %d)\n", FB, ms / K, K);
}
// the shipped pair: library trial kernel (plain partial stores) + library finalize/decide kernel
template <bool NT, int S> void run_lib_loop(const Bufs& B, int T) {
    const int64_t n2 = B.n / 2;
    const int64_t nt_all = (n2 + ZF_TILE_UNITS - 1) / ZF_TILE_UNITS;
    int grid = (int)((nt_all + T - 1) / T);
    zf_control h; memset(&h, 0, sizeof(h)); h.lr = 0.45; h.status = ZF_RUNNING; h.world = 1; h.max_iter = 1 << 30;
    h.max_backtrack = 100; h.decay_rate = 0.5; h.tol_internal = 1e300; h.F_old = 1e300; h.nesterov = 1;
    constexpr int RINGSZ = S > 1 ? 4 : 3;
    h.ring_size = RINGSZ; h.sub_iters = S; h.prev = RINGSZ - 1; h.plan_n = S; h.cut_at = -1; h.ncuts = 0;
    zf_control* ctl; CK(hipMalloc(&ctl, sizeof(h))); CK(hipMemcpy(ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    double *beta, *trace, *pack; zf_step_args A; zf_finalize_args F;
    CK(hipMalloc(&beta, 8 * ZF_RING)); CK(hipMemset(beta, 0, 8 * ZF_RING)); CK(hipMalloc(&trace, 8 * ZF_RING * 8));
    CK(hipMalloc(&pack, 64 * S)); CK(hipMalloc(&F.cnt, 64)); CK(hipMemset(F.cnt, 0, 64));
    CK(hipMalloc(&A.blk_part, 8 * ZF_NPART * S * grid)); CK(hipMalloc(&F.slice_part, 8 * ZF_NPART * S * ZF_FIN_WGS));
    A.ctl = ctl; A.beta_ring = beta;
    double* extra[ZF_MAX_RING] = {};
    for (int k = 0; k < ZF_MAX_RING; ++k) A.xb[k] = B.xk;
    A.xb[0] = B.xk; A.xb[1] = B.xn; A.xb[RINGSZ - 1] = B.xo;     // ring: x_k, free..., x_{k-1}
    for (int k = 2; k < RINGSZ - 1; ++k) { CK(hipMalloc(&extra[k], 8 * B.n)); A.xb[k] = extra[k]; }
    A.p0 = B.d; A.p1 = B.c;
    A.lam = 0.1; A.lo = 0; A.hi = 0; A.n = B.n; A.tiles_per_wg = T;
    F.blk_part = A.blk_part; F.nblocks = grid; F.sub_iters = S; for (int k = 0; k < ZF_NPART; ++k) F.scale[k] = 1.0;
    F.f_y_ext = F.f_x_ext = nullptr; F.contribute_f = 1; F.pack = pack; F.ctl = ctl; F.decide = 1; F.trace = trace; F.beta_ring = beta;
    int wgs = (grid + ZF_FIN_THREADS - 1) / ZF_FIN_THREADS; if (wgs > ZF_FIN_WGS) wgs = ZF_FIN_WGS; if (wgs < 1) wgs = 1;
    const int K = 20;
    double ms = time_ms([&] { for (int k = 0; k < K; ++k) {
        hipLaunchKernelGGL((zf_trial_kernel<true, true, false, NT, S>), dim3(grid), dim3(ZF_BLOCK), 0, 0, A);
        hipLaunchKernelGGL(zf_finalize_kernel<S>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, 0, F); } }, 5);
    double ms_f = time_ms([&] { for (int k = 0; k < K; ++k)
        hipLaunchKernelGGL(zf_finalize_kernel<S>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, 0, F); }, 5);
    F.decide = 0;
    double ms_r = time_ms([&] { for (int k = 0; k < K; ++k)
        hipLaunchKernelGGL(zf_finalize_kernel<S>, dim3(wgs), dim3(ZF_FIN_THREADS), 0, 0, F); }, 5);
    F.decide = 1;
    printf("        finalize alone S=%d T=%d (%d workgroups over %d partial rows): %7.1f us per launch (%.1f without the decide pass)\n", S, T, wgs, grid, ms_f / K * 1e3, ms_r / K * 1e3);
    printf("LIBRARY trial + finalize  S=%d T=%d nt=%d : %7.3f ms per pass = %7.3f ms per iteration (loop of %d)\n",
           S, T, (int)NT, ms / K, ms / K / S, K);
    for (int k = 2; k < RINGSZ - 1; ++k) CK(hipFree(extra[k]));
}
template <int LEVEL> void run_dbg(const Bufs& B) {
    const int64_t n2 = B.n / 2;
    int grid = (int)(n2 / ZF_TILE_UNITS);   // full tiles only
    int ngroups = (grid + ZF_GROUP - 1) / ZF_GROUP;
    zf_control h; memset(&h, 0, sizeof(h)); h.lr = 0.45; h.status = ZF_RUNNING; h.beta_next = 0.3;
    zf_control* ctl; CK(hipMalloc(&ctl, sizeof(h))); CK(hipMemcpy(ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    unsigned* cnt; tune_args A;
    CK(hipMalloc(&cnt, 4 * (ngroups + 16))); CK(hipMemset(cnt, 0, 4 * (ngroups + 16)));
    CK(hipMalloc(&A.ws.blk_part, 8 * ZF_NPART * grid)); CK(hipMalloc(&A.ws.totals, 64)); A.ws.grp_cnt = cnt;
    A.ctl = ctl; A.xb[0] = B.xk; A.xb[1] = B.xn; A.xb[2] = B.xo; A.p0 = B.d; A.p1 = B.c; A.lam = 0.1; A.n = B.n;
    double ms = time_ms([&] { hipLaunchKernelGGL((trial_dbg<LEVEL>), dim3(grid), dim3(ZF_BLOCK), 0, 0, A); });
    printf("staged LEVEL %d grid=%6d : %7.3f ms  %7.1f GB/s\n", LEVEL, grid, ms, 40.0 * B.n / ms / 1e6);
}
template <int U, int BS, int MASK, bool PF> void run_p(const Bufs& B, int grid) {
    double ms = time_ms([&] { hipLaunchKernelGGL((trial_p<U, BS, MASK, PF>), dim3(grid), dim3(BS), 0, 0, B.xk, B.xo, B.d, B.c,
                                                 B.xn, 0.3, 0.45, 0.1, B.n, B.partials); });
    printf("persist U=%d BS=%4d mask=%2d pf=%d grid=%6d : %7.3f ms  %7.1f GB/s\n", U, BS, MASK, (int)PF, grid, ms, 40.0 * B.n / ms / 1e6);
}

template <int R, int U, int BS> void run_stream(const Bufs& B, int grid) {
    double ms = time_ms([&] { hipLaunchKernelGGL((stream_v<R, U, BS>), dim3(grid), dim3(BS), 0, 0, (const double2*)B.xk,
                                                 (const double2*)B.xo, (const double2*)B.d, (const double2*)B.c,
                                                 (double2*)B.xn, B.n / 2); });
    printf("stream R=%d U=%d BS=%4d grid=%6d          : %7.3f ms  %7.1f GB/s (%d B/elem)\n", R, U, BS, grid, ms,
           8.0 * (R + 1) * B.n / ms / 1e6, 8 * (R + 1));
}

__global__ void fill(double* p, int64_t n, double s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = s * (double)((i * 2654435761u) % 1000003) / 1000003.0 + 0.5;
}

int main(int argc, char** argv) {
    Bufs B; B.n = argc > 1 ? atoll(argv[1]) : 100000000LL;
    size_t bytes = sizeof(double) * B.n;
    CK(hipMalloc(&B.xk, bytes)); CK(hipMalloc(&B.xo, bytes)); CK(hipMalloc(&B.d, bytes)); CK(hipMalloc(&B.c, bytes));
    CK(hipMalloc(&B.xn, bytes)); CK(hipMalloc(&B.partials, sizeof(double) * ZF_NPART * 1000000));
    fill<<<2048, 256>>>(B.xk, B.n, 1.0); fill<<<2048, 256>>>(B.xo, B.n, 0.9); fill<<<2048, 256>>>(B.d, B.n, 1.5);
    fill<<<2048, 256>>>(B.c, B.n, -2.0); CK(hipDeviceSynchronize());
    const int64_t n2 = B.n / 2;
    auto full = [&](int U, int BS) { return (int)((n2 + (int64_t)U * BS - 1) / ((int64_t)U * BS)); };
    printf("n = %lld\n", (long long)B.n);
    for (int rep = 0; rep < 2; ++rep) {
        run_c<4, 256, 9>(B);
        run_split<16>(B);
        run_lib_loop<true, 1>(B, 1); run_lib_loop<true, 1>(B, 2); run_lib_loop<true, 1>(B, 4); run_lib_loop<false, 1>(B, 1);
        run_lib_loop<true, 2>(B, 1); run_lib_loop<true, 2>(B, 2); run_lib_loop<false, 2>(B, 1);
        run_lib_loop<true, 4>(B, 1); run_lib_loop<true, 4>(B, 2); run_lib_loop<true, 4>(B, 4); run_lib_loop<false, 4>(B, 1);
        run_lib_loop<true, 8>(B, 1); run_lib_loop<true, 8>(B, 2); run_lib_loop<true, 8>(B, 4); run_lib_loop<true, 8>(B, 8); run_lib_loop<false, 8>(B, 8);
    }
    return 0;
}
