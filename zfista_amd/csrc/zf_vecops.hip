// zf_vecops.hip - the solver's own O(n) arithmetic for opaque host callbacks.
//
// When f / g / jac_f / prox_wsum_g are arbitrary Python callables the iterates
// live in host memory (the callbacks need NumPy arrays); what remains of
// zfista/proximal_gradient.py is three vector expressions, computed here on the
// GPU with host pointers in and out (staged through a library-owned device
// workspace):
//   v = y - lr * jac                                   :148
//   <jac, x - y>, |x - y|^2, max|x - y|                :150-152, :510
//   y' = x + beta * (x - x_old)                        :534
#include <mutex>
#include <vector>

#include "zf_common.h"

namespace {

// Staging workspace of the host-pointer entry points: one per host thread AND device (a caller
// that switches devices between calls gets that device's buffers, never pointers into another
// GPU's memory), registered so that zf_shutdown() can release all of them.
struct zf_workspace {
    int device = -1;
    double* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t cap = 0;
    double* partials = nullptr;
    double* out = nullptr;
};
constexpr int ZF_WS_DEVICES = 16;
thread_local zf_workspace g_ws_tab[ZF_WS_DEVICES];
thread_local zf_workspace* g_ws_cur = nullptr;
#define g_ws (*g_ws_cur)
std::mutex g_ws_mu;
std::vector<zf_workspace*> g_ws_all;   // every workspace ever used, for zf_shutdown

void zf_ws_release(zf_workspace* w) {
    if (w->device < 0) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(w->device);
    for (int k = 0; k < 4; ++k)
        if (w->buf[k]) (void)hipFree(w->buf[k]);
    if (w->partials) (void)hipFree(w->partials);
    if (w->out) (void)hipFree(w->out);
    *w = zf_workspace();
    (void)hipSetDevice(cur);
}

// a thread that ends takes its workspaces with it (and out of the registry)
struct zf_ws_owner {
    ~zf_ws_owner() {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        for (zf_workspace& w : g_ws_tab) {
            for (size_t k = 0; k < g_ws_all.size(); ++k)
                if (g_ws_all[k] == &w) {
                    g_ws_all.erase(g_ws_all.begin() + k);
                    break;
                }
            zf_ws_release(&w);
        }
    }
};
thread_local zf_ws_owner g_ws_owner;

int zf_ws_reserve(int64_t n) {
    (void)&g_ws_owner;   // (constructs the owner of this thread's table)
    int dev = 0;
    ZF_HIP(hipGetDevice(&dev));
    ZF_REQUIRE(dev >= 0 && dev < ZF_WS_DEVICES, "zf_host_*: device index beyond the workspace table");
    g_ws_cur = &g_ws_tab[dev];
    if (g_ws.device < 0) {
        g_ws.device = dev;
        std::lock_guard<std::mutex> lock(g_ws_mu);
        g_ws_all.push_back(g_ws_cur);
    }
    if (!g_ws.partials) {
        ZF_HIP(hipMalloc(&g_ws.partials, sizeof(double) * 12 * ZF_MAX_GRID));
        ZF_HIP(hipMalloc(&g_ws.out, sizeof(double) * 8));
    }
    if (n <= g_ws.cap) return ZF_OK;
    int64_t cap = (n + 1023) & ~int64_t(1023);
    for (int k = 0; k < 4; ++k) {
        if (g_ws.buf[k]) ZF_HIP(hipFree(g_ws.buf[k]));
        g_ws.buf[k] = nullptr;
        ZF_HIP(hipMalloc(&g_ws.buf[k], sizeof(double) * cap));
    }
    g_ws.cap = cap;
    return ZF_OK;
}

__global__ __launch_bounds__(ZF_BLOCK) void k_grad_step(double* __restrict__ v, const double* __restrict__ y,
                                                        const double* __restrict__ jac, double lr, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride)
        v[i] = y[i] - lr * jac[i];
}

__global__ __launch_bounds__(ZF_BLOCK) void k_momentum(double* __restrict__ out, const double* __restrict__ x,
                                                       const double* __restrict__ xo, double beta, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride)
        out[i] = x[i] + beta * (x[i] - xo[i]);
}

// partials: [0] <jac, x-y>  [1] |x-y|^2  [2] max|x-y|
__global__ __launch_bounds__(ZF_BLOCK) void k_model_terms(const double* __restrict__ jac,
                                                          const double* __restrict__ x,
                                                          const double* __restrict__ y, int64_t n,
                                                          double* partials) {
    __shared__ double lds[ZF_WAVES * 3];
    double dot = 0.0, ss = 0.0, mx = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride) {
        const double dx = x[i] - y[i];
        dot += jac[i] * dx;
        ss += dx * dx;
        mx = fmax(mx, fabs(dx));
    }
    const double sums[2] = {dot, ss};
    const double maxs[1] = {mx};
    double out = 0.0;
    zf_block_reduce<2, 1, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x < 3) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// partials: [0] sum d (x-c)^2  [1] |x|_1
__global__ __launch_bounds__(ZF_BLOCK) void k_eval_diag(const double* __restrict__ x, const double* __restrict__ d,
                                                        const double* __restrict__ c, int64_t n,
                                                        double* partials) {
    __shared__ double lds[ZF_WAVES * 2];
    double f = 0.0, l1 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride) {
        const double r = x[i] - c[i];
        f += d[i] * (r * r);
        l1 += fabs(x[i]);
    }
    const double sums[2] = {f, l1};
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<2, 0, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x < 2) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// ---- multi-objective trial with callbacks on device tensors (m >= 2) --------------------------------
// the two vector expressions of _dual_minimized_fun_jac (proximal_gradient.py:162-173) that are the
// solver's own; prox_wsum_g and g in between are the caller's
constexpr int ZF_DEV_MO_MAX_M = 8;
struct zf_dev_mo_w {
    double w[ZF_DEV_MO_MAX_M];
};
// v = y - lr * (w @ J)   (:164, the prox argument);  partial [0] = |w @ J|^2   (:171)
__global__ __launch_bounds__(ZF_BLOCK) void k_dev_mo_combine(double* __restrict__ v, const double* __restrict__ y,
                                                             const double* __restrict__ J, zf_dev_mo_w W, double lr,
                                                             int m, int64_t n, double* partials) {
    __shared__ double lds[ZF_WAVES];
    double ss = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        double wJ = 0.0;
#pragma unroll
        for (int i = 0; i < ZF_DEV_MO_MAX_M; ++i)
            if (i < m) wJ += W.w[i] * J[(int64_t)i * n + j];
        v[j] = y[j] - lr * wJ;
        ss += wJ * wJ;
    }
    const double sums[1] = {ss};
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<1, 0, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x == 0) partials[blockIdx.x] = out;
}
// partials [0 .. m) = J_i . (p - y)   (:173),  [m] = |p - v|^2   (:168)
__global__ __launch_bounds__(ZF_BLOCK) void k_dev_mo_post(const double* __restrict__ J, const double* __restrict__ y,
                                                          const double* __restrict__ p, const double* __restrict__ v,
                                                          int m, int64_t n, double* partials) {
    constexpr int NS = ZF_DEV_MO_MAX_M + 1;
    __shared__ double lds[ZF_WAVES * NS];
    double acc[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        const double pj = p[j];
        const double dy = pj - y[j], dv = pj - v[j];
#pragma unroll
        for (int i = 0; i < ZF_DEV_MO_MAX_M; ++i)
            if (i < m) acc[i] += J[(int64_t)i * n + j] * dy;
        acc[ZF_DEV_MO_MAX_M] += dv * dv;
    }
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<NS, 0, ZF_WAVES>(acc, maxs, lds, out);
    // rows 0 .. m-1: the dots, row m: |p - v|^2 (compacted: the reduce kernel reads m + 1 rows)
    if (threadIdx.x < m) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
    if (threadIdx.x == ZF_DEV_MO_MAX_M) partials[(int64_t)m * gridDim.x + blockIdx.x] = out;
}

// one-block fixed-order reduce of `nq` quantities; quantity max_index is a max
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partials, int nblocks, int nq,
                                                         int max_index, double* out) {
    __shared__ double lds[4 * 12];   // (nq <= 12)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < nq; ++k) {
        const bool is_max = (k == max_index);
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) {
            const double p = partials[(int64_t)k * nblocks + b];
            v = is_max ? fmax(v, p) : v + p;
        }
        v = is_max ? zf_wave_max(v) : zf_wave_sum(v);
        if (lane == 0) lds[wave * 12 + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < nq) {
        const int k = threadIdx.x;
        double v = lds[k];
        for (int w = 1; w < 4; ++w) v = (k == max_index) ? fmax(v, lds[w * 12 + k]) : v + lds[w * 12 + k];
        out[k] = v;
    }
}

}  // namespace

extern "C" int zf_host_grad_step(double* v_host, const double* y_host, const double* jac_host, double lr,
                                 int64_t n) {
    ZF_REQUIRE(v_host && y_host && jac_host && n >= 0, "zf_host_grad_step: bad argument");
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(n);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * n;
    ZF_HIP(hipMemcpyAsync(g_ws.buf[0], y_host, bytes, hipMemcpyHostToDevice, nullptr));
    ZF_HIP(hipMemcpyAsync(g_ws.buf[1], jac_host, bytes, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_grad_step, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, nullptr, g_ws.buf[2], g_ws.buf[0],
                       g_ws.buf[1], lr, n);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(v_host, g_ws.buf[2], bytes, hipMemcpyDeviceToHost, nullptr));
    ZF_HIP(hipStreamSynchronize(nullptr));
    return ZF_OK;
}

extern "C" int zf_host_model_terms(const double* jac_host, const double* x_host, const double* y_host,
                                   int64_t n, double out3[3]) {
    ZF_REQUIRE(jac_host && x_host && y_host && out3 && n >= 0, "zf_host_model_terms: bad argument");
    out3[0] = out3[1] = out3[2] = 0.0;
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(n);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * n;
    ZF_HIP(hipMemcpyAsync(g_ws.buf[0], jac_host, bytes, hipMemcpyHostToDevice, nullptr));
    ZF_HIP(hipMemcpyAsync(g_ws.buf[1], x_host, bytes, hipMemcpyHostToDevice, nullptr));
    ZF_HIP(hipMemcpyAsync(g_ws.buf[2], y_host, bytes, hipMemcpyHostToDevice, nullptr));
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_model_terms, dim3(g), dim3(ZF_BLOCK), 0, nullptr, g_ws.buf[0], g_ws.buf[1], g_ws.buf[2],
                       n, g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, nullptr, g_ws.partials, g, 3, 2, g_ws.out);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(out3, g_ws.out, sizeof(double) * 3, hipMemcpyDeviceToHost, nullptr));
    ZF_HIP(hipStreamSynchronize(nullptr));
    return ZF_OK;
}

extern "C" int zf_host_momentum(double* y_out_host, const double* x_host, const double* x_old_host, double beta,
                                int64_t n) {
    ZF_REQUIRE(y_out_host && x_host && x_old_host && n >= 0, "zf_host_momentum: bad argument");
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(n);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * n;
    ZF_HIP(hipMemcpyAsync(g_ws.buf[0], x_host, bytes, hipMemcpyHostToDevice, nullptr));
    ZF_HIP(hipMemcpyAsync(g_ws.buf[1], x_old_host, bytes, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_momentum, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, nullptr, g_ws.buf[2], g_ws.buf[0],
                       g_ws.buf[1], beta, n);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(y_out_host, g_ws.buf[2], bytes, hipMemcpyDeviceToHost, nullptr));
    ZF_HIP(hipStreamSynchronize(nullptr));
    return ZF_OK;
}

// ---- the same three expressions on DEVICE vectors (tensor callbacks: x0 and every callback
// result live in HBM; nothing but three scalars crosses PCIe per trial) ---------------------
extern "C" int zf_dev_grad_step(double* v_dev, const double* y_dev, const double* jac_dev, double lr, int64_t n,
                                void* stream) {
    ZF_REQUIRE(v_dev && y_dev && jac_dev && n >= 0, "zf_dev_grad_step: bad argument");
    if (n == 0) return ZF_OK;
    hipLaunchKernelGGL(k_grad_step, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, (hipStream_t)stream, v_dev, y_dev,
                       jac_dev, lr, n);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

extern "C" int zf_dev_model_terms(const double* jac_dev, const double* x_dev, const double* y_dev, int64_t n,
                                  double out3_host[3], void* stream) {
    ZF_REQUIRE(jac_dev && x_dev && y_dev && out3_host && n >= 0, "zf_dev_model_terms: bad argument");
    out3_host[0] = out3_host[1] = out3_host[2] = 0.0;
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_model_terms, dim3(g), dim3(ZF_BLOCK), 0, st, jac_dev, x_dev, y_dev, n, g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, g_ws.partials, g, 3, 2, g_ws.out);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(out3_host, g_ws.out, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
    ZF_HIP(hipStreamSynchronize(st));
    return ZF_OK;
}

// as zf_dev_model_terms, but the three scalars stay on the device (out3_dev) and nothing is
// synchronised: the caller fetches them together with its own scalars in one transfer
extern "C" int zf_dev_model_terms_async(const double* jac_dev, const double* x_dev, const double* y_dev, int64_t n,
                                        double* out3_dev, void* stream) {
    ZF_REQUIRE(jac_dev && x_dev && y_dev && out3_dev && n >= 1, "zf_dev_model_terms_async: bad argument");
    int rc = zf_ws_reserve(1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_model_terms, dim3(g), dim3(ZF_BLOCK), 0, st, jac_dev, x_dev, y_dev, n, g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, g_ws.partials, g, 3, 2, out3_dev);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

extern "C" int zf_dev_momentum(double* y_out_dev, const double* x_dev, const double* x_old_dev, double beta,
                               int64_t n, void* stream) {
    ZF_REQUIRE(y_out_dev && x_dev && x_old_dev && n >= 0, "zf_dev_momentum: bad argument");
    if (n == 0) return ZF_OK;
    hipLaunchKernelGGL(k_momentum, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, (hipStream_t)stream, y_out_dev, x_dev,
                       x_old_dev, beta, n);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// Multi-objective trial on device tensors (m >= 2): v = y - lr (w @ J) and |w @ J|^2 (ss_dev[0]); J is m x n
// row-major on the device, w comes from the host (m doubles).  Stream-ordered, nothing synchronised.
extern "C" int zf_dev_mo_combine(double* v_dev, const double* y_dev, const double* J_dev, const double* w_host,
                                 double lr, int32_t m, int64_t n, double* ss_dev, void* stream) {
    ZF_REQUIRE(v_dev && y_dev && J_dev && w_host && ss_dev && n >= 1, "zf_dev_mo_combine: bad argument");
    ZF_REQUIRE(m >= 1 && m <= ZF_DEV_MO_MAX_M, "zf_dev_mo_combine: 1 <= m <= 8");
    int rc = zf_ws_reserve(1);
    if (rc) return rc;
    zf_dev_mo_w W;
    for (int i = 0; i < ZF_DEV_MO_MAX_M; ++i) W.w[i] = i < m ? w_host[i] : 0.0;
    hipStream_t st = (hipStream_t)stream;
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_dev_mo_combine, dim3(g), dim3(ZF_BLOCK), 0, st, v_dev, y_dev, J_dev, W, lr, (int)m, n,
                       g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, g_ws.partials, g, 1, -1, ss_dev);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// out_dev[0 .. m) = J_i . (p - y), out_dev[m] = |p - v|^2 after the caller's prox produced p from v
extern "C" int zf_dev_mo_post_terms(const double* J_dev, const double* y_dev, const double* p_dev,
                                    const double* v_dev, int32_t m, int64_t n, double* out_dev, void* stream) {
    ZF_REQUIRE(J_dev && y_dev && p_dev && v_dev && out_dev && n >= 1, "zf_dev_mo_post_terms: bad argument");
    ZF_REQUIRE(m >= 1 && m <= ZF_DEV_MO_MAX_M, "zf_dev_mo_post_terms: 1 <= m <= 8");
    int rc = zf_ws_reserve(1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_dev_mo_post, dim3(g), dim3(ZF_BLOCK), 0, st, J_dev, y_dev, p_dev, v_dev, (int)m, n,
                       g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, g_ws.partials, g, (int)m + 1, -1, out_dev);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

extern "C" int zf_eval_diag_l1(const double* x_dev, const double* d_dev, const double* c_dev, double lam,
                               int64_t n, double out2_host[2], void* stream) {
    ZF_REQUIRE(x_dev && d_dev && c_dev && out2_host && n >= 1, "zf_eval_diag_l1: bad argument");
    int rc = zf_ws_reserve(0);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_eval_diag, dim3(g), dim3(ZF_BLOCK), 0, st, x_dev, d_dev, c_dev, n, g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, g_ws.partials, g, 2, -1, g_ws.out);
    ZF_HIP(hipGetLastError());
    double tmp[2];
    ZF_HIP(hipMemcpyAsync(tmp, g_ws.out, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    ZF_HIP(hipStreamSynchronize(st));
    out2_host[0] = 0.5 * tmp[0];
    out2_host[1] = lam * tmp[1];
    return ZF_OK;
}

namespace {
__global__ __launch_bounds__(ZF_BLOCK) void k_prox_l1_box(double* __restrict__ out, const double* __restrict__ x,
                                                          double tau, double lo, double hi, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride)
        out[i] = zf_clip(zf_soft_threshold(x[i], tau), lo, hi);
}
__global__ __launch_bounds__(ZF_BLOCK) void k_diag_grad(double* __restrict__ out, const double* __restrict__ x,
                                                        const double* __restrict__ d,
                                                        const double* __restrict__ c, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride)
        out[i] = d[i] * (x[i] - c[i]);
}
}  // namespace

extern "C" int zf_host_prox_l1_box(double* out_host, const double* x_host, double tau, double lo, double hi,
                                   int64_t n) {
    ZF_REQUIRE(out_host && x_host && n >= 0, "zf_host_prox_l1_box: bad argument");
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(n);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * n;
    ZF_HIP(hipMemcpyAsync(g_ws.buf[0], x_host, bytes, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_prox_l1_box, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, nullptr, g_ws.buf[1], g_ws.buf[0],
                       tau, lo, hi, n);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(out_host, g_ws.buf[1], bytes, hipMemcpyDeviceToHost, nullptr));
    ZF_HIP(hipStreamSynchronize(nullptr));
    return ZF_OK;
}

extern "C" int zf_host_diag_grad(double* out_host, const double* x_host, const double* d_dev,
                                 const double* c_dev, int64_t n) {
    ZF_REQUIRE(out_host && x_host && d_dev && c_dev && n >= 0, "zf_host_diag_grad: bad argument");
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(n);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * n;
    ZF_HIP(hipMemcpyAsync(g_ws.buf[0], x_host, bytes, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_diag_grad, dim3(zf_grid_for(n)), dim3(ZF_BLOCK), 0, nullptr, g_ws.buf[1], g_ws.buf[0],
                       d_dev, c_dev, n);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(out_host, g_ws.buf[1], bytes, hipMemcpyDeviceToHost, nullptr));
    ZF_HIP(hipStreamSynchronize(nullptr));
    return ZF_OK;
}

namespace {
__global__ __launch_bounds__(ZF_BLOCK) void k_asum(const double* __restrict__ x, int64_t n, double* partials) {
    __shared__ double lds[ZF_WAVES];
    double l1 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride) l1 += fabs(x[i]);
    const double sums[1] = {l1};
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<1, 0, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x == 0) partials[blockIdx.x] = out;
}
}  // namespace

extern "C" int zf_host_asum(const double* x_host, int64_t n, double* out) {
    ZF_REQUIRE(x_host && out && n >= 0, "zf_host_asum: bad argument");
    *out = 0.0;
    if (n == 0) return ZF_OK;
    int rc = zf_ws_reserve(n);
    if (rc) return rc;
    ZF_HIP(hipMemcpyAsync(g_ws.buf[0], x_host, sizeof(double) * n, hipMemcpyHostToDevice, nullptr));
    const int g = zf_grid_for(n);
    hipLaunchKernelGGL(k_asum, dim3(g), dim3(ZF_BLOCK), 0, nullptr, g_ws.buf[0], n, g_ws.partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, nullptr, g_ws.partials, g, 1, -1, g_ws.out);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(out, g_ws.out, sizeof(double), hipMemcpyDeviceToHost, nullptr));
    ZF_HIP(hipStreamSynchronize(nullptr));
    return ZF_OK;
}

// Release the staging workspaces of the host-pointer entry points (all threads, all devices).  Call
// when no zf_host_* / zf_dev_* call is in flight; later calls allocate again.
extern "C" int zf_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (zf_workspace* w : g_ws_all) zf_ws_release(w);
    g_ws_all.clear();
    return ZF_OK;
}
