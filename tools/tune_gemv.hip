// tune_gemv.hip - A^T r column sweep of the dense fp64 least-squares gradient:
// VALU (v_fma_f64, 16-B loads, 1 KiB per wave instruction) against
// v_mfma_f64_16x16x4_f64 (development tool; prints GB/s and the max deviation).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I include tools/tune_gemv.hip -o tools/tune_gemv
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// ---- VALU: thread owns 2 adjacent columns, walks its row slice (the library's kernel) ----
template <int UNROLL>
__global__ __launch_bounds__(256) void gemvT_valu(const double* __restrict__ A, const double* __restrict__ r,
                                                  double* __restrict__ slab, int64_t m, int64_t n, int64_t rps) {
    const int64_t c2 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n2 = n >> 1;
    const int64_t r0 = (int64_t)blockIdx.y * rps;
    int64_t r1 = r0 + rps; if (r1 > m) r1 = m;
    if (c2 >= n2) return;
    const d2* Ac = reinterpret_cast<const d2*>(A) + c2;
    double ax = 0.0, ay = 0.0;
    int64_t i = r0;
    for (; i + UNROLL <= r1; i += UNROLL) {
        d2 a[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a[u] = Ac[(i + u) * n2];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const double rv = r[i + u]; ax += a[u].x * rv; ay += a[u].y * rv; }
    }
    for (; i < r1; ++i) { const d2 a = Ac[i * n2]; const double rv = r[i]; ax += a.x * rv; ay += a.y * rv; }
    d2 o; o.x = ax; o.y = ay;
    reinterpret_cast<d2*>(slab + (int64_t)blockIdx.y * n)[c2] = o;
}

// ---- MFMA: a wave owns a 32-column panel; per step it takes 4 rows x 32 columns ------------------
// B operand of v_mfma_f64_16x16x4_f64: lane l holds B[k = l>>4][col = l&15]; with 16-B loads a lane
// fetches columns 2c, 2c+1 (c = l&15) of row i0 + (l>>4): two MFMAs (even / odd columns).
// A operand: lane l holds A[row = l&15][k = l>>4] = r[i0 + (l>>4)] for every row (rows identical).
// C/D: 4 f64 per lane, col = l&15, row = (l>>4) + 4*reg: row 0 = lanes 0..15, reg 0.
template <int UNROLL>
__global__ __launch_bounds__(256) void gemvT_mfma(const double* __restrict__ A, const double* __restrict__ r,
                                                  double* __restrict__ slab, int64_t m, int64_t n, int64_t rps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = ((int64_t)blockIdx.x * 4 + wave) * 32;   // panel start column
    const int64_t r0 = (int64_t)blockIdx.y * rps;
    int64_t r1 = r0 + rps; if (r1 > m) r1 = m;
    if (j0 >= n) return;
    const int kq = lane >> 4, c = lane & 15;
    const double* Ap = A + j0 + 2 * c;
    d4 acc_e = {0.0, 0.0, 0.0, 0.0}, acc_o = {0.0, 0.0, 0.0, 0.0};
    int64_t i = r0;
    for (; i + 4 * UNROLL <= r1; i += 4 * UNROLL) {
        d2 b[UNROLL];
        double rv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            b[u] = *reinterpret_cast<const d2*>(Ap + (i + 4 * u + kq) * n);
            rv[u] = r[i + 4 * u + kq];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc_e = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[u], b[u].x, acc_e, 0, 0, 0);
            acc_o = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[u], b[u].y, acc_o, 0, 0, 0);
        }
    }
    double te = acc_e[0], to = acc_o[0];           // row 0 of D lives in lanes 0..15, register 0
    if (kq == 0) {
        for (; i < r1; ++i) {                        // ragged rows (fewer than 4*UNROLL left)
            const d2 a = *reinterpret_cast<const d2*>(Ap + i * n);
            const double x = r[i];
            te += a.x * x; to += a.y * x;
        }
        d2 o; o.x = te; o.y = to;
        *reinterpret_cast<d2*>(slab + (int64_t)blockIdx.y * n + j0 + 2 * c) = o;
    }
}

__global__ void fill(double* p, int64_t n, unsigned seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        p[i] = (double)(z >> 11) / 9007199254740992.0 - 0.5;
    }
}

template <typename F> double time_ms(F launch, int reps = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const int64_t m = argc > 1 ? atoll(argv[1]) : 16384, n = argc > 2 ? atoll(argv[2]) : 65536;
    double *A, *r, *slab1, *slab2;
    const int max_slices = 64;
    CK(hipMalloc(&A, sizeof(double) * m * n)); CK(hipMalloc(&r, sizeof(double) * m));
    CK(hipMalloc(&slab1, sizeof(double) * max_slices * n)); CK(hipMalloc(&slab2, sizeof(double) * max_slices * n));
    fill<<<4096, 256>>>(A, m * n, 1u); fill<<<64, 256>>>(r, m, 7u); CK(hipDeviceSynchronize());
    printf("A %lld x %lld fp64 (%.2f GiB)\n", (long long)m, (long long)n, m * n * 8.0 / (1 << 30));
    const double gb = m * n * 8.0 / 1e9;
    std::vector<double> h1(n), h2(n);
    for (int slices : {4, 8, 16, 32}) {
        const int64_t rps = (m + slices - 1) / slices;
        dim3 gv((unsigned)((n / 2 + 255) / 256), slices), gm((unsigned)((n / 32 + 3) / 4), slices);
        double t_v4 = time_ms([&] { hipLaunchKernelGGL(gemvT_valu<4>, gv, dim3(256), 0, 0, A, r, slab1, m, n, rps); });
        double t_v8 = time_ms([&] { hipLaunchKernelGGL(gemvT_valu<8>, gv, dim3(256), 0, 0, A, r, slab1, m, n, rps); });
        double t_m2 = time_ms([&] { hipLaunchKernelGGL(gemvT_mfma<2>, gm, dim3(256), 0, 0, A, r, slab2, m, n, rps); });
        double t_m4 = time_ms([&] { hipLaunchKernelGGL(gemvT_mfma<4>, gm, dim3(256), 0, 0, A, r, slab2, m, n, rps); });
        double t_m8 = time_ms([&] { hipLaunchKernelGGL(gemvT_mfma<8>, gm, dim3(256), 0, 0, A, r, slab2, m, n, rps); });
        // compare slice 0 of the two forms
        CK(hipMemcpy(h1.data(), slab1, sizeof(double) * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), slab2, sizeof(double) * n, hipMemcpyDeviceToHost));
        double dev = 0, mag = 0;
        for (int64_t j = 0; j < n; ++j) { dev = std::max(dev, std::fabs(h1[j] - h2[j])); mag = std::max(mag, std::fabs(h1[j])); }
        printf("slices=%2d  VALU u4 %.3f ms %6.0f GB/s | VALU u8 %.3f ms %6.0f GB/s | MFMA u2 %.3f ms %6.0f GB/s | MFMA u4 %.3f ms %6.0f GB/s | MFMA u8 %.3f ms %6.0f GB/s | max|dev| %.2e (|g| %.2e)\n",
               slices, t_v4, gb / t_v4 * 1e3, t_v8, gb / t_v8 * 1e3, t_m2, gb / t_m2 * 1e3, t_m4, gb / t_m4 * 1e3, t_m8,
               gb / t_m8 * 1e3, dev, mag);
    }
    return 0;
}
