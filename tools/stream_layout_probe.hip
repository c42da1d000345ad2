// Probe (no product code): does the NUMBER of concurrent streams matter at equal bytes?  A chained pass moves 48 B per
// element as 4 read streams + 2 write streams of 16-byte pieces (x_k, x_{k-1}, d, c -> x+, x+_prev).  The same bytes as
// 2 read streams + 1 write stream of 32-byte pieces (iterate pairs interleaved, d and c interleaved) - would they move
// faster?   hipcc -O3 --offload-arch=gfx950 tools/stream_layout_probe.hip -o stream_layout_probe && ./stream_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// 4 in + 2 out, 16 B per lane and stream; tiles of 256 x 4 units interleaved over the grid as the trial kernel does
__global__ __launch_bounds__(256) void k6(const d2* __restrict__ a, const d2* __restrict__ b, const d2* __restrict__ c, const d2* __restrict__ d,
                                           d2* __restrict__ o0, d2* __restrict__ o1, long units, int tiles_per_wg) {
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)t * gridDim.x + blockIdx.x;
        const long base = tile * 1024 + threadIdx.x;
        if (base + 768 >= units) break;
        d2 va[4], vb[4], vc[4], vd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = __builtin_nontemporal_load(a + base + u * 256); vb[u] = __builtin_nontemporal_load(b + base + u * 256);
            vc[u] = __builtin_nontemporal_load(c + base + u * 256); vd[u] = __builtin_nontemporal_load(d + base + u * 256);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_nontemporal_store(va[u] * vc[u] + vb[u], o0 + base + u * 256);
            __builtin_nontemporal_store(vb[u] * vd[u] + va[u], o1 + base + u * 256);
        }
    }
}
// 2 in + 1 out, 32 B per lane and stream
__global__ __launch_bounds__(256) void k3(const d4* __restrict__ ab, const d4* __restrict__ cd, d4* __restrict__ o, long units, int tiles_per_wg) {
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)t * gridDim.x + blockIdx.x;
        const long base = tile * 1024 + threadIdx.x;
        if (base + 768 >= units) break;
        d4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = __builtin_nontemporal_load(ab + base + u * 256); y[u] = __builtin_nontemporal_load(cd + base + u * 256); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d4 r;
            r.x = x[u].x * y[u].x + x[u].y; r.y = x[u].y * y[u].y + x[u].x; r.z = x[u].z * y[u].z + x[u].w; r.w = x[u].w * y[u].w + x[u].z;
            __builtin_nontemporal_store(r, o + base + u * 256);
        }
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 100000000;   // elements
    const long units = n / 2;                               // 16-byte units per stream (k6) = 32-byte units per stream (k3)
    double* buf[7];
    for (int k = 0; k < 7; ++k) { CK(hipMalloc(&buf[k], n * 8 * (k == 6 ? 6 : 1))); CK(hipMemset(buf[k], 0, n * 8 * (k == 6 ? 6 : 1))); }
    double* big = buf[6];   // 6 n doubles: ab (2n), cd (2n), o (2n)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int T : {8, 24}) {
        const int grid = (int)((units / 1024 + T - 1) / T);
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k6, dim3(grid), dim3(256), 0, 0, (const d2*)buf[0], (const d2*)buf[1], (const d2*)buf[2], (const d2*)buf[3], (d2*)buf[4], (d2*)buf[5], units, T);
                else hipLaunchKernelGGL(k3, dim3(grid), dim3(256), 0, 0, (const d4*)big, (const d4*)(big + 2 * n), (d4*)(big + 4 * n), units, T);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            printf("n = %ld, T = %2d, %s: mean %.4f ms (best %.4f) = %.0f GB/s of 48 B x n\n", n, T,
                   mode == 0 ? "4 read + 2 write streams x 16 B" : "2 read + 1 write streams x 32 B", sum / 10, best, 48.0 * n / (sum / 10 * 1e-3) / 1e9);
        }
    }
    return 0;
}
