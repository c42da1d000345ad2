// Probe (no product code): can pass k+1 of a chain of dependent launches start its workgroups while pass k is still
// finalising?  Each workgroup j of launch s waits for flag[j] >= s-1 (its own predecessor only - the element-wise data
// dependency of a chained pass), works ~W us, publishes flag[j] = s; the last arriver then spends ~T us alone (the
// finalisation tail).  Modes: 0 plain launches on one stream; 1 hipExtAnyOrderLaunch on one stream; 2 alternating
// between two streams; 3 between three.   hipcc -O3 --offload-arch=gfx950 tools/runahead_probe.hip -o runahead_probe && ./runahead_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned* flags, unsigned* tickets, long long* stamps, int seq, int work, int tail,
                                             unsigned spin_limit) {
    __shared__ double pad[6800];   // ~54 KB: two workgroups per CU, as the full chain
    const int j = blockIdx.x;
    long long t0 = wall_clock64();
    unsigned spins = 0;
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(flags + j * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(seq - 1)) {
            if (++spins > spin_limit) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    long long t1 = wall_clock64();
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < work; ++i) { a = a * b + 1e-9; b = b * 0.9999999 + 1e-8; }
    pad[threadIdx.x] = a + b;
    __syncthreads();
    long long t2 = wall_clock64();
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(flags + j * 32, (unsigned)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_fetch_add(tickets + (seq & 3) * 32, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == gridDim.x - 1);
        if (last) {
            __hip_atomic_store(tickets + (seq & 3) * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned v = 0;   // a chain of dependent trips to memory
            for (int i = 0; i < tail; ++i) v += __hip_atomic_fetch_add(tickets + 4 * 32 + (v & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pad[1] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && stamps) {
        long long* s = stamps + ((long long)seq * gridDim.x + j) * 4;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = wall_clock64() | ((long long)last << 62) | ((long long)(spins > spin_limit) << 61);
    }
    if (pad[threadIdx.x] == 123.456) flags[0] = 7;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 489, work = argc > 2 ? atoi(argv[2]) : 6000, tail = argc > 3 ? atoi(argv[3]) : 12;
    const int L = 40;
    unsigned *flags, *tickets; long long* stamps;
    CK(hipMalloc(&flags, grid * 128)); CK(hipMalloc(&tickets, 8 * 128)); CK(hipMalloc(&stamps, (size_t)(L + 2) * grid * 32));
    hipStream_t st[3]; for (int k = 0; k < 3; ++k) CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    std::vector<long long> h((size_t)(L + 2) * grid * 4);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(flags, 0, grid * 128)); CK(hipMemset(tickets, 0, 8 * 128)); CK(hipMemset(stamps, 0, (size_t)(L + 2) * grid * 32));
            CK(hipDeviceSynchronize());
            auto w0 = std::chrono::steady_clock::now();
            for (int s = 1; s <= L; ++s) {
                if (mode == 1) hipExtLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, st[0], nullptr, nullptr, hipExtAnyOrderLaunch, flags, tickets, stamps, s, work, tail, 2000000u);
                else hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, st[mode == 2 ? (s & 1) : mode == 3 ? (s % 3) : 0], flags, tickets, stamps, s, work, tail, 2000000u);
            }
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            if (rep < 2) continue;
            const long long M = (1ll << 61) - 1;
            long long base = h[(size_t)1 * grid * 4];
            for (int j = 0; j < grid; ++j) base = std::min(base, h[((size_t)1 * grid + j) * 4]);
            int timeouts = 0; double body = 0, waited = 0;
            std::vector<double> first_start(L + 1), last_end(L + 1), last_body(L + 1);
            for (int s = 1; s <= L; ++s) {
                double fs = 1e18, le = 0, lb = 0;
                for (int j = 0; j < grid; ++j) {
                    const long long* q = &h[((size_t)s * grid + j) * 4];
                    fs = std::min(fs, (double)(q[0] - base)); le = std::max(le, (double)((q[3] & M) - base)); lb = std::max(lb, (double)(q[2] - base));
                    timeouts += (int)((q[3] >> 61) & 1); body += q[2] - q[1]; waited += q[1] - q[0];
                }
                first_start[s] = fs / 100; last_end[s] = le / 100; last_body[s] = lb / 100;
            }
            printf("mode %d (%s): wall %.0f us for %d launches = %.1f us each; device: period %.1f us, body %.1f us, waited at start %.1f us mean, tail after last body %.1f us, timeouts %d\n",
                   mode, mode == 0 ? "one stream" : mode == 1 ? "hipExtAnyOrderLaunch" : mode == 2 ? "two streams" : "three streams", wall, L, wall / L,
                   (last_end[L] - last_end[L / 2]) / (L - L / 2), body / 100 / (L * grid), waited / 100 / (L * grid), last_end[L] - last_body[L], timeouts);
            printf("   launch s: first start / last body end / end (us):");
            for (int s = L - 3; s <= L; ++s) printf("  %d: %.1f / %.1f / %.1f", s, first_start[s], last_body[s], last_end[s]);
            printf("\n");
        }
    }
    return 0;
}
