// event_probe.hip - what timing a kernel costs the queue (a probe, not product code).
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/event_probe tools/event_probe.hip
// N back-to-back launches of a kernel of ~T us on one stream, timed on the host from first launch to synchronize:
//   A  plain launches
//   B  hipEventRecord before and after every launch (what zf_solver's timing mode does)
//   C  hipExtLaunchKernelGGL with a start and a stop event ATTACHED to the launch (no packets of their own)
//   D  as C, and a second stream waits for every stop event (hipStreamWaitEvent) and runs a tiny kernel behind it
// and for B, C the durations the events report against the kernel's own clock.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void spin(long long ticks, long long* out) {   // ticks of the 100 MHz clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = wall_clock64() - t0;
}
__global__ void tiny(int* p) {
    if (p && threadIdx.x == 0) *p += 1;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int N = 16, reps = 200;
    const double kernel_us = argc > 1 ? atof(argv[1]) : 100.0;
    const long long ticks = (long long)(kernel_us * 100.0);
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    std::vector<hipEvent_t> e0(N), e1(N);
    for (int i = 0; i < N; ++i) {
        CK(hipEventCreate(&e0[i]));
        CK(hipEventCreate(&e1[i]));
    }
    int* counter;
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(counter, 0, 4));
    const char* names[4] = {"A plain", "B hipEventRecord pairs", "C events attached to the launch", "D attached + second stream waits"};
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> t;
        double ev_sum = 0.0;
        int ev_n = 0;
        for (int r = 0; r < reps + 10; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) {
                if (mode == 0) {
                    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, (long long*)nullptr);
                } else if (mode == 1) {
                    CK(hipEventRecord(e0[i], st));
                    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, (long long*)nullptr);
                    CK(hipEventRecord(e1[i], st));
                } else {
                    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, e0[i], e1[i], 0, ticks, (long long*)nullptr);
                    if (mode == 3) {
                        CK(hipStreamWaitEvent(st2, e1[i], 0));
                        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st2, counter);
                    }
                }
            }
            CK(hipStreamSynchronize(st));
            const double t1 = now_us();
            CK(hipStreamSynchronize(st2));
            if (r >= 10) {
                t.push_back(t1 - t0);
                if (mode >= 1)
                    for (int i = 0; i < N; ++i) {
                        float ms = 0.f;
                        CK(hipEventElapsedTime(&ms, e0[i], e1[i]));
                        ev_sum += ms * 1e3;
                        ev_n += 1;
                    }
            }
        }
        std::sort(t.begin(), t.end());
        const double med = t[t.size() / 2];
        printf("%-40s %d launches of %.0f us: median %8.1f us = %6.2f us per launch over the kernel", names[mode], N, kernel_us, med,
               med / N - kernel_us);
        if (ev_n) printf("; events report %7.2f us per kernel", ev_sum / ev_n);
        printf("\n");
    }
    int h = 0;
    CK(hipMemcpy(&h, counter, 4, hipMemcpyDeviceToHost));
    printf("second-stream kernels that ran: %d (expected %d)\n", h, N * (reps + 10));
    return 0;
}
