// poll_probe.hip - what one look at the device costs the host, four ways (a probe, not product code).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/poll_probe tools/poll_probe.hip && /tmp/poll_probe
// Behind a short kernel on a stream:
//   A  hipMemcpyAsync(512 B, D2H, pinned) + hipStreamSynchronize            (zf_solver_poll without a trace)
//   B  the same with 66 KB                                                   (zf_solver_poll with the trace ring)
//   C  a one-workgroup kernel that copies 512 B into mapped host memory + hipStreamSynchronize
//   D  the same kernel, then a token; the host spins on the token in host memory (no synchronize)
//   E  as D with 66 KB copied by the kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void work(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}

__global__ void mirror(const unsigned long long* src, unsigned long long* dst, int words, unsigned long long* token, unsigned long long seq) {
    for (int i = threadIdx.x; i < words; i += blockDim.x)
        __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && token) __hip_atomic_store(token, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const int n = 1 << 16;
    double* buf;
    CK(hipMalloc(&buf, n * 8));
    CK(hipMemset(buf, 0, n * 8));
    const size_t big = 66 * 1024;
    unsigned long long* ctl;
    CK(hipMalloc(&ctl, big));
    CK(hipMemset(ctl, 1, big));
    unsigned char* pinned;
    CK(hipHostMalloc(&pinned, big + 4096, hipHostMallocMapped));
    memset(pinned, 0, big + 4096);
    unsigned long long* pin_dev;
    CK(hipHostGetDevicePointer((void**)&pin_dev, pinned, 0));
    volatile unsigned long long* token = reinterpret_cast<volatile unsigned long long*>(pinned + big);
    unsigned long long* token_dev = pin_dev + big / 8;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long seq = 0;
    const char* names[5] = {"A memcpyAsync 512 B + sync", "B memcpyAsync 66 KB + sync", "C kernel mirror 512 B + sync",
                            "D kernel mirror 512 B + token spin", "E kernel mirror 66 KB + token spin"};
    for (int passes = 1; passes <= 4; passes += 3) {
        for (int mode = 0; mode < 5; ++mode) {
            std::vector<double> t;
            for (int r = 0; r < reps + 50; ++r) {
                CK(hipStreamSynchronize(st));
                const double t0 = now_us();
                for (int p = 0; p < passes; ++p) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, buf, n);
                const double t1 = now_us();
                if (mode == 0) {
                    CK(hipMemcpyAsync(pinned, ctl, 512, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                } else if (mode == 1) {
                    CK(hipMemcpyAsync(pinned, ctl, big, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                } else if (mode == 2) {
                    hipLaunchKernelGGL(mirror, dim3(1), dim3(256), 0, st, ctl, pin_dev, 64, (unsigned long long*)nullptr, 0ull);
                    CK(hipStreamSynchronize(st));
                } else {
                    ++seq;
                    hipLaunchKernelGGL(mirror, dim3(1), dim3(256), 0, st, ctl, pin_dev, mode == 3 ? 64 : (int)(big / 8), token_dev, seq);
                    while (*token != seq) __builtin_ia32_pause();
                }
                const double t2 = now_us();
                if (r >= 50) t.push_back(t2 - t0), (void)t1;
            }
            std::sort(t.begin(), t.end());
            printf("passes %d  %-40s median %7.2f us  p10 %7.2f  p90 %7.2f\n", passes, names[mode], t[t.size() / 2], t[t.size() / 10],
                   t[t.size() * 9 / 10]);
        }
    }
    // the launches alone (no look at the device): host time of 4 launches, and their completion by synchronize
    {
        std::vector<double> t;
        for (int r = 0; r < reps; ++r) {
            CK(hipStreamSynchronize(st));
            const double t0 = now_us();
            for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, buf, n);
            CK(hipStreamSynchronize(st));
            t.push_back(now_us() - t0);
        }
        std::sort(t.begin(), t.end());
        printf("4 launches + synchronize (no copy)                    median %7.2f us\n", t[t.size() / 2]);
    }
    return 0;
}
