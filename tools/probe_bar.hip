// tools/probe_bar.hip -- can the host store directly into device memory (large BAR) on this box?  Decides how the streaming
// "doorbell" is rung: a posted PCIe write into fine-grained device memory (the device polls its own memory) or a word in pinned
// host memory (the device polls across PCIe).  Also times: host->device doorbell visibility and device->host completion, with a
// resident kernel that echoes a sequence number.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_bar tools/probe_bar.hip && tools/probe_bar
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

// one wave: wait for *bell == s (s = 1, 2, ...), then write s to *done; exits at s == last or after ~2 s without a ring
__global__ void echo(volatile unsigned long long *bell, volatile unsigned long long *done, unsigned long long last)
{
    unsigned long long s = 1;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    while (s <= last) {
        unsigned long long v = __hip_atomic_load((unsigned long long *)bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v >= s) {
            __hip_atomic_store((unsigned long long *)done, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ++s;
        } else if (__builtin_amdgcn_s_memrealtime() - t_start > 400000000ull) break;          // 4 s at 100 MHz: never hang the box
    }
}

static void run(const char *name, volatile unsigned long long *bell_host, unsigned long long *bell_dev, volatile unsigned long long *done_host,
                unsigned long long *done_dev)
{
    const unsigned long long iters = 20000;
    *bell_host = 0; *done_host = 0;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, st, bell_dev, done_dev, iters);
    std::vector<double> us;
    for (unsigned long long s = 1; s <= iters; ++s) {
        const auto t0 = std::chrono::steady_clock::now();
        *bell_host = s;
        __sync_synchronize();
        while (*done_host < s) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) { printf("%s: no echo for seq %llu\n", name, s); goto out; }
        }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
out:
    CK(hipStreamSynchronize(st));
    if (!us.empty()) {
        std::sort(us.begin(), us.end());
        printf("%-60s round trip p50 %.2f us  p99 %.2f us  min %.2f us (%zu echoes)\n", name, us[us.size() / 2], us[us.size() * 99 / 100], us[0], us.size());
    }
}

int main()
{
    unsigned long long *pin;                                     // pinned, coherent host memory: [0] bell, [16] done
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin), 4096, hipHostMallocCoherent | hipHostMallocMapped));
    unsigned long long *pin_dev;
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&pin_dev), pin, 0));
    run("bell in pinned host memory, done in pinned host memory", pin, pin_dev, pin + 16, pin_dev + 16);

    unsigned long long *fg = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&fg), 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        CK(hipMemset(fg, 0, 4096));
        CK(hipDeviceSynchronize());
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            volatile unsigned long long *h = fg;
            *h = 0;                                              // faults here when device memory is not host-mapped
            unsigned long long back = *h;
            printf("host can store to / load from fine-grained device memory (read back %llu)\n", back);
            run("bell in fine-grained DEVICE memory, done in pinned host memory", fg, fg, pin + 16, pin_dev + 16);
        } else {
            printf("host access to fine-grained device memory FAULTS: the doorbell has to live in pinned host memory\n");
        }
    }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, isLargeBar %d\n", p.name, p.isLargeBar);
    return 0;
}
