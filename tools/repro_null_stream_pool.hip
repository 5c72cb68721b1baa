// tools/repro_null_stream_pool.hip -- ADVICE r03: is it true that a stream-ordered pool with a finite release threshold pulls memory from
// under kernels that are still queued on the legacy NULL stream?  (Round 3 saw garbage in examples/rowband_demo.c with the DEFAULT pool and
// answered with a private pool that never releases; the diagnosis was never confirmed.)  This program does what the row-band path does --
// scratch = hipMallocAsync(stream), a kernel fills it, a kernel consumes it into `out`, hipFreeAsync(scratch, stream), and straight away
// another allocation of a different size that is overwritten with garbage -- a few hundred times without any synchronisation, on the NULL
// stream and on a created stream, from the default pool and from a private pool with release thresholds 0 / 64 MiB / UINT64_MAX, and checks
// every byte of `out` at the end.
//   hipcc --offload-arch=gfx950 -O2 -o tools/repro_null_stream_pool tools/repro_null_stream_pool.hip && tools/repro_null_stream_pool
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(unsigned *p, size_t n, unsigned tag) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = tag ^ (unsigned)i; }
__global__ void k_consume(const unsigned *p, unsigned *out, size_t n, int spin)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        unsigned v = p[i];
        for (int s = 0; s < spin; ++s) v = v * 1664525u + 1013904223u, v = (v - 1013904223u) * 4276115653u;   // x -> x (inverse LCG step): keeps the kernel long
        out[i] = v;
    }
}

static int run(const char *name, hipMemPool_t pool, hipStream_t st, int iters)
{
    const size_t n = 16u << 20;                                        // 64 MiB of scratch per iteration
    unsigned *out = nullptr;
    CK(hipMalloc(&out, n * 4 * (size_t)iters / 8 + n * 4));
    std::vector<unsigned> host(n);
    long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned *a = nullptr, *b = nullptr;
        if (pool) CK(hipMallocFromPoolAsync((void **)&a, n * 4, pool, st)); else CK(hipMallocAsync((void **)&a, n * 4, st));
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, a, n, 0xA5A50000u + it);
        hipLaunchKernelGGL(k_consume, dim3(256), dim3(256), 0, st, a, out + (size_t)(it % 8) * (n / 8), n / 8, 200);
        CK(hipFreeAsync(a, st));
        const size_t nb = n / 2 + (size_t)(it % 5) * (1u << 20);
        if (pool) CK(hipMallocFromPoolAsync((void **)&b, nb * 4, pool, st)); else CK(hipMallocAsync((void **)&b, nb * 4, st));
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, b, nb, 0xDEAD0000u);
        CK(hipFreeAsync(b, st));
        if (it % 8 == 7) {                                             // check the eight slices written since the last check
            CK(hipMemcpyAsync(host.data(), out, n * 4, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            for (int s = 0; s < 8; ++s)
                for (size_t i = 0; i < n / 8; ++i)
                    if (host[(size_t)s * (n / 8) + i] != ((0xA5A50000u + (it - 7 + s)) ^ (unsigned)i)) ++bad;
        }
    }
    CK(hipStreamSynchronize(st));
    CK(hipFree(out));
    printf("%-58s %s (%lld wrong words in %d iterations)\n", name, bad ? "CORRUPT" : "ok", bad, iters);
    return bad != 0;
}

int main()
{
    int rc = 0;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    rc |= run("default pool, NULL stream", nullptr, nullptr, 400);
    rc |= run("default pool, created stream", nullptr, st, 400);
    for (uint64_t keep : {0ull, 64ull << 20, ~0ull}) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = 0;
        hipMemPool_t pool;
        CK(hipMemPoolCreate(&pool, &props));
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
        char name[96];
        snprintf(name, sizeof(name), "private pool, release threshold %llu MiB, NULL stream", (unsigned long long)(keep >> 20));
        rc |= run(name, pool, nullptr, 400);
        snprintf(name, sizeof(name), "private pool, release threshold %llu MiB, created stream", (unsigned long long)(keep >> 20));
        rc |= run(name, pool, st, 400);
        CK(hipMemPoolDestroy(pool));
    }
    return rc;
}
