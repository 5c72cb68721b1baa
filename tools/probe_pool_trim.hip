// tools/probe_pool_trim.hip -- what does a private stream-ordered pool give back, and when?  (round 4: the scratch pool's threshold / trim)
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_pool_trim tools/probe_pool_trim.hip && tools/probe_pool_trim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static void show(const char *what, hipMemPool_t pool)
{
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    uint64_t res = 0, used = 0;
    hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &res);
    hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &used);
    printf("%-44s free %8.1f MiB   pool reserved %8.1f MiB used %8.1f MiB\n", what, fr / 1048576.0, res / 1048576.0, used / 1048576.0);
}
int main()
{
    for (uint64_t keep : {0ull, 64ull << 20, ~0ull})
        for (int null_stream = 0; null_stream < 2; ++null_stream) {
            hipStream_t st = nullptr;
            if (!null_stream) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned; props.location.type = hipMemLocationTypeDevice; props.location.id = 0;
            hipMemPool_t pool; CK(hipMemPoolCreate(&pool, &props));
            hipError_t e = hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            printf("== release threshold %llu MiB (set: %s), %s stream\n", (unsigned long long)(keep >> 20), hipGetErrorString(e), null_stream ? "NULL" : "created");
            show("start", pool);
            void *p; CK(hipMallocFromPoolAsync(&p, 512u << 20, pool, st)); CK(hipMemsetAsync(p, 1, 512u << 20, st));
            CK(hipStreamSynchronize(st)); show("512 MiB allocated", pool);
            CK(hipFreeAsync(p, st)); show("freed (not synchronised)", pool);
            CK(hipStreamSynchronize(st)); show("stream synchronised", pool);
            CK(hipDeviceSynchronize()); show("device synchronised", pool);
            e = hipMemPoolTrimTo(pool, 0); printf("hipMemPoolTrimTo(0): %s\n", hipGetErrorString(e)); show("trimmed", pool);
            CK(hipMemPoolDestroy(pool)); 
            if (st) CK(hipStreamDestroy(st));
        }
    return 0;
}
