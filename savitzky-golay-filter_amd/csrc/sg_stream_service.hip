// sg_stream_service.hip -- the per-tick streaming path without a kernel launch per tick.
//
// savgol_streambank_push costs a launch and a synchronise per tick: 14.6 us p50 wall from plain C at 65 536 streams, of which
// 3.8 us is the kernel (profiles/r01_stream_kernel_stats.csv).  The service keeps one kernel RESIDENT instead: 256 waves (one
// per CU; a lane owns 4 adjacent streams) that hold the 2n in-flight accumulators of their streams in registers and wait for
// a doorbell.  A tick is then
//     host  : {in, out} + sequence number -> mailbox                     (one posted PCIe write when the mailbox lives in
//                                                                          host-writable device memory, large BAR)
//     wave  : sees the new sequence number, loads its 4 samples per lane (system-scope 16-byte load), advances every
//             accumulator by one tap, stores the finished output (write-through), stores the sample into the bank's ring
//             in HBM, then writes the sequence number into ITS word of a done array in pinned host memory
//     host  : spins on the done array until every wave has reported
// No launch, no stream synchronise, no atomics.  Arithmetic per output is the reference's, exactly as in the block-push kernel
// (sg_stream_roll.hip, bank_accroll_item): one accumulator starting at 0, taps ascending, multiply and add rounded separately,
// then * dt_inv -> bit-identical to savgol_stream_push (reference src/savgol_stream.c:25-38, 152-178).
// The ring in HBM is kept current every tick, so stopping the service (or its idle time-out) needs no state hand-back: the
// stream-ordered calls (push, push_block, flush, save ...) continue from where the service stopped.
//
// Safety: a wave that sees no doorbell for `idle_ms` leaves on its own (reporting EXITED), so nothing can spin forever; the next
// tick restarts the service.  While it is resident, a DEVICE-WIDE synchronise waits for it (up to idle_ms) -- synchronise
// streams, not the device, or stop the service first.
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include <csetjmp>
#include <csignal>
#include <immintrin.h>

#include "sg_internal.h"
#include "sg_pk.hpp"
#include "sg_runtime.hpp"
#include "sg_stream.hpp"

namespace sg {

constexpr unsigned SERVICE_EXITED = 0xffffffffu;
constexpr unsigned long long SERVICE_CMD_TICK = 0, SERVICE_CMD_STOP = 1;      // top two bits of ServiceMailbox::seq_b
constexpr int SERVICE_STREAMS_PER_WAVE = 256;                // 64 lanes x 4 streams
constexpr unsigned SERVICE_MAX_WAVES = 1024;                 // must all be resident: 4 per CU at most
constexpr unsigned SERVICE_BELLS = 64;                       // most copies of the mailbox (one 64-byte line each): wave w polls copy w % bells
                                                             // (256 waves hammering ONE uncached line queue behind each other)

// One 64-byte line, written by the host, polled by the waves with two 16-byte loads issued together: each half carries the
// sequence number (the second one with the command in its top bits), and a wave acts only when both halves show the number
// it is waiting for -- so it never pairs a new number with an old pointer, and needs no second round trip for the payload.
struct alignas(64) ServiceMailbox {
    const float       *in;
    unsigned long long seq_a;
    float             *out;
    unsigned long long seq_b;                                // sequence number | command << 62
};
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

struct ServiceArgs {
    ServiceMailbox *bell;                                    // device view of the mailbox
    unsigned       *done;                                    // device view of the pinned done array, one word per wave
    float          *ring;                                    // the bank's [ws][streams] ring in HBM
    size_t          streams;
    int             wp0;                                     // write position when the service starts
    unsigned long long received0;
    float           dt_inv;
    unsigned        bells;                                   // mailbox copies in use (<= SERVICE_BELLS)
    unsigned long long idle_ticks;                           // s_memrealtime ticks (100 MHz) without a doorbell before a wave leaves
};

template <int N> struct ServiceTaps { f32x2 w[N + 1]; };

__device__ __forceinline__ f32x4 load16_system(const float *p)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store16_system(float *p, const f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// every accumulator one tap further with sample pair x: slot a holds the output that has seen a samples; w[a]*x is added and
// the sum moves to slot a+1 (walked from the top down).  Returns the output that has just seen its last tap.
template <int N>
__device__ __forceinline__ f32x2 advance(f32x2 (&acc)[2 * N + 1], const ServiceTaps<N> &taps, const f32x2 x)
{
    constexpr int WS = 2 * N + 1;
    f32x2 done;
    static_for<WS>([&](auto ic) -> bool {
        constexpr int a = WS - 1 - decltype(ic)::value;
        f32x2 p;
        if constexpr ((a & 1) == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(taps.w[a >> 1]), "v"(x));
        else                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(taps.w[a >> 1]), "v"(x));
        if constexpr (a == WS - 1)  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(done) : "v"(acc[a]), "v"(p));
        else if constexpr (a == 0)  asm volatile("v_pk_add_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(acc[1]) : "v"(p));
        else                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[a + 1]) : "v"(acc[a]), "v"(p));
        return true;
    });
    return done;
}

template <int N>
__global__ __launch_bounds__(64) void sg_bank_service_kernel(const ServiceArgs a, const ServiceTaps<N> taps)
{
    constexpr int WS = 2 * N + 1;
    const int lane = threadIdx.x;
    const unsigned wave = blockIdx.x;
    ServiceMailbox *const bell = a.bell + (wave % a.bells);
    const size_t s0 = ((size_t)wave * 64 + lane) * 4;
    const bool live = s0 < a.streams;                        // streams is a multiple of 4

    // warm-up: the last 2N samples of the ring (oldest first) through the accumulators, so that the next sample completes
    // the output whose window is {those 2N, the new one}
    f32x2 acc0[WS], acc1[WS];
#pragma unroll
    for (int i = 0; i < WS; ++i) { acc0[i] = f32x2{0.0f, 0.0f}; acc1[i] = f32x2{0.0f, 0.0f}; }
    for (int k = 2 * N; k >= 1; --k) {                       // sample -k of the history sits in slot (wp0 - k) mod WS
        int slot = a.wp0 - k;
        slot = slot < 0 ? slot + WS : slot;
        f32x4 x = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (live) x = *reinterpret_cast<const f32x4 *>(a.ring + (size_t)slot * a.streams + s0);
        (void)advance<N>(acc0, taps, f32x2{x.x, x.y});
        (void)advance<N>(acc1, taps, f32x2{x.z, x.w});
    }

    int wp = a.wp0;
    unsigned long long received = a.received0;
    unsigned long long seq = 1;
    unsigned long long idle_since = __builtin_amdgcn_s_memrealtime();
    const f32x2 s2 = f32x2{a.dt_inv, a.dt_inv};
    for (;;) {
        u64x2 ha, hb;                                        // {in, seq_a}, {out, seq_b}
        asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(ha), "=&v"(hb) : "v"(bell) : "memory");
        auto uniform64 = [](unsigned long long v) -> unsigned long long {      // readfirstlane returns a SIGNED int: widen through unsigned
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
            return (unsigned long long)lo | ((unsigned long long)hi << 32);
        };
        const unsigned long long sa = uniform64(ha.y), sb = uniform64(hb.y);
        if (sa != seq || (sb & 0x3fffffffffffffffull) != seq) {
            if (__builtin_amdgcn_s_memrealtime() - idle_since > a.idle_ticks) break;
            continue;
        }
        if ((sb >> 62) == SERVICE_CMD_STOP) break;
        const float *in = reinterpret_cast<const float *>(uniform64(ha.x));
        float *out = reinterpret_cast<float *>(uniform64(hb.x));
        ++received;
        if (live) {
            const f32x4 x = load16_system(in + s0);
            const f32x2 y0 = advance<N>(acc0, taps, f32x2{x.x, x.y});
            const f32x2 y1 = advance<N>(acc1, taps, f32x2{x.z, x.w});
            if (received >= (unsigned long long)WS) {        // uniform: the windows are full (reference :166-170)
                const f32x2 o0 = y0 * s2, o1 = y1 * s2;
                store16_system(out + s0, f32x4{o0.x, o0.y, o1.x, o1.y});
            }
            *reinterpret_cast<f32x4 *>(a.ring + (size_t)wp * a.streams + s0) = x;      // the ring in HBM stays current
        }
        wp = wp + 1 == WS ? 0 : wp + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's output has left for memory before it reports
        if (lane == 0) __hip_atomic_store(a.done + wave, (unsigned)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++seq;
        idle_since = __builtin_amdgcn_s_memrealtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(a.done + wave, SERVICE_EXITED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct BankService {
    ServiceMailbox *bell_host = nullptr, *bell_dev = nullptr;
    bool            bell_in_device = false;
    volatile unsigned *done_host = nullptr;
    unsigned       *done_dev = nullptr;
    unsigned        waves = 0;
    unsigned long long seq = 0;                              // ticks posted to the CURRENT kernel instance
    hipStream_t     stream = nullptr;
    bool            running = false;
    unsigned        idle_ms = 1000;
    unsigned        bells = 16;
};

template <int N>
static bool launch_service(int n, const ServiceArgs &args, const float *cw, unsigned waves, hipStream_t st)
{
    if (n == N) {
        ServiceTaps<N> taps;
        memset(&taps, 0, sizeof(taps));
        for (int k = 0; k < 2 * N + 1; ++k) { if (k & 1) taps.w[k >> 1].y = cw[k]; else taps.w[k >> 1].x = cw[k]; }
        hipLaunchKernelGGL((sg_bank_service_kernel<N>), dim3(waves), dim3(64), 0, st, args, taps);
        return true;
    }
    if constexpr (N < SAVGOL_MAX_HALF_WINDOW) return launch_service<N + 1>(n, args, cw, waves, st);
    else return false;
}

// How many waves of the service kernel for half window n the device holds at once: every wave of the service must be resident
// (a wave that is not scheduled never answers the doorbell), and the n = 32 instance keeps 260+ VGPRs of accumulators -- one wave
// per SIMD.  Occupancy API x compute units of THIS device (a partition may have fewer than 256).
template <int N>
static long service_resident_waves(int n, int cu_count)
{
    if (n == N) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sg_bank_service_kernel<N>, 64, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); return -1; }
        return (long)nb * cu_count;
    }
    if constexpr (N < SAVGOL_MAX_HALF_WINDOW) return service_resident_waves<N + 1>(n, cu_count);
    else return -1;
}

// OPT-IN since round 4 (SAVGOL_HIP_BAR_DOORBELL=1): the probe below swaps the process's SIGSEGV / SIGBUS
// handlers for the duration of one store, which a host with fault handlers of its own (Python's faulthandler, a JVM, a sanitizer
// run-time) should not have done to it by a filter library for 0.7 us per tick.  Without the switch the doorbell lives in pinned host
// memory (2.5 us per tick instead of 1.8) and no handler is ever touched.
// Can the host really store into fine-grained device memory on this box?  isLargeBar says the BAR covers the memory, not that
// the allocation is mapped for the CPU; a store into an unmapped one is a SIGSEGV / SIGBUS.  Probed ONCE per process under a
// guard (handlers saved and restored around the single store), because the alternative is to crash the caller.
// Multithreaded hosts: the probe runs once, under a mutex; the handler only jumps when the fault happened on the probing thread
// inside the probe window (thread-local flag) -- a fault of any other thread during that window puts the previous handler back
// and returns, so the faulting instruction runs again and reaches the handler it would have reached without us.
static sigjmp_buf g_probe_jmp;
static thread_local volatile sig_atomic_t tl_probing = 0;
static struct sigaction g_probe_old_segv, g_probe_old_bus;
static void probe_fault(int sig)
{
    if (tl_probing) siglongjmp(g_probe_jmp, 1);
    sigaction(sig, sig == SIGSEGV ? &g_probe_old_segv : &g_probe_old_bus, nullptr);
}
bool host_can_write_device_memory(void *p)          // also used by the small-call service (sg_k1d_misc.hip)
{
    static const bool opted_in = [] { const char *e = getenv("SAVGOL_HIP_BAR_DOORBELL"); return e && atoi(e) == 1; }();
    if (!opted_in) return false;
    static std::mutex mu;
    static int verdict = -1;                                 // -1 unknown, 0 no, 1 yes
    std::lock_guard<std::mutex> lock(mu);
    if (verdict >= 0) return verdict == 1;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = probe_fault;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &g_probe_old_segv);
    sigaction(SIGBUS, &sa, &g_probe_old_bus);
    if (sigsetjmp(g_probe_jmp, 1) == 0) {
        tl_probing = 1;
        volatile unsigned long long *w = static_cast<volatile unsigned long long *>(p);
        w[0] = 0x5347u;
        _mm_sfence();
        verdict = (w[0] == 0x5347u) ? 1 : 0;
        w[0] = 0;
        _mm_sfence();
    } else {
        verdict = 0;
    }
    tl_probing = 0;
    sigaction(SIGSEGV, &g_probe_old_segv, nullptr);
    sigaction(SIGBUS, &g_probe_old_bus, nullptr);
    return verdict == 1;
}

static double service_timeout_s()                            // SAVGOL_HIP_SERVICE_TIMEOUT_MS: how long a tick waits for every wave (tests shorten it)
{
    static const double v = [] { const char *e = getenv("SAVGOL_HIP_SERVICE_TIMEOUT_MS"); const int ms = e ? atoi(e) : 0; return ms > 0 ? ms * 1e-3 : 5.0; }();
    return v;
}

static void service_free(BankService *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->bell_in_device && s->bell_dev) (void)hipFree(s->bell_dev);
    if (!s->bell_in_device && s->bell_host) (void)hipHostFree(s->bell_host);
    if (s->done_host) (void)hipHostFree(const_cast<unsigned *>(s->done_host));
    delete s;
}

// (re)launch the resident kernel from the bank's current state
static int service_launch(SavgolStreamBank *bank, BankService *s, const char *who)
{
    for (unsigned b = 0; b < SERVICE_BELLS; ++b) { s->bell_host[b].in = nullptr; s->bell_host[b].seq_a = 0; s->bell_host[b].out = nullptr; s->bell_host[b].seq_b = 0; }
    _mm_sfence();
    for (unsigned w = 0; w < s->waves; ++w) s->done_host[w] = 0;
    s->seq = 0;
    ServiceArgs args;
    memset(&args, 0, sizeof(args));
    args.bell = s->bell_dev; args.done = s->done_dev; args.ring = bank->d_ring; args.streams = bank->streams;
    args.wp0 = bank->wp; args.received0 = bank->received; args.dt_inv = bank->dt_inv;
    args.idle_ticks = (unsigned long long)s->idle_ms * 100000ull;
    args.bells = s->bells;
    if (!launch_service<1>(bank->filter->config.half_window, args, bank->filter->center_weights, s->waves, s->stream)) {
        sg_set_error("%s: no service kernel for half_window %d", who, bank->filter->config.half_window);
        return -1;
    }
    if (!hip_ok(hipGetLastError(), who)) return -1;
    s->running = true;
    return 0;
}

}  // namespace sg

extern "C" {

int savgol_streambank_service_start(SavgolStreamBank *bank, unsigned idle_ms)
{
    const char *who = "savgol_streambank_service_start";
    if (!bank) { sg_set_error("%s: NULL bank", who); return -1; }
    if (bank->service) return 0;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != bank->device) { sg_set_error("%s: the bank lives on device %d, this thread uses %d", who, bank->device, cur); return -1; }
    if (bank->streams % 4 != 0) { sg_set_error("%s: the tick service needs a multiple of 4 streams (got %zu)", who, bank->streams); return -1; }
    const size_t waves = (bank->streams + sg::SERVICE_STREAMS_PER_WAVE - 1) / sg::SERVICE_STREAMS_PER_WAVE;
    if (waves > sg::SERVICE_MAX_WAVES) { sg_set_error("%s: at most %u streams (every wave must be resident)", who, sg::SERVICE_MAX_WAVES * sg::SERVICE_STREAMS_PER_WAVE); return -1; }
    sg::BankService *s = new sg::BankService();
    s->waves = (unsigned)waves;
    s->idle_ms = idle_ms ? idle_ms : 1000;
    hipDeviceProp_t prop;
    bool ok = sg::hip_ok(hipGetDeviceProperties(&prop, bank->device), who);
    if (ok) {
        const long fit = sg::service_resident_waves<1>(bank->filter->config.half_window, prop.multiProcessorCount);
        if (fit < (long)waves) {
            sg_set_error("%s: %zu streams need %zu resident waves but this device holds %ld of the half_window %d service kernel at once "
                         "(%d compute units); use savgol_streambank_push", who, bank->streams, waves, fit, bank->filter->config.half_window,
                         prop.multiProcessorCount);
            ok = false;
        }
    }
    ok = ok && sg::hip_ok(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), who);
    unsigned *done = nullptr;
    ok = ok && sg::hip_ok(hipHostMalloc(reinterpret_cast<void **>(&done), sizeof(unsigned) * waves, hipHostMallocCoherent | hipHostMallocMapped), who) &&
         sg::hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->done_dev), done, 0), who);
    s->done_host = done;
    // the mailbox: host-writable device memory where the whole of it is visible to the host (large BAR: the doorbell is one
    // posted write and the waves poll their own memory), pinned host memory otherwise
    if (ok && prop.isLargeBar) {
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, sizeof(sg::ServiceMailbox) * sg::SERVICE_BELLS, hipDeviceMallocFinegrained) == hipSuccess) {
            if (sg::host_can_write_device_memory(p)) {
                s->bell_dev = static_cast<sg::ServiceMailbox *>(p);
                s->bell_host = s->bell_dev;
                s->bell_in_device = true;
            } else (void)hipFree(p);
        } else (void)hipGetLastError();
    }
    if (ok && !s->bell_dev) {
        void *p = nullptr;
        ok = sg::hip_ok(hipHostMalloc(&p, sizeof(sg::ServiceMailbox) * sg::SERVICE_BELLS, hipHostMallocCoherent | hipHostMallocMapped), who) &&
             sg::hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->bell_dev), p, 0), who);
        s->bell_host = static_cast<sg::ServiceMailbox *>(p);
    }
    if (!ok || sg::service_launch(bank, s, who) != 0) { sg::service_free(s); return -1; }
    bank->service = s;
    return 0;
}

// 1 when d_out[0..streams) holds this tick's centre outputs, 0 while the windows are filling, -1 on error.  Returns when every
// wave has reported: d_out is complete in device memory (and d_samples may be reused).
int savgol_streambank_service_tick(SavgolStreamBank *bank, const float *d_samples, float *d_out)
{
    const char *who = "savgol_streambank_service_tick";
    if (!bank || !d_samples || !d_out) { sg_set_error("%s: NULL pointer", who); return -1; }
    sg::BankService *s = static_cast<sg::BankService *>(bank->service);
    if (!s) { sg_set_error("%s: call savgol_streambank_service_start first", who); return -1; }
    if ((reinterpret_cast<uintptr_t>(d_samples) | reinterpret_cast<uintptr_t>(d_out)) & 15u) { sg_set_error("%s: d_samples and d_out must be 16-byte aligned", who); return -1; }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!s->running && sg::service_launch(bank, s, who) != 0) return -1;
        sg::ServiceMailbox *mb = s->bell_host;
        const unsigned long long seq = ++s->seq;
        for (unsigned b = 0; b < s->bells; ++b) { mb[b].in = d_samples; mb[b].seq_a = seq; mb[b].out = d_out; mb[b].seq_b = seq | (sg::SERVICE_CMD_TICK << 62); }
        _mm_sfence();                                        // out of the write-combining buffers now (a line that leaves in pieces is
                                                             // caught by the two sequence numbers)
        const auto t0 = std::chrono::steady_clock::now();
        bool exited = false;
        unsigned w = 0;
        unsigned long spins = 0;
        while (w < s->waves) {
            const unsigned d = s->done_host[w];
            if (d == (unsigned)seq) { ++w; continue; }
            if (d == sg::SERVICE_EXITED) { exited = true; break; }
            if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > sg::service_timeout_s()) {
                // Some waves may have taken this tick (accumulators advanced, slot wp of the ring and part of d_out written), the
                // bank's counters have not moved.  Do not leave that kernel resident: stop it -- waves still waiting for this tick
                // first, then those already waiting for the next one; a wave that never became resident leaves on its idle
                // time-out -- and mark the service not running.  The next tick relaunches from the bank's state, which is
                // idempotent: the warm-up reads the 2n newest ring slots, and slot wp is not one of them.
                for (int phase = 0; phase < 2; ++phase) {
                    const unsigned long long sq = seq + (unsigned)phase;
                    for (unsigned b = 0; b < sg::SERVICE_BELLS; ++b) { mb[b].seq_a = sq; mb[b].seq_b = sq | (sg::SERVICE_CMD_STOP << 62); }
                    _mm_sfence();
                    const auto p0 = std::chrono::steady_clock::now();
                    while (phase == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count() < 0.02) {}
                }
                (void)hipStreamSynchronize(s->stream);
                (void)hipGetLastError();
                s->running = false;
                sg_set_error("%s: no answer from the resident kernel for %.1f s (wave %u reports %u, expected %llu); the service kernel was "
                             "stopped and restarts from the bank's state on the next tick", who, sg::service_timeout_s(), w, d, seq);
                return -1;
            }
        }
        if (!exited) {
            const int ws = bank->filter->window_size;
            bank->wp = (bank->wp + 1) % ws;
            bank->received++;
            const int emit = bank->received >= (unsigned long long)ws ? 1 : 0;
            if (emit) bank->emitted++;
            return emit;
        }
        // the kernel left on its idle time-out before it saw this tick (it consumes nothing once a wave has left): wait for the
        // rest of it, then start a fresh one from the bank's state and post the tick again
        if (!sg::hip_ok(hipStreamSynchronize(s->stream), who)) return -1;
        s->running = false;
    }
    sg_set_error("%s: the resident kernel keeps exiting", who);
    return -1;
}

int savgol_streambank_service_stop(SavgolStreamBank *bank)
{
    const char *who = "savgol_streambank_service_stop";
    if (!bank) { sg_set_error("%s: NULL bank", who); return -1; }
    sg::BankService *s = static_cast<sg::BankService *>(bank->service);
    if (!s) return 0;
    int rc = 0;
    if (s->running) {
        for (unsigned b = 0; b < sg::SERVICE_BELLS; ++b) { s->bell_host[b].seq_a = s->seq + 1; s->bell_host[b].seq_b = (s->seq + 1) | (sg::SERVICE_CMD_STOP << 62); }
        _mm_sfence();
        if (!sg::hip_ok(hipStreamSynchronize(s->stream), who)) rc = -1;
        s->running = false;
    }
    sg::service_free(s);
    bank->service = nullptr;
    return rc;
}

int savgol_streambank_service_running(const SavgolStreamBank *bank) { return bank && bank->service ? 1 : 0; }

}  // extern "C"
