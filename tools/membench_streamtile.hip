// tools/membench_streamtile.hip -- round 5: bare access patterns for the stream block push (BASELINE config 3: [ticks][streams] fp32,
// 4096 x 65 536, a 2n = 32-row halo down the tick axis), before any arithmetic is added.
//   reg   : the shipped tile form (csrc/sg_stream_roll.hip sg_bank_tile_kernel): a wave loads TR + 32 rows of SPL*64 streams into registers
//   lds   : the same tile staged by LDS-DMA (global_load_lds_dwordx4, no VGPRs in flight) into a wave-private slab, rows read back from LDS
//   ldsblk: a BLOCK owns 256 streams x TR ticks; its waves split the row loads, one barrier, then split the streams (64 per wave)
// every form: out[t] = in[t] + 1e-30 * (in[t - 16] + in[t + 16]) so that every loaded row is used.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench_streamtile tools/membench_streamtile.hip && tools/membench_streamtile
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int kHalo = 32;            // 2n at n = 16
constexpr int kH2 = kHalo / 2;

__global__ __launch_bounds__(256) void k_init(v4f *p, size_t nvec)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x5A17601Aull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
        p[i] = v4f{(float)(z & 0xffff) * 1e-4f, (float)((z >> 16) & 0xffff) * 1e-4f, (float)((z >> 32) & 0xffff) * 1e-4f, (float)(z >> 48) * 1e-4f};
    }
}

__global__ __launch_bounds__(256) void k_flat(const v4f *__restrict__ in, v4f *__restrict__ out, size_t nvec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) __builtin_nontemporal_store(__builtin_bit_cast(u4, in[i]), (u4 *)(out + i));
}

struct Shape {
    int streams, ticks;
    unsigned strips, bands, group;
    unsigned long long total;
    int xcd, fma;                    // fma: dummy packed multiply-adds per output vector (0 = bare)
};

// tile index -> (strip, band): groups of `group` neighbouring strips, inside a group band after band, strips fastest
__device__ __forceinline__ bool tile_of(const Shape &s, unsigned long long t, unsigned &strip, unsigned &band)
{
    const unsigned long long per_group = (unsigned long long)s.group * s.bands;
    const unsigned grp = (unsigned)(t / per_group);
    const unsigned long long rem = t % per_group;
    const unsigned left = s.strips - grp * s.group;
    const unsigned gs = left < s.group ? left : s.group;
    band = (unsigned)(rem / gs); strip = grp * s.group + (unsigned)(rem % gs);
    return band < s.bands;
}

__device__ __forceinline__ v4f busy(v4f v, int n)
{
    // n dependent-free packed multiply-adds on the value (two chains per half), standing in for the taps
    v2f a = {v.x, v.y}, b = {v.z, v.w};
    const v2f w = {1.0000001f, 0.9999999f};
    for (int i = 0; i < n; i += 2) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(w), "v"(b));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b) : "v"(w), "v"(a));
    }
    return v4f{a.x, a.y, b.x, b.y};
}

// ---- reg: rows in registers (SPL = 4: 16 B per lane and row, 256 streams per wave; SPL = 2: 8 B, 128 streams) ----
template <int TR, int SPL, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_reg(const float *__restrict__ in, float *__restrict__ out, const Shape s)
{
    extern __shared__ float occupancy_pad[];
    typedef typename std::conditional<SPL == 4, v4f, v2f>::type VT;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = s.xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned long long t = (unsigned long long)blk * WPB + wv;
    if (t >= s.total) return;
    unsigned strip, band;
    if (!tile_of(s, t, strip, band)) return;
    const int t0 = (int)band * TR;
    const size_t col = (size_t)strip * (64 * SPL) + (size_t)lane * SPL;
    VT tile[TR + kHalo];
#pragma unroll
    for (int i = 0; i < TR + kHalo; ++i) {
        int y = t0 - kH2 + i;
        y = y < 0 ? 0 : (y >= s.ticks ? s.ticks - 1 : y);
        tile[i] = *(const VT *)(in + (size_t)y * s.streams + col);
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        VT v = tile[r + kH2] + 1e-30f * (tile[r] + tile[r + kHalo]);
        if (s.fma) {
            if constexpr (SPL == 4) v = busy(v, s.fma);
            else { v4f q = busy(v4f{v.x, v.y, v.x, v.y}, s.fma / 2); v = VT{q.x, q.y}; }
        }
        const int y = t0 + r;
        if (y < s.ticks) {
            if constexpr (SPL == 4) __builtin_nontemporal_store(__builtin_bit_cast(u4, v), (u4 *)(out + (size_t)y * s.streams + col));
            else __builtin_nontemporal_store(__builtin_bit_cast(u2, v), (u2 *)(out + (size_t)y * s.streams + col));
        }
    }
    (void)occupancy_pad;
}

// ---- lds: a wave-private slab filled by LDS-DMA ----
// WS streams per wave (64 / 128 / 256: a row is 256 / 512 / 1024 bytes, one DMA instruction moves 4 / 2 / 1 rows)
template <int TR, int WS, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_lds(const float *__restrict__ in, float *__restrict__ out, const Shape s)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RB = WS * 4, RPI = 1024 / RB, ROWS = TR + kHalo, NI = ROWS / RPI, SLAB = ROWS * RB;
    static_assert(ROWS % RPI == 0 && TR % RPI == 0 && kH2 % RPI == 0, "whole instructions");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = s.xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned long long t = (unsigned long long)blk * WPB + wv;
    if (t >= s.total) return;
    unsigned strip, band;
    if (!tile_of(s, t, strip, band)) return;
    const int t0 = (int)band * TR;
    const int rsub = lane / (RB / 16), chunk = lane % (RB / 16);          // row inside one instruction's rows, 16-byte chunk of the row
    const size_t col = (size_t)strip * WS + (size_t)chunk * 4;
    char *slab = lds + wv * SLAB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int y = t0 - kH2 + i * RPI + rsub;
        y = y < 0 ? 0 : (y >= s.ticks ? s.ticks - 1 : y);
        const float *g = in + (size_t)y * s.streams + col;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(slab + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char *mine = slab + lane * 16;
#pragma unroll
    for (int j = 0; j < TR / RPI; ++j) {
        const v4f c = *(const v4f *)(mine + (kH2 / RPI + j) * 1024);
        const v4f a = *(const v4f *)(mine + j * 1024);
        const v4f b = *(const v4f *)(mine + (kHalo / RPI + j) * 1024);
        v4f v = c + 1e-30f * (a + b);
        if (s.fma) v = busy(v, s.fma);
        const int y = t0 + j * RPI + rsub;
        if (y < s.ticks) __builtin_nontemporal_store(__builtin_bit_cast(u4, v), (u4 *)(out + (size_t)y * s.streams + col));
    }
}

// ---- ldsblk: a block of WPB waves owns 256 streams x TR ticks in ONE slab; waves split the row loads, then the output rows ----
template <int TR, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_ldsblk(const float *__restrict__ in, float *__restrict__ out, const Shape s)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ROWS = TR + kHalo;
    static_assert(ROWS % WPB == 0 && TR % WPB == 0, "rows split over the waves");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = s.xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned long long t = blk;
    if (t >= s.total) return;
    unsigned strip, band;
    if (!tile_of(s, t, strip, band)) return;
    const int t0 = (int)band * TR;
    const size_t col = (size_t)strip * 256 + (size_t)lane * 4;
#pragma unroll
    for (int i = 0; i < ROWS / WPB; ++i) {
        const int r = i * WPB + wv;
        int y = t0 - kH2 + r;
        y = y < 0 ? 0 : (y >= s.ticks ? s.ticks - 1 : y);
        const float *g = in + (size_t)y * s.streams + col;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(lds + r * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char *mine = lds + lane * 16;
#pragma unroll
    for (int j = 0; j < TR / WPB; ++j) {
        const int r = j * WPB + wv;
        const v4f c = *(const v4f *)(mine + (kH2 + r) * 1024);
        const v4f a = *(const v4f *)(mine + r * 1024);
        const v4f b = *(const v4f *)(mine + (kHalo + r) * 1024);
        v4f v = c + 1e-30f * (a + b);
        if (s.fma) v = busy(v, s.fma);
        const int y = t0 + r;
        if (y < s.ticks) __builtin_nontemporal_store(__builtin_bit_cast(u4, v), (u4 *)(out + (size_t)y * s.streams + col));
    }
}


// ---- walkdma: a wave walks down a band of ticks of its 128 streams; the rows arrive by LDS-DMA into a wave-private RING of P rows (prefetch
// distance P rows, no VGPR in flight), each row is read from LDS once and stored; no halo re-reads except the 2n warm-up rows of a band ----
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int K> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory"); }

template <int P, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_walkdma(const float *__restrict__ in, float *__restrict__ out, const Shape s, int band_ticks)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int HP = P / 2;                                    // row pairs in the ring
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = s.xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned item = blk * WPB + wv;
    if (item >= s.total) return;
    const unsigned strip = item % s.strips, band = item / s.strips;
    const int t0 = (int)band * band_ticks;
    int nt = s.ticks - t0 < band_ticks ? s.ticks - t0 : band_ticks;
    const int rows = nt + kHalo, pairs = (rows + 1) / 2;         // warm-up rows included
    const unsigned ring = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)lds) + (unsigned)wv * (unsigned)(P * 512);
    const char *mine = lds + (size_t)wv * (P * 512) + lane * 8;
    const int sub = lane >> 5, chunk = lane & 31;
    const size_t col = (size_t)strip * 128 + (size_t)chunk * 4;
    auto src = [&](int pair) -> const float * {
        int y = t0 - kHalo + 2 * pair + sub;
        y = y < 0 ? 0 : (y >= s.ticks ? s.ticks - 1 : y);
        return in + (size_t)y * s.streams + col;
    };
    const size_t ocol = (size_t)strip * 128 + (size_t)lane * 2;
    auto step = [&](int g, v2f &carry) {
        const int slot = g % HP;
        const v2f a = *(const v2f *)(mine + slot * 1024), b = *(const v2f *)(mine + slot * 1024 + 512);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma16(src(g + HP), ring + (unsigned)slot * 1024u);       // refill the slot just read (rows past the band: clamped, never used)
        const int y = t0 - kHalo + 2 * g;
        v2f o0 = a + 1e-30f * carry, o1 = b + 1e-30f * a;
        carry = b;
        if (s.fma) { v4f q = busy(v4f{o0.x, o0.y, o1.x, o1.y}, s.fma); o0 = v2f{q.x, q.y}; o1 = v2f{q.z, q.w}; }
        // one store per row, always (rows without an output go to an empty descriptor so that the vmcnt arithmetic stays static)
        const bool ok0 = y >= t0 && y < t0 + nt, ok1 = y + 1 >= t0 && y + 1 < t0 + nt;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(ok0 ? y : 0) * s.streams, 0, ok0 ? s.streams * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(ok1 ? y + 1 : 0) * s.streams, 0, ok1 ? s.streams * 4 : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, o0), r0, (int)(ocol * 4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, o1), r1, (int)(ocol * 4), 0, 0);
    };
#pragma unroll
    for (int g = 0; g < HP; ++g) dma16(src(g), ring + (unsigned)g * 1024u);
    v2f carry = {0.0f, 0.0f};
    // first HP pairs: the queue fills up with this loop's own DMAs and stores (literal counts), then the steady state 3 HP - 1
    [&]<int... G>(std::integer_sequence<int, G...>) {
        ((wait_vm<HP - 1 + 2 * G>(), step(G, carry)), ...);
    }(std::make_integer_sequence<int, HP>{});
    for (int g = HP; g < pairs; ++g) {
        wait_vm<3 * HP - 1>();
        step(g, carry);
    }
}


// ---- dmatile: the product's LDS-DMA tile (csrc/sg_stream_dma.hip) with phase stamps: 128 streams x TR ticks per wave, all row pairs issued up
// front, consumed in arrival order behind counted vmcnt waits, input-stationary multiply-adds (MODE 1: 33 taps, two chains; MODE 0: none) ----
struct Taps33 { v2f w[17]; };
template <int SEL> __device__ __forceinline__ void pkfma(v2f &acc, const v2f wpair, const v2f x)
{
    if constexpr (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
    else                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
}
template <int SEL> __device__ __forceinline__ v2f pkmul(const v2f wpair, const v2f x)
{
    v2f p;
    if constexpr (SEL == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(wpair), "v"(x));
    else                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(wpair), "v"(x));
    return p;
}
__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
template <int TR, int MODE, int WPB, bool STAMP>
__global__ __launch_bounds__(64 * WPB) void k_dmatile(const float *__restrict__ in, float *__restrict__ out, const Shape s, const Taps33 taps, unsigned long long *stamps)
{
    constexpr int N = 16, ROWS = TR + 2 * N, NI = ROWS / 2, RB = 512, SLAB = ROWS * RB;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = gridDim.x;
    const unsigned blk = s.xcd ? (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned t = blk * WPB + wv;
    if (t >= s.total) return;
    unsigned strip, band;
    if (!tile_of(s, t, strip, band)) return;
    unsigned long long st[8];
    if constexpr (STAMP) st[0] = now();
    const int t0 = (int)band * TR;
    const unsigned slab = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)lds) + (unsigned)wv * (unsigned)SLAB;
    const int sub = lane >> 5, chunk = lane & 31;
    const size_t col = (size_t)strip * 128 + (size_t)chunk * 4;
    {
        int y0 = t0 - 2 * N + sub;
        const bool inside = t0 >= 2 * N && t0 + TR <= s.ticks;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int y = y0 + 2 * i;
            if (!inside) y = y < 0 ? 0 : (y >= s.ticks ? s.ticks - 1 : y);
            dma16(in + (size_t)y * s.streams + col, slab + (unsigned)i * 1024u);
        }
    }
    if constexpr (STAMP) st[1] = now();
    const char *mine = lds + (size_t)wv * SLAB + lane * 8;
    const unsigned voff = (strip * 128u + 2u * (unsigned)lane) * 4u;
    v2f acc[2][TR];
    auto row_in = [&](auto rc) -> v2f { constexpr int r = decltype(rc)::value; return *reinterpret_cast<const v2f *>(mine + r * RB); };
    auto feed = [&](auto rc, const v2f x) {
        constexpr int r = decltype(rc)::value;
        constexpr int mlo = r - 2 * N > 0 ? r - 2 * N : 0, mhi = r < TR - 1 ? r : TR - 1;
        if constexpr (MODE == 1) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ([&] {
                    constexpr int m = mlo + I, k = r - m;
                    if constexpr (k < 2) acc[k][m] = pkmul<k>(taps.w[0], x);
                    else pkfma<(k & 1)>(acc[k & 1][m], taps.w[k >> 1], x);
                }(), ...);
            }(std::make_integer_sequence<int, mhi - mlo + 1>{});
        } else {
            if constexpr (r >= 2 * N && r - 2 * N < TR) acc[0][r - 2 * N] = x;
            else if constexpr (r < TR) acc[1][r] = x;
        }
        if constexpr (r >= 2 * N && r - 2 * N < TR) {
            constexpr int m = r - 2 * N;
            v2f a = acc[0][m] + (MODE == 1 ? acc[1][m] : 1e-30f * acc[1][m]);
            const int tt = t0 + m;
            const bool has_out = tt < s.ticks;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(has_out ? tt : 0) * s.streams, 0, has_out ? s.streams * 4 : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, a), rs, (int)voff, 0, 0);
        }
    };
    wait_vm<NI - 1>();
    if constexpr (STAMP) st[2] = now();
    v2f xa = row_in(std::integral_constant<int, 0>{}), xb = row_in(std::integral_constant<int, 1>{});
    [&]<int... G>(std::integer_sequence<int, G...>) {
        ([&] {
            constexpr int g = G;
            v2f na = xa, nb = xb;
            if constexpr (g + 1 < NI) {
                constexpr int done = 2 * g - 2 * N < 0 ? 0 : (2 * g - 2 * N > TR ? TR : 2 * g - 2 * N);
                wait_vm<NI - 2 - g + done>();
                if constexpr (STAMP) { if constexpr (g + 1 == NI / 4) st[3] = now(); if constexpr (g + 1 == NI / 2) st[4] = now(); if constexpr (g + 1 == 3 * NI / 4) st[5] = now(); if constexpr (g + 2 == NI) st[6] = now(); }
                na = row_in(std::integral_constant<int, 2 * g + 2>{});
                nb = row_in(std::integral_constant<int, 2 * g + 3>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            feed(std::integral_constant<int, 2 * g>{}, xa);
            feed(std::integral_constant<int, 2 * g + 1>{}, xb);
            xa = na; xb = nb;
        }(), ...);
    }(std::make_integer_sequence<int, NI>{});
    if constexpr (STAMP) {
        st[7] = now();
        if (lane == 0 && t < 65536) { for (int i = 0; i < 8; ++i) stamps[(size_t)t * 8 + i] = st[i]; }
    }
}

static hipEvent_t ev_a, ev_b;
template <typename F>
static double time_ms(F launch, int iters = 9)
{
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(ev_a)); launch(); CK(hipEventRecord(ev_b)); CK(hipEventSynchronize(ev_b));
        float t; CK(hipEventElapsedTime(&t, ev_a, ev_b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

static int kStreams = 65536, kTicks = 4096;
static float *g_in, *g_out;

static void report(const char *name, int tr, int w, int wpb, const Shape &s, int cap, int vgpr, unsigned grid, size_t lds, double ms)
{
    const double bytes = 2.0 * (double)kStreams * kTicks * 4;
    printf("%-6s TR=%3d W=%3d WPB=%d group=%3u xcd=%d cap=%d fma=%3d vgpr=%3d lds=%6zu grid=%7u : %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", name, tr, w, wpb, s.group,
           s.xcd, cap, s.fma, vgpr, lds, grid, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

static Shape shape(int tr, int w, int group, int xcd, int fma)
{
    Shape s;
    s.streams = kStreams; s.ticks = kTicks;
    s.strips = (unsigned)(kStreams / w);
    s.bands = (unsigned)((kTicks + tr - 1) / tr);
    s.group = (unsigned)group > s.strips ? s.strips : (unsigned)group;
    const unsigned groups = (s.strips + s.group - 1) / s.group;
    s.total = (unsigned long long)groups * s.group * s.bands;
    s.xcd = xcd; s.fma = fma;
    return s;
}

template <int TR, int SPL, int WPB>
static void run_reg(int group, int xcd, int cap, int fma = 0)
{
    const Shape s = shape(TR, 64 * SPL, group, xcd, fma);
    const void *fn = reinterpret_cast<const void *>(k_reg<TR, SPL, WPB>);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, fn));
    const size_t lds = cap > 0 ? (size_t)(160 * 1024 / cap) - 1024 : 0;
    if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (unsigned)((s.total + WPB - 1) / WPB);
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_reg<TR, SPL, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s); });
    report("reg", TR, 64 * SPL, WPB, s, cap, fa.numRegs, grid, lds, ms);
}

template <int TR, int WS, int WPB>
static void run_lds(int group, int xcd, int fma = 0)
{
    const Shape s = shape(TR, WS, group, xcd, fma);
    const void *fn = reinterpret_cast<const void *>(k_lds<TR, WS, WPB>);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, fn));
    const size_t lds = (size_t)WPB * (TR + kHalo) * WS * 4;
    if (lds > 160 * 1024) { printf("lds    TR=%d W=%d WPB=%d: %zu bytes of LDS, skipped\n", TR, WS, WPB, lds); return; }
    if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (unsigned)((s.total + WPB - 1) / WPB);
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_lds<TR, WS, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s); });
    report("lds", TR, WS, WPB, s, 0, fa.numRegs, grid, lds, ms);
}

template <int TR, int WPB>
static void run_ldsblk(int group, int xcd, int fma = 0)
{
    const Shape s = shape(TR, 256, group, xcd, fma);
    const void *fn = reinterpret_cast<const void *>(k_ldsblk<TR, WPB>);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, fn));
    const size_t lds = (size_t)(TR + kHalo) * 1024;
    if (lds > 160 * 1024) { printf("ldsblk TR=%d WPB=%d: %zu bytes of LDS, skipped\n", TR, WPB, lds); return; }
    if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (unsigned)s.total;
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_ldsblk<TR, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s); });
    report("ldsblk", TR, 256, WPB, s, 0, fa.numRegs, grid, lds, ms);
}


template <int P, int WPB>
static void run_walkdma(int bands, int xcd, int fma = 0)
{
    Shape s = shape(16, 128, 512, xcd, fma);
    const int band_ticks = (kTicks + bands - 1) / bands;
    s.bands = (unsigned)bands; s.total = (unsigned long long)s.strips * bands;
    const void *fn = reinterpret_cast<const void *>(k_walkdma<P, WPB>);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, fn));
    const size_t lds = (size_t)WPB * P * 512;
    if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (unsigned)((s.total + WPB - 1) / WPB);
    grid = (grid + 7u) & ~7u;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_walkdma<P, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s, band_ticks); });
    char name[32]; snprintf(name, sizeof name, "wdma%d", P);
    report(name, band_ticks, 128, WPB, s, bands, fa.numRegs, grid, lds, ms);
}


static unsigned long long *g_stamps;
template <int TR, int MODE, int WPB>
static void run_dmatile(int group, int xcd)
{
    const Shape s = shape(TR, 128, group, xcd, MODE);
    Taps33 taps;
    for (int k = 0; k < 34; ++k) { const float w = 0.03f + 0.001f * (float)k; if (k & 1) taps.w[k >> 1].y = w; else taps.w[k >> 1].x = w; }
    const size_t lds = (size_t)WPB * (TR + 32) * 512;
    unsigned grid = (unsigned)((s.total + WPB - 1) / WPB);
    grid = (grid + 7u) & ~7u;
    {
        const void *fn = reinterpret_cast<const void *>(k_dmatile<TR, MODE, WPB, false>);
        hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, fn));
        if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_dmatile<TR, MODE, WPB, false>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s, taps, g_stamps); });
        report(MODE ? "dmaFMA" : "dmaBARE", TR, 128, WPB, s, 0, fa.numRegs, grid, lds, ms);
    }
    {
        const void *fn = reinterpret_cast<const void *>(k_dmatile<TR, MODE, WPB, true>);
        if (lds > 65536) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipMemset(g_stamps, 0, 65536 * 8 * 8));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_dmatile<TR, MODE, WPB, true>), dim3(grid), dim3(64 * WPB), lds, 0, g_in, g_out, s, taps, g_stamps);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(65536 * 8);
        CK(hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost));
        // medians of the phase lengths over the middle half of the tiles (cycles of s_memtime = 100 MHz?  no: shader clock ticks)
        const char *names[7] = {"issue", "first pair", "to 1/4", "to 1/2", "to 3/4", "to last", "tail"};
        printf("        stamps (median ticks over tiles %u..%u):", (unsigned)(s.total / 4), (unsigned)(3 * s.total / 4));
        for (int ph = 0; ph < 7; ++ph) {
            std::vector<long long> d;
            for (size_t t = s.total / 4; t < 3 * s.total / 4 && t < 65536; ++t) if (h[t * 8 + 7]) d.push_back((long long)(h[t * 8 + ph + 1] - h[t * 8 + ph]));
            if (d.empty()) { printf(" -"); continue; }
            std::sort(d.begin(), d.end());
            printf("  %s %lld", names[ph], d[d.size() / 2]);
        }
        std::vector<long long> life;
        for (size_t t = s.total / 4; t < 3 * s.total / 4 && t < 65536; ++t) if (h[t * 8 + 7]) life.push_back((long long)(h[t * 8 + 7] - h[t * 8]));
        std::sort(life.begin(), life.end());
        if (!life.empty()) printf("  | life %lld", life[life.size() / 2]);
        printf("\n"); fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    if (argc > 1) kTicks = atoi(argv[1]);
    const int fma = argc > 2 ? atoi(argv[2]) : 0;
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t bytes = (size_t)kStreams * kTicks * 4;
    CK(hipMalloc(&g_in, bytes)); CK(hipMalloc(&g_out, bytes));
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_in, bytes / 16);
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, (v4f *)g_out, bytes / 16);
    CK(hipDeviceSynchronize());
    {
        const size_t nvec = bytes / 16;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const v4f *)g_in, (v4f *)g_out, nvec); });
        printf("flat copy, one vector per thread: %7.3f ms  %6.1f GB/s  (%.3f of 8 TB/s)\n", ms, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 8000.0);
    }
    const int only = argc > 3 ? atoi(argv[3]) : 0;
    CK(hipMalloc(&g_stamps, 65536 * 8 * 8));
    if (only == 2) {       // the product's DMA tile with phase stamps, bare and with the 33-tap multiply-adds
        for (int group : {64, 128}) {
            run_dmatile<32, 0, 4>(group, 1); run_dmatile<32, 1, 4>(group, 1);
            run_dmatile<32, 0, 2>(group, 1); run_dmatile<32, 1, 2>(group, 1);
            run_dmatile<32, 0, 1>(group, 1); run_dmatile<32, 1, 1>(group, 1);
            run_dmatile<16, 0, 2>(group, 1); run_dmatile<16, 1, 2>(group, 1);
            run_dmatile<16, 0, 3>(group, 1); run_dmatile<16, 1, 3>(group, 1);
            run_dmatile<48, 0, 1>(group, 1); run_dmatile<48, 1, 1>(group, 1);
        }
        return 0;
    }
    // the DMA ring walk: ring depth x waves per block x bands (items = 512 strips x bands; 2560 waves fit at 10 per CU)
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int bands : {2, 4, 5, 8, 16, 64}) {
            run_walkdma<16, 4>(bands, xcd, fma);
            run_walkdma<32, 4>(bands, xcd, fma);
            run_walkdma<32, 2>(bands, xcd, fma);
            run_walkdma<32, 1>(bands, xcd, fma);
            run_walkdma<40, 4>(bands, xcd, fma);
            run_walkdma<24, 4>(bands, xcd, fma);
        }
    if (only == 1) return 0;
    // the shipped tile shape, bare, at the occupancy its registers allow and capped
    for (int cap = 0; cap <= 2; ++cap) {
        run_reg<16, 4, 2>(64, 1, cap, fma);
        run_reg<16, 4, 4>(64, 1, cap, fma);
        run_reg<32, 2, 4>(128, 1, cap, fma);
        run_reg<48, 2, 4>(128, 1, cap, fma);
        run_reg<64, 2, 4>(128, 1, cap, fma);
    }
    // LDS-DMA, wave-private slabs
    for (int group : {16, 64, 256}) {
        run_lds<32, 256, 2>(group / 1 > 0 ? group : 1, 1, fma);      // 64 KiB per wave
        run_lds<48, 128, 4>(group * 2, 1, fma);                       // 40 KiB per wave, 4 waves per CU
        run_lds<32, 128, 4>(group * 2, 1, fma);                       // 32 KiB per wave
        run_lds<32, 128, 2>(group * 2, 1, fma);
        run_lds<128, 64, 4>(group * 4, 1, fma);                       // 40 KiB per wave
        run_lds<96, 64, 4>(group * 4, 1, fma);                        // 32 KiB per wave
        run_lds<96, 64, 1>(group * 4, 1, fma);
        run_lds<64, 64, 4>(group * 4, 1, fma);                        // 24 KiB
        run_lds<32, 64, 4>(group * 4, 1, fma);                        // 16 KiB: 10 waves per CU
        run_lds<32, 64, 8>(group * 4, 1, fma);
    }
    // LDS-DMA, block-shared slab (256 streams)
    for (int group : {16, 64, 256}) {
        run_ldsblk<128, 4>(group, 1, fma);
        run_ldsblk<96, 4>(group, 1, fma);
        run_ldsblk<96, 8>(group, 1, fma);
        run_ldsblk<48, 4>(group, 1, fma);                            // 80 KiB: two blocks per CU
        run_ldsblk<48, 8>(group, 1, fma);
        run_ldsblk<32, 4>(group, 1, fma);                            // 64 KiB
        run_ldsblk<16, 4>(group, 1, fma);                            // 48 KiB: three blocks
    }
    return 0;
}
