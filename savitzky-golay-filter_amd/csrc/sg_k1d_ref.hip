// sg_k1d_ref.hip -- the reference's own 1-D summation order on packed math (the bit-identical batch path).
//
// convolve_ilp (src/savgolFilter.c:547-580) deals the taps of one output to four accumulator chains -- the first
// ws&3 taps to chains 0..2, the rest round robin -- adds each product with its own rounding and returns
// (c0+c1)+(c2+c3).  sg1d_reference_order_kernel (sg_k1d_misc.hip) does exactly that one output per thread; this file
// does it for long batches at a few times that speed: a wave stages a tile of 512 outputs + halo in its own LDS slab
// (boundary remap of the padded modes applied while staging, like sg1d_center_kernel), a lane owns 8 consecutive
// outputs = 4 packed pairs x 4 chains = 16 accumulator pairs, walks its window once (input stationary) and feeds
// every input pair into the chain its tap index selects, with v_pk_mul_f32 + v_pk_add_f32 (two outputs per
// instruction, the reference's two roundings per tap).  Taps sit in SGPR pairs (by-value kernarg).
// The POLYNOMIAL edge samples (2n per channel) are left to sg1d_reference_order_kernel.
#include <cstdint>
#include <cstring>

#include "sg_k1d.hpp"

namespace sg {

template <int N>
struct RefK {
    static constexpr int WS = 2 * N + 1;
    static constexpr int R = 8;                              // outputs per lane
    static constexpr int TW = 64 * R;                        // outputs per tile (one wave)
    static constexpr int NA = (N + 3) & ~3;                  // halo rounded to whole 16-byte vectors
    static constexpr int OFF = NA - N;
    static constexpr int SV = (TW + 2 * NA) / 4;             // vectors in a slab
    static constexpr int WQ = (R + 2 * NA) / 4;              // vectors a lane reads: its 8 outputs' window
    static constexpr int SLAB = 16 * (SV + (SV + 1) / 2);    // bytes: one pad vector after every 2 -> lane stride 48 B
    static constexpr int CH0 = WS & 3;                       // taps that go straight to chains 0 .. CH0-1
};
__device__ __forceinline__ constexpr int ref_vec_off(int v) { return 16 * (v + v / 2); }
__device__ __forceinline__ constexpr int ref_chain(int k, int ch0) { return k < ch0 ? k : ((k - ch0) & 3); }

template <int N>
struct RefConv {
    typedef RefK<N> K;
    // input pair (x[i], x[i+1]) of the lane's window: tap k = i - 2J - OFF of output pair J, chain ref_chain(k).
    // All (up to four) products first, then the adds: an asm result that the next instruction consumes costs an s_nop
    // (sg_pk.hpp), this way it is one per input pair instead of one per product.
    template <int I, int J = 0>
    static __device__ __forceinline__ void products(f32x2 (&P)[4], const f32x2 (&W)[N + 1], const f32x2 x)
    {
        if constexpr (J < 4) {
            constexpr int k = I - 2 * J - K::OFF;
            if constexpr (k >= 0 && k <= 2 * N) P[J] = pk_mul_sgpr<(k & 1)>(W[k >> 1], x);
            products<I, J + 1>(P, W, x);
        }
    }
    template <int I, int J = 0>
    static __device__ __forceinline__ void sums(f32x2 (&A)[4][4], const f32x2 (&P)[4])
    {
        if constexpr (J < 4) {
            constexpr int k = I - 2 * J - K::OFF;
            if constexpr (k >= 0 && k <= 2 * N) A[J][ref_chain(k, K::CH0)] = A[J][ref_chain(k, K::CH0)] + P[J];
            sums<I, J + 1>(A, P);
        }
    }
    template <int I>
    static __device__ __forceinline__ void feed(f32x2 (&A)[4][4], const f32x2 (&W)[N + 1], const f32x2 x)
    {
        f32x2 P[4];
        products<I>(P, W, x);
        sums<I>(A, P);
    }
    template <int Q>
    static __device__ __forceinline__ void quads(const char *win, f32x2 (&A)[4][4], const f32x2 (&W)[N + 1], f32x2 prev)
    {
        if constexpr (Q < K::WQ) {
            const float4 v = *reinterpret_cast<const float4 *>(win + ref_vec_off(Q));
            const f32x2 e0 = {v.x, v.y}, e1 = {v.z, v.w};
            if constexpr (Q > 0) feed<4 * Q - 1>(A, W, pk_straddle(prev, e0));
            feed<4 * Q>(A, W, e0);
            feed<4 * Q + 1>(A, W, pk_straddle(e0, e1));
            feed<4 * Q + 2>(A, W, e1);
            quads<Q + 1>(win, A, W, e1);
        }
    }
};

struct RefJob {
    const float *in;
    float       *out;
    long long    in_ld, out_ld;
    int          length, mode;
    int          store_lo, store_hi, out_shift;
    unsigned     tiles_per_channel, total_tiles;
    unsigned     tpc_magic, tpc_shift;                       // tile / tiles_per_channel on the scalar unit (division_magic)
    float        dt_inv;
    int          vec_in, vec_out;                            // rows 16-byte aligned
};

template <int N>
__global__ __launch_bounds__(256) void sg1d_refpk_kernel(const RefJob job, const Taps taps)
{
    typedef RefK<N> K;
    __shared__ __attribute__((aligned(16))) char smem[4 * K::SLAB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *slab = smem + wave * K::SLAB;
    // one tile per wave, blocks dispatched in order, each XCD sweeping its own eighth of the batch (as sg1d_center_kernel, round 2)
    const unsigned nb8 = gridDim.x >> 3;
    unsigned blk = blockIdx.x;
    if (blk < nb8 * 8u) blk = (blk & 7u) * nb8 + (blk >> 3);
    const unsigned tile = blk * 4u + (unsigned)wave;
    if (tile >= job.total_tiles) return;
    const int L = job.length;

    f32x2 W[N + 1];
#pragma unroll
    for (int p = 0; p < N + 1; ++p) W[p] = f32x2{taps.w[2 * p], taps.w[2 * p + 1]};

    {
        const unsigned c = job.tpc_shift >= 32 ? tile : (__umulhi(tile, job.tpc_magic) >> job.tpc_shift);
        const int t0 = (int)(tile - c * job.tiles_per_channel) * K::TW;          // first output of the tile
        const float *x = job.in + (long long)c * job.in_ld;
        float *y = job.out + (long long)c * job.out_ld;

        // ---- stage samples t0-NA .. t0+TW+NA-1 (remapped at the channel ends) ----
        const int g0 = t0 - K::NA;
        if (job.vec_in && g0 >= 0 && g0 + K::SV * 4 <= L) {
#pragma unroll
            for (int v = lane; v < K::SV; v += 64)
                *reinterpret_cast<float4 *>(slab + ref_vec_off(v)) = *reinterpret_cast<const float4 *>(x + g0 + 4 * v);
        } else {
            for (int s = lane; s < K::SV * 4; s += 64) {
                int g = g0 + s;
                bool zero = false;
                if (g < 0 || g >= L) g = remap_index(g, L, job.mode, zero);
                reinterpret_cast<float *>(slab + ref_vec_off(s >> 2))[s & 3] = zero ? 0.0f : x[g];
            }
        }
        wave_lds_sync();

        // ---- the four chains of every output of this lane ----
        f32x2 A[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) A[j][q] = f32x2{0.0f, 0.0f};
        RefConv<N>::template quads<0>(slab + ref_vec_off(2 * lane), A, W, f32x2{0.0f, 0.0f});
        wave_lds_sync();                                     // the slab is free for the next tile

        // ---- (c0+c1)+(c2+c3), * dt_inv, store ----
        const f32x2 s2 = f32x2{job.dt_inv, job.dt_inv};
        f32x2 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ((A[j][0] + A[j][1]) + (A[j][2] + A[j][3])) * s2;
        const int j0 = t0 + 8 * lane;                        // this lane's first output
        if (job.vec_out && j0 >= job.store_lo && j0 + 8 <= job.store_hi) {
            float *dst = y + (j0 - job.out_shift);
            *reinterpret_cast<float4 *>(dst) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e;
                const float v = (e & 1) ? o[e >> 1].y : o[e >> 1].x;
                if (j >= job.store_lo && j < job.store_hi) y[j - job.out_shift] = v;
            }
        }
    }
}

template <int N>
static int launch_refpk(const RefJob &job, const Taps &taps, int cu_count, hipStream_t st)
{
    unsigned blocks = (job.total_tiles + 3u) / 4u;             // one tile per wave
    blocks = (blocks + 7u) & ~7u;
    (void)cu_count;
    hipLaunchKernelGGL((sg1d_refpk_kernel<N>), dim3(blocks), dim3(256), 0, st, job, taps);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int N>
static int dispatch_refpk(int n, const RefJob &job, const Taps &taps, int cu_count, hipStream_t st)
{
    if (n == N) return launch_refpk<N>(job, taps, cu_count, st);
    if constexpr (N < SAVGOL_MAX_HALF_WINDOW) return dispatch_refpk<N + 1>(n, job, taps, cu_count, st);
    else return -1;
}

}  // namespace sg

// Samples [store_lo, store_hi) of every channel in the reference's order; rows of `length` samples, `mode` decides
// what lies beyond the channel ends (POLYNOMIAL callers store only [n, L-n) here).  0 on success.
extern "C" int sg1d_launch_refpk_f32(const float *in, float *out, long long in_ld, long long out_ld, long long length, int n,
                                     const float *center, float dt_inv, int mode, int store_lo, int store_hi, int out_shift,
                                     size_t channels, int cu_count, void *stream)
{
    using namespace sg;
    Taps taps;
    memset(&taps, 0, sizeof(taps));
    memcpy(taps.w, center, sizeof(float) * (size_t)(2 * n + 1));
    RefJob job;
    memset(&job, 0, sizeof(job));
    job.in_ld = in_ld; job.out_ld = out_ld; job.length = (int)length; job.mode = mode;
    job.store_lo = store_lo; job.store_hi = store_hi; job.out_shift = out_shift; job.dt_inv = dt_inv;
    job.tiles_per_channel = (unsigned)((length + 511) / 512);
    division_magic(job.tiles_per_channel, &job.tpc_magic, &job.tpc_shift);
    job.vec_in = (in_ld % 4 == 0);
    job.vec_out = (out_ld % 4 == 0) && (out_shift % 4 == 0);
    // one tile per wave, four waves per block: a launch holds < 2^24 blocks (HIP rejects gridDim.x * blockDim.x >= 2^32)
    size_t max_ch = (size_t)sg::MAX_TILES_PER_LAUNCH / job.tiles_per_channel;
    if (max_ch == 0) max_ch = 1;                     // cannot happen: a channel of 2^30 samples is 2^21 tiles
    for (size_t c0 = 0; c0 < channels; c0 += max_ch) {
        const size_t nc = channels - c0 < max_ch ? channels - c0 : max_ch;
        job.in = in + c0 * in_ld; job.out = out + c0 * out_ld;
        job.total_tiles = (unsigned)(nc * job.tiles_per_channel);
        const int vi = job.vec_in && (reinterpret_cast<uintptr_t>(job.in) % 16 == 0);
        const int vo = job.vec_out && (reinterpret_cast<uintptr_t>(job.out) % 16 == 0);
        RefJob j2 = job; j2.vec_in = vi; j2.vec_out = vo;
        if (dispatch_refpk<1>(n, j2, taps, cu_count, static_cast<hipStream_t>(stream)) != 0) return -1;
    }
    return 0;
}
