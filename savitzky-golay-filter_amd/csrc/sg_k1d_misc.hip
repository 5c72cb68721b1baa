// sg_k1d_misc.hip -- the small 1-D kernels around sg1d_center_kernel: POLYNOMIAL edge rows,
// array-of-structs gather/scatter for the strided entry point, and the synthetic-signal generator
// used by bench.py and the full-size tests.
#include "sg_k1d.hpp"

namespace sg {

// element i of channel c is the float at base + c*pitch + i*stride + offset (bytes); any alignment
__global__ __launch_bounds__(256) void sg_gather_f32_kernel(const char *__restrict__ base, size_t stride, size_t offset,
                                                            size_t pitch, float *__restrict__ dst, size_t dst_ld,
                                                            size_t count)
{
    const size_t c = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float v;
        __builtin_memcpy(&v, base + c * pitch + i * stride + offset, sizeof(float));
        dst[c * dst_ld + i] = v;
    }
}

// the POLYNOMIAL edge rows of sg1d_edges_kernel (sg_k1d.hpp) on strided data (always the edges of a strided call: the reference's
// savgol_apply_strided ignores config.boundary, :902-928)
__global__ __launch_bounds__(64) void sg1d_edges_strided_kernel(const char *__restrict__ in, char *__restrict__ out, long long in_pitch,
                                                                long long out_pitch, long long in_stride, long long out_stride,
                                                                long long L, int n, const float *__restrict__ ew, float dt_inv, int flags)
{
    const int lane = threadIdx.x;
    const long long c = blockIdx.x;
    const bool trailing = blockIdx.y != 0;
    const int ws = 2 * n + 1;
    const char *row = in + c * in_pitch;
    char *orow = out + c * out_pitch;
    auto sample = [&](long long i) { return *reinterpret_cast<const float *>(row + i * in_stride); };
    const int k0 = lane, k1 = lane + 64;
    float x0 = 0.0f, x1 = 0.0f;
    if (k0 < ws) x0 = sample(trailing ? (L - ws + k0) : (long long)(2 * n - k0));
    if (k1 < ws) x1 = sample(trailing ? (L - ws + k1) : (long long)(2 * n - k1));
    for (int e = 0; e < n; ++e) {
        const float *w = ew + e * ws;
        float p = 0.0f;
        if (k0 < ws) p = w[k0] * x0;
        if (k1 < ws) p = __builtin_fmaf(w[k1], x1, p);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) p += __shfl_xor(p, m, 64);
        if (flags & 1) p *= dt_inv;
        if ((flags & 2) && !trailing) p = -p;
        if (lane == 0) *reinterpret_cast<float *>(orow + (trailing ? (L - 1 - e) : (long long)e) * out_stride) = p;
    }
}

__global__ __launch_bounds__(256) void sg_scatter_f32_kernel(const float *__restrict__ src, size_t src_ld,
                                                             char *__restrict__ base, size_t stride, size_t offset,
                                                             size_t pitch, size_t count)
{
    const size_t c = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const float v = src[c * src_ld + i];
        __builtin_memcpy(base + c * pitch + i * stride + offset, &v, sizeof(float));
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// x[c][i] = sin(2 pi f_c i) + 0.5 sin(2 pi 7.3 f_c i + phi_c) + 0.1 u(c,i), evaluated in
// fp64 and rounded once to T.  Counter based: every value depends only on (seed, c, i).
template <typename T>
__global__ __launch_bounds__(256) void sg_synth_kernel(T *__restrict__ dst, size_t channel0, size_t length, size_t ld,
                                                       uint64_t seed)
{
    const size_t c = blockIdx.y;
    const uint64_t cg = channel0 + c;
    const double two_pi = 6.283185307179586476925286766559;
    const double f = (double)(1 + (cg % 97)) / 4096.0;
    const double phi = two_pi * (double)(cg % 13) / 13.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (size_t)gridDim.x * blockDim.x) {
        const double u = (double)(mix64(seed ^ (cg << 32) ^ (uint64_t)i) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        const double t = (double)i;
        dst[c * ld + i] = (T)(sin(two_pi * f * t) + 0.5 * sin(two_pi * 7.3 * f * t + phi) + 0.1 * u);
    }
}

// ---- the reference's own summation order (savgol_apply, src/savgolFilter.c:743-804), one output per thread ----
// convolve_ilp (:547-580) / convolve_ilp_reverse (:593-623): the first ws&3 taps go to chains 0..2, the rest are dealt
// round robin to four chains, result (c0+c1)+(c2+c3), every multiply and add rounded on its own.  Used by the
// host-pointer drop-in calls (which are bounded by the host link anyway) and, on request, by the batch entry points:
// outputs are bit-identical to the reference library's.  x(k) returns tap k's sample.
template <typename Sample>
__device__ __forceinline__ float dot_reference_order(const float *__restrict__ w, int ws, Sample x)
{
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    const int r = ws & 3;                                    // ws is odd: 1 or 3
    if (r > 0) c0 = __fadd_rn(c0, __fmul_rn(w[0], x(0)));
    if (r > 1) c1 = __fadd_rn(c1, __fmul_rn(w[1], x(1)));
    if (r > 2) c2 = __fadd_rn(c2, __fmul_rn(w[2], x(2)));
    for (int k = r; k < ws; k += 4) {
        c0 = __fadd_rn(c0, __fmul_rn(w[k], x(k)));
        c1 = __fadd_rn(c1, __fmul_rn(w[k + 1], x(k + 1)));
        c2 = __fadd_rn(c2, __fmul_rn(w[k + 2], x(k + 2)));
        c3 = __fadd_rn(c3, __fmul_rn(w[k + 3], x(k + 3)));
    }
    return __fadd_rn(__fadd_rn(c0, c1), __fadd_rn(c2, c3));
}

// table: row 0 = centre taps, row 1+e = edge row e, ws floats each.  mode = SavgolBoundaryMode (anything else: zeros
// outside the signal, reference :478-480).  Samples g in [store_lo, store_hi) are written to out[g - out_shift].
// negate_leading: SAVGOL_HIP_OPT_CORRECT_LEADING_EDGE for odd derivatives (POLYNOMIAL mode only).
__global__ __launch_bounds__(256) void sg1d_reference_order_kernel(const float *__restrict__ in, float *__restrict__ out, long long in_ld,
                                                                   long long out_ld, int L, int n, const float *__restrict__ table,
                                                                   float dt_inv, int mode, int store_lo, int store_hi, int out_shift,
                                                                   int negate_leading, int wide)
{
    const int ws = 2 * n + 1;
    const float *x = in + (long long)blockIdx.y * in_ld;
    float *y = out + (long long)blockIdx.y * out_ld;
    // a thread owns 4 consecutive outputs; in the interior they share one sliding 4-sample window (ws + 3 loads for
    // 4 outputs), each output still adding its own taps in the reference's chain order
    for (int j0 = store_lo + 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x); j0 < store_hi; j0 += 4 * (int)(gridDim.x * blockDim.x)) {
        if (wide && j0 >= n && j0 + 3 < L - n && j0 + 3 < store_hi) {
            const float *p = x + (j0 - n);
            float c[4][4];
#pragma unroll
            for (int o = 0; o < 4; ++o) { c[o][0] = 0.0f; c[o][1] = 0.0f; c[o][2] = 0.0f; c[o][3] = 0.0f; }
            float v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
            const int r = ws & 3;
            int k = 0;
            for (; k < r; ++k) {                             // taps 0 .. r-1 go to chains 0 .. r-1
                const float w = table[k];
                const float t0 = __fmul_rn(w, v0), t1 = __fmul_rn(w, v1), t2 = __fmul_rn(w, v2), t3 = __fmul_rn(w, v3);
                if (k == 0)      { c[0][0] = __fadd_rn(c[0][0], t0); c[1][0] = __fadd_rn(c[1][0], t1); c[2][0] = __fadd_rn(c[2][0], t2); c[3][0] = __fadd_rn(c[3][0], t3); }
                else if (k == 1) { c[0][1] = __fadd_rn(c[0][1], t0); c[1][1] = __fadd_rn(c[1][1], t1); c[2][1] = __fadd_rn(c[2][1], t2); c[3][1] = __fadd_rn(c[3][1], t3); }
                else             { c[0][2] = __fadd_rn(c[0][2], t0); c[1][2] = __fadd_rn(c[1][2], t1); c[2][2] = __fadd_rn(c[2][2], t2); c[3][2] = __fadd_rn(c[3][2], t3); }
                v0 = v1; v1 = v2; v2 = v3; v3 = (k + 4 < ws + 3) ? p[k + 4] : 0.0f;
            }
            for (; k < ws; k += 4) {                         // then round robin over the four chains
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float w = table[k + q];
                    c[0][q] = __fadd_rn(c[0][q], __fmul_rn(w, v0));
                    c[1][q] = __fadd_rn(c[1][q], __fmul_rn(w, v1));
                    c[2][q] = __fadd_rn(c[2][q], __fmul_rn(w, v2));
                    c[3][q] = __fadd_rn(c[3][q], __fmul_rn(w, v3));
                    v0 = v1; v1 = v2; v2 = v3;
                    v3 = (k + q + 4 < ws + 3) ? p[k + q + 4] : 0.0f;     // the last three shifts run past the 4 windows
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o)
                y[j0 + o - out_shift] = __fmul_rn(__fadd_rn(__fadd_rn(c[o][0], c[o][1]), __fadd_rn(c[o][2], c[o][3])), dt_inv);
            continue;
        }
        for (int j = j0; j < j0 + 4 && j < store_hi; ++j) {
            float v;
            if (j >= n && j < L - n) {
                const float *p = x + (j - n);
                v = dot_reference_order(table, ws, [&](int k) { return p[k]; });
            } else if (mode == SAVGOL_BOUNDARY_POLYNOMIAL) {
                if (j < n) {                                 // leading edge: row j on the first ws samples walked backwards
                    const float *p = x + (ws - 1);
                    v = dot_reference_order(table + (size_t)(1 + j) * ws, ws, [&](int k) { return p[-k]; });
                    if (negate_leading) v = -v;
                } else {                                     // trailing edge: row L-1-j on the last ws samples
                    const float *p = x + (L - ws);
                    v = dot_reference_order(table + (size_t)(1 + (L - 1 - j)) * ws, ws, [&](int k) { return p[k]; });
                }
            } else {                                         // get_padded_sample :442-482
                v = dot_reference_order(table, ws, [&](int k) {
                    bool zero;
                    int i = j - n + k;
                    if (i >= 0 && i < L) return x[i];
                    i = remap_index(i, L, mode, zero);
                    return zero ? 0.0f : x[i];
                });
            }
            y[j - out_shift] = __fmul_rn(v, dt_inv);
        }
    }
}

template <typename T>
static int enqueue_edges(const T *in, T *out, long long in_ld, long long out_ld, long long L, int n,
                        const float *d_edges, float dt_inv, int apply_scale, size_t channels, hipStream_t st)
{
    size_t done = 0;
    while (done < channels) {                       // gridDim.x limit
        const size_t chunk = (channels - done) < 1048576 ? (channels - done) : 1048576;
        hipLaunchKernelGGL((sg1d_edges_kernel<T>), dim3((unsigned)chunk, 2), dim3(64), 0, st,
                           in + done * in_ld, out + done * out_ld, in_ld, out_ld, L, n, d_edges, dt_inv, apply_scale);
        done += chunk;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T>
static int enqueue_synth(T *dst, size_t channel0, size_t channels, size_t length, size_t ld, uint64_t seed, hipStream_t st)
{
    unsigned gx = (unsigned)((length + 255) / 256);
    if (gx > 4096) gx = 4096;
    size_t done = 0;
    while (done < channels) {
        const size_t chunk = (channels - done) < 65535 ? (channels - done) : 65535;
        hipLaunchKernelGGL((sg_synth_kernel<T>), dim3(gx, (unsigned)chunk), dim3(256), 0, st,
                           dst + done * ld, channel0 + done, length, ld, seed);
        done += chunk;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace sg

extern "C" {

int sg1d_launch_edges_f32(const float *in, float *out, long long in_ld, long long out_ld, long long L, int n,
                          const float *d_edges, float dt_inv, int apply_scale, size_t channels, void *st)
{
    return sg::enqueue_edges<float>(in, out, in_ld, out_ld, L, n, d_edges, dt_inv, apply_scale, channels,
                                   static_cast<hipStream_t>(st));
}

int sg1d_launch_edges_f64(const double *in, double *out, long long in_ld, long long out_ld, long long L, int n,
                          const float *d_edges, float dt_inv, int apply_scale, size_t channels, void *st)
{
    return sg::enqueue_edges<double>(in, out, in_ld, out_ld, L, n, d_edges, dt_inv, apply_scale, channels,
                                    static_cast<hipStream_t>(st));
}

int sg1d_launch_reference_order_f32(const float *in, float *out, long long in_ld, long long out_ld, long long L, int n,
                                    const float *d_table, float dt_inv, int mode, int store_lo, int store_hi, int out_shift,
                                    int negate_leading, size_t channels, void *stream)
{
    if (store_hi <= store_lo) return 0;
    // the shared-window path (4 outputs per thread) wins up to n = 10: 548 / 338 / 280 Gsamples/s at n = 2 / 5 / 8 against
    // 347 / 219 / 231 one output per thread; from n = 12 on it loses (146 vs 164 at n = 16, 75 vs 106 at n = 32)
    const int wide = n <= 10;
    unsigned gx = (unsigned)(((size_t)(store_hi - store_lo) + 1023) / 1024);          // a thread owns 4 consecutive outputs
    if (gx > 16384u) gx = 16384u;
    size_t done = 0;
    while (done < channels) {                       // gridDim.y limit
        const size_t chunk = (channels - done) < 65535 ? (channels - done) : 65535;
        hipLaunchKernelGGL(sg::sg1d_reference_order_kernel, dim3(gx, (unsigned)chunk), dim3(256), 0, static_cast<hipStream_t>(stream),
                           in + done * in_ld, out + done * out_ld, in_ld, out_ld, (int)L, n, d_table, dt_inv, mode, store_lo, store_hi,
                           out_shift, negate_leading, wide);
        done += chunk;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int sg1d_launch_edges_strided_f32(const void *in, void *out, long long in_pitch, long long out_pitch, long long in_stride, long long out_stride,
                                  long long L, int n, const float *d_edges, float dt_inv, int flags, size_t channels, void *st)
{
    for (size_t done = 0; done < channels; done += 1048576) {
        const size_t chunk = (channels - done) < 1048576 ? (channels - done) : 1048576;
        hipLaunchKernelGGL(sg::sg1d_edges_strided_kernel, dim3((unsigned)chunk, 2), dim3(64), 0, static_cast<hipStream_t>(st),
                           static_cast<const char *>(in) + (long long)done * in_pitch, static_cast<char *>(out) + (long long)done * out_pitch, in_pitch, out_pitch,
                           in_stride, out_stride, L, n, d_edges, dt_inv, flags);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int sg_launch_gather_f32(const void *base, size_t stride, size_t offset, size_t pitch, float *dst, size_t dst_ld,
                         size_t channels, size_t count, void *st)
{
    unsigned gx = (unsigned)((count + 255) / 256);
    if (gx > 2048) gx = 2048;
    for (size_t done = 0; done < channels; done += 65535) {
        const size_t chunk = (channels - done) < 65535 ? (channels - done) : 65535;
        hipLaunchKernelGGL(sg::sg_gather_f32_kernel, dim3(gx, (unsigned)chunk), dim3(256), 0, static_cast<hipStream_t>(st),
                           static_cast<const char *>(base) + done * pitch, stride, offset, pitch, dst + done * dst_ld,
                           dst_ld, count);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int sg_launch_scatter_f32(const float *src, size_t src_ld, void *base, size_t stride, size_t offset, size_t pitch,
                          size_t channels, size_t count, void *st)
{
    unsigned gx = (unsigned)((count + 255) / 256);
    if (gx > 2048) gx = 2048;
    for (size_t done = 0; done < channels; done += 65535) {
        const size_t chunk = (channels - done) < 65535 ? (channels - done) : 65535;
        hipLaunchKernelGGL(sg::sg_scatter_f32_kernel, dim3(gx, (unsigned)chunk), dim3(256), 0, static_cast<hipStream_t>(st),
                           src + done * src_ld, src_ld, static_cast<char *>(base) + done * pitch, stride, offset, pitch,
                           count);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int savgol_hip_synth_f32(float *d_dst, size_t channel0, size_t channels, size_t length, size_t ld, uint64_t seed, void *st)
{
    if (!d_dst || ld < length) { sg_set_error("savgol_hip_synth_f32: bad arguments"); return -1; }
    if (channels == 0 || length == 0) return 0;
    return sg::enqueue_synth<float>(d_dst, channel0, channels, length, ld, seed, static_cast<hipStream_t>(st));
}

int savgol_hip_synth_f64(double *d_dst, size_t channel0, size_t channels, size_t length, size_t ld, uint64_t seed, void *st)
{
    if (!d_dst || ld < length) { sg_set_error("savgol_hip_synth_f64: bad arguments"); return -1; }
    if (channels == 0 || length == 0) return 0;
    return sg::enqueue_synth<double>(d_dst, channel0, channels, length, ld, seed, static_cast<hipStream_t>(st));
}

}  // extern "C"
