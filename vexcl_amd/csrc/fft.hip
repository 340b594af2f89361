// vex::FFT on gfx950 (reference: vexcl/fft.hpp, vexcl/fft/plan.hpp:214-330, fft/kernels.hpp -- a generator of
// global-memory radix passes, one launch per radix, plus a transpose per dimension).
//
// MI355X design.  A transform is HBM-bound, so the plan is built around ONE kernel that moves every element
// once per pass and does as much of the transform as fits in LDS in that pass.  `fft_lines_kernel` brings a
// tile of LINES (1-D sub-sequences of the data: any element stride, line offsets given by a small mixed-radix
// table) into LDS, runs ALL radix stages of their length there (Stockham autosort between two LDS buffers,
// 256 lanes, radix 8/4/2 for the power-of-two part and 3/5/7/11/13 for the rest, twiddles from a per-length
// table that stays in L2), optionally multiplies by the inter-pass twiddle, and writes the lines back through
// a second offset table.  Global accesses are coalesced along whichever direction is contiguous: along the
// line when its element stride is 1, ACROSS the lines of the tile otherwise (LDS rows are padded by one element
// so both directions are conflict-free).  With that one kernel
//   * a contiguous row of up to 2048 (fp64) / 4096 (fp32) elements -- twice that for powers of two, which run their
//     stages register-resident in ONE LDS buffer (stage_inplace) -- is ONE pass, whatever its number of radix
//     stages (the reference launches one global-memory pass per radix: 3-4 for n = 4096);
//   * a longer length n = n1 n2 [n3] is 2 [3] passes and NO transpose: pass t transforms digit t in place (lines
//     strided by the product of the later factors, tile = neighbouring lines, twiddle W^(k_t J) fused into the
//     store), the last pass reads contiguous lines and scatters them to the digit-reversed position -- a scatter
//     whose tile writes runs of consecutive elements, because the tile's lines are enumerated in output order;
//   * the dimensions of an n-D transform (and batches) need no transposes either: a dimension with element
//     stride s is a set of lines with stride s, neighbouring lines adjacent in memory.
// The reference rotates the array with one transpose per dimension and the textbook four-step algorithm
// has three; here 4096 x 4096 fp64 is 3 passes over the data instead of 12.  Lengths with a prime factor above
// 13 go through Bluestein's chirp-z over a 2^a 3^b 5^c 7^d convolution length (rows gathered by a tiled LDS
// transpose when the dimension is strided).  The plan (factorizations, passes, twiddle / chirp tables, work
// buffers) is native C++ behind four C entry points.
// Measured on MI355X (tools/fft_bench.py): fp64 1024-point rows 0.54 ms per 2.1 GB moved (3.9 TB/s; torch.fft 0.81 ms, rocFFT's
// kernel alone 0.38), 2^24 points 0.50 ms in 3 passes (torch.fft 0.61), 4096 x 4096 0.48 ms in 3 passes (0.50).  Tried and dropped: an LDS
// layout skewed by one element per eight (removes the bank conflicts of the first stage's writes, but the extra
// index arithmetic on every access cost more: 0.76 -> 0.94 ms on the 1024-point rows).
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

namespace vexhip {
namespace {

template <typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };

template <typename T> __host__ __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __host__ __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __host__ __device__ __forceinline__ cx<T> operator*(cx<T> a, cx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

constexpr int FB = 256;                                                  // lanes per workgroup
template <typename T> constexpr int lds_elems() { return sizeof(T) == 8 ? 2048 : 4096; }   // per LDS buffer: 32 KiB
constexpr int MAX_STAGES = 16;

struct stage_list { int count; int radix[MAX_STAGES]; };

// ---- butterflies -------------------------------------------------------------------------------------------
// root[t] = W_R^t with the sign of the direction already applied.
template <typename T, int R> struct dft;

template <typename T> struct dft<T, 2> {
    static __device__ __forceinline__ void run(cx<T> (&v)[2], const cx<T> *, bool) {
        const cx<T> a = v[0], b = v[1];
        v[0] = a + b; v[1] = a - b;
    }
};
/// v * (-i) for the forward transform, v * (+i) for the inverse.
template <typename T> __device__ __forceinline__ cx<T> mul_w4(cx<T> v, bool inverse) {
    return inverse ? cx<T>{-v.y, v.x} : cx<T>{v.y, -v.x};
}
template <typename T> __device__ __forceinline__ void dft4(cx<T> &a, cx<T> &b, cx<T> &c, cx<T> &d, bool inverse) {
    const cx<T> t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_w4(b - d, inverse);
    a = t0 + t2; b = t1 + t3; c = t0 - t2; d = t1 - t3;
}
template <typename T> struct dft<T, 4> {
    static __device__ __forceinline__ void run(cx<T> (&v)[4], const cx<T> *, bool inverse) { dft4(v[0], v[1], v[2], v[3], inverse); }
};
template <typename T> struct dft<T, 8> {
    static __device__ __forceinline__ void run(cx<T> (&v)[8], const cx<T> *, bool inverse) {
        dft4(v[0], v[2], v[4], v[6], inverse);          // even samples -> E[0..3] in v[0], v[2], v[4], v[6]
        dft4(v[1], v[3], v[5], v[7], inverse);          // odd samples  -> O[0..3]
        const T h = (T)0.70710678118654752440;
        const cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        // O[k] * W_8^k: W_8 = (1 -+ i) / sqrt 2, W_8^2 = -+i, W_8^3 = (-1 -+ i) / sqrt 2
        const cx<T> o0 = v[1];
        const cx<T> r1 = mul_w4(v[3], inverse), r3 = mul_w4(v[7], inverse);
        const cx<T> o1 = {(v[3].x + r1.x) * h, (v[3].y + r1.y) * h};
        const cx<T> o2 = mul_w4(v[5], inverse);
        const cx<T> o3 = {(r3.x - v[7].x) * h, (r3.y - v[7].y) * h};
        v[0] = e0 + o0; v[4] = e0 - o0;
        v[1] = e1 + o1; v[5] = e1 - o1;
        v[2] = e2 + o2; v[6] = e2 - o2;
        v[3] = e3 + o3; v[7] = e3 - o3;
    }
};
/// Odd primes: the definition, unrolled (the exponents fold to constants).
template <typename T, int R> struct dft {
    static __device__ __forceinline__ void run(cx<T> (&v)[R], const cx<T> *root, bool) {
        cx<T> out[R];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            cx<T> acc = v[0];
#pragma unroll
            for (int q = 1; q < R; ++q) acc = acc + v[q] * root[(q * s) % R];
            out[s] = acc;
        }
#pragma unroll
        for (int s = 0; s < R; ++s) v[s] = out[s];
    }
};

/// Division by a run-time constant that is usually a power of two (shift < 0: not a power of two).
struct divisor {
    int d, shift;
    __host__ __device__ static divisor make(int v) {
        divisor r; r.d = v; r.shift = -1;
        if (v > 0 && (v & (v - 1)) == 0) { r.shift = 0; while ((1 << r.shift) < v) ++r.shift; }
        return r;
    }
    __device__ __forceinline__ int div(int x) const { return shift >= 0 ? (x >> shift) : (x / d); }
};

/// Pointwise work fused into the global accesses of contiguous lines (Bluestein): element i of a line is read as
/// i < pre_n ? x[i] * pre[i] : 0 (pre_n = 0: no limit, pre = null: no factor) and written, only if i < post_n, as
/// X[i] * post[i] * scale.
template <typename T> struct io_ops { const cx<T> *pre; int pre_n; const cx<T> *post; int post_n; T scale; };

/// One Stockham stage over the lines held in LDS (line l at l * pitch): `src` -> `dst`, sub-transform length p -> p * R.
/// `gin` / `gout` non-null: the stage reads its inputs straight from global memory (first stage of contiguous lines:
/// lane j reads elements j + q * nb, coalesced for every q) / writes its outputs straight to global memory (last
/// stage: p = nb, so lane j writes elements j + s * nb) instead of going through LDS.
template <typename T, int R, bool FUSED>
__device__ __forceinline__ void stage(const cx<T> *__restrict__ src, cx<T> *__restrict__ dst,
        const cx<T> *__restrict__ tw, int n, int pitch, int p, int nlines, bool inverse,
        const cx<T> *__restrict__ gin, const long long *in_off, cx<T> *__restrict__ gout, const long long *out_off,
        const io_ops<T> &io)
{
    const int nb = n / R;                        // butterflies per line
    const int tstride = n / (p * R);             // W_(pR)^k = tw[k * tstride]
    const divisor dnb = divisor::make(nb), dp = divisor::make(p);
    constexpr bool needs_roots = R != 2 && R != 4 && R != 8;
    cx<T> root[needs_roots ? R : 1];
    if constexpr (needs_roots) {
#pragma unroll
        for (int t = 0; t < R; ++t) { root[t] = tw[(size_t)t * nb]; if (inverse) root[t].y = -root[t].y; }
    }
    const int total = nlines * nb;
    for (int b = threadIdx.x; b < total; b += (int)blockDim.x) {
        const int line = dnb.div(b), j = b - line * nb;
        const int k = j - dp.div(j) * p;
        cx<T> v[R];
        if (gin) {
            const cx<T> *in = gin + in_off[line] + j;
            if constexpr (!FUSED) {
#pragma unroll
                for (int q = 0; q < R; ++q) v[q] = in[q * nb];
            } else {
                // all loads unconditional (clamped index), so that they stay in flight together; the cut is a select
                const cx<T> *line0 = gin + in_off[line];
                const int last = (io.pre_n ? io.pre_n : n) - 1;
                cx<T> f[R];
#pragma unroll
                for (int q = 0; q < R; ++q) { const int i = min(j + q * nb, last); v[q] = line0[i]; f[q] = io.pre ? io.pre[i] : cx<T>{T(1), T(0)}; }
#pragma unroll
                for (int q = 0; q < R; ++q) { const cx<T> x = v[q] * f[q]; v[q] = (j + q * nb <= last) ? x : cx<T>{T(0), T(0)}; }
            }
        } else {
            const cx<T> *in = src + line * pitch + j;
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = in[q * nb];
        }
        if (p > 1) {
            cx<T> w1 = tw[(size_t)k * tstride];
            if (inverse) w1.y = -w1.y;
            cx<T> w = w1;
#pragma unroll
            for (int q = 1; q < R; ++q) { v[q] = v[q] * w; if (q + 1 < R) w = w * w1; }
        }
        dft<T, R>::run(v, root, inverse);
        if (gout) {
            const int o = (j - k) * R + k;
            cx<T> *out = gout + out_off[line] + o;
            if constexpr (!FUSED) {
#pragma unroll
                for (int s = 0; s < R; ++s) out[s * p] = v[s];
            } else {
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const int i = o + s * p;
                    if (io.post_n == 0 || i < io.post_n) {
                        const cx<T> x = io.post ? v[s] * io.post[i] : v[s];
                        out[s * p] = {x.x * io.scale, x.y * io.scale};
                    }
                }
            }
        } else {
            cx<T> *out = dst + line * pitch + (j - k) * R + k;
#pragma unroll
            for (int s = 0; s < R; ++s) out[s * p] = v[s];
        }
    }
}

/// The same stage for the register-resident organisation of power-of-two lengths: the workgroup has one lane per EPT (8 or
/// 16) elements of its tile, so a lane owns EPT / R butterflies of every stage and keeps its elements in registers;
/// all of them are read before the workgroup's barrier and written after it, which lets the stage work IN PLACE in a
/// single LDS buffer (half the LDS of the two-buffer scheme: twice the workgroups per CU).
template <typename T, int R, int EPT, bool FUSED>
__device__ __forceinline__ void stage_inplace(cx<T> *__restrict__ buf, const cx<T> *__restrict__ tw, int n, int pitch, int p,
        int nlines, bool inverse, const cx<T> *__restrict__ gin, const long long *in_off, cx<T> *__restrict__ gout, const long long *out_off,
        const io_ops<T> &io)
{
    constexpr int K = (EPT + R - 1) / R;         // butterflies per lane: blockDim >= tile elements / EPT covers every stage
    const int nb = n / R, tstride = n / (p * R);
    const divisor dnb = divisor::make(nb), dp = divisor::make(p);
    const int total = nlines * nb;
    constexpr bool needs_roots = R != 2 && R != 4 && R != 8;
    cx<T> root[needs_roots ? R : 1];
    if constexpr (needs_roots) {
#pragma unroll
        for (int t = 0; t < R; ++t) { root[t] = tw[(size_t)t * nb]; if (inverse) root[t].y = -root[t].y; }
    }
    cx<T> v[K][R];
    int line[K], j[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const int b = threadIdx.x + t * (int)blockDim.x;
        line[t] = dnb.div(b); j[t] = b - line[t] * nb;
        if (b < total) {
            if (FUSED && gin) {                  // clamped, unconditional loads; the cut at pre_n is a select
                const cx<T> *line0 = gin + in_off[line[t]];
                const int last = (io.pre_n ? io.pre_n : n) - 1;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int i = min(j[t] + q * nb, last);
                    const cx<T> x = io.pre ? line0[i] * io.pre[i] : line0[i];
                    v[t][q] = (j[t] + q * nb <= last) ? x : cx<T>{T(0), T(0)};
                }
            } else {
                const cx<T> *in = gin ? gin + in_off[line[t]] + j[t] : buf + line[t] * pitch + j[t];
#pragma unroll
                for (int q = 0; q < R; ++q) v[t][q] = in[q * nb];
            }
        }
    }
    if (!gin) __syncthreads();                   // every lane has its inputs: the buffer may be overwritten
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const int b = threadIdx.x + t * (int)blockDim.x;
        if (b < total) {
            const int k = j[t] - dp.div(j[t]) * p;
            if (p > 1) {
                cx<T> w1 = tw[(size_t)k * tstride];
                if (inverse) w1.y = -w1.y;
                cx<T> w = w1;
#pragma unroll
                for (int q = 1; q < R; ++q) { v[t][q] = v[t][q] * w; if (q + 1 < R) w = w * w1; }
            }
            dft<T, R>::run(v[t], root, inverse);
            const int o = (j[t] - k) * R + k;
            cx<T> *out = gout ? gout + out_off[line[t]] + o : buf + line[t] * pitch + o;
            if (FUSED && gout) {
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const int i = o + s * p;
                    if (io.post_n == 0 || i < io.post_n) {
                        const cx<T> x = io.post ? v[t][s] * io.post[i] : v[t][s];
                        out[s * p] = {x.x * io.scale, x.y * io.scale};
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < R; ++s) out[s * p] = v[t][s];
            }
        }
    }
}

/// Where the lines of a pass live: line g (counted over the whole launch) is decomposed in the mixed radix
/// `extent[0]` (fastest) ... `extent[nlv - 1]`; its first element is at sum digit_i * in_stride[i] in the source and at
/// sum digit_i * out_stride[i] in the destination; element k of the line is k * in_es / k * out_es further.
constexpr int MAX_LEVELS = 4;
struct line_map {
    int nlv;
    long long extent[MAX_LEVELS], in_stride[MAX_LEVELS], out_stride[MAX_LEVELS];
    long long in_es, out_es;
    // inter-pass twiddle: element k of line g is multiplied by W_M^(k * J), J = (g / tw_div) % tw_mod; tw_M = 0: none
    long long tw_M, tw_div, tw_mod;
    // pointwise factors fused into the global accesses (io_ops), contiguous lines only
    const void *pre, *post; int pre_n, post_n; double post_scale;
};

constexpr int MAX_LINES = 256;                   // lines per workgroup (size of the offset tables in LDS)

/// A tile of lines, each transformed completely in LDS.
/// ODD = false: the instantiation for lengths 2^a (radix 2 / 4 / 8 stages only: half the registers of the general one).
/// FUSED = true: the instantiation whose global accesses carry pointwise factors (io_ops; Bluestein).
/// EPT > 0 (2^a lengths, no fused factors): register-resident stages in one LDS buffer, blockDim = tile elements / EPT.
template <typename T, bool ODD, bool FUSED, int EPT>
__global__ __launch_bounds__(FB)
void fft_lines_kernel(const cx<T> *__restrict__ in, cx<T> *__restrict__ out, const cx<T> *__restrict__ tw,
        int n, long long lines, int lines_per_wg, int pitch, stage_list st, int inverse, line_map map)
{
    extern __shared__ __attribute__((aligned(16))) char fft_smem[];
    __shared__ long long in_off[MAX_LINES], out_off[MAX_LINES], tw_j[MAX_LINES];
    const long long g0 = (long long)blockIdx.x * lines_per_wg;
    const int nl = (int)min((long long)lines_per_wg, lines - g0);
    const int E = nl * n;
    cx<T> *A = reinterpret_cast<cx<T> *>(fft_smem);
    constexpr bool SINGLE = EPT > 0;
    cx<T> *B = SINGLE ? A : A + (size_t)lines_per_wg * pitch;

    for (int l = threadIdx.x; l < nl; l += (int)blockDim.x) {
        long long g = g0 + l, io = 0, oo = 0;
        tw_j[l] = map.tw_M ? (g / map.tw_div) % map.tw_mod : 0;
        for (int i = 0; i < map.nlv; ++i) {
            const long long q = (i + 1 < map.nlv) ? g / map.extent[i] : 0;
            const long long d = (i + 1 < map.nlv) ? g - q * map.extent[i] : g;
            io += d * map.in_stride[i]; oo += d * map.out_stride[i];
            g = q;
        }
        in_off[l] = io; out_off[l] = oo;
    }
    __syncthreads();

    const divisor dn = divisor::make(n), dl = divisor::make(nl);
    // contiguous lines: the first stage reads global memory itself, the last one writes it (no staging copy in LDS);
    // strided lines: neighbouring lanes move the same element of neighbouring lines through LDS
    const bool direct_in = map.in_es == 1, direct_out = map.out_es == 1 && map.tw_M == 0;
    const io_ops<T> io = {static_cast<const cx<T> *>(map.pre), map.pre_n, static_cast<const cx<T> *>(map.post), map.post_n, (T)map.post_scale};
    if (!direct_in) {
        for (int e = threadIdx.x; e < E; e += (int)blockDim.x) {
            const int k = dl.div(e), l = e - k * nl;
            A[l * pitch + k] = in[in_off[l] + k * map.in_es];
        }
        __syncthreads();
    }

    int p = 1;
    for (int s = 0; s < st.count; ++s) {
        const int R = st.radix[s];
        const cx<T> *gi = (s == 0 && direct_in) ? in : nullptr;
        cx<T> *go = (s + 1 == st.count && direct_out) ? out : nullptr;
        if constexpr (SINGLE && ODD) {
            switch (R) {
                case 2:  stage_inplace<T, 2, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 3:  stage_inplace<T, 3, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 4:  stage_inplace<T, 4, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 5:  stage_inplace<T, 5, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 7:  stage_inplace<T, 7, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 8:  stage_inplace<T, 8, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 11: stage_inplace<T, 11, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                default: stage_inplace<T, 13, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
            }
        } else if constexpr (SINGLE) {
            switch (R) {
                case 2:  stage_inplace<T, 2, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 4:  stage_inplace<T, 4, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                default: stage_inplace<T, 8, EPT, FUSED>(A, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
            }
        } else if constexpr (ODD) {
            switch (R) {
                case 2:  stage<T, 2, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 3:  stage<T, 3, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 4:  stage<T, 4, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 5:  stage<T, 5, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 7:  stage<T, 7, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 8:  stage<T, 8, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 11: stage<T, 11, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                default: stage<T, 13, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
            }
        } else {
            switch (R) {
                case 2:  stage<T, 2, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                case 4:  stage<T, 4, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
                default: stage<T, 8, FUSED>(A, B, tw, n, pitch, p, nl, inverse, gi, in_off, go, out_off, io); break;
            }
        }
        p *= R;
        cx<T> *t = A; A = B; B = t;
        __syncthreads();
    }

    if (direct_out) return;
    const bool along = map.out_es == 1;
    for (int e = threadIdx.x; e < E; e += (int)blockDim.x) {
        int l, k;
        if (along) { l = dn.div(e); k = e - l * n; } else { k = dl.div(e); l = e - k * nl; }
        cx<T> v = A[l * pitch + k];
        if (map.tw_M) {
            const double frac = 2.0 * (double)(tw_j[l] * k) / (double)map.tw_M;        // J * k < M: no reduction needed
            double sn, cs;
            sincospi(frac, &sn, &cs);
            const cx<T> w = {(T)cs, (T)(inverse ? sn : -sn)};
            v = v * w;
        }
        out[out_off[l] + k * map.out_es] = v;
    }
}

/// in[b][R][C] -> out[b][C][R], 32 x 32 tiles through LDS.
template <typename T>
__global__ __launch_bounds__(FB)
void fft_transpose_kernel(const cx<T> *__restrict__ in, cx<T> *__restrict__ out, long long R, long long C, long long tiles_c, long long tiles_per_mat)
{
    __shared__ cx<T> tile[32][33];
    const long long b = blockIdx.x / tiles_per_mat, t = blockIdx.x % tiles_per_mat;
    const long long tr = t / tiles_c, tc = t % tiles_c;
    const cx<T> *src = in + b * R * C;
    cx<T> *dst = out + b * R * C;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const long long r = tr * 32 + ly + i, c = tc * 32 + lx;
        if (r < R && c < C) tile[ly + i][lx] = src[r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const long long c = tc * 32 + ly + i, r = tr * 32 + lx;
        if (r < R && c < C) dst[c * R + r] = tile[lx][ly + i];
    }
}

/// Bluestein, in: a[row][k] = k < n ? x[row][k] * chirp[k] : 0   (rows of length m)
template <typename T>
__global__ void bluestein_in_kernel(const cx<T> *__restrict__ x, const cx<T> *__restrict__ chirp, cx<T> *__restrict__ a,
        long long n, long long m, long long total /* rows * m */)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / m, k = i - row * m;
        cx<T> v = {T(0), T(0)};
        if (k < n) v = x[row * n + k] * chirp[k];
        a[i] = v;
    }
}
/// a[row][k] *= bhat[k]
template <typename T>
__global__ void bluestein_mul_kernel(cx<T> *__restrict__ a, const cx<T> *__restrict__ bhat, long long m, long long total)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        a[i] = a[i] * bhat[i % m];
}
/// y[row][k] = a[row][k] * chirp[k] / m, k < n
template <typename T>
__global__ void bluestein_out_kernel(const cx<T> *__restrict__ a, const cx<T> *__restrict__ chirp, cx<T> *__restrict__ y,
        long long n, long long m, long long total /* rows * n */, T inv_m)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / n, k = i - row * n;
        cx<T> v = a[row * m + k] * chirp[k];
        y[i] = {v.x * inv_m, v.y * inv_m};
    }
}

// ---- planning ------------------------------------------------------------------------------------------------
const int kPrimes[] = {2, 3, 5, 7, 11, 13};

/// true and the radix list if n factors into the supported primes.
bool factor(size_t n, stage_list &st) {
    st.count = 0;
    int twos = 0;
    while (n % 2 == 0) { n /= 2; ++twos; }
    // power-of-two part: radix 8 as far as possible, the remainder as 4 / 2
    std::vector<int> r2;
    while (twos >= 3 && twos != 4) { r2.push_back(8); twos -= 3; }
    while (twos >= 2) { r2.push_back(4); twos -= 2; }
    if (twos) r2.push_back(2);
    std::vector<int> rest;
    for (int p : {3, 5, 7, 11, 13}) while (n % p == 0) { rest.push_back(p); n /= p; }
    if (n != 1 || r2.size() + rest.size() > (size_t)MAX_STAGES) return false;
    for (int r : r2) st.radix[st.count++] = r;
    for (int r : rest) st.radix[st.count++] = r;
    return true;
}
bool smooth(size_t n) { for (int p : kPrimes) while (n % p == 0) n /= p; return n == 1; }

size_t best_size(size_t n) {
    // smallest 2^a 3^b 5^c 7^d >= n
    if (n <= 1) return 1;
    size_t best = ~size_t(0);
    for (size_t p7 = 1; p7 < 2 * n && p7 < best; p7 *= 7)
        for (size_t p5 = p7; p5 < 2 * n && p5 < best; p5 *= 5)
            for (size_t p3 = p5; p3 < 2 * n && p3 < best; p3 *= 3) {
                size_t v = p3;
                while (v < n) v *= 2;
                best = std::min(best, v);
            }
    return best;
}

enum buf_id { B_IN = 0, B_OUT = 1, B_WORK = 2, B_FIRST_OWNED = 3 };

struct step {
    enum kind_t { LINES, TRANSPOSE, BLUE_IN, BLUE_MUL, BLUE_OUT, COPY } kind;
    int src, dst;
    // LINES
    int n = 0; long long lines = 0; int lines_per_wg = 1, pitch = 0, threads = FB, ept = 0; stage_list st{}; int inverse = 0; int table = -1; line_map map{};
    // TRANSPOSE: [batch][R][C] -> [batch][C][R]
    long long batch = 0, R = 0, C = 0;
    // BLUESTEIN
    long long rows = 0, bn = 0, bm = 0; int chirp = -1, bhat = -1;
    long long elems = 0;        // COPY
};

/// Split n (smooth) into m factors, each as close to n^(1/m) as its divisors allow; the largest last.
std::vector<size_t> split(size_t n, int m) {
    std::vector<size_t> f;
    size_t rest = n;
    for (int left = m; left > 1; --left) {
        const double target = std::pow((double)rest, 1.0 / left);
        size_t best = 1;
        for (size_t d = 1; d * d <= rest; ++d) if (rest % d == 0) {
            for (size_t c : {d, rest / d})
                if (std::fabs(std::log((double)c / target)) < std::fabs(std::log((double)best / target))) best = c;
        }
        f.push_back(best);
        rest /= best;
    }
    f.push_back(rest);
    std::sort(f.begin(), f.end());
    return f;
}

template <typename T>
struct plan_t {
    int dev = 0;
    size_t total = 0;
    std::vector<step> steps;
    std::vector<void *> owned;                 // device buffers of the plan: work buffer, Bluestein buffers, tables
    std::vector<std::pair<size_t, int>> tw_tables;     // (n, owned index) twiddle tables already built

    ~plan_t() { (void)hipSetDevice(dev); for (void *p : owned) if (p) (void)hipFree(p); }

    int alloc(size_t bytes, int &id) {
        void *p = nullptr;
        VEXHIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        owned.push_back(p);
        id = B_FIRST_OWNED + (int)owned.size() - 1;
        return 0;
    }
    int upload(const std::vector<cx<T>> &h, int &id) {
        if (int rc = alloc(h.size() * sizeof(cx<T>), id)) return rc;
        VEXHIP_TRY(hipMemcpy(owned[id - B_FIRST_OWNED], h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
        return 0;
    }
    /// W_n^t = exp(-2 pi i t / n), t < n
    int twiddles(size_t n, int &id) {
        for (auto &t : tw_tables) if (t.first == n) { id = t.second; return 0; }
        std::vector<cx<T>> h(n);
        const long double w = -2.0L * 3.141592653589793238462643383279502884L / (long double)n;
        for (size_t t = 0; t < n; ++t) h[t] = {(T)std::cos(w * t), (T)std::sin(w * t)};
        if (int rc = upload(h, id)) return rc;
        tw_tables.push_back({n, id});
        return 0;
    }

    static bool writable(int b) { return b != B_IN; }

    /// One launch of the lines kernel.  `in_place`: source and destination offsets coincide, so a writable source
    /// is transformed where it is; otherwise the pass goes to the other buffer of the pair (a, b).
    int emit_pass(size_t n, bool inverse, const line_map &map, long long lines, bool in_place, int cur, int a, int b, int &res, int dst = -1) {
        step s; s.kind = step::LINES; s.src = cur;
        s.dst = dst >= 0 ? dst : (in_place && writable(cur)) ? cur : (cur == a ? b : a);
        s.n = (int)n; s.lines = lines; s.inverse = inverse; s.map = map;
        if (!factor(n, s.st)) return fail(__FILE__, __LINE__, "fft: internal error, unsupported pass length");
        if (int rc = twiddles(n, s.table)) return rc;
        // lines per workgroup: what fits the LDS buffers (one padding element per line when there are several);
        // contiguous lines may take fewer to keep >= ~4 workgroups per CU, strided ones keep the tile wide
        // (the tile's width IS the contiguous run of the global accesses)
        long long L = std::min<long long>(MAX_LINES, std::max<long long>(1, lds_elems<T>() / (long long)n));
        while (L > 1 && L * ((long long)n + 1) > lds_elems<T>() + MAX_LINES) --L;
        if (!(map.in_es == 1 && map.out_es == 1)) {                    // strided passes: tiles of 2048 elements measured best for
            long long cap = 2048;                                      // both types (tools/fft_strided_experiment.py)
            if (const char *e = env(ENV_VEXHIP_FFT_STRIDED_ELEMS)) cap = std::max(1, atoi(e));
            if (cap >= (long long)n) L = std::max<long long>(1, std::min<long long>(L, cap / (long long)n));
        }
        long long row_elems = lds_elems<T>() / 2;                     // contiguous lines: <= 16 KiB per buffer
        if (const char *e = env(ENV_VEXHIP_FFT_ROW_ELEMS)) row_elems = std::max(1, atoi(e));      // tuning knob (tools/fft_bench.py)
        if (map.in_es == 1 && map.out_es == 1) {
            L = std::max<long long>(1, std::min(L, row_elems / (long long)n));
            L = std::max<long long>(1, std::min(L, (lines + 2047) / 2048));
        }
        L = std::min(L, std::max<long long>(lines, 1));
        s.lines_per_wg = (int)L;
        s.pitch = (int)n + (L > 1 ? 1 : 0);
        // lanes: one per butterfly of the widest stage, in whole waves
        int min_radix = s.st.radix[0];
        for (int i = 1; i < s.st.count; ++i) min_radix = std::min(min_radix, s.st.radix[i]);
        long long want = L * ((long long)n / min_radix);
        if (const char *e = env(ENV_VEXHIP_FFT_LANES_DIV)) want /= std::max(1, atoi(e));
        s.threads = (int)std::min<long long>(FB, std::max<long long>(kWave, (want + kWave - 1) / kWave * kWave));
        // register-resident stages in place in one LDS buffer: one lane per 8 (16, 32) elements of the tile
        const long long E = L * (long long)n;
        const bool pow2 = (n & (n - 1)) == 0, plain = !map.pre && !map.pre_n && !map.post && !map.post_n;
        if ((plain || pow2) && !env(ENV_VEXHIP_FFT_NO_SINGLE)) {
            for (int ept : {8, 16, 32}) {
                const long long lanes = ((E + ept - 1) / ept + kWave - 1) / kWave * kWave;
                const bool allowed = ept == 8 || (ept == 16 && (sizeof(T) == 4 || (pow2 && E > 2048))) || (ept == 32 && sizeof(T) == 4 && pow2);
                if (allowed && lanes <= FB) { s.ept = ept; s.threads = (int)lanes; break; }
            }
        }
        if ((long long)n > lds_elems<T>() && !s.ept) return fail(__FILE__, __LINE__, "fft: internal error, row does not fit LDS");
        steps.push_back(s);
        res = s.dst;
        return 0;
    }

    /// Transform along one dimension: `outer` blocks of [w][s] elements, lines of length w with element stride s.
    int emit_dim(size_t w, size_t s, size_t outer, bool inverse, int cur, int a, int b, int &res) {
        res = cur;
        if (w == 1) return 0;
        const long long W = (long long)w, S = (long long)s, O = (long long)outer;
        if (!smooth(w)) {
            if (s == 1) return emit_bluestein(w, O, inverse, cur, a, b, res);
            // gather the lines into rows, chirp-z them, scatter back: [outer][w][s] -> [outer][s][w] -> ... -> [outer][w][s]
            auto other = [&](int c) { return c == a ? b : a; };
            step t1; t1.kind = step::TRANSPOSE; t1.src = cur; t1.dst = writable(cur) ? other(cur) : a; t1.batch = O; t1.R = W; t1.C = S;
            steps.push_back(t1);
            int c = t1.dst;
            if (int rc = emit_bluestein(w, O * S, inverse, c, a, b, c)) return rc;
            step t2; t2.kind = step::TRANSPOSE; t2.src = c; t2.dst = other(c); t2.batch = O; t2.R = S; t2.C = W;
            steps.push_back(t2);
            res = t2.dst;
            return 0;
        }
        // one pass when a useful tile of lines fits LDS: any contiguous row that fits; strided lines need >= 4 per tile
        size_t cap = s == 1 ? (size_t)lds_elems<T>() : (size_t)lds_elems<T>() / 4;
        // a contiguous power-of-two row of twice that length still fits the single-buffer organisation (64 KiB of LDS)
        if (s == 1 && (w & (w - 1)) == 0 && !env(ENV_VEXHIP_FFT_NO_SINGLE)) cap = 2 * (size_t)lds_elems<T>();
        if (w <= cap) {
            line_map m{};
            m.in_es = m.out_es = S;
            m.nlv = 0;
            if (s > 1) { m.extent[m.nlv] = S; m.in_stride[m.nlv] = m.out_stride[m.nlv] = 1; ++m.nlv; }
            m.extent[m.nlv] = O; m.in_stride[m.nlv] = m.out_stride[m.nlv] = W * S; ++m.nlv;
            return emit_pass(w, inverse, m, O * S, true, cur, a, b, res);
        }
        // w = n_1 ... n_m: pass t transforms digit t of the index j = j_1 (n_2..n_m) + ... + j_m
        const int nf = w <= (size_t)1 << 16 ? 2 : 3;
        const std::vector<size_t> f = split(w, nf);
        for (size_t x : f) if (x > (size_t)lds_elems<T>()) return fail(__FILE__, __LINE__, "fft: length too large");
        long long M = W;                                 // n_t ... n_m
        long long done = 1;                              // n_1 ... n_(t-1)
        for (int t = 0; t + 1 < nf; ++t) {
            const long long nt = (long long)f[t], rest = M / nt;
            line_map m{};
            m.in_es = m.out_es = rest * S;
            m.nlv = 0;
            m.extent[m.nlv] = rest * S; m.in_stride[m.nlv] = m.out_stride[m.nlv] = 1; ++m.nlv;            // (j_(t+1..m), i)
            if (done > 1) { m.extent[m.nlv] = done; m.in_stride[m.nlv] = m.out_stride[m.nlv] = M * S; ++m.nlv; }   // (k_1..k_(t-1))
            m.extent[m.nlv] = O; m.in_stride[m.nlv] = m.out_stride[m.nlv] = W * S; ++m.nlv;
            m.tw_M = M; m.tw_div = S; m.tw_mod = rest;
            if (int rc = emit_pass((size_t)nt, inverse, m, O * (W / nt) * S, true, res, a, b, res)) return rc;
            M = rest; done *= nt;
        }
        {   // last digit: lines contiguous in the source when s == 1; written to the digit-reversed position,
            // lines enumerated in OUTPUT order (i, k_1, k_2, ...) so that a tile writes runs of consecutive elements
            const long long nm = (long long)f[nf - 1];
            line_map m{};
            m.in_es = S; m.out_es = (W / nm) * S;
            m.nlv = 0;
            if (s > 1) { m.extent[m.nlv] = S; m.in_stride[m.nlv] = m.out_stride[m.nlv] = 1; ++m.nlv; }
            long long in_str = W * S, out_str = S;
            for (int q = 0; q + 1 < nf; ++q) {
                in_str /= (long long)f[q];
                m.extent[m.nlv] = (long long)f[q]; m.in_stride[m.nlv] = in_str; m.out_stride[m.nlv] = out_str; ++m.nlv;
                out_str *= (long long)f[q];
            }
            m.extent[m.nlv] = O; m.in_stride[m.nlv] = m.out_stride[m.nlv] = W * S; ++m.nlv;
            if (m.nlv > MAX_LEVELS) return fail(__FILE__, __LINE__, "fft: internal error, too many levels");
            if (int rc = emit_pass((size_t)nm, inverse, m, O * (W / nm) * S, false, res, a, b, res)) return rc;
        }
        return 0;
    }

    /// Bluestein on `rows` contiguous rows of length n (a prime factor above 13):
    /// X[k] = c[k] * sum_j (x[j] c[j]) conj(c)[k - j], c[t] = exp(-+ pi i t^2 / n)
    int emit_bluestein(size_t n, long long rows, bool inverse, int cur, int a, int b, int &res) {
        if (n >= (size_t(1) << 31)) return fail(__FILE__, __LINE__, "fft: length too large for the chirp-z path");
        // convolution length: the radix-8 stages of a power of two are much cheaper per element than stages of 3s and 5s
        size_t m = best_size(2 * n - 1), m2 = 1;
        while (m2 < 2 * n - 1) m2 *= 2;
        if (m2 <= m + m / 2) m = m2;
        std::vector<cx<T>> chirp(n), bseq(m, cx<T>{T(0), T(0)});
        const long double pi = 3.141592653589793238462643383279502884L;
        for (size_t t = 0; t < n; ++t) {
            const size_t q = (size_t)(((unsigned long long)t * t) % (2ull * n));     // t^2 mod 2n: the angle is exact (n < 2^32)
            const long double ang = pi * (long double)q / (long double)n;
            const T cs = (T)std::cos(ang), sn = (T)std::sin(ang);
            chirp[t] = {cs, inverse ? sn : -sn};
            const cx<T> bt = {cs, inverse ? -sn : sn};                                // conj(c[t])
            bseq[t] = bt;
            if (t) bseq[m - t] = bt;
        }
        int chirp_id, bhat_id, ba, bb;
        if (int rc = upload(chirp, chirp_id)) return rc;
        if (int rc = alloc((size_t)rows * m * sizeof(cx<T>), ba)) return rc;
        bb = ba;
        const bool fused = m <= (size_t)lds_elems<T>() || (m <= 2 * (size_t)lds_elems<T>() && (m & (m - 1)) == 0 && !env(ENV_VEXHIP_FFT_NO_SINGLE));
        if (!fused)                                // multi-pass convolution transforms alternate between two buffers
            if (int rc = alloc((size_t)rows * m * sizeof(cx<T>), bb)) return rc;
        {   // bhat = FFT_m(b), computed once with a plan of its own
            plan_t<T> sub; sub.dev = dev; sub.total = m;
            int in_id, r, w1;
            if (int rc = upload(bseq, in_id)) return rc;
            if (int rc = alloc(m * sizeof(cx<T>), bhat_id)) return rc;
            if (int rc = sub.alloc(m * sizeof(cx<T>), w1)) return rc;
            if (int rc = sub.emit_dim(m, 1, 1, false, B_IN, B_OUT, B_WORK, r)) return rc;
            void *dst = owned[bhat_id - B_FIRST_OWNED];
            if (int rc = sub.run(nullptr, owned[in_id - B_FIRST_OWNED], dst, sub.owned[w1 - B_FIRST_OWNED])) return rc;
            if (r == B_WORK) VEXHIP_TRY(hipMemcpyAsync(dst, sub.owned[w1 - B_FIRST_OWNED], m * sizeof(cx<T>), hipMemcpyDeviceToDevice, nullptr));
            VEXHIP_TRY(hipDeviceSynchronize());
        }
        if (m <= (size_t)lds_elems<T>() || (m <= 2 * (size_t)lds_elems<T>() && (m & (m - 1)) == 0 && !env(ENV_VEXHIP_FFT_NO_SINGLE))) {
            // two row passes carry all the pointwise work: chirp + zero padding on the way in and the product with
            // FFT(b) on the way out of the forward transform; chirp, 1/m and the cut to n on the way out of the inverse
            const int final_dst = writable(cur) ? cur : a;
            line_map f{};
            f.in_es = f.out_es = 1; f.nlv = 1; f.extent[0] = rows; f.in_stride[0] = (long long)n; f.out_stride[0] = (long long)m;
            f.pre = owned[chirp_id - B_FIRST_OWNED]; f.pre_n = (int)n;
            f.post = owned[bhat_id - B_FIRST_OWNED]; f.post_n = 0; f.post_scale = 1.0;
            int c;
            if (int rc = emit_pass(m, false, f, rows, false, cur, ba, bb, c, ba)) return rc;
            line_map g{};
            g.in_es = g.out_es = 1; g.nlv = 1; g.extent[0] = rows; g.in_stride[0] = (long long)m; g.out_stride[0] = (long long)n;
            g.post = owned[chirp_id - B_FIRST_OWNED]; g.post_n = (int)n; g.post_scale = 1.0 / (double)m;
            return emit_pass(m, true, g, rows, false, c, ba, bb, res, final_dst);
        }
        step s1; s1.kind = step::BLUE_IN; s1.src = cur; s1.dst = ba; s1.bn = (long long)n; s1.bm = (long long)m; s1.rows = rows; s1.chirp = chirp_id;
        steps.push_back(s1);
        int c = ba;
        if (int rc = emit_dim(m, 1, (size_t)rows, false, c, ba, bb, c)) return rc;
        step s2; s2.kind = step::BLUE_MUL; s2.src = c; s2.dst = c; s2.bm = (long long)m; s2.rows = rows; s2.bhat = bhat_id;
        steps.push_back(s2);
        if (int rc = emit_dim(m, 1, (size_t)rows, true, c, ba, bb, c)) return rc;
        step s3; s3.kind = step::BLUE_OUT; s3.src = c; s3.dst = writable(cur) ? cur : a; s3.bn = (long long)n; s3.bm = (long long)m; s3.rows = rows; s3.chirp = chirp_id;
        steps.push_back(s3);
        res = s3.dst;
        return 0;
    }

    /// n-D: every dimension in place where it is (last first), `none` dimensions are batches.
    int build(const std::vector<size_t> &sizes, const std::vector<int> &dirs, int a, int b, int &res) {
        steps.clear();
        int cur = B_IN;
        size_t inner = 1;
        for (size_t j = sizes.size(); j-- > 0;) {
            const size_t w = sizes[j];
            if (dirs[j] != VEXHIP_FFT_NONE && w > 1)
                if (int rc = emit_dim(w, inner, total / (w * inner), dirs[j] == VEXHIP_FFT_INVERSE, cur, a, b, cur)) return rc;
            inner *= w;
        }
        res = cur;
        return 0;
    }

    void *resolve(int id, const void *in, void *out, void *work) const {
        if (id == B_IN) return const_cast<void *>(in);
        if (id == B_OUT) return out;
        if (id == B_WORK) return work;
        return owned[id - B_FIRST_OWNED];
    }

    /// Executes the steps.
    int run(hipStream_t stream, const void *in, void *out, void *work) const {
        for (const step &s : steps) {
            const cx<T> *src = static_cast<const cx<T> *>(resolve(s.src, in, out, work));
            cx<T> *dst = static_cast<cx<T> *>(resolve(s.dst, in, out, work));
            switch (s.kind) {
                case step::LINES: {
                    const long long grid = (s.lines + s.lines_per_wg - 1) / s.lines_per_wg;
                    size_t lds = (s.ept ? 1 : 2) * (size_t)s.lines_per_wg * s.pitch * sizeof(cx<T>);
                    if (const char *e = env(ENV_VEXHIP_FFT_EXTRA_LDS)) lds += (size_t)atoi(e);       // occupancy experiments
                    const bool pow2 = (s.n & (s.n - 1)) == 0;
                    const bool fused = s.map.pre || s.map.pre_n || s.map.post || s.map.post_n;
                    auto kernel = fused ? (pow2 ? (s.ept == 8 ? &fft_lines_kernel<T, false, true, 8> : s.ept == 16 ? &fft_lines_kernel<T, false, true, 16>
                                                                     : s.ept == 32 ? &fft_lines_kernel<T, false, true, (sizeof(T) == 4 ? 32 : 16)>
                                                                     : &fft_lines_kernel<T, false, true, 0>)
                                                : &fft_lines_kernel<T, true, true, 0>)
                                : s.ept == 8 ? (pow2 ? &fft_lines_kernel<T, false, false, 8> : &fft_lines_kernel<T, true, false, 8>)
                                : s.ept == 16 ? (pow2 ? &fft_lines_kernel<T, false, false, 16> : &fft_lines_kernel<T, true, false, 16>)
                                : s.ept == 32 ? &fft_lines_kernel<T, false, false, (sizeof(T) == 4 ? 32 : 16)>
                                : (pow2 ? &fft_lines_kernel<T, false, false, 0> : &fft_lines_kernel<T, true, false, 0>);
                    if (lds > 48 * 1024)          // per device, and cheap: raise the dynamic LDS limit for the padded tiles
                        VEXHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (lds_elems<T>() + MAX_LINES) * (int)sizeof(cx<T>)));
                    kernel<<<dim3((unsigned)grid), dim3(s.threads), lds, stream>>>(src, dst,
                            static_cast<const cx<T> *>(owned[s.table - B_FIRST_OWNED]), s.n, s.lines, s.lines_per_wg, s.pitch, s.st, s.inverse, s.map);
                    break;
                }
                case step::TRANSPOSE: {
                    const long long tr = (s.R + 31) / 32, tc = (s.C + 31) / 32;
                    fft_transpose_kernel<T><<<dim3((unsigned)(s.batch * tr * tc)), dim3(FB), 0, stream>>>(src, dst, s.R, s.C, tc, tr * tc);
                    break;
                }
                case step::BLUE_IN: {
                    const long long tot = s.rows * s.bm;
                    bluestein_in_kernel<T><<<dim3((unsigned)std::min<long long>((tot + FB - 1) / FB, 1 << 20)), dim3(FB), 0, stream>>>(
                            src, static_cast<const cx<T> *>(owned[s.chirp - B_FIRST_OWNED]), dst, s.bn, s.bm, tot);
                    break;
                }
                case step::BLUE_MUL: {
                    const long long tot = s.rows * s.bm;
                    bluestein_mul_kernel<T><<<dim3((unsigned)std::min<long long>((tot + FB - 1) / FB, 1 << 20)), dim3(FB), 0, stream>>>(
                            dst, static_cast<const cx<T> *>(owned[s.bhat - B_FIRST_OWNED]), s.bm, tot);
                    break;
                }
                case step::BLUE_OUT: {
                    const long long tot = s.rows * s.bn;
                    bluestein_out_kernel<T><<<dim3((unsigned)std::min<long long>((tot + FB - 1) / FB, 1 << 20)), dim3(FB), 0, stream>>>(
                            src, static_cast<const cx<T> *>(owned[s.chirp - B_FIRST_OWNED]), dst, s.bn, s.bm, tot, (T)(1.0 / (double)s.bm));
                    break;
                }
                case step::COPY:
                    VEXHIP_TRY(hipMemcpyAsync(dst, src, (size_t)s.elems * sizeof(cx<T>), hipMemcpyDeviceToDevice, stream));
                    break;
            }
            VEXHIP_LAUNCH_CHECK();
        }
        return 0;
    }
};

struct any_plan {
    int dtype;
    int work_id = -1;
    std::unique_ptr<plan_t<float>> f;
    std::unique_ptr<plan_t<double>> d;
};

template <typename T>
int make_plan(int dev, const std::vector<size_t> &sizes, const std::vector<int> &dirs, std::unique_ptr<plan_t<T>> &out, int &work_id) {
    std::unique_ptr<plan_t<T>> p(new plan_t<T>);
    p->dev = dev;
    p->total = 1;
    for (size_t s : sizes) p->total *= s;
    int res;
    // out-of-place passes alternate between two buffers; which of them receives the first write decides where
    // the result ends: build with (OUT, WORK) and, if the result lands in WORK, again with the roles swapped
    if (int rc = p->build(sizes, dirs, B_OUT, B_WORK, res)) return rc;
    if (res == B_WORK) {
        std::unique_ptr<plan_t<T>> q(new plan_t<T>);
        q->dev = dev; q->total = p->total;
        p.reset();                                  // release the first attempt's buffers before building the second
        if (int rc = q->build(sizes, dirs, B_WORK, B_OUT, res)) return rc;
        p = std::move(q);
    }
    if (res != B_OUT) {                             // no pass at all (res == B_IN), or a parity the swap did not fix
        step c; c.kind = step::COPY; c.src = res; c.dst = B_OUT; c.elems = (long long)p->total;
        p->steps.push_back(c);
    }
    // the work buffer exists only if some step touches it (a single in-place row pass needs none)
    work_id = -1;
    for (const step &s : p->steps) if (s.src == B_WORK || s.dst == B_WORK) { if (int rc = p->alloc(p->total * sizeof(cx<T>), work_id)) return rc; break; }
    out = std::move(p);
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

size_t vexhip_fft_best_size(size_t n) { return best_size(n); }

int vexhip_fft_plan_create(int dev, int dtype, int ndim, const size_t *sizes, const int *dirs, void **plan) {
    VEXHIP_REQUIRE(plan && sizes && dirs && ndim >= 1, "fft: bad arguments");
    VEXHIP_REQUIRE(dtype == VEXHIP_F32 || dtype == VEXHIP_F64, "fft: only float and double data are supported");
    VEXHIP_SET_DEVICE(dev);
    std::vector<size_t> sz(sizes, sizes + ndim);
    std::vector<int> dr(dirs, dirs + ndim);
    for (size_t s : sz) VEXHIP_REQUIRE(s >= 1, "fft: empty dimension");
    std::unique_ptr<any_plan> p(new any_plan);
    p->dtype = dtype;
    int rc = dtype == VEXHIP_F32 ? make_plan<float>(dev, sz, dr, p->f, p->work_id) : make_plan<double>(dev, sz, dr, p->d, p->work_id);
    if (rc) return rc;
    *plan = p.release();
    return 0;
}

int vexhip_fft_plan_destroy(void *plan) {
    delete static_cast<any_plan *>(plan);
    return 0;
}

int vexhip_fft_plan_steps(void *plan, int *rows_passes, int *transposes, int *others) {
    VEXHIP_REQUIRE(plan, "fft: null plan");
    any_plan *p = static_cast<any_plan *>(plan);
    int r = 0, t = 0, o = 0;
    auto count = [&](const std::vector<step> &steps) {
        for (const step &s : steps) { if (s.kind == step::LINES) ++r; else if (s.kind == step::TRANSPOSE) ++t; else ++o; }
    };
    if (p->f) count(p->f->steps); else count(p->d->steps);
    if (rows_passes) *rows_passes = r;
    if (transposes) *transposes = t;
    if (others) *others = o;
    return 0;
}

int vexhip_fft_exec(void *plan, void *stream, const void *in, void *out) {
    VEXHIP_REQUIRE(plan && in && out, "fft: null argument");
    VEXHIP_REQUIRE(in != out, "fft: the transform is out of place");
    any_plan *p = static_cast<any_plan *>(plan);
    if (p->f) {
        VEXHIP_SET_DEVICE(p->f->dev);
        return p->f->run(as_stream(stream), in, out, p->work_id >= 0 ? p->f->owned[p->work_id - B_FIRST_OWNED] : nullptr);
    }
    VEXHIP_SET_DEVICE(p->d->dev);
    return p->d->run(as_stream(stream), in, out, p->work_id >= 0 ? p->d->owned[p->work_id - B_FIRST_OWNED] : nullptr);
}

} // extern "C"

VEXHIP_WARM_TU(fft)
