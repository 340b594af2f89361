// vex::FFT on gfx950 (reference: vexcl/fft.hpp, vexcl/fft/plan.hpp:214-330, fft/kernels.hpp -- a generator of
// global-memory radix passes, one launch per radix, plus a transpose per dimension).
//
// MI355X design.  A transform is HBM-bound, so the plan is built around ONE kernel that moves every element
// once: `fft_rows_kernel` brings a batch of contiguous rows into LDS with 16-byte coalesced loads, runs ALL
// radix stages of the row there (Stockham autosort between two LDS buffers, 256 lanes, radix 8/4/2 for the
// power-of-two part and 3/5/7/11/13 for the rest, twiddles from a per-length table that stays in L2) and writes
// the rows back -- 1 read + 1 write of the data per row pass, whatever the number of stages (the reference
// launches one global-memory pass per radix: 3-4 for n = 4096).  A row of up to 2048 (fp64) / 4096 (fp32)
// complex elements fits (2 x 32 KiB of LDS, two workgroups per CU).  On top of that kernel:
//   * longer rows: four-step decomposition n = n1 * n2 (transpose, n2 x FFT(n1) with the W_n^(j2 k1) twiddle fused
//     into the store, transpose, n1 x FFT(n2) -- recursively --, transpose);
//   * lengths with a prime factor above 13: Bluestein's chirp-z over a 2^a 3^b 5^c 7^d convolution length;
//   * n-D transforms and batches: the reference's scheme of rotating the dimensions with one tiled LDS transpose
//     per transformed dimension (plan.hpp:243-256), rows always contiguous; `none` dimensions on the left cost nothing.
// The plan (factorizations, twiddle / chirp tables, work buffers) is native C++ behind four C entry points.
#include "common.hpp"

#include <algorithm>
#include <cmath>
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
    static __device__ __forceinline__ void run(cx<T> (&v)[2], const cx<T> *) {
        const cx<T> a = v[0], b = v[1];
        v[0] = a + b; v[1] = a - b;
    }
};
template <typename T> __device__ __forceinline__ void dft4(cx<T> &a, cx<T> &b, cx<T> &c, cx<T> &d, cx<T> w4) {
    const cx<T> t0 = a + c, t1 = a - c, t2 = b + d, t3 = (b - d) * w4;
    a = t0 + t2; b = t1 + t3; c = t0 - t2; d = t1 - t3;
}
template <typename T> struct dft<T, 4> {
    static __device__ __forceinline__ void run(cx<T> (&v)[4], const cx<T> *root) { dft4(v[0], v[1], v[2], v[3], root[1]); }
};
template <typename T> struct dft<T, 8> {
    static __device__ __forceinline__ void run(cx<T> (&v)[8], const cx<T> *root) {
        dft4(v[0], v[2], v[4], v[6], root[2]);          // even samples -> E[0..3] in v[0], v[2], v[4], v[6]
        dft4(v[1], v[3], v[5], v[7], root[2]);          // odd samples  -> O[0..3]
        const cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        const cx<T> o0 = v[1], o1 = v[3] * root[1], o2 = v[5] * root[2], o3 = v[7] * root[3];
        v[0] = e0 + o0; v[4] = e0 - o0;
        v[1] = e1 + o1; v[5] = e1 - o1;
        v[2] = e2 + o2; v[6] = e2 - o2;
        v[3] = e3 + o3; v[7] = e3 - o3;
    }
};
/// Odd primes: the definition, unrolled (the exponents fold to constants).
template <typename T, int R> struct dft {
    static __device__ __forceinline__ void run(cx<T> (&v)[R], const cx<T> *root) {
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

/// One Stockham stage over the rows held in LDS: `src` -> `dst`, sub-transform length p -> p * R.
template <typename T, int R>
__device__ __forceinline__ void stage(const cx<T> *__restrict__ src, cx<T> *__restrict__ dst,
        const cx<T> *__restrict__ tw, int n, int p, int nrows, bool inverse)
{
    const int nb = n / R;                        // butterflies per row
    const int tstride = n / (p * R);             // W_(pR)^k = tw[k * tstride]
    cx<T> root[R];
#pragma unroll
    for (int t = 0; t < R; ++t) { root[t] = tw[(size_t)t * nb]; if (inverse) root[t].y = -root[t].y; }
    const int total = nrows * nb;
    for (int b = threadIdx.x; b < total; b += FB) {
        const int row = b / nb, j = b - row * nb;
        const int k = j % p;
        const cx<T> *in = src + (size_t)row * n + j;
        cx<T> v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = in[q * nb];
        if (p > 1) {
            cx<T> w1 = tw[(size_t)k * tstride];
            if (inverse) w1.y = -w1.y;
            cx<T> w = w1;
#pragma unroll
            for (int q = 1; q < R; ++q) { v[q] = v[q] * w; if (q + 1 < R) w = w * w1; }
        }
        dft<T, R>::run(v, root);
        cx<T> *out = dst + (size_t)row * n + (size_t)(j - k) * R + k;
#pragma unroll
        for (int s = 0; s < R; ++s) out[s * p] = v[s];
    }
}

/// A batch of contiguous rows, each transformed completely in LDS.
/// Optional post-multiplication of element (row, k) by W_N^((row % tw_rows) * k) (four-step twiddle), tw_rows = 0: none.
template <typename T>
__global__ __launch_bounds__(FB)
void fft_rows_kernel(const cx<T> *__restrict__ in, cx<T> *__restrict__ out, const cx<T> *__restrict__ tw,
        int n, long long rows, int rows_per_wg, stage_list st, int inverse, long long tw_N, long long tw_rows)
{
    extern __shared__ __attribute__((aligned(16))) char fft_smem[];
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    const int nrows = (int)min((long long)rows_per_wg, rows - r0);
    const int E = nrows * n;
    cx<T> *A = reinterpret_cast<cx<T> *>(fft_smem);
    cx<T> *B = A + (size_t)rows_per_wg * n;

    const cx<T> *gin = in + r0 * n;
    for (int e = threadIdx.x; e < E; e += FB) A[e] = gin[e];
    __syncthreads();

    int p = 1;
    for (int s = 0; s < st.count; ++s) {
        const int R = st.radix[s];
        switch (R) {
            case 2:  stage<T, 2>(A, B, tw, n, p, nrows, inverse); break;
            case 3:  stage<T, 3>(A, B, tw, n, p, nrows, inverse); break;
            case 4:  stage<T, 4>(A, B, tw, n, p, nrows, inverse); break;
            case 5:  stage<T, 5>(A, B, tw, n, p, nrows, inverse); break;
            case 7:  stage<T, 7>(A, B, tw, n, p, nrows, inverse); break;
            case 8:  stage<T, 8>(A, B, tw, n, p, nrows, inverse); break;
            case 11: stage<T, 11>(A, B, tw, n, p, nrows, inverse); break;
            default: stage<T, 13>(A, B, tw, n, p, nrows, inverse); break;
        }
        p *= R;
        cx<T> *t = A; A = B; B = t;
        __syncthreads();
    }

    cx<T> *gout = out + r0 * n;
    if (tw_rows == 0) {
        for (int e = threadIdx.x; e < E; e += FB) gout[e] = A[e];
    } else {
        for (int e = threadIdx.x; e < E; e += FB) {
            const int row = e / n, k = e - row * n;
            const long long j2 = (r0 + row) % tw_rows;
            const double frac = 2.0 * (double)(j2 * k) / (double)tw_N;        // j2 * k < N: no reduction needed
            double sn, cs;
            sincospi(frac, &sn, &cs);
            cx<T> w = {(T)cs, (T)(inverse ? sn : -sn)};
            gout[e] = A[e] * w;
        }
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
    enum kind_t { ROWS, TRANSPOSE, BLUE_IN, BLUE_MUL, BLUE_OUT, COPY } kind;
    int src, dst;
    // ROWS
    int n = 0; long long rows = 0; stage_list st{}; int inverse = 0; int table = -1; long long tw_N = 0, tw_rows = 0;
    // TRANSPOSE: [batch][R][C] -> [batch][C][R]
    long long batch = 0, R = 0, C = 0;
    // BLUESTEIN
    long long bn = 0, bm = 0; int chirp = -1, bhat = -1;
    long long elems = 0;        // COPY
};

template <typename T>
struct plan_t {
    int dev = 0;
    size_t total = 0;
    std::vector<step> steps;
    std::vector<void *> owned;                 // device buffers of the plan: work buffer, Bluestein buffers, tables
    std::vector<size_t> owned_bytes;
    std::vector<std::pair<size_t, int>> tw_tables;     // (n, owned index) twiddle tables already built

    ~plan_t() { (void)hipSetDevice(dev); for (void *p : owned) if (p) (void)hipFree(p); }

    int alloc(size_t bytes, int &id) {
        void *p = nullptr;
        VEXHIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        owned.push_back(p); owned_bytes.push_back(bytes);
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

    /// Transform `rows` contiguous rows of length n held in `cur`; (a, b) are two writable buffers of at least
    /// rows * n elements that the steps may alternate between.  Returns the buffer holding the result in `res`.
    int emit_rows(size_t n, long long rows, bool inverse, int cur, int a, int b, int &res, long long tw_N = 0, long long tw_rows = 0) {
        auto other = [&](int c) { return c == a ? b : a; };
        if (n == 1) {                                   // nothing to transform (a twiddle on k = 0 is 1 as well)
            res = cur;
            return 0;
        }
        stage_list st;
        if (n <= (size_t)lds_elems<T>() && factor(n, st)) {
            step s; s.kind = step::ROWS; s.src = cur; s.dst = writable(cur) ? cur : a;
            s.n = (int)n; s.rows = rows; s.st = st; s.inverse = inverse; s.tw_N = tw_N; s.tw_rows = tw_rows;
            if (int rc = twiddles(n, s.table)) return rc;
            steps.push_back(s);
            res = s.dst;
            return 0;
        }
        if (smooth(n)) {
            if (tw_rows) return fail(__FILE__, __LINE__, "fft: nested four-step twiddle is not supported");    // n1 always fits the kernel
            // four-step: n = n1 * n2, n1 the largest divisor that fits the row kernel
            size_t n1 = 1;
            for (size_t d = std::min<size_t>(n, lds_elems<T>()); d >= 2; --d) if (n % d == 0) { n1 = d; break; }
            const size_t n2 = n / n1;
            // x[j1 * n2 + j2] viewed as [rows][n1][n2] -> [rows][n2][n1]
            step t1; t1.kind = step::TRANSPOSE; t1.src = cur; t1.dst = writable(cur) ? other(cur) : a; t1.batch = rows; t1.R = (long long)n1; t1.C = (long long)n2;
            steps.push_back(t1);
            int c = t1.dst;
            // n2 transforms of length n1 per row, each multiplied by W_n^(j2 k1)
            if (int rc = emit_rows(n1, rows * (long long)n2, inverse, c, a, b, c, (long long)n, (long long)n2)) return rc;
            // [rows][n2][n1] -> [rows][n1][n2]
            step t2; t2.kind = step::TRANSPOSE; t2.src = c; t2.dst = other(c); t2.batch = rows; t2.R = (long long)n2; t2.C = (long long)n1;
            steps.push_back(t2);
            c = t2.dst;
            // n1 transforms of length n2
            if (int rc = emit_rows(n2, rows * (long long)n1, inverse, c, a, b, c)) return rc;
            // Z[k1][k2] = X[k1 + n1 k2]: [rows][n1][n2] -> [rows][n2][n1]
            step t3; t3.kind = step::TRANSPOSE; t3.src = c; t3.dst = other(c); t3.batch = rows; t3.R = (long long)n1; t3.C = (long long)n2;
            steps.push_back(t3);
            c = t3.dst;
            res = c;
            return 0;
        }
        // Bluestein: a prime factor above 13.  X[k] = c[k] * sum_j (x[j] c[j]) conj(c)[k - j], c[t] = exp(-+ pi i t^2 / n)
        if (tw_rows) return fail(__FILE__, __LINE__, "fft: Bluestein inside a four-step pass is not supported");
        if (n >= (size_t(1) << 31)) return fail(__FILE__, __LINE__, "fft: length too large for the chirp-z path");
        const size_t m = best_size(2 * n - 1);
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
        if (int rc = alloc((size_t)rows * m * sizeof(cx<T>), bb)) return rc;
        {   // bhat = FFT_m(b), computed once with a plan of its own
            plan_t<T> sub; sub.dev = dev; sub.total = m;
            int in_id, r;
            if (int rc = upload(bseq, in_id)) return rc;
            if (int rc = alloc(m * sizeof(cx<T>), bhat_id)) return rc;
            int w1;
            if (int rc = sub.alloc(m * sizeof(cx<T>), w1)) return rc;
            if (int rc = sub.emit_rows(m, 1, false, B_IN, B_OUT, B_WORK, r)) return rc;
            void *dst = owned[bhat_id - B_FIRST_OWNED];
            if (int rc = sub.run(nullptr, owned[in_id - B_FIRST_OWNED], dst, sub.owned[w1 - B_FIRST_OWNED], r)) return rc;
            if (r == B_WORK) VEXHIP_TRY(hipMemcpyAsync(dst, sub.owned[w1 - B_FIRST_OWNED], m * sizeof(cx<T>), hipMemcpyDeviceToDevice, nullptr));
            VEXHIP_TRY(hipDeviceSynchronize());
        }
        step s1; s1.kind = step::BLUE_IN; s1.src = cur; s1.dst = ba; s1.bn = (long long)n; s1.bm = (long long)m; s1.rows = rows; s1.chirp = chirp_id;
        steps.push_back(s1);
        int c = ba;
        if (int rc = emit_rows(m, rows, false, c, ba, bb, c)) return rc;
        step s2; s2.kind = step::BLUE_MUL; s2.src = c; s2.dst = c; s2.bm = (long long)m; s2.rows = rows; s2.bhat = bhat_id;
        steps.push_back(s2);
        if (int rc = emit_rows(m, rows, true, c, ba, bb, c)) return rc;
        step s3; s3.kind = step::BLUE_OUT; s3.src = c; s3.dst = writable(cur) ? cur : a; s3.bn = (long long)n; s3.bm = (long long)m; s3.rows = rows; s3.chirp = chirp_id;
        steps.push_back(s3);
        res = s3.dst;
        return 0;
    }

    /// n-D: rotate the dimensions, last first (plan.hpp:243-256 of the reference); `none` dimensions are skipped and,
    /// once no transformed dimension remains on the left, the rotation is undone with one transpose.
    int build(const std::vector<size_t> &sizes, const std::vector<int> &dirs, int a, int b, int &res) {
        steps.clear();
        auto other = [&](int c) { return c == a ? b : a; };
        int cur = B_IN;
        int leftmost = -1;
        for (size_t j = 0; j < sizes.size(); ++j) if (dirs[j] != VEXHIP_FFT_NONE && sizes[j] > 1) { leftmost = (int)j; break; }
        size_t P = 1;                               // product of the dimensions already rotated to the front
        if (leftmost >= 0) {
            for (int j = (int)sizes.size() - 1; j >= leftmost; --j) {
                const size_t w = sizes[j], h = total / w;
                if (dirs[j] != VEXHIP_FFT_NONE && w > 1)
                    if (int rc = emit_rows(w, (long long)h, dirs[j] == VEXHIP_FFT_INVERSE, cur, a, b, cur)) return rc;
                if (j > leftmost && w > 1 && h > 1) {
                    step t; t.kind = step::TRANSPOSE; t.src = cur; t.dst = writable(cur) ? other(cur) : a; t.batch = 1; t.R = (long long)h; t.C = (long long)w;
                    steps.push_back(t);
                    cur = t.dst;
                    P *= w;
                }
            }
            if (P > 1) {                                // [P][Q] -> [Q][P]
                step t; t.kind = step::TRANSPOSE; t.src = cur; t.dst = writable(cur) ? other(cur) : a; t.batch = 1; t.R = (long long)P; t.C = (long long)(total / P);
                steps.push_back(t);
                cur = t.dst;
            }
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

    /// Executes the steps; the result is left in buffer `res` (the caller knows which one that is).
    int run(hipStream_t stream, const void *in, void *out, void *work, int /*res*/) const {
        for (const step &s : steps) {
            const cx<T> *src = static_cast<const cx<T> *>(resolve(s.src, in, out, work));
            cx<T> *dst = static_cast<cx<T> *>(resolve(s.dst, in, out, work));
            switch (s.kind) {
                case step::ROWS: {
                    // rows per workgroup: as many as fit the LDS buffers, but keep >= ~4 workgroups per CU when the batch allows
                    long long rpw = std::max<long long>(1, lds_elems<T>() / s.n);
                    rpw = std::max<long long>(1, std::min(rpw, (s.rows + 1023) / 1024));
                    const long long grid = (s.rows + rpw - 1) / rpw;
                    const size_t lds = 2 * (size_t)rpw * s.n * sizeof(cx<T>);
                    fft_rows_kernel<T><<<dim3((unsigned)grid), dim3(FB), lds, stream>>>(src, dst,
                            static_cast<const cx<T> *>(owned[s.table - B_FIRST_OWNED]), s.n, s.rows, (int)rpw, s.st, s.inverse, s.tw_N, s.tw_rows);
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
    if (int rc = p->alloc(p->total * sizeof(cx<T>), work_id)) return rc;
    int res;
    // the steps alternate between two buffers; which of them receives the first write decides where the result
    // ends: build with (OUT, WORK) and, if the result lands in WORK, again with the roles swapped
    if (int rc = p->build(sizes, dirs, B_OUT, work_id, res)) return rc;
    if (res == work_id) {
        std::unique_ptr<plan_t<T>> q(new plan_t<T>);
        q->dev = dev; q->total = p->total;
        int w2;
        if (int rc = q->alloc(q->total * sizeof(cx<T>), w2)) return rc;
        p.reset();                                  // release the first attempt's tables before building the second
        if (int rc = q->build(sizes, dirs, w2, B_OUT, res)) return rc;
        p = std::move(q);
        work_id = w2;
    }
    if (res != B_OUT) {                             // no step at all (res == B_IN), or a parity the swap did not fix
        step c; c.kind = step::COPY; c.src = res; c.dst = B_OUT; c.elems = (long long)p->total;
        p->steps.push_back(c);
    }
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
        for (const step &s : steps) { if (s.kind == step::ROWS) ++r; else if (s.kind == step::TRANSPOSE) ++t; else ++o; }
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
        return p->f->run(as_stream(stream), in, out, p->f->owned[p->work_id - B_FIRST_OWNED], B_OUT);
    }
    VEXHIP_SET_DEVICE(p->d->dev);
    return p->d->run(as_stream(stream), in, out, p->d->owned[p->work_id - B_FIRST_OWNED], B_OUT);
}

} // extern "C"
