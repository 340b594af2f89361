// vex::Reductor on gfx950 (reductor.hpp:302-439).  Both stages run on the
// device: stage 1 = per-lane accumulation over a grid-stride range of 16-byte
// loads (reductor.hpp:511-564), wave-64 shuffle fold, one LDS hop across the
// workgroup's waves, one partial per workgroup; stage 2 = one workgroup folds
// the partials (the reference does this fold on the host, reductor.hpp:412-436)
// so that a multi-GPU combine needs one scalar per GPU.
#include "common.hpp"

#include <algorithm>
#include <limits>

namespace vexhip {
namespace {

constexpr int RBLOCK = 256;
constexpr int RWAVES = RBLOCK / kWave;

template <typename T> struct lim {
    __host__ __device__ static T lowest() { return std::numeric_limits<T>::lowest(); }
    __host__ __device__ static T highest() { return std::numeric_limits<T>::max(); }
};

// ---- accumulators ----------------------------------------------------------
template <typename T> struct AccSum {
    T s;
    __device__ void init() { s = T(0); }
    __device__ void add(T x) { s += x; }
    __device__ void merge(const AccSum &o) { s += o.s; }
    __device__ void shfl_merge(int off) { s += __shfl_down(s, off, 64); }
    static constexpr int NOUT = 1;
    __device__ void store(T *out) const { out[0] = s; }
    __device__ void load(const T *in) { s = in[0]; }
};
// Per-lane Kahan recurrence exactly as reductor.hpp:537-564; lanes are then
// folded with plain additions (the reference tree-reduces mySum the same way).
template <typename T> struct AccKahan {
    T s, c;
    __device__ void init() { s = T(0); c = T(0); }
    __device__ void add(T x) { T y = x - c; T t = s + y; c = (t - s) - y; s = t; }
    __device__ void merge(const AccKahan &o) { add(o.s); }
    __device__ void shfl_merge(int off) { T o = __shfl_down(s, off, 64); add(o); }
    static constexpr int NOUT = 1;
    __device__ void store(T *out) const { out[0] = s; }
    __device__ void load(const T *in) { s = in[0]; c = T(0); }
};
template <typename T> struct AccMin {
    T s;
    __device__ void init() { s = lim<T>::highest(); }
    __device__ void add(T x) { s = x < s ? x : s; }
    __device__ void merge(const AccMin &o) { add(o.s); }
    __device__ void shfl_merge(int off) { add(__shfl_down(s, off, 64)); }
    static constexpr int NOUT = 1;
    __device__ void store(T *out) const { out[0] = s; }
    __device__ void load(const T *in) { s = in[0]; }
};
template <typename T> struct AccMax {
    T s;
    __device__ void init() { s = lim<T>::lowest(); }
    __device__ void add(T x) { s = x > s ? x : s; }
    __device__ void merge(const AccMax &o) { add(o.s); }
    __device__ void shfl_merge(int off) { add(__shfl_down(s, off, 64)); }
    static constexpr int NOUT = 1;
    __device__ void store(T *out) const { out[0] = s; }
    __device__ void load(const T *in) { s = in[0]; }
};
template <typename T> struct AccMinMax {
    T lo, hi;
    __device__ void init() { lo = lim<T>::highest(); hi = lim<T>::lowest(); }
    __device__ void add(T x) { lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
    __device__ void merge(const AccMinMax &o) { lo = o.lo < lo ? o.lo : lo; hi = o.hi > hi ? o.hi : hi; }
    __device__ void shfl_merge(int off) {
        T a = __shfl_down(lo, off, 64), b = __shfl_down(hi, off, 64);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    static constexpr int NOUT = 2;
    __device__ void store(T *out) const { out[0] = lo; out[1] = hi; }
    __device__ void load(const T *in) { lo = in[0]; hi = in[1]; }
};

template <typename Acc>
__device__ __forceinline__ void block_fold(Acc &a, Acc *s_acc) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a.shfl_merge(off);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (lane == 0) s_acc[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < RWAVES; ++w) a.merge(s_acc[w]);
    }
}

template <typename T> struct vec16 { static constexpr int N = 16 / sizeof(T); typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };

template <typename T, typename Acc, bool DOT>
__global__ __launch_bounds__(RBLOCK)
void reduce_stage1(const T *__restrict__ a, const T *__restrict__ b, long long n, T *__restrict__ partials, int vec_ok)
{
    __shared__ Acc s_acc[RWAVES];
    constexpr int VN = vec16<T>::N;
    typedef typename vec16<T>::type VT;
    Acc acc; acc.init();
    const long long tid = (long long)blockIdx.x * RBLOCK + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * RBLOCK;
    long long done = 0;
    if (vec_ok) {
        const long long nv = n / VN;
        const VT *av = reinterpret_cast<const VT *>(a);
        const VT *bv = reinterpret_cast<const VT *>(b);
        long long i = tid;
        // two independent 16-byte loads in flight per lane per trip
        for (; i + nthreads < nv; i += 2 * nthreads) {
            VT x0 = av[i], x1 = av[i + nthreads];
            if constexpr (DOT) {
                VT y0 = bv[i], y1 = bv[i + nthreads];
#pragma unroll
                for (int k = 0; k < VN; ++k) acc.add(x0[k] * y0[k]);
#pragma unroll
                for (int k = 0; k < VN; ++k) acc.add(x1[k] * y1[k]);
            } else {
#pragma unroll
                for (int k = 0; k < VN; ++k) acc.add(x0[k]);
#pragma unroll
                for (int k = 0; k < VN; ++k) acc.add(x1[k]);
            }
        }
        for (; i < nv; i += nthreads) {
            VT x0 = av[i];
            if constexpr (DOT) {
                VT y0 = bv[i];
#pragma unroll
                for (int k = 0; k < VN; ++k) acc.add(x0[k] * y0[k]);
            } else {
#pragma unroll
                for (int k = 0; k < VN; ++k) acc.add(x0[k]);
            }
        }
        done = nv * VN;
    }
    for (long long i = done + tid; i < n; i += nthreads) {
        if constexpr (DOT) acc.add(a[i] * b[i]); else acc.add(a[i]);
    }
    block_fold(acc, s_acc);
    if (threadIdx.x == 0) acc.store(partials + (long long)blockIdx.x * Acc::NOUT);
}

template <typename T, typename Acc>
__global__ __launch_bounds__(RBLOCK)
void reduce_stage2(const T *__restrict__ partials, long long nparts, T *__restrict__ out)
{
    __shared__ Acc s_acc[RWAVES];
    Acc acc; acc.init();
    for (long long i = threadIdx.x; i < nparts; i += RBLOCK) {
        Acc p; p.load(partials + i * Acc::NOUT);
        acc.merge(p);
    }
    block_fold(acc, s_acc);
    if (threadIdx.x == 0) acc.store(out);
}

inline int stage1_groups(int dev) { return info(dev).cus * 8; }   // 8 x CU, as reductor.hpp:463-471

template <typename T, typename Acc, bool DOT>
int run(int dev, hipStream_t s, const T *a, const T *b, int64_t n, T *out, T *tmp) {
    int groups = (int)std::max<int64_t>(1, std::min<int64_t>(stage1_groups(dev), (n + RBLOCK * 4 - 1) / (RBLOCK * 4)));
    int vec_ok = ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && (!DOT || (reinterpret_cast<uintptr_t>(b) & 15) == 0);
    reduce_stage1<T, Acc, DOT><<<groups, RBLOCK, 0, s>>>(a, b, n, tmp, vec_ok);
    reduce_stage2<T, Acc><<<1, RBLOCK, 0, s>>>(tmp, groups, out);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

template <typename T, bool DOT>
int run_op(int dev, hipStream_t s, int op, const T *a, const T *b, int64_t n, T *out, T *tmp) {
    switch (op) {
        case VEXHIP_SUM:       return run<T, AccSum<T>, DOT>(dev, s, a, b, n, out, tmp);
        case VEXHIP_SUM_KAHAN: return run<T, AccKahan<T>, DOT>(dev, s, a, b, n, out, tmp);
        case VEXHIP_MIN:       return run<T, AccMin<T>, DOT>(dev, s, a, b, n, out, tmp);
        case VEXHIP_MAX:       return run<T, AccMax<T>, DOT>(dev, s, a, b, n, out, tmp);
        case VEXHIP_MIN_MAX:   return run<T, AccMinMax<T>, DOT>(dev, s, a, b, n, out, tmp);
    }
    return fail(__FILE__, __LINE__, "unknown reduction op");
}

template <typename T>
int finish_op(hipStream_t s, int op, const T *p, int64_t np, T *out) {
    switch (op) {
        case VEXHIP_SUM:       reduce_stage2<T, AccSum<T>><<<1, RBLOCK, 0, s>>>(p, np, out); break;
        case VEXHIP_SUM_KAHAN: reduce_stage2<T, AccKahan<T>><<<1, RBLOCK, 0, s>>>(p, np, out); break;
        case VEXHIP_MIN:       reduce_stage2<T, AccMin<T>><<<1, RBLOCK, 0, s>>>(p, np, out); break;
        case VEXHIP_MAX:       reduce_stage2<T, AccMax<T>><<<1, RBLOCK, 0, s>>>(p, np, out); break;
        case VEXHIP_MIN_MAX:   reduce_stage2<T, AccMinMax<T>><<<1, RBLOCK, 0, s>>>(p, np, out); break;
        default: return fail(__FILE__, __LINE__, "unknown reduction op");
    }
    VEXHIP_LAUNCH_CHECK();
    return 0;
}

} // namespace
} // namespace vexhip

using namespace vexhip;

namespace {
template <typename T> int reduce_t(int dev, void *stream, int op, const void *in, int64_t n, void *out, void *tmp) {
    return run_op<T, false>(dev, as_stream(stream), op, (const T *)in, (const T *)nullptr, n, (T *)out, (T *)tmp);
}
template <typename T> int dot_t(int dev, void *stream, const void *a, const void *b, int64_t n, void *out, void *tmp) {
    return run_op<T, true>(dev, as_stream(stream), VEXHIP_SUM, (const T *)a, (const T *)b, n, (T *)out, (T *)tmp);
}
template <typename T> int finish_t(void *stream, int op, const void *p, int64_t np, void *out) {
    return finish_op<T>(as_stream(stream), op, (const T *)p, np, (T *)out);
}
}

extern "C" {

size_t vexhip_reduce_tmp_bytes(void) { return (size_t)8192 * 2 * 8; }   // >= 8*CU partials x 2 outputs x 8 B

int vexhip_reduce_num_groups(int dev, int *groups, int *block) {
    if (groups) *groups = stage1_groups(dev);
    if (block) *block = RBLOCK;
    return 0;
}

#define DISPATCH(FN, ...)                                                                         \
    switch (dtype) {                                                                              \
        case VEXHIP_F64: return FN<double>(__VA_ARGS__);                                          \
        case VEXHIP_F32: return FN<float>(__VA_ARGS__);                                           \
        case VEXHIP_I32: return FN<int>(__VA_ARGS__);                                             \
        case VEXHIP_U32: return FN<unsigned>(__VA_ARGS__);                                        \
        case VEXHIP_I64: return FN<long long>(__VA_ARGS__);                                       \
        case VEXHIP_U64: return FN<unsigned long long>(__VA_ARGS__);                              \
    }                                                                                             \
    return fail(__FILE__, __LINE__, "unknown dtype");


int vexhip_reduce(int dev, void *stream, int op, int dtype, const void *in, int64_t n, void *out, void *tmp) {
    VEXHIP_REQUIRE(out && tmp && n >= 0, "bad argument");
    VEXHIP_REQUIRE(info(dev).cus * 8 * 16 <= (int64_t)vexhip_reduce_tmp_bytes(), "tmp too small for this device");
    VEXHIP_SET_DEVICE(dev);
    DISPATCH(reduce_t, dev, stream, op, in, n, out, tmp)
}

int vexhip_reduce_dot(int dev, void *stream, int dtype, const void *a, const void *b, int64_t n, void *out, void *tmp) {
    VEXHIP_REQUIRE(out && tmp && n >= 0, "bad argument");
    VEXHIP_SET_DEVICE(dev);
    DISPATCH(dot_t, dev, stream, a, b, n, out, tmp)
}

int vexhip_reduce_finish(int dev, void *stream, int op, int dtype, const void *partials, int64_t nparts, void *out) {
    VEXHIP_REQUIRE(out && nparts >= 0, "bad argument");
    VEXHIP_SET_DEVICE(dev);
    DISPATCH(finish_t, stream, op, partials, nparts, out)
}

} // extern "C"

VEXHIP_WARM_TU(reduce)
