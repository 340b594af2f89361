/*
 * vex_oracle.c -- CPU restatement of the reference (ddemidov/vexcl) algorithms
 * on the vector-expression hot path.  TEST INFRASTRUCTURE ONLY: nothing in the
 * product (vexcl_amd/, include/, vexcl/) may link, import or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * HOW PARITY IS PINNED.  The reference holds no golden vectors (every reference
 * test recomputes its expectation on the host from time(0)-seeded inputs,
 * tests/context_setup.hpp:19-22) and its LIBRARY cannot be compiled here (Boost
 * is a hard dependency of every hot-path header and is absent; no CPU OpenCL
 * device).  What can be executed is the reference's TEST code, and that is what
 * pins this file:
 *   * SpMV (the headline path) -- directly: oracle/ref_fixture_driver.cpp runs
 *     the reference's generators (tests/random_matrix.hpp, random_vector.hpp,
 *     fixed srand) and the host loop its test asserts against
 *     (tests/spmv.cpp:28-32); tests/golden/ref_fixtures.npz holds its output and
 *     tests/test_oracle.py::test_oracle_against_reference_run_fixtures requires
 *     vxo_spmv_csr_* (plain, alpha / append, OpenMP, through hybrid ELL) to
 *     reproduce it BIT FOR BIT (round 4);
 *   * everything else -- through the reference's own test programs: oracle/_ref
 *     holds /root/reference/tests/*.cpp compiled where they lie against this
 *     repository's vexcl/ headers (oracle/build_ref.sh), their host loops and
 *     BOOST_CHECK_CLOSE assertions run against the HIP path on the GPU box
 *     (tests/test_reference_suite.py), and the HIP path is compared with this
 *     file bit for bit (integers, SpMV) or within the stated tolerance.
 * Each function below cites the reference file:line it restates.
 *
 * Built with:  gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC  (oracle/Makefile)
 * -ffp-contract=off keeps "sum += val[j]*x[col[j]]" as a rounded multiply
 * followed by a rounded add, which is what a non-FMA x86-64 build of the
 * reference's host loops (tests/spmv.cpp:28-32, examples/benchmark.cpp:447-453)
 * computes.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* 3-D Poisson matrix, examples/benchmark.cpp:364-415 (same generator in      */
/* vexcl/spmat.hpp:415-466 docs example).  Boundary rows are identity rows;   */
/* interior rows hold -h2i x3, 6*h2i, -h2i x3 in ascending column order.      */
/* ------------------------------------------------------------------------- */
int64_t vxo_poisson3d_nnz(int64_t n)
{
    int64_t N = n * n * n;
    int64_t in = (n > 2) ? (n - 2) * (n - 2) * (n - 2) : 0;
    return 7 * in + (N - in);
}

#define POISSON_BODY(PTR_T, COL_T)                                            \
    const double h2i = (double)(n - 1) * (double)(n - 1);                     \
    int64_t idx = 0, nz = 0;                                                  \
    ptr[0] = 0;                                                               \
    for (int64_t k = 0; k < n; k++)                                           \
        for (int64_t j = 0; j < n; j++)                                       \
            for (int64_t i = 0; i < n; i++, idx++) {                          \
                if (i == 0 || i == n - 1 || j == 0 || j == n - 1 ||           \
                    k == 0 || k == n - 1) {                                   \
                    col[nz] = (COL_T)idx; val[nz] = 1; nz++;                  \
                } else {                                                      \
                    col[nz] = (COL_T)(idx - n * n); val[nz] = -h2i; nz++;     \
                    col[nz] = (COL_T)(idx - n);     val[nz] = -h2i; nz++;     \
                    col[nz] = (COL_T)(idx - 1);     val[nz] = -h2i; nz++;     \
                    col[nz] = (COL_T)(idx);         val[nz] = 6 * h2i; nz++;  \
                    col[nz] = (COL_T)(idx + 1);     val[nz] = -h2i; nz++;     \
                    col[nz] = (COL_T)(idx + n);     val[nz] = -h2i; nz++;     \
                    col[nz] = (COL_T)(idx + n * n); val[nz] = -h2i; nz++;     \
                }                                                             \
                ptr[idx + 1] = (PTR_T)nz;                                     \
            }

void vxo_poisson3d_csr_i32(int64_t n, int32_t *ptr, int32_t *col, double *val)
{ POISSON_BODY(int32_t, int32_t) }

void vxo_poisson3d_csr_i64(int64_t n, int64_t *ptr, int64_t *col, double *val)
{ POISSON_BODY(int64_t, int64_t) }

void vxo_poisson3d_csr_f32_i32(int64_t n, int32_t *ptr, int32_t *col, float *val)
{ POISSON_BODY(int32_t, int32_t) }

/* ------------------------------------------------------------------------- */
/* Variable-coefficient 7-point operator (bench.py's general-matrix row; NOT  */
/* a reference generator): the Poisson pattern above with a coefficient per   */
/* face, k = 0.5 + u, u = (splitmix64 finalizer of seed + (3*lo + axis + 1)*g) */
/* >> 11 scaled by 2^-53; lo = the lower grid point of the face.  Restates     */
/* vexcl_amd/csrc/misc.hip poisson_kernel<V, true> operation for operation.    */
/* ------------------------------------------------------------------------- */
static uint64_t vxo_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static double vxo_face(uint64_t seed, int64_t lo, int axis)
{
    uint64_t h = vxo_mix64(seed + ((uint64_t)lo * 3ull + (uint64_t)axis + 1ull) * 0x9E3779B97F4A7C15ull);
    return 0.5 + (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

void vxo_diffusion3d_csr_i32(int64_t n, uint64_t seed, int32_t *ptr, int32_t *col, double *val)
{
    const double h2i = (double)(n - 1) * (double)(n - 1);
    const int64_t nn = n * n;
    int64_t idx = 0, nz = 0;
    ptr[0] = 0;
    for (int64_t k = 0; k < n; k++)
        for (int64_t j = 0; j < n; j++)
            for (int64_t i = 0; i < n; i++, idx++) {
                if (i == 0 || i == n - 1 || j == 0 || j == n - 1 || k == 0 || k == n - 1) {
                    col[nz] = (int32_t)idx; val[nz] = 1; nz++;
                } else {
                    const double k0 = vxo_face(seed, idx - nn, 2), k1 = vxo_face(seed, idx - n, 1), k2 = vxo_face(seed, idx - 1, 0);
                    const double k4 = vxo_face(seed, idx, 0), k5 = vxo_face(seed, idx, 1), k6 = vxo_face(seed, idx, 2);
                    col[nz] = (int32_t)(idx - nn); val[nz] = -h2i * k0; nz++;
                    col[nz] = (int32_t)(idx - n);  val[nz] = -h2i * k1; nz++;
                    col[nz] = (int32_t)(idx - 1);  val[nz] = -h2i * k2; nz++;
                    col[nz] = (int32_t)(idx);      val[nz] = h2i * (((((k0 + k1) + k2) + k4) + k5) + k6); nz++;
                    col[nz] = (int32_t)(idx + 1);  val[nz] = -h2i * k4; nz++;
                    col[nz] = (int32_t)(idx + n);  val[nz] = -h2i * k5; nz++;
                    col[nz] = (int32_t)(idx + nn); val[nz] = -h2i * k6; nz++;
                }
                ptr[idx + 1] = (int32_t)nz;
            }
}

/* ------------------------------------------------------------------------- */
/* CSR SpMV.  Kernel text vexcl/spmat/csr.inl:163-170:                        */
/*     sum = 0; for j in [row[i], row[i+1]): sum += val[j]*in[col[j]];        */
/*     out[i] (= | +=) scale*sum;                                             */
/* append semantics vexcl/spmat.hpp:120-121 and csr.inl:186-200.  The host    */
/* check loop of tests/spmv.cpp:28-32 is the same arithmetic with scale=1.    */
/* ------------------------------------------------------------------------- */
#define SPMV_ROW(i)                                                           \
    {                                                                         \
        VAL_T sum = 0;                                                        \
        for (int64_t j = ptr[i], e = ptr[(i) + 1]; j < e; ++j)                \
            sum += val[j] * x[col[j]];                                        \
        if (append) y[i] += alpha * sum; else y[i] = alpha * sum;             \
    }

#define VAL_T double
void vxo_spmv_csr_f64_i32(int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val,
        const double *x, double *y)
{ for (int64_t i = 0; i < n; ++i) SPMV_ROW(i) }

void vxo_spmv_csr_f64_i64(int64_t n, double alpha, int append,
        const int64_t *ptr, const int64_t *col, const double *val,
        const double *x, double *y)
{ for (int64_t i = 0; i < n; ++i) SPMV_ROW(i) }

/*
 * The reference's CPU-device execution shape: work-group
 * size 1 (backend/opencl/kernel.hpp:193-194), 8 x compute-units work-items
 * (:166-171), each taking ONE contiguous chunk of rows
 * (backend/opencl/source.hpp:255-268; the JIT/OpenMP backend is identical,
 * backend/jit/source.hpp:503-519,565-573).  Used as bench.py's cpu_baseline.
 */
int vxo_spmv_csr_f64_i32_omp(int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val,
        const double *x, double *y)
{
    int P = 1;
#ifdef _OPENMP
    P = omp_get_max_threads();
#endif
    int64_t G = 8 * (int64_t)P;
    int64_t chunk = (n + G - 1) / G;
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < G; ++g) {
        int64_t b = g * chunk, e = b + chunk; if (e > n) e = n;
        for (int64_t i = b; i < e; ++i) SPMV_ROW(i)
    }
    return P;
}
#undef VAL_T

#define VAL_T float
void vxo_spmv_csr_f32_i32(int64_t n, float alpha, int append,
        const int32_t *ptr, const int32_t *col, const float *val,
        const float *x, float *y)
{ for (int64_t i = 0; i < n; ++i) SPMV_ROW(i) }
#undef VAL_T

/*
 * bench.py's cpu_baseline in one call, with the arrays first-touched by the threads that use them.
 * Builds the n^3 Poisson matrix (generator above; offsets from the closed form of the row lengths) in
 * malloc'ed memory, every one of the 8 x threads row chunks written by the thread that will multiply
 * it (same static schedule as vxo_spmv_csr_f64_i32_omp), then times
 *   out[0] = seconds per product, OpenMP chunked csr_spmv (the reference's CPU-device shape), run for >= `seconds`
 *   out[1] = threads used
 *   out[2] = seconds per product of the single-thread loop examples/benchmark.cpp:447-453 ("C++" line of the
 *            reference harness), `single_reps` products
 *   out[3] = sum(y) of the last product (x = 0.01, as benchmark.cpp:369)      out[4] = products timed
 * Returns 0, or -1 if the arrays cannot be allocated.
 */
static int64_t vxo_interior_below(int64_t t, int64_t n) { int64_t v = t - 1; if (v < 0) v = 0; if (v > n - 2) v = n - 2; return v; }
static int64_t vxo_poisson_nnz_before(int64_t idx, int64_t n)
{
    if (n < 3) return idx;
    int64_t N = n * n * n;
    if (idx >= N) return N + 6 * (n - 2) * (n - 2) * (n - 2);
    int64_t i = idx % n, j = (idx / n) % n, k = idx / (n * n), m = n - 2;
    int64_t cnt = vxo_interior_below(k, n) * m * m;
    if (k >= 1 && k <= n - 2) { cnt += vxo_interior_below(j, n) * m; if (j >= 1 && j <= n - 2) cnt += vxo_interior_below(i, n); }
    return idx + 6 * cnt;
}

int vxo_cpu_baseline_poisson(int64_t n, double seconds, int single_reps, double *out)
{
    const int64_t N = n * n * n, nnz = vxo_poisson3d_nnz(n), nn = n * n;
    int32_t *ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)nnz);
    double *val = (double *)malloc(sizeof(double) * (size_t)nnz);
    double *x = (double *)malloc(sizeof(double) * (size_t)N), *y = (double *)malloc(sizeof(double) * (size_t)N);
    if (!ptr || !col || !val || !x || !y) { free(ptr); free(col); free(val); free(x); free(y); return -1; }
    int P = 1;
#ifdef _OPENMP
    P = omp_get_max_threads();
#endif
    const int64_t G = 8 * (int64_t)P, chunk = (N + G - 1) / G;
    const double h2i = (double)(n - 1) * (double)(n - 1);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < G; ++g) {
        int64_t b = g * chunk, e = b + chunk; if (e > N) e = N;
        for (int64_t idx = b; idx < e; ++idx) {
            int64_t nz = vxo_poisson_nnz_before(idx, n);
            ptr[idx] = (int32_t)nz;
            int64_t i = idx % n, j = (idx / n) % n, k = idx / nn;
            if (i == 0 || i == n - 1 || j == 0 || j == n - 1 || k == 0 || k == n - 1) { col[nz] = (int32_t)idx; val[nz] = 1; }
            else {
                col[nz] = (int32_t)(idx - nn); val[nz++] = -h2i; col[nz] = (int32_t)(idx - n); val[nz++] = -h2i;
                col[nz] = (int32_t)(idx - 1);  val[nz++] = -h2i; col[nz] = (int32_t)idx;       val[nz++] = 6 * h2i;
                col[nz] = (int32_t)(idx + 1);  val[nz++] = -h2i; col[nz] = (int32_t)(idx + n); val[nz++] = -h2i;
                col[nz] = (int32_t)(idx + nn); val[nz] = -h2i;
            }
            x[idx] = 0.01; y[idx] = 0;
        }
    }
    ptr[N] = (int32_t)nnz;
    double t0 = 0, t1 = 0;
    int64_t reps = 0;
    vxo_spmv_csr_f64_i32_omp(N, 1.0, 0, ptr, col, val, x, y);      /* warm-up */
#ifdef _OPENMP
    t0 = omp_get_wtime();
    do { vxo_spmv_csr_f64_i32_omp(N, 1.0, 0, ptr, col, val, x, y); ++reps; t1 = omp_get_wtime(); } while (t1 - t0 < seconds && reps < 100000);
#else
    reps = 1;
#endif
    out[0] = reps ? (t1 - t0) / (double)reps : 0; out[1] = P; out[4] = (double)reps;
    out[2] = 0;
    if (single_reps > 0) {
#ifdef _OPENMP
        double s0 = omp_get_wtime();
        for (int r = 0; r < single_reps; ++r) vxo_spmv_csr_f64_i32(N, 1.0, 0, ptr, col, val, x, y);
        out[2] = (omp_get_wtime() - s0) / single_reps;
#endif
    }
    double sum = 0;
    for (int64_t i = 0; i < N; ++i) sum += y[i];
    out[3] = sum;
    free(ptr); free(col); free(val); free(x); free(y);
    return 0;
}

int vxo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Hybrid ELL (+CSR tail).  vexcl/spmat/hybrid_ell.inl:66-114 (width choice:  */
/* smallest w such that 3 * #rows-wider-than-w < n), :138-198 (fill, column-  */
/* major, pitch = alignup(n,16), pad column = -1), kernel :238-269.           */
/* Single-device restatement (all columns local).                             */
/* ------------------------------------------------------------------------- */
int64_t vxo_hell_width_i32(int64_t n, const int32_t *ptr)
{
    int64_t maxw = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t w = ptr[i + 1] - ptr[i];
        if (w > maxw) maxw = w;
    }
    int64_t *hist = (int64_t *)calloc((size_t)maxw + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) hist[ptr[i + 1] - ptr[i]]++;
    const double ell_vs_csr = 3.0;
    int64_t rows = n, w = maxw;
    for (int64_t i = 0; i < maxw; ++i) {
        rows -= hist[i];                 /* rows wider than i */
        if (ell_vs_csr * (double)rows < (double)n) { w = i; break; }
    }
    free(hist);
    return w;
}

int64_t vxo_hell_pitch(int64_t n) { return (n + 15) / 16 * 16; }

/* returns the nnz of the CSR tail; csr_* must hold (n+1), tail-nnz entries   */
int64_t vxo_hell_build_f64_i32(int64_t n, const int32_t *ptr, const int32_t *col,
        const double *val, int64_t width, int64_t pitch,
        int32_t *ell_col, double *ell_val,
        int32_t *csr_ptr, int32_t *csr_col, double *csr_val)
{
    for (int64_t k = 0; k < pitch * width; ++k) { ell_col[k] = -1; ell_val[k] = 0; }
    int64_t tail = 0;
    csr_ptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t cnt = 0;
        for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) {
            if (cnt < width) {
                ell_col[i + pitch * cnt] = col[j];
                ell_val[i + pitch * cnt] = val[j];
                ++cnt;
            } else {
                if (csr_col) { csr_col[tail] = col[j]; csr_val[tail] = val[j]; }
                ++tail;
            }
        }
        csr_ptr[i + 1] = (int32_t)tail;
    }
    return tail;
}

void vxo_spmv_hell_f64_i32(int64_t n, double alpha, int append,
        int64_t width, int64_t pitch,
        const int32_t *ell_col, const double *ell_val,
        const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0;
        for (int64_t j = 0; j < width; ++j) {
            int32_t c = ell_col[i + j * pitch];
            if (c != -1) sum += ell_val[i + j * pitch] * x[c];
        }
        if (csr_ptr)
            for (int64_t j = csr_ptr[i], e = csr_ptr[i + 1]; j < e; ++j)
                sum += csr_val[j] * x[csr_col[j]];
        if (append) y[i] += alpha * sum; else y[i] = alpha * sum;
    }
}

/* ------------------------------------------------------------------------- */
/* Elementwise.  Generated statement operations.hpp:1856-1880 /               */
/* SURVEY appendix A.1:  prm_1[idx] = ((prm_2[idx]*prm_3[idx]) + sin(prm_4)). */
/* ------------------------------------------------------------------------- */
void vxo_ew_mul_add_sin_f64(int64_t n, const double *b, const double *c,
        const double *d, double *a)
{
    for (int64_t i = 0; i < n; ++i) a[i] = (b[i] * c[i]) + sin(d[i]);
}

/* ------------------------------------------------------------------------- */
/* Reductor.  SUM: plain accumulation (reductor.hpp:511-534); SUM_Kahan:      */
/* compensated accumulation (reductor.hpp:537-564, same recurrence the        */
/* reference test checks against via boost kahan accumulator,                 */
/* tests/vector_arithmetics.cpp:72-86).  Order of a parallel reduction is     */
/* implementation defined => SUM parity is by tolerance; the Kahan value is   */
/* the reference for both.                                                    */
/* ------------------------------------------------------------------------- */
double vxo_sum_f64(const double *x, int64_t n)
{ double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i]; return s; }

double vxo_sum_kahan_f64(const double *x, int64_t n)
{
    double s = 0, c = 0;
    for (int64_t i = 0; i < n; ++i) {
        double y = x[i] - c;
        double t = s + y;
        c = (t - s) - y;
        s = t;
    }
    return s;
}

double vxo_dot_kahan_f64(const double *a, const double *b, int64_t n)
{
    double s = 0, c = 0;
    for (int64_t i = 0; i < n; ++i) {
        double y = a[i] * b[i] - c;
        double t = s + y;
        c = (t - s) - y;
        s = t;
    }
    return s;
}

/* ------------------------------------------------------------------------- */
/* Deterministic input generators with the SHAPE of tests/random_matrix.hpp   */
/* (per row a uniform width in [0, nnz_per_row-1] of distinct sorted columns, */
/* values U[0,1)) and tests/random_vector.hpp (floats U[0,1), ints U[0,100]). */
/* The reference seeds std::default_random_engine from time(0); there is no   */
/* fixed stream to reproduce, so a splitmix64 stream with a caller-given seed */
/* is used instead.                                                           */
/* ------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

void vxo_random_f64(uint64_t seed, int64_t n, double *x)
{ uint64_t s = seed; for (int64_t i = 0; i < n; ++i) x[i] = u01(&s); }

void vxo_random_i32(uint64_t seed, int64_t n, int32_t lo, int32_t hi, int32_t *x)
{
    uint64_t s = seed;
    for (int64_t i = 0; i < n; ++i)
        x[i] = lo + (int32_t)(splitmix64(&s) % (uint64_t)(hi - lo + 1));
}

void vxo_random_u32(uint64_t seed, int64_t n, uint32_t *x)
{ uint64_t s = seed; for (int64_t i = 0; i < n; ++i) x[i] = (uint32_t)(splitmix64(&s) >> 32); }

static int cmp_i32(const void *a, const void *b)
{ int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

/* ptr must hold n+1; col/val must hold n*(nnz_per_row-1) (upper bound).      */
/* Returns nnz.  empty_tail rows at the end are left empty                    */
/* (tests/spmv.cpp:116-146 "768 trailing empty rows").                        */
int64_t vxo_random_matrix_f64_i32(uint64_t seed, int64_t n, int64_t m,
        int64_t nnz_per_row, int64_t empty_tail,
        int32_t *ptr, int32_t *col, double *val)
{
    uint64_t s = seed;
    int64_t nz = 0;
    ptr[0] = 0;
    for (int64_t k = 0; k < n; ++k) {
        int64_t width = (k >= n - empty_tail) ? 0 : (int64_t)(splitmix64(&s) % (uint64_t)nnz_per_row);
        if (width > m) width = m;
        int64_t got = 0;
        while (got < width) {
            int32_t c = (int32_t)(splitmix64(&s) % (uint64_t)m);
            int dup = 0;
            for (int64_t q = 0; q < got; ++q) if (col[nz + q] == c) { dup = 1; break; }
            if (!dup) col[nz + got++] = c;
        }
        qsort(col + nz, (size_t)width, sizeof(int32_t), cmp_i32);
        nz += width;
        ptr[k + 1] = (int32_t)nz;
    }
    for (int64_t j = 0; j < nz; ++j) val[j] = u01(&s);
    return nz;
}
