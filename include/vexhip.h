/*
 * vexhip.h -- C ABI of libvexhip.so: the MI355X (gfx950) launch layer under the
 * vex:: header API (vexcl/ *.hpp) and the Python harness (vexcl_amd/).
 *
 * The reference (ddemidov/vexcl) has no FFI; its seam is the compile-time
 * `vex::backend` concept (vexcl/backend.hpp:40-96).  Every entry point below
 * names the reference interface it stands in for.  Conventions:
 *   - every function returns 0 on success, non-zero on failure; the failure
 *     text (file:line + HIP error string, like backend/cuda/error.hpp:119-145)
 *     is available from vexhip_last_error() on the calling thread;
 *   - plain pointers and sizes only; `stream` is a hipStream_t passed as
 *     void* (NULL = the device's null stream); `dev` is a HIP device ordinal;
 *   - all device pointers must belong to `dev`; calls are asynchronous on
 *     `stream` unless stated otherwise (the reference's assignments are
 *     enqueue-only too, operations.hpp:1886-1894).
 */
#ifndef VEXHIP_H
#define VEXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VEXHIP_ABI_VERSION 1

/* ---- errors (backend/cuda/error.hpp:119-156, util.hpp:67-77) ------------ */
const char *vexhip_last_error(void);
/* the hipError_t behind the last failure on the calling thread (0: a check of the library's own), so that callers can tell an
 * allocation that did not fit (VEXHIP_ERROR_OUT_OF_MEMORY) from a fault without parsing the text                              */
int vexhip_last_error_code(void);
enum { VEXHIP_ERROR_OUT_OF_MEMORY = 2 };   /* hipErrorOutOfMemory */
int vexhip_abi_version(void);

/* ---- devices (backend/cuda/context.hpp:96-203,383-413; devlist.hpp) ----- */
typedef struct vexhip_device_props {
    char     name[256];
    char     arch[64];            /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    int32_t  compute_units;       /* 256 on MI355X */
    int32_t  wavefront_size;      /* 64 */
    int32_t  max_threads_per_block;
    int32_t  lds_bytes_per_block; /* max shared memory per block */
    int32_t  clock_khz;
    int32_t  l2_bytes;
    uint64_t global_mem_bytes;
    int32_t  pci_bus_id;
    int32_t  reserved;
} vexhip_device_props;

int vexhip_device_count(int *count);
int vexhip_device_get_props(int dev, vexhip_device_props *props);
int vexhip_device_sync(int dev);
int vexhip_mem_info(int dev, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- command queues = streams (backend/cuda/context.hpp:205-260) -------- */
int vexhip_stream_create(int dev, void **stream);
int vexhip_stream_destroy(int dev, void *stream);
int vexhip_stream_sync(int dev, void *stream);                 /* command_queue::finish() */

/* ---- events (backend/cuda/event.hpp:51-124; enqueue_marker/barrier) ----- */
int vexhip_event_create(int dev, int timing, void **event);
int vexhip_event_destroy(int dev, void *event);
int vexhip_event_record(int dev, void *event, void *stream);   /* enqueue_marker */
int vexhip_event_sync(int dev, void *event);                   /* event::wait()  */
int vexhip_stream_wait_event(int dev, void *stream, void *event); /* enqueue_barrier(q, wait_list) */
int vexhip_event_elapsed_ms(int dev, void *start, void *stop, float *ms);

/* ---- device_vector<T> storage (backend/cuda/device_vector.hpp:66-214) --- */
/* The library's switches (VEXHIP_* / VEXCL_* environment variables: A/B experiments, diagnostics, test hooks -- none is needed in
 * normal use) are looked up in a snapshot of the environment taken at first use and again whenever an object is created (a matrix,
 * a plan, a window, a step, a communicator); products never read the environment.  This call takes the snapshot NOW.           */
int vexhip_reload_env(void);
int vexhip_malloc(int dev, size_t bytes, void **ptr);
int vexhip_free(int dev, void *ptr);
/* Where vexhip_malloc places an allocation of 64 MiB or more (round 6; host arithmetic, no device): bytes to skip from the raw
 * address hipMalloc returned so that the vector starts at a multiple of 64 MiB plus a stagger of 0 / 2 / 4 / 6 / 8 MiB (by the
 * allocation's ordinal) -- vectors a product reads and writes in lockstep then differ by less than 10 MiB mod 64 MiB, where the
 * headline product runs at its fast end (profiles/r06_xy_gap.json).  Smaller allocations: 0.  VEXHIP_MALLOC_STAGGER=0 turns it off.
 * (vexhip_free takes the pointer vexhip_malloc returned.)                                                                        */
size_t vexhip_malloc_placement(size_t bytes, uint64_t raw_address, unsigned ordinal);
size_t vexhip_malloc_stagger(size_t bytes, unsigned ordinal);
/* svm_vector<T> storage (backend/cuda/svm_vector.hpp:57-62: cuMemAllocManaged): memory addressable by the host and
 * the device alike (hipMallocManaged); released with vexhip_free. */
int vexhip_malloc_managed(int dev, size_t bytes, void **ptr);
int vexhip_memcpy_h2d(int dev, void *dst, const void *src, size_t bytes, void *stream, int blocking); /* device_vector::write */
int vexhip_memcpy_d2h(int dev, void *dst, const void *src, size_t bytes, void *stream, int blocking); /* device_vector::read  */
int vexhip_memcpy_d2d(int dev, void *dst, const void *src, size_t bytes, void *stream);
int vexhip_memcpy_peer(int dst_dev, void *dst, int src_dev, const void *src, size_t bytes, void *stream);
int vexhip_memset(int dev, void *ptr, int byte, size_t bytes, void *stream);
int vexhip_host_alloc(size_t bytes, void **ptr);               /* pinned staging (device_vector::map) */
int vexhip_host_free(void *ptr);

/* ---- JIT: build_sources + kernel (backend/cuda/compiler.hpp:53-116,
 *      backend/cuda/kernel.hpp:45-244, cache dir backend/common.hpp:215-285) */
int vexhip_module_compile(int dev, const char *source, const char *options, void **module);
int vexhip_module_unload(int dev, void *module);
int vexhip_module_get_function(int dev, void *module, const char *name, void **function);
int vexhip_function_max_threads(int dev, void *function, int *max_threads_per_block, int *static_lds_bytes);
int vexhip_launch(int dev, void *function,
        unsigned grid_x, unsigned grid_y, unsigned grid_z,
        unsigned block_x, unsigned block_y, unsigned block_z,
        unsigned dynamic_lds_bytes, void *stream, void **args);
/* number of JIT compilations that missed both the in-process and the on-disk
 * cache since load (the reference's VEXCL_CACHE_KERNELS behaviour is testable
 * through it) */
int vexhip_jit_stats(uint64_t *compiled, uint64_t *disk_hits);
/* compile-only check of a kernel source for `arch` (e.g. "gfx950"); needs no
 * GPU, loads nothing, bypasses the caches */
int vexhip_jit_check(const char *source, const char *options, const char *arch);

/* ---- fixed primitive: CSR SpMV  (spmat/csr.inl:153-185 `csr_spmv`) ------
 * y[i] (= | +=) alpha * sum_{j in [ptr[i],ptr[i+1])} val[j]*x[col[j]],
 * summation in CSR order, scale applied after the sum.  append!=0 => "+=".
 * Hand-written LDS-staged kernel: one workgroup streams the (col,val) range of
 * 256 consecutive rows with 16-byte loads, products are staged in LDS and each
 * lane folds its own row in CSR order.                                        */
int vexhip_spmv_csr_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y);
int vexhip_spmv_csr_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y);
int vexhip_spmv_csr_f64_i64(int dev, void *stream, int64_t n, double alpha, int append,
        const int64_t *ptr, const int64_t *col, const double *val, const double *x, double *y);
/* tuning variant selector for the CSR kernel (bench / sweep tool only):
 * 0 = default. */
/* Row-subset CSR, always "+=": y[rows[k]] += alpha * sum_{j in [ptr[k], ptr[k+1])} val[j] * x[col[j]], k < nrows.
 * The remote part of a partitioned matrix (spmat/csr.inl:92-131 `rem`, applied by `mul_remote`, spmat.hpp:177-183)
 * has entries only in rows next to a partition boundary; rows = their (strictly increasing) local ids.   */
int vexhip_spmv_csr_rows_f64_i32(int dev, void *stream, int64_t nrows, double alpha, const int32_t *rows,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y);
int vexhip_spmv_csr_rows_f32_i32(int dev, void *stream, int64_t nrows, float alpha, const int32_t *rows,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y);
int vexhip_spmv_csr_set_variant(int variant);
/* The CSR product with the strip traversal of the SELL kernels (see vexhip_traversal below): for banded /
 * stencil matrices whose far diagonals are "planes" apart, every XCD owns a strip of every plane, so x is
 * fetched from HBM about once.  vexhip_csr_traversal_i32 inspects the first 64 entries of every row
 * (blocking, set-up time); grid_blocks = 0 means "no reordering pays".  rows_per_block = 256 for the CSR kernel. */
struct vexhip_traversal;
int vexhip_csr_traversal_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col,
        int rows_per_block, struct vexhip_traversal *traversal);
int vexhip_spmv_csr_ordered_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        const int32_t *ptr, const int32_t *col, const double *val, const double *x, double *y,
        const struct vexhip_traversal *traversal);
int vexhip_spmv_csr_ordered_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        const int32_t *ptr, const int32_t *col, const float *val, const float *x, float *y,
        const struct vexhip_traversal *traversal);
int vexhip_spmv_hell_set_variant(int variant);

/* ---- fixed primitive: hybrid ELL SpMV (spmat/hybrid_ell.inl:238-300) ----
 * ELL part column-major with pitch (multiple of 16), padding column -1; CSR
 * tail may be absent (csr_ptr == NULL), as the reference passes 0 for an empty
 * part (hybrid_ell.inl:283-296).  ell_width == 0 => CSR tail only.            */
int vexhip_spmv_hell_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        int64_t ell_width, int64_t ell_pitch, const int32_t *ell_col, const double *ell_val,
        const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y);
int vexhip_spmv_hell_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        int64_t ell_width, int64_t ell_pitch, const int32_t *ell_col, const float *ell_val,
        const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y);

/* Traversal order for banded / stencil matrices (setup, blocking).
 * Detects constant column offsets (e.g. +-1, +-n, +-n^2 of a 3-D stencil) in the
 * ELL part.  If the farthest one (S_big, a "plane") is too long for the x values of
 * three planes to stay in one XCD's 4 MiB L2, the traversal gives every XCD a strip
 * of consecutive row-blocks of EVERY plane and sweeps plane after plane: both the
 * +-n neighbours and the +-n^2 re-reads stay in ONE private L2 (workgroup b runs on
 * XCD b % 8), so x is fetched from HBM about once instead of three times.
 * grid_blocks == 0 means "use the plain order".  The ordered product computes
 * exactly what the plain one computes (any permutation of row-blocks is correct). */
typedef struct vexhip_traversal {
    int64_t grid_blocks;      /* workgroups to launch; 0 = plain order                          */
    int64_t chunk;            /* > 0: strip order, computed arithmetically inside the kernel:   */
    int64_t planes;           /*   workgroup b -> row-block p*plane_blocks + t*8*chunk          */
    int64_t plane_blocks;     /*   + (b%8)*chunk + (b/8)%chunk, p = (b/(8*chunk)) % planes      */
    const int32_t *order;     /* != NULL: explicit workgroup -> row-block map in device memory  */
} vexhip_traversal;
int64_t vexhip_hell_order_capacity(int64_t n);     /* ints needed by an explicit map */
/* mode 0 = default (arithmetic strips, `order` may be NULL), 1 = per-XCD slabs,
 * 2 = round-robin tiles, 100+k = strips of k row-blocks (tuning; modes 1, 2 write `order`) */
int vexhip_hell_order_i32(int dev, void *stream, int64_t n, int64_t ell_width, int64_t ell_pitch,
        const int32_t *ell_col, int mode, int32_t *order, int64_t capacity, vexhip_traversal *out);
int vexhip_sell_order_i32(int dev, void *stream, int64_t n, int64_t ell_width, int value_bytes,
        const void *sell, int mode, int32_t *order, int64_t capacity, vexhip_traversal *out);
int vexhip_spmv_hell_ordered_f64_i32(int dev, void *stream, int64_t n, double alpha, int append,
        int64_t ell_width, int64_t ell_pitch, const int32_t *ell_col, const double *ell_val,
        const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal);
int vexhip_spmv_hell_ordered_f32_i32(int dev, void *stream, int64_t n, float alpha, int append,
        int64_t ell_width, int64_t ell_pitch, const int32_t *ell_col, const float *ell_val,
        const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal);

/* Sliced ELL (SELL-512): the ELL part stored slice-major.  One slice = the 512 rows
 * of one workgroup = ONE contiguous region of w*512*(4 + sizeof(value)) bytes:
 * first its w*512 int32 columns (element (r, j) at j*512 + r), then its w*512
 * values in the same order.  Same width rule, same CSR tail, same arithmetic and
 * summation order as hybrid ELL.  The buffer holds vexhip_sell_bytes(n, w,
 * value_bytes) bytes; `traversal` as for the ordered HELL product (NULL = plain).  */
int64_t vexhip_sell_bytes(int64_t n, int64_t ell_width, int value_bytes);
int vexhip_sell_fill_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t ell_width, void *sell);
int vexhip_sell_fill_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t ell_width, void *sell);
int vexhip_spmv_sell_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width,
        const void *sell, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal);
int vexhip_spmv_sell_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width,
        const void *sell, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal);

/* SELL-512 with 8-bit diagonal codes ("SELL8").  For banded / stencil matrices (column -
 * row) takes few distinct values: if the ELL part uses at most 255 distinct diagonals,
 * every column index is stored as ONE byte (position of its diagonal in the sorted table
 * `deltas`, 255 = padding) and rebuilt in the kernel as row + deltas[code]: 9 instead of
 * 12 bytes per fp64 entry, same arithmetic, same order, bit-identical results.
 *   analyze: *ndeltas = number of diagonals (table written to deltas[256], device memory),
 *            or -1 if the matrix has more than 255 (keep 32-bit columns then);
 *   fill:    encodes the ELL part (width/tail as for hybrid ELL) into `buf`
 *            (vexhip_sell8_bytes) and returns the strip traversal for it;
 *   slice layout: ceil(w/2) KiB of codes (word [jp][t] = codes of columns 2jp, 2jp+1 for
 *            rows 2t, 2t+1), then w*512 values, j-major.                                 */
int64_t vexhip_sell8_bytes(int64_t n, int64_t ell_width, int value_bytes);
int vexhip_sell8_analyze_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col,
        int64_t ell_width, int32_t *deltas, int *ndeltas);
int vexhip_sell8_fill_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t ell_width, const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *traversal);
int vexhip_sell8_fill_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t ell_width, const int32_t *deltas, int ndeltas, void *buf, vexhip_traversal *traversal);
int vexhip_spmv_sell8_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal);
int vexhip_spmv_sell8_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal);

/* SELL8V: SELL8 whose VALUES are coded too.  When the ELL part holds at most 255 distinct values (bit patterns;
 * matrices assembled from a constant-coefficient stencil: the 7-point Poisson matrix has three), each value is
 * stored as one byte -- its position in a sorted table kept in LDS by the kernel -- next to the diagonal code:
 * 2 bytes per entry instead of 9 (fp64).  Same arithmetic in the same order: bit-identical to SELL8 / HELL / CSR.
 * The reference keeps a separate class for such matrices (SpMatCCSR, spmat/ccsr.hpp:55-280); here vex::SpMat detects
 * them.   values: 256 entries on the device, sorted by bit pattern, nvalues valid (-1: not applicable);
 * slice layout: ceil(w/2) KiB of diagonal codes, then ceil(w/2) KiB of value codes, both packed as in SELL8.     */
int64_t vexhip_sell8v_bytes(int64_t n, int64_t ell_width);
/* Slice dictionary.  The CODES of a 512-row slice repeat on a structured grid: the 262 144 slices of the 512^3 Poisson
 * matrix hold TWO distinct code blocks, and so do those of a variable-coefficient operator with the same pattern.
 * vexhip_slice_dictionary numbers the distinct code blocks -- the first slice_bytes bytes of every slice of `buf`
 * (nslices slices, stride_bytes apart, device memory) -- in order of first appearance: a 64-bit hash per block, then a
 * word-by-word comparison of every block with the representative of its number.  It writes the number of every slice to
 * blocks[nslices] (device) and the representatives to pool[*nblocks x slice_bytes] (device, capacity max_blocks).
 * *nblocks = -1 when there are more than max_blocks distinct blocks (or two different blocks share a hash): nothing
 * valid was written.  The _dict products read the codes of slice s at pool + blocks[s] * slice_bytes -- value-coded
 * storage (SELL8V): the whole slice, `pool` replaces the buffer; diagonal codes with stored values (SELL8): the code
 * part, the values stay in `buf`.  4 bytes per slice instead of the code stream, the pool stays in L1 / L2; same
 * codes, same arithmetic, same results.                                                                             */
int vexhip_slice_dictionary(int dev, void *stream, int64_t nslices, int64_t stride_bytes, int64_t slice_bytes, const void *buf,
        int64_t max_blocks, int32_t *blocks, void *pool, int64_t *nblocks);
int vexhip_spmv_sell8_dict_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width, const void *buf, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal);
int vexhip_spmv_sell8_dict_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width, const void *buf, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal);
/* March products (round 3): the _dict products with the x window of the NEAR diagonals staged in an LDS ring that a
 * workgroup carries along a run of consecutive slices -- one coalesced load of the 512 new elements per slice instead of
 * one gather per near column; up to two far diagonals (+-n^2 of a 3-D grid operator) are requested one slice ahead as two
 * more coalesced streams and parked in LDS; a matrix with a third far diagonal is declined (round 4).  Replaces, like every SELL product, the per-row
 * gathers of the reference's ELL kernel (vexcl/spmat/hybrid_ell.inl:238-269).  Bit-identical to the _dict products.
 * vexhip_sell8_march_plan decides from the diagonal table and the slice numbers whether it applies (usable = 0: call the
 * _dict product; value-coded storage only -- with stored values the product is bound by the value stream, which the pair
 * kernel already moves at 0.9 of the copy rate): the code block must rarely change from slice to slice, and the near
 * diagonals (grown from 0 outwards) are those whose ring + mirror + far slots + tables fit 48 KiB of LDS and whose first
 * window is at most 2048 elements.  run: from the slice count (32 for large matrices; shorter so that a CU sees at least
 * two rounds of workgroups), VEXHIP_MARCH_RUN overrides.  x_last = largest valid index of x (the fills report the largest
 * ELL column through vexhip_sell8_last_fill_max_col, per thread).                                                        */
typedef struct vexhip_march { int32_t lo, hi;      /* smallest / largest NEAR diagonal (lo <= 0 <= hi)                */
                              int32_t run;         /* consecutive slices per workgroup (divides the strip length)     */
                              int32_t usable;
                              int64_t x_last;
                              int32_t nfar, far[3]; /* far[0..nfar-1]: the (<= 2) far diagonals nearest to the window, requested one slice ahead; far[2] reserved */
                            } vexhip_march;
int vexhip_sell8_march_plan(int dev, void *stream, const int32_t *deltas, int ndeltas, const int32_t *blocks, int64_t nslices,
        int value_bytes, const vexhip_traversal *traversal, int64_t x_last, vexhip_march *out);
/* The PLANE product (plane.hip, round 4; same semantics, hybrid_ell.inl:238-269): value-coded storage with a slice dictionary
 * whose diagonals are {0, +-1, +-512, +-P}, P = 512 * lines_per_plane -- a 7-point operator on a grid with 512-point lines.
 * A workgroup owns two adjacent grid lines and walks through `depth` planes; the +-512 and +-P neighbours of a lane's rows
 * are pairs the same lane loaded (registers), the +-1 neighbours come by DPP wave shifts: no LDS, every x line requested
 * twice instead of three times.  The plan checks on the host that every dictionary block keeps its rows' diagonals in
 * ascending order of position (position order = storage order), that at most 1/16 of the slices use another block than the
 * most frequent one (hot_block), that there is no CSR tail and that x can be read in whole lines ((x_last + 1) % 512 == 0);
 * usable = 0 otherwise and the march / pair products stay.  VEXHIP_PLANE_DEPTH overrides depth.  value_bytes 8 or 4: the
 * fp32 product (plane32.hip, round 5) reads the same storage with FOUR rows per lane -- a 512-point line of floats is 128
 * lanes x 16 bytes, a workgroup is two waves and walks half of `depth` (VEXHIP_PLANE32_DEPTH overrides).               */
typedef struct vexhip_plane { int32_t usable;
                              int32_t lines_per_plane;   /* the far diagonals are +-512 * lines_per_plane                     */
                              int32_t planes;            /* ceil(slices / lines_per_plane)                                    */
                              int32_t depth;             /* planes one workgroup walks through                                */
                              int32_t hot_block;         /* dictionary block kept decoded in registers                        */
                              int32_t tile;              /* grid lines per workgroup: 2 or 4 (divides lines_per_plane)        */
                              int32_t store_policy;      /* y stores: 0 non-temporal, 1 non-temporal + sc1, 2 sc0 sc1, 3 plain  */
                              int32_t table_pitch;       /* 0: `pool` holds SELL-512 code blocks; > 0: class tables of the grid storage (vexhip_grid.pitch) */
                              int32_t flat;              /* 1: no entry at +-512 anywhere (vexhip_grid.flat: a 2-D operator on virtual lines): the walk requests no neighbour lines */
                              int32_t reserved;
                              int64_t x_last;
                            } vexhip_plane;
int vexhip_sell8_plane_plan(int dev, void *stream, const int32_t *deltas, int ndeltas, const int32_t *blocks, int64_t nslices,
        const void *pool, int64_t dictionary_blocks, int64_t ell_width, int64_t rows, int64_t tail_nnz, int value_bytes,
        int64_t x_last, vexhip_plane *out);
/* the plan's choice of walks by itself (host arithmetic, no device; usable stays 0): ONE workgroup per CU where the tiles come
 * out that way, else at least six short walks per CU (round 5) -- so that the rule can be checked for any device size     */
int vexhip_sell8_plane_geometry(int cus, int64_t lines_per_plane, int64_t planes, vexhip_plane *out);
/* y = x with one 16-byte pair per lane and non-temporal stores: the measured ceiling for a product whose HBM traffic is x once
 * + y once (bench.py roofline.device_copy_hand); the reference times its copies through clEnqueueCopyBuffer
 * (vexcl/backend/opencl/device_vector.hpp) -- this is the device-side counterpart used as a yardstick only.             */
int vexhip_stream_copy_f64(int dev, void *stream, const double *x, double *y, int64_t n);
int vexhip_spmv_sell8v_plane_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const double *x, double *y, const vexhip_plane *plane);
int vexhip_spmv_sell8v_plane_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const float *x, float *y, const vexhip_plane *plane);
/* planes per workgroup the fp32 plane product takes for this grid on a device of `cus` CUs (host arithmetic, no device): its walks
 * are chosen per launch and must keep (depth + 4) planes of 2048-byte lines below 2^32 bytes; 0 = no depth fits (the launch fails) */
int64_t vexhip_sell8_plane_f32_depth(int cus, int64_t lines_per_plane, int64_t planes);
/* The GRID product (grid.hip, round 4; same semantics, hybrid_ell.inl:238-269; the size-agnostic stencil form the reference
 * reaches through SpMatCCSR, spmat/ccsr.hpp:55-113): the plane product for grids of any line length.  Value-coded storage
 * (with or without a slice dictionary: `blocks` may be NULL, `codes` is then the per-slice buffer) whose diagonals are
 * {0, +-1, +-nx, +-nx * lines_per_plane}, rows = a whole number of lines.  The plan re-expresses the matrix by grid line on the
 * device: seven bytes per row (the value code per position, 255 = no entry), lines with equal rows form a class (<= 128), every
 * line is verified against its class; it declines (usable = 0) when a row's entries do not ascend by position, when more
 * than 1/4 of the lines use another class than the most frequent one, for fp32, a CSR tail, fewer than 2^23 rows (x within the caches: the pair product is as fast) or 4 planes.
 * The product owns two adjacent lines (one segment of <= 512 rows of them, <= 1024 for nx > 768) per workgroup and walks through `depth` planes;
 * lines need not be 16-byte aligned (odd nx), lines_per_plane may be odd.  line_class / table are device memory owned by the
 * plan: vexhip_sell8_grid_release frees them.  VEXHIP_PLANE_DEPTH / VEXHIP_PLANE_STORE override, VEXHIP_NO_GRID declines.
 * Round 6: a matrix with the diagonals {0, +-1, +-W} only (a 5-point operator on a 2-D grid; the reference's SpMatCCSR has no notion
 * of dimension either, spmat/ccsr.hpp:55-113) is stored the same way with its rows cut into VIRTUAL lines -- nx = 512 where an even
 * number >= 4 of them make a row (the plane product), else the longest even divisor of W in [128, min(1024, W / 10)] --,
 * lines_per_plane = W / nx, +-W as the far pair, flat = 1; rows without such a divisor keep the SELL-512 products.               */
typedef struct vexhip_grid { int32_t usable;
                             int32_t nx;                /* rows per grid line: the middle diagonals are +-nx                  */
                             int32_t lines_per_plane;   /* the far diagonals are +-nx * lines_per_plane                       */
                             int32_t planes;            /* ceil(lines / lines_per_plane)                                      */
                             int32_t depth;             /* planes one workgroup walks through                                 */
                             int32_t segments;          /* segments per line: ceil(nx / 512), ceil(nx / 1024) for nx > 768    */
                             int32_t segment_rows;      /* rows per segment (even, <= 512; <= 1024 for nx > 768)              */
                             int32_t threads;           /* lanes per workgroup: 64 * ceil(segment_rows / 128), <= 512         */
                             int32_t hot_class;         /* line class kept decoded in registers                               */
                             int32_t classes;           /* distinct line classes                                              */
                             int32_t pitch;             /* bytes per position row of a class table (>= what the lanes read)   */
                             int32_t store_policy;      /* as vexhip_plane.store_policy                                       */
                             int32_t flat;              /* 1: no line has an entry at +-nx (a 5-point operator on a 2-D grid whose rows are cut into
                                                           virtual lines, +-row length = the far pair): the walk requests no neighbour lines */
                             int32_t reserved;
                             int64_t x_last;
                             const int32_t *line_class; /* device: class of every grid line                                   */
                             const void *table;         /* device: classes x 7 positions x pitch value codes                  */
                           } vexhip_grid;
int vexhip_sell8_grid_plan(int dev, void *stream, const int32_t *deltas, int ndeltas, const void *codes, const int32_t *blocks,
        int64_t ell_width, int64_t rows, int64_t tail_nnz, int value_bytes, int64_t x_last, vexhip_grid *out);
int vexhip_sell8_grid_release(int dev, vexhip_grid *grid);
/* The geometry the two set-ups choose for an nx x lines_per_plane x planes grid on a device with `cus` compute units -- host
 * arithmetic only, no device is touched: nx .. threads, pitch and store_policy are filled, usable = 0, no tables (depth = 0: no
 * geometry, a plane too large for 32-bit offsets).  vexhip_sell8_grid_check returns 0 when the product accepts the plan's geometry
 * for a matrix of n rows (the same test the product runs before every launch).  Both exist so that the plan of EVERY line length
 * can be checked where there is no GPU (tests/test_capi_exports.py).                                                            */
int vexhip_sell8_grid_geometry(int cus, int64_t nx, int64_t lines_per_plane, int64_t planes, vexhip_grid *out);
/* Round 6: the virtual grid line (in points) a matrix with the diagonals {0, +-1, +-row_length} -- a 5-point operator on a 2-D grid -- is
 * stored by: 512 where an even number >= 4 of them make a row, else the longest even divisor of row_length in [128, min(1024,
 * row_length / 10)], else 0 (such a matrix keeps the SELL-512 products).  Host arithmetic only, like _grid_geometry.                */
int64_t vexhip_sell8_grid_virtual_line(int64_t row_length);
int vexhip_sell8_grid_check(const vexhip_grid *grid, int64_t n);
int vexhip_spmv_sell8v_grid_f64(int dev, void *stream, int64_t n, double alpha, int append, const double *values,
        const double *x, double *y, const vexhip_grid *grid);
/* the same product for float matrices (grid32.hip, round 5): four rows per lane, 16-byte requests at 4-byte addresses; the plan
 * (vexhip_grid) is the fp64 one, the walk is cut shorter at launch (VEXHIP_GRID32_DEPTH overrides)                      */
int vexhip_spmv_sell8v_grid_f32(int dev, void *stream, int64_t n, float alpha, int append, const float *values,
        const float *x, float *y, const vexhip_grid *grid);
int64_t vexhip_sell8_last_fill_max_col(void);
int vexhip_spmv_sell8v_march_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal, const vexhip_march *march);
int vexhip_spmv_sell8v_march_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal, const vexhip_march *march);
int vexhip_spmm_sell8_dict_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t ell_width,
        const void *buf, const void *pool, const int32_t *blocks, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *const *x, double *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell8_dict_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t ell_width,
        const void *buf, const void *pool, const int32_t *blocks, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *const *x, float *const *y, const vexhip_traversal *traversal);
int vexhip_spmv_sell8v_dict_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const double *values, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal);
int vexhip_spmv_sell8v_dict_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width, const void *pool,
        const int32_t *blocks, const int32_t *deltas, const float *values, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell8v_dict_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t ell_width,
        const void *pool, const int32_t *blocks, const int32_t *deltas, const double *values, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *const *x, double *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell8v_dict_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t ell_width,
        const void *pool, const int32_t *blocks, const int32_t *deltas, const float *values, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *const *x, float *const *y, const vexhip_traversal *traversal);
int vexhip_sell8v_analyze_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const double *val,
        int64_t ell_width, double *values, int *nvalues);
int vexhip_sell8v_analyze_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const float *val,
        int64_t ell_width, float *values, int *nvalues);
int vexhip_sell8v_fill_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t ell_width, const int32_t *deltas, int ndeltas, const double *values, int nvalues, void *buf, vexhip_traversal *traversal);
int vexhip_sell8v_fill_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t ell_width, const int32_t *deltas, int ndeltas, const float *values, int nvalues, void *buf, vexhip_traversal *traversal);
int vexhip_spmv_sell8v_f64_i32(int dev, void *stream, int64_t n, double alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const double *values, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *x, double *y, const vexhip_traversal *traversal);
int vexhip_spmv_sell8v_f32_i32(int dev, void *stream, int64_t n, float alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const float *values, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *x, float *y, const vexhip_traversal *traversal);

/* A/B switch of the SELL / SELL8 / SELL8V products (results are bit-identical either way):
 *   variant 0 (default): pair kernels -- a lane reads x for its two rows with ONE 16-byte load per ELL column
 *                        wherever the fill kernels could align the two rows (see the storage notes above);
 *   variant 1:           one 8-byte gather per entry (round 1);
 *   variant 2:           pair kernels even where a march plan is given (the _march entry points then behave as _dict). */
int vexhip_spmv_sell8_set_variant(int variant);

/* ---- vex::SpMat on one device: ONE object that owns the storage selection -------------------------------------
 * Replaces the per-device matrix objects the reference builds in SpMat's constructor (vexcl/spmat.hpp:84-104:
 * SpMatCSR for CPU devices, SpMatHELL for GPUs; spmat/hybrid_ell.inl:60-216) and their mul() entry points
 * (spmat/csr.inl:186-232, hybrid_ell.inl:218-300).  Built from DEVICE CSR arrays (int32 indices, sorted or not);
 * nothing is staged through the host.  create() decides, in this order: hybrid-ELL width and CSR tail
 * (hybrid_ell.inl:103-110) -> 1-byte diagonal codes if the ELL part uses <= 254 diagonals -> 1-byte value codes if
 * it holds <= 255 distinct values -> otherwise 32-bit columns; plain CSR when the ELL part would be empty.
 * Round 4, in front of all that (fp64, format AUTO / SELL8V, >= 2^23 rows): a probe of 8192 rows names the diagonals; if they
 * are {0, +-1, +-nx, +-nx * ny} -- a 7-point operator on a grid -- ONE pass over the CSR arrays stores the matrix BY GRID LINE
 * (vexhip_grid: a class per line, a table of value codes per class; values numbered on the device in the order they are met,
 * <= 254 of them, <= 128 classes) and the plane / grid products run on it; info then reports format SELL8V, grid.usable,
 * sell = code_pool = NULL.  Anything that does not fit gives the pass up and the selection above takes over
 * (VEXHIP_SPMAT_NO_GRID_BUILD skips the attempt).
 * `format` pins a less compact storage (tests, A/B): AUTO = most compact the matrix allows.
 * apply: y (=|+=) alpha * A * x, bit-identical for every storage (products rounded, rows folded in CSR order).   */
typedef struct vexhip_spmat vexhip_spmat;
enum { VEXHIP_SPMAT_AUTO = 0,      /* create(): most compact storage; info: never reported                     */
       VEXHIP_SPMAT_SELL8V = 1,    /* diagonal codes + value codes (2 B per entry)                                */
       VEXHIP_SPMAT_SELL8 = 2,     /* diagonal codes, values as they are (1 + sizeof(V) B per entry)              */
       VEXHIP_SPMAT_SELL = 3,      /* 32-bit columns (4 + sizeof(V) B per entry)                                  */
       VEXHIP_SPMAT_CSR = 4 };     /* the CSR arrays themselves (csr_stream_kernel)                               */
enum { VEXHIP_SPMAT_BORROW_CSR = 1,      /* format CSR: keep the caller's arrays instead of copying them (caller keeps them alive) */
       VEXHIP_SPMAT_NO_DICTIONARY = 2,   /* value-coded storage: keep one block per slice even if the slices repeat (A/B, tests)   */
       VEXHIP_SPMAT_NO_MARCH = 4,        /* keep the pair products where the march / plane products would apply (A/B, tests)       */
       VEXHIP_SPMAT_NO_PLANE = 8,        /* keep the march product where the plane product would apply (A/B, tests)                */
       VEXHIP_SPMAT_NO_GRID_BUILD = 16,  /* build the SELL-512 storage even where the matrix could be stored by grid line (A/B, tests) */
       VEXHIP_SPMAT_PLAIN_ORDER = 64,    /* 32-bit-column storage: slices dealt to the XCDs round-robin even where the default gives every XCD a contiguous eighth (A/B) */
       VEXHIP_SPMAT_SQUARE = 32 };       /* x has at least `rows` elements whatever the largest column that occurs (the set-up assumes     *
                                          * only max column + 1 otherwise: vex::SpMat takes n AND m, spmat.hpp:56-60)                     */
typedef struct vexhip_spmat_info {
    int32_t format, value_type, device, ndeltas, nvalues, reserved;
    int64_t rows, nnz, ell_width, tail_nnz, sell_bytes;
    int64_t matrix_bytes;           /* bytes of matrix data one product streams (storage actually read)          */
    const void *sell; const int32_t *deltas; const void *values;           /* SELL storage (make_inline reads it)  */
    const int32_t *csr_ptr, *csr_col; const void *csr_val;                 /* CSR tail, or the matrix (format CSR) */
    vexhip_traversal traversal;
    const int32_t *slice_blocks;    /* slice dictionary (NULL = none): the codes of slice s are block slice_blocks[s] of       */
    const void *code_pool;          /* code_pool, which holds dictionary_blocks distinct code blocks.  SELL8V: `sell` IS the   */
    int64_t dictionary_blocks;      /* pool (no per-slice storage left); SELL8: `sell` keeps the values, slice-major           */
    vexhip_march march;             /* march product (usable = 1: apply() runs it; see vexhip_sell8_march_plan)                */
    vexhip_plane plane;             /* plane product (usable = 1: apply() prefers it to the march product; vexhip_sell8_plane_plan) */
    vexhip_grid grid;               /* grid product (usable = 1: apply() runs it where the plane product does not apply; vexhip_sell8_grid_plan) */
    char product[64];               /* round 6: the kernel a product of this matrix launches (spmat.hip select_product: ONE table) ...          */
    char reason[320];               /* ... why this storage was chosen and why that product; also printed under VEXHIP_DEBUG                    */
} vexhip_spmat_info;
int vexhip_spmat_create_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int format, int flags, vexhip_spmat **out);
int vexhip_spmat_create_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int format, int flags, vexhip_spmat **out);
/* The same with 64-bit ROW POINTERS (a device may hold 2^31 entries or more: the reference's default index type is
 * size_t, vexcl/spmat.hpp:56-57; a 288 GB part holds such matrices).  Columns stay 32-bit (< 2^31 columns per device);
 * the CSR tail of a hybrid-ELL storage must stay below 2^31 entries.  Same storage selection, same products.        */
int vexhip_spmat_create_f64_p64(int dev, void *stream, int64_t n, const int64_t *ptr, const int32_t *col, const double *val,
        int format, int flags, vexhip_spmat **out);
int vexhip_spmat_create_f32_p64(int dev, void *stream, int64_t n, const int64_t *ptr, const int32_t *col, const float *val,
        int format, int flags, vexhip_spmat **out);
int vexhip_spmat_destroy(vexhip_spmat *A);
int vexhip_spmat_apply_f64(const vexhip_spmat *A, void *stream, double alpha, int append, const double *x, double *y);
/* round 6: y = alpha A x + beta z (z != NULL; z may be x or y; x != y) -- what the reference's additive expressions amount to when their vector
 * part is one vector (vector.hpp:698-801: `y = z - A * x`, a residual, is "y = z" then "y -= A * x": two passes over y; spmat.hpp:120-185).  One
 * pass where the product takes the addend (the plane product; z == x costs no byte more than y = A x), otherwise y = beta z, then
 * y += alpha A x.  Per element: round(beta z) + round(alpha (A x)_i), one addition -- the bits of the two-pass form.                       */
int vexhip_spmat_apply_axpby_f64(const vexhip_spmat *A, void *stream, double alpha, const double *x, double beta, const double *z, double *y);
int vexhip_spmat_apply_axpby_f32(const vexhip_spmat *A, void *stream, float alpha, const float *x, float beta, const float *z, float *y);
/* 1: that call runs as ONE pass on these vectors; 0: as two (vex::SpMat then keeps its own general route: vexcl/spmat.hpp apply_axpby) */
int vexhip_spmat_axpby_fused(const vexhip_spmat *A, const void *x, const void *z, const void *y);
int vexhip_spmat_apply_f32(const vexhip_spmat *A, void *stream, float alpha, int append, const float *x, float *y);
/* Y[k] (=|+=) alpha * A * X[k], k < nrhs, reading the matrix once per group of four (x, y: HOST arrays of device pointers) */
int vexhip_spmat_apply_multi_f64(const vexhip_spmat *A, void *stream, int nrhs, double alpha, int append, const double *const *x, double *const *y);
int vexhip_spmat_apply_multi_f32(const vexhip_spmat *A, void *stream, int nrhs, float alpha, int append, const float *const *x, float *const *y);
int vexhip_spmat_get_info(const vexhip_spmat *A, vexhip_spmat_info *info);

/* ---- partition set-up on the device (vexcl/spmat.hpp:291-378, spmat/csr.inl:92-131, hybrid_ell.inl:132-136) -------
 * One device's row strip, in HBM with GLOBAL column ids, is split into the LOCAL part (columns inside [col_begin,
 * col_end), renumbered c - col_begin), the REMOTE part as a row-subset CSR (only the rows that reach a column outside
 * the range; columns renumbered to their rank in the ghost set) and the sorted ghost set -- the reference builds all
 * of this on the host from a std::set per device.  Two calls: sizes() -> {local nnz, remote nnz, rows with remote
 * entries, -1}; the caller allocates lptr[n+1], lcol/lval[local nnz], rem_rows[rows], rem_ptr[rows+1],
 * rem_col/rem_val[remote nnz], ghosts[remote nnz] (capacity); split() fills them and sets sizes[3] = ghost count.   */
int vexhip_csr_split_sizes_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col,
        int64_t col_begin, int64_t col_end, int64_t *sizes);
int vexhip_csr_split_f64_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const double *val,
        int64_t col_begin, int64_t col_end, int64_t *sizes, int32_t *lptr, int32_t *lcol, double *lval,
        int32_t *rem_rows, int32_t *rem_ptr, int32_t *rem_col, double *rem_val, int32_t *ghosts);
int vexhip_csr_split_f32_i32(int dev, void *stream, int64_t n, const int32_t *ptr, const int32_t *col, const float *val,
        int64_t col_begin, int64_t col_end, int64_t *sizes, int32_t *lptr, int32_t *lcol, float *lval,
        int32_t *rem_rows, int32_t *rem_ptr, int32_t *rem_col, float *rem_val, int32_t *ghosts);

/* The strip of one device WITH its ghost planes as one square matrix -- the operand of the one-launch step below (the reference keeps
 * the remote columns in a second matrix, vexcl/spmat.hpp:291-378): `lo` empty rows, the strip's n rows, `hi` empty rows; columns
 * col - (col_begin - lo), i.e. counted from the first element of the lower ghost plane.  ptr_ext[lo + n + hi + 1], col_ext[nnz]
 * (values are shared with the strip).  *out_of_range = entries whose column lies outside the two ghost planes (the caller declines). */
int vexhip_csr_extend_halo_i32(int dev, void *stream, int64_t n, int64_t nnz, const int32_t *ptr, const int32_t *col, int64_t col_begin,
        int64_t lo, int64_t hi, int32_t *ptr_ext, int32_t *col_ext, int64_t *out_of_range);

/* ---- RCCL transport over xGMI -------------------------------------------------------------
 * Replaces the host-staged ghost exchange of vexcl/spmat.hpp:125-183 / sparse/distributed.hpp:347-428 (device ->
 * host -> device, four finish() fences), the host fold of the Reductor partials (reductor.hpp:412-436) and the host
 * carry of multi-device scans (scan.hpp:445-457).  librccl is loaded on first use.
 * A communicator spans the devices of ONE process (vexhip_comm_init: the reference's model, one Context drives every
 * GPU; every call then takes arrays with one entry per local device and issues ONE RCCL group) or is one rank of a
 * one-process-per-GPU job (vexhip_comm_init_rank; rank 0 makes the 128-byte id, the launcher distributes it).        */
typedef struct vexhip_comm vexhip_comm;
int vexhip_comm_unique_id(void *id128);
enum { VEXHIP_COMM_AUTO = 0,   /* RCCL when the devices are distinct GPUs and there are at least two, else PEER          */
       VEXHIP_COMM_RCCL = 1,   /* grouped ncclSend / ncclRecv, ncclAllReduce, ncclAllGather                               */
       VEXHIP_COMM_PEER = 2,   /* single process only: event-ordered device-to-device copies (no communicator; the only  *
                                * option when logical devices share one GPU -- the reference's own test fixture,           *
                                * tests/context_setup.hpp:24-39 -- and the host fold of the reference for reductions)      */
       VEXHIP_COMM_IPC = 3,    /* one process per GPU: peer-mapped ghost windows (vexhip_ipc_window_*), reported by       *
                                * vexhip_dist_spmv_status; not a value for vexhip_comm_init                               */
       VEXHIP_COMM_HALO = 4 }; /* the same windows, the whole step in ONE product launch (vexhip_dist_spmv_create_halo)   */
int vexhip_comm_init(int ndev, const int *devs, int transport, vexhip_comm **out);
int vexhip_comm_init_rank(int dev, int rank, int world, const void *id128, vexhip_comm **out);
int vexhip_comm_destroy(vexhip_comm *comm);
int vexhip_comm_size(const vexhip_comm *comm, int *world, int *nlocal, int *transport);
/* Ghost exchange: local device d sends send_counts[d*world + p] elements to rank p, taken from send_bufs[d] in rank
 * order (contiguous), and receives recv_counts[d*world + p] from it into recv_bufs[d] in rank order: grouped
 * ncclSend / ncclRecv on streams[d].  A rank's share for itself is a device copy.                                  */
int vexhip_halo_exchange(vexhip_comm *comm, int dtype, const void *const *send_bufs, const int64_t *send_counts,
        void *const *recv_bufs, const int64_t *recv_counts, void *const *streams);
/* In-place all-reduce of `count` elements per device (op: VEXHIP_SUM / SUM_KAHAN -> sum, MIN, MAX) */
int vexhip_allreduce_scalar(vexhip_comm *comm, int op, int dtype, void *const *bufs, int64_t count, void *const *streams);
/* recv[d] = concatenation over ranks of their `count` elements */
int vexhip_allgather(vexhip_comm *comm, int dtype, const void *const *send, void *const *recv, int64_t count, void *const *streams);

/* One rank's product step of a row-partitioned SpMat, issued from C++: pack (gather of the owned values the peers
 * need) -> grouped send/recv on a second stream -> local part (overlapped) -> remote part after the receive
 * (the five phases of spmat.hpp:125-183 with one xGMI hop).  All arrays are device memory owned by the caller:
 * local = vexhip_spmat of the owned columns (NULL: none); remote part = row-subset CSR over the ghost buffer
 * (rows_idx[rem_rows], rem_ptr[rem_rows + 1], columns = positions in ghost_buf); send_idx[nsend] = local ids packed
 * into send_buf in peer order; send_counts / recv_counts[world].  set_graph(1): a step WITHOUT exchange (one rank, or
 * a block-diagonal partition) is captured into a hipGraph on first use and replayed while the operands stay the same
 * (needs a non-default stream); steps with RCCL operations are always issued directly.                              */
typedef struct vexhip_dist_spmv vexhip_dist_spmv;
int vexhip_dist_spmv_create(vexhip_comm *comm, int dtype, int64_t rows, const vexhip_spmat *local,
        int64_t rem_rows, const int32_t *rows_idx, const int32_t *rem_ptr, const int32_t *rem_col, const void *rem_val,
        int64_t nsend, const int32_t *send_idx, void *send_buf, const int64_t *send_counts,
        int64_t nghost, void *ghost_buf, const int64_t *recv_counts, vexhip_dist_spmv **out);
int vexhip_dist_spmv_destroy(vexhip_dist_spmv *step);
int vexhip_dist_spmv_set_graph(vexhip_dist_spmv *step, int enable);
int vexhip_dist_spmv_apply(vexhip_dist_spmv *step, void *stream, double alpha, int append, const void *x, void *y);
/* What RCCL itself reports for the communicator of local device 0 (ncclCommCount / ncclCommCuDevice / ncclCommUserRank);
 * for the PEER transport: the values the communicator was built with.                                                  */
int vexhip_comm_rccl_info(const vexhip_comm *comm, int *nranks, int *device, int *user_rank);

/* The same product step over PEER-MAPPED GHOST WINDOWS instead of a communicator (second transport of the
 * one-process-per-GPU job; replaces the host staging of vexcl/spmat.hpp:125-183 like the RCCL step does).  Every rank
 * allocates one uncached window (its ghost values + two arrays of 64-bit step counters), exports it
 * (hipIpcGetMemHandle; the launcher distributes the 64-byte handles) and opens the windows of the ranks it exchanges
 * with.  Per product ONE kernel per rank writes every neighbour's share straight into that neighbour's window over
 * xGMI and raises `arrive` there; the consumer's stream waits on its own flags with a one-wave kernel, runs the remote
 * part on the window and raises `consumed` at the owners, which gates their next write.  No pack buffer, no receive,
 * no collective kernel.  Spins are bounded (20 s; VEXHIP_IPC_TIMEOUT_MS): a wait that runs out is an ERROR -- the ghost
 * values are overwritten with NaN (the remote part can not turn stale ghosts into plausible numbers), a flag in pinned
 * host memory is set, and vexhip_dist_spmv_apply / _profile fail from then on (vexhip_dist_spmv_status reports it too).
 * Several plans may share a window (each counts its own launches).  dst_offsets[p] = element offset of this rank's
 * share in rank p's ghost set.                                                                                         */
typedef struct vexhip_ipc_window vexhip_ipc_window;
int vexhip_ipc_window_create(int dev, int rank, int world, int64_t data_bytes, vexhip_ipc_window **out);
int vexhip_ipc_window_export(const vexhip_ipc_window *win, void *handle64);
int vexhip_ipc_window_open(vexhip_ipc_window *win, int peer, const void *handle64);
/* ONE process driving every GPU (vex::Context; the reference's model, vexcl/spmat.hpp:120-185): the peer's window is an address
 * this process already holds -- no handle; distinct GPUs get hipDeviceEnablePeerAccess (which also covers the vectors a pull
 * step reads in place).  `peer_win` must be rank `peer` of the same world.                                                       */
int vexhip_ipc_window_attach(vexhip_ipc_window *win, int peer, const vexhip_ipc_window *peer_win);
/* Any device allocation shown to another process (the pull step BETWEEN processes reads the neighbours' x in place): the 64-byte handle of
 * the allocation that holds `ptr` and ptr's offset in it; the peer opens the handle (base of the mapping) and closes it when done.   */
int vexhip_ipc_export(int dev, const void *ptr, void *handle64, int64_t *offset);
int vexhip_ipc_open(int dev, const void *handle64, void **base);
int vexhip_ipc_close(int dev, void *base);
int vexhip_ipc_window_data(const vexhip_ipc_window *win, void **data);
int vexhip_ipc_window_destroy(vexhip_ipc_window *win);
int vexhip_dist_spmv_create_ipc(vexhip_ipc_window *win, int dtype, int64_t rows, const vexhip_spmat *local,
        int64_t rem_rows, const int32_t *rows_idx, const int32_t *rem_ptr, const int32_t *rem_col, const void *rem_val,
        int64_t nsend, const int32_t *send_idx, const int64_t *send_counts, const int64_t *dst_offsets,
        int64_t nghost, const int64_t *recv_counts, vexhip_dist_spmv **out);
/* Round 5 -- the step as ONE launch (replaces all five phases of vexcl/spmat.hpp:120-185 for plane partitions of a 7-point
 * operator on 512-point lines): `ext` is the rank's strip stored as ONE grid matrix that keeps the entries reaching into the
 * neighbours' planes -- (has lower neighbour ? `halo` empty rows : none) + the rank's rows + (has upper ? `halo` empty rows :
 * none), columns counted from the first element of the lower ghost plane -- and must have come out of vexhip_spmat_create_*
 * with a plane plan.  The window holds [lower ghost plane | upper ghost plane] (2 * halo * 8 bytes); lower / upper are the
 * neighbours' ranks (-1: none; their windows opened).  vexhip_dist_spmv_apply then issues the plane product whose first
 * workgroups copy the rank's first / last plane of x into the neighbours' windows and whose other workgroups read the ghost
 * planes from the rank's own window behind the owners' `arrive` flags, followed by one single-thread kernel that raises
 * `consumed`: no second stream, no event, no remote part, two launches per product.  x and y are the rank's own segments.
 * The plan owns the window (no other plan on it).  Timeouts as above (NaN ghosts, sticky error).                          */
int vexhip_dist_spmv_create_halo(vexhip_ipc_window *win, const vexhip_spmat *ext, int64_t rows, int64_t halo, int lower, int upper,
        vexhip_dist_spmv **out);
/* Round 6 -- the PULL form of the one-launch step, for ONE process that drives every GPU (vex::SpMat on a multi-GPU vex::Context,
 * vexcl/spmat.hpp; replaces the five phases of /root/reference/vexcl/spmat.hpp:120-185 and the set-up of :291-378): nothing is
 * copied -- the planes next to a ghost plane read the NEIGHBOURS' boundary planes of x where they lie (peer access), passed with
 * every product: x_below = the lower neighbour's LAST `halo` elements of its segment, x_above = the upper neighbour's FIRST ones
 * (NULL where the plan has no neighbour).  order = VEXHIP_PULL_FLAGS: the windows (data_bytes 0, vexhip_ipc_window_attach) carry
 * "x is final" (raised by the first workgroup of the owner's launch) and `consumed` (raised behind the launch; the same kernel
 * waits for the neighbours' `consumed`, so that nothing behind it in the stream overwrites a plane still being read).
 * order = VEXHIP_PULL_EVENTS: no flag, `win` may be NULL -- the CALLER orders the devices' streams with events (logical devices
 * sharing one GPU may share a hardware queue, where a launch that waits for a later launch would never end).  Time-outs as above.
 * `ext` may have a plane plan, a grid plan (any line length, fp64 or fp32) or -- no geometry at all -- be stored as SELL-512 with
 * diagonal codes (storage "sell8" / "sell8v": a value per entry or value codes), provided no diagonal reaches further than `halo`
 * elements beyond the rank's rows: any banded operator whose band fits one ghost range, e.g. the variable-coefficient 7-point
 * problem.  The launch is then the pair product in its HALO role (csrc/sell8.hip: the slices that touch a ghost range run last and
 * translate a column into [x_below | x | x_above]).  Anything else: a non-zero return with the reason in vexhip_last_error(); the caller keeps the five-phase step.    */
enum { VEXHIP_PULL_FLAGS = 1, VEXHIP_PULL_EVENTS = 2 };
int vexhip_dist_spmv_create_halo_pull(vexhip_ipc_window *win, const vexhip_spmat *ext, int64_t rows, int64_t halo, int lower, int upper,
        int order, vexhip_dist_spmv **out);
int vexhip_dist_spmv_apply_pull(vexhip_dist_spmv *step, void *stream, double alpha, int append, const void *x, void *y,
        const void *x_below, const void *x_above);
/* Diagnostics of the one-launch step (VEXHIP_HALO_DEBUG=1 in the environment when the plan is created): six 64-bit words per
 * workgroup of the last launch -- start, ghost flag seen, first ghost line in registers, end (100 MHz ticks), first plane, end
 * plane (push workgroups: ~0, side) -- copied to `out` (at most 4096 workgroups).  tools/r05_halo_timeline.py.                    */
int vexhip_dist_spmv_debug(vexhip_dist_spmv *step, void *out, int64_t bytes);
/* timed_out: a flag wait ran into its bound; transport: VEXHIP_COMM_RCCL, VEXHIP_COMM_IPC or VEXHIP_COMM_HALO; direct: the shares are
 * runs of x and no pack kernel / index list is used.                                                                 */
int vexhip_dist_spmv_status(vexhip_dist_spmv *step, int *timed_out, int *transport, int *direct);
/* One product with its phases timed (HIP events on both streams; synchronises): ms6 = total, local part, wait for the
 * ghosts after the local part, remote part, pack, exchange (the last two run beside the local part).                 */
int vexhip_dist_spmv_profile(vexhip_dist_spmv *step, void *stream, double alpha, int append, const void *x, void *y, float *ms6);

/* Multi-right-hand-side products  y[k] (+)= alpha * A * x[k],  k < nrhs  -- `SpMat * multivector`
 * (vexcl/spmat.hpp:388-398, which applies the product once per component; tests/spmv.cpp:262-305).
 * One launch per group of up to four right-hand sides reads the matrix ONCE; each y[k] is
 * bit-identical to the corresponding vexhip_spmv_sell*_ call.  x and y are HOST arrays of
 * nrhs device pointers; every other argument is as in the single-vector calls.               */
int vexhip_spmm_sell8_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *const *x, double *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell8_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *const *x, float *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell8v_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const double *values, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *const *x, double *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell8v_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t ell_width,
        const void *buf, const int32_t *deltas, const float *values, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *const *x, float *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell_f64_i32(int dev, void *stream, int64_t n, int nrhs, double alpha, int append, int64_t ell_width,
        const void *sell, const int32_t *csr_ptr, const int32_t *csr_col, const double *csr_val,
        const double *const *x, double *const *y, const vexhip_traversal *traversal);
int vexhip_spmm_sell_f32_i32(int dev, void *stream, int64_t n, int nrhs, float alpha, int append, int64_t ell_width,
        const void *sell, const int32_t *csr_ptr, const int32_t *csr_col, const float *csr_val,
        const float *const *x, float *const *y, const vexhip_traversal *traversal);

/* CSR -> hybrid ELL conversion on the device (sparse/ell.hpp:400-508,
 * `convert_csr2ell` :348-397; width rule hybrid_ell.inl:66-114).
 * Step 1 (blocking): row-width histogram -> ELL width by the reference's
 * "3 x rows-wider-than-w < n" rule, and the nnz of the CSR tail.
 * Step 2: fill caller-allocated ell_col/ell_val (pitch*width each, pitch =
 * alignup(n,16)) and csr_ptr (n+1) / csr_col / csr_val (tail_nnz).            */
int vexhip_hell_analyze_i32(int dev, void *stream, int64_t n, const int32_t *ptr,
        int64_t *ell_width, int64_t *tail_nnz);
int vexhip_hell_fill_f64_i32(int dev, void *stream, int64_t n,
        const int32_t *ptr, const int32_t *col, const double *val,
        int64_t ell_width, int64_t ell_pitch, int32_t *ell_col, double *ell_val,
        int32_t *csr_ptr, int32_t *csr_col, double *csr_val);
int vexhip_hell_fill_f32_i32(int dev, void *stream, int64_t n,
        const int32_t *ptr, const int32_t *col, const float *val,
        int64_t ell_width, int64_t ell_pitch, int32_t *ell_col, float *ell_val,
        int32_t *csr_ptr, int32_t *csr_col, float *csr_val);

/* ---- gather for the ghost exchange (spmat.hpp:129-133 `permutation(cols)(x)`,
 *      sparse/distributed.hpp:352-398 `vexcl_sparse_gather`) --------------- */
int vexhip_gather_f64_i32(int dev, void *stream, int64_t n, const int32_t *idx, const double *src, double *dst);
int vexhip_gather_f32_i32(int dev, void *stream, int64_t n, const int32_t *idx, const float *src, float *dst);

/* ---- Reductor (reductor.hpp:302-439) ------------------------------------
 * Two-stage, both stages on the device; result lands in *out_dev (device
 * memory, one element of the value type; MIN_MAX writes two).  `tmp` must hold
 * vexhip_reduce_tmp_bytes() bytes.  ops: */
enum { VEXHIP_SUM = 0, VEXHIP_SUM_KAHAN = 1, VEXHIP_MIN = 2, VEXHIP_MAX = 3, VEXHIP_MIN_MAX = 4 };
/* dtypes shared by reduce / scan / sort / fill: */
enum { VEXHIP_F64 = 0, VEXHIP_F32 = 1, VEXHIP_I32 = 2, VEXHIP_U32 = 3, VEXHIP_I64 = 4, VEXHIP_U64 = 5 };
size_t vexhip_reduce_tmp_bytes(void);
int vexhip_reduce(int dev, void *stream, int op, int dtype, const void *in, int64_t n, void *out_dev, void *tmp);
/* sum(a*b), the reduce section of examples/benchmark.cpp:224-246 */
int vexhip_reduce_dot(int dev, void *stream, int dtype, const void *a, const void *b, int64_t n, void *out_dev, void *tmp);
/* stage 2 alone: fold `nparts` partials produced by a JIT-generated stage-1
 * kernel (reductor.hpp:412-436 does this fold on the host).                   */
int vexhip_reduce_finish(int dev, void *stream, int op, int dtype, const void *partials, int64_t nparts, void *out_dev);
/* number of partials (= workgroups) a JIT stage-1 kernel should produce       */
int vexhip_reduce_num_groups(int dev, int *groups, int *block);

/* ---- scan (scan.hpp:66-414: inclusive_scan / exclusive_scan with vex::plus)
 * exclusive != 0: out[i] = init + in[0] + ... + in[i-1].  In-place allowed.
 * `init` points to ONE host element of the dtype (ignored for inclusive).     */
size_t vexhip_scan_tmp_bytes(int dtype, int64_t n);
/* integer scans use a single-pass decoupled look-back kernel (1 read + 1 write per
 * element); 0 forces the deterministic reduce-then-scan path (A/B, tests)       */
int vexhip_scan_set_lookback(int enable);
int vexhip_scan(int dev, void *stream, int dtype, int exclusive, const void *init_host,
        const void *in, void *out, int64_t n, void *tmp);

/* ---- sort (sort.hpp:2158-2182: vex::sort / vex::sort_by_key, stable) -----
 * Stable LSD radix sort, ascending (descending != 0: vex::greater<T>).
 * keys are sorted in place; keys_tmp (n keys) and, for pairs, vals_tmp are
 * ping-pong buffers; tmp holds vexhip_sort_tmp_bytes().  value_bytes in {0,4,8}. */
size_t vexhip_sort_tmp_bytes(int key_dtype, int64_t n);
/* How the scatter ranks the keys of a tile: -1 / 6 (default) = half-wave ranking units, one returning 64-bit LDS atomic per key whose
 * return carries the rank AND the lanes served before it -- a lane served out of lane order is SEEN, and the tile is then ranked again
 * by ballots inside the same launch (counted: vexhip_sort_status); 0 = ranks from match words in every tile (ordered by construction:
 * no assumption about the order in which the LDS serves the lanes of one atomic; 25 % slower; A/B, tests); 7 = the default with
 * every tile taking the ballot path (tests of that path).                                                                          */
int vexhip_sort_set_rank(int mode);
int vexhip_sort(int dev, void *stream, int key_dtype, int descending,
        void *keys, void *keys_tmp, int value_bytes, void *vals, void *vals_tmp,
        int64_t n, void *tmp);
/* What the last vexhip_sort of n keys on the workspace `tmp` met (waits for the stream): tiles ranked a second time because the LDS
 * served a lane out of lane order (the result is correct all the same), tiles dropped because their keys no longer matched the
 * pass's histogram -- the caller changed the input while the sort ran; then the call FAILS (vexhip_last_error).  Neither traps. */
int vexhip_sort_status(int dev, void *stream, int64_t n, const void *tmp, int64_t *reranked_tiles, int64_t *dropped_tiles);

/* ---- compressed-stencil SpMV (spmat/ccsr.hpp:40-53,184-200; vex::SpMatCCSR) ----
 * y[i] (=|+=) alpha * sum_{j in [row[idx[i]], row[idx[i]+1])} val[j] * x[i + col[j]];
 * m unique rows, `entries` = row[m] table entries (device arrays; idx, row 32-bit).
 * far_offset: the largest |col| shared by most rows (0 if unknown) -- used only to
 * pick the strip traversal that keeps x in one XCD's L2.  Caller guarantees that
 * every i + col[j] addressed lies inside x.                                          */
int vexhip_spmv_ccsr_f64(int dev, void *stream, int64_t n, double alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const double *val, int64_t entries, int64_t far_offset,
        const double *x, double *y);
int vexhip_spmv_ccsr_f32(int dev, void *stream, int64_t n, float alpha, int append, const uint32_t *idx, int64_t m,
        const uint32_t *row, const int32_t *col, const float *val, int64_t entries, int64_t far_offset,
        const float *x, float *y);
/* CCSR -> CSR on the device (so that vex::SpMatCCSR can hand its operator to vexhip_spmat).  Two calls: with out_col = NULL
 * it writes the row pointers ptr[n + 1] and *nnz; with out_col / out_val (nnz entries each) it fills them in table order
 * (the CCSR product's summation order) and sets *nnz = -1 if an entry refers to a column outside [0, n), else 0.          */
int vexhip_ccsr_to_csr_f64_i32(int dev, void *stream, int64_t n, const uint32_t *idx, const uint32_t *row, const int32_t *col, const double *val,
        int32_t *ptr, int32_t *out_col, double *out_val, int64_t *nnz);
int vexhip_ccsr_to_csr_f32_i32(int dev, void *stream, int64_t n, const uint32_t *idx, const uint32_t *row, const int32_t *col, const float *val,
        int32_t *ptr, int32_t *out_col, float *out_val, int64_t *nnz);
int vexhip_spmv_ccsr_set_rows_per_lane(int rows);    /* 0 (default): pair form, rows 2t and 2t+1 per lane with 16-byte x loads; 1, 2, 4, 8: rows per lane of the first form (A/B) */

/* ---- stencil convolution (stencil.hpp:306-405 `slow_conv` / `fast_conv`) ----
 * y[i] = beta*y[i] + alpha * sum_{j=0}^{lhalo+rhalo} s[j] * X(i + j - lhalo), where
 * X(g) = x[g] inside [0,n); outside it reads the halo buffer xrem (lhalo values of the
 * left neighbour, then rhalo values of the right one) when has_left / has_right, else
 * the edge element x[0] / x[n-1] (stencil.hpp:264-290).  LDS-staged.               */
int vexhip_stencil_conv_f64(int dev, void *stream, int64_t n, int has_left, int has_right, int lhalo, int rhalo,
        const double *s, const double *x, const double *xrem, double *y, double beta, double alpha);
int vexhip_stencil_conv_f32(int dev, void *stream, int64_t n, int has_left, int has_right, int lhalo, int rhalo,
        const float *s, const float *x, const float *xrem, float *y, float beta, float alpha);

/* ---- vex::FFT (vexcl/fft.hpp:69-148; fft/plan.hpp:214-330 plan + transform) -------------------------------
 * Complex-to-complex transforms of interleaved (re, im) data, dtype VEXHIP_F32 / VEXHIP_F64 = the scalar type.
 * `sizes` are row-major (last dimension contiguous); `dirs[j]` says what happens along dimension j -- a `none`
 * dimension is a batch.  Any length: 2,3,5,7,11,13-smooth lengths run in the LDS row kernel (four-step
 * beyond one row of LDS), others through Bluestein.  Unnormalized in both directions (the header applies 1/n
 * for inverse dimensions, plan.hpp:236-241).  Out of place: `in` and `out` are distinct buffers of
 * prod(sizes) complex elements on the plan's device.                                                        */
enum { VEXHIP_FFT_FORWARD = 0, VEXHIP_FFT_INVERSE = 1, VEXHIP_FFT_NONE = 2 };      /* fft::direction, plan.hpp:49-53 */
size_t vexhip_fft_best_size(size_t n);     /* fft::planner::best_size: smallest 2^a 3^b 5^c 7^d >= n (plan.hpp:126-128) */
int vexhip_fft_plan_create(int dev, int dtype, int ndim, const size_t *sizes, const int *dirs, void **plan);
int vexhip_fft_plan_destroy(void *plan);
int vexhip_fft_exec(void *plan, void *stream, const void *in, void *out);
/* what the plan consists of: LDS row passes, transposes, other launches (Bluestein, copy) */
int vexhip_fft_plan_steps(void *plan, int *row_passes, int *transposes, int *others);

/* ---- vex::mba (vexcl/mba.hpp:233-330: lattice hierarchy fitted on the host in the reference) ------------------
 * Multilevel B-spline fit of `npts` scattered points (coo: npts x ndim row-major, val: npts, both on the device;
 * val is overwritten with the final residuals) over the domain [cmin, cmax], starting from control grid `grid`,
 * at most `levels` levels, stopping when the squared residual falls below tol * its initial value.  ndim 1..3.
 * Outputs the final lattice: origin, inverse spacing, extents, strides and a device buffer of the control values
 * (allocated here, released by the caller with vexhip_free).                                                     */
int vexhip_mba_fit(int dev, void *stream, int dtype, int ndim, const double *cmin, const double *cmax,
        const void *coo, void *val, int64_t npts, const size_t *grid, int levels, double tol,
        double *xmin, double *hinv, size_t *n, size_t *stride, void **phi, size_t *phi_elems);

/* ---- benchmark input generators (examples/benchmark.cpp:364-415; SURVEY
 *      section 8(d): 512^3 is built on the device, never uploaded) ---------- */
int64_t vexhip_poisson3d_nnz(int64_t n);
int vexhip_poisson3d_csr_f64_i32(int dev, void *stream, int64_t n, int32_t *ptr, int32_t *col, double *val);
/* rows [row_begin,row_end) of the same matrix with GLOBAL column ids and a
 * strip-local ptr (ptr[0] = 0): the per-rank strip of the 8-GPU run.          */
int vexhip_poisson3d_strip_f64_i32(int dev, void *stream, int64_t n, int64_t row_begin, int64_t row_end,
        int32_t *ptr, int32_t *col, double *val);
int64_t vexhip_poisson3d_strip_nnz(int64_t n, int64_t row_begin, int64_t row_end);
/* the same strip with 64-bit row pointers: grids with 2^31 entries and more (700^3: 2.38e9); columns stay 32-bit */
int vexhip_poisson3d_strip_f64_p64(int dev, void *stream, int64_t n, int64_t row_begin, int64_t row_end,
        int64_t *ptr, int32_t *col, double *val);
/* The same 7-point pattern with a different coefficient on every face: -div(k grad u), k = 0.5 + u(hash(seed, face)),
 * u in [0,1) -- about 4 N distinct values (every coupling appears in the two rows it joins), what a finite-volume code assembles (no value coding applies).  Same row strip
 * convention and nnz as the Poisson generator; restated on the host in oracle/vex_oracle.c (bit-identical values).   */
int vexhip_diffusion3d_strip_f64_i32(int dev, void *stream, int64_t n, int64_t row_begin, int64_t row_end, uint64_t seed,
        int32_t *ptr, int32_t *col, double *val);
/* counter-hash pseudo-random fill (same hash on host: tests restate it):
 * u32: full range; f64/f32: U[0,1).                                           */
int vexhip_fill_hash(int dev, void *stream, int dtype, uint64_t seed, void *out, int64_t n);
int vexhip_fill_value(int dev, void *stream, int dtype, const void *value_host, void *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* VEXHIP_H */
