"""Round 2: where does the value-coded product spend its time?  JIT variants of ONE branch-free kernel with one
ingredient changed at a time (several produce WRONG results on purpose -- they only answer "what would it cost
without X").  Diagnostic; the product path is libvexhip's kernel.  Output: gpurun_out/r02_sell8v_ablation.json"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
L = lib(); dev = torch.device("cuda:0")
n = 512
ptr, col, val = ops.poisson3d(n, device=dev)
N = n ** 3
S = ops.SlicedELL(ptr, col, val)
del ptr, col, val
# x sits in the middle of a larger array: the variants that use the wrong codes on purpose read up to n^2 elements outside it
xbig = ops.fill_hash(torch.empty(N + 2 * n * n + 1024, dtype=torch.float64, device=dev), 1)
x = xbig[n * n + 512: n * n + 512 + N]; y = torch.empty(N, dtype=torch.float64, device=dev)
yref = torch.empty_like(x); S.mul(x, yref)

SRC = r'''
#define W 7
#define WP 4
typedef double d2 __attribute__((ext_vector_type(2)));
struct trav { int chunk, planes, plane_blocks; };
__device__ inline long long slot_of(const trav t, long long nblocks, const unsigned vb) {
  if (t.chunk > 0) {
    const unsigned k = vb & 7u, q = vb >> 3, chunk = t.chunk, planes = t.planes;
    const unsigned r = q / chunk, i = q - r * chunk, tile = r / planes, p = r - tile * planes;
    const long long l = (long long)tile * (8 * chunk) + k * chunk + i, lb = (long long)p * t.plane_blocks + l;
    return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
  }
  return vb < nblocks ? (long long)vb : -1;
}
// GATHER: 0 full | 1 every tap reads x[i+q] (14 loads, one line) | 2 ONE 16-byte load of x[i..i+1], reused by every tap
//         3 seven 16-byte loads x[i+d .. i+d+1] (both rows take the lane's first code: wrong on boundary rows)
//         4 full, but lanes own rows t and t+256 (coalesced 8-byte gathers; codes read as stored: wrong results)
//         5 +-1 taps from the centre load by lane shuffles (wrong at wave edges / boundary rows), 4 far taps as 16-byte loads
// CODES:  0 loaded | 1 not loaded (synthesized: interior pattern)
// STORE:  0 nontemporal 16-byte | 1 none | 2 plain
// ROWS:   slices per workgroup (consecutive in the XCD strip), sequential
extern "C" __global__ void __launch_bounds__(256) k(long long n, long long ns, unsigned nvirtual, const char *buf, const int *deltas,
    const double *values, const double *x, double *y, trav tr) {
  __shared__ int s_delta[256]; __shared__ double s_value[256];
  s_delta[threadIdx.x] = deltas[threadIdx.x]; s_value[threadIdx.x] = values[threadIdx.x];
  __syncthreads();
  const int t = threadIdx.x;
  for (int kk = 0; kk < ROWS; ++kk) {
    const unsigned vb = (((blockIdx.x >> 3) * ROWS + kk) << 3) | (blockIdx.x & 7);
    const long long s = vb < nvirtual ? slot_of(tr, ns, vb) : -1;
    if (s < 0) continue;
#if GATHER == 4
    const long long i = s * 512 + t;  const int rstride = 256;
#else
    const long long i = s * 512 + 2 * t; const int rstride = 1;
#endif
    unsigned c[WP], vc[WP];
#if CODES == 0 || CODES == 3 || CODES == 4
    // CODES 3: every slice reads the code block of ONE interior slice (what a dictionary of distinct slice blocks would make of
    // this matrix: the block stays in L2; wrong results on the boundary slices)
#if CODES == 3 || CODES == 4
    const long long sc = 300 * 512 + 300 + (s & 7);
#else
    const long long sc = s;
#endif
    const unsigned *cw = (const unsigned *)(buf + sc * (WP * 2048ll)) + t; const unsigned *vw = cw + WP * 256;
    #pragma unroll
#if CODES == 4
    for (int jp = 0; jp < WP; ++jp) { c[jp] = cw[jp * 256]; vc[jp] = vw[jp * 256]; }      // plain (cached) loads
#else
    for (int jp = 0; jp < WP; ++jp) { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(vw + jp * 256); }
#endif
#elif CODES == 2 || CODES == 5
    { typedef unsigned u4 __attribute__((ext_vector_type(4)));
      // lane t: 16 bytes at t*16 of the first 4 KiB (its four diagonal-code words) and of the second 4 KiB (value codes)
#if CODES == 5   // ... of one of 8 fixed slices, plain (cached) loads: a lane-major pool of distinct code blocks
      const u4 *cw4 = (const u4 *)(buf + (300 * 512 + 300 + (s & 7)) * (WP * 2048ll)) + t;
      const u4 a4 = cw4[0], b4 = cw4[256];
#else
      const u4 *cw4 = (const u4 *)(buf + s * (WP * 2048ll)) + t;
      const u4 a4 = __builtin_nontemporal_load(cw4), b4 = __builtin_nontemporal_load(cw4 + 256);
#endif
      const unsigned never = (a4.x == 0xdeadbeefu) + (a4.y == 0xdeadbeefu) + (a4.z == 0xdeadbeefu) + (a4.w == 0xdeadbeefu)
                           + (b4.x == 0xdeadbeefu) + (b4.y == 0xdeadbeefu) + (b4.z == 0xdeadbeefu) + (b4.w == 0xdeadbeefu);
      c[0] = 0x01010000u + never; c[1] = 0x03030202u; c[2] = 0x05050404u; c[3] = 0xffff0606u;
      vc[0] = 0; vc[1] = 0x01010000u; vc[2] = 0; vc[3] = 0; }
#else
    c[0] = 0x01010000u; c[1] = 0x03030202u; c[2] = 0x05050404u; c[3] = 0xffff0606u;
    vc[0] = 0; vc[1] = 0x01010000u; vc[2] = 0; vc[3] = 0;
    if (t == 300) { c[0] += (unsigned)s; }   /* keep it a run-time value */
#endif
    int dl[W][2];
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) dl[j][q] = s_delta[(c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u];
    double xv[W][2];
#if GATHER == 2
    { const d2 p = *(const d2 *)(x + i);
      #pragma unroll
      for (int j = 0; j < W; ++j) { xv[j][0] = p.x + dl[j][0]; xv[j][1] = p.y + dl[j][1]; } }
#elif GATHER == 3
    #pragma unroll
    for (int j = 0; j < W; ++j) {
      const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2))) & 255u;
      const double *px = code != 255u ? x + (i + dl[j][0]) : values + 254;
      d2 p; __builtin_memcpy(&p, px, 16);
      xv[j][0] = p.x; xv[j][1] = p.y;
    }
#elif GATHER == 6 || GATHER == 7
    // uniform base (SGPR pair) + 32-bit unsigned byte offset per lane: the saddr form of global_load
    { const char *base = (const char *)(x + (s * 512 - 262144 - 512));       // below every address this slice gathers
      #pragma unroll
      for (int j = 0; j < W; ++j) {
#if GATHER == 6
        const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2))) & 255u;
        const unsigned off = code != 255u ? (unsigned)(2 * t + dl[j][0] + 262144 + 512) * 8u : 0u;
        d2 p; __builtin_memcpy(&p, base + off, 16);
        xv[j][0] = p.x; xv[j][1] = p.y;
#else
        #pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
          const unsigned off = code != 255u ? (unsigned)(2 * t + q + dl[j][q] + 262144 + 512) * 8u : 0u;
          xv[j][q] = *(const double *)(base + off);
        }
#endif
      } }
#elif GATHER == 8
    // raw buffer loads: 128-bit resource over x, 32-bit byte offset per lane
    { __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (s * 512 - 262144 - 512)), 0, (2 * 262144 + 2048) * 8, 0x00020000);
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      #pragma unroll
      for (int j = 0; j < W; ++j) {
        const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2))) & 255u;
        const unsigned off = code != 255u ? (unsigned)(2 * t + dl[j][0] + 262144 + 512) * 8u : 0u;
        const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        unsigned long long lo = ((unsigned long long)r.y << 32) | r.x, hi = ((unsigned long long)r.w << 32) | r.z;
        xv[j][0] = __builtin_bit_cast(double, lo); xv[j][1] = __builtin_bit_cast(double, hi);
      } }
#elif GATHER == 9
    // x for EVERY diagonal of the table (7 here: uniform offsets, no dependence on the slice's codes) loaded up front with
    // 16-byte loads, parked in a lane-private LDS slot (no barrier) and picked by code afterwards
    { __shared__ d2 xs[W][256];
      d2 xr[W];
      #pragma unroll
      for (int j = 0; j < W; ++j) __builtin_memcpy(&xr[j], x + (i + deltas[j]), 16);
      #pragma unroll
      for (int j = 0; j < W; ++j) xs[j][t] = xr[j];
      #pragma unroll
      for (int j = 0; j < W; ++j)
      #pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
          const double v = ((const double *)&xs[code < 254u ? code : 0][t])[q];
          xv[j][q] = code < 254u ? v : 0.0;
        } }
#elif GATHER == 5
    { const d2 p = *(const d2 *)(x + i);
      const double left = __shfl_up(p.y, 1, 64), right = __shfl_down(p.x, 1, 64);
      xv[2][0] = left; xv[2][1] = p.x; xv[3][0] = p.x; xv[3][1] = p.y; xv[4][0] = p.y; xv[4][1] = right;
      #pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= 2 && j <= 4) continue;
        const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2))) & 255u;
        const double *px = code != 255u ? x + (i + dl[j][0]) : values + 254;
        d2 q2; __builtin_memcpy(&q2, px, 16);
        xv[j][0] = q2.x; xv[j][1] = q2.y;
      } }
#else
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
#if GATHER == 1
        const double *px = code != 255u ? x + (i + q + (dl[j][q] & 0)) : values + 255;
#else
        const double *px = code != 255u ? x + (i + q * rstride + dl[j][q]) : values + 255;
#endif
        xv[j][q] = *px;
      }
#endif
    double a[W][2];
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int sh = 8 * ((j & 1) * 2 + q);
        const unsigned code = (c[j >> 1] >> sh) & 255u;
        a[j][q] = s_value[code != 255u ? (vc[j >> 1] >> sh) & 255u : 255u];
      }
    double sum[2] = {0, 0};
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) sum[q] += a[j][q] * xv[j][q];
#if GATHER == 4
    if (i + 256 < n) {
#if STORE == 1
      if (sum[0] + sum[1] == 12345.678) y[i] = sum[0];
#else
      __builtin_nontemporal_store(sum[0], y + i); __builtin_nontemporal_store(sum[1], y + i + 256);
#endif
    }
#else
    if (i + 1 < n) {
      d2 o; o.x = sum[0]; o.y = sum[1];
#if STORE == 0
      __builtin_nontemporal_store(o, (d2 *)(y + i));
#elif STORE == 1
      if (sum[0] + sum[1] == 12345.678) y[i] = sum[0];
#else
      *(d2 *)(y + i) = o;
#endif
    }
#endif
  }
}
'''

SRC_PIPE = r"""
#define W 7
#define WP 4
typedef double d2 __attribute__((ext_vector_type(2)));
struct trav { int chunk, planes, plane_blocks; };
__device__ inline long long slot_of(const trav t, long long nblocks, const unsigned vb) {
  if (t.chunk > 0) {
    const unsigned k = vb & 7u, q = vb >> 3, chunk = t.chunk, planes = t.planes;
    const unsigned r = q / chunk, i = q - r * chunk, tile = r / planes, p = r - tile * planes;
    const long long l = (long long)tile * (8 * chunk) + k * chunk + i, lb = (long long)p * t.plane_blocks + l;
    return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
  }
  return vb < nblocks ? (long long)vb : -1;
}
// resident grid; per slice: seven 16-byte gathers (both rows take the lane's first code: timing only), the NEXT slice's
// codes are loaded behind the gathers
extern "C" __global__ void __launch_bounds__(256) kp(long long n, long long ns, unsigned nvirtual, const char *buf, const int *deltas,
    const double *values, const double *x, double *y, trav tr) {
  __shared__ int s_delta[256]; __shared__ double s_value[256];
  const int t = threadIdx.x;
  unsigned vb = blockIdx.x;
  long long s = -1;
  for (; vb < nvirtual; vb += gridDim.x) { s = slot_of(tr, ns, vb); if (s >= 0) break; }
  unsigned c[WP], vc[WP];
  { const unsigned *cw = (const unsigned *)(buf + (s < 0 ? 0 : s) * (WP * 2048ll)) + t;
    #pragma unroll
    for (int jp = 0; jp < WP; ++jp) { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(cw + (WP + jp) * 256); } }
  s_delta[t] = deltas[t]; s_value[t] = values[t];
  __syncthreads();
  if (s < 0) return;
  for (;;) {
    const long long i = s * 512 + 2 * t;
    int dl[W];
    #pragma unroll
    for (int j = 0; j < W; ++j) dl[j] = s_delta[(c[j >> 1] >> (16 * (j & 1))) & 255u];
    d2 xv[W];
    #pragma unroll
    for (int j = 0; j < W; ++j) {
      const unsigned code = (c[j >> 1] >> (16 * (j & 1))) & 255u;
      const double *px = code < 254u ? x + (i + dl[j]) : values + 254;
      __builtin_memcpy(&xv[j], px, 16);
    }
    long long sn = -1;
    for (vb += gridDim.x; vb < nvirtual; vb += gridDim.x) { sn = slot_of(tr, ns, vb); if (sn >= 0) break; }
    unsigned cn[WP], vn[WP];
#if PIPE
    { const unsigned *cw = (const unsigned *)(buf + (sn >= 0 ? sn : s) * (WP * 2048ll)) + t;
      __builtin_amdgcn_sched_barrier(0);
      #pragma unroll
      for (int jp = 0; jp < WP; ++jp) { cn[jp] = __builtin_nontemporal_load(cw + jp * 256); vn[jp] = __builtin_nontemporal_load(cw + (WP + jp) * 256); }
      __builtin_amdgcn_sched_barrier(0); }
#endif
    double sum[2] = {0, 0};
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int sh = 16 * (j & 1) + 8 * q;
        const unsigned code = (c[j >> 1] >> sh) & 255u;
        const double a = s_value[code < 254u ? (vc[j >> 1] >> sh) & 255u : 255u];
        sum[q] += a * (q ? xv[j].y : xv[j].x);
      }
    if (i + 1 < n) { d2 o; o.x = sum[0]; o.y = sum[1]; __builtin_nontemporal_store(o, (d2 *)(y + i)); }
    if (sn < 0) break;
    s = sn;
#if PIPE
    #pragma unroll
    for (int jp = 0; jp < WP; ++jp) { c[jp] = cn[jp]; vc[jp] = vn[jp]; }
#else
    { const unsigned *cw = (const unsigned *)(buf + s * (WP * 2048ll)) + t;
      #pragma unroll
      for (int jp = 0; jp < WP; ++jp) { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(cw + (WP + jp) * 256); } }
#endif
  }
}
"""
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
class Trav(ctypes.Structure):
    _fields_ = [("chunk", ctypes.c_int), ("planes", ctypes.c_int), ("plane_blocks", ctypes.c_int)]
tr = Trav(int(S.trav.chunk), int(S.trav.planes), int(S.trav.plane_blocks))
nvirtual = int(S.trav.grid_blocks); ns = (N + 511) // 512
variants = [
    ("full (branch-free, one slice per workgroup)", dict()),
    ("no y store", dict(STORE=1)),
    ("plain y store", dict(STORE=2)),
    ("codes not loaded", dict(CODES=1)),
    ("every tap reads x[i+q] (14 loads, no new lines)", dict(GATHER=1)),
    ("one 16-byte x load per lane, no gathers", dict(GATHER=2)),
    ("seven 16-byte gathers (both rows of a lane share the code)", dict(GATHER=3)),
    ("seven 16-byte gathers, codes not loaded", dict(GATHER=3, CODES=1)),
    ("seven 16-byte gathers, no store", dict(GATHER=3, STORE=1)),
    ("rows t and t+256 per lane (coalesced 8-byte gathers)", dict(GATHER=4)),
    ("+-1 taps by lane shuffle, four 16-byte far gathers", dict(GATHER=5)),
    ("+-1 taps by lane shuffle, four far gathers, codes not loaded", dict(GATHER=5, CODES=1)),
    ("seven 16-byte gathers, codes as two 16-byte loads per lane", dict(GATHER=3, CODES=2)),
    ("seven 16-byte gathers, codes as two 16-byte loads, no store", dict(GATHER=3, CODES=2, STORE=1)),
    ("seven 16-byte gathers, SGPR base + 32-bit offsets", dict(GATHER=6)),
    ("seven 16-byte gathers, SGPR base + 32-bit offsets, codes 2 x 16 B", dict(GATHER=6, CODES=2)),
    ("fourteen 8-byte gathers, SGPR base + 32-bit offsets", dict(GATHER=7)),
    ("seven 16-byte raw buffer loads", dict(GATHER=8)),
    ("seven 16-byte raw buffer loads, codes 2 x 16 B", dict(GATHER=8, CODES=2)),
    ("+-1 taps by lane shuffle, four far gathers, codes 2 x 16 B", dict(GATHER=5, CODES=2)),
    ("x of all 7 table diagonals loaded up front (16-byte, independent of the codes), picked by code via lane-private LDS", dict(GATHER=9)),
    ("x of all 7 table diagonals up front, no store", dict(GATHER=9, STORE=1)),
    ("seven 16-byte gathers, codes read from 8 fixed slices (L2-resident code blocks)", dict(GATHER=3, CODES=3)),
    ("full, codes read from 8 fixed slices (L2-resident code blocks)", dict(CODES=3)),
    ("seven 16-byte gathers, codes from 8 fixed slices with plain (L1-cached) loads", dict(GATHER=3, CODES=4)),
    ("seven 16-byte gathers, codes from 8 fixed slices, two cached 16-byte loads per lane", dict(GATHER=3, CODES=5)),
    ("+-1 taps by lane shuffle, four far gathers, codes from 8 fixed slices with cached loads", dict(GATHER=5, CODES=4)),
    ("+-1 taps by lane shuffle, four far gathers, codes two cached 16-byte loads", dict(GATHER=5, CODES=5)),
]
res = []
mods = []
for label, d in variants:
    f = dict(GATHER=0, CODES=0, STORE=0, ROWS=1); f.update(d)
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    L.module_compile(0, ("".join("#define %s %d\n" % kv for kv in f.items()) + SRC).encode(), b"-ffp-contract=off", ctypes.byref(mod))
    L.module_get_function(0, mod, b"k", ctypes.byref(fn))
    mods.append((label, f, mod, fn))
def launcher(f, fn):
    grid = (nvirtual + 8 * f["ROWS"] - 1) // (8 * f["ROWS"]) * 8
    args = [ctypes.c_longlong(N), ctypes.c_longlong(ns), ctypes.c_uint(nvirtual), ctypes.c_void_p(S.sell.data_ptr()), ctypes.c_void_p(S.deltas.data_ptr()),
            ctypes.c_void_p(S.values.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), tr]
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    return lambda keep=(args, arr): L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr)
runs = [(label, f, launcher(f, fn)) for label, f, mod, fn in mods]
for pipe in (0, 1):
    for bpc in (2, 4, 8):
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        L.module_compile(0, ("#define PIPE %d\n" % pipe + SRC_PIPE).encode(), b"-ffp-contract=off", ctypes.byref(mod))
        L.module_get_function(0, mod, b"kp", ctypes.byref(fn))
        def mk(fn=fn, g=bpc * 256):
            args = [ctypes.c_longlong(N), ctypes.c_longlong(ns), ctypes.c_uint(nvirtual), ctypes.c_void_p(S.sell.data_ptr()), ctypes.c_void_p(S.deltas.data_ptr()),
                    ctypes.c_void_p(S.values.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), tr]
            arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
            return lambda keep=(args, arr): L.launch(0, fn, g, 1, 1, 256, 1, 1, 0, stream, arr)
        runs.append(("seven 16-byte gathers, resident grid %d workgroups per CU, %s" % (bpc, "next codes prefetched behind the gathers" if pipe else "no prefetch"), dict(), mk()))
times = {label: [] for label, _, _ in runs}
same = {}
for rnd in range(3):
    for label, f, run in runs:
        y.zero_()
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        times[label].append(round(e0.elapsed_time(e1) / 30, 4))
        same[label] = bool(torch.equal(y, yref))
    t0 = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): S.mul(x, y)
    e1.record(); torch.cuda.synchronize()
    times.setdefault("library kernel", []).append(round(e0.elapsed_time(e1) / 30, 4)); same["library kernel"] = True
for label, ts in times.items():
    print("%-70s %s  best %.4f ms  identical %s" % (label, ts, min(ts), same[label]), flush=True)
    res.append({"variant": label, "ms": ts, "best_ms": min(ts), "identical": same[label]})
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r02_sell8v_ablation.json", "w"), indent=1)
