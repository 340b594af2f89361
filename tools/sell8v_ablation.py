"""Where does the SELL8V product spend its time?  JIT variants of the kernel with one ingredient removed
(diagnostic; the product path is libvexhip's sell8v_kernel)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
L = lib(); dev = torch.device("cuda:0")
n = 512
ptr, col, val = ops.poisson3d(n, device=dev)
N = n ** 3
S = ops.SlicedELL(ptr, col, val)
del ptr, col, val
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1); y = torch.empty_like(x)
SRC = r'''
#define W 7
#define WP 4
struct trav { int chunk, planes, plane_blocks; };
__device__ inline long long slot(const trav t, long long nblocks) {
  const long long b = blockIdx.x;
  if (t.chunk > 0) {
    const long long k = b & 7, q = b >> 3, i = q % t.chunk, r = q / t.chunk, p = r % t.planes, tile = r / t.planes;
    const long long l = tile * 8 * t.chunk + k * t.chunk + i, lb = p * t.plane_blocks + l;
    return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
  }
  return b < nblocks ? b : -1;
}
__device__ inline long long slot_of(const trav t, long long nblocks, const long long b) {
  if (t.chunk > 0) {
    const long long k = b & 7, q = b >> 3, i = q % t.chunk, r = q / t.chunk, p = r % t.planes, tile = r / t.planes;
    const long long l = tile * 8 * t.chunk + k * t.chunk + i, lb = p * t.plane_blocks + l;
    return (l < t.plane_blocks && lb < nblocks) ? lb : -1;
  }
  return b < nblocks ? b : -1;
}
#ifndef SPW
#define SPW 4
#endif
// MODE 9: SPW consecutive slices of this XCD's strip per workgroup, the next slice's codes loaded while the current one is processed
extern "C" __global__ void __launch_bounds__(256) kp(long long n, long long ns, long long nslots, const char *buf, const int *deltas, const double *values,
    const double *x, double *y, trav tr) {
  __shared__ int s_delta[256]; __shared__ double s_value[256];
  const int t = threadIdx.x;
  unsigned c[WP], vc[WP], cn[WP], vn[WP];
  long long s = -1, sn = -1;
  {
    const long long b = (((long long)blockIdx.x >> 3) * SPW) * 8 + (blockIdx.x & 7);
    sn = b < nslots ? slot_of(tr, ns, b) : -1;
    if (sn >= 0) {
      const unsigned *cw = (const unsigned *)(buf + sn * (WP * 2048ll)) + t; const unsigned *vw = cw + WP * 256;
      #pragma unroll
      for (int jp = 0; jp < WP; ++jp) { cn[jp] = __builtin_nontemporal_load(cw + jp * 256); vn[jp] = __builtin_nontemporal_load(vw + jp * 256); }
    }
  }
  s_delta[t] = deltas[t]; s_value[t] = values[t];
  __syncthreads();
  #pragma unroll 1
  for (int kk = 0; kk < SPW; ++kk) {
    s = sn;
    #pragma unroll
    for (int jp = 0; jp < WP; ++jp) { c[jp] = cn[jp]; vc[jp] = vn[jp]; }
    sn = -1;
    if (kk + 1 < SPW) {
      const long long b = (((long long)blockIdx.x >> 3) * SPW + kk + 1) * 8 + (blockIdx.x & 7);
      sn = b < nslots ? slot_of(tr, ns, b) : -1;
      if (sn >= 0) {
        const unsigned *cw = (const unsigned *)(buf + sn * (WP * 2048ll)) + t; const unsigned *vw = cw + WP * 256;
        #pragma unroll
        for (int jp = 0; jp < WP; ++jp) { cn[jp] = __builtin_nontemporal_load(cw + jp * 256); vn[jp] = __builtin_nontemporal_load(vw + jp * 256); }
      }
    }
    if (s < 0) continue;
    const long long i = s * 512 + 2 * t;
    double sum[2] = {0, 0}, xv[W][2];
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
        xv[j][q] = code != 255u ? x[i + q + s_delta[code]] : 0.0;
      }
    #pragma unroll
    for (int j = 0; j < W; ++j)
    #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int sh = 8 * ((j & 1) * 2 + q);
        const unsigned code = (c[j >> 1] >> sh) & 255u;
        if (code != 255u) sum[q] += s_value[(vc[j >> 1] >> sh) & 255u] * xv[j][q];
      }
    typedef double d2 __attribute__((ext_vector_type(2)));
    if (i + 1 < n) { d2 o; o.x = sum[0]; o.y = sum[1]; __builtin_nontemporal_store(o, (d2 *)(y + i)); }
  }
}
extern "C" __global__ void __launch_bounds__(256) k(long long n, long long ns, const char *buf, const int *deltas, const double *values,
    const double *x, double *y, trav tr) {
  __shared__ int s_delta[256]; __shared__ double s_value[256];
#if MODE == 8
  const long long s = slot(tr, ns);
  const int t = threadIdx.x; const long long i = s * 512 + 2 * t;
  unsigned c[WP], vc[WP];
  if (s >= 0) {
    const unsigned *cw = (const unsigned *)(buf + s * (WP * 2048ll)) + t; const unsigned *vw = cw + WP * 256;
    #pragma unroll
    for (int jp = 0; jp < WP; ++jp) { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(vw + jp * 256); }
  }
  s_delta[threadIdx.x] = deltas[threadIdx.x]; s_value[threadIdx.x] = values[threadIdx.x];
  __syncthreads();
  if (s < 0) return;
#else
#if MODE != 5
  s_delta[threadIdx.x] = deltas[threadIdx.x]; s_value[threadIdx.x] = values[threadIdx.x];
  __syncthreads();
#endif
  const long long s = slot(tr, ns); if (s < 0) return;
  const int t = threadIdx.x; const long long i = s * 512 + 2 * t;
  const unsigned *cw = (const unsigned *)(buf + s * (WP * 2048ll)) + t; const unsigned *vw = cw + WP * 256;
  unsigned c[WP], vc[WP];
  #pragma unroll
  for (int jp = 0; jp < WP; ++jp) { c[jp] = __builtin_nontemporal_load(cw + jp * 256); vc[jp] = __builtin_nontemporal_load(vw + jp * 256); }
#endif
  double sum[2] = {0, 0}, xv[W][2];
  #pragma unroll
  for (int j = 0; j < W; ++j)
  #pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned code = (c[j >> 1] >> (8 * ((j & 1) * 2 + q))) & 255u;
#if MODE == 7          /* rows 2t and 2t+1 on the same even diagonal: ONE 16-byte load of x */
      if (q == 0) {
        const unsigned code1 = (c[j >> 1] >> (8 * ((j & 1) * 2 + 1))) & 255u;
        const int d0 = code != 255u ? s_delta[code] : 1;
        if (code == code1 && !(d0 & 1)) {
          const double2 p = *(const double2 *)(x + i + d0);
          xv[j][0] = p.x; xv[j][1] = p.y;
        } else {
          xv[j][0] = code != 255u ? x[i + d0] : 0.0;
          xv[j][1] = code1 != 255u ? x[i + 1 + s_delta[code1]] : 0.0;
        }
      }
#elif MODE == 1          /* no gathers: x[i+q] only */
      xv[j][q] = code != 255u ? x[i + q] * (double)s_delta[code] : 0.0;
#elif MODE == 4 || MODE == 5        /* no delta table: offsets by arithmetic on the code */
      xv[j][q] = code != 255u ? x[i + q + ((long long)code - 3)] : 0.0;
#else
      xv[j][q] = code != 255u ? x[i + q + s_delta[code]] : 0.0;
#endif
    }
  #pragma unroll
  for (int j = 0; j < W; ++j)
  #pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int sh = 8 * ((j & 1) * 2 + q);
      const unsigned code = (c[j >> 1] >> sh) & 255u;
#if MODE == 2 || MODE == 5         /* no value table */
      if (code != 255u) sum[q] += (double)((vc[j >> 1] >> sh) & 255u) * xv[j][q];
#else
      if (code != 255u) sum[q] += s_value[(vc[j >> 1] >> sh) & 255u] * xv[j][q];
#endif
    }
#if MODE == 3          /* no store */
  if (sum[0] + sum[1] == 12345.678) y[i] = sum[0];
#elif MODE == 6        /* non-temporal store */
  typedef double d2 __attribute__((ext_vector_type(2)));
  if (i + 1 < n) { d2 o; o.x = sum[0]; o.y = sum[1]; __builtin_nontemporal_store(o, (d2 *)(y + i)); }
#else
  if (i + 1 < n) { double2 o; o.x = sum[0]; o.y = sum[1]; *(double2 *)(y + i) = o; }
#endif
}
'''
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
class Trav(ctypes.Structure):
    _fields_ = [("chunk", ctypes.c_int), ("planes", ctypes.c_int), ("plane_blocks", ctypes.c_int)]
tr = Trav(int(S.trav.chunk), int(S.trav.planes), int(S.trav.plane_blocks))
grid = int(S.trav.grid_blocks); ns = (N + 511) // 512
names = {0: "full", 1: "no gathers (x[i] only)", 2: "no value table", 3: "no y store", 4: "no delta table (+-3 window)", 5: "no LDS at all", 6: "non-temporal y store", 7: "paired 16-byte x loads", 8: "code loads before the table barrier"}
for spw in (2, 4, 8):
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    L.module_compile(0, ("#define MODE 0\n#define SPW %d\n" % spw + SRC).encode(), b"-ffp-contract=off", ctypes.byref(mod))
    L.module_get_function(0, mod, b"kp", ctypes.byref(fn))
    nslots = grid
    g2 = (nslots + 8 * spw - 1) // (8 * spw) * 8
    args = [ctypes.c_longlong(N), ctypes.c_longlong(ns), ctypes.c_longlong(nslots), ctypes.c_void_p(S.sell.data_ptr()), ctypes.c_void_p(S.deltas.data_ptr()),
            ctypes.c_void_p(S.values.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), tr]
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    def run(): L.launch(0, fn, g2, 1, 1, 256, 1, 1, 0, stream, arr)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    yr = torch.empty_like(y); S.mul(x, yr)
    print("prefetching, %d slices per workgroup  %.3f ms  identical %s" % (spw, e0.elapsed_time(e1) / 30, torch.equal(y, yr)), flush=True)
    L.module_unload(0, mod)
for mode in (0, 8):
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    L.module_compile(0, ("#define MODE %d\n" % mode + SRC).encode(), b"-ffp-contract=off", ctypes.byref(mod))
    L.module_get_function(0, mod, b"k", ctypes.byref(fn))
    args = [ctypes.c_longlong(N), ctypes.c_longlong(ns), ctypes.c_void_p(S.sell.data_ptr()), ctypes.c_void_p(S.deltas.data_ptr()),
            ctypes.c_void_p(S.values.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), tr]
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    def run(): L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    print("%-32s %.3f ms" % (names[mode], e0.elapsed_time(e1) / 30), flush=True)
    L.module_unload(0, mod)
