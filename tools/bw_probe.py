"""What bounds streaming kernels?  Copy / triad variants at 1e8 fp64 (diagnostic, not a product path)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import lib
L = lib(); dev = torch.device("cuda:0")
SRC = r'''
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void copy8(ulong n, double *a, const double *b) {
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n; i += g) a[i] = b[i];
}
extern "C" __global__ void copy16(ulong n, double *a, const double *b) {
  const ulong n2 = n / 2;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g) ((d2*)a)[i] = ((const d2*)b)[i];
}
extern "C" __global__ void copy16nt(ulong n, double *a, const double *b) {
  const ulong n2 = n / 2;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g)
    __builtin_nontemporal_store(__builtin_nontemporal_load(((const d2*)b) + i), ((d2*)a) + i);
}
// one pass, no loop: a workgroup per 256 x 16 B x UNROLL contiguous bytes
extern "C" __global__ void copy16u4(ulong n, double *a, const double *b) {
  const ulong n2 = n / 2;
  const ulong base = (ulong)blockIdx.x * 1024 + threadIdx.x;
  d2 v[4];
  #pragma unroll
  for (int k = 0; k < 4; ++k) if (base + k * 256 < n2) v[k] = ((const d2*)b)[base + k * 256];
  #pragma unroll
  for (int k = 0; k < 4; ++k) if (base + k * 256 < n2) ((d2*)a)[base + k * 256] = v[k];
}
extern "C" __global__ void read16(ulong n, double *a, const double *b) {
  const ulong n2 = n / 2; d2 s = {0, 0};
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g) { d2 v = ((const d2*)b)[i]; s.x += v.x; s.y += v.y; }
  if (s.x + s.y == 12345.678) a[0] = s.x;
}
extern "C" __global__ void write16(ulong n, double *a, const double *b) {
  const ulong n2 = n / 2; d2 s = {1.5, 2.5};
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g) ((d2*)a)[i] = s;
}
extern "C" __global__ void triad16u2(ulong n, double *a, const double *b, const double *c, const double *d) {
  const ulong n2 = n / 2;
  const ulong base = (ulong)blockIdx.x * 512 + threadIdx.x;
  d2 vb[2], vc[2], vd[2];
  #pragma unroll
  for (int k = 0; k < 2; ++k) if (base + k * 256 < n2) { vb[k] = ((const d2*)b)[base + k * 256]; vc[k] = ((const d2*)c)[base + k * 256]; vd[k] = ((const d2*)d)[base + k * 256]; }
  #pragma unroll
  for (int k = 0; k < 2; ++k) if (base + k * 256 < n2) { d2 r; r.x = vb[k].x * vc[k].x + vd[k].x; r.y = vb[k].y * vc[k].y + vd[k].y; ((d2*)a)[base + k * 256] = r; }
}
extern "C" __global__ void triad8(ulong n, double *a, const double *b, const double *c, const double *d) {
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n; i += g) a[i] = b[i] * c[i] + d[i];
}
extern "C" __global__ void triad8two(ulong n, double *a, const double *b, const double *c, const double *d) {   // the generated kernel's shape
  const ulong g = blockDim.x * (ulong)gridDim.x;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x; i < n; i += 2 * g) {
    const bool two = i + g < n; const ulong j = two ? i + g : i;
    double r0 = b[i] * c[i] + d[i], r1 = b[j] * c[j] + d[j];
    a[i] = r0; if (two) a[j] = r1;
  }
}
#define NOLOOP(NAME, U) extern "C" __global__ void NAME(ulong n, double *a, const double *b, const double *c, const double *d) { \
  const ulong base = (ulong)blockIdx.x * (256 * U) + threadIdx.x; double r[U]; \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; r[k] = i < n ? b[i] * c[i] + d[i] : 0; } \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; if (i < n) a[i] = r[k]; } }
NOLOOP(triad8u1, 1) NOLOOP(triad8u2, 2) NOLOOP(triad8u4, 4) NOLOOP(triad8u8, 8)
#define NOLOOPS(NAME, U) extern "C" __global__ void NAME(ulong n, double *a, const double *b, const double *c, const double *d) { \
  const ulong base = (ulong)blockIdx.x * (256 * U) + threadIdx.x; double r[U]; \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; r[k] = i < n ? b[i] * c[i] + sin(d[i]) : 0; } \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; if (i < n) a[i] = r[k]; } }
NOLOOPS(sin8u2, 2) NOLOOPS(sin8u4, 4)
#define NOLOOPNT(NAME, U) extern "C" __global__ void NAME(ulong n, double *a, const double *b, const double *c, const double *d) { \
  const ulong base = (ulong)blockIdx.x * (256 * U) + threadIdx.x; double r[U]; \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; r[k] = i < n ? __builtin_nontemporal_load(b + i) * __builtin_nontemporal_load(c + i) + __builtin_nontemporal_load(d + i) : 0; } \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; if (i < n) __builtin_nontemporal_store(r[k], a + i); } }
NOLOOPNT(triad8u2nt, 2)
#define NOLOOPNTS(NAME, U) extern "C" __global__ void NAME(ulong n, double *a, const double *b, const double *c, const double *d) { \
  const ulong base = (ulong)blockIdx.x * (256 * U) + threadIdx.x; double r[U]; \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; r[k] = i < n ? b[i] * c[i] + d[i] : 0; } \
  _Pragma("unroll") for (int k = 0; k < U; ++k) { const ulong i = base + k * 256; if (i < n) __builtin_nontemporal_store(r[k], a + i); } }
NOLOOPNTS(triad8u2nts, 2)
extern "C" __global__ void dot8(ulong n, double *a, const double *b, const double *c) {
  double s = 0;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n; i += g) s += b[i] * c[i];
  if (s == 12345.678) a[0] = s;
}
extern "C" __global__ void dot8x2(ulong n, double *a, const double *b, const double *c) {
  double s = 0; const ulong g = blockDim.x * (ulong)gridDim.x;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x; i < n; i += 2 * g) {
    const ulong j = i + g < n ? i + g : i; const double w = i + g < n ? 1.0 : 0.0;
    double v0 = b[i] * c[i], v1 = b[j] * c[j]; s += v0; s += w * v1; }
  if (s == 12345.678) a[0] = s;
}
extern "C" __global__ void dot8x4(ulong n, double *a, const double *b, const double *c) {
  double s = 0; const ulong g = blockDim.x * (ulong)gridDim.x;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x; i < n; i += 4 * g) {
    double v[4];
    #pragma unroll
    for (int k = 0; k < 4; ++k) { const ulong j = i + k * g; v[k] = j < n ? b[j] * c[j] : 0.0; }
    s += v[0]; s += v[1]; s += v[2]; s += v[3]; }
  if (s == 12345.678) a[0] = s;
}
'''
mod = ctypes.c_void_p(); L.module_compile(0, SRC.encode(), b"", ctypes.byref(mod))
n = 100_000_000
a, b, c, d = (torch.rand(n, dtype=torch.float64, device=dev) for _ in range(4))
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def bench(name, grid, nargs, bytes_per_elem):
    fn = ctypes.c_void_p(); L.module_get_function(0, mod, name.encode(), ctypes.byref(fn))
    args = [ctypes.c_uint64(n)] + [ctypes.c_void_p(t.data_ptr()) for t in (a, b, c, d)[:nargs]]
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(x), ctypes.c_void_p) for x in args])
    def run(): L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-10s grid %7d  %.4f ms  %5.0f GB/s" % (name, grid, ms, bytes_per_elem * n / ms / 1e6), flush=True)
for g in (2048, 8192, 32768):
    bench("copy8", g, 2, 16); bench("copy16", g, 2, 16); bench("copy16nt", g, 2, 16)
bench("copy16u4", (n // 2 + 1023) // 1024, 2, 16)
for g in (2048, 8192):
    bench("read16", g, 2, 8); bench("write16", g, 2, 8)
bench("triad16u2", (n // 2 + 511) // 512, 4, 32)

for g in (2048, 8192, 32768, 131072):
    bench("triad8", g, 4, 32); bench("triad8two", g, 4, 32)
for name, u in (("triad8u1", 1), ("triad8u2", 2), ("triad8u4", 4), ("triad8u8", 8), ("sin8u2", 2), ("sin8u4", 4)):
    bench(name, (n + 256 * u - 1) // (256 * u), 4, 32)

for rep in range(2):
    for name, u in (("triad8u2", 2), ("triad8u2nt", 2), ("triad8u2nts", 2)):
        bench(name, (n + 256 * u - 1) // (256 * u), 4, 32)

for g in (2048, 4096, 8192, 32768):
    bench("dot8", g, 3, 16); bench("dot8x2", g, 3, 16); bench("dot8x4", g, 3, 16)
