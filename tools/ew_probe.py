"""Upside check: scalar vs 16-byte-per-lane versions of a = b*c + sin(d) (diagnostic)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import lib
L = lib(); dev = torch.device("cuda:0")
SRC = r'''
extern "C" __global__ void k1(ulong n, double *a, double *b, double *c, double *d) {
  for (ulong idx = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; idx < n; idx += g)
    a[idx] = ( ( b[idx] * c[idx] ) + sin( d[idx] ) );
}
extern "C" __global__ void k2(ulong n, double *a, double *b, double *c, double *d) {
  const ulong n2 = n / 2;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g) {
    double2 vb = ((double2*)b)[i], vc = ((double2*)c)[i], vd = ((double2*)d)[i], r;
    r.x = vb.x * vc.x + sin(vd.x); r.y = vb.y * vc.y + sin(vd.y);
    ((double2*)a)[i] = r;
  }
}
typedef double d2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void k3(ulong n, double *a, double *b, double *c, double *d) {
  const ulong n2 = n / 2;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g) {
    d2 vb = __builtin_nontemporal_load(((d2*)b) + i), vc = __builtin_nontemporal_load(((d2*)c) + i), vd = __builtin_nontemporal_load(((d2*)d) + i), r;
    r.x = vb.x * vc.x + sin(vd.x); r.y = vb.y * vc.y + sin(vd.y);
    __builtin_nontemporal_store(r, ((d2*)a) + i);
  }
}
extern "C" __global__ void k4(ulong n, double *a, double *b, double *c, double *d) {
  for (ulong idx = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; idx < n; idx += g)
    a[idx] = ( ( b[idx] * c[idx] ) + d[idx] );
}
extern "C" __global__ void k5(ulong n, double *a, double *b, double *c, double *d) {
  const ulong n2 = n / 2;
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x, g = blockDim.x * (ulong)gridDim.x; i < n2; i += g) {
    double2 vb = ((double2*)b)[i], vc = ((double2*)c)[i], vd = ((double2*)d)[i], r;
    r.x = vb.x * vc.x + vd.x; r.y = vb.y * vc.y + vd.y;
    ((double2*)a)[i] = r;
  }
}
'''
mod = ctypes.c_void_p(); L.module_compile(0, SRC.encode(), b"", ctypes.byref(mod))
n = 100_000_000
a, b, c, d = (torch.rand(n, dtype=torch.float64, device=dev) for _ in range(4))
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, grids in (("k1", (2048, 4096, 8192)), ("k2", (2048, 4096, 8192)), ("k3", (2048, 4096)), ("k4", (2048, 8192)), ("k5", (2048, 8192))):
    fn = ctypes.c_void_p(); L.module_get_function(0, mod, name.encode(), ctypes.byref(fn))
    for grid in grids:
        args = [ctypes.c_uint64(n)] + [ctypes.c_void_p(t.data_ptr()) for t in (a, b, c, d)]
        arr = (ctypes.c_void_p * 5)(*[ctypes.cast(ctypes.pointer(x), ctypes.c_void_p) for x in args])
        def run(): L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(name, "grid", grid, "ms", round(ms, 4), "GB/s", round(32 * n / ms / 1e6), flush=True)
