"""Experiment: SELL8V codes packed per ROW (4 columns of one row per word; a lane owns rows t and t + 256 of the
slice) instead of per row PAIR -- every gather of x is then a contiguous 512-byte wave access."""
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
ns = (N + 511) // 512
old = S.sell.view(ns, 2, 4, 256, 4)                     # [slice][col/val][jp][lane][jj*2+q]
new = torch.empty(ns, 2, 2, 512, 4, dtype=torch.uint8, device=dev)
for kind in range(2):
    for j in range(8):
        for q in range(2):
            src = old[:, kind, j // 2, :, (j % 2) * 2 + q] if j < 7 else None      # rows 2t+q
            dst = new[:, kind, j // 4, q::2, j % 4]
            if j < 7: dst.copy_(src)
            else: dst.fill_(255 if kind == 0 else 0)
new = new.contiguous()
x = ops.fill_hash(torch.empty(N, dtype=torch.float64, device=dev), 1); y = torch.empty_like(x); yref = torch.empty_like(x)
S.mul(x, yref)
SRC = r'''
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
extern "C" __global__ void __launch_bounds__(256) k(long long n, long long ns, const char *buf, const int *deltas, const double *values,
    const double *x, double *y, trav tr) {
  __shared__ int s_delta[256]; __shared__ double s_value[256];
  s_delta[threadIdx.x] = deltas[threadIdx.x]; s_value[threadIdx.x] = values[threadIdx.x];
  __syncthreads();
  const long long s = slot(tr, ns); if (s < 0) return;
  const int t = threadIdx.x;
  const unsigned *cw = (const unsigned *)(buf + s * 8192ll); const unsigned *vw = cw + 1024;
  unsigned c[2][2], vc[2][2];
  #pragma unroll
  for (int q = 0; q < 2; ++q)
  #pragma unroll
    for (int j4 = 0; j4 < 2; ++j4) { c[q][j4] = __builtin_nontemporal_load(cw + j4 * 512 + q * 256 + t); vc[q][j4] = __builtin_nontemporal_load(vw + j4 * 512 + q * 256 + t); }
  double sum[2] = {0, 0}, xv[7][2];
  #pragma unroll
  for (int j = 0; j < 7; ++j)
  #pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned code = (c[q][j >> 2] >> (8 * (j & 3))) & 255u;
      xv[j][q] = code != 255u ? x[s * 512 + q * 256 + t + s_delta[code]] : 0.0;
    }
  #pragma unroll
  for (int j = 0; j < 7; ++j)
  #pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned code = (c[q][j >> 2] >> (8 * (j & 3))) & 255u;
      if (code != 255u) sum[q] += s_value[(vc[q][j >> 2] >> (8 * (j & 3))) & 255u] * xv[j][q];
    }
  #pragma unroll
  for (int q = 0; q < 2; ++q) { const long long i = s * 512 + q * 256 + t; if (i < n) __builtin_nontemporal_store(sum[q], y + i); }
}
'''
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
class Trav(ctypes.Structure):
    _fields_ = [("chunk", ctypes.c_int), ("planes", ctypes.c_int), ("plane_blocks", ctypes.c_int)]
tr = Trav(int(S.trav.chunk), int(S.trav.planes), int(S.trav.plane_blocks))
grid = int(S.trav.grid_blocks)
mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
L.module_compile(0, SRC.encode(), b"-ffp-contract=off", ctypes.byref(mod))
L.module_get_function(0, mod, b"k", ctypes.byref(fn))
args = [ctypes.c_longlong(N), ctypes.c_longlong(ns), ctypes.c_void_p(new.data_ptr()), ctypes.c_void_p(S.deltas.data_ptr()),
        ctypes.c_void_p(S.values.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), tr]
arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
def run(): L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr)
def timed(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rep in range(2):
    print("row layout %.3f ms   library (pair layout) %.3f ms   identical %s" % (timed(run), timed(lambda: S.mul(x, yref)), torch.equal(y, yref)), flush=True)
