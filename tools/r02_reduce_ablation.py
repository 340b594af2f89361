"""Round 2: how should the Reductor's stage-1 loop read?  JIT variants of sum(a*b), n = 1e8 fp64, 8 x CU workgroups of 256 as the
library launches them (and 4x as many).  Diagnostic; output gpurun_out/r02_reduce_ablation.json"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
L = lib(); dev = torch.device("cuda:0")
n = 10**8
a = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 1); b = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 2)
SRC = r'''
typedef unsigned long ulong;
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ inline void finish(double s, double *out) {
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ double sd[16];
  if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += sd[w]; out[blockIdx.x] = s; }
}
extern "C" __global__ void __launch_bounds__(256) k(ulong n, const double *a, const double *b, double *out) {
  double s = 0;
  const ulong gs = blockDim.x * (ulong)gridDim.x; ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x;
#if MODE == 0        // one element per trip (round-1 / round-2 library loop)
  for (; i < n; i += gs) s += a[i] * b[i];
#elif MODE == 1      // BATCH strided elements per trip, loads first, folds in loop order
  for (; i + (BATCH - 1) * gs < n; i += BATCH * gs) {
    double v[BATCH];
    #pragma unroll
    for (int e = 0; e < BATCH; ++e) v[e] = a[i + e * gs] * b[i + e * gs];
    #pragma unroll
    for (int e = 0; e < BATCH; ++e) s += v[e];
  }
  for (; i < n; i += gs) s += a[i] * b[i];
#elif MODE == 2      // two CONSECUTIVE elements per lane and trip: one 16-byte load per array (n even, arrays 16-byte aligned)
  const d2 *a2 = (const d2 *)a, *b2 = (const d2 *)b; const ulong n2 = n / 2;
  for (; i + (BATCH - 1) * gs < n2; i += BATCH * gs) {
    d2 x[BATCH], y[BATCH];
    #pragma unroll
    for (int e = 0; e < BATCH; ++e) { x[e] = a2[i + e * gs]; y[e] = b2[i + e * gs]; }
    #pragma unroll
    for (int e = 0; e < BATCH; ++e) { s += x[e].x * y[e].x; s += x[e].y * y[e].y; }
  }
  for (; i < n2; i += gs) { const d2 x = a2[i], y = b2[i]; s += x.x * y.x; s += x.y * y.y; }
#elif MODE == 3      // each workgroup owns one CONTIGUOUS chunk of the arrays (no grid stride), 16-byte loads, BATCH in flight
  const d2 *a2 = (const d2 *)a, *b2 = (const d2 *)b; const ulong n2 = n / 2;
  const ulong per = (n2 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n2 ? lo + per : n2;
  ulong j = lo + threadIdx.x;
  for (; j + (BATCH - 1) * 256 < hi; j += BATCH * 256) {
    d2 x[BATCH], y[BATCH];
    #pragma unroll
    for (int e = 0; e < BATCH; ++e) { x[e] = a2[j + e * 256]; y[e] = b2[j + e * 256]; }
    #pragma unroll
    for (int e = 0; e < BATCH; ++e) { s += x[e].x * y[e].x; s += x[e].y * y[e].y; }
  }
  for (; j < hi; j += 256) { const d2 x = a2[j], y = b2[j]; s += x.x * y.x; s += x.y * y.y; }
#endif
  finish(s, out);
}
'''
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
cus = torch.cuda.get_device_properties(0).multi_processor_count
ref = float((a * b).sum())
res = []
runs = []
for label, mode, batch in (("one element per trip", 0, 1), ("2 strided per trip", 1, 2), ("4 strided per trip", 1, 4), ("8 strided per trip", 1, 8),
                           ("16-byte loads, 1 per trip", 2, 1), ("16-byte loads, 2 per trip", 2, 2), ("16-byte loads, 4 per trip", 2, 4),
                           ("contiguous chunk per workgroup, 16-byte, 2 in flight", 3, 2), ("contiguous chunk per workgroup, 16-byte, 4 in flight", 3, 4)):
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    L.module_compile(0, ("#define MODE %d\n#define BATCH %d\n" % (mode, batch) + SRC).encode(), b"-ffp-contract=off", ctypes.byref(mod))
    L.module_get_function(0, mod, b"k", ctypes.byref(fn))
    for mult in (8, 32):
        grid = cus * mult
        out = torch.zeros(grid, dtype=torch.float64, device=dev)
        args = [ctypes.c_ulong(n), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.data_ptr())]
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(x), ctypes.c_void_p) for x in args])
        runs.append(("%s, %d x CU workgroups" % (label, mult), out, (lambda fn=fn, grid=grid, arr=arr, keep=args: L.launch(0, fn, grid, 1, 1, 256, 1, 1, 0, stream, arr))))
times = {r[0]: [] for r in runs}; ok = {}
for rnd in range(3):
    for label, out, run in runs:
        for _ in range(60): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): run()
        e1.record(); torch.cuda.synchronize()
        times[label].append(round(e0.elapsed_time(e1) / 40, 4))
        ok[label] = abs(float(out.sum()) - ref) <= 1e-9 * abs(ref)
for label, ts in times.items():
    m = min(ts)
    print("%-72s %s best %.4f ms  %.2f TB/s  sum ok %s" % (label, ts, m, 16.0 * n / m / 1e9, ok[label]), flush=True)
    res.append({"variant": label, "ms": ts, "best_ms": m, "tbps": round(16.0 * n / m / 1e9, 3), "sum_ok": ok[label]})
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r02_reduce_ablation.json", "w"), indent=1)
