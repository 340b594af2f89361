"""Does one LDS `atomicAdd` with return value hand the lanes of a wave that hit the SAME address their old values in
lane order (lane i < lane j  =>  returned(i) < returned(j))?  Not an architectural promise; the stable radix sort could
rank a wave's keys with ONE LDS operation per key instead of five if it holds.  Random digit patterns, 16 waves per
workgroup on private counters, many rounds; reports the number of order violations."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import lib
L = lib(); dev = torch.device("cuda:0")
SRC = r'''
extern "C" __global__ void __launch_bounds__(1024) k(const unsigned *digits, int rounds, int nd, unsigned long long *violations, unsigned long long *checked) {
  __shared__ unsigned cnt[16][256];
  const int t = threadIdx.x, wave = t / 64, lane = t % 64;
  unsigned long long bad = 0, seen = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = t; i < 16 * 256; i += 1024) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const unsigned d = digits[((size_t)blockIdx.x * rounds + r) * 1024 + t] % (unsigned)nd;
    const unsigned old = atomicAdd(&cnt[wave][d], 1u);
    // expected: the number of lower lanes of this wave with the same digit
    unsigned expect = 0;
    for (int l = 0; l < 64; ++l) { const unsigned dl = __shfl(d, l, 64); if (l < lane && dl == d) ++expect; }
    if (old != expect) ++bad;
    ++seen;
    __syncthreads();
  }
  atomicAdd(violations, bad); atomicAdd(checked, seen);
}
'''
mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
L.module_compile(0, SRC.encode(), b"", ctypes.byref(mod))
L.module_get_function(0, mod, b"k", ctypes.byref(fn))
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
blocks, rounds = 2048, 64
dig = torch.randint(0, 2**31 - 1, (blocks * rounds * 1024,), dtype=torch.int32, device=dev)
for nd in (1, 2, 3, 7, 16, 64, 256):
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    args = [ctypes.c_void_p(dig.data_ptr()), ctypes.c_int(rounds), ctypes.c_int(nd), ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(res.data_ptr() + 8)]
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    L.launch(0, fn, blocks, 1, 1, 1024, 1, 1, 0, stream, arr)
    torch.cuda.synchronize()
    print("distinct digits %3d: %d order violations in %d returned values" % (nd, int(res[0]), int(res[1])), flush=True)
