"""Does the plane product's time depend on where x and y lie?  One process, 512^3: x and y are views into one large buffer at
varying distances; the product, the hand copy and the march product timed for each placement.  JSON on stdout."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vexcl_amd import lib, ops
L = lib()
dev = torch.device("cuda:0"); n = 512; N = n ** 3
def timed(fn, reps=30, rounds=3):
    best = 1e30
    for _ in range(rounds):
        for _ in range(30): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
p, c, v = ops.poisson3d(n, dev)
cfgs = [(4, 256, 2), (2, 512, 1), (2, 128, 1)]
mats = {}
for tile, d, st in cfgs:
    os.environ["VEXHIP_PLANE_DEPTH"] = str(d); os.environ["VEXHIP_PLANE_TILE"] = str(tile); os.environ["VEXHIP_PLANE_STORE"] = str(st)
    mats["plane %dx%dx%d" % (tile, d, st)] = ops.SpMat(p, c, v)
for k in ("VEXHIP_PLANE_DEPTH", "VEXHIP_PLANE_TILE", "VEXHIP_PLANE_STORE"): os.environ.pop(k)
mats["march"] = ops.SpMat(p, c, v, plane=False)
del p, c, v
for A in mats.values(): A.ptr = A.col = A.val = None
torch.cuda.empty_cache()
big = torch.empty(3 * N + (1 << 24), dtype=torch.float64, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {"base_address_mod_2MiB": big.data_ptr() % (1 << 21), "rows": []}
for gap in [int(g) for g in os.environ.get('GAPS', '0,512,4096,32768,262144,263680,1048576,1572864,8388608,67108864,134217728').split(',')]:
    x = big[:N]; y = big[N + gap:2 * N + gap]
    ops.fill_hash(x, 42)
    row = {"gap_elements": gap, "y_minus_x_bytes": (N + gap) * 8}
    for k, A in mats.items():
        row[k] = round(timed(lambda: A.apply(x, y)), 4)
    row["hand_copy"] = round(timed(lambda: L.stream_copy_f64(0, stream, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), N)), 4)
    out["rows"].append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps(out))
