"""What the memory system gives a plain copy (1 read : 1 write) and a 2 : 1 stream at the headline's size (fp64, 512^3 elements):
the ceiling for a product whose HBM traffic is x once + y once.  JSON on stdout (profiles/r03_copy_rate.json)."""
import json
import torch

dev = torch.device("cuda:0")
N = 512 ** 3
x = torch.empty(N, dtype=torch.float64, device=dev).normal_()
z = torch.empty(N, dtype=torch.float64, device=dev).normal_()
y = torch.empty_like(x)


def timed(fn, reps=30, rounds=3):
    best = 1e30
    for _ in range(rounds):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


out = {}
ms = timed(lambda: y.copy_(x)); out["copy_1r1w"] = {"ms": round(ms, 4), "TBps": round(16 * N / ms / 1e9, 3)}
ms = timed(lambda: torch.add(x, z, out=y)); out["add_2r1w"] = {"ms": round(ms, 4), "TBps": round(24 * N / ms / 1e9, 3)}
ms = timed(lambda: y.fill_(1.0)); out["fill_0r1w"] = {"ms": round(ms, 4), "TBps": round(8 * N / ms / 1e9, 3)}
ms = timed(lambda: torch.sum(x)); out["sum_1r0w"] = {"ms": round(ms, 4), "TBps": round(8 * N / ms / 1e9, 3)}
print(json.dumps(out))
