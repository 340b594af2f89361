"""Round 6 EXPERIMENT (not the product): needs tools/r06_sort_onesweep_experiment.hip built in place of vexcl_amd/csrc/sort.hip -- rank modes 8 / 9
exist only there (profiles/r06_sort_chain_experiments.md).  The chained scatter (vexhip_sort_set_rank(8)...) against the default on hashed u32 keys: same result (keys and, for pairs, the
permutation), time per sort; SORT_MODES = comma list of rank modes, SORT_N = keys."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexcl_amd import ops, lib
dev = torch.device("cuda:0"); L = lib()
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
modes = [int(m) for m in os.environ.get("SORT_MODES", "-1,8").split(",")]
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
out = {"checks": [], "times": {}}

def run(mode, k, ktmp, v, vtmp, tmp, n, vb):
    L.sort_set_rank(mode)
    L.sort(0, stream, 3, 0, p(k), p(ktmp), vb, p(v), p(vtmp), n, p(tmp))
    a, b = ctypes.c_int64(), ctypes.c_int64()
    L.sort_status(0, stream, n, p(tmp), ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value

# correctness at small and ragged sizes, keys only and pairs (the permutation = stability)
for n in ((1, 2, 777, 12288, 12289, 8 * 12288, 8 * 12288 + 5, 9 * 12288 + 1, 1000003, 12288 * 1000, 50_000_017) if os.environ.get('SORT_CHECKS', '1') != '0' else ()):
    for vb, vdt in ((0, None), (4, torch.int32), (8, torch.int64)):
        res = {}
        for mode in modes:
            k = torch.empty(n, dtype=torch.int32, device=dev); ops.fill_hash(k, 7)
            if n > 100:
                k &= 0x00FFFFFF if n % 2 else -1          # (odd sizes: a constant top digit)
            ktmp = torch.empty_like(k)
            v = torch.arange(n, dtype=vdt, device=dev) if vb else None
            vtmp = torch.empty_like(v) if vb else None
            tmp = torch.empty(L.sort_tmp_bytes(3, n), dtype=torch.uint8, device=dev)
            st = run(mode, k, ktmp, v, vtmp, tmp, n, vb)
            res[mode] = (k, v, st)
        ref = res[modes[0]]
        for mode in modes[1:]:
            same = bool(torch.equal(res[mode][0], ref[0])) and (vb == 0 or bool(torch.equal(res[mode][1], ref[1])))
            out["checks"].append({"n": n, "value_bytes": vb, "mode": mode, "same_as_%d" % modes[0]: same, "status": res[mode][2]})
            if not same:
                print("MISMATCH", n, vb, mode, flush=True)
        del res
L.sort_set_rank(-1)
print(json.dumps({"all_same": all(c["same_as_%d" % modes[0]] for c in out["checks"]), "checks": len(out["checks"])}), flush=True)

n = int(float(os.environ.get("SORT_N", "1e9")))
k = torch.empty(n, dtype=torch.int32, device=dev); ktmp = torch.empty_like(k)
tmp = torch.empty(L.sort_tmp_bytes(3, n), dtype=torch.uint8, device=dev)
ref = None
for mode in modes:
    best = None
    for _ in range(4):
        ops.fill_hash(k, 42); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st = run(mode, k, ktmp, None, None, tmp, n, 0) if False else None
        L.sort_set_rank(mode)
        L.sort(0, stream, 3, 0, p(k), p(ktmp), 0, None, None, n, p(tmp))
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1); best = t if best is None else min(best, t)
    a, b = ctypes.c_int64(), ctypes.c_int64()
    L.sort_status(0, stream, n, p(tmp), ctypes.byref(a), ctypes.byref(b))
    if ref is None:
        ref = k.clone()
    out["times"][str(mode)] = {"ms": round(best, 3), "gkeys_per_s": round(n / best / 1e6, 1), "same_as_first": bool(torch.equal(k, ref)), "status": [a.value, b.value], "lookback_words_per_tile": (b.value & 0xFFFFF) / 100, "polls_per_tile": ((b.value >> 20) & 0xFFFFF) / 100, "max_words": b.value >> 40}
    print(mode, out["times"][str(mode)], flush=True)
L.sort_set_rank(-1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.environ.get("SORT_OUT", "gpurun_out/r06_sort_chain.json"), "w"), indent=1)
if os.environ.get("SORT_PROFILE"):
    import numpy as np
    words = tmp.view(torch.int32)
    redo_off = tmp.numel() // 4 - 8 - ((n + 3071) // 3072 + 4)
    d = words[redo_off + 100000: redo_off + 100000 + 512 * 12 * 12].cpu().numpy().astype(np.int64).reshape(512, 12, 12) * 16
    tiles = n // 12288 / 512
    names = ["find+barrier A", "look-back / fetch", "barrier B", "write-out", "barrier C + zero + barrier", "rank", "barriers + counts + off + rr", "reorder + first round", "rounds", "polls", "words polled", "-"]
    for wv in (0, 3, 5, 11):
        print("wave %2d cycles per tile:" % wv, {names[i]: int(d[:, wv, i].mean() / tiles) for i in range(8)}, "sum", int(d[:, wv, :8].sum(axis=1).mean() / tiles), {names[i]: round(float(d[:, wv, i].mean() / tiles / 16), 2) for i in (8, 9, 10)})
