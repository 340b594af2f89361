#!/bin/bash
# round 6, call 25: the plane product's flat instantiation (2-D rows of 512-point virtual lines): tests, 16384^2 and 8192 x 20000
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_spmv.py -q -x -m gpu -k "two_dimensional or plane_product or vector_added" 2>&1 | tail -4 > gpurun_out/r06_gpu25_tests.log
timeout 900 python - > gpurun_out/r06_gpu25_2d.log 2>&1 <<'PY'
import sys, os, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import unstructured as U, bench
from vexcl_amd import ops
dev = torch.device("cuda:0")
for W, H in ((16384, 16384), (8192, 20000), (512, 8)):
    if W == 512: break
    ptr, col, val, h2i = U.stencil2d(W, H, dev)
    n = W * H
    x = ops.fill_hash(torch.empty(n, dtype=torch.float64, device=dev), 42); y = torch.empty_like(x)
    A = ops.SpMat(ptr, col, val)
    A.apply(x, y)
    yr, mag = U.stencil2d_reference(x, W, H, h2i)
    bad = int(((y - yr).abs() > 1e-10 * mag).sum()); del yr, mag
    t = min(bench.timed_events(torch, lambda: A.apply(x, y), 20) for _ in range(3))
    print(json.dumps({"grid": "%d x %d" % (W, H), "product": A.product, "plane": A.plane, "ms": round(t, 5), "frac_of_8TBps": round((A.matrix_bytes() + 16 * n) / t / 1e6 / 8000, 4), "rows_outside_tolerance": bad}), flush=True)
    del A, ptr, col, val, x, y; torch.cuda.empty_cache()
PY
cat gpurun_out/r06_gpu25_tests.log; grep "^{" gpurun_out/r06_gpu25_2d.log | cut -c1-500; tail -2 gpurun_out/r06_gpu25_2d.log | cut -c1-300
