#!/bin/bash
# One gpurun call: GPU tests, smoke, kernel-variant sweep, bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; free -g | head -2 >> gpurun_out/device.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout 600 python tools/spmv_sweep.py > gpurun_out/sweep.log 2>&1; echo "sweep exit $?"; cat gpurun_out/sweep.log | tail -25
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -3 gpurun_out/bench.log
