#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ns in 0 1; do w=4; export NOSYNC=$ns; echo "NOSYNC=$ns"
echo "== world $w"
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2954$w tools/r06_n4_probe.py 2>&1 | grep "^\[rank\|Timeout\|File \"/root/repo\|Error\|error" | grep "B \|Timeout\|File" | head -40 | cut -c1-200
done
