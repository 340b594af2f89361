#!/usr/bin/env python
"""How much HOST time one halo exchange costs through torch.distributed (batch_isend_irecv of one send + one receive,
RCCL underneath): one rank exchanging a 512^2 plane with itself on the 1-GPU box.  This is the part of the 8-GPU
strong-scaling step (~0.12 ms of kernels per GPU) that cannot be measured otherwise here.  Measured: 43 us.
(Issuing the same ncclSend/ncclRecv group through ctypes on torch's librccl.so was tried to shave that further; a
second communicator could not bootstrap in the sandbox -- "remote process exited or there was a network error" -- and
with 43 us there is little left to gain.)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 260100
send = torch.arange(n, dtype=torch.float64, device=dev)
recv = torch.zeros(n, dtype=torch.float64, device=dev)

def torch_exchange():
    ops = [dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()

res = {}
try:
    torch_exchange(); torch.cuda.synchronize()
    assert torch.equal(send, recv)
    t0 = time.perf_counter()
    for _ in range(200):
        torch_exchange()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    res["torch batch_isend_irecv host us"] = (t1 - t0) / 200 * 1e6
except Exception as e:
    res["torch error"] = repr(e)

print(res)
dist.destroy_process_group()
