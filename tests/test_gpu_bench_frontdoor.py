"""The front door of the multi-GPU measurement: `python bench.py --gpus N` as the driver types it (no launcher, no
WORLD_SIZE) must start N ranks itself, must REFUSE when the box has fewer GPUs than ranks (round 2 silently measured one
GPU and printed n_gpus: 1), and its N > 1 line must carry a check that trusts no transport.  On the one GPU of the
gpurun box the N-rank path is exercised with --one-device (all ranks on cuda:0, gloo for the set-up collectives, the
peer-mapped-window transport for the ghost exchange between the processes)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")


def test_more_ranks_than_gpus_is_refused():
    have = torch.cuda.device_count()
    rc, out, err = _run(["--gpus", str(have + 1), "--steps", "2", "--warmup", "1", "--grid", "32"])
    assert rc != 0
    assert "this box has %d GPU" % have in (out + err)
    assert not any(l.startswith("{") for l in out.splitlines())          # no result line


@pytest.mark.parametrize("world,grid", [(2, 64), (3, 48)])
def test_self_launch_one_device(world, grid):
    rc, out, err = _run(["--gpus", str(world), "--one-device", "--grid", str(grid), "--steps", "5", "--warmup", "2",
                         "--trial-steps", "10"])
    assert rc == 0, (out[-2000:], err[-4000:])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == world and r["steps"] == 5 and r["scaling"] == "strong"
    assert r["config"]["one_device_debug"] is True and r["config"]["backend"] == "gloo"
    d = r["distributed"]
    assert d["transports_tried"]["torch"]["valid"] is True
    assert d["transports_tried"]["ipc"]["valid"] is True, d["transports_tried"]["ipc"]
    assert len(d["per_rank"]) == world and all(p["rows_within_tolerance"] and not p["timed_out"] for p in d["per_rank"])
    if d["transport_chosen"] != "torch":
        assert set(d["slowest_rank_step_ms"]) == {"total", "local", "wait_for_ghosts", "remote", "pack", "exchange"}
    c = r["checksum"]
    assert abs(c["sum_y"] - c["sum_y_independent"]) <= 1e-6 * max(1.0, abs(c["sum_y_independent"]))
