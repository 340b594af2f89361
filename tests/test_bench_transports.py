"""CPU: bench.py's `--transport auto` never dies of a transport (round 6, review item 1d): a candidate that raises while it is being
enabled, validated, timed or disabled, one that times out (invalid), and one that validates but cannot be enabled a second time are all
skipped; the fastest valid one is chosen.  The selection logic is exercised with fakes -- the transports themselves need GPUs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_a_failing_transport_is_skipped_never_fatal():
    import bench
    state = {"on": None, "log": []}
    behaviour = {"halo": "raises_on_enable", "ipc": "times_out", "rccl": "ok_slow", "peer": "raises_in_trial", "fast": "ok_fast", "odd": "raises_on_disable"}

    def enable(tr):
        state["on"] = tr; state["log"].append(("enable", tr))
        if behaviour[tr] == "raises_on_enable":
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument")
        return True

    def validate():
        if behaviour[state["on"]] == "times_out":
            return False, float("nan")
        return True, 1e-12

    def trial():
        b = behaviour[state["on"]]
        if b == "raises_in_trial":
            raise RuntimeError("an earlier product of this plan timed out")
        return {"ok_slow": 0.2, "ok_fast": 0.05, "raises_on_disable": 0.1}[b]

    def disable():
        state["log"].append(("disable", state["on"]))
        if behaviour[state["on"]] == "raises_on_disable":
            raise RuntimeError("hipFree failed")

    tried = bench.try_transports(list(behaviour), enable, validate, trial, disable, lambda: "flag wait ran into its bound")
    assert set(tried) == set(behaviour)
    assert not tried["halo"]["valid"] and "hipIpcOpenMemHandle" in tried["halo"]["error"]
    assert not tried["ipc"]["valid"] and tried["ipc"]["error"] == "flag wait ran into its bound"
    assert not tried["peer"]["valid"] and "timed out" in tried["peer"]["error"]
    assert tried["rccl"]["valid"] and tried["fast"]["valid"] and tried["odd"]["valid"] and "disable_error" in tried["odd"]
    # every candidate was disabled again, whatever happened to it
    assert [t for k, t in state["log"] if k == "disable"] == list(behaviour)
    tried["torch"] = {"valid": True, "trial_ms_per_step": 150.0}
    tried["untimed"] = {"valid": True}
    assert bench.transports_by_time(tried) == ["fast", "odd", "rccl", "torch", "untimed"]
    assert bench.transports_by_time({"a": {"valid": False}}) == []
