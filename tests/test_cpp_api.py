"""The C++ header API (vexcl/*.hpp over libvexhip.so): source-compatible
vex::Context / vector / Reductor / SpMat / sparse:: / scan / sort.

CPU (`-m "not gpu"`): every test program compiles with plain g++ (the headers
need no HIP toolchain), and the expression engine's generated kernels compile
for gfx950 through hiprtc.  GPU: the ports of the reference's own Boost.Test
cases run on a 2-"device" context (tests/cpp/*.cpp name the reference lines)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
GPU_TESTS = ["vector_tests", "spmv_tests", "primitives_tests", "multivector_tests", "expression_tests", "view_tests", "extras_tests"]


def _build(name):
    import vexcl_amd
    if not os.path.exists(vexcl_amd.LIB_PATH):
        vexcl_amd.build()
    subprocess.check_call(["make", "-C", CPP, "-s", "build/" + name])
    return os.path.join(CPP, "build", name)


def test_codegen_and_hiprtc_on_cpu():
    exe = _build("codegen_cpu")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures" in out.stdout


@pytest.mark.parametrize("name", GPU_TESTS)
def test_cpp_programs_compile_without_hip_toolchain(name):
    assert os.path.exists(_build(name))


def test_headers_are_odr_safe(tmp_path):
    # reference: tests/dummy1.cpp + dummy2.cpp -- the headers included from two translation units
    for i in (1, 2):
        (tmp_path / ("tu%d.cpp" % i)).write_text(
            "#include <vexcl/vexcl.hpp>\nint f%d() { vex::vector<double> x; return (int)x.size(); }\n" % i)
    (tmp_path / "main.cpp").write_text("int f1(); int f2(); int main() { return f1() + f2(); }\n")
    exe = str(tmp_path / "odr")
    subprocess.check_call(["g++", "-std=c++17", "-I" + ROOT, str(tmp_path / "tu1.cpp"), str(tmp_path / "tu2.cpp"),
                           str(tmp_path / "main.cpp"), "-o", exe, "-L" + os.path.join(ROOT, "vexcl_amd", "lib"),
                           "-lvexhip", "-Wl,-rpath," + os.path.join(ROOT, "vexcl_amd", "lib")])
    assert subprocess.run([exe]).returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_TESTS)
def test_cpp_api_on_gpu(name):
    exe = _build(name)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(out.stdout[-4000:])
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-3000:]
    assert "0 failures" in out.stdout


@pytest.mark.gpu
def test_every_baseline_config_is_verified_at_its_stated_size():
    """BASELINE.json configs[1] (a = b*c + sin(d), n = 1e8) and configs[4] (vex::sort + vex::inclusive_scan of 1e9 uint32
    keys; sort_by_key at 2.5e8) through the vex:: API at FULL size, every element checked on the host
    (tests/cpp/configs_at_size.cpp): sortedness + multiset equality, scan differences, elementwise against libm."""
    import json
    exe = _build("configs_at_size")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=1800)
    print(out.stdout[-4000:])
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-3000:]
    assert "0 failures" in out.stdout
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    by = {r["what"]: r for r in rows}
    assert by["a = b*c + sin(d)"]["n"] == 100000000 and by["a = b*c + sin(d)"]["outside_tolerance"] == 0
    assert by["vex::sort uint32"]["n"] == 1000000000 and by["vex::sort uint32"]["inversions"] == 0 and by["vex::sort uint32"]["multiset_equal"] is True
    assert by["vex::inclusive_scan uint32"]["n"] == 1000000000 and by["vex::inclusive_scan uint32"]["differences_wrong"] == 0
    assert by["vex::sort_by_key uint32 -> uint32"]["n"] == 250000000 and by["vex::sort_by_key uint32 -> uint32"]["violations"] == 0


def test_configs_at_size_compiles():
    assert os.path.exists(_build("configs_at_size"))


@pytest.mark.gpu
def test_cpp_api_single_device_context():
    exe = _build("spmv_tests")
    env = dict(os.environ, VEX_TEST_SINGLE_DEVICE="1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["vector_tests", "primitives_tests"])
def test_cpp_reductor_combine_through_the_comm_layer(name):
    """VEXCL_REDUCTOR_COMBINE=rccl: the per-device scalars of vex::Reductor are combined by vexhip_allreduce_scalar
    (RCCL between distinct GPUs; on the two logical devices of this box the PEER transport's fold) instead of the host
    fold -- every reduction assertion of the suite must still hold."""
    exe = _build(name)
    env = dict(os.environ, VEXCL_REDUCTOR_COMBINE="rccl")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-3000:]
    assert "0 failures" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["release", "relaxed", "two_launch"])
def test_cpp_reductor_order_modes(order):
    """VEXCL_REDUCTOR_ORDER: the default (`tagged`: every partial travels as 8-byte words that carry the number of the reduction,
    the folding workgroup re-reads a word until it does -- single-object coherence only, correct by the HIP memory model without
    a fence) runs in test_cpp_api_on_gpu; the fence form (`release`), round 4's exchange form and the two-launch form
    (vexhip_reduce_finish as stage 2) must pass the same assertions, the back-to-back stress included: all fold in the same
    order (reference: vexcl/reductor.hpp:412-436, a host fold that has no such hazard)."""
    exe = _build("vector_tests")
    env = dict(os.environ, VEXCL_REDUCTOR_ORDER=order)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-3000:]
    assert "0 failures" in out.stdout


def test_by_key_wave_scan_on_dpp_matches_the_shuffle_form():
    """Host model of the by-key wave scan (vexcl/scan_by_key.hpp): the DPP steps (VEXCL_SBK_DPP=1: row_shr 1/2/4/8, row_bcast:15,
    row_bcast:31) give every live lane the sum the six shuffle steps give it, on random head / live patterns (tools/r04_sbk_dpp_sim.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sbk_dpp_sim", os.path.join(ROOT, "tools", "r04_sbk_dpp_sim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(4000, seed=7)
