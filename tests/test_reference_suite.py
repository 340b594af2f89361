"""GPU: the REFERENCE'S OWN test programs, compiled unchanged against this repository's vexcl/ headers.

oracle/build_ref.sh (run by __graft_entry__.build() where /root/reference is present) compiles
/root/reference/tests/<name>.cpp in place -- Boost.Test replaced by oracle/ref_shim -- into
oracle/_ref/<name>.  Those binaries travel to the GPU box; this module only RUNS them (it never reads
/root/reference), so every assertion the reference makes about vex:: results is checked against the
HIP path.  Where the reference is present (/root/reference: the build container) missing or stale binaries FAIL; a
checkout that never saw the reference skips.  The binaries embed the vexcl/ headers, so they are only valid for the
headers they were built from: oracle/_ref/HEADERS_SHA256 must match the tree (checked on CPU and on the GPU box).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _names():
    manifest = os.path.join(REF, "MANIFEST")
    if not os.path.exists(manifest):
        return []
    with open(manifest) as f:
        return [l.strip() for l in f if l.strip() and l.strip() != "MANIFEST"]


NAMES = _names()
HAVE_REFERENCE = os.path.isdir("/root/reference/tests")


def _stale():
    """None if oracle/_ref was built from the headers in the tree, else a message."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from headers_hash import headers_hash
    rec = os.path.join(REF, "HEADERS_SHA256")
    if not os.path.exists(rec):
        return "oracle/_ref/HEADERS_SHA256 is missing: run oracle/build_ref.sh (python -c 'import __graft_entry__ as g; g.build()')"
    if open(rec).read().strip() != headers_hash():
        return "oracle/_ref was built from other vexcl/ headers than the tree holds: rebuild with oracle/build_ref.sh"
    return None


@pytest.mark.gpu
@pytest.mark.skipif(not NAMES, reason="oracle/_ref holds no reference test binaries (oracle/build_ref.sh was not run)")
@pytest.mark.parametrize("name", NAMES or ["none"])
def test_reference_program(name):
    assert _stale() is None, _stale()
    exe = os.path.join(REF, name)
    env = dict(os.environ)
    args = {"example_mba_benchmark": ["65536"],
            "example_benchmark": ["--bm_cpu", "0"],                     # BASELINE.json configs[0]; the host-CPU comparison loops are skipped
            "example_fft_benchmark": ["--runs", "20", "--max", "65536", "-p", "-c"]}.get(name, [])
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT, stdin=subprocess.DEVNULL)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, f"reference program {name} failed:\n{tail}"
    if not name.startswith("example_"):          # the reference's examples/*.cpp print results, not a test summary
        assert "0 failures" in r.stdout, tail


def test_every_listed_reference_program_built():
    """CPU: where the reference is present the binaries must exist, be current and complete -- a missing MANIFEST, a
    stale build or a program that no longer compiles against vexcl/ is a failure, not a skip.  (A failed build leaves
    <name>.build.log behind and drops the program from the GPU run.)"""
    if not os.path.isdir(REF) or not NAMES:
        if HAVE_REFERENCE:
            pytest.fail("/root/reference is present but oracle/_ref holds no reference test binaries: run oracle/build_ref.sh "
                        "(python -c 'import __graft_entry__ as g; g.build()')")
        pytest.skip("oracle/_ref absent and no reference in this checkout")
    assert _stale() is None, _stale()
    failed = sorted(f for f in os.listdir(REF) if f.endswith(".build.log"))
    assert not failed, "reference programs that no longer compile: %s" % failed
    assert len(NAMES) >= 48, "expected 35 test programs, 3 of them a second time with VEXCL_CHECK_SIZES, and 10 examples; found %d" % len(NAMES)
