#!/bin/bash
# Builds the REFERENCE'S OWN test programs against this repository's vexcl/ headers.
#
# Test infrastructure.  The sources are compiled where they lie (/root/reference/tests/*.cpp, never
# copied); Boost.Test is replaced by oracle/ref_shim/boost/test/unit_test.hpp (Boost is not in this
# image).  Outputs go to oracle/_ref/ only (git-ignored, shipped to the GPU box by gpurun), where
# tests/test_reference_suite.py runs them.  A test that builds and passes demonstrates source
# compatibility of the vex:: API with the reference's own callers and the reference's own
# assertions on the results.
#
# -DVEXCL_BACKEND_CUDA: the reference's tests use that macro to leave out what its CUDA backend cannot
# run -- user functions written in OpenCL C syntax (`(int4)(x, x, x, x)`), the constant address space.
# Kernels here are HIP C++, so the same cases are left out; nothing in vexcl/ looks at the macro.
#
# usage: oracle/build_ref.sh [-j N] [name ...]     (default: every test listed in TESTS)
set -u
here="$(cd "$(dirname "$0")" && pwd)"
repo="$(dirname "$here")"
ref="${VEXCL_REFERENCE:-/root/reference}"
out="$here/_ref"
jobs=8
if [ "${1:-}" = "-j" ]; then jobs="$2"; shift 2; fi

# Every test source of the reference except image (OpenCL images / CUDA texture objects through the vendor API),
# cusparse, boost_compute_*, clogs_* (adapters to other libraries) and boost_version (prints Boost's version).
TESTS="${*:-vector_create vector_copy vector_arithmetics vector_view vector_pointer vector_io \
tagged_terminal temporary cast constants logical reinterpret types eval events \
multivector_create multivector_arithmetics spmv sparse_matrices stencil random sort scan \
scan_by_key reduce_by_key context threads custom_kernel tensordot multi_array mba deduce svm generator fft}"

if [ ! -d "$ref/tests" ]; then echo "build_ref: $ref/tests not present, nothing to do"; exit 0; fi
mkdir -p "$out"
build_one() {
    name="$1"
    if g++ -std=c++17 -O1 -w -DVEXCL_BACKEND_CUDA -I "$here/ref_shim" -I "$repo" "$ref/tests/$name.cpp" -o "$out/$name" \
        -L "$repo/vexcl_amd/lib" -lvexhip -Wl,-rpath,'$ORIGIN/../../vexcl_amd/lib' -pthread 2> "$out/$name.build.log"; then
        rm -f "$out/$name.build.log"; echo "built   $name"
    else
        rm -f "$out/$name"; echo "FAILED  $name ($(grep -c 'error' "$out/$name.build.log") error lines, see oracle/_ref/$name.build.log)"
    fi
}
# The reference's example programs that need nothing but vex:: and the small stand-ins of oracle/ref_shim (Boost.Test,
# program_options, ios_state, two Phoenix placeholders, odeint's Runge-Kutta stepper; not cuFFT / ViennaCL): built the same way as example_<name>; the pytest module checks that they run to completion.
EXAMPLES="benchmark fft_benchmark symbolic devlist complex_simple complex_spmv mba_benchmark fft_profile exclusive simple/hello"
build_example() {
    src="$1"; name="example_$(basename "$1")"
    if g++ -std=c++17 -O1 -w -DVEXCL_BACKEND_CUDA -I "$here/ref_shim" -I "$repo" "$ref/examples/$src.cpp" -o "$out/$name" \
        -L "$repo/vexcl_amd/lib" -lvexhip -Wl,-rpath,'$ORIGIN/../../vexcl_amd/lib' -pthread 2> "$out/$name.build.log"; then
        rm -f "$out/$name.build.log"; echo "built   $name"
    else
        rm -f "$out/$name"; echo "FAILED  $name (see oracle/_ref/$name.build.log)"
    fi
}
# The reference's size checks are a compile-time option (VEXCL_CHECK_SIZES, CMakeLists.txt:23); the tests that hold
# `#if (VEXCL_CHECK_SIZES > 0)` cases are built a second time with it as <name>_checked.
CHECKED="vector_arithmetics vector_view multivector_arithmetics"
build_checked() {
    name="$1"
    if g++ -std=c++17 -O1 -w -DVEXCL_BACKEND_CUDA -DVEXCL_CHECK_SIZES=2 -I "$here/ref_shim" -I "$repo" "$ref/tests/$name.cpp" -o "$out/${name}_checked" \
        -L "$repo/vexcl_amd/lib" -lvexhip -Wl,-rpath,'$ORIGIN/../../vexcl_amd/lib' -pthread 2> "$out/${name}_checked.build.log"; then
        rm -f "$out/${name}_checked.build.log"; echo "built   ${name}_checked"
    else
        rm -f "$out/${name}_checked"; echo "FAILED  ${name}_checked (see oracle/_ref/${name}_checked.build.log)"
    fi
}
export -f build_one build_example build_checked; export here repo ref out
printf '%s\n' $TESTS | xargs -P "$jobs" -I{} bash -c 'build_one {}'
if [ $# -eq 0 ]; then
    printf '%s\n' $EXAMPLES | xargs -P "$jobs" -I{} bash -c 'build_example {}'
    printf '%s\n' $CHECKED | xargs -P "$jobs" -I{} bash -c 'build_checked {}'
fi
# Fixtures from reference-run code (the reference's generators + the host loop of tests/spmv.cpp:28-32): pins oracle/vex_oracle.c
# directly (tests/test_oracle.py::test_oracle_against_reference_run_fixtures).  The .npz under tests/golden/ is committed; it is
# rewritten here only when VEXCL_REF_FIXTURES=1 (std::default_random_engine is the same stream for one libstdc++).
if [ $# -eq 0 ]; then
    if g++ -std=c++17 -O1 -w -I "$ref/tests" -I "$repo" "$here/ref_fixture_driver.cpp" -o "$out/ref_fixture_driver" 2> "$out/ref_fixture_driver.build.log"; then
        rm -f "$out/ref_fixture_driver.build.log"; echo "built   ref_fixture_driver"
        "$out/ref_fixture_driver" "$out/ref_fixtures.bin" && [ "${VEXCL_REF_FIXTURES:-0}" = "1" ] && python3 "$here/ref_fixtures_to_npz.py"
    else
        echo "FAILED  ref_fixture_driver (see oracle/_ref/ref_fixture_driver.build.log)"
    fi
fi
ls "$out" | grep -v '\.log$' | grep -v '^HEADERS_SHA256$' | grep -v '^ref_fixture' > "$out/MANIFEST" || true
# the binaries embed the vexcl/ headers (code generators included): record what they were built from, so that
# tests/test_reference_suite.py can refuse to run binaries that are older than the headers they claim to test
python3 "$here/headers_hash.py" > "$out/HEADERS_SHA256"
exit 0
