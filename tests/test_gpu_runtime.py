"""GPU tests of the runtime half of the C ABI (include/vexhip.h): devices,
streams, events, memory, hiprtc JIT with the on-disk code-object cache, generic
launch -- the pieces that stand in for the reference's backend concept
(backend/cuda/{context,device_vector,kernel,compiler,event}.hpp)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SRC = r'''
extern "C" __global__ void axpb(ulong n, double a, const double *x, double b, double *y) {
  for (ulong i = blockDim.x * (ulong)blockIdx.x + threadIdx.x; i < n; i += blockDim.x * (ulong)gridDim.x)
    y[i] = a * x[i] + b;
}
extern "C" __global__ void lds_reverse(int *x) {
  extern __shared__ int buf[];
  buf[threadIdx.x] = x[threadIdx.x];
  __syncthreads();
  x[threadIdx.x] = buf[blockDim.x - 1 - threadIdx.x];
}
'''


def test_device_props_are_mi355x(built_lib):
    from vexcl_amd._capi import DeviceProps
    n = ctypes.c_int(0)
    built_lib.device_count(ctypes.byref(n))
    assert n.value >= 1
    p = DeviceProps()
    built_lib.device_get_props(0, ctypes.byref(p))
    assert p.arch.decode().startswith("gfx950") and p.wavefront_size == 64 and p.compute_units == 256
    assert p.global_mem_bytes > 200 * 2 ** 30 and p.lds_bytes_per_block >= 64 * 1024
    free, total = ctypes.c_uint64(), ctypes.c_uint64()
    built_lib.mem_info(0, ctypes.byref(free), ctypes.byref(total))
    assert 0 < free.value <= total.value


def test_memory_streams_events(built_lib):
    L = built_lib
    n = 1 << 20
    host = np.arange(n, dtype=np.float64)
    back = np.empty_like(host)
    s1, s2, d1, d2, e = (ctypes.c_void_p() for _ in range(5))
    L.stream_create(0, ctypes.byref(s1)); L.stream_create(0, ctypes.byref(s2))
    L.malloc(0, host.nbytes, ctypes.byref(d1)); L.malloc(0, host.nbytes, ctypes.byref(d2))
    L.event_create(0, 0, ctypes.byref(e))
    L.memcpy_h2d(0, d1, host.ctypes.data_as(ctypes.c_void_p), host.nbytes, s1, 0)   # async on s1
    L.event_record(0, e, s1)
    L.stream_wait_event(0, s2, e)                                                  # enqueue_barrier semantics
    L.memcpy_d2d(0, d2, d1, host.nbytes, s2)
    L.memcpy_d2h(0, back.ctypes.data_as(ctypes.c_void_p), d2, host.nbytes, s2, 1)  # blocking
    assert np.array_equal(back, host)
    L.memset(0, d2, 0, host.nbytes, s2)
    L.memcpy_peer(0, d1, 0, d2, host.nbytes, s2)                                    # same device: plain copy
    L.memcpy_d2h(0, back.ctypes.data_as(ctypes.c_void_p), d1, host.nbytes, s2, 1)
    assert not back.any()
    t0, t1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.event_create(0, 1, ctypes.byref(t0)); L.event_create(0, 1, ctypes.byref(t1))
    L.event_record(0, t0, s1); L.memset(0, d1, 1, host.nbytes, s1); L.event_record(0, t1, s1); L.event_sync(0, t1)
    ms = ctypes.c_float(-1)
    L.event_elapsed_ms(0, t0, t1, ctypes.byref(ms))
    assert ms.value >= 0
    for ev in (e, t0, t1):
        L.event_destroy(0, ev)
    L.free(0, d1); L.free(0, d2); L.stream_destroy(0, s1); L.stream_destroy(0, s2)
    L.free(0, None)                                                                # freeing NULL is fine
    pin = ctypes.c_void_p(); L.host_alloc(4096, ctypes.byref(pin)); L.host_free(pin)


def test_jit_compile_launch_and_disk_cache(built_lib, tmp_path):
    L = built_lib
    os.environ["VEXCL_CACHE_DIR"] = str(tmp_path)
    try:
        c0, h0, c1, h1 = (ctypes.c_uint64() for _ in range(4))
        L.jit_stats(ctypes.byref(c0), ctypes.byref(h0))
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        L.module_compile(0, SRC.encode(), b"", ctypes.byref(mod))
        L.jit_stats(ctypes.byref(c1), ctypes.byref(h1))
        assert c1.value == c0.value + 1 and h1.value == h0.value        # compiled, not found on disk
        L.module_get_function(0, mod, b"axpb", ctypes.byref(fn))
        mt, lds = ctypes.c_int(), ctypes.c_int()
        L.function_max_threads(0, fn, ctypes.byref(mt), ctypes.byref(lds))
        assert mt.value >= 256 and lds.value == 0
        n = 100003
        x = np.random.default_rng(1).random(n)
        dx, dy = ctypes.c_void_p(), ctypes.c_void_p()
        L.malloc(0, x.nbytes, ctypes.byref(dx)); L.malloc(0, x.nbytes, ctypes.byref(dy))
        L.memcpy_h2d(0, dx, x.ctypes.data_as(ctypes.c_void_p), x.nbytes, None, 1)
        args = [ctypes.c_uint64(n), ctypes.c_double(2.5), dx, ctypes.c_double(-1.0), dy]
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        L.launch(0, fn, 64, 1, 1, 256, 1, 1, 0, None, arr)
        y = np.empty_like(x)
        L.memcpy_d2h(0, y.ctypes.data_as(ctypes.c_void_p), dy, x.nbytes, None, 1)
        assert np.allclose(y, 2.5 * x - 1.0, rtol=1e-15, atol=1e-15)      # hiprtc contracts a*x+b into an FMA
        # dynamic LDS through the launch call
        fr = ctypes.c_void_p(); L.module_get_function(0, mod, b"lds_reverse", ctypes.byref(fr))
        v = np.arange(256, dtype=np.int32)
        dv = ctypes.c_void_p(); L.malloc(0, v.nbytes, ctypes.byref(dv))
        L.memcpy_h2d(0, dv, v.ctypes.data_as(ctypes.c_void_p), v.nbytes, None, 1)
        a2 = (ctypes.c_void_p * 1)(ctypes.cast(ctypes.pointer(dv), ctypes.c_void_p))
        L.launch(0, fr, 1, 1, 1, 256, 1, 1, 1024, None, a2)
        L.memcpy_d2h(0, v.ctypes.data_as(ctypes.c_void_p), dv, v.nbytes, None, 1)
        assert np.array_equal(v, np.arange(255, -1, -1))
        # second compile of the same source: code object comes from the disk cache
        mod2 = ctypes.c_void_p()
        L.module_compile(0, SRC.encode(), b"", ctypes.byref(mod2))
        L.jit_stats(ctypes.byref(c0), ctypes.byref(h0))
        assert c0.value == c1.value and h0.value == h1.value + 1
        assert any(f == "kernel.hsaco" for _, _, fs in os.walk(tmp_path) for f in fs)
        L.module_unload(0, mod); L.module_unload(0, mod2)
        for p in (dx, dy, dv):
            L.free(0, p)
    finally:
        del os.environ["VEXCL_CACHE_DIR"]


def test_errors_carry_file_line_and_hip_text(built_lib):
    from vexcl_amd import Error
    with pytest.raises(Error) as ei:
        built_lib.module_compile(0, b"this is not HIP", b"", ctypes.byref(ctypes.c_void_p()))
    assert "hiprtc" in str(ei.value) and ".hip:" in str(ei.value)
    with pytest.raises(Error) as ei:
        built_lib.device_get_props(99, ctypes.byref(__import__("vexcl_amd")._capi.DeviceProps()))
    assert "hip" in str(ei.value).lower()
    with pytest.raises(Error):
        built_lib.sort(0, None, 3, 0, None, None, 0, None, None, 10, None)          # NULL buffers are refused
