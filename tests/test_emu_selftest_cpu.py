"""The host SIMT emulator (tests/emu) against numpy, one emulated operation at a time: operand layouts of the three MFMA forms,
the transposed LDS read, the permlane swaps, and the LDS-DMA queue with its counted waits and range check.  These pin the
emulator itself (its semantics table: tests/emu/README.md); the product kernels are compared with it in
tests/test_emu_kernels_cpu.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

from harness import emu_lib  # noqa: E402


def _bf16(x):
    """float32 array -> (uint16 bf16 bits, the float32 values those bits stand for), truncation is enough for a test."""
    u = (np.asarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16)
    return u, (u.astype(np.uint32) << 16).view(np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    lib = emu_lib()
    for n in ("mfma16", "mfma32", "mfma_f32", "tr16", "permlane", "dma"):
        getattr(lib, "emu_selftest_" + n).restype = None
    return lib


@pytest.mark.parametrize("name,m,n,k", [("mfma16", 16, 16, 32), ("mfma32", 32, 32, 16)])
def test_bf16_mfma_layouts(lib, name, m, n, k):
    rng = np.random.default_rng(1)
    ab, a = _bf16(rng.standard_normal((m, k)))
    bb, b = _bf16(rng.standard_normal((k, n)))
    c = rng.standard_normal((m, n)).astype(np.float32)
    d = np.zeros((m, n), np.float32)
    getattr(lib, "emu_selftest_" + name)(_p(ab), _p(bb), _p(c), _p(d))
    ref = a.astype(np.float64) @ b.astype(np.float64) + c
    assert np.abs(d - ref).max() < 1e-5 * np.abs(ref).max()


def test_f32_mfma_is_a_k_ordered_fma_chain(lib):
    rng = np.random.default_rng(2)
    a, b = rng.standard_normal((16, 4)).astype(np.float32), rng.standard_normal((4, 16)).astype(np.float32)
    c = rng.standard_normal((16, 16)).astype(np.float32)
    d = np.zeros((16, 16), np.float32)
    lib.emu_selftest_mfma_f32(_p(a), _p(b), _p(c), _p(d))
    ref = a.astype(np.float64) @ b.astype(np.float64) + c
    assert np.abs(d - ref).max() < 2e-6 * np.abs(ref).max()


def test_transposed_lds_read(lib):
    src = np.arange(256, dtype=np.uint16).reshape(16, 16)           # [row][column]
    out = np.zeros((64, 4), np.uint16)
    lib.emu_selftest_tr16(_p(src), _p(out))
    for lane in range(64):
        g, i = divmod(lane, 16)
        assert list(out[lane]) == [src[4 * g + q, i] for q in range(4)], lane


def test_permlane_swaps(lib):
    out = np.zeros((4, 64), np.uint32)
    lib.emu_selftest_permlane(_p(out))
    lane = np.arange(64, dtype=np.uint32)
    vdst, src = lane, 100 + lane
    # v_permlane32_swap: lanes 32..63 of vdst <-> lanes 0..31 of src
    assert np.array_equal(out[0], np.where(lane < 32, vdst, np.roll(src, 32)))
    assert np.array_equal(out[1], np.where(lane < 32, np.roll(vdst, -32), src))
    # v_permlane16_swap: odd 16-lane rows of vdst <-> even rows of src
    odd = (lane // 16) % 2 == 1
    assert np.array_equal(out[2], np.where(odd, np.roll(src, 16), vdst))
    assert np.array_equal(out[3], np.where(odd, src, np.roll(vdst, -16)))


@pytest.mark.parametrize("lazy", [1, 0])
def test_lds_dma_queue_and_range_check(lib, lazy):
    lib.emu_set_dma_mode(lazy)
    try:
        src = np.arange(512, dtype=np.uint32)                        # 2 KiB: both in-range loads fit
        out = np.zeros(768, np.uint32)
        lib.emu_selftest_dma(_p(src), C.c_uint(src.nbytes), _p(out))
    finally:
        lib.emu_set_dma_mode(1)
    assert np.array_equal(out[:256], src[:256])                      # the load the counted wait retired
    if lazy:
        assert np.all(out[256:512] == 0xA5A5A5A5)                    # still in flight: LDS holds its start pattern
    else:
        assert np.array_equal(out[256:512], src[256:512])            # landed at issue
    assert np.all(out[512:] == 0)                                    # out-of-range voffset: zeros
