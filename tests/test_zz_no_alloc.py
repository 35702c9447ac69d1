"""ldm_groupnorm_noalloc (include/ldm_amd.h): the allocation-free variant of ldm_groupnorm, whose partial-sum workspace is a
static device array inside the library.  Called through the C-ABI (ctypes) and compared bit for bit with ldm_groupnorm on the
same inputs, twice in a row, the second call on a smaller input (the array keeps the first call's contents), on shapes that
take the three-kernel form (float32; bf16 960 channels at 64x64) and one that takes the one-kernel form.  The call must not
allocate device memory: once the entry point has run on the device (its first call looks the scratch array's address up, which
may load the code object), the DRIVER's free-memory figure (hipMemGetInfo -- a hipMalloc inside the library would never pass
through torch's allocator) and torch's allocator statistics are unchanged across a call.  A captured and replayed call
(torch.cuda.graph) gives the same bits."""
import ctypes as C

import pytest
import torch

from conftest import same_bits, seeded
from ldm import _lib

pytestmark = pytest.mark.gpu

MAX_FLOATS = 1 << 20


def _run(fn_noalloc, dtype, B, hw, c0, c1, groups, silu, seed):
    lib = _lib.load()
    dev = torch.device("cuda")
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    a0 = seeded((B, hw, c0), seed).to(dev).to(dtype)
    a1 = seeded((B, hw, c1), seed + 1).to(dev).to(dtype) if c1 else None
    gamma = (1.0 + 0.1 * seeded((c0 + c1,), seed + 2)).to(dev)
    beta = (0.05 * seeded((c0 + c1,), seed + 3)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for noalloc in (False, True):
        out = torch.full((B, hw, c0 + c1), float("nan"), dtype=dtype, device=dev)
        stats = torch.full((B, groups, 2), float("nan"), dtype=torch.float32, device=dev)
        args = [a0.data_ptr(), 0 if a1 is None else a1.data_ptr(), c0, c1, code, B, hw, groups, 1e-5, gamma.data_ptr(),
                beta.data_ptr(), 1 if silu else 0, out.data_ptr(), stats.data_ptr()]
        if noalloc:
            _lib.check(fn_noalloc(*args, st), "ldm_groupnorm_noalloc")       # warm: symbol lookup / code-object load done
            torch.cuda.synchronize()
            before = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved(), torch.cuda.mem_get_info()[0])
            out.fill_(float("nan"))
            stats.fill_(float("nan"))
            _lib.check(fn_noalloc(*args, st), "ldm_groupnorm_noalloc")
            torch.cuda.synchronize()
            after = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved(), torch.cuda.mem_get_info()[0])
            assert before == after, f"device memory moved across the call: {before} -> {after}"
        else:
            ws = torch.empty(lib.ldm_groupnorm_workspace_floats(B, groups), dtype=torch.float32, device=dev)
            _lib.check(lib.ldm_groupnorm(*args, ws.data_ptr(), st), "ldm_groupnorm")
        outs.append((out, stats))
    (o_ref, s_ref), (o_new, s_new) = outs
    assert torch.isfinite(o_ref.float()).all() and torch.isfinite(s_ref).all()
    assert same_bits(o_new, o_ref) and same_bits(s_new, s_ref), (dtype, B, hw, c0, c1)


def test_groupnorm_noalloc_matches_groupnorm_twice_in_a_row():
    lib = _lib.load()
    fn = lib.ldm_groupnorm_noalloc
    # first call: the larger inputs; second call right behind it: smaller ones (fewer samples, fewer slabs)
    _run(fn, torch.float32, 6, 32 * 32, 64, 0, 32, True, 100)        # float32: always the three-kernel form
    _run(fn, torch.float32, 2, 8 * 8, 32, 32, 32, False, 110)        # smaller, two-source concat
    _run(fn, torch.bfloat16, 4, 64 * 64, 640, 320, 32, True, 120)    # bf16 960 channels at 64x64: three-kernel form
    _run(fn, torch.bfloat16, 1, 64 * 64, 640, 320, 32, True, 130)    # smaller batch
    _run(fn, torch.bfloat16, 3, 16 * 16, 1280, 0, 32, True, 140)     # one-kernel form (scratch untouched)
    _run(fn, torch.float32, 1, 7 * 5, 32, 0, 8, False, 150)          # ragged tiny input after the large ones


def test_groupnorm_noalloc_rejects_oversize_and_empty():
    lib = _lib.load()
    dev = torch.device("cuda")
    x = torch.zeros(8, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    # batch * groups * 128 floats > the built-in buffer
    B, groups = 300, 32
    assert lib.ldm_groupnorm_workspace_floats(B, groups) > MAX_FLOATS
    rc = lib.ldm_groupnorm_noalloc(x.data_ptr(), 0, 32, 0, _lib.F32, B, 4, groups, 1e-5, x.data_ptr(), x.data_ptr(), 0,
                                   x.data_ptr(), x.data_ptr(), st)
    assert rc == -1 and b"scratch floats" in lib.ldm_last_error()
    # the largest admitted size passes the size check (256 x 32 groups = the buffer exactly) and an empty batch is a no-op
    assert lib.ldm_groupnorm_workspace_floats(256, 32) == MAX_FLOATS
    assert lib.ldm_groupnorm_noalloc(0, 0, 32, 0, _lib.F32, 0, 4, 32, 1e-5, 0, 0, 0, 0, 0, st) == 0
    torch.cuda.synchronize()


def test_groupnorm_noalloc_captured_and_replayed():
    """The documented use: one ordinary call per device, then the call is captured into a graph (no allocation, no host
    synchronisation, no symbol lookup under capture) and the replay reproduces the eager result bit for bit."""
    lib = _lib.load()
    dev = torch.device("cuda")
    B, hw, c, groups = 2, 32 * 32, 64, 32
    a0 = seeded((B, hw, c), 200).to(dev)
    gamma, beta = (1.0 + 0.1 * seeded((c,), 201)).to(dev), (0.05 * seeded((c,), 202)).to(dev)
    ref, ref_stats = torch.empty_like(a0), torch.empty((B, groups, 2), device=dev)
    out, stats = torch.zeros_like(a0), torch.zeros((B, groups, 2), device=dev)

    def call(o, s, stream):
        _lib.check(lib.ldm_groupnorm_noalloc(a0.data_ptr(), 0, c, 0, _lib.F32, B, hw, groups, 1e-5, gamma.data_ptr(),
                                             beta.data_ptr(), 1, o.data_ptr(), s.data_ptr(), stream), "ldm_groupnorm_noalloc")

    call(ref, ref_stats, torch.cuda.current_stream().cuda_stream)       # the ordinary call (also resolves the scratch address)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(out, stats, torch.cuda.current_stream().cuda_stream)
    out.zero_()
    stats.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert same_bits(out, ref) and same_bits(stats, ref_stats)
