"""Runs the product's Python front ends (ldm.ops, ldm.unet ...) on CPU tensors against libldm_emu.so, the kernel sources
compiled for the host on the SIMT emulator (tests/emu/emu.h).  TEST INFRASTRUCTURE: the patches below exist only inside
the ``emulated()`` context of a test; the product has no CPU path and refuses CPU tensors everywhere else."""
import contextlib
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import build_emu  # noqa: E402

_emu = None


def emu_lib():
    """libldm_emu.so with the product's ctypes signatures (built on first use)."""
    global _emu
    if _emu is not None:
        return _emu
    from ldm import _lib
    path = build_emu.build()
    lib = C.CDLL(path)
    lib.ldm_abi_version.restype = C.c_int
    lib.ldm_abi_version.argtypes = []
    lib.ldm_last_error.restype = C.c_char_p
    lib.ldm_last_error.argtypes = []
    for name, argtypes in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    lib.ldm_conv_wgrad_workspace_bytes.restype = C.c_longlong
    lib.ldm_conv_wgrad_workspace_bytes.argtypes = [C.POINTER(_lib.ConvGemmDesc)]
    for name, argtypes in _lib._LONGLONG_QUERIES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_longlong
        fn.argtypes = argtypes
    lib.emu_set_dma_mode.argtypes = [C.c_int]
    lib.emu_set_dma_mode.restype = None
    lib.emu_alloc_guarded.argtypes = [C.c_size_t]
    lib.emu_alloc_guarded.restype = C.c_void_p
    lib.emu_free_guarded.argtypes = [C.c_void_p]
    lib.emu_free_guarded.restype = None
    lib.emu_launch_count.restype = C.c_longlong
    lib.emu_waveop_count.restype = C.c_longlong
    assert lib.ldm_abi_version() == _lib.ABI_VERSION
    _emu = lib
    return lib


class _NullStream:
    cuda_stream = 0

    def wait_stream(self, other):
        pass

    def synchronize(self):
        pass


_leaky = None


def _guarded_empty(*size, dtype=None, device=None, **kw):
    """torch.empty on guarded pages (never freed: the tensors escape into the test)."""
    global _leaky
    if _leaky is None:
        _leaky = Guarded()
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    t = _leaky.empty(tuple(int(v) for v in size), dtype or torch.get_default_dtype())
    if t.numel():      # 0xFF bytes: NaN in float32 / bfloat16, -1 in integers -- an element a kernel forgets to write shows up
        t.reshape(-1).view(torch.uint8).fill_(0xFF)
    return t


_FACTORIES = ("randn", "rand", "zeros", "ones", "empty", "full", "tensor", "arange", "randint", "zeros_like", "ones_like",
              "empty_like", "randn_like", "as_tensor", "linspace", "eye")


def _cpu_device(fn):
    """fn with device="cuda..." rewritten to the CPU (the GPU tests name the device in their factory calls)."""
    def wrapped(*a, **k):
        d = k.get("device")
        if d is not None and str(d).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


class _CpuGenerator(torch.Generator):
    def __new__(cls, device="cpu"):
        return super().__new__(cls, "cpu")


_real_autocast = torch.autocast


class _Autocast:
    """torch.autocast for the emulated session: a "cuda" region sets the CUDA autocast state the product looks at
    (unet.compute_dtype_of) -- torch itself refuses to on a host without the device -- and opens a CPU autocast region of the
    same dtype, so that torch's own ops on the CPU tensors cast the way they would on the device."""

    def __init__(self, device_type, dtype=None, enabled=True, cache_enabled=None):
        self.cuda = str(device_type).startswith("cuda")
        self.dtype, self.enabled = dtype, enabled
        self.inner = _real_autocast("cpu", dtype=dtype if dtype is not None else torch.bfloat16, enabled=enabled) if self.cuda \
            else _real_autocast(device_type, dtype=dtype, enabled=enabled, cache_enabled=cache_enabled)

    def __enter__(self):
        if self.cuda:
            self.prev = (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"))
            torch.set_autocast_enabled("cuda", bool(self.enabled))
            if self.dtype is not None:
                torch.set_autocast_dtype("cuda", self.dtype)
        self.inner.__enter__()
        return self

    def __exit__(self, *exc):
        self.inner.__exit__(*exc)
        if self.cuda:
            torch.set_autocast_enabled("cuda", self.prev[0])
            torch.set_autocast_dtype("cuda", self.prev[1])
        return False


@contextlib.contextmanager
def emulated(dma="lazy", guard=False):
    """Inside: ldm._lib.load() returns the emulator library and ldm.ops accepts CPU tensors.  dma: "lazy" (an LDS-DMA
    lands when a counted wait retires it) or "eager" (when it is issued).  guard: every torch.empty (the outputs and
    workspaces the front ends allocate) ends on a PROT_NONE page."""
    from ldm import _lib, ops
    lib = emu_lib()
    lib.emu_set_dma_mode(1 if dma == "lazy" else 0)
    saved = (_lib._lib, ops._gpu, ops._stream, torch.cuda.device, torch.cuda.current_stream, torch.empty)
    saved_cuda = (torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.synchronize)
    saved_cap = torch.cuda.is_current_stream_capturing
    saved_pin = torch.Tensor.pin_memory
    saved_mem = (torch.cuda.memory_allocated, torch.cuda.memory_reserved, torch.cuda.mem_get_info)
    torch.cuda.memory_allocated = lambda *a, **k: 0
    torch.cuda.memory_reserved = lambda *a, **k: 0
    torch.cuda.mem_get_info = lambda *a, **k: (1 << 38, 1 << 38)
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.cuda.is_current_stream_capturing = lambda: False
    torch.Tensor.cuda = lambda self, *a, **k: self          # the bodies of the GPU tests run unchanged on CPU tensors
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.Tensor.is_cuda = property(lambda self: True)      # shadows TensorBase.is_cuda for the product's device checks
    if guard:
        torch.empty = _guarded_empty
    saved_fact = {n: getattr(torch, n) for n in _FACTORIES}
    for n in _FACTORIES:
        setattr(torch, n, _cpu_device(getattr(torch, n)))
    saved_gen, saved_to = torch.Generator, torch.Tensor.to
    torch.Generator = _CpuGenerator
    torch.autocast = _Autocast

    def _to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, (str, torch.device)) and str(x).startswith("cuda")) else x for x in a)
        if "device" in k and k["device"] is not None and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return saved_to(self, *a, **k)
    torch.Tensor.to = _to
    _lib._lib = lib
    ops._gpu = lambda *ts: None
    ops._stream = lambda t: 0
    torch.cuda.device = lambda dev: contextlib.nullcontext()
    torch.cuda.current_stream = lambda device=None: _NullStream()
    try:
        yield ops
    finally:
        _lib._lib, ops._gpu, ops._stream, torch.cuda.device, torch.cuda.current_stream, torch.empty = saved
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.synchronize = saved_cuda
        torch.cuda.is_current_stream_capturing = saved_cap
        torch.Tensor.pin_memory = saved_pin
        torch.cuda.memory_allocated, torch.cuda.memory_reserved, torch.cuda.mem_get_info = saved_mem
        del torch.Tensor.is_cuda
        for n, f in saved_fact.items():
            setattr(torch, n, f)
        torch.Generator, torch.Tensor.to = saved_gen, saved_to
        torch.autocast = _real_autocast
        lib.emu_set_dma_mode(1)


class Guarded:
    """CPU tensors whose storage ends on a PROT_NONE page: one byte past the end is a SIGSEGV (and one page before the
    start too).  Keep the object alive as long as the tensors are used."""

    def __init__(self):
        self._ptrs = []

    def like(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        lib = emu_lib()
        p = lib.emu_alloc_guarded(max(nbytes, 1))
        assert p
        self._ptrs.append(p)
        if nbytes == 0:
            return t.clone()
        buf = (C.c_char * nbytes).from_address(p)
        g = torch.frombuffer(buf, dtype=torch.uint8, count=nbytes).view(t.dtype).view(t.shape)
        g.copy_(t)
        return g

    def empty(self, shape, dtype) -> torch.Tensor:
        return self.like(torch.zeros(shape, dtype=dtype))

    def close(self):
        lib = emu_lib()
        for p in self._ptrs:
            lib.emu_free_guarded(p)
        self._ptrs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
