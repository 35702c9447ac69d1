import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "latent-diffusion_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


EMULATE = os.environ.get("LDM_EMU") == "1"      # run the bodies of the -m gpu tests on the host SIMT emulator (tests/emu)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available() or EMULATE:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


if EMULATE:
    # LDM_EMU=1: the whole session (module-scoped fixtures included) runs inside tests/emu/harness.emulated(guard=True) -- the
    # kernel sources compiled for the host, CPU tensors on guard pages.  LDM_EMU_DMA=eager selects the other LDS-DMA model.
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    from harness import emulated as _emulated
    _emu_session = _emulated(os.environ.get("LDM_EMU_DMA", "lazy"), guard=True)
    _emu_session.__enter__()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
            return {k: (str(z[k]) if z[k].dtype.kind in 'US' else torch.from_numpy(z[k])) for k in z.files}
    return load


def seeded(shape, seed, dtype=torch.float32):
    """Same generator recipe as tests/golden/make_golden.py."""
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=dtype)


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def same_bits(a, b) -> bool:
    """Bit equality with a compact failure report (never dumps tensors into the log)."""
    a = a.detach().cpu()
    b = b.detach().cpu()
    if a.shape != b.shape or a.dtype != b.dtype:
        print(f"shape/dtype mismatch: {tuple(a.shape)} {a.dtype} vs {tuple(b.shape)} {b.dtype}")
        return False
    if a.numel() == 0:
        return True
    ai = a.contiguous().view(torch.int32 if a.element_size() == 4 else (torch.int16 if a.element_size() == 2 else torch.int64))
    bi = b.contiguous().view(ai.dtype)
    ne = ai != bi
    n = int(ne.sum())
    if n:
        idx = ne.flatten().nonzero().flatten()[:3].tolist()
        print(f"{n}/{a.numel()} elements differ; first at flat idx {idx}: "
              f"{[a.flatten()[i].item() for i in idx]} vs {[b.flatten()[i].item() for i in idx]}")
    return n == 0


def max_ulp(a, b) -> int:
    a = a.detach().cpu().contiguous().view(torch.int32).long()
    b = b.detach().cpu().contiguous().view(torch.int32).long()
    return int((a - b).abs().max()) if a.numel() else 0
