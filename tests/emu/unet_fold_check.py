"""One-off check on the emulator (tests/emu): the DEFAULT UNet (859.5 M parameters, 320/640/1280 channels: every
TransformerModel takes the LayerNorm fold) on a small latent, entered the way the trainer's validation does -- float32 module,
bf16 autocast region around the call (the entry that faulted the GPU in round 2: DESIGN.md section 9) -- with every buffer the
front ends allocate on guard pages, against (a) the same forward without the fold and (b) the float64 oracle.

    python tests/emu/unet_fold_check.py [latent_hw=8] [--selfcheck]

--selfcheck also runs ldm.ops.ln_fold_selfcheck at its real size (32768 rows: minutes on the emulator); otherwise it is
marked as passed beforehand.  Prints relative L2 errors; exits non-zero on a mismatch.  Not collected by pytest (minutes)."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "latent-diffusion_amd"), HERE, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from conftest import rel_l2, seeded  # noqa: E402
from harness import emulated  # noqa: E402


def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
    from ldm import ops as _ops
    from ldm import unet as U
    from oracle import unet_ref as uref
    cfg = uref.UNetConfig()
    sd = uref.fill_state_dict(uref.param_shapes(cfg), seed=7, **uref.WELL_CONDITIONED)
    net = U.UNet()
    net.load_state_dict(sd)
    net.eval()
    x, ctx = seeded((1, 4, hw, hw), 11), seeded((1, 77, 768), 12)
    t = torch.tensor([500])
    with torch.no_grad():
        ref = uref.unet_forward({k: v.double() for k, v in sd.items()}, cfg, x.double(), t, ctx.double())
    out = {}
    with emulated("lazy", guard=True), torch.no_grad():
        if "--selfcheck" in sys.argv:
            t0 = time.time()
            assert _ops.ln_fold_selfcheck(torch.device("cpu")), "self-check failed on the emulator"
            print(f"ln_fold_selfcheck at its real size: passed ({time.time() - t0:.0f} s)", flush=True)
        else:
            _ops._ln_fold_checked[-1] = True
        for fold in (True, False):
            U._LN_FOLD = fold
            net.invalidate_packed_weights()
            t0 = time.time()
            # the stand-in of torch.autocast("cuda", bf16) on this GPU-less host: the "cuda" autocast state is what the product
            # looks at (unet.compute_dtype_of), the "cpu" autocast makes torch's own ops on these CPU tensors cast the way they
            # would on the device (a matmul inside the region comes back in bf16: the round-2 bug needs exactly that)
            torch.set_autocast_enabled("cuda", True)
            torch.set_autocast_dtype("cuda", torch.bfloat16)
            try:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    assert (torch.ones(2, 2) @ torch.ones(2, 2)).dtype == torch.bfloat16
                    y = net(x, t, ctx)
            finally:
                torch.set_autocast_enabled("cuda", False)
            P = net._packed_for(torch.bfloat16)
            nfold = sum(1 for k, v in P.items() if k.endswith("ln_fold") and v is True)
            print(f"fold={fold}: {nfold} TransformerModels folded, output {y.dtype}, {time.time() - t0:.0f} s, "
                  f"rel-L2 vs float64 oracle {rel_l2(y, ref):.3e}", flush=True)
            assert (nfold == 16) == fold, nfold
            out[fold] = y.float().clone()
    d = rel_l2(out[True], out[False])
    print(f"fold vs no fold: {d:.3e}")
    # two bf16 evaluations with independent rounding differ from each other by about the root sum of squares of their own
    # errors; the folded form must not be worse than the unfused one by more than a quarter
    e_fold, e_plain = rel_l2(out[True], ref), rel_l2(out[False], ref)
    ok = e_fold < max(2e-2, 1.25 * e_plain) and d < 1.5 * (e_fold ** 2 + e_plain ** 2) ** 0.5
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
