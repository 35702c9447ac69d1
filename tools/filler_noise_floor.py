"""bf16 noise floor of the small (config-1) UNet under different weight fillers: torch-CPU bf16 and float32 evaluations of the
oracle against its float64 evaluation (host only; DESIGN.md section 2 quotes the table).  Usage: python tools/filler_noise_floor.py"""
import sys, torch, math
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_ref as uref
torch.manual_seed(0)
cfg = uref.UNetConfig(block_out_channels=[32,64,64,64])
shapes = uref.param_shapes(cfg)
g = torch.Generator().manual_seed(0)
x = torch.randn(4,4,32,32,generator=g); ctx = torch.randn(4,77,768,generator=g); t = torch.tensor([999,500,20,0])
def run(sd):
    with torch.no_grad():
        r64 = uref.unet_forward({k:v.double() for k,v in sd.items()}, cfg, x.double(), t, ctx.double())
        r32 = uref.unet_forward(sd, cfg, x, t, ctx).double()
        rb = uref.unet_forward({k:v.bfloat16() for k,v in sd.items()}, cfg, x.bfloat16(), t, ctx.bfloat16()).double()
    e = lambda a: float((a-r64).norm()/r64.norm())
    return e(r32), e(rb)
for name, kw, post in [("plain gain=1", dict(gain=1.0), None),
                 ("well-conditioned (gate x0.25, +2)", uref.WELL_CONDITIONED, None),
                 ("proj_out & ff.3 x 3", dict(gain=1.0), 3.0),
                 ("proj_out & ff.3 x 0.3", dict(gain=1.0), 0.3),
                 ("default-init variance (gain 1/sqrt3)", dict(gain=1/math.sqrt(3)), None)]:
    sd = uref.fill_state_dict(shapes, seed=7, **kw)
    if post:
        for k in sd:
            if k.endswith("proj_out.weight") or k.endswith("feed_forward.3.weight"): sd[k] = sd[k]*post
    print(f"{name:40s} fp32 vs f64 {run(sd)[0]:.2e}   torch bf16 vs f64 {run(sd)[1]:.3e}")
print("--- gate variants (scale, shift)")
for sc, sh in [(0.5,1.0),(0.5,0.5),(1.0,1.0),(0.35,1.0),(0.25,1.0)]:
    sd = uref.fill_state_dict(shapes, seed=7, gain=1.0, glu_gate=(sc,sh))
    a,b = run(sd)
    print(f"gate x{sc} +{sh}: fp32 {a:.2e} torch bf16 {b:.3e}")
