"""hipGraph replay of the UNet forward for launch-bound (small-batch, serving) use.

One denoiser forward is ~700 kernel launches.  Every C-ABI entry point launches on the stream it is given, never
allocates and never synchronises, so a whole forward is capturable: ``GraphedUNet`` runs it once under
``torch.cuda.graph`` (hipGraph) on static input buffers and replays the graph per call -- one host-side launch per
denoising step instead of ~700, which takes the Python / ctypes issue cost off the host.

Measured (tools/latency_bench.py, profiles/r1_latency_eager_vs_graph.txt): replay is bit-identical to eager and, on this
stack, NOT faster -- 6.7 ms per step at batch 1, 9.0 at 4, 34.3 at 32 either way.  The eager path is already GPU-bound at
every batch size (the host issues launches faster than the GPU retires ~10 us kernels), so what limits small-batch latency
is per-kernel time at tiny shapes, not launch issue.  The wrapper is kept for hosts that are busier than a benchmark loop.

The reference has no counterpart (its sampling loop is eager PyTorch, train_latent_diffusion.py:579-593); the wrapper
keeps the call signature ``(sample, timestep, encoder_hidden_states)`` so that loop can use it unchanged.  Limits: fixed
batch / latent extent / context length (one graph per shape), inference only, and the timestep must reach the graph as a
tensor (a Python number is copied into the static buffer first).  The scheduler update stays outside the graph: its
coefficients are host scalars that change every step.
"""
from __future__ import annotations

from typing import Optional

import torch


class GraphedUNet:
    def __init__(self, net, batch: int, latent_hw, ctx_len: int = 77, ctx_dim: int = 768,
                 sample_dtype: torch.dtype = torch.float32, compute_dtype: torch.dtype = torch.bfloat16, device=None):
        p = next(net.parameters())
        self.net = net
        self.device = torch.device(device) if device is not None else p.device
        self.compute_dtype = compute_dtype
        h, w = latent_hw
        cin = net.conv_in.weight.shape[1]
        self.sample = torch.zeros(batch, cin, h, w, dtype=sample_dtype, device=self.device)
        self.t = torch.zeros(batch, dtype=torch.float32, device=self.device)
        self.ctx = torch.zeros(batch, ctx_len, ctx_dim, dtype=compute_dtype, device=self.device)
        self._versions = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out = None
        self._capture()

    def _param_versions(self):
        from .unet import param_state_key
        return param_state_key(self.net)

    def _forward(self):
        if self.compute_dtype == torch.bfloat16 and self.sample.dtype != torch.bfloat16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.net(self.sample, self.t, self.ctx)
        return self.net(self.sample, self.t, self.ctx)

    def _capture(self):
        with torch.no_grad(), torch.cuda.device(self.device):
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):          # warm-up off the capture: packs the weights, sets kernel attributes
                for _ in range(2):
                    self._forward()
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._forward()
            # The graph holds raw device pointers into the packed (derived) filters and no tensor references.  Keep the
            # dict alive here: a training-form forward evicts the inference set from the module's cache
            # (UNet._packed_for(train=True)) without changing any parameter, and a replay would then read freed memory.
            self._packed = self.net._packed_for(self.compute_dtype)
        self._versions = self._param_versions()

    def __call__(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        """Same arguments as ``UNet.forward``.  Returns the graph's static output buffer: consume (or clone) it before the
        next call."""
        if self._versions != self._param_versions():   # weights changed (load_state_dict, optimizer step): re-capture
            self._capture()
        if sample.shape != self.sample.shape or encoder_hidden_states.shape != self.ctx.shape:
            raise ValueError(f"GraphedUNet was captured for sample {tuple(self.sample.shape)} / context {tuple(self.ctx.shape)}; "
                             f"got {tuple(sample.shape)} / {tuple(encoder_hidden_states.shape)}")
        self.sample.copy_(sample, non_blocking=True)
        self.ctx.copy_(encoder_hidden_states, non_blocking=True)
        if torch.is_tensor(timestep):
            self.t.copy_(timestep.reshape(-1).expand(self.t.shape[0]) if timestep.numel() == 1 else timestep, non_blocking=True)
        else:
            self.t.fill_(float(timestep))
        self.graph.replay()
        return self.out
