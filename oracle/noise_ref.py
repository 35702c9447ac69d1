"""TEST INFRASTRUCTURE (CPU oracle; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it).

numpy restatement of ldm_gaussian_noise (latent-diffusion_amd/csrc/scheduler.hip): per-sample N(0,1) noise as a function of
(seed, step, global sample index, element).  Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11; constants M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9, W1 = 0xBB67AE85) keyed by the 64-bit seed with
counter (element // 4, row lo, row hi, step + 1); the four outputs give two Box-Muller pairs from 23-bit uniforms
u = ((x >> 9) + 0.5) * 2^-23.  The reference has no counterpart: its noise is torch's global generator
(ldm/scheduling_ddpm.py:212, train_latent_diffusion.py:566-576), which a batch-sharded sampler cannot reproduce for any world
size; SURVEY.md 8e asks for noise that is a function of (seed, sample, step).  Pinned by the published Philox known-answer
vectors (tests/test_noise_cpu.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c, key):
    """c: (..., 4) uint32 counters; key: (k0, k1).  Returns (..., 4) uint32."""
    c0, c1, c2, c3 = (c[..., i].astype(np.uint64) for i in range(4))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        c1, c3, c0, c2 = p1 & MASK, p0 & MASK, n0, n2
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def gaussian_rows(seed: int, step: int, row0: int, rows: int, row_elems: int) -> np.ndarray:
    """float32 (rows, row_elems): rows [row0, row0 + rows) of the noise of (seed, step)."""
    quads = (row_elems + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    r = np.arange(row0, row0 + rows, dtype=np.uint64)
    c = np.empty((rows, quads, 4), dtype=np.uint32)
    c[..., 0] = (q & MASK)[None, :]
    c[..., 1] = (r & MASK)[:, None]
    c[..., 2] = (r >> np.uint64(32))[:, None]
    c[..., 3] = np.uint32((step + 1) & 0xFFFFFFFF)
    x = philox4x32_10(c, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    z = np.empty((rows, quads, 4), dtype=np.float32)
    for h in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[..., 2 * h]))
        th = np.float32(6.283185307179586) * u[..., 2 * h + 1]
        z[..., 2 * h] = rad * np.cos(th)
        z[..., 2 * h + 1] = rad * np.sin(th)
    return z.reshape(rows, quads * 4)[:, :row_elems].copy()
