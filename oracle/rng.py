"""ORACLE (test infrastructure, never shipped): numpy restatement of the device's counter-based RNG.

The reference draws its 34 dropout masks and the reparameterisation noise from torch's CPU generator
(pace.py:1659-1662, nn.Dropout), which no device can reproduce.  The HIP path instead uses a stateless hash of
(seed, site, global DAG index, element) — dags_vae_search_amd/csrc/dvs_device.h: dvs_fmix32, dvs_site_key, dvs_draw —
so that the backward pass regenerates masks and a sharded batch draws the same bits as the same batch on one GPU.
This file restates those functions so that parity tests can run the oracle with the SAME masks (dropout-on parity
is then exact up to fp32 rounding).  The stream is the build's own choice; only its distribution (Bernoulli keep
probability 1 - round(p*65536)/65536, standard normal) matters for agreement with the reference, and that is
tested statistically.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return (np.asarray(x, dtype=np.uint64) & M32)


def fmix32(x):
    x = _u32(x)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M32
    x ^= x >> np.uint64(16)
    return x


def site_key(seed: int, site: int, dag):
    seed_lo, seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    k = fmix32(np.uint64(seed_lo ^ (((site + 1) * 0x632BE5AB) & 0xFFFFFFFF)))
    dag = _u32(dag)
    return fmix32(k ^ np.uint64(seed_hi) ^ ((dag * np.uint64(0x9E3779B1)) & M32))


def draw(key, pair):
    return fmix32(_u32(key) ^ ((_u32(pair) * np.uint64(0x9E3779B1)) & M32))


def keep_elements(key, elems, thr16):
    """Bernoulli keep decision for element indices `elems` (same shape broadcast with key)."""
    h = draw(key, _u32(elems) >> np.uint64(1))
    half = np.where((_u32(elems) & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return half >= np.uint64(thr16)


class DeviceMasks:
    """Dropout masks (already multiplied by 1/keep) and eps exactly as the HIP kernels generate them."""

    def __init__(self, seed: int, p: float, dag_offset: int = 0, eps_scale: float = 0.01):
        self.seed = int(seed)
        self.thr16 = int(np.rint(np.float32(p) * np.float32(65536.0)))
        self.scale = np.float32(1.0) / (np.float32(1.0) - np.float32(self.thr16) / np.float32(65536.0))
        self.dag_offset = int(dag_offset)
        self.eps_scale = np.float32(eps_scale)

    def tile(self, site: int, B: int, N: int, width: int = 64):
        """[B, N, width] mask*scale for a [16][64]-indexed site: element = tok*64 + feature."""
        dag = np.arange(B, dtype=np.uint64) + np.uint64(self.dag_offset)
        key = site_key(self.seed, site, dag)[:, None, None]
        tok = np.arange(N, dtype=np.uint64)[None, :, None]
        f = np.arange(width, dtype=np.uint64)[None, None, :]
        keep = keep_elements(key, tok * np.uint64(64) + f, self.thr16)
        return keep.astype(np.float32) * self.scale

    def attn(self, site: int, B: int, N: int, heads: int = 8):
        """[B*heads, N, N] mask*scale for attention probabilities: element = (h*T + i)*T + j with T = 16*ceil(N/16)
        token slots per DAG (16 on the one-tile path)."""
        T = np.uint64(16 * ((N + 15) // 16))
        dag = np.arange(B, dtype=np.uint64) + np.uint64(self.dag_offset)
        key = site_key(self.seed, site, dag)[:, None, None, None]
        h = np.arange(heads, dtype=np.uint64)[None, :, None, None]
        i = np.arange(N, dtype=np.uint64)[None, None, :, None]
        j = np.arange(N, dtype=np.uint64)[None, None, None, :]
        keep = keep_elements(key, (h * T + i) * T + j, self.thr16)
        return (keep.astype(np.float32) * self.scale).reshape(B * heads, N, N)

    def eps(self, B: int, latent: int = 32):
        """[B, latent] reparameterisation noise (Box-Muller on two 24-bit uniforms), times eps_scale; site 100."""
        dag = np.arange(B, dtype=np.uint64) + np.uint64(self.dag_offset)
        key = site_key(self.seed, 100, dag)[:, None]
        o = np.arange(latent, dtype=np.uint64)[None, :]
        h1 = draw(key, np.uint64(2) * o)
        h2 = draw(key, np.uint64(2) * o + np.uint64(1))
        u1 = ((h1 >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        u2 = (h2 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        n = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
        return (n * self.eps_scale).astype(np.float32)
