"""Oracle: Philox4x32-10 + Box-Muller, numpy (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates the published Philox4x32-10 generator (Salmon et al., SC'11: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl
constants 0x9E3779B9 / 0xBB67AE85, 10 rounds) with the counter / key layout of ``natinf_randn_philox_f32``
(include/natinf.h).  The reference has no counterpart (it uses one sequential torch.randn stream,
src/CIFAR10NaturalInference.py:285-290); this oracle pins the integer stream exactly and the float transform to a
few ulp.  Known-answer test: the Random123 distribution's kat vector for philox4x32-10 (all-zero counter and key)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c, k):
    """c: uint32 [..., 4], k: uint32 [..., 2] -> uint32 [..., 4]"""
    c0, c1, c2, c3 = (c[..., i].astype(np.uint64) for i in range(4))
    k0, k1 = (k[..., i].astype(np.uint64) for i in range(2))
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def randn(indices, elems_per_image, seed):
    """float32 [len(indices), elems_per_image] as the kernel lays it out."""
    idx = np.asarray(indices, dtype=np.uint64)
    q = np.arange(elems_per_image // 4, dtype=np.uint64)
    c = np.zeros((len(idx), len(q), 4), dtype=np.uint32)
    c[..., 0] = (idx & np.uint64(0xFFFFFFFF))[:, None]
    c[..., 1] = (idx >> np.uint64(32))[:, None]
    c[..., 2] = (q & np.uint64(0xFFFFFFFF))[None, :]
    c[..., 3] = (q >> np.uint64(32))[None, :]
    k = np.zeros(c.shape[:-1] + (2,), dtype=np.uint32)
    k[..., 0] = np.uint32(seed & 0xFFFFFFFF); k[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(c, k)
    u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    out = np.empty(c.shape[:-1] + (4,), dtype=np.float32)
    for h in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[..., 2 * h]))
        th = np.float32(6.28318530717958647692) * u[..., 2 * h + 1]
        out[..., 2 * h] = rad * np.cos(th)
        out[..., 2 * h + 1] = rad * np.sin(th)
    return out.reshape(len(idx), elems_per_image), r
