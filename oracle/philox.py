"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the engine's in-kernel noise stream
(pytorch_mppi_amd/csrc/common.hpp: philox4x32_10, box_muller, philox_normal4).  This is NOT
reference behaviour (the reference draws torch.randn, mppi.py:203); it is the CPU checker of
the engine-defined counter scheme:
    counter = (k_global, jb, call_lo, call_hi ^ k_global_hi), key = (seed_lo, seed_hi)
    -> 4 words -> two Box-Muller pairs -> the row-of-4 at [jb][k] of the TNK4 layout.
Philox4x32-10 itself is pinned by the Random123 known-answer vectors."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-R (R = 10: the default everywhere; R = 7: rng="philox7", Random123's philox4x32_R<7>)"""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def box_muller(a, b):
    u1 = (a.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float32)
    u2 = (b.astype(np.float32) * np.float32(2.0 ** -32)).astype(np.float32)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float64(2.0 * np.pi) * u2.astype(np.float64))
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)


def rows4(T, nu):
    g = 4 if nu % 4 == 0 else (2 if nu % 2 == 0 else 1)
    p4, tt = nu // g, 4 // g
    return -(-T // tt) * p4


def normals_tnk4(seed, call, K, T, nu, k_offset=0, rounds=10):
    """(J4, K, 4) float32 -- the engine's native layout."""
    J4 = rows4(T, nu)
    kg = (np.arange(K, dtype=np.uint64) + np.uint64(k_offset))[None, :].repeat(J4, 0)
    jb = np.arange(J4, dtype=np.uint32)[:, None].repeat(K, 1)
    c0 = kg.astype(np.uint32)
    c2 = np.full_like(c0, np.uint32(call & 0xFFFFFFFF))
    c3 = np.uint32((call >> 32) & 0xFFFFFFFF) ^ (kg >> np.uint64(32)).astype(np.uint32)
    r0, r1, r2, r3 = philox4x32_10(c0, jb, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds)
    a, b = box_muller(r0, r1)
    c, d = box_muller(r2, r3)
    return np.stack([a, b, c, d], axis=-1)


def normals_ktn(seed, call, K, T, nu, k_offset=0, rounds=10):
    """The same draws re-indexed to the reference's (K,T,nu) layout: element (k,t,n) is
    component (t*nu+n)%4 of row (t*nu+n)//4."""
    z4 = normals_tnk4(seed, call, K, T, nu, k_offset, rounds)          # (J4,K,4)
    flat = z4.transpose(1, 0, 2).reshape(K, -1)                # (K, J4*4)
    return np.ascontiguousarray(flat[:, :T * nu].reshape(K, T, nu))


PROCESS_NOISE_KEY_TAG = 0x5A5A5A5AA5A5A5A5


def process_normals(seed, call, K, T, M, nx, MM=4, k_offset=0):
    """(M, K, T, nx) float32 -- the process-noise draws of the fused multi-rollout kernel
    (pytorch_mppi_amd/csrc/rollout.hpp, rollout_stream_multi): Philox key = seed ^ PROCESS_NOISE_KEY_TAG,
    counter = (k_global, (t * MM + m) * ceil(nx/4) + block, call), four normals per call."""
    nxb = -(-nx // 4)
    key = (seed ^ PROCESS_NOISE_KEY_TAG) & 0xFFFFFFFFFFFFFFFF
    out = np.zeros((M, K, T, 4 * nxb), dtype=np.float32)
    kg = np.arange(K, dtype=np.uint64) + np.uint64(k_offset)
    c0 = kg.astype(np.uint32)
    c2 = np.full(K, call & 0xFFFFFFFF, dtype=np.uint32)
    c3 = np.uint32((call >> 32) & 0xFFFFFFFF) ^ (kg >> np.uint64(32)).astype(np.uint32)
    for m in range(M):
        for t in range(T):
            for q in range(nxb):
                jb = np.full(K, (t * MM + m) * nxb + q, dtype=np.uint32)
                r0, r1, r2, r3 = philox4x32_10(c0, jb, c2, c3, key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
                a, b = box_muller(r0, r1)
                c, d = box_muller(r2, r3)
                out[m, :, t, 4 * q:4 * q + 4] = np.stack([a, b, c, d], axis=-1)
    return out[..., :nx]
