"""Shared test helpers: encoding of oracle values into the C-ABI layouts and back."""
import random

from oracle import bn254_ref as o


def rand_fq(rng):
    return rng.randrange(o.P)


def rand_point(rng):
    """Random G1 point as canonical affine (via a random multiple of the generator)."""
    return o.scalar_mul(rng.randrange(1, o.R_ORDER), o.GEN)


def rand_jac(rng, aff):
    """Re-randomise the Jacobian representative of an affine point."""
    if aff is None:
        return None
    l = rng.randrange(1, o.P)
    return (aff[0] * l * l % o.P, aff[1] * l * l * l % o.P, l)


def fq_be32(x):
    """Montgomery residue of canonical x in the reference wire layout (8 x u32, MS first)."""
    return o.int_to_be32_limbs(o.fq_to_mont(x))


def be32_fq(limbs):
    return o.fq_from_mont(o.be32_limbs_to_int(limbs))


def decode_be32_affine(limbs24):
    pj = o.decode_point_be32(limbs24)
    return o.to_affine(pj) if pj is not None else None


def h2c_instance_bytes(points, scalars):
    """(scalars bytes, points bytes) exactly as &[bn256::Fr] / &[bn256::G1Affine] lie in memory."""
    return b"".join(o.encode_scalar_h2c(k) for k in scalars), b"".join(o.encode_affine_h2c(p) for p in points)


def small_instance(seed, n):
    rng = random.Random(seed)
    pts = [rand_point(rng) for _ in range(n)]
    sc = [rng.randrange(o.R_ORDER) for _ in range(n)]
    return pts, sc


def dlog_expected(a0, d, scalars_mont_le: bytes, n: int):
    """(sum_i k_i (a0 + i d)) G for scalars handed over as Montgomery residues k R mod r (32 B little-endian each):
    the answer of an MSM over the bases P_i = (a0 + i d) G, with big integers only -- no MSM code on this side.
    sum k_i (a0 + i d) = a0 sum k_i + d sum i k_i; both sums are taken over 16-bit quarter limbs with numpy (a quarter
    limb times an index below 2^28 stays below 2^44, 2^28 of them below 2^64)."""
    import numpy as np
    assert n < (1 << 20) * 256
    q = np.frombuffer(scalars_mont_le, dtype="<u2").reshape(n, 16).astype(np.uint64)
    idx = np.arange(n, dtype=np.uint64)
    s0 = [int(x) for x in q.sum(axis=0)]
    s1 = [int(x) for x in (q * idx[:, None]).sum(axis=0)]
    sum_k = sum(v << (16 * j) for j, v in enumerate(s0))
    sum_ik = sum(v << (16 * j) for j, v in enumerate(s1))
    r_inv = pow(o.MONT_R, -1, o.R_ORDER)
    s = (a0 * sum_k + d * sum_ik) * r_inv % o.R_ORDER
    return o.scalar_mul(s, o.GEN)
