"""CPU oracle (TEST INFRASTRUCTURE, not product code) -- Python big-int restatement.

Plain Python big-integer BN254 G1 arithmetic plus mirrors of the reference's
per-stage CPU test oracles.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path (the HIP library under
metal-msm-gpu-acceleration_amd/csrc) never does.

PARITY PINNING: the reference (ElusAegis/metal-msm-gpu-acceleration, crate mopro-msm)
holds NO golden vectors or known-answer tests for this path -- every test compares the
GPU against a third-party CPU library (halo2curves 0.7.0 `msm_best`, ark-ec 0.4.1
`VariableBaseMSM::msm`; Cargo.toml:42-48) on random inputs, and those crates are not
vendored (no Cargo.lock, `.gitignore:14`).  At the literal-fixture level this oracle is
therefore "parity unpinned".  What pins it instead:
  * the MSM value sum(k_i * P_i) is a unique group element and its canonical affine
    (x, y) mod p is representation independent -- the reference's own parity criterion
    (`src/metal/msm.rs:604-608`: `to_affine()` equality);
  * the constants N, R^2, R-N, MU hard-coded in `src/metal/shader/fields/fp_bn254.h.metal:25-46`
    are re-derived below and asserted at import;
  * curve/group laws checked in tests (generator on curve, r*G = O, known 2G, dlog identity
    sum(k_i*(a+i*d))*G == MSM(k, (a+i*d)*G)) which do not depend on any MSM code;
  * the reference's only literal fixtures -- the 17 bucket index lists of
    `src/metal/msm/bucket_wise_accumulation.rs:232-487` and the pair list at
    `src/metal/msm/sort_buckets.rs:98` -- are reproduced as data in tests/golden/.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------- constants
# BN254 base field / scalar field (SURVEY.md "Quick facts"; fp_bn254.h.metal:25-46 holds N
# as 8 big-endian-order u32 limbs).
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT_R = 1 << 256                      # both halo2curves and arkworks use R = 2^256
MONT_R_INV_P = pow(MONT_R, -1, P)
MONT_R_INV_R = pow(MONT_R, -1, R_ORDER)
CURVE_B = 3                            # y^2 = x^3 + 3, a = 0
GEN = (1, 2)
MODULUS_BIT_SIZE = 254                 # limbs_conversion.rs:172 (ark), :344 (h2c)

# constants of fp_bn254.h.metal:25-46, re-derived (limb 0 = most significant there)
REF_N_LIMBS_BE = [0x30644E72, 0xE131A029, 0xB85045B6, 0x8181585D, 0x97816A91, 0x6871CA8D, 0x3C208C16, 0xD87CFD47]
REF_MU = 3834012553                    # -N^-1 mod 2^32


def _limbs_be32(x: int):
    return [(x >> (32 * (7 - i))) & 0xFFFFFFFF for i in range(8)]


assert _limbs_be32(P) == REF_N_LIMBS_BE
assert (-pow(P, -1, 1 << 32)) % (1 << 32) == REF_MU
assert P % 4 == 3 and P.bit_length() == 254 and R_ORDER.bit_length() == 254


# ----------------------------------------------------------------------------- field helpers
def fq_to_mont(x: int) -> int:
    return (x * MONT_R) % P


def fq_from_mont(x: int) -> int:
    return (x * MONT_R_INV_P) % P


def fr_to_mont(x: int) -> int:
    return (x * MONT_R) % R_ORDER


def fr_from_mont(x: int) -> int:
    return (x * MONT_R_INV_R) % R_ORDER


def mont_mul_p(a: int, b: int) -> int:
    """Montgomery product a*b*R^-1 mod p (fp_bn254.h.metal:237-290 computes this by CIOS)."""
    return (a * b * MONT_R_INV_P) % P


# ----------------------------------------------------------------------------- curve (Jacobian, plain ints)
# A point is None (identity) or a tuple (X, Y, Z) of canonical (non-Montgomery) ints.
def is_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % P == 0


def jac_double(p1):
    """dbl-2009-l (a = 0). The reference uses dbl-2007-bl (ec_point.h.metal:141-175);
    both give the same group element."""
    if p1 is None:
        return None
    X1, Y1, Z1 = p1
    if Y1 == 0:
        return None
    A = X1 * X1 % P
    B = Y1 * Y1 % P
    C = B * B % P
    D = 2 * ((X1 + B) * (X1 + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y1 * Z1 % P
    return (X3, Y3, Z3)


def jac_add(p1, p2):
    """add-2007-bl with the equality->double fallback, as ec_point.h.metal:13-69."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(p1)
        return None
    H = (U2 - U1) % P
    I = (2 * H) * (2 * H) % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


def to_jac(aff):
    return None if aff is None else (aff[0], aff[1], 1)


def to_affine(pj):
    """Canonical affine (x, y) or None -- the parity representation (msm.rs:604-608)."""
    if pj is None:
        return None
    X, Y, Z = pj
    if Z % P == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def aff_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def aff_add(a, b):
    return to_affine(jac_add(to_jac(a), to_jac(b)))


def scalar_mul(k: int, aff):
    """Double-and-add, MSB first (the reference's operate_with_self is LSB-first,
    ec_point.h.metal:110-131; same group element)."""
    k %= R_ORDER
    acc = None
    base = to_jac(aff)
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, base)
    return to_affine(acc)


def scalar_mul_jac(k: int, pj):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, pj)
    return acc


# ----------------------------------------------------------------------------- MSM
def msm_naive(scalars, points):
    """sum(k_i * P_i) by repeated double-and-add: the definition, no bucket method."""
    acc = None
    for k, pt in zip(scalars, points):
        acc = jac_add(acc, to_jac(scalar_mul(k, pt)))
    return to_affine(acc)


def window_params(n: int, window_size=None):
    """encode_instances (msm.rs:132-146): c = 3 if n < 32 else 15; starts 0,c,2c,.. < 254."""
    c = window_size if window_size is not None else (3 if n < 32 else 15)
    starts = list(range(0, MODULUS_BIT_SIZE, c))
    return c, starts, len(starts), (1 << c) - 1


def get_scalar_fragment(k: int, window_start: int) -> int:
    """get_scalar_fragment (prepare_buckets_indices.rs:59-90): the low 32-bit limb of the
    256-bit value k >> window_start."""
    return (k >> window_start) & 0xFFFFFFFF


def prepare_buckets_indices(scalars, window_size: int, num_windows: int):
    """prepare_buckets_indices_rust (prepare_buckets_indices.rs:92-118) in the kernel's output
    order (msm.h.metal:50-57): entry [t*W + i] = (i*(2^c-1) + m - 1, t) or the sentinel pair."""
    buckets_len = (1 << window_size) - 1
    out = []
    for t, k in enumerate(scalars):
        for i in range(num_windows):
            m = get_scalar_fragment(k, i * window_size) & buckets_len
            if m != 0:
                out.append((i * buckets_len + m - 1, t))
            else:
                out.append((0xFFFFFFFF, 0xFFFFFFFF))
    return out


def sort_buckets_indices(pairs):
    """sort_buckets_indices (sort_buckets.rs:15-34): stable sort by .0 (tests only require the
    multiset to be preserved and keys non-decreasing, sort_buckets.rs:111-125)."""
    return sorted(pairs, key=lambda pr: pr[0])


def bucket_wise_accumulation(pairs, points_jac, total_buckets=None):
    """bucket_wise_accumulation_rust (bucket_wise_accumulation.rs:662-681).  points are Jacobian
    tuples / None; returns a list of Jacobian points indexed by bucket."""
    if total_buckets is None:
        total_buckets = max([a for a, _ in pairs if a != 0xFFFFFFFF], default=0) + 1
    res = [None] * total_buckets
    for b, pi in pairs:
        if b == 0xFFFFFFFF:
            continue
        if b < total_buckets:
            res[b] = jac_add(res[b], points_jac[pi])
    return res


def sum_reduction(num_windows: int, buckets_matrix):
    """sum_reduction_rust (sum_reduction.rs:358-378): res[j] = sum_b (b+1) * B[j*len + b],
    computed here with the running-sum identity instead of scalar multiplications."""
    bl = len(buckets_matrix) // num_windows
    out = []
    for j in range(num_windows):
        run = None
        acc = None
        for b in range(bl - 1, -1, -1):
            run = jac_add(run, buckets_matrix[j * bl + b])
            acc = jac_add(acc, run)
        out.append(acc)
    return out


def final_accumulation(window_sums, window_size: int):
    """Horner over the window sums, highest window first (final_accumulation.rs:19-39); unlike
    the reference this also handles window_num == 1."""
    acc = None
    for ws in reversed(window_sums):
        for _ in range(window_size):
            acc = jac_double(acc)
        acc = jac_add(acc, ws)
    return acc


def msm_pippenger(scalars, points, window_size=None):
    """The reference pipeline (msm.rs:189-217) restated on the CPU: digits -> sort ->
    bucket accumulate -> weighted window sums -> Horner."""
    n = min(len(scalars), len(points))
    if n == 0:
        return None
    c, _starts, W, _bl = window_params(n, window_size)
    pairs = sort_buckets_indices(prepare_buckets_indices(scalars[:n], c, W))
    pj = [to_jac(pt) for pt in points[:n]]
    buckets = bucket_wise_accumulation(pairs, pj, W * ((1 << c) - 1))
    return to_affine(final_accumulation(sum_reduction(W, buckets), c))


# ----------------------------------------------------------------------------- layouts
def int_to_be32_limbs(x: int):
    """Reference device limb order: 8 x u32, limb 0 most significant (unsigned_int.h.metal:12-17,
    limbs_conversion.rs:87-106)."""
    return _limbs_be32(x)


def be32_limbs_to_int(limbs) -> int:
    x = 0
    for l in limbs:
        x = (x << 32) | (int(l) & 0xFFFFFFFF)
    return x


def int_to_le_bytes32(x: int) -> bytes:
    """[u64;4] little-endian host representation of halo2curves / arkworks field elements."""
    return int(x).to_bytes(32, "little")


def encode_scalar_h2c(k: int) -> bytes:
    """bn256::Fr in memory: Montgomery form, [u64;4] LE (msm.rs:258-270 reinterprets these)."""
    return int_to_le_bytes32(fr_to_mont(k % R_ORDER))


def encode_affine_h2c(pt) -> bytes:
    """bn256::G1Affine {x, y}: two Fq in Montgomery form, LE; identity is (0, 0)."""
    if pt is None:
        return bytes(64)
    return int_to_le_bytes32(fq_to_mont(pt[0])) + int_to_le_bytes32(fq_to_mont(pt[1]))


def encode_projective_ark(pj) -> bytes:
    """ark_bn254::G1Projective {x, y, z}: Jacobian, Montgomery, LE (limbs_conversion.rs:123-130).
    Identity is written as (1, 1, 0) in Montgomery form like arkworks' `zero()`."""
    if pj is None:
        pj = (1, 1, 0)
    return b"".join(int_to_le_bytes32(fq_to_mont(c)) for c in pj)


def decode_jacobian_mont_le(buf: bytes):
    """96-byte result buffer (x, y, z Montgomery LE) -> canonical affine tuple or None."""
    assert len(buf) == 96
    X, Y, Z = (fq_from_mont(int.from_bytes(buf[32 * i:32 * i + 32], "little")) for i in range(3))
    return to_affine((X, Y, Z)) if Z != 0 else None


def encode_point_be32(pj):
    """Reference wire layout of a point: 24 x u32 (x, y, z), each MS-limb first, Montgomery
    (limbs_conversion.rs:123-130, :313-327)."""
    if pj is None:
        pj = (1, 1, 0)
    out = []
    for c in pj:
        out += int_to_be32_limbs(fq_to_mont(c))
    return out


def decode_point_be32(limbs):
    X, Y, Z = (fq_from_mont(be32_limbs_to_int(limbs[8 * i:8 * i + 8])) for i in range(3))
    return None if Z == 0 else (X, Y, Z)


def encode_scalar_be32(k: int):
    """Reference wire layout of a scalar: canonical value, MS-limb first (limbs_conversion.rs:116-121)."""
    return int_to_be32_limbs(k % R_ORDER)


# ----------------------------------------------------------------------------- deterministic synthetic inputs
# Counter-based generator shared (bit for bit) by this file, oracle/msm_oracle.c and the device
# generator kernel of the product library (csrc/msm_kernels.hip: gen_instance_kernel).  It plays the
# role of the reference's random instance generation (src/utils/preprocess.rs:113-138).
M64 = (1 << 64) - 1
STREAM_BASES = 0
STREAM_SCALARS = 1
SEED_BASE = 0xB2540000


def splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rnd64(seed: int, stream: int, ctr: int) -> int:
    return splitmix64((splitmix64(seed ^ (stream << 56)) + ctr) & M64)


def _rnd256(seed, stream, ctr4):
    w = [rnd64(seed, stream, ctr4 * 4 + k) for k in range(4)]
    return w[0] | (w[1] << 64) | (w[2] << 128) | (w[3] << 192)


def gen_scalar(seed: int, i: int) -> int:
    """Uniform scalar mod r by rejection: 254 random bits per attempt (accept if < r, p = 0.756), up to 16
    attempts; the 2^-32-probability fallback subtracts r once."""
    v = 0
    for attempt in range(16):
        v = _rnd256(seed, STREAM_SCALARS, i * 16 + attempt) & ((1 << 254) - 1)
        if v < R_ORDER:
            return v
    return v - R_ORDER


def gen_point(seed: int, i: int):
    """Try-and-increment: candidate x from 254 random bits, y = (x^3+3)^((p+1)/4), sign from the
    top random bit.  BN254 G1 has cofactor 1 so every curve point is in the group."""
    for attempt in range(64):
        raw = _rnd256(seed, STREAM_BASES, i * 64 + attempt)
        x = raw & ((1 << 254) - 1)
        if x >= P:
            continue
        rhs = (x * x * x + CURVE_B) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P != rhs:
            continue
        if (raw >> 255) & 1:
            y = (P - y) % P
        return (x, y)
    raise RuntimeError("gen_point: 64 failed attempts")


def gen_instance(seed: int, n: int):
    return [gen_point(seed, i) for i in range(n)], [gen_scalar(seed, i) for i in range(n)]
