"""Third-party cross-check of the oracle's group law (CPU, build container only).

sympy.ntheory.elliptic_curve.EllipticCurve(0, 3, modulus=p) is an independent implementation of the short-Weierstrass
group law that happens to be importable in the build image.  It is NOT the reference (the reference's oracle is
halo2curves / arkworks, absent here), so parity stays "unpinned" by the rules of this project; what this buys is that
oracle/bn254_ref.py is no longer checked only against itself, its C twin and two EIP-196 constants.

Writes tests/golden/sympy_crosscheck.json: inputs (seeded) and sympy's answers for
  add      P + Q for random P, Q, plus P + P, P + (-P), O + P, P + O, O + O
  mul      k * P for random 254-bit k and the edge scalars 0, 1, 2, r - 1, r, r + 1
  msm      sum k_i P_i for n = 1 .. 32
tests/test_sympy_crosscheck.py replays the file against oracle/bn254_ref.py (aff_add, scalar_mul, msm_naive,
msm_pippenger) and against the C oracle (oracle_msm_best / oracle_msm_naive), without needing sympy again.

Run:  python tools/gen_sympy_crosscheck.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sympy.ntheory.elliptic_curve import EllipticCurve  # noqa: E402

P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def main():
    E = EllipticCurve(0, 3, modulus=P_MOD)
    G = E(1, 2)
    O = G + (-G)
    rng = random.Random(0x5EED_0003)

    def enc(pt):
        return None if int(pt.z) == 0 else [hex(int(pt.x)), hex(int(pt.y))]   # sympy's identity is (0 : 1 : 0)

    def rand_point():
        return rng.randrange(1, R_ORDER) * G

    add = []
    for _ in range(48):
        a, b = rand_point(), rand_point()
        add.append({"a": enc(a), "b": enc(b), "sum": enc(a + b)})
    for _ in range(8):
        a = rand_point()
        add.append({"a": enc(a), "b": enc(a), "sum": enc(a + a)})          # doubling
        add.append({"a": enc(a), "b": enc(-a), "sum": enc(a + (-a))})      # cancellation
        add.append({"a": None, "b": enc(a), "sum": enc(O + a)})
        add.append({"a": enc(a), "b": None, "sum": enc(a + O)})
    add.append({"a": None, "b": None, "sum": enc(O + O)})

    mul = []
    edge = [0, 1, 2, 3, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 254) - 1, 1 << 253]
    for k in edge + [rng.randrange(R_ORDER) for _ in range(40)]:
        a = rand_point()
        mul.append({"k": hex(k), "p": enc(a), "prod": enc(k * a)})
    mul.append({"k": hex(R_ORDER), "p": enc(G), "prod": enc(R_ORDER * G)})   # r G = O

    msm = []
    for n in [1, 2, 3, 5, 8, 13, 21, 32]:
        pts = [rand_point() for _ in range(n)]
        ks = [rng.randrange(R_ORDER) for _ in range(n)]
        if n >= 5:
            ks[1] = 0                       # a zero scalar
            pts[2] = pts[0]                 # a repeated base
            ks[3] = R_ORDER - ks[0]         # k0 P0 + (r - k0) P2 cancels when P2 = P0 ... with pts[2]: partial cancel
        acc = O
        for k, pt in zip(ks, pts):
            acc = acc + k * pt
        msm.append({"scalars": [hex(k) for k in ks], "points": [enc(q) for q in pts], "sum": enc(acc)})

    out = {"generator": "tools/gen_sympy_crosscheck.py", "library": "sympy.ntheory.elliptic_curve.EllipticCurve(0, 3, modulus=p)",
           "note": "independent third-party group law; not the reference's oracle (parity stays unpinned)",
           "add": add, "mul": mul, "msm": msm}
    path = os.path.join(ROOT, "tests", "golden", "sympy_crosscheck.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(f"wrote {path}: {len(add)} additions, {len(mul)} scalar multiplications, {len(msm)} MSMs")


if __name__ == "__main__":
    main()
