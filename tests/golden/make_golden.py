"""Generates the golden fixtures in this directory from the repository's own big-int oracle
(oracle/bn254_ref.py).  The reference holds no known-answer vectors for this path,
so these pin the build's oracle against regressions and give the GPU tests literal expected values;
the reference's only literal fixtures (bucket index lists, sort list) are reproduced in
reference_index_lists.json as data.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bn254_ref as o  # noqa: E402


def main():
    rng = random.Random(20250117)
    # ---- field op vectors (canonical values and Montgomery residues, hex)
    edge = [0, 1, 2, o.P - 1, o.P - 2, o.MONT_R % o.P, (o.MONT_R * o.MONT_R) % o.P]
    vals = edge + [rng.randrange(o.P) for _ in range(57)]
    mul = []
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        mul.append({"a": hex(a), "b": hex(b), "ab": hex(a * b % o.P), "sum": hex((a + b) % o.P),
                    "diff": hex((a - b) % o.P), "neg": hex((-a) % o.P),
                    "mont_a": hex(o.fq_to_mont(a)), "mont_b": hex(o.fq_to_mont(b)),
                    "mont_ab": hex(o.mont_mul_p(o.fq_to_mont(a), o.fq_to_mont(b)))})
    with open(os.path.join(HERE, "field_ops.json"), "w") as f:
        json.dump({"p": hex(o.P), "r": hex(o.R_ORDER), "mont_R": "2^256", "mul": mul}, f, indent=1)

    # ---- point op vectors: generic, P+P, P+(-P), inf+P, P+inf, inf+inf (cases of test_bn254.rs:373-458)
    def aff(pt):
        return None if pt is None else [hex(pt[0]), hex(pt[1])]
    pts = [o.scalar_mul(rng.randrange(1, o.R_ORDER), o.GEN) for _ in range(20)]
    cases = [(pts[i], pts[i + 1]) for i in range(0, 16, 2)]
    cases += [(pts[16], pts[16]), (pts[17], o.aff_neg(pts[17])), (None, pts[18]), (pts[19], None), (None, None)]
    padd = [{"p": aff(p), "q": aff(q), "sum": aff(o.aff_add(p, q)), "dbl_p": aff(o.aff_add(p, p))} for p, q in cases]
    ks = [0, 1, 2, o.R_ORDER - 1, o.R_ORDER, (1 << 14) + 1] + [rng.randrange(o.R_ORDER) for _ in range(10)]
    pmul = [{"k": hex(k), "p": aff(pts[i % len(pts)]), "kp": aff(o.scalar_mul(k, pts[i % len(pts)]))}
            for i, k in enumerate(ks)]
    with open(os.path.join(HERE, "point_ops.json"), "w") as f:
        json.dump({"generator": aff(o.GEN), "two_g": aff(o.scalar_mul(2, o.GEN)), "add": padd, "mul": pmul}, f, indent=1)

    # ---- digit extraction vectors (prepare_buckets_indices semantics) for c in {3, 14, 15, 16}
    dig = []
    for c in (3, 14, 15, 16):
        sc = [(1 << 14) + 1, 0, 1, o.R_ORDER - 1] + [rng.randrange(o.R_ORDER) for _ in range(4)]
        W = len(range(0, o.MODULUS_BIT_SIZE, c))
        dig.append({"window_size": c, "num_windows": W, "scalars": [hex(k) for k in sc],
                    "pairs": o.prepare_buckets_indices(sc, c, W)})
    with open(os.path.join(HERE, "digits.json"), "w") as f:
        json.dump({"cases": dig}, f)

    # ---- whole-MSM answers on generator-defined instances (seed, n) -> canonical affine result
    msm = []
    for seed, n in [(o.SEED_BASE, 1), (o.SEED_BASE, 2), (o.SEED_BASE + 1, 31), (o.SEED_BASE + 2, 32),
                    (o.SEED_BASE + 3, 256), (o.SEED_BASE + 4, 1024)]:
        p, s = o.gen_instance(seed, n)
        res = o.msm_naive(s, p)
        msm.append({"seed": seed, "n": n, "result_affine": [hex(res[0]), hex(res[1])]})
    with open(os.path.join(HERE, "msm_small.json"), "w") as f:
        json.dump({"generator": "oracle/bn254_ref.py gen_instance(seed, n); scalars/points in h2c layout",
                   "cases": msm}, f, indent=1)


if __name__ == "__main__":
    main()
