"""Worst-case limb-bound verifier for the lazily reduced 29-bit-limb arithmetic (csrc/bn254_fq29.hip.h,
bn254_ec29.hip.h).

The hot kernels never propagate carries between field operations: limbs are only BOUNDED, and a 64-bit column sum
of a multiplication that overflowed would be silent.  This script re-states the group-law formulas over intervals
(per-limb maxima, value bound in multiples of p) and checks, for every multiplication, squaring and double
product, that no column of products + Montgomery terms + carries can reach 2^64, that no lifted subtraction can
go negative in any limb, that no limb leaves 32 bits, that the one-limb zero filters are given a large enough
bound, and that every point an addition returns satisfies the invariant the next addition assumes
(X < 10 p, Y < 6 p, ZZ < 2.8 p, ZZZ < 2 p, limbs 0..7 < 2^29 + 8).  Run: python tools/fq29_bounds.py  (exit 0 = all hold).
The constants are read from the header, the formulas below mirror bn254_ec29.hip.h line by line -- when one
changes, change the other."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "csrc", "bn254_fq29.hip.h")
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
MASK = (1 << 29) - 1
RHO = 1 << 261
src = open(HDR).read()


def _table(name, rows=1):
    m = re.search(name + r"\(int[^)]*\)\s*\{\s*constexpr uint32_t c(?:\[\d+\])?\[9\] = \{(.*?)\};", src, re.S)
    nums = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]+)u", m.group(1))]
    return [nums[9 * r:9 * r + 9] for r in range(len(nums) // 9)]


PL = _table("p")[0]
assert sum(l << (29 * i) for i, l in enumerate(PL)) == P
KC = _table("kc")
KNAMES = ["K4E30", "K8E30", "K8E31", "K16E30", "K16E31"]
KMULT = {}
for name, limbs in zip(KNAMES, KC):
    v = sum(l << (29 * i) for i, l in enumerate(limbs))
    assert v % P == 0
    KMULT[name] = v // P
KL = dict(zip(KNAMES, KC))
problems = []


def fail(msg):
    problems.append(msg)


class Fe:
    """limb maxima (limbs are unsigned, minima are 0) + value bound in multiples of p"""

    def __init__(self, mx, val, name="?"):
        self.mx, self.val, self.name = list(mx), float(val), name

    def __repr__(self):
        return f"{self.name}: val<{self.val:.2f}p, limbs<=2^{max(self.mx[:8]).bit_length()} top<={self.mx[8]}"


def top_from_value(val):
    return int(val * P) >> 232


def canonical(name):          # an unpacked base coordinate: canonical limbs of a value < p
    return Fe([MASK] * 8 + [P >> 232], 1.0, name)


def _reduce(cols, what):
    """Montgomery reduction of worst-case column sums; returns the worst-case top limb (final carry)."""
    A = list(cols) + [0] * (17 - len(cols))
    carry = 0
    for k in range(9):
        A[k] += carry
        for j in range(9):
            A[k + j] += MASK * PL[j]
        if A[k] >= 1 << 64:
            fail(f"{what}: column {k} can reach 2^{A[k].bit_length()} during the reduction")
        carry = A[k] >> 29
    for k in range(9, 17):
        A[k] += carry
        if A[k] >= 1 << 64:
            fail(f"{what}: column {k} can reach 2^{A[k].bit_length()}")
        carry = A[k] >> 29
    return carry


def _mul_cols(pairs):
    cols = [0] * 17
    for a, b in pairs:
        for i in range(9):
            for j in range(9):
                cols[i + j] += a.mx[i] * b.mx[j]
    return cols


def _out(pairs, name, what):
    for a, b in pairs:
        for f in (a, b):
            if max(f.mx) >= 1 << 32:
                fail(f"{what}: operand {f.name} has a limb beyond 32 bits")
    carry = _reduce(_mul_cols(pairs), what)
    val = sum(a.val * b.val for a, b in pairs) * P / RHO + 1.0
    top = min(carry, top_from_value(val))
    return Fe([MASK] * 8 + [top], val, name)


def mul(a, b, name):
    return _out([(a, b)], name, f"mul {name} = {a.name} * {b.name}")


def mul2(a, b, c, d, name):
    return _out([(a, b), (c, d)], name, f"mul2 {name} = {a.name}*{b.name} + {c.name}*{d.name}")


def sqr(a, name):
    if max(a.mx) * 2 >= 1 << 32:
        fail(f"sqr {name}: doubled limb of {a.name} leaves 32 bits")
    return _out([(a, a)], name, f"sqr {name} = {a.name}^2")


def add(a, b, name=None):
    r = Fe([x + y for x, y in zip(a.mx, b.mx)], a.val + b.val, name or f"({a.name}+{b.name})")
    if max(r.mx) >= 1 << 32:
        fail(f"add {r.name}: limb beyond 32 bits")
    return r


def sub(sel, a, b, name):
    K = KL[sel]
    for i in range(9):
        if b.mx[i] > K[i]:
            fail(f"sub<{sel}> {name} = {a.name} - {b.name}: limb {i} of the subtrahend may exceed the lift "
                 f"({b.mx[i]:#x} > {K[i]:#x})")
    if b.val > KMULT[sel]:
        fail(f"sub<{sel}> {name}: subtrahend value {b.val:.2f}p above the {KMULT[sel]}p of the constant")
    r = Fe([x + k for x, k in zip(a.mx, K)], a.val + KMULT[sel], name)
    if max(r.mx) >= 1 << 32:
        fail(f"sub<{sel}> {name}: limb beyond 32 bits")
    return r


def norm(a, name=None):
    mx = [min(a.mx[0], MASK)]
    for i in range(1, 8):
        mx.append(min(a.mx[i], MASK) + (a.mx[i - 1] >> 29))
    mx.append(a.mx[8] + (a.mx[7] >> 29))
    top = min(mx[8], top_from_value(a.val) + 1)
    mx[8] = top
    return Fe(mx, a.val, name or a.name)


def zero():
    return Fe([0] * 9, 0.0, "0")


def neg(a, name):
    return norm(sub("K4E30", zero(), a, name), name)


def maybe_zero(a, bound):
    if a.val >= bound:
        fail(f"maybe_zero({a.name}, {bound}): value bound {a.val:.2f}p is not below the filter's {bound}")


# the invariant of every stored / loop-carried point (multiples of p); pti_mmadd sets it: ZZ3 = P^2 with P < 17.1 p
# gives 2.72 p, X3 = R^2 - ... + 8 p with R < 12.1 p gives 9.86 p
INV_X, INV_Y, INV_ZZ, INV_ZZZ = 10.0, 6.0, 2.8, 2.0


def check_point(x, y, zz, zzz, where):
    for f, lim in ((x, INV_X), (y, INV_Y), (zz, INV_ZZ), (zzz, INV_ZZZ)):
        if f.val > lim:
            fail(f"{where}: {f.name} may reach {f.val:.2f}p, the invariant says < {lim}p")
        if max(f.mx[:8]) > MASK + 8:
            fail(f"{where}: {f.name} leaves limbs above 2^29 + 8")


def point_invariant():
    def coord(name, val):
        return Fe([MASK + 8] * 8 + [top_from_value(val) + 1], val, name)
    return coord("X1", INV_X), coord("Y1", INV_Y), coord("ZZ1", INV_ZZ), coord("ZZZ1", INV_ZZZ)


# ---- the formulas of bn254_ec29.hip.h ----------------------------------------------------------------------
def pti_double(px, py, pzz, pzzz, where):
    U = add(py, py, "U")
    V = sqr(U, "V")
    W = mul(U, V, "W")
    S = mul(px, V, "S")
    XX = sqr(px, "XX")
    M = norm(add(XX, add(XX, XX)), "M")
    MM = sqr(M, "MM")
    X3 = norm(sub("K4E30", MM, add(S, S), "X3"), "X3")
    T = norm(sub("K8E30", S, X3, "T"), "T")
    Y3 = norm(sub("K4E30", mul(M, T, "MT"), mul(W, py, "WY"), "Y3"), "Y3")
    ZZ3 = mul(V, pzz, "ZZ3")
    ZZZ3 = mul(W, pzzz, "ZZZ3")
    check_point(X3, Y3, ZZ3, ZZZ3, where + " (doubling)")


def neg_wide(a, name):
    return sub("K4E30", zero(), a, name)


def negated_base_y(qy, name="+-y2"):
    """accumulate_kernel: cur.y = negate ? Fq29::neg_wide(cur.y) : cur.y  (no carry round)"""
    n = neg_wide(qy, "-y")
    return Fe([max(a, b) for a, b in zip(n.mx, qy.mx)], max(n.val, qy.val), name)


def stored_base_y(qy, name="+-y1"):
    """pti_from_affi: the y of a point that becomes an accumulator is normalised"""
    return norm(negated_base_y(qy, name), name)


def pti_madd(where="pti_madd"):
    px, py, pzz, pzzz = point_invariant()
    qx, qy = canonical("x2"), negated_base_y(canonical("y2"))
    U2 = mul(qx, pzz, "U2")
    S2 = mul(qy, pzzz, "S2")
    Pd = norm(sub("K16E30", U2, px, "P"), "P")
    R = norm(sub("K8E30", S2, py, "R"), "R")
    maybe_zero(Pd, 18)
    PP = sqr(Pd, "PP")
    PPP = mul(Pd, PP, "PPP")
    Q = mul(px, PP, "Q")
    RR = sqr(R, "RR")
    X3 = norm(sub("K8E31", RR, add(PPP, add(Q, Q)), "X3"), "X3")
    T = sub("K16E30", Q, X3, "T")                            # un-normalised: its partner R is normalised
    Y3 = mul2(R, T, py, neg_wide(PPP, "-PPP"), "Y3")
    ZZ3 = mul(pzz, PP, "ZZ3")
    ZZZ3 = mul(pzzz, PPP, "ZZZ3")
    check_point(X3, Y3, ZZ3, ZZZ3, where)
    one = Fe([MASK] * 8 + [P >> 232], 1.0, "one")           # the doubling path restarts from pti_from_affi(q)
    pti_double(qx, stored_base_y(canonical("y2")), one, one, where)


def pti_mmadd(where="pti_mmadd"):
    # both operands affine; p = a previous base (its y possibly a lazily negated value < 4 p), q likewise
    px, py = canonical("x1"), stored_base_y(canonical("y1"))
    qx, qy = canonical("x2"), negated_base_y(canonical("y2"))
    Pd = norm(sub("K16E30", qx, px, "P"), "P")
    R = norm(sub("K8E30", qy, py, "R"), "R")
    maybe_zero(Pd, 18)
    PP = sqr(Pd, "PP")
    PPP = mul(Pd, PP, "PPP")
    Q = mul(px, PP, "Q")
    RR = sqr(R, "RR")
    X3 = norm(sub("K8E31", RR, add(PPP, add(Q, Q)), "X3"), "X3")
    T = norm(sub("K16E30", Q, X3, "T"), "T")
    Y3 = mul2(R, T, py, neg_wide(PPP, "-PPP"), "Y3")
    check_point(X3, Y3, PP, PPP, where)
    one = Fe([MASK] * 8 + [P >> 232], 1.0, "one")
    pti_double(qx, stored_base_y(canonical("y2")), one, one, where)


def pti_add_nz(where="pti_add_nz"):
    px, py, pzz, pzzz = point_invariant()
    qx, qy, qzz, qzzz = point_invariant()
    for f, n in ((qx, "X2"), (qy, "Y2"), (qzz, "ZZ2"), (qzzz, "ZZZ2")):
        f.name = n
    U1 = mul(px, qzz, "U1")
    U2 = mul(qx, pzz, "U2")
    S1 = mul(py, qzzz, "S1")
    S2 = mul(qy, pzzz, "S2")
    Pd = norm(sub("K4E30", U2, U1, "P"), "P")
    R = norm(sub("K4E30", S2, S1, "R"), "R")
    maybe_zero(Pd, 6)
    PP = sqr(Pd, "PP")
    PPP = mul(Pd, PP, "PPP")
    Q = mul(U1, PP, "Q")
    RR = sqr(R, "RR")
    X3 = norm(sub("K8E31", RR, add(PPP, add(Q, Q)), "X3"), "X3")
    T = sub("K16E30", Q, X3, "T")
    Y3 = mul2(R, T, S1, neg_wide(PPP, "-PPP"), "Y3")
    ZZ3 = mul(mul(pzz, qzz, "ZZ12"), PP, "ZZ3")
    ZZZ3 = mul(mul(pzzz, qzzz, "ZZZ12"), PPP, "ZZZ3")
    check_point(X3, Y3, ZZ3, ZZZ3, where)
    pti_double(px, py, pzz, pzzz, where)


def main():
    pti_madd()
    pti_mmadd()
    pti_add_nz()
    if problems:
        print("LIMB BOUNDS VIOLATED:")
        for p in sorted(set(problems)):
            print("  -", p)
        return 1
    print("fq29 bounds: every column sum < 2^64, every lifted subtraction non-negative, every returned point within "
          "the invariant (pti_madd, pti_mmadd, pti_add_nz, pti_double)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
