"""Generates the constants of csrc/bn254_fq29.hip.h (29-bit-limb internal field representation).

Internal Montgomery radix rho = 2^261 (9 limbs x 29 bits).  External representation (what the host
libraries and the reference use) is Montgomery with R = 2^256 on 8 x 32-bit limbs.
"""
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
B, L = 29, 9
M = (1 << B) - 1
RHO = 1 << (B * L)


def limbs(x):
    out = [(x >> (B * i)) & M for i in range(L - 1)]
    out.append(x >> (B * (L - 1)))
    return out


def fmt(name, arr, comment=""):
    return f"  // {comment}\n  {name} = {{" + ", ".join(f"0x{v:08X}u" for v in arr) + "};"


def lifted(k, e):
    c = limbs(k * P)
    bor = 1 << (e - B)
    out = [c[0] + (1 << e)] + [c[i] + (1 << e) - bor for i in range(1, L - 1)] + [c[L - 1] - bor]
    assert sum(v << (B * i) for i, v in enumerate(out)) == k * P
    assert all(0 <= v < (1 << 32) for v in out)
    return out


print("P29      ", [hex(v) for v in limbs(P)])
print("INV29     0x%08X   (-p^-1 mod 2^29)" % ((-pow(P, -1, 1 << B)) % (1 << B)))
print("PINV29    0x%08X   ( p^-1 mod 2^29)" % (pow(P, -1, 1 << B)))
print("ONE29    ", [hex(v) for v in limbs(RHO % P)], " rho mod p (internal one)")
print("C_IN     ", [hex(v) for v in limbs(pow(2, 2 * B * L - 256, P))], " 2^(2*261-256) mod p: ext -> int")
print("D_OUT    ", [hex(v) for v in limbs(pow(2, 256, P))], " 2^256 mod p: int -> ext")
for k, e in ((4, 30), (8, 30), (8, 31), (16, 30), (16, 31)):
    print(f"K{k}E{e}  ", [hex(v) for v in lifted(k, e)], f" top-limb headroom {limbs(k*P)[8] - (1 << (e-B))}")
print("p/rho =", P / RHO, " p>>232 =", P >> 232, hex(P >> 232))
