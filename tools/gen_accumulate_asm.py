"""Generates metal-msm-gpu-acceleration_amd/csrc/k_accumulate_asm.inc: the body of accumulate_kernel_asm as ONE
inline-assembly statement with hand-allocated registers.

Why: the mixed addition of the accumulate kernel (madd-2008-s on 9 x 29-bit limbs) needs ~170 live registers the way
the compiler schedules it, i.e. 2 waves per SIMD, and at 2 waves a SIMD issues a v_mad_u64_u32 only every 4-5 cycles
(each wave can issue one every 8-9.5, whatever the other wave does): tools/microbench/mul_occ.hip measures the field
multiplication at 986 cycles per SIMD with 2 waves resident, 715 with 4, 572 with 5.  The compiler cannot be talked
below 128 registers without spilling (profiles/r04_ab_accumulate_occupancy.txt), so this generator allocates them:

  * the multiplication runs row by row (operand scanning with the Montgomery reduction of row i interleaved): only
    9 of the 17 column sums are open at any time -- a rotating window of 9 register pairs;
  * the accumulator's Y, ZZ and ZZZ coordinates live in LDS between their uses (27 dwords per lane, 6.75 KiB per
    wave), X stays in registers;
  * the 64-byte record of the next point is gathered in the middle of the addition into 16 registers that are free
    from the unpacking of one point to the middle of its addition;
  * 87 registers for the assembly + what the compiler needs for the operands = at most 96: FIVE waves per SIMD.

The statement handles the common path only.  A lane that meets an identity base or a possible exceptional case of the
addition law (P = U2 - X1 = 0 mod p by the one-limb filter) sets its flag, finishes with whatever it has, and appends
its work item to a redo list; accumulate_redo_kernel (the compiler-built accumulate_body) recomputes those items.

`python tools/gen_accumulate_asm.py --selftest` runs the emitted instruction stream of one point addition through a
small interpreter and compares limbs and group elements with Python big integers (no GPU needed);
`python tools/gen_accumulate_asm.py out.inc` writes the include file (done by the Makefile).
"""
import re
import sys
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "csrc", "bn254_fq29.hip.h")

# ---- constants from the header (single source of truth) ---------------------------------------------------------------
_h = open(HDR).read()


def _arr(name):
    m = re.search(name + r"\(int i\) \{[^{]*\{([^}]*)\}", _h, re.S)
    return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]


P_LIMBS = _arr("p")
ONE = _arr("one_c")
_kc = re.search(r"kc\(int sel, int i\) \{.*?c\[5\]\[9\] = \{(.*?)\};", _h, re.S).group(1)
KC = [[int(x.strip().rstrip("u"), 16) for x in row.split(",")] for row in re.findall(r"\{([^{}]*)\}", _kc)]
K4E30, K8E30, K8E31, K16E30, K16E31 = range(5)
MASK = 0x1FFFFFFF
INV = int(re.search(r"INV = (0x[0-9A-Fa-f]+)u", _h).group(1), 16)
PINV = int(re.search(r"PINV = (0x[0-9A-Fa-f]+)u", _h).group(1), 16)
P = sum(v << (29 * i) for i, v in enumerate(P_LIMBS))
RHO = 1 << 261
assert len(KC) == 5 and all(len(r) == 9 for r in KC) and (INV * P + 1) % (1 << 29) == 0 and (PINV * P) % (1 << 29) == 1
assert sum(v << (29 * i) for i, v in enumerate(ONE)) == RHO % P

# ---- register map (v0 .. v86) -----------------------------------------------------------------------------------------
W = [(2 * k, 2 * k + 1) for k in range(9)]          # window: 9 even-aligned pairs, v0..v17
VM, VT = 18, 19                                       # quotient digit, scratch
PRE = list(range(20, 36))                             # gathered record: x = v20..27, y = v28..35 (4 quads)
X = list(range(36, 45))                               # accumulator X (two quads + one)
V_I, V_FLAG, V_CUR, V_NEXT, V_NN = 45, 46, 47, 48, 49
QUADS = [[b, b + 1, b + 2, b + 3] for b in range(52, 84, 4)]   # 8 quads
SINGLES = [50, 51, 84, 85, 86]
N_ASM_VGPRS = 87
S_SAVE, S_TMP = 44, 46                                # SGPR pairs the statement clobbers (s44:45, s46:47)

# LDS parking: coordinate c (0 = Y, 1 = ZZ, 2 = ZZZ): limbs 0-3 and 4-7 as two b128 slots, limb 8 as one b32
LDS_B128_STRIDE = 64 * 16
LDS_B32_BASE = 6 * LDS_B128_STRIDE
LDS_BYTES_PER_WAVE = LDS_B32_BASE + 3 * 64 * 4
CY, CZZ, CZZZ = 0, 1, 2


class Pool:
    def __init__(self):
        self.quads = [list(q) for q in QUADS]
        self.singles = list(SINGLES)
        self.peak = 0

    def fe(self):
        """9 registers: two aligned quads + one single."""
        if len(self.quads) < 2 or not self.singles:
            raise RuntimeError("register pool exhausted")
        r = self.quads.pop(0) + self.quads.pop(0) + [self.singles.pop(0)]
        return r

    def single(self):
        return self.singles.pop(0)

    def free(self, fe):
        if isinstance(fe, int):
            self.singles.append(fe)
            return
        assert len(fe) == 9
        self.quads.append(fe[0:4])
        self.quads.append(fe[4:8])
        self.singles.append(fe[8])


# ---- instruction stream -------------------------------------------------------------------------------------------------
class Prog:
    """Instructions as (text, kind, writes, reads, sem) -- text for the assembler, sem for the interpreter."""

    def __init__(self):
        self.ins = []

    def add(self, text, kind, writes=(), reads=(), sem=None):
        self.ins.append((text, kind, frozenset(writes), frozenset(reads), sem))

    def raw(self, text):
        self.ins.append((text, "raw", frozenset(), frozenset(), None))

    def render(self, nops=True):
        """Text with the wait states between dependent neighbours (the compiler puts an s_nop 0 between back-to-back
        dependent VALU instructions on gfx950; inside an inline-assembly statement nobody does)."""
        out = []
        prev = None
        for text, kind, wr, rd, _ in self.ins:
            if prev is not None and prev[1] == "valu" and kind in ("valu", "lds", "vmem", "salu") and (prev[2] & rd):
                if nops:
                    out.append("s_nop 1" if kind == "salu" else "s_nop 0")
            out.append(text)
            if kind != "comment":
                prev = (text, kind, wr, rd)
        return out


def v(r):
    return "v%d" % r


def vp(lo):
    return "v[%d:%d]" % (lo, lo + 1)


def vq(lo):
    return "v[%d:%d]" % (lo, lo + 3)


def R(*regs):
    return {"v%d" % r for r in regs}


# operand placeholders of the inline-assembly statement (named operands)
OP_P = ["%%[p%d]" % j for j in range(9)]


class Gen:
    def __init__(self, prog, pool):
        self.p = prog
        self.pool = pool
        self.sdst = "vcc"        # where the (unused) carry-out of every v_mad_u64_u32 goes
        self.adjacent_m = False  # experiment: quotient digit mul_lo / and / first use back to back (the wave stalls)
        self.interleave = False  # experiment: the reduction of row i interleaved with the products of row i + 1
        self.p_vgprs = None      # experiment: the limbs of p in VGPRs instead of SGPR operands

    # -- basic emitters (each with interpreter semantics on a register dict `g`) ----------------------------------------
    def comment(self, s):
        self.p.add("; " + s, "comment")

    def mad(self, col_pair, a, b, first, b_is_p=None):
        lo = col_pair[0]
        src2 = "0" if first else vp(lo)
        btxt = (OP_P[b_is_p] if self.p_vgprs is None else v(self.p_vgprs[b_is_p])) if b_is_p is not None else v(b)
        reads = R(a) | (set() if first else R(lo, lo + 1)) | (set() if b_is_p is not None else R(b))

        def sem(g, lo=lo, a=a, b=b, first=first, b_is_p=b_is_p):
            bv = P_LIMBS[b_is_p] if b_is_p is not None else g[b]
            acc = 0 if first else (g[lo] | (g[lo + 1] << 32))
            s = acc + g[a] * bv
            assert s < (1 << 64), "column overflow"
            g[lo], g[lo + 1] = s & 0xFFFFFFFF, s >> 32
        self.p.add("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(lo), self.sdst, v(a), btxt, src2), "valu", R(lo, lo + 1) | {self.sdst}, reads, sem)

    def op2(self, name, d, a_txt, b, areg=None, imm=None):
        """VOP2 d = a (op) b with a = register or literal."""
        reads = R(b) | (R(areg) if areg is not None else set())

        def sem(g, name=name, d=d, areg=areg, imm=imm, b=b):
            a = g[areg] if areg is not None else imm
            bb = g[b]
            if name == "v_add_u32":
                g[d] = (a + bb) & 0xFFFFFFFF
            elif name == "v_sub_u32":
                g[d] = (a - bb) & 0xFFFFFFFF
            elif name == "v_and_b32":
                g[d] = a & bb
            elif name == "v_or_b32":
                g[d] = a | bb
            elif name == "v_lshrrev_b32":
                g[d] = bb >> (a & 31)
            elif name == "v_lshlrev_b32":
                g[d] = (bb << (a & 31)) & 0xFFFFFFFF
            else:
                raise KeyError(name)
        self.p.add("%s %s, %s, %s" % (name, v(d), a_txt, v(b)), "valu", R(d), reads, sem)

    def add_lit(self, d, lit, b):
        self.op2("v_add_u32", d, "0x%x" % lit, b, imm=lit)

    def add(self, d, a, b):
        self.op2("v_add_u32", d, v(a), b, areg=a)

    def sub(self, d, a, b):
        self.op2("v_sub_u32", d, v(a), b, areg=a)

    def sub_lit(self, d, lit, b):   # d = lit - b
        self.op2("v_sub_u32", d, "0x%x" % lit, b, imm=lit)

    def and_mask(self, d, b):
        self.op2("v_and_b32", d, "0x%x" % MASK, b, imm=MASK)

    def shr(self, d, n, b):
        self.op2("v_lshrrev_b32", d, str(n), b, imm=n)

    def shl(self, d, n, b):
        self.op2("v_lshlrev_b32", d, str(n), b, imm=n)

    def mov(self, d, a):
        def sem(g, d=d, a=a):
            g[d] = g[a]
        self.p.add("v_mov_b32 %s, %s" % (v(d), v(a)), "valu", R(d), R(a), sem)

    def mov_lit(self, d, lit):
        def sem(g, d=d, lit=lit):
            g[d] = lit
        self.p.add("v_mov_b32 %s, 0x%x" % (v(d), lit), "valu", R(d), set(), sem)

    def mul_lo_inv(self, d, a):
        def sem(g, d=d, a=a):
            g[d] = (g[a] * INV) & 0xFFFFFFFF
        self.p.add("v_mul_lo_u32 %s, %s, %%[inv]" % (v(d), v(a)), "valu", R(d), R(a), sem)

    def shr64(self, pair, n):
        lo = pair[0]

        def sem(g, lo=lo, n=n):
            x = (g[lo] | (g[lo + 1] << 32)) >> n
            g[lo], g[lo + 1] = x & 0xFFFFFFFF, x >> 32
        self.p.add("v_lshrrev_b64 %s, %d, %s" % (vp(lo), n, vp(lo)), "valu", R(lo, lo + 1), R(lo, lo + 1), sem)

    def add64(self, dpair, apair):   # d += a
        d, a = dpair[0], apair[0]

        def sem(g, d=d, a=a):
            s = (g[d] | (g[d + 1] << 32)) + (g[a] | (g[a + 1] << 32))
            assert s < (1 << 64), "column overflow in carry"
            g[d], g[d + 1] = s & 0xFFFFFFFF, s >> 32
        self.p.add("v_lshl_add_u64 %s, %s, 0, %s" % (vp(d), vp(a), vp(d)), "valu", R(d, d + 1), R(d, d + 1, a, a + 1), sem)

    def alignbit(self, d, hi, lo, n):
        def sem(g, d=d, hi=hi, lo=lo, n=n):
            g[d] = (((g[hi] << 32) | g[lo]) >> n) & 0xFFFFFFFF
        self.p.add("v_alignbit_b32 %s, %s, %s, %d" % (v(d), v(hi), v(lo), n), "valu", R(d), R(hi, lo), sem)

    # -- LDS parking ----------------------------------------------------------------------------------------------------
    def park(self, coord, fe):
        def sem(g, coord=coord, fe=tuple(fe)):
            g["lds"][coord] = [g[r] for r in fe]
        a, b = fe[0], fe[4]
        assert fe[1:4] == [a + 1, a + 2, a + 3] and fe[5:8] == [b + 1, b + 2, b + 3] and a % 2 == 0 and b % 2 == 0
        self.p.add("ds_write_b128 %%[lds128], %s offset:%d" % (vq(a), (2 * coord) * LDS_B128_STRIDE), "lds", (), R(*fe[0:4]), sem)
        self.p.add("ds_write_b128 %%[lds128], %s offset:%d" % (vq(b), (2 * coord + 1) * LDS_B128_STRIDE), "lds", (), R(*fe[4:8]))
        self.p.add("ds_write_b32 %%[lds32], %s offset:%d" % (v(fe[8]), coord * 256), "lds", (), R(fe[8]))

    def unpark(self, coord, fe):
        def sem(g, coord=coord, fe=tuple(fe)):
            for r, val in zip(fe, g["lds"][coord]):
                g[r] = val
        a, b = fe[0], fe[4]
        self.p.add("ds_read_b128 %s, %%[lds128] offset:%d" % (vq(a), (2 * coord) * LDS_B128_STRIDE), "lds", R(*fe[0:4]), (), sem)
        self.p.add("ds_read_b128 %s, %%[lds128] offset:%d" % (vq(b), (2 * coord + 1) * LDS_B128_STRIDE), "lds", R(*fe[4:8]), ())
        self.p.add("ds_read_b32 %s, %%[lds32] offset:%d" % (v(fe[8]), coord * 256), "lds", R(fe[8]), ())

    def wait_lds(self):
        self.p.add("s_waitcnt lgkmcnt(0)", "wait")

    # -- field operations -------------------------------------------------------------------------------------------------
    def mul_interleaved(self, a, b, out):
        """Single product, the reduction mads of row i alternating with the product mads of row i + 1."""
        touched = set()

        def col_mad(c, x, y, b_is_p=None):
            self.mad(W[c % 9], x, y, c not in touched, b_is_p)
            touched.add(c)
        for j in range(9):
            col_mad(j, a[j], b[0])
        for i in range(9):
            self.mul_lo_inv(VM, W[i % 9][0])
            self.and_mask(VM, VM)
            # reduction row i: columns i .. i + 8; product row i + 1: columns i + 1 .. i + 9 (column i + 9 only after the
            # pair of column i has been retired)
            col_mad(i, VM, None, b_is_p=0)
            self.shr64(W[i % 9], 29)
            for j in range(1, 9):
                col_mad(i + j, VM, None, b_is_p=j)
                if j == 1:
                    self.add64(W[(i + 1) % 9], W[i % 9])
                if i < 8:
                    col_mad(i + j, a[j - 1], b[i + 1])
            if i < 8:
                col_mad(i + 9, a[8], b[i + 1])
        for k in range(8):
            c = 9 + k
            self.and_mask(out[k], W[c % 9][0])
            if k < 7:
                self.shr64(W[c % 9], 29)
                self.add64(W[(c + 1) % 9], W[c % 9])
            else:
                self.alignbit(out[8], W[c % 9][1], W[c % 9][0], 29)

    def mul(self, prods, out):
        """out = sum of the products (Montgomery, radix 2^261), row by row.  prods: ('mul', a, b) or ('sqr', a).
        out: 9 registers; they MAY be the registers of an input: every input limb is read inside the nine rows, the
        output limbs are only written by the final carry pass after them."""
        touched = set()

        def col_mad(c, a, b, b_is_p=None):
            self.mad(W[c % 9], a, b, c not in touched, b_is_p)
            touched.add(c)
        for i in range(9):
            seq = []
            for pr in prods:
                if pr[0] == "mul":
                    _, a, b = pr
                    seq += [(i + j, a[j], b[i]) for j in range(9)]
                else:
                    _, a = pr
                    seq.append((2 * i, a[i], a[i]))
                    if i < 8:
                        self.shl(VT, 1, a[i])
                        seq += [(i + j, VT, a[j]) for j in range(i + 1, 9)]
            # column i first (its sum is complete after this row's contribution); the quotient digit is computed in the
            # shadow of the remaining products so that no dependent instructions sit next to each other
            seq.sort(key=lambda t: (t[0] != i,))
            last_i = max([idx for idx, t in enumerate(seq) if t[0] == i], default=-1)
            pending = {} if self.adjacent_m else {last_i + 2: "mul_lo", last_i + 4: "and"}
            done = set()
            for k, (c, a, b) in enumerate(seq):
                if pending.get(k) and pending[k] not in done:
                    (self.mul_lo_inv(VM, W[i % 9][0]) if pending[k] == "mul_lo" else self.and_mask(VM, VM))
                    done.add(pending[k])
                col_mad(c, a, b)
            if "mul_lo" not in done:
                self.mul_lo_inv(VM, W[i % 9][0])
            if "and" not in done:
                self.and_mask(VM, VM)
            for j in range(9):
                col_mad(i + j, VM, None, b_is_p=j)
                if j == 2:
                    self.shr64(W[i % 9], 29)
                if j == 4:
                    self.add64(W[(i + 1) % 9], W[i % 9])
        # columns 9 .. 16 hold the result
        for k in range(8):
            c = 9 + k
            self.and_mask(out[k], W[c % 9][0])
            if k < 7:
                self.shr64(W[c % 9], 29)
                self.add64(W[(c + 1) % 9], W[c % 9])
            else:
                self.alignbit(out[8], W[c % 9][1], W[c % 9][0], 29)

    def norm_inplace(self, a):
        """One parallel carry round, top limb first (limb i needs the ORIGINAL limb i - 1)."""
        self.shr(VT, 29, a[7])
        self.add(a[8], a[8], VT)
        for i in range(7, 0, -1):
            self.shr(VT, 29, a[i - 1])
            self.and_mask(a[i], a[i])
            self.add(a[i], a[i], VT)
        self.and_mask(a[0], a[0])

    def sub_k(self, d, a, ksel, b):
        """d = a + K - b (limbs; d may alias a or b)."""
        for i in range(9):
            self.add_lit(VT, KC[ksel][i], a[i])
            self.sub(d[i], VT, b[i])

    # -- one point addition, common path ------------------------------------------------------------------------------------
    def unpack(self, words, out):
        """256-bit little-endian integer in 8 registers -> 9 limbs of 29 bits."""
        for i in range(9):
            bit = 29 * i
            w, s = bit >> 5, bit & 31
            if i == 0:
                self.and_mask(out[0], words[0])
            elif i == 8:
                self.shr(out[8], s, words[7])
            else:
                self.alignbit(out[i], words[w + 1], words[w], s)
                self.and_mask(out[i], out[i])

    def flag_if(self, cond_text, reads):
        """flag |= 1 where the compare (text writing vcc) holds."""
        self.p.add(cond_text, "valu", {"vcc"}, reads, ("cmp", cond_text))
        self.p.add("v_cndmask_b32 %s, 0, 1, vcc" % v(VT), "valu", R(VT), {"vcc"}, ("cnd01", VT))
        self.op2("v_or_b32", V_FLAG, v(V_FLAG), VT, areg=V_FLAG)

    def load_point(self, xy, negate_from):
        """PRE (packed x | y) -> xy = (x2, y2) with y2 negated (lazily) where the index word has its sign bit set."""
        x2, y2 = xy
        self.comment("unpack the gathered base; an identity base (x word 7 = 0xffffffff) flags the lane")
        self.flag_if("v_cmp_eq_u32 vcc, -1, %s" % v(PRE[7]), R(PRE[7]))
        self.unpack(PRE[0:8], x2)
        self.unpack(PRE[8:16], y2)
        self.comment("negative digit: y := K4E30 - y (limbs < 2^30.5, never normalised: see pti_madd)")
        self.p.add("v_cmp_gt_i32 vcc, 0, %s" % v(negate_from), "valu", {"vcc"}, R(negate_from), ("cmp_neg", negate_from))
        for i in range(9):
            self.sub_lit(VT, KC[K4E30][i], y2[i])
            self.p.add("v_cndmask_b32 %s, %s, %s, vcc" % (v(y2[i]), v(y2[i]), v(VT)), "valu", R(y2[i]), R(y2[i], VT) | {"vcc"},
                       ("cndsel", y2[i], VT))

    def madd(self, midway):
        """acc += (PRE unpacked): X in registers, Y / ZZ / ZZZ parked in LDS.  `midway()` emits the prefetch."""
        pool = self.pool
        x2, y2 = pool.fe(), pool.fe()
        self.load_point((x2, y2), V_CUR)
        zz = pool.fe()
        self.unpark(CZZ, zz)
        self.wait_lds()
        self.comment("U2 = x2 * ZZ1 ; P = norm(U2 + K16E30 - X1)")
        Pv = x2                                   # the product lands in the registers of its dead operand
        self.mul([("mul", x2, zz)], Pv)
        self.sub_k(Pv, Pv, K16E30, X)
        self.norm_inplace(Pv)
        self.comment("exceptional case filter: (P0 * p^-1 mod 2^29) < 18  =>  P may be 0 mod p: flag the lane")
        self.and_mask(VT, Pv[0])
        self.p.add("v_mul_lo_u32 %s, %s, %%[pinv]" % (v(VT), v(VT)), "valu", R(VT), R(VT), ("mul_pinv", VT))
        self.and_mask(VT, VT)
        self.p.add("v_cmp_gt_u32 vcc, 18, %s" % v(VT), "valu", {"vcc"}, R(VT), ("cmp_lt18", VT))
        self.p.add("v_cndmask_b32 %s, 0, 1, vcc" % v(VT), "valu", R(VT), {"vcc"}, ("cnd01", VT))
        self.op2("v_or_b32", V_FLAG, v(V_FLAG), VT, areg=V_FLAG)
        self.comment("S2 = y2 * ZZZ1 ; R = norm(S2 + K8E30 - Y1)")
        zzz = pool.fe()
        self.unpark(CZZZ, zzz)
        self.wait_lds()
        Rv = y2
        self.mul([("mul", y2, zzz)], Rv)
        pool.free(zzz)
        y1 = pool.fe()
        self.unpark(CY, y1)
        self.wait_lds()
        self.sub_k(Rv, Rv, K8E30, y1)
        pool.free(y1)
        self.norm_inplace(Rv)
        self.comment("PP = P^2 ; ZZ3 = ZZ1 * PP (parked) ; Q = X1 * PP ; PPP = P * PP")
        PP = pool.fe()
        self.mul([("sqr", Pv)], PP)
        zz3 = zz
        self.mul([("mul", zz, PP)], zz3)
        self.park(CZZ, zz3)
        pool.free(zz3)
        Q = pool.fe()
        self.mul([("mul", X, PP)], Q)
        PPP = Pv
        self.mul([("mul", Pv, PP)], PPP)
        pool.free(PP)
        self.comment("ZZZ3 = ZZZ1 * PPP (parked)")
        zzz = pool.fe()
        self.unpark(CZZZ, zzz)
        self.wait_lds()
        zzz3 = zzz
        self.mul([("mul", zzz, PPP)], zzz3)
        self.park(CZZZ, zzz3)
        pool.free(zzz3)
        midway()
        self.comment("X3 = norm(R^2 + K8E31 - (PPP + 2 Q)) -> the X registers")
        RR = pool.fe()
        self.mul([("sqr", Rv)], RR)
        for i in range(9):
            self.add(VT, Q[i], Q[i])
            self.add(VT, VT, PPP[i])
            self.add_lit(X[i], KC[K8E31][i], RR[i])
            self.sub(X[i], X[i], VT)
        pool.free(RR)
        self.norm_inplace(X)
        self.comment("T = Q + K16E30 - X3 (not normalised) ; -PPP = K4E30 - PPP ; Y3 = R * T + Y1 * (-PPP) (parked)")
        self.sub_k(Q, Q, K16E30, X)
        for i in range(9):
            self.sub_lit(PPP[i], KC[K4E30][i], PPP[i])
        y1 = pool.fe()
        self.unpark(CY, y1)
        self.wait_lds()
        y3 = y1
        self.mul([("mul", Rv, Q), ("mul", y1, PPP)], y3)
        pool.free(Rv)
        pool.free(Q)
        pool.free(PPP)
        self.park(CY, y3)
        pool.free(y3)

    def first_point(self):
        """acc := the first base: X = x, Y = norm(+-y), ZZ = ZZZ = one."""
        pool = self.pool
        x2, y2 = pool.fe(), pool.fe()
        self.load_point((x2, y2), V_CUR)
        for i in range(9):
            self.mov(X[i], x2[i])
        self.norm_inplace(y2)
        self.park(CY, y2)
        for i in range(9):
            self.mov_lit(x2[i], ONE[i])
        self.park(CZZ, x2)
        self.park(CZZZ, x2)
        pool.free(x2)
        pool.free(y2)


# ---- interpreter (one lane, straight-line arithmetic only) -------------------------------------------------------------
def run(prog, g):
    vcc = 0
    for text, kind, wr, rd, sem in prog.ins:
        if sem is None:
            continue
        if callable(sem):
            sem(g)
            continue
        tag = sem[0]
        if tag == "cmp":          # v_cmp_eq_u32 vcc, -1, PRE7
            vcc = 1 if g[PRE[7]] == 0xFFFFFFFF else 0
        elif tag == "cmp_neg":
            vcc = 1 if g[sem[1]] & 0x80000000 else 0
        elif tag == "cmp_lt18":
            vcc = 1 if g[sem[1]] < 18 else 0
        elif tag == "cnd01":
            g[sem[1]] = vcc
        elif tag == "cndsel":
            if vcc:
                g[sem[1]] = g[sem[2]]
        elif tag == "mul_pinv":
            g[sem[1]] = (g[sem[1]] * PINV) & 0xFFFFFFFF
        else:
            raise KeyError(tag)


# ---- Python twins of the C++ field / group code (limb exact) -------------------------------------------------------------
def limbs_of(x):
    return [(x >> (29 * i)) & MASK for i in range(8)] + [x >> 232]


def val(l):
    return sum(v << (29 * i) for i, v in enumerate(l))


def py_mul2(pairs):
    A = [0] * 17
    for a, b in pairs:
        for i in range(9):
            for j in range(9):
                A[i + j] += a[i] * b[j]
    carry = 0
    for k in range(9):
        A[k] += carry
        m = ((A[k] & 0xFFFFFFFF) * INV) & MASK
        for j in range(9):
            A[k + j] += m * P_LIMBS[j]
        carry = A[k] >> 29
    r = []
    for k in range(9, 17):
        A[k] += carry
        r.append(A[k] & MASK)
        carry = A[k] >> 29
    r.append(carry & 0xFFFFFFFF)
    return r


def py_norm(a):
    r = [a[0] & MASK] + [(a[i] & MASK) + (a[i - 1] >> 29) for i in range(1, 8)] + [(a[8] + (a[7] >> 29)) & 0xFFFFFFFF]
    return r


def py_sub(a, k, b):
    return [((a[i] + KC[k][i]) - b[i]) & 0xFFFFFFFF for i in range(9)]


def py_madd(acc, q):
    X1, Y1, ZZ1, ZZZ1 = acc
    x2, y2 = q
    U2 = py_mul2([(x2, ZZ1)])
    S2 = py_mul2([(y2, ZZZ1)])
    Pv = py_norm(py_sub(U2, K16E30, X1))
    Rv = py_norm(py_sub(S2, K8E30, Y1))
    PP = py_mul2([(Pv, Pv)])
    PPP = py_mul2([(Pv, PP)])
    Q = py_mul2([(X1, PP)])
    RR = py_mul2([(Rv, Rv)])
    t = [(PPP[i] + 2 * Q[i]) & 0xFFFFFFFF for i in range(9)]
    X3 = py_norm(py_sub(RR, K8E31, t))
    T = py_sub(Q, K16E30, X3)
    nP = py_sub([0] * 9, K4E30, PPP)
    Y3 = py_mul2([(Rv, T), (Y1, nP)])
    return X3, Y3, py_mul2([(ZZ1, PP)]), py_mul2([(ZZZ1, PPP)])


def selftest(seed=1, rounds=6):
    sys.path.insert(0, ROOT)
    from oracle import bn254_ref as o
    rng = random.Random(seed)
    to_int = lambda x: x * RHO % P          # canonical -> internal Montgomery domain

    def pack(x):    # canonical internal value -> 8 little-endian words
        return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    for rnd in range(rounds):
        pts = [o.scalar_mul(rng.randrange(1, o.R_ORDER), o.GEN) for _ in range(4)]
        negs = [rng.random() < 0.5 for _ in pts]
        prog, pool = Prog(), Pool()
        gen = Gen(prog, pool)
        g = {r: rng.getrandbits(32) for r in range(N_ASM_VGPRS)}
        g["lds"] = {}
        g[V_FLAG] = 0
        expect = None
        acc_py = None
        for k, (pt, ng) in enumerate(zip(pts, negs)):
            prog.ins.clear()
            for j, wd in enumerate(pack(to_int(pt[0])) + pack(to_int(pt[1]))):
                g[PRE[j]] = wd
            g[V_CUR] = (0x80000000 if ng else 0) | rng.getrandbits(20)
            q_aff = (pt[0], (-pt[1]) % P if ng else pt[1])
            qx = limbs_of(to_int(pt[0]))
            qy = limbs_of(to_int(pt[1]))
            if ng:
                qy = py_sub([0] * 9, K4E30, qy)
            if k == 0:
                gen.first_point()
                run(prog, g)
                acc_py = (qx, py_norm(qy), list(ONE), list(ONE))
                expect = q_aff
            else:
                gen.madd(lambda: None)
                run(prog, g)
                acc_py = py_madd(acc_py, (qx, qy))
                expect = o.aff_add(expect, q_aff)
            got = ([g[r] for r in X], g["lds"][CY], g["lds"][CZZ], g["lds"][CZZZ])
            assert got[0] == acc_py[0] and got[1] == acc_py[1] and got[2] == acc_py[2] and got[3] == acc_py[3], (rnd, k, "limbs differ")
            # the group element: x = X / ZZ, y = Y / ZZZ (all in the internal domain: the factors cancel)
            Xv, Yv, ZZv, ZZZv = [val(c) % P for c in got]
            assert (Xv * pow(ZZv, -1, P) % P, Yv * pow(ZZZv, -1, P) % P) == expect, (rnd, k, "point differs")
            assert g[V_FLAG] == 0
            assert sorted(sum(pool.quads, [])) == sorted(sum(QUADS, [])) and sorted(pool.singles) == sorted(SINGLES), "leak"
    # the flags
    prog, pool = Prog(), Pool()
    gen = Gen(prog, pool)
    g = {r: 0 for r in range(N_ASM_VGPRS)}
    g["lds"] = {}
    for j in range(16):
        g[PRE[j]] = 0xFFFFFFFF if j < 8 else 0
    gen.first_point()
    run(prog, g)
    assert g[V_FLAG] == 1
    print("selftest ok: %d rounds x 4 points, limbs and group elements agree; one addition: %s" % (rounds, count_madd()))


def count_madd():
    prog = Prog()
    Gen(prog, Pool()).madd(lambda: None)
    lines = prog.render()
    valu = sum(1 for l in lines if l.startswith("v_"))
    mult = sum(1 for l in lines if l.startswith("v_mad_u64") or l.startswith("v_mul_lo"))
    nops = sum(1 for l in lines if l.startswith("s_nop"))
    return "%d VALU (%d multiplier), %d s_nop, %d LDS" % (valu, mult, nops, sum(1 for l in lines if l.startswith("ds_")))


# ---- the whole statement --------------------------------------------------------------------------------------------------
def build():
    prog, pool = Prog(), Pool()
    gen = Gen(prog, pool)
    raw, c = prog.raw, gen.comment
    sv, st = "s[%d:%d]" % (S_SAVE, S_SAVE + 1), "s[%d:%d]" % (S_TMP, S_TMP + 1)

    def gather(idx_reg):
        """64 bytes of bases[idx & 0x7fffffff] -> PRE (address pair: window pair 0, free between multiplications)."""
        prog.add("v_and_b32 %s, 0x7fffffff, %s" % (v(VT), v(idx_reg)), "valu", R(VT), R(idx_reg))
        prog.add("v_mad_u64_u32 %s, vcc, %s, 64, %%[bases]" % (vp(0), v(VT)), "valu", R(0, 1) | {"vcc"}, R(VT))
        for k in range(4):
            prog.add("global_load_dwordx4 %s, %s, off offset:%d" % (vq(PRE[4 * k]), vp(0), 16 * k), "vmem", R(*PRE[4 * k:4 * k + 4]), R(0, 1))

    def masked(cond_lhs, body):
        """body() for the lanes with cond_lhs < cnt (cond_lhs: a register name or a constant)."""
        raw("v_cmp_lt_u32 vcc, %s, %%[cnt]" % cond_lhs)
        raw("s_nop 1")
        raw("s_and_saveexec_b64 %s, vcc" % st)
        body()
        raw("s_mov_b64 exec, %s" % st)

    c("accumulate_kernel_asm: generated by tools/gen_accumulate_asm.py -- do not edit")
    raw("s_mov_b64 %s, exec" % sv)
    raw("v_mov_b32 %s, 0" % v(V_FLAG))
    c("indices of points 0, 1, 2 (the lanes that have them); the record of point 0")
    raw("global_load_dword %s, %%[idxp], off" % v(V_CUR))
    masked("1", lambda: raw("global_load_dword %s, %%[idxp], off offset:4" % v(V_NEXT)))
    masked("2", lambda: raw("global_load_dword %s, %%[idxp], off offset:8" % v(V_NN)))
    raw("s_waitcnt vmcnt(0)")
    gather(V_CUR)
    raw("s_waitcnt vmcnt(0)")
    gen.first_point()
    raw("v_mov_b32 %s, 1" % v(V_I))
    c("record of point 1")
    masked("1", lambda: gather(V_NEXT))
    raw("1:")
    c("---- loop: lanes with i < cnt; on entry the record of point i is in flight, V_NEXT = idx[i], V_NN = idx[i + 1]")
    raw("v_cmp_lt_u32 vcc, %s, %%[cnt]" % v(V_I))
    raw("s_nop 1")
    raw("s_and_b64 exec, exec, vcc")
    raw("s_cbranch_execz 2f")
    raw("s_waitcnt vmcnt(0)")
    raw("v_mov_b32 %s, %s" % (v(V_CUR), v(V_NEXT)))
    raw("v_mov_b32 %s, %s" % (v(V_NEXT), v(V_NN)))
    raw("v_lshl_add_u64 %[idxp], %[idxp], 0, 4")

    def midway():
        c("middle of the addition: gather the record of point i + 1, fetch idx[i + 2]")
        raw("v_add_u32 %s, 1, %s" % (v(VM), v(V_I)))
        raw("s_nop 0")
        masked(v(VM), lambda: gather(V_NEXT))
        raw("v_add_u32 %s, 2, %s" % (v(VM), v(V_I)))
        raw("s_nop 0")
        masked(v(VM), lambda: raw("global_load_dword %s, %%[idxp], off offset:8" % v(V_NN)))
    gen.madd(midway)
    raw("v_add_u32 %s, 1, %s" % (v(V_I), v(V_I)))
    raw("s_branch 1b")
    raw("2:")
    c("---- every lane of the wave again: results to memory")
    raw("s_mov_b64 exec, %s" % sv)
    raw("s_waitcnt vmcnt(0)")
    fe = pool.fe()
    for coord, off in ((None, 0), (CY, 36), (CZZ, 72), (CZZZ, 108)):
        src = X if coord is None else fe
        if coord is not None:
            gen.unpark(coord, fe)
            gen.wait_lds()
        prog.add("global_store_dwordx4 %%[outp], %s, off offset:%d" % (vq(src[0]), off), "vmem", (), R(*src[0:4]))
        prog.add("global_store_dwordx4 %%[outp], %s, off offset:%d" % (vq(src[4]), off + 16), "vmem", (), R(*src[4:8]))
        prog.add("global_store_dword %%[outp], %s, off offset:%d" % (v(src[8]), off + 32), "vmem", (), R(src[8]))
        if coord is not None:
            raw("s_waitcnt vmcnt(0)")   # the registers are reused for the next coordinate
    pool.free(fe)
    c("flagged lanes append their work item to the redo list")
    raw("v_cmp_ne_u32 vcc, 0, %s" % v(V_FLAG))
    raw("s_nop 1")
    raw("s_and_saveexec_b64 %s, vcc" % st)
    raw("s_cbranch_execz 4f")
    raw("v_mov_b32 %s, 0" % v(VT))
    raw("v_mov_b32 %s, 1" % v(VM))
    raw("s_nop 0")
    raw("global_atomic_add %s, %s, %s, %%[redo_count] sc0" % (v(VM), v(VT), v(VM)))
    raw("s_waitcnt vmcnt(0)")
    raw("v_lshlrev_b32 %s, 2, %s" % (v(VM), v(VM)))
    raw("s_nop 0")
    raw("global_store_dword %s, %%[slot], %%[redo_list]" % v(VM))
    raw("4:")
    raw("s_mov_b64 exec, %s" % sv)
    raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return prog, pool


def build_bench(kind):
    """Timing-only statements for tools/microbench/mul_occ.hip: `iters` trips of
         mul      x = x * y            (the row-form multiplication alone)
         mulsub   x = norm(x * y + K - y)   (+ the simple-instruction tail of a field subtraction)
         mulpark  x = x * y with x parked in LDS and fetched back every trip
       x, y start from whatever the registers hold (values do not matter for the timing)."""
    prog, pool = Prog(), Pool()
    gen = Gen(prog, pool)
    x, y = pool.fe(), pool.fe()
    if kind == "mul_sgpr":
        gen.sdst = "s[%d:%d]" % (S_TMP, S_TMP + 1)
    if kind == "mul_nonop":
        gen.adjacent_m = True
    if kind == "mul_banks":      # operands in registers whose numbers are not multiples of 4 apart
        x = [52, 57, 62, 67, 72, 77, 82, 55, 60]
        y = [53, 58, 63, 68, 73, 78, 83, 56, 61]
    prog.raw("v_mov_b32 %s, 0" % v(V_I))
    for r in x + y:
        prog.raw("v_and_b32 %s, 0x%x, %s" % (v(r), MASK, v(r)))
    prog.raw("1:")
    if kind == "mulpark":
        gen.unpark(CY, x)
        gen.wait_lds()
    if kind == "mul_pv":
        gen.p_vgprs = [20 + j for j in range(9)]
        for j in range(9):
            prog.ins.insert(0, ("v_mov_b32 v%d, %%[p%d]" % (20 + j, j), "raw", frozenset(), frozenset(), None))
    if kind == "mul_il":
        gen.mul_interleaved(x, y, x)
    else:
        gen.mul([("mul", x, y)], x)
    if kind == "mulsub":
        gen.sub_k(x, x, K16E30, y)
        gen.norm_inplace(x)
    if kind == "mulpark":
        gen.park(CY, x)
    prog.raw("v_add_u32 %s, 1, %s" % (v(V_I), v(V_I)))
    prog.raw("s_nop 0")
    prog.raw("v_cmp_lt_u32 vcc, %s, %%[cnt]" % v(V_I))
    prog.raw("s_nop 1")
    prog.raw("s_cbranch_vccnz 1b")
    return prog


def write_inc(path):
    prog, _ = build()
    lines = prog.render()
    with open(path, "w") as f:
        f.write("// generated by tools/gen_accumulate_asm.py -- do not edit (make regenerates it)\n")
        f.write("#define MSM_ACC_ASM_LDS_BYTES %d\n" % LDS_BYTES_PER_WAVE)
        f.write("#define MSM_ACC_ASM_TEXT \\\n")
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln.replace('"', '\\"'))
        f.write('  ""\n')
        if os.environ.get("MSM_ASM_BENCH"):
            for kind in ("mul", "mulsub", "mulpark", "mul_sgpr", "mul_banks", "mul_nonop", "mul_il", "mul_pv"):
                f.write("#define MSM_ACC_ASM_BENCH_%s \\\n" % kind.upper())
                for ln in build_bench(kind).render(nops=(kind != "mul_nonop")):
                    f.write('  "%s\\n\\t" \\\n' % ln.replace('"', '\\"'))
                f.write('  ""\n')
        clob = ['"v%d"' % r for r in range(N_ASM_VGPRS)] + ['"s%d"' % r for r in range(S_SAVE, S_TMP + 2)] + ['"vcc"', '"scc"', '"memory"']
        f.write("#define MSM_ACC_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    print("wrote %s: %d lines; one addition: %s" % (path, len(lines), count_madd()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--selftest":
        selftest()
    elif len(sys.argv) > 1:
        write_inc(sys.argv[1])
    else:
        print(__doc__)
