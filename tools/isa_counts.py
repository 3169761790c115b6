"""Instruction counts of the shipped accumulate kernel, taken from the compiler's own assembly.

`make -C metal-msm-gpu-acceleration_amd/csrc isa-counts` (part of `all`) compiles k_accumulate.hip once more with
`--cuda-device-only -S` (same flags as the object that goes into libmsm_amd.so) and runs this script on the result.
It writes metal-msm-gpu-acceleration_amd/isa_counts.json, which bench.py reads for `roofline.secondary` (multiplier
instructions per mixed addition / per affine start) instead of constants typed into the bench by hand.

How the paths are told apart: the source carries comment marks (MSM_ISA_MARK, bn254_fq29.hip.h) that the compiler
copies into the assembly without emitting an instruction:
    "begin X"   first statement of path X inside the loop (X = mixed_addition: pti_madd, madd-2008-s 8M + 2S;
                X = affine_start: pti_mmadd, mmadd-2008-s 4M + 2S)
    "rare X"    entry of the exceptional-case block (exact zero tests, doubling) -- not counted
    "resume X"  first statement after that block: the compiler lays this part out as ONE basic block
    "end"       last statement of the path
The count of a path = instructions from "begin X" to "rare X", plus the basic block that starts at "resume X", plus
whatever multiplication the compiler hoisted above the path split (it speculates the first product of pti_madd
for every lane; those regions sit between the loop header and the first mark).  The script fails loudly if the
totals do not look like 8M + 2S / 4M + 2S on 9 x 29-bit limbs.

Usage: python tools/isa_counts.py k_accumulate.s out.json
"""
import collections
import json
import re
import sys

MULT = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_i64_i32")


def functions(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(ln)
    if name:
        yield name, body


def is_instr(s):
    return bool(s) and not s.startswith(";") and not s.startswith(".") and not s.endswith(":")


def path_counts(body):
    """Counters per tag: 'hoisted', 'mixed_addition', 'affine_start'.  Tagging is per basic-block region (the run of
    instructions between labels / branches): arithmetic may move across a mark inside a block, a block cannot."""
    header = next(i for i, ln in enumerate(body) if "This Inner Loop Header" in ln)
    regs = []          # (Counter, [marks])
    cur, marks = collections.Counter(), []
    for ln in body[header:]:
        s = ln.strip()
        m = re.match(r"^; MSM_MARK (\w+)\s*(\w*)", s)
        if m:
            marks.append((m.group(1), m.group(2)))
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            if cur or marks:
                regs.append((cur, marks))
            cur, marks = collections.Counter(), []
            continue
        if not is_instr(s):
            continue
        op = s.split()[0]
        cur[op] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            regs.append((cur, marks))
            cur, marks = collections.Counter(), []
    if cur or marks:
        regs.append((cur, marks))
    out = collections.defaultdict(collections.Counter)
    return regs, out


def tally(body):
    """A path's tag carries over region boundaries until its "rare" mark (the exceptional block starts there) and
    is picked up again by the region that holds its "resume" mark; the region holding "end" closes it."""
    header = next(i for i, ln in enumerate(body) if "This Inner Loop Header" in ln)
    out = collections.defaultdict(collections.Counter)
    # pass 0: the exceptional-case code is skipped as a whole -- the stretch that the closest `s_cbranch_execz` above a
    # "rare" mark jumps over.  (A mark drifts inside its basic block: it may sit at the END of the exact-zero test it
    # announces; until round 4 the 171 multiplier instructions of that test were counted into the affine start.)
    label_at = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln.strip())
        if m:
            label_at[m.group(1)] = i
    rare_line = [False] * len(body)
    for r, ln in enumerate(body):
        if "MSM_MARK rare" not in ln:
            continue
        for i in range(r, -1, -1):
            m = re.match(r"^\s*s_cbranch_execz\s+(\.LBB\d+_\d+)", body[i])
            if m and label_at.get(m.group(1), -1) > r:
                for k in range(i + 1, label_at[m.group(1)]):
                    rare_line[k] = True
                break
    # pass 1: regions as lists of (kind, payload) events
    regs, cur = [], []
    for off, ln in enumerate(body[header:]):
        if rare_line[header + off]:
            continue
        s = ln.strip()
        m = re.match(r"^; MSM_MARK (\w+)\s*(\w*)", s)
        if m:
            cur.append(("mark", (m.group(1), m.group(2))))
        elif re.match(r"^\.LBB\d+_\d+:", s):
            if cur:
                regs.append(cur)
            cur = []
        elif is_instr(s):
            op = s.split()[0]
            cur.append(("op", op))
            if op.startswith("s_cbranch") or op == "s_branch":
                regs.append(cur)
                cur = []
    if cur:
        regs.append(cur)
    tag, seen_mark = "hoisted", False
    for reg in regs:
        marks = [p for k, p in reg if k == "mark"]
        whole = next((n for k, n in marks if k in ("begin", "resume")), None)
        if whole:
            tag = whole            # arithmetic floats across a mark inside a block: the whole block is the path's
        for k, p in reg:
            if k == "op":
                if tag:
                    out[tag][p] += 1
            elif p[0] == "rare":
                tag = None
        if any(k in ("end", "resume") for k, _n in marks):   # the path's last block, or its one-block main part
            tag = None
        seen_mark = seen_mark or bool(marks)
        if tag == "hoisted" and seen_mark:
            tag = None
    return out


def summarise(c):
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    mult = sum(c[k] for k in MULT)
    return {"valu": valu, "multiplier": mult, "v_mad_u64_u32": c["v_mad_u64_u32"], "v_mul_lo_u32": c["v_mul_lo_u32"],
            "other_valu": valu - mult,
            "top_other": dict(sorted(((k, v) for k, v in c.items() if k.startswith("v_") and k not in MULT),
                                     key=lambda kv: -kv[1])[:14])}


def main(src, dst):
    lines = open(src).read().splitlines()
    out = {"source": "compiler assembly of k_accumulate.hip (hipcc --cuda-device-only -S, flags of the shipped object)",
           "multiplier_instructions": list(MULT), "kernels": {}}
    for name, body in functions(lines):
        if "accumulate_kernel" not in name or "accumulate_kernel_park" in name or "accumulate_kernel_asm" in name:
            continue   # (the experimental builds with other code shapes are not tallied)
        lean = "accumulate_kernel_lean" in name
        if lean:
            variant = "lean_4_waves"
        elif "accumulate_kernelILb1E" in name:
            variant = "low_occupancy_2_waves"       # the shipped kernel
        elif "accumulate_kernelILb0E" in name:
            variant = "without_register_pin"
        else:
            variant = re.sub(r"^_ZN7msm_amd\d+", "", name)[:40]   # experiments build: keyed by symbol
        try:
            pc = tally(body)
        except StopIteration:
            continue
        hoisted = collections.Counter({k: v for k, v in pc["hoisted"].items()})
        hoisted_mult = sum(hoisted[k] for k in MULT)
        madd = pc["mixed_addition"] + (hoisted if hoisted_mult else collections.Counter())
        mmadd = pc["affine_start"]
        ms, as_ = summarise(madd), summarise(mmadd)
        # 8M + 2S with one shared reduction: 6 x 162 + 2 x 126 + 243 products + 9 x 9 v_mul_lo = 1548; 4M + 2S: 864
        if lean:
            # the exact-zero test's multiplication floats above its "rare" mark in this build: 171 of the count
            # belong to the exceptional block
            pass
        elif ms["multiplier"] and not 1400 <= ms["multiplier"] <= 1700:
            raise SystemExit(f"{name}: mixed addition counts {ms['multiplier']} multiplier instructions, expected ~1548")
        if not lean and as_["multiplier"] and not 780 <= as_["multiplier"] <= 1100:
            raise SystemExit(f"{name}: affine start counts {as_['multiplier']} multiplier instructions, expected ~864")
        out["kernels"][variant] = {"symbol": name, "mixed_addition": ms, "affine_start": as_,
                                   "hoisted_above_the_path_split": summarise(hoisted)}
    if "low_occupancy_2_waves" not in out["kernels"]:
        raise SystemExit("accumulate_kernel<true> not found")
    k = out["kernels"]["low_occupancy_2_waves"]
    out["multiplier_per_mixed_addition"] = k["mixed_addition"]["multiplier"]
    out["multiplier_per_affine_start"] = k["affine_start"]["multiplier"]
    out["valu_per_mixed_addition"] = k["mixed_addition"]["valu"]
    out["valu_per_affine_start"] = k["affine_start"]["valu"]
    json.dump(out, open(dst, "w"), indent=1)
    print(f"isa_counts: mixed addition {k['mixed_addition']['multiplier']} multiplier / {k['mixed_addition']['valu']} VALU "
          f"instructions, affine start {k['affine_start']['multiplier']} / {k['affine_start']['valu']} -> {dst}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
