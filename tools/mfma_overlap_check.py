"""Scan gfx950 assembly for MFMA instructions whose destination registers overlap those of source A or source B.

r05: such an instruction ("v_mfma_f32_32x32x16_f16 v[2:17], v[70:73], v[2:5], 0": the compiler gives the destination the registers of
a source that dies there) produced results that differed from run to run on MI355X (csrc/emd.hip, pair_exponents).  A destination
identical to source C is the normal accumulate form and is not reported.

    python tools/mfma_overlap_check.py file.s [...]     -> one line per offending instruction; exit status 1 if any
"""
import re
import sys

REG = r"([va])\[(\d+):(\d+)\]|([va])(\d+)"


def regs(tok):
    m = re.fullmatch(REG, tok.strip())
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(4), int(m.group(5)), int(m.group(5))


def overlaps(a, b):
    return a is not None and b is not None and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


def scan(path):
    bad, kernel, total = [], None, 0
    for line in open(path):
        m = re.match(r"^(\S+):\s", line)
        if m and not m.group(1).startswith("."):
            kernel = m.group(1)
        m = re.match(r"\s*(v_mfma_\S+|v_smfmac_\S+)\s+(.*)", line)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        if len(ops) < 3:
            continue
        total += 1
        dst, a, b = regs(ops[0]), regs(ops[1]), regs(ops[2])
        if overlaps(dst, a) or overlaps(dst, b):
            bad.append((kernel, line.strip()))
    return total, bad


def main(paths):
    rc = 0
    for p in paths:
        total, bad = scan(p)
        print("%s: %d MFMA instructions, %d with the destination on a source's registers" % (p, total, len(bad)))
        for k, l in bad:
            print("   %s: %s" % (k[:60], l))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
