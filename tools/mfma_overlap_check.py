"""Scan gfx950 assembly for MFMA instructions whose destination registers overlap those of source A or source B.

r05: such an instruction ("v_mfma_f32_32x32x16_f16 v[2:17], v[70:73], v[2:5], 0": the compiler gives the destination the registers of
a source that dies there) produced results that differed from run to run on MI355X (csrc/emd.hip, pair_exponents).  A destination
identical to source C is the normal accumulate form and is not reported.

    python tools/mfma_overlap_check.py [--require-register-c] [--no-scratch PREFIX] file.s [...]
        -> one line per offending instruction; exit status 1 if any
    --require-register-c   also report MFMAs whose C operand is not a register tuple (the literal-0 form is the one the compiler
                           pairs with an overlapping destination)
    --no-scratch PREFIX    also fail if a kernel whose name contains PREFIX touches scratch memory (a spilled build of the
                           approx-EMD passes flickered too)
    --no-packed-f32 PREFIX r06: also fail if a kernel whose name contains PREFIX holds v_pk_{fma,mul,add}_f32 (the SLP vectoriser's
                           packed accumulations: see the next option; the approx-EMD passes keep none at all)
    --ignore-overlap       count the overlapping destinations but do not fail on them (csrc/Makefile's gate for every object)
    --no-packed-with-mfma  r06, the isolated form of the same defect (tools/ubench/pk_vs_mfma_waves2.hip, profiles/r06_packed_f32_vs_mfma.txt):
                           v_pk_{fma,mul,add}_f32 loses the LOW half of its result in lanes 48-63 while ANOTHER wave of the SIMD
                           issues MFMAs at certain distances -- nothing in the wave's own instruction stream prevents it.  Fails if
                           any kernel of the file holds both MFMAs and packed fp32 VALU instructions
    --no-packed-low-from-high
                           r06, the narrowest statement of the defect (tools/ubench/pk_vs_mfma_forms.hip): of eight forms of
                           v_pk_{fma,mul,add}_f32 only the one whose LOW half reads the HIGH word of a VGPR source pair ("op_sel" with
                           a 1 at a VGPR operand, e.g. op_sel:[0,1,0]) loses that half in lanes 48-63 under a neighbouring wave's
                           MFMAs -- the plain forms, op_sel_hi forms and op_sel on an SGPR pair do not.  Fails on every such
                           instruction in ANY kernel, with or without MFMAs of its own: the neighbour may belong to a kernel of
                           another stream (tests/test_gpu_interference.py)
    --no-packed-before-mfma
                           r06, the instruction pair tools/asm_bisect found behind r05's / r06's run-to-run differing bits (DESIGN
                           4.6): a packed fp32 VALU instruction whose NEXT instruction is an MFMA -- in emd_mfma_cols_kernel<0> of
                           the vectorised build, "v_pk_fma_f32 v[132:133], ...; v_mfma_f32_32x32x16_f16 v[34:49], ..." lost the
                           packed result in about half of the launches; one s_nop 0, any instruction between the two, or the
                           two v_fma_f32 the packed one stands for made 48 of 48 launches repeat.  Fails on every such pair in
                           ANY kernel of the file (csrc/Makefile runs it on every object)
    --war                  r06: also report a VMEM / LDS / scratch LOAD whose destination registers are the A or B source of an
                           MFMA that was issued before it and whose result nothing has read yet (the MFMA may still be queued in
                           the matrix pipe).  An INVENTORY, not a defect list: tools/ubench/mfma_war.hip / mfma_valu_war.hip
                           show the hardware interlocks this pattern (a rejected hypothesis for r05's run-to-run
                           differences; flow.hip has thousands of such sites and has never flickered)
csrc/Makefile runs it on emd.s with the first three options before emd.o may be linked, and on EVERY object's assembly with
--no-packed-with-mfma --no-packed-before-mfma: determinism of those kernels is a property of the compiled code (ADVICE r05,
include/dpf_hip.h at dpf_emd_set_matrix_path).
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


PACKED_F32 = re.compile(r"^v_pk_(fma|mul|add)_f32\b")


def scan_packed_before_mfma(path):
    """[(kernel, packed line, mfma line)]: packed fp32 VALU instructions directly followed by an MFMA"""
    out, kernel, prev = [], None, None
    for line in open(path):
        m = re.match(r"^(\S+):\s", line)
        if m and not m.group(1).startswith("."):
            kernel, prev = m.group(1), None
            continue
        body = line.split(";")[0].strip()
        if not body or body.startswith((".", "//")) or body.endswith(":"):
            continue
        if re.match(r"(v_mfma_|v_smfmac_)", body) and prev is not None and PACKED_F32.match(prev):
            out.append((kernel, prev, body))
        prev = body
    return out


def scan_packed_low_from_high(path):
    """[(kernel, line)]: packed fp32 VALU instructions whose op_sel takes the low half's operand from the high word of a VGPR pair"""
    out, kernel = [], None
    for line in open(path):
        m = re.match(r"^(\S+):\s", line)
        if m and not m.group(1).startswith("."):
            kernel = m.group(1)
            continue
        body = line.split(";")[0].strip()
        if not PACKED_F32.match(body):
            continue
        sel = re.search(r"\bop_sel:\[([01,]+)\]", body)
        if not sel:
            continue
        bits = [int(b) for b in sel.group(1).split(",")]
        ops = [o.strip() for o in body.split(None, 1)[1].split(",")]
        srcs = ops[1:1 + len(bits)]                      # operand 0 is the destination
        for bit, src in zip(bits, srcs):
            if bit and src.startswith("v"):
                out.append((kernel, body))
                break
    return out


def scan_packed_with_mfma(path):
    """[(kernel, packed count, mfma count)]: kernels that hold both"""
    counts, kernel = {}, None
    for line in open(path):
        m = re.match(r"^(\S+):\s", line)
        if m and not m.group(1).startswith("."):
            kernel = m.group(1)
            counts[kernel] = [0, 0]
            continue
        body = line.split(";")[0].strip()
        if kernel is None or not body:
            continue
        if PACKED_F32.match(body):
            counts[kernel][0] += 1
        elif re.match(r"(v_mfma_|v_smfmac_)", body):
            counts[kernel][1] += 1
    return [(k, a, b) for k, (a, b) in counts.items() if a and b]


def scan(path, require_register_c=False, no_scratch=None, no_packed=None):
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
        elif require_register_c and regs(ops[3].split()[0] if len(ops) > 3 else "") is None:
            bad.append((kernel, line.strip() + "    ; C is not a register tuple"))
    if no_packed:
        kernel = None
        for line in open(path):
            m = re.match(r"^(\S+):\s", line)
            if m and not m.group(1).startswith("."):
                kernel = m.group(1)
            if kernel and no_packed in kernel and re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
                bad.append((kernel, line.strip() + "    ; packed fp32 VALU in a kernel that must not have any"))
    if no_scratch:
        kernel = None
        for line in open(path):
            m = re.match(r"^(\S+):\s", line)
            if m and not m.group(1).startswith("."):
                kernel = m.group(1)
            if kernel and no_scratch in kernel and re.search(r"\bscratch_(load|store)", line):
                bad.append((kernel, line.strip() + "    ; scratch in a kernel that must not spill"))
                break
    return total, bad


LOADS = re.compile(r"\s*(global_load_\S+|buffer_load_\S+|scratch_load_\S+|flat_load_\S+|ds_read\S*|ds_load\S*)\s+(.*)")
ANYREG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def scan_war(path):
    """[(kernel, load line, mfma line)]: loads into the A / B registers of an MFMA nobody has consumed yet"""
    out, kernel, pending = [], None, []
    for line in open(path):
        m = re.match(r"^(\S+):\s", line)
        if m and not m.group(1).startswith("."):
            kernel, pending = m.group(1), []
        body = line.split(";")[0]
        mm = re.match(r"\s*(v_mfma_\S+|v_smfmac_\S+)\s+(.*)", body)
        if mm:
            ops = [o.strip() for o in mm.group(2).split(",")]
            if len(ops) >= 4:
                c = regs(ops[3].split()[0])
                # an MFMA that accumulates onto an earlier result does not prove that one finished: keep both pending
                pending.append((regs(ops[0]), regs(ops[1]), regs(ops[2]), body.strip()))
            continue
        ml = LOADS.match(body)
        if ml:
            dst = regs(ml.group(2).split(",")[0].strip())
            for (d, a, b, txt) in pending:
                if overlaps(dst, a) or overlaps(dst, b):
                    out.append((kernel, body.strip(), txt))
                    break
        if not pending or not body.strip() or body.strip().startswith((".", "s_", ";")):
            continue
        # any other instruction that READS a pending MFMA's destination waits for it (and for every MFMA before it)
        toks = body.split(None, 1)
        srcs = toks[1].split(",")[1:] if len(toks) > 1 and not ml else (toks[1].split(",")[1:] if len(toks) > 1 else [])
        if toks and toks[0].startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "ds_write", "ds_store")):
            srcs = toks[1].split(",")
        used = []
        for sx in srcs:
            for t in ANYREG.finditer(sx):
                used.append((t.group(1) or t.group(4), int(t.group(2) or t.group(5)), int(t.group(3) or t.group(5))))
        last = -1
        for i, (d, a, b, txt) in enumerate(pending):
            if any(overlaps(d, u) for u in used):
                last = i
        if last >= 0:
            pending = pending[last + 1:]
    return out


def main(argv):
    rc = 0
    req = "--require-register-c" in argv
    nos = argv[argv.index("--no-scratch") + 1] if "--no-scratch" in argv else None
    nop = argv[argv.index("--no-packed-f32") + 1] if "--no-packed-f32" in argv else None
    paths = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] not in ("--no-scratch", "--no-packed-f32"))]
    for p in paths:
        total, bad = scan(p, req, nos, nop)
        if "--war" in argv:
            war = scan_war(p)
            print("%s: %d loads into the A / B registers of an MFMA still in flight" % (p, len(war)))
            for k, l, mf in war:
                print("   %s: %s    <- after %s" % ((k or "?")[:50], l, mf))
                rc = 1
        if "--no-packed-with-mfma" in argv:
            both = scan_packed_with_mfma(p)
            print("%s: %d kernels with both MFMAs and packed fp32 VALU instructions" % (p, len(both)))
            for k, a, b in both:
                print("   %s: %d packed fp32, %d MFMA" % ((k or "?")[:70], a, b))
                rc = 1
        if "--no-packed-low-from-high" in argv:
            lows = scan_packed_low_from_high(p)
            print("%s: %d packed fp32 VALU instructions whose low half reads the high word of a VGPR pair" % (p, len(lows)))
            for k, l in lows[:20]:
                print("   %s: %s" % ((k or "?")[:50], l))
            rc = rc or (1 if lows else 0)
        if "--no-packed-before-mfma" in argv:
            pairs = scan_packed_before_mfma(p)
            print("%s: %d packed fp32 VALU instructions directly in front of an MFMA" % (p, len(pairs)))
            for k, a, b in pairs:
                print("   %s: %s  ->  %s" % ((k or "?")[:50], a, b))
                rc = 1
        print("%s: %d MFMA instructions, %d with the destination on a source's registers" % (p, total, len(bad)))
        if "--ignore-overlap" in argv:          # (the library-wide gate: the packed-fp32 rules only)
            continue
        for k, l in bad:
            print("   %s: %s" % (k[:60], l))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
