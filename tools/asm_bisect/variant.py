"""r06: instruction-level bisect of the run-to-run differing bits of the matrix-core approx-EMD passes.

The library built with the SLP vectoriser ON flickers, the same source built without it does not (profiles/r06_emd_bisect.txt: first at
`emd_mfma_rows_kernel<4>`, level 6).  This tool takes the DEVICE ASSEMBLY of the flickering build, edits the instruction stream of
one kernel, and carries the result through assembler -> lld -> offload bundle -> host object -> variants/libdpf_<name>.so, so that
single instructions can be exchanged while everything else (register allocation, schedule, every other kernel) stays as it was.

    python tools/asm_bisect/variant.py <name> [<name> ...]      (names: see VARIANTS; `all` builds every one)

On the GPU box: tools/asm_bisect/run.sh  (copies each variant over dpf_nets_amd/libdpf_hip.so in the box's scratch copy and runs
tests/diag/emd_bisect.py on the reproducer).  Scratch tooling: nothing here is part of the product path."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dpf_nets_amd", "csrc")
OUT = os.path.join(ROOT, "variants")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]
ROWS4 = "emd_mfma_rows_kernelILi4E"
COLS0 = "emd_mfma_cols_kernelILi0E"
MFMA = re.compile(r"^\tv_mfma_f32_32x32x16_f16 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]")
VWRITE = re.compile(r"^\t(v_[a-z0-9_]+) v(\d+)(?:,|\s)")

PK = re.compile(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[1,0,1\]\s*$")


def run(cmd, **kw):
    subprocess.run(cmd, check=True, **kw)


def device_asm(slp=True, extra=()):
    tag = ("slp" if slp else "noslp") + "".join(e.replace("=", "_").replace("-", "") for e in extra)
    path = os.path.join(OUT, "emd_%s.s" % tag)
    fl = FLAGS + ([] if slp else ["-fno-slp-vectorize"]) + list(extra)
    run(["/opt/rocm/bin/hipcc"] + fl + ["-S", "--cuda-device-only", os.path.join(CSRC, "emd.hip"), "-o", path], stderr=subprocess.DEVNULL)
    return open(path).read().split("\n")


def kernel_span(lines, key):
    """[first, last) line numbers of the function whose symbol contains key."""
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_Z") and key in ln and ln.rstrip().endswith(key_tail(ln)))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return start, end


def key_tail(ln):
    return ln.rstrip()[-1]


def unpack_fma(ln):
    """v_pk_fma_f32 D, S0, S1, D op_sel_hi:[1,0,1]  ->  the two v_fma_f32 it stands for (S1's LOW word multiplies both halves)."""
    m = PK.match(ln)
    if not m:
        return None
    d0, d1, a0, a1, b0, _b1, c0, c1 = (int(v) for v in m.groups())
    return ["\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, a0, b0, c0), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d1, a1, b0, c1)]


def edit(lines, key, fn):
    a, b = kernel_span(lines, key)
    body, n = [], 0
    for i in range(a, b):
        r = fn(lines, i)
        if r is None:
            body.append(lines[i])
        else:
            body.extend(r)
            n += 1
    return lines[:a] + body + lines[b:], n


def v_unpack_all(lines, i):
    return unpack_fma(lines[i])


def v_nop_before_pk(k):
    def fn(lines, i):
        return ["\ts_nop %d" % k, lines[i]] if PK.match(lines[i]) else None
    return fn


def v_nop_after_pk(k):
    def fn(lines, i):
        return [lines[i], "\ts_nop %d" % k] if PK.match(lines[i]) else None
    return fn


def v_nop_first_of_group(k):
    """a wait only in front of the FIRST packed fma behind a run of v_exp_f32 (the transcendental -> packed use distance)"""
    def fn(lines, i):
        if PK.match(lines[i]) and "v_exp_f32" in lines[i - 1]:
            return ["\ts_nop %d" % k, lines[i]]
        return None
    return fn


def v_nop_before_exp_after_pk(k):
    """a wait in front of the first v_exp_f32 behind a packed fma (the packed read -> transcendental overwrite distance)"""
    def fn(lines, i):
        if "v_exp_f32" in lines[i] and PK.match(lines[i - 1]):
            return ["\ts_nop %d" % k, lines[i]]
        return None
    return fn


def v_unpack_subset(pred):
    count = [0]

    def fn(lines, i):
        if PK.match(lines[i]):
            count[0] += 1
            if pred(count[0] - 1):
                return unpack_fma(lines[i])
        return None
    return fn


def v_nop_behind_mfma_source_overwrite(k, window=3):
    """s_nop k in front of a VALU instruction that overwrites an A / B source register of an MFMA issued at most `window` lines
    before it (the compiler counts the register dead once its last reader has ISSUED)"""
    def fn(lines, i):
        m = VWRITE.match(lines[i])
        if not m or m.group(1).startswith("v_mfma"):
            return None
        reg = int(m.group(2))
        for j in range(max(0, i - window), i):
            mm = MFMA.match(lines[j])
            if mm:
                a0, a1, b0, b1 = (int(mm.group(g)) for g in (3, 4, 5, 6))
                if a0 <= reg <= a1 or b0 <= reg <= b1:
                    return ["\ts_nop %d" % k, lines[i]]
        return None
    return fn


def v_unpack_any(lines, i):
    """every v_pk_fma_f32 of any op_sel form -> two v_fma_f32"""
    m = re.match(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\](.*)$", lines[i])
    if not m:
        return None
    d0, d1, a0, a1, b0, b1, c0, c1 = (int(m.group(g)) for g in range(1, 9))
    mods = m.group(9)
    sel, sel_hi = [0, 0, 0], [1, 1, 1]
    ms = re.search(r"op_sel:\[(\d),(\d),(\d)\]", mods)
    if ms:
        sel = [int(v) for v in ms.groups()]
    ms = re.search(r"op_sel_hi:\[(\d),(\d),(\d)\]", mods)
    if ms:
        sel_hi = [int(v) for v in ms.groups()]
    src = [(a0, a1), (b0, b1), (c0, c1)]
    lo = [src[u][sel[u]] for u in range(3)]
    hi = [src[u][sel_hi[u]] for u in range(3)]
    # the low result must not clobber a register the high one still reads
    if d0 in hi:
        return ["\tv_fma_f32 v%d, v%d, v%d, v%d" % (d1, hi[0], hi[1], hi[2]), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, lo[0], lo[1], lo[2])] \
            if d1 not in lo else None
    return ["\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, lo[0], lo[1], lo[2]), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d1, hi[0], hi[1], hi[2])]


def v_drain_after_mfma(which=None, nops=("s_nop 15", "s_nop 15", "s_nop 15")):
    """wait states behind MFMAs (all, or the ones whose index is in `which`): no VALU instruction issues in their shadow"""
    count = [0]

    def fn(lines, i):
        if MFMA.match(lines[i]):
            count[0] += 1
            if which is None or (count[0] - 1) in which:
                return [lines[i]] + ["\t" + n for n in nops]
        return None
    return fn


def v_unpack_indices(which):
    count = [0]

    def fn(lines, i):
        if lines[i].startswith("\tv_pk_fma_f32"):
            count[0] += 1
            if (count[0] - 1) in which:
                return v_unpack_any(lines, i)
        return None
    return fn


def v_nop_around_pk(index, before=None, after=None, after_next=None):
    """s_nop in front of / behind the index-th v_pk_fma_f32, or behind the instruction that follows it"""
    count = [0]
    hit = [-1]

    def fn(lines, i):
        if lines[i].startswith("\tv_pk_fma_f32"):
            count[0] += 1
            if count[0] - 1 == index:
                hit[0] = i
                out = ([] if before is None else ["\ts_nop %d" % before]) + [lines[i]] + ([] if after is None else ["\ts_nop %d" % after])
                return out if len(out) > 1 else None
        if after_next is not None and hit[0] >= 0 and i == hit[0] + 1:
            return [lines[i], "\ts_nop %d" % after_next]
        return None
    return fn


def v_shift_code(k):
    """k x s_nop 0 (4 bytes each) in front of the kernel's first instruction: everything behind moves by 4k bytes"""
    done = [False]

    def fn(lines, i):
        if not done[0] and lines[i].startswith("\ts_") :
            done[0] = True
            return ["\ts_nop 0"] * k + [lines[i]]
        return None
    return fn


def v_swap_mfma_behind_pk(index):
    """the MFMA that follows the index-th v_pk_fma_f32 moves behind the two v_fma_f32 after it (which do not depend on it)"""
    count = [0]
    hold = {}

    def fn(lines, i):
        if lines[i].startswith("\tv_pk_fma_f32"):
            count[0] += 1
            if count[0] - 1 == index and MFMA.match(lines[i + 1]) and lines[i + 2].startswith("\tv_fma_f32") and lines[i + 3].startswith("\tv_fma_f32"):
                hold["at"] = i
                return [lines[i], lines[i + 2], lines[i + 3], lines[i + 1]]
        if "at" in hold and hold["at"] < i <= hold["at"] + 3:
            return []
        return None
    return fn


def v_rewrite_at_pk(index, how):
    """edits around the index-th v_pk_fma_f32.  how = "swap_with_previous": ..., pk[index-1], s_nop 0, pk[index]  ->  pk[index], s_nop 0,
    pk[index-1] (both add a term into the same accumulators);  "valu_after": a harmless v_mov_b32 v140, v140 behind it;
    "nop0_after": s_nop 0 behind it"""
    count = [0]
    state = {}

    def fn(lines, i):
        if lines[i].startswith("\tv_pk_fma_f32"):
            count[0] += 1
            k = count[0] - 1
            if how == "swap_with_previous" and k == index - 1 and lines[i + 1].strip() == "s_nop 0" and lines[i + 2].startswith("\tv_pk_fma_f32"):
                state["skip"] = (i + 1, i + 2)
                count[0] += 1
                return [lines[i + 2], lines[i + 1], lines[i]]
            if k == index and how == "valu_after":
                return [lines[i], "\tv_mov_b32_e32 v140, v140"]
            if k == index and how == "nop0_after":
                return [lines[i], "\ts_nop 0"]
        if "skip" in state and i in state["skip"]:
            return []
        return None
    return fn


VARIANTS = {
    # name: (slp build?, kernel key, edit) -- None = the assembly as the compiler left it
    "slp_asis": (True, None, None),
    "slp_fix_pairs": (True, None, "fix_every_pair"),
    "slp_unpack_affected_form": (True, None, "unpack_low_from_high"),   # the vectorised build minus the ONE form: everything else packed stays       # the vectorised build + one wait state at each of its packed -> MFMA pairs
    "noslp_asis": (False, None, None),
    "unpack_all": (True, ROWS4, v_unpack_all),
    "nop7_before_pk": (True, ROWS4, v_nop_before_pk(7)),
    "nop7_after_pk": (True, ROWS4, v_nop_after_pk(7)),
    "nop7_exp_to_pk": (True, ROWS4, v_nop_first_of_group(7)),
    "nop7_pk_to_exp": (True, ROWS4, v_nop_before_exp_after_pk(7)),
    "unpack_even": (True, ROWS4, v_unpack_subset(lambda k: k % 2 == 0)),
    "unpack_first_half": (True, ROWS4, v_unpack_subset(lambda k: k < 16)),
    "unpack_second_half": (True, ROWS4, v_unpack_subset(lambda k: k >= 16)),
    # the columns kernel of pass 1 (the launch whose OUTPUT differs first inside one process: emd_flicker_values.py 5,6)
    "c0_unpack": (True, COLS0, v_unpack_any),
    "c0_nop_war": (True, COLS0, v_nop_behind_mfma_source_overwrite(7)),
    "c0_nop_war8": (True, COLS0, v_nop_behind_mfma_source_overwrite(7, 8)),
    "c0_drain_all": (True, COLS0, v_drain_after_mfma()),
    "c0_drain_744": (True, COLS0, v_drain_after_mfma({13})),      # the MFMA in front of the B tile's r = 11..13 terms
    "c0_unpack_44": (True, COLS0, v_unpack_indices({44})),
    "c0_44_nop_after": (True, COLS0, v_nop_around_pk(44, after=1)),             # pk_fma; s_nop 1; v_mfma; v_fma (reader)
    "c0_44_nop_after_mfma": (True, COLS0, v_nop_around_pk(44, after_next=1)),   # pk_fma; v_mfma; s_nop 1; v_fma (reader)
    "c0_44_nop_before": (True, COLS0, v_nop_around_pk(44, before=3)),           # pk_fma #43; s_nop 0; s_nop 3; pk_fma #44
    "c0_44_mfma_later": (True, COLS0, v_swap_mfma_behind_pk(44)),               # pk_fma #44; v_fma; v_fma; v_mfma
    "c0_44_swap_43": (True, COLS0, v_rewrite_at_pk(44, "swap_with_previous")),  # pk_fma #44; s_nop 0; pk_fma #43; v_mfma
    "c0_44_valu_after": (True, COLS0, v_rewrite_at_pk(44, "valu_after")),       # pk_fma #44; v_mov_b32 v140, v140; v_mfma
    "c0_44_nop0_after": (True, COLS0, v_rewrite_at_pk(44, "nop0_after")),       # pk_fma #44; s_nop 0; v_mfma
    "c0_shift_1": (True, COLS0, v_shift_code(1)),
    "c0_shift_2": (True, COLS0, v_shift_code(2)),
    "c0_shift_3": (True, COLS0, v_shift_code(3)),
    "c0_shift_4": (True, COLS0, v_shift_code(4)),
    "c0_shift_5": (True, COLS0, v_shift_code(5)),
    "c0_shift_6": (True, COLS0, v_shift_code(6)),
    "c0_shift_7": (True, COLS0, v_shift_code(7)),
    "c0_shift_8": (True, COLS0, v_shift_code(8)),
    "c0_shift_10": (True, COLS0, v_shift_code(10)),
    "c0_shift_12": (True, COLS0, v_shift_code(12)),
    "c0_shift_14": (True, COLS0, v_shift_code(14)),
    "c0_shift_16": (True, COLS0, v_shift_code(16)),
    "c0_unpack_43": (True, COLS0, v_unpack_indices({43})),
    "c0_unpack_not44": (True, COLS0, v_unpack_indices(set(range(61)) - {44})),
    "c0_unpack_Btile": (True, COLS0, v_unpack_indices(set(range(32, 61)))),
    "c0_unpack_Atile": (True, COLS0, v_unpack_indices(set(range(0, 32)))),
}


def fix_every_pair(lines):
    """s_nop 0 between every packed fp32 VALU instruction and an MFMA that directly follows it, in every kernel of the file"""
    out, n, prev = [], 0, None
    for ln in lines:
        body = ln.split(";")[0].strip()
        is_inst = bool(body) and not body.startswith((".", "//")) and not body.endswith(":")
        if is_inst and re.match(r"(v_mfma_|v_smfmac_)", body) and prev is not None and re.match(r"v_pk_(fma|mul|add)_f32\b", prev):
            out.append("\ts_nop 0")
            n += 1
        out.append(ln)
        if ln.startswith("_Z") and ":" in ln:
            prev = None
        elif is_inst:
            prev = body
    return out, n


def unpack_low_from_high(lines):
    """every packed fma whose low half reads the high word of a VGPR pair (op_sel with a 1) -> its two v_fma_f32, in every kernel;
    the other packed forms (op_sel_hi, plain) stay"""
    out, n = [], 0
    for i, ln in enumerate(lines):
        if ln.startswith("\tv_pk_fma_f32") and re.search(r"op_sel:\[[01,]*1[01,]*\]", ln):
            r = v_unpack_any(lines, i)
            if r is not None:
                out.extend(r)
                n += 1
                continue
        out.append(ln)
    return out, n


def build(name):
    slp, key, fn = VARIANTS[name]
    lines = device_asm(slp)
    n = 0
    if fn == "fix_every_pair":
        lines, n = fix_every_pair(lines)
    elif fn == "unpack_low_from_high":
        lines, n = unpack_low_from_high(lines)
    elif fn is not None:
        lines, n = edit(lines, key, fn)
    s = os.path.join(OUT, "emd_%s.s" % name)
    open(s, "w").write("\n".join(lines))
    o, hsaco, fb, ho = (os.path.join(OUT, "emd_%s.%s" % (name, e)) for e in ("dev.o", "out", "hipfb", "o"))
    run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, o])
    run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fb])
    run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c",
                                           os.path.join(CSRC, "emd.hip"), "-o", ho], stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "emd.o"
              and not f.endswith(("_ab.o", "_prof.o"))]
    lib = os.path.join(OUT, "libdpf_%s.so" % name)
    run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, ho] + others)
    for f in (o, hsaco, fb, ho):
        os.remove(f)
    print("%-22s %4d edits -> %s" % (name, n, os.path.relpath(lib, ROOT)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:]
    if names == ["all"]:
        names = list(VARIANTS)
    for nm in names:
        build(nm)
