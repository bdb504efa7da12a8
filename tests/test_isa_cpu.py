"""Properties of the COMPILED kernels that the r03 performance work depends on, checked on the ISA the cross-compiler emits
(hipcc runs without a GPU).  Each of them was found by reading `hipcc -S` output, cost measurable time when it was violated,
and is invisible to every numerical test:

* flow.hip / flow_train.hip carry no packed-f32 VALU (v_pk_fma/add/mul_f32: +16 cycles each beside MFMAs on gfx950 -- the SLP
  vectoriser creates them under plain -O3, hence -fno-slp-vectorize in the Makefile) and no scratch memory (a run-time index
  into a small array had put `tbwd2`'s per-point values there);
* the training kernels stage their weights through registers, not LDS-DMA (the compiler turns every wait behind a
  global_load_lds into vmcnt(0) and treats every later LDS access as its alias);
* the skewed eval kernel's layer loop contains an explicit `s_waitcnt vmcnt(0)` (the compiler emits none at its barriers:
  a landed weight DMA used to be a matter of timing).
Skipped when hipcc is absent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dpf_nets_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _isa(tmp_path_factory, name):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-fno-slp-vectorize",
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, name + ".hip")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


@pytest.fixture(scope="module")
def isa_flow_train(tmp_path_factory):
    return _isa(tmp_path_factory, "flow_train")


@pytest.fixture(scope="module")
def isa_flow(tmp_path_factory):
    return _isa(tmp_path_factory, "flow")


@pytest.fixture(scope="module")
def isa_flow16(tmp_path_factory):
    return _isa(tmp_path_factory, "flow16")


def _kernels(isa):
    """{mangled name: body} of every kernel of an assembly listing"""
    parts = re.split(r"^(_Z[A-Za-z0-9_]+):[^\n]*$", isa, flags=re.M)
    # (up to the function's end label, not its first s_endpgm: a kernel whose workgroups take roles -- tbwd1's column sums --
    # has an early exit)
    return {parts[i]: re.split(r"^\.Lfunc_end\d+:", parts[i + 1], flags=re.M)[0] for i in range(1, len(parts) - 1, 2)}


HOT_EVAL = "flow_kernelILi2ELi8ELi1ELb1ELb1ELb1E"          # NS = 2, 8 waves, LPB = 1, pipelined, fp16 operands, skewed ring
HOT_TRAIN = ("tstats_h1_kernel", "tbwd1_kernel", "tbwd2_kernel")


def test_makefile_compiles_every_object_without_the_slp_vectoriser():
    """r06: the vectoriser's packed fp32 code is what made the approx-EMD passes return run-to-run differing bits;
    since r06 it is off for the whole library (FLAGS), not only for the flow stack (FLOWFLAGS, r03: speed)."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("FLAGS :=")]
    assert len(flags) == 1 and "-fno-slp-vectorize" in flags[0], flags
    for ln in mk.splitlines():                      # every compile line goes through $(FLAGS)
        if "$(HIPCC)" in ln and " -c " in ln:
            assert "$(FLAGS)" in ln, ln


def test_makefile_compiles_the_flow_stack_without_the_slp_vectoriser():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for target in ("flow.o", "flow_train.o"):
        rule = mk[mk.index(target + ":"):]
        assert "$(FLOWFLAGS)" in rule.split("\n\n")[0].split("\n", 2)[1], target
    assert "FLOWFLAGS := -fno-slp-vectorize" in mk


def test_no_packed_f32_and_no_scratch_in_the_flow_kernels(isa_flow, isa_flow_train):
    hot = {n: b for n, b in _kernels(isa_flow).items() if HOT_EVAL in n}
    hot.update({n: b for n, b in _kernels(isa_flow_train).items() if any(k in n for k in HOT_TRAIN)})
    assert len(hot) >= 2 + 9, sorted(hot)                      # direct / inverse of the eval kernel, 3 precisions of 3 training kernels
    for name, body in hot.items():
        assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", body), name
        assert "scratch_" not in body, name
    for name, isa in (("flow.hip", isa_flow), ("flow_train.hip", isa_flow_train)):
        assert "scratch_" not in isa, name
        spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", isa)]
        assert spills and max(spills) == 0, (name, spills)


def test_tile16_kernels_keep_their_registers(isa_flow16):
    """r04: the 16-point-tile kernels hold a layer's accumulators, the ring's addresses and the next head in registers; the
    first versions spilled (an `if` around the head prefetch kept two copies alive; ~100 hoisted LDS addresses) -- the
    launder-the-base idiom of flow16.hip is what keeps them under the cap, and nothing numerical would notice its loss."""
    ks = {n: b for n, b in _kernels(isa_flow16).items() if "flow16_kernel" in n or "flow16s_kernel" in n}
    assert len(ks) >= 4, sorted(ks)
    for name, body in ks.items():
        assert "scratch_" not in body, name
        assert "v_mfma_f32_16x16x32_f16" in body, name
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", isa_flow16)]
    assert spills and max(spills) == 0, spills


def test_training_kernels_stage_weights_through_registers(isa_flow_train):
    assert "global_load_lds" not in isa_flow_train
    ks = _kernels(isa_flow_train)
    for key in HOT_TRAIN:
        bodies = [b for n, b in ks.items() if key in n]
        assert bodies, key
        for b in bodies:
            # the prologue waits for its loads one group at a time: partial vmcnt waits, not only vmcnt(0)
            partial = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", b) if int(x) > 0]
            assert len(partial) >= 3, (key, partial)


def test_skewed_eval_kernel_waits_for_its_weight_dma_inside_the_layer_loop(isa_flow):
    ks = _kernels(isa_flow)
    skew = [b for n, b in ks.items() if HOT_EVAL in n]
    assert skew
    for b in skew:
        lines = b.split("\n")
        dma = [i for i, l in enumerate(lines) if "global_load_lds" in l]
        waits = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt vmcnt\(0\)\s*$", l)]
        # the DMA of the loop's bottom (after the prologue's two layers) and, before it in program order, the explicit wait
        loop_dma = [i for i in dma if i > len(lines) // 2]
        assert loop_dma, "no in-loop DMA found"
        assert any(len(lines) // 4 < w < loop_dma[0] for w in waits), "no vmcnt(0) between the loop's top and its DMA issue"


def test_emd_matrix_passes_keep_the_mfma_destination_apart(tmp_path_factory):
    """csrc/emd.hip, pair_exponents (r05): with a literal-zero C operand the compiler gives the MFMA's destination the registers
    of a dying source, and in the builds where it did a pass returned sums that differed from run to run on MI355X; with C in
    registers (an opaque zero) it uses the early-clobber form.  Checked on the ISA: every MFMA of emd.hip has a register C and
    a destination apart from both sources (tools/mfma_overlap_check.py)."""
    import sys
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "emd.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-ffp-contract=off",
           "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "emd.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_overlap_check as chk
    total, bad = chk.scan(str(out), require_register_c=True, no_scratch="emd_mfma_", no_packed="emd_mfma_")     # (what csrc/Makefile gates emd.o on)
    assert total >= 20 and not bad, (total, bad[:3])
    text = out.read_text()
    for ln in text.splitlines():
        if "v_mfma_" in ln:
            assert re.search(r",\s*[va]\[\d+:\d+\]\s*$", ln), ln       # C is a register tuple, not the literal 0
    mk = open(os.path.join(CSRC, "Makefile")).read()
    rule = mk.split("emd.o:")[1].split("\n\n")[0]
    assert "mfma_overlap_check.py --require-register-c --no-scratch emd_mfma_ --no-packed-f32 emd_mfma_ " in rule and rule.count(" emd.s") >= 2
    assert rule.count("-fno-slp-vectorize") == 2              # the gated listing and the object are the same build
    # ... and none of the matrix-core kernels spills (r05: a build forced to 128 registers -- 40 to 123 spilled -- failed the
    # parity and repeat tests on the GPU, besides being 2 - 6 x slower)
    for name, body in _kernels(text).items():
        if "emd_mfma_" in name:
            assert "scratch_" not in body, name


# ---- r06: the packed-fp32 rules of every object ---------------------------------------------------------------
def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_overlap_check", os.path.join(ROOT, "tools", "mfma_overlap_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_checker_knows_the_affected_instruction_forms(tmp_path):
    """tools/ubench/pk_vs_mfma_forms.hip on MI355X: of eight forms only the one whose LOW half reads the HIGH word of a VGPR pair
    loses results beside another wave's MFMAs.  The checker must flag exactly that form, a packed instruction directly in front of
    an MFMA, and a kernel that holds both kinds."""
    chk = _checker()
    text = """
kern_a: ; @kern_a
\tv_pk_fma_f32 v[132:133], v[172:173], v[112:113], v[132:133] op_sel:[0,1,0]
\tv_mfma_f32_32x32x16_f16 v[34:49], v[142:145], v[78:81], v[34:49]
\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel_hi:[1,0,1]
\ts_nop 0
\tv_mfma_f32_32x32x16_f16 v[34:49], v[142:145], v[78:81], v[34:49]
.Lfunc_end0:
kern_b: ; @kern_b
\tv_pk_add_f32 v[2:3], s[4:5], v[6:7] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]
\tv_pk_mul_f32 v[2:3], v[2:3], v[6:7]
\tv_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]
.Lfunc_end1:
"""
    p = tmp_path / "t.s"
    p.write_text(text)
    low = chk.scan_packed_low_from_high(str(p))
    assert [(k, l.split()[0], l.split("op_sel:")[1][:7]) for k, l in low] == [("kern_a", "v_pk_fma_f32", "[0,1,0]"), ("kern_b", "v_pk_mul_f32", "[0,1]")]
    pairs = chk.scan_packed_before_mfma(str(p))
    assert len(pairs) == 1 and pairs[0][0] == "kern_a" and "op_sel:[0,1,0]" in pairs[0][1]
    both = chk.scan_packed_with_mfma(str(p))
    assert both == [("kern_a", 2, 2)]
    assert chk.main(["--ignore-overlap", "--no-packed-with-mfma", "--no-packed-before-mfma", "--no-packed-low-from-high", str(p)]) == 1


def test_makefile_gates_every_object_on_the_packed_fp32_rules():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    gate = [ln for ln in mk.splitlines() if ln.startswith("GATE :=")]
    assert len(gate) == 1 and all(o in gate[0] for o in ("--no-packed-with-mfma", "--no-packed-before-mfma", "--no-packed-low-from-high")), gate
    rules = re.findall(r"^([\w%]+\.o): .*\n((?:\t.*\n)+)", mk, flags=re.M)
    objs = re.search(r"^OBJS := (.*)$", mk, flags=re.M).group(1).split()
    named = {t for t, _ in rules}
    for t, body in rules:
        if t.endswith(("_prof.o", "_ab.o")) or t not in set(objs) | {"%.o"}:
            continue                                    # (profiling / ablation builds are scratch: nothing but libdpf_hip.so ships)
        if t == "emd.o":
            assert all(o in body for o in ("--no-packed-with-mfma", "--no-packed-before-mfma", "--no-packed-low-from-high")), body
        else:
            assert "$(call isa_gate," in body, (t, body)
    assert "%.o" in named and "emd.o" in named


def test_the_built_objects_hold_no_affected_packed_form():
    """the assembly listings `make` leaves beside the objects (csrc/*.s; __graft_entry__.build() has run make) through the same gate"""
    chk = _checker()
    listings = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".s")]
    fresh = [p for p in listings if os.path.exists(p[:-2] + ".hip") and os.path.getmtime(p) >= os.path.getmtime(p[:-2] + ".hip")]
    if len(fresh) < 10:
        pytest.skip("no fresh listings beside the objects (run make -C dpf_nets_amd/csrc)")
    for p in fresh:
        assert chk.scan_packed_low_from_high(p) == [], p
        assert chk.scan_packed_before_mfma(p) == [], p
        assert chk.scan_packed_with_mfma(p) == [], p
