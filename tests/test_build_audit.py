"""CPU-side audit of the emitted gfx950 ISA (hipcc cross-compiles without a GPU).

The persistent bf16 training kernel owns all 256 AGPRs by hand (sixteen literal accumulator tiles,
reni_amd/csrc/reni_dev_common.inc: mfma_bf16_agpr_tile).  That is only sound if NO compiler-generated
instruction of that kernel touches an AGPR and the kernel does not spill.  Both are checked here on the
assembly hipcc emits, so an edit that breaks the invariant fails the CPU suite."""
import os
import re
import subprocess
import tempfile

from tests import isa_audit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CSRC = os.path.join(ROOT, "reni_amd", "csrc")
TUS = ("core", "main_f32", "main_bf16", "film_f32", "film_bf16", "train_film", "wide")
_ISA = {}


def _isa(tu):
    """gfx950 assembly of one translation unit, with build.sh's flags (all five are compiled in parallel once)."""
    if not _ISA:
        with tempfile.TemporaryDirectory() as d:
            procs = [subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                       "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-I" + os.path.join(ROOT, "include"),
                                       "-I" + CSRC, os.path.join(CSRC, f"reni_tu_{t}.hip"), "-o", os.path.join(d, t + ".s")],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE) for t in TUS]
            for t, pr in zip(TUS, procs):
                _, err = pr.communicate()
                assert pr.returncode == 0, err.decode()[-2000:]
                _ISA[t] = open(os.path.join(d, t + ".s")).read()
    return _ISA[tu]


def _train_kernel_counts(text, mangled_prefix):
    fn = [x for x in re.split(r"\n\s*\.globl\s+", text) if x.startswith(mangled_prefix)]
    assert len(fn) == 1, mangled_prefix
    in_asm, touching, outside, scratch, mfma = False, 0, 0, 0, 0
    for line in fn[0].split("\n"):
        if "ASMSTART" in line:
            in_asm = True
        elif "ASMEND" in line:
            in_asm = False
        else:
            code = line.split(";")[0]
            if "v_mfma" in code:
                mfma += 1
                assert in_asm, "compiler-generated MFMA in the training kernel: " + code
            if "v_accvgpr" in code or re.search(r"\ba\[\d", code) or re.search(r"\ba\d+\b", code):
                touching += 1
                outside += 0 if in_asm else 1
            if "scratch_" in code:
                scratch += 1
    return mfma, touching, outside, scratch


def test_training_kernel_agprs_only_in_hand_written_asm():
    text = _isa("core")
    # k_reni_train_bf16<128, true>: the training instance owns the AGPRs by hand
    # (... ELb0EEE: the generic instance; ... ELb1EEE: SPEC -- linear head, tanh, WeightedMSE as compile-time constants: what config 2 runs)
    # (template <H, DW, STATS, FILM, SPEC, L0X>; ... ELb1ELb1EEE: SPEC + L0X, round 5 -- the tile ends with layer 2's step, what config 2 runs now)
    for inst in ("_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb0ELb0ELb0EEE", "_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb0ELb1ELb0EEE",
                 "_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb0ELb1ELb1EEE"):
        mfma, touching, outside, scratch = _train_kernel_counts(text, inst)
        assert mfma > 100 and touching >= 512
        assert outside == 0, f"{outside} compiler-generated instructions touch AGPRs"
        assert scratch == 0, f"{scratch} scratch (spill) instructions in the training kernel"
    # k_reni_train_bf16<128, false>: the frozen-decoder instance has no weight-gradient accumulators at all
    mfma, touching, outside, scratch = _train_kernel_counts(text, "_ZN4reni17k_reni_train_bf16ILi128ELb0ELb0ELb0ELb0ELb0EEE")
    assert mfma > 50 and touching == 0 and scratch == 0
    # k_reni_train_bf16<128, false, true>: the forward-only statistics instance
    mfma, touching, outside, scratch = _train_kernel_counts(text, "_ZN4reni17k_reni_train_bf16ILi128ELb0ELb1ELb0ELb0ELb0EEE")
    assert mfma > 20 and touching == 0 and scratch == 0
    # k_reni_l0_ring (two waves per SIMD: 256 registers each): no spill either -- a scratch access is a vector-memory operation, and
    # the kernel's LDS-DMA ring counts its own
    fn = [x for x in re.split(r"\n\s*\.globl\s+", text) if x.startswith("_ZN4reni14k_reni_l0_ringILi128EEE")]
    assert len(fn) == 1 and "scratch_" not in fn[0] and fn[0].count("v_mfma") >= 40
    # the FiLM instances (reni_tu_train_film.hip): same rules
    text = _isa("train_film")
    for inst in ("_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb1ELb0ELb0EEE", "_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb1ELb1ELb0EEE"):  # generic, SPEC
        mfma, touching, outside, scratch = _train_kernel_counts(text, inst)
        assert mfma > 100 and touching >= 512 and outside == 0 and scratch == 0
    mfma, touching, outside, scratch = _train_kernel_counts(text, "_ZN4reni17k_reni_train_bf16ILi128ELb0ELb1ELb1ELb0ELb0EEE")
    assert mfma > 20 and touching == 0 and scratch == 0


def test_mfma_results_are_never_read_before_their_write_back():
    """Every MFMA of every kernel: strict wait-state count (taken branches count 0) to the first
    non-chained access of its result reaches the XDL write-back latency.  Guards against the hipcc
    hazard-padding defect that produced nondeterministic H = 256 head gradients (tests/isa_audit.py)."""
    kernels, bad = set(), []
    for t in TUS:  # per translation unit: local labels (.LBBn_m) repeat across units
        text = _isa(t)
        kernels |= {f for f, _ in isa_audit.audit(text)} | isa_audit.mfma_functions(text)
        bad += isa_audit.violations(text)
        early = isa_audit.valu_to_mfma(text)
        assert not early, f"VALU write within 2 wait states of an MFMA reading it ({t}): " + "; ".join(
            f"{f[:40]} asm lines {a}->{b} ({st} states)" for f, a, b, st in early[:8])
        # the hazard hipcc does not see across an asm statement's edge -- a transcendental's result read by the next
        # VALU (round 5: an asm v_cvt_pk_bf16_f32 straight behind v_sin_f32 in k_reni_l0_ring gave run-to-run different bits) -- on EVERY
        # instance of every translation unit: the sparse / compact frozen instances, the FiLM instances, k_reni_wide256<0/1/2>,
        # k_reni_l0_ring, the generic kernels; and no asm SDWA instruction that writes part of a dword (dst-forwarding hazard)
        tv = isa_audit.trans_to_valu(text)
        assert not tv, f"transcendental result read by the next VALU without a wait state ({t}): " + "; ".join(
            f"{f[:48]} asm lines {a}->{b}" for f, a, b, st in tv[:8])
        sd = isa_audit.sdwa_partial_dst(text)
        assert not sd, f"asm SDWA instruction with a partial destination ({t}): " + "; ".join(f"{f[:40]} line {a}: {c}" for f, a, c in sd[:8])
    assert any("k_reni_train_bf16" in f for f in kernels) and any("k_reni_main" in f for f in kernels)
    # (the instances the round-4 intermittent failure could have come from are all among the audited functions)
    for need in ("k_reni_l0_ring", "k_reni_wide256ILi0", "k_reni_wide256ILi1", "k_reni_wide256ILi2", "k_reni_train_bf16ILi128ELb0ELb0ELb0",
                 "k_reni_train_bf16ILi128ELb0ELb1ELb0", "k_reni_train_bf16ILi128ELb1ELb0ELb1", "k_reni_dw1_ring", "k_dw_frag"):   # (k_wide_head_dw: no MFMA -- audited by trans_to_valu all the same)
        assert any(need in f for f in kernels), need
    assert sum("k_reni_main" in f for f in kernels) >= 40  # 2 precisions x 4 widths x 3 modes x {concat, FiLM}, minus MFMA-free ones
    assert not bad, "MFMA result accessed too early: " + "; ".join(
        f"{f[:48]} {k} {st} states (asm lines {a}->{b})" for f, k, st, a, b in bad[:8])


def test_release_library_has_no_result_changing_debug_knobs():
    """The ablation mask (RENI_DEBUG_MASK: skips weight-gradient phases / LDS staging, i.e. WRONG results by design) and
    the cycle-trace pointer exist only in -DRENI_DEBUG / -DRENI_TRACE builds: the shipped library must not even contain
    the environment variables' names, so a stray variable cannot change what it computes."""
    so = os.path.join(ROOT, "reni_amd", "lib", "libreni_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    blob = open(so, "rb").read()
    for name in (b"RENI_DEBUG_MASK", b"RENI_TRACE_PTR"):
        assert name not in blob, name
    # the two path selectors that remain pick between paths that are both correct (and both tested)
    assert b"RENI_NO_PERSIST" in blob and b"RENI_NO_SIDE_STREAM" in blob


def _vm_ops_before(lines, site, need):
    """The `need` newest vector-memory operations in the text in front of line `site` (newest first).  hipcc places out-of-line blocks
    (e.g. a lane-conditional LDS-DMA issued early in the step, which jumps back to where it came from) between a kernel's last stores
    and the wait behind them: a block that cannot be fallen into, and that ends in an unconditional branch BACK to a label in front of
    every operation counted here, cannot run between those operations and the wait -- its operations are left out."""
    code = [l.split(";")[0].strip() for l in lines]
    label_at = {t[:-1]: i for i, t in enumerate(code) if re.match(r"\.LBB\d+_\d+:$", t)}
    uncond = ("s_branch", "s_endpgm", "s_setpc")
    skip_to, back_target = {}, {}   # last line of an out-of-line block -> its first line / the line its closing branch targets
    for lab, i in label_at.items():
        prev = next((code[q] for q in range(i - 1, 0, -1) if code[q]), "")
        if not prev.startswith(uncond):
            continue
        j = next((q for q in range(i, len(code)) if code[q].startswith(uncond)), None)
        if j is None or not code[j].startswith("s_branch"):
            continue
        tgt = label_at.get(code[j].split()[1])
        if tgt is not None and tgt < i and j < site:
            skip_to[j], back_target[j] = i, tgt
    vm, k, latest_target = [], site - 1, None
    while k > 0 and len(vm) < need:
        if k in skip_to:
            latest_target = back_target[k] if latest_target is None else max(latest_target, back_target[k])
            k = skip_to[k] - 1
            continue
        t = code[k]
        if t.startswith(("global_", "scratch_", "buffer_", "flat_")):
            vm.append(t.split()[0])
        k -= 1
    assert latest_target is None or latest_target < k + 1, "an out-of-line block returns to a point behind the counted operations"
    return vm


def test_counted_wait_in_front_of_the_dA_phase_covers_the_weight_image():
    """k_reni_train_bf16<128,true> waits `vmcnt(8)` (not 0) for the next tile's first weight image in front of the layer-0
    dA phase: legal only if the eight newest vector-memory operations at that point are the g_1 stream's stores and every
    LDS-DMA piece of the image is older.  Checked on every asm `s_waitcnt vmcnt(8)` of the emitted kernel."""
    text = _isa("core")
    # (the L0X instance: the same wait at the end of the tile -- its eight g_1 stores leave from layer 2's stream, behind the slots that
    # carry the image's LDS-DMA groups)
    for inst in ("_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb0ELb0ELb0EEE", "_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb0ELb1ELb0EEE",
                 "_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0ELb0ELb1ELb1EEE"):
        fn = [x for x in re.split(r"\n\s*\.globl\s+", text) if x.startswith(inst)][0]
        lines = [l.strip() for l in fn.split("\n")]
        sites = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(8)") and "ASMSTART" in lines[i - 1]]
        assert sites, "the counted wait is gone: update this audit with the code"
        for i in sites:
            vm = _vm_ops_before(lines, i - 1, 17)
            assert vm[:8] == ["global_store_dwordx4"] * 8, (inst[-24:], vm[:10])
            assert vm[8:17] == ["global_load_lds_dwordx4"] * 9, (inst[-24:], vm[8:17])
