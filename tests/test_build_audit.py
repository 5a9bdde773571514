"""CPU-side audit of the emitted gfx950 ISA (hipcc cross-compiles without a GPU).

The persistent bf16 training kernel owns all 256 AGPRs by hand (sixteen literal accumulator tiles,
reni_amd/csrc/reni_kernels.hip: mfma_bf16_agpr_tile).  That is only sound if NO compiler-generated
instruction of that kernel touches an AGPR and the kernel does not spill.  Both are checked here on the
assembly hipcc emits, so an edit that breaks the invariant fails the CPU suite."""
import os
import re
import subprocess
import tempfile

from tests import isa_audit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_kernel_agprs_only_in_hand_written_asm():
    src = os.path.join(ROOT, "reni_amd", "csrc", "reni_kernels.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-DRENI_ONLY_TRAIN", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    fn = [x for x in re.split(r"\n\s*\.globl\s+", text) if x.startswith("_ZN4reni17k_reni_train_bf16")]
    assert len(fn) == 1
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
    assert mfma > 100 and touching >= 512
    assert outside == 0, f"{outside} compiler-generated instructions touch AGPRs"
    assert scratch == 0, f"{scratch} scratch (spill) instructions in the training kernel"


def test_mfma_results_are_never_read_before_their_write_back():
    """Every MFMA of every kernel: strict wait-state count (taken branches count 0) to the first
    non-chained access of its result reaches the XDL write-back latency.  Guards against the hipcc
    hazard-padding defect that produced nondeterministic H = 256 head gradients (tests/isa_audit.py)."""
    src = os.path.join(ROOT, "reni_amd", "csrc", "reni_kernels.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "all.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = isa_audit.audit(text)
    kernels = {f for f, _ in res}
    assert any("k_reni_train_bf16" in f for f in kernels) and any("k_reni_main" in f for f in kernels)
    bad = isa_audit.violations(text)
    assert not bad, "MFMA result accessed too early: " + "; ".join(
        f"{f[:48]} {k} {st} states (asm lines {a}->{b})" for f, k, st, a, b in bad[:8])
