"""MFMA write-back hazard audit of hipcc's gfx950 assembly (helper of test_build_audit.py).

For every `v_mfma_*` the audit walks every control-flow path that leaves it and counts the wait states
until the first instruction that touches the MFMA's destination registers and is not the next MFMA of
the same accumulate chain.  `s_nop N` counts N + 1, every other instruction 1, a TAKEN branch 0 (the
strict reading; hipcc's own hazard recogniser counts it as 1 and was seen to misplace its padding behind
a block-ending MFMA -- see PolF32::drain in reni_amd/csrc/reni_dev_common.inc).  The kernels pad their own chains, so the
strict count must still reach the XDL write-back latency.

Usage as a script:  python tests/isa_audit.py file.s
"""
import bisect
import collections
import re
import sys

# wait states an MFMA result needs before a non-chained access (MI355X ISA: passes + 4 / LLVM gfx940 tables)
NEED = {"v_mfma_f32_32x32x16_bf16": 12, "v_mfma_f32_32x32x2_f32": 18}
HORIZON = 24


def _regs(tok):
    out = set()
    for m in re.finditer(r"\b([av])\[(\d+):(\d+)\]", tok):
        out |= {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([av])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def _states(t):
    m = re.match(r"s_nop (\d+)", t)
    return int(m.group(1)) + 1 if m else 1


def parse(text):
    labels, prog, fnames = {}, [], {}
    for i, line in enumerate(text.split("\n")):
        m = re.match(r"^([.\w$]+):", line)
        if m:
            labels[m.group(1)] = len(prog)
            if m.group(1).startswith("_Z"):
                fnames[len(prog)] = m.group(1)
            continue
        t = line.strip()
        if not line.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        t = t.split(";")[0].strip()
        if t:
            prog.append((i + 1, t))
    return labels, prog, fnames


def audit(text, taken_branch_states=0):
    """-> {(function, mfma opcode): [(states, line of the MFMA, line of the access), ...]} minimal per access."""
    labels, prog, fnames = parse(text)
    fk = sorted(fnames)

    def fn(k):
        j = bisect.bisect_right(fk, k) - 1
        return fnames[fk[j]] if j >= 0 else "?"

    res = collections.defaultdict(list)
    for k, (ln, t) in enumerate(prog):
        if not t.startswith("v_mfma"):
            continue
        ops = t.split(None, 1)[1].split(",")
        dtxt = ops[0].strip()
        dest = _regs(dtxt)
        stack, best = [(k + 1, 0)], {}
        while stack:
            pc, st = stack.pop()
            while pc < len(prog) and st <= HORIZON:
                if best.get(pc, 1 << 30) <= st:
                    break
                best[pc] = st
                l2, t2 = prog[pc]
                op = t2.split()[0]
                if op == "s_endpgm":
                    break
                if op.startswith("v_mfma"):
                    o2 = t2.split(None, 1)[1].split(",")
                    chained = (o2[0].strip() == dtxt and o2[-1].strip().split()[0] == dtxt
                               and not (_regs(o2[1]) | _regs(o2[2])) & dest)
                    if chained:
                        break  # back-to-back accumulate: no wait needed, the chain restarts from that MFMA
                if op == "s_branch" or op.startswith("s_cbranch"):
                    tgt = t2.split()[1]
                    if tgt in labels:
                        stack.append((labels[tgt], st + taken_branch_states))
                    if op == "s_branch":
                        break
                    pc += 1
                    st += 1
                    continue
                if _regs(t2) & dest:
                    res[(fn(k), t.split()[0])].append((st, ln, l2))
                    break
                st += _states(t2)
                pc += 1
    return res


def mfma_functions(text):
    """-> the set of functions that contain an MFMA at all (what the audits above walk; a function whose every MFMA result is first touched
    beyond the horizon has no entry in audit()'s result -- k_dw_frag since its builder was pipelined under the MFMAs)."""
    labels, prog, fnames = parse(text)
    fk = sorted(fnames)
    out = set()
    for k, (_, t) in enumerate(prog):
        if t.startswith("v_mfma"):
            j = bisect.bisect_right(fk, k) - 1
            if j >= 0:
                out.add(fnames[fk[j]])
    return out


def valu_to_mfma(text, need=2):
    """VALU write of a VGPR -> MFMA reading it as SrcA/SrcB/SrcC needs `need` wait states.  hipcc pads its own MFMAs;
    an asm MFMA without a leading s_nop relies on nothing of the kind sitting in front of it, which is what this
    checks: -> [(function, line of the VALU, line of the MFMA, states between)] for every violation, following
    fall-through and branch predecessors."""
    labels, prog, fnames = parse(text)
    fk = sorted(fnames)

    def fn(k):
        j = bisect.bisect_right(fk, k) - 1
        return fnames[fk[j]] if j >= 0 else "?"

    label_at = collections.defaultdict(list)
    for name, pc in labels.items():
        label_at[pc].append(name)
    preds = collections.defaultdict(list)  # pc -> [pc of branches that jump here]
    for pc, (_, t) in enumerate(prog):
        op = t.split()[0]
        if op == "s_branch" or op.startswith("s_cbranch"):
            tgt = t.split()[1]
            if tgt in labels:
                preds[labels[tgt]].append(pc)
    bad = []
    for k, (ln, t) in enumerate(prog):
        if not t.startswith("v_mfma"):
            continue
        ops = t.split(None, 1)[1].split(",")
        src = _regs(ops[1]) | _regs(ops[2]) | {r for r in _regs(ops[3]) if r[0] == "v"}
        src = {r for r in src if r[0] == "v"}
        stack, seen = [(k - 1, 0)], set()
        while stack:
            pc, st = stack.pop()
            while pc >= 0 and st < need:
                if (pc, st) in seen:
                    break
                seen.add((pc, st))
                for b in preds.get(pc + 1, ()):  # instruction pc+1 is a branch target: also come from the branch
                    if b != pc:
                        stack.append((b, st))
                l2, t2 = prog[pc]
                op = t2.split()[0]
                if op == "s_branch":  # no fall-through into pc+1 from here
                    break
                if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane")):
                    dst = _regs(t2.split(None, 1)[1].split(",")[0]) if " " in t2 else set()
                    if dst & src:
                        bad.append((fn(k), l2, ln, st))
                        break
                st += _states(t2)
                pc -= 1
    return bad


# transcendental VALU opcodes of gfx950 (LLVM: SIInstrFlags::TRANS): a NON-transcendental VALU that reads such a result needs one wait
# state in between (GCNHazardRecognizer: hasTransForwardingHazard, TransDefWaitstates = 1).  hipcc pads its own code; it does not look
# into an asm statement, nor across the boundary between an asm statement and its own code -- where the kernels' sine / cosine epilogues sit.
TRANS = re.compile(r"v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)(_legacy)?_(f32|f16|bf16)(_e32|_e64|_sdwa|_dpp)?$")


def trans_to_valu(text, need=1):
    """-> [(function, line of the transcendental, line of the reader, states between)] for every non-transcendental VALU (or MFMA) that
    reads a transcendental's destination VGPR with fewer than `need` wait states between them, on any control-flow path (`s_nop N` counts
    N + 1, any other instruction 1, a taken branch 0)."""
    labels, prog, fnames = parse(text)
    fk = sorted(fnames)

    def fn(k):
        j = bisect.bisect_right(fk, k) - 1
        return fnames[fk[j]] if j >= 0 else "?"

    bad = []
    for k, (ln, t) in enumerate(prog):
        parts = t.split(None, 1)
        if not TRANS.match(parts[0]) or len(parts) < 2:
            continue
        dst = {r for r in _regs(parts[1].split(",")[0]) if r[0] == "v"}
        if not dst:
            continue
        stack, seen = [(k + 1, 0)], set()
        while stack:
            pc, st = stack.pop()
            while pc < len(prog) and st < need:
                if (pc, st) in seen:
                    break
                seen.add((pc, st))
                l2, t2 = prog[pc]
                p2 = t2.split(None, 1)
                op = p2[0]
                if op == "s_endpgm":
                    break
                if op == "s_branch" or op.startswith("s_cbranch"):
                    tgt = t2.split()[1]
                    if tgt in labels:
                        stack.append((labels[tgt], st))
                    if op == "s_branch":
                        break
                    pc += 1
                    st += 1
                    continue
                if (op.startswith("v_") and len(p2) > 1):
                    ops = p2[1].split(",")
                    srcs = set()
                    for o in ops[1:]:  # (every operand behind the destination; an MFMA: SrcA, SrcB, SrcC)
                        srcs |= _regs(o)
                    if srcs & dst and not TRANS.match(op):
                        bad.append((fn(k), ln, l2, st))
                        break
                    if _regs(ops[0]) & dst and not op.startswith("v_cmp"):
                        break  # overwritten: the hazard ends here
                st += _states(t2)
                pc += 1
    return bad


def sdwa_partial_dst(text):
    """asm SDWA / op_sel instructions that write PART of a dword (dst_sel other than DWORD): gfx950's dst-forwarding hazard then needs a
    wait state in front of a VALU reading the register, which hipcc inserts for its own code only -> [(function, line, text)] inside
    ASMSTART .. ASMEND blocks."""
    out, in_asm, cur = [], False, "?"
    for i, line in enumerate(text.split("\n")):
        m = re.match(r"^(_Z[\w$]+):", line)
        if m:
            cur = m.group(1)
        if "ASMSTART" in line:
            in_asm = True
        elif "ASMEND" in line:
            in_asm = False
        elif in_asm:
            code = line.split(";")[0]
            m = re.search(r"dst_sel:(\w+)", code)
            if m and m.group(1) != "DWORD":
                out.append((cur, i + 1, code.strip()))
    return out


def violations(text, skip=("k_probe",)):
    bad = []
    for (f, kind), v in audit(text).items():
        if any(s in f for s in skip) or kind not in NEED:
            continue
        bad += [(f, kind, st, a, b) for st, a, b in v if st < NEED[kind]]
    return bad


if __name__ == "__main__":
    txt = open(sys.argv[1]).read()
    for (f, kind), v in sorted(audit(txt).items()):
        print(f[:70].ljust(70), kind[7:], "min states", min(x[0] for x in v), "accesses", len(v))
    for b in violations(txt):
        print("VIOLATION", b)
    for b in valu_to_mfma(txt):
        print("VALU->MFMA", b)
    for b in trans_to_valu(txt):
        print("TRANS->VALU", b)
    for b in sdwa_partial_dst(txt):
        print("SDWA partial dst in asm", b)
