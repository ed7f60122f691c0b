#!/usr/bin/env python
"""Lint for the hand-placed LDS reads of conv_board.hip (the CDNA guide's inline-asm rule "an asm load's destination counts as
written at ;;#ASMEND, so the compiler may read, copy, spill or reuse it before the data lands").  Walks one kernel of a
`hipcc -save-temps` .s file in program order, tracks the destinations of ds_read_b128 instructions that no s_waitcnt lgkmcnt(N)
has covered yet (LDS returns in order: a wait for N leaves the youngest N in flight), and reports every instruction outside
the asm statements that touches such a register.  The walk is in text order and starts afresh at every branch target.

    python tools/asm_lint.py <file.s> <kernel-name-substring>

Second check (kernels that issue MFMAs from inline asm, Split<16>): hipcc pads no hazard around an asm statement, so a VALU write
(v_mov, v_accvgpr_write / _mov, ...) of a register an asm MFMA reads must be at least two wait states ahead of it.  Reports
every asm MFMA with such a write within the two preceding instructions.
"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def any_regs(tok):
    """VGPRs as n, AGPRs as 1000 + n."""
    m = re.fullmatch(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.fullmatch(r"([av])(\d+)", tok)
    return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))} if m else set()


def main():
    path, needle = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if needle in l and re.match(r"^\S+:", l))
    inflight = []          # list of register sets, oldest first
    vm_inflight = []       # the same for vector memory operations (an empty set: no register destination)
    vm_reads = 0
    findings = 0
    reads = 0
    in_asm = False
    recent = []            # the last two instructions: (text, registers written by a VALU instruction)
    mfma_hazards = 0
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith("s_endpgm"):
            break
        if l.startswith(";;#ASMSTART"):
            in_asm = True
        elif l.startswith(";;#ASMEND"):
            in_asm = False
        if re.match(r"^\.LBB\d+_\d+:", l):
            # a branch target: what is in flight on the OTHER way in is unknown to a walk in text order.  The hand-placed
            # pipelines are straight-line code; start afresh (a wait precedes every join in them).
            inflight, vm_inflight, recent = [], [], []
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        op = l.split()[0]
        toks = [t.strip(",") for t in l.split()[1:]]
        if op.startswith("v_mfma") and in_asm:
            reads_of_mfma = set()
            for t in toks[1:]:
                reads_of_mfma |= any_regs(t)
            for text, written in recent:
                if written & reads_of_mfma:
                    mfma_hazards += 1
                    if mfma_hazards <= 10:
                        print(f"line {i + 1}: asm MFMA `{l}` reads what `{text}` wrote less than two wait states earlier")
        if op == "s_nop":
            states = int(toks[0]) + 1 if toks else 1
            recent = recent[states:] if states < len(recent) else []
        elif not op.startswith("s_waitcnt"):
            written = any_regs(toks[0]) if (op.startswith("v_") and not op.startswith("v_mfma") and toks) else set()
            recent = ([(l, written)] + recent)[:2]
        if op == "ds_read_b128":
            inflight.append(regs(toks[0]))
            reads += 1
            continue
        # vector memory operations return (loads) or retire (stores, LDS-DMA) in order: vmcnt(N) leaves the youngest N
        if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store", "scratch_")):
            dest = regs(toks[0]) if (op.startswith("global_load_dword") and in_asm) else set()   # loads hipcc issues itself are its own business
            vm_inflight.append(dest)
            if dest:
                vm_reads += 1
            if not in_asm:
                pass
            else:
                continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
                if n == 0:
                    inflight = []
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                vm_inflight = vm_inflight[len(vm_inflight) - n:] if n < len(vm_inflight) else vm_inflight
                if n == 0:
                    vm_inflight = []
            continue
        if op in ("s_barrier",):
            continue
        touched = set()
        for t in toks:
            touched |= regs(t)
        for k, dest in enumerate(inflight):
            if touched & dest:
                findings += 1
                if findings <= 25:
                    print(f"line {i + 1}: `{l}` touches v{sorted(touched & dest)} of an LDS read still in flight ({len(inflight) - k} youngest)")
        for k, dest in enumerate(vm_inflight):
            if touched & dest:
                findings += 1
                if findings <= 25:
                    print(f"line {i + 1}: `{l}` touches v{sorted(touched & dest)} of an asm global load still in flight ({len(vm_inflight) - k} youngest)")
    print(f"{reads} LDS reads and {vm_reads} asm global loads walked, {findings} findings; {mfma_hazards} asm MFMAs behind a fresh VALU write of an operand")


if __name__ == "__main__":
    main()
