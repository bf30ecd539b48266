"""Checks on the gfx950 ISA hipcc emitted for this library (build.py runs them on every translation unit that carries a
hand-placed wait; ``python -m sparsefactorization_amd.isa_lint file.s ...`` runs them by hand).

The kernels are correct only if a few properties of the *emitted* instruction stream hold which the C++ source cannot
express; each rule below names the source construct it guards.

R0  no vector instruction may read or write a register that a ds_read issued in the same basic block still owns by the
    in-order lgkmcnt count (hipcc's own wait insertion guarantees it; the rule catches a hand-placed wait that is too weak).

R1  counted vmcnt before a barrier (fwd_window.h, MODE 2: ``s_waitcnt vmcnt(NF*R + [RES] R)`` + bare ``s_barrier``).
    vmcnt retires in order, so after ``vmcnt(n)`` only the n youngest vector-memory operations may be outstanding: the
    LDS-DMAs (``global_load_lds_*``) have landed iff AT LEAST n counted operations were issued behind the last of them,
    in straight-line code. Exactly n is what the source intends (more = a longer wait than necessary). hipcc merging,
    dropping, hoisting or sinking one of the loads the source counts breaks the first and is reported as an error.

R2  no scalar-memory load may be outstanding at a counted ``lgkmcnt(n > 0)`` wait that guards LDS results: SMEM returns out
    of order, the count then proves nothing. (hipcc keeps to this by itself; the rule is for hand-placed waits and was
    the first suspect for profiles/r04b_mixer_lds_wait.md — it does not occur anywhere in the library.)

R3  in a kernel that issues MFMAs, a packed-f32 instruction (``v_pk_*_f32``: 64-bit register-pair operands) must not be
    among the first K instructions behind an ``s_waitcnt lgkmcnt(n)`` while reading a register that this very wait may
    just have released (the destination of a ``ds_read`` that was still countable as outstanding before the wait).
    That is the one pattern the ISA of the sporadically wrong build of chord_fwd_mlp_k has and the two clean builds
    (full wait before the arithmetic; -fno-slp-vectorize) do not: see profiles/r04b_mixer_lds_wait.md, "What the ISA says".
"""
from __future__ import annotations

import re
import sys
from dataclasses import dataclass, field

VM_COUNTED = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
REG_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
K_FIRST = 2  # R3: how many instructions behind the wait count as "right behind it"


@dataclass
class Insn:
    line: int
    op: str
    args: str
    in_asm: bool = False


@dataclass
class Kernel:
    name: str
    items: list = field(default_factory=list)  # Insn or ("label", name, line)


def _vregs(text: str) -> set:
    out = set()
    for m in REG_RANGE.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _split_operands(args: str):
    """(destination text, source text) of 'dst, src0, src1 ...'."""
    depth, cut = 0, None
    for i, ch in enumerate(args):
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        elif ch == "," and depth == 0:
            cut = i
            break
    return (args, "") if cut is None else (args[:cut], args[cut + 1:])


def parse(path: str):
    kernels, cur, in_asm = [], None, False
    with open(path) as fh:
        for ln, raw in enumerate(fh, 1):
            s = raw.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            m = re.match(r"^\.type\s+(\S+),@function", s)
            if m:
                cur = Kernel(m.group(1))
                kernels.append(cur)
                continue
            if cur is None:
                continue
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            m = re.match(r"^(\.LBB\S+|[A-Za-z_]\w*):", s)
            if m:
                cur.items.append(("label", m.group(1), ln))
                continue
            if not s or s[0] in ".;":
                continue
            s = s.split(";")[0].strip()
            parts = s.split(None, 1)
            cur.items.append(Insn(ln, parts[0], parts[1] if len(parts) > 1 else "", in_asm))
    return kernels


def _wait_counts(args: str):
    """{'vmcnt': n, 'lgkmcnt': n} named in an s_waitcnt (absent = not waited for)."""
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", args)}


def check_kernel(k: Kernel):
    errs, notes = [], []
    has_mfma = any(isinstance(i, Insn) and i.op.startswith("v_mfma") for i in k.items)
    # (DMAs written as inline assembly — x3_gemm.h's glds16 — are counted by the source itself, per loop iteration: R1 is
    # for loads the compiler emits from builtins and may merge, drop or move)
    has_dma = any(isinstance(i, Insn) and i.op.startswith("global_load_lds") and not i.in_asm for i in k.items)

    # ---- R1 ----
    if has_dma:
        for idx, it in enumerate(k.items):
            if not (isinstance(it, Insn) and it.in_asm and it.op == "s_waitcnt"):
                continue
            n = _wait_counts(it.args).get("vmcnt")
            if n is None or n == 0:
                continue
            behind, found, straight = 0, False, True
            for prev in reversed(k.items[:idx]):
                if not isinstance(prev, Insn):
                    straight = False
                    break
                if prev.op.startswith("global_load_lds"):
                    found = True
                    break
                if prev.op.startswith(("s_cbranch", "s_branch")):
                    straight = False
                    break
                if VM_COUNTED.match(prev.op):
                    behind += 1
            if not found:
                if not straight:
                    errs.append(f"{k.name}: line {it.line}: hand-placed vmcnt({n}) is not in straight-line code behind the last LDS-DMA")
                continue
            if behind < n:
                errs.append(f"{k.name}: line {it.line}: vmcnt({n}) but only {behind} counted operations behind the last "
                            f"global_load_lds: the barrier can be passed with a DMA in flight")
            elif behind > n:
                notes.append(f"{k.name}: line {it.line}: vmcnt({n}) with {behind} counted operations behind the last DMA (waits longer than intended)")

    # ---- R2, R3: one pass in textual order; the LDS queue is forgotten at labels (unknown predecessors) ----
    smem_pending = False
    queue = []  # ds_read destinations in issue order (None for LDS operations that return nothing but still count)
    check_regs, check_left = set(), 0
    for it in k.items:
        if not isinstance(it, Insn):
            queue, check_left = [], 0
            continue
        op = it.op
        if op.startswith(("s_load", "s_buffer_load")):
            smem_pending = True
        if op == "s_waitcnt":
            n = _wait_counts(it.args).get("lgkmcnt")
            if n is not None:
                if n == 0:
                    smem_pending = False
                elif smem_pending and queue:
                    errs.append(f"{k.name}: line {it.line}: counted lgkmcnt({n}) with a scalar-memory load possibly outstanding")
                released = set()
                while len(queue) > n:
                    d = queue.pop(0)
                    if d:
                        released |= d
                check_regs, check_left = released, K_FIRST
            continue
        if op.startswith("ds_"):
            dst, _ = _split_operands(it.args)
            queue.append(_vregs(dst) if op.startswith("ds_read") or "_rtn" in op else None)
            continue
        if queue and op.startswith(("v_", "global_", "buffer_")) and not op.startswith("v_mfma"):
            pending = set().union(*[d for d in queue if d])
            hit = _vregs(it.args) & pending
            if hit:
                errs.append(f"{k.name}: line {it.line}: {op} touches v{sorted(hit)} while a ds_read into it is still outstanding "
                            f"by the lgkmcnt count (R0)")
        if check_left > 0 and op.startswith("v_"):
            if has_mfma and op.startswith("v_pk_") and op.endswith("_f32"):
                _, src = _split_operands(it.args)
                hit = _vregs(src) & check_regs
                if hit:
                    errs.append(f"{k.name}: line {it.line}: {op} reads v{sorted(hit)} right behind the lgkmcnt wait that "
                                f"released it (R3)")
            check_left -= 1
    return errs, notes


def lint_file(path: str):
    errs, notes, n = [], [], 0
    for k in parse(path):
        e, w = check_kernel(k)
        errs += e
        notes += w
        n += 1
    return errs, notes, n


def main(argv):
    bad = 0
    for p in argv:
        errs, notes, n = lint_file(p)
        print(f"{p}: {n} functions, {len(errs)} errors, {len(notes)} notes")
        for e in errs:
            print("  ERROR", e)
        for w in notes:
            print("  note ", w)
        bad += len(errs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
