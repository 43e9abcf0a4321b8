#!/usr/bin/env python3
"""Build-time check of vit_ws_gemm.hip's hand-counted panel fetch (run by the Makefile on the generated assembly).

The kernel loads the next activation panel with inline-asm global_load_dwordx4 into four registers quads and waits for them
with its own `s_waitcnt vmcnt(N)` one loop iteration later (the compiler cannot see that the registers are pending).  That is
only correct if, between a fetch and the wait that follows it, the compiler never touches those registers: no spill, no copy,
no reuse.  This script proves it on the emitted ISA: every instruction whose nearest preceding asm block (in layout order,
which is also execution order inside the single-loop kernel) is a FETCH must not mention a fetched register or use scratch (spills and
reloads OUTSIDE a fetch span are tolerated: they cannot involve the fetch registers).  The other half of the protocol is checked too: `s_waitcnt vmcnt(N)` only proves that the fetch has
landed if at least N vector-memory operations were issued AFTER it (vmcnt retires in issue order): the script counts the
global_/buffer_/flat_ loads, stores and atomics between each fetch and its wait -- instructions inside an inner loop (a label
with a backward branch inside that span: the two half-panel steps) count twice -- and fails if there are fewer than N, i.e. if
an epilogue edit or a compiler change removed memory operations the hand count relies on.  Exit status 1 with the offending
line otherwise."""

INNER_TRIPS = 2     # the only loop between a fetch and its wait is the `h2` loop over the two half-panels
import re
import sys


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


VMEM = re.compile(r"^(global|buffer|flat)_(load|store|atomic)")


def vmem_ops_issued(span):
    """vector-memory instructions executed between a fetch and its wait: layout order, with the body of an inner loop (label ...
    backward branch to it, both inside the span) counted INNER_TRIPS times"""
    labels = {t[:-1]: i for i, t in enumerate(span) if re.match(r"^\.LBB\w+:$", t)}
    weight = [1] * len(span)
    for i, t in enumerate(span):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\w+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            for j in range(labels[m.group(1)], i):
                weight[j] = INNER_TRIPS
    return sum(w for t, w in zip(span, weight) if VMEM.match(t))


def main(path):
    src = open(path).read()
    bad = 0
    funcs = re.findall(r"^(_Z\d+vit_ws_gemm_kernel\w+):[^\n]*\n(.*?)\n\s*s_endpgm", src, re.S | re.M)
    if not funcs:
        print("check_ws_gemm_isa: no vit_ws_gemm_kernel in", path)
        return 1
    for name, body in funcs:
        lines = body.split("\n")
        state, pending, in_asm, block = "idle", set(), False, []
        n_fetch = n_wait = 0
        span = []                 # instructions (and labels) between the current fetch and its wait
        for ln in lines:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm, block = True, []
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                txt = "\n".join(block)
                if "global_load_dwordx" in txt:
                    # alternative fetch blocks (with / without the row statistics) are laid out one after the other and only one of them
                    # executes: until the wait, the registers of ALL of them count as pending (round 4: a copy of st2 placed right behind
                    # the block that loads it went unnoticed because a later alternative had replaced the pending set)
                    if state != "pending":
                        pending, span = set(), []
                    state, n_fetch = "pending", n_fetch + 1
                    for b in block:
                        m = re.match(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", b.strip())
                        if m:
                            pending.update(range(int(m.group(1)), int(m.group(2)) + 1))
                elif "s_waitcnt vmcnt" in txt and state == "pending":
                    state, n_wait = "idle", n_wait + 1
                    need = int(re.search(r"vmcnt\((\d+)\)", txt).group(1))
                    have = vmem_ops_issued(span)
                    if have < need:
                        print(f"{name}: s_waitcnt vmcnt({need}) follows a panel fetch after only {have} younger vector-memory "
                              f"operation(s): the wait can pass before the fetch has landed")
                        bad += 1
                elif state == "pending":
                    span.extend(block)                     # another asm block between fetch and wait (none today)
                continue
            if in_asm:
                block.append(t)
                continue
            if state == "pending" and t and not t.startswith(";"):
                span.append(t.split(";")[0].strip())
            if state == "pending" and t and not t.startswith(";") and not t.startswith("."):
                code = t.split(";")[0]
                if "scratch_" in code:
                    # a spill or reload while a fetch is in flight: never accepted (a spill OUTSIDE a fetch span -- a loop-invariant value the
                    # allocator parks before the loop and reloads at the top of a panel, round 4's LN = 5 variant -- cannot involve the fetch
                    # registers, and extra vector-memory operations only make the counted waits stricter)
                    print(f"{name}: `{code.strip()}` uses scratch memory while a panel fetch is in flight")
                    bad += 1
                hit = regs_of(code) & pending
                if hit:
                    print(f"{name}: `{code.strip()}` touches v{sorted(hit)} while the panel fetch into them is in flight")
                    bad += 1
        if n_fetch < 2 or n_wait < 2 or state != "idle":
            print(f"{name}: expected fetch sites in the prologue and in the loop, each followed by its wait: found {n_fetch} fetches, "
                  f"{n_wait} waits, final state {state}")
            bad += 1
    print(f"check_ws_gemm_isa: {len(funcs)} kernels, {bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
