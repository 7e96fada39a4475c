#!/usr/bin/env python3
"""Static check of the gfx950 ISA (csrc/bp5_device.s from `make asm`): every s_barrier must be reached with no LDS
write of the same wave still in flight, i.e. an `s_waitcnt lgkmcnt(0)` must lie between a ds_write/ds_add and the next
s_barrier on every path, including loop back-edges.

Why: the compiler was seen to drop the wait in front of a barrier at a loop header (the LDS write at the end of the loop
body reaches the barrier through the back-edge).  Waves of one workgroup run on different SIMDs; without the wait another
wave can pass the barrier and read-modify-write the same LDS word before the write has landed (lost update in the block
kernel's multi-round accumulation, seen as rare wrong sums at full size)."""
import re
import sys


def functions(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name, body = None, []


def check(body):
    labels, blocks, cur = {}, [], []
    for ln in body:
        s = ln.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            labels[m.group(1)] = len(blocks)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur.append(s.split(";")[0].strip())
        if re.match(r"s_branch|s_cbranch|s_endpgm", s):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        out = []
        last = b[-1] if b else ""
        m = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)", last)
        if m:
            if m.group(2) in labels and labels[m.group(2)] < len(blocks):
                out.append(labels[m.group(2)])
            if m.group(1) != "s_branch" and i + 1 < len(blocks):
                out.append(i + 1)
        elif "s_endpgm" not in last and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    pend_in = [False] * len(blocks)
    bad = []

    def transfer(i, p, report):
        for ins in blocks[i]:
            if re.match(r"ds_(write|add|wrxchg|max|min|and|or|xor|sub|inc|dec|cmpst)", ins):
                p = True
            elif ins.startswith("s_waitcnt"):
                if "lgkmcnt(0)" in ins:
                    p = False
            elif ins.startswith("s_barrier") and p and report:
                bad.append((i, ins))
        return p

    changed = True
    while changed:
        changed = False
        for i in range(len(blocks)):
            out = transfer(i, pend_in[i], False)
            for j in succ[i]:
                if out and not pend_in[j]:
                    pend_in[j] = True
                    changed = True
    for i in range(len(blocks)):
        transfer(i, pend_in[i], True)
    return bad


def main(path):
    n_bad = 0
    for name, body in functions(path):
        bad = check(body)
        if bad:
            n_bad += len(bad)
            print(f"{name}: {len(bad)} barrier(s) reachable with an LDS write in flight")
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
