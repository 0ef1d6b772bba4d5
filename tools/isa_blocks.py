#!/usr/bin/env python3
"""Offline (no GPU): basic blocks of one kernel of a disassembled code object (llvm-objdump -d), with the
VALU / SALU / other instruction counts of each block and where it branches -- for working out what a
wavefront executes per loop iteration.  Usage: isa_blocks.py <objdump.s> <kernel-name-substring> [lo hi]"""
import re
import sys


def parse(path, kernel):
    insts, on = [], False
    for line in open(path):
        if re.match(r"^[0-9a-f]+ <", line):
            on = kernel in line
            continue
        if not on:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):(.*)$", line)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    return insts


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    insts = parse(path, kernel)
    base = insts[0][0]
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    leaders = {0}
    targets = {}
    for i, (a, op, args) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"\+0x([0-9a-f]+)>", args)
            t = base + int(m.group(1), 16) if m else None
            targets[i] = t
            if t in addr_index:
                leaders.add(addr_index[t])
            leaders.add(i + 1)
        elif op == "s_endpgm":
            leaders.add(i + 1)
    order = sorted(x for x in leaders if x < len(insts))
    lo = int(sys.argv[3], 16) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4], 16) if len(sys.argv) > 4 else 1 << 62
    for bi, start in enumerate(order):
        end = order[bi + 1] if bi + 1 < len(order) else len(insts)
        a0 = insts[start][0] - base
        if not (lo <= a0 < hi):
            continue
        valu = sum(1 for _, op, _ in insts[start:end] if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
        lanes = sum(1 for _, op, _ in insts[start:end] if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
        trans = sum(1 for _, op, _ in insts[start:end] if re.match(r"v_(rsq|rcp|sqrt|sin|cos|exp|log)", op))
        salu = sum(1 for _, op, _ in insts[start:end] if op.startswith("s_") and op not in ("s_nop", "s_waitcnt"))
        nops = sum(1 for _, op, _ in insts[start:end] if op == "s_nop")
        mem = sum(1 for _, op, _ in insts[start:end] if op.startswith(("global_", "ds_", "buffer_", "scratch_", "flat_")))
        last = insts[end - 1]
        tail = ""
        if end - 1 in targets:
            t = targets[end - 1]
            tail = "%s -> +0x%x" % (last[1], (t or 0) - base)
        elif last[1] == "s_endpgm":
            tail = "end"
        print("+0x%05x  n=%4d valu=%4d (trans %d) lane=%3d salu=%4d nop=%3d mem=%2d  %s" % (a0, end - start, valu, trans, lanes, salu, nops, mem, tail))


if __name__ == "__main__":
    main()
