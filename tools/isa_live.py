#!/usr/bin/env python
"""tools/isa_live.py <kernel.s dumped by isa_count.py --dump> <section>: which VGPRs are live THROUGH a section (between two s_barrier
instructions) without being touched in it -- straight-line approximation (branches ignored), good enough to find what a loop body carries."""
import re
import sys

lines = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith((";", "."))]
sec = int(sys.argv[2])
secs, cur = [], []
for l in lines:
    if l.startswith("s_barrier"):
        secs.append(cur)
        cur = []
    else:
        cur.append(l)
secs.append(cur)


def regs(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", tok):
        if m.group(1):
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def defs_uses(l):
    l = l.split(";")[0]
    parts = l.split(None, 1)
    if len(parts) < 2:
        return [], []
    op, rest = parts
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(("global_store", "scratch_store", "ds_write", "ds_add_u32", "buffer_store", "v_cmp", "v_cmpx", "s_")) and not op.startswith(("v_cmp_", "v_cmpx_")) or op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store")):
        return [], [r for o in ops for r in regs(o)]
    if op.startswith(("v_cmp_", "v_cmpx_")):
        return [], [r for o in ops for r in regs(o)]
    d = regs(ops[0])
    u = [r for o in ops[1:] for r in regs(o)]
    if op.startswith(("v_mac", "v_fmac", "v_writelane", "v_bfi")) or "dst_unused:UNUSED_PRESERVE" in l:
        u += d
    return d, u


def touched(section):
    t = set()
    for l in section:
        d, u = defs_uses(l)
        t.update(d)
        t.update(u)
    return t


# live-in of everything after `sec`: used before defined
after = [l for s in secs[sec + 1:] for l in s] + [l for s in secs[:sec] for l in s]     # the loop wraps around
live, dead = set(), set()
for l in after:
    d, u = defs_uses(l)
    for r in u:
        if r not in dead:
            live.add(r)
    for r in d:
        if r not in live:
            dead.add(r)
t = touched(secs[sec])
through = sorted(live - t)
print("sections:", len(secs), "| touched in section", sec, ":", len(t), "| live after it (first use before def, wrap-around):", len(live))
print("live through the section without being touched:", through)
defd = set()
for l in secs[sec]:
    d, u = defs_uses(l)
    defd.update(d)
print("defined in the section and live after it:", len(sorted(live & defd)))
