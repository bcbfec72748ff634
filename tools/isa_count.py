#!/usr/bin/env python
"""tools/isa_count.py <file.s> <kernel-substring> [--dump out.s]: static instruction counts of one kernel of a `hipcc -S --cuda-device-only`
listing, per section between s_barrier instructions (VALU / SALU / LDS / VMEM / branch / waitcnt), with its register / scratch / LDS figures.
Used to see where an issue-bound kernel spends its instructions."""
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
lines = open(src).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.endswith(":") is False and re.match(r"^(_Z\w+):", l) and pat in l:
        start = i
        name = l.split(":")[0]
        break
if start is None:
    sys.exit("kernel not found")
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
if dump:
    open(dump, "w").write("\n".join(body))
meta = {}
for l in lines[end:end + 400]:
    m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\S+)", l)
    if m:
        meta[m.group(1)] = m.group(2)
    m = re.match(r";\s*(ScratchSize|Occupancy|NumVgprs|NumAgprs|TotalNumSgprs|LDSByteSize|codeLenInByte)[^:]*:\s*(\S+)", l)
    if m:
        meta[m.group(1)] = m.group(2)
    if ".end_amdhsa_kernel" in l:
        pass
print(name[:110])
print(" ", meta)


def cls(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


sec, secs, label = {}, [], "entry"
for l in body[1:]:
    t = l.strip()
    if not t or t.startswith((";", ".", "//")):
        continue
    if re.match(r"^\.?\w+:", t):
        continue
    op = t.split()[0]
    c = cls(op)
    if c == "barrier":
        secs.append(sec)
        sec = {}
        continue
    sec[c] = sec.get(c, 0) + 1
secs.append(sec)
keys = ["valu", "salu", "lds", "vmem", "smem", "branch", "wait", "other"]
print("  section  " + "  ".join(f"{k:>6}" for k in keys))
tot = {}
for i, s in enumerate(secs):
    print(f"  {i:>7}  " + "  ".join(f"{s.get(k, 0):>6}" for k in keys))
    for k in keys:
        tot[k] = tot.get(k, 0) + s.get(k, 0)
print("    total  " + "  ".join(f"{tot.get(k, 0):>6}" for k in keys))
