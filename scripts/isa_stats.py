"""Static instruction mix of one kernel in a hipcc -S dump, per basic block (loop bodies show up as
blocks that end in a backward branch).  usage: isa_stats.py nmpc.s <mangled-kernel-name-substring>"""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def klass(op):
    if "dpp" in op:
        return "dpp"
    if op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "xlane"
    if op.startswith("ds_bpermute"):
        return "bperm"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_") and op.endswith("_f64") or "_f64_" in op:
        return "vf64"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "vcmp"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem:" + op.split("_")[0]
    return "other"


blocks, cur, name = [], Counter(), "entry"
label_line = {}
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if sum(cur.values()):
                blocks.append((name, cur, None))
            name, cur = m.group(1), Counter()
            label_line[name] = len(blocks)
        continue
    op = l.split()[0]
    cur[klass(op)] += 1
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        tgt = l.split()[-1]
        blocks.append((name, cur, tgt))
        name, cur = name + "'", Counter()
if sum(cur.values()):
    blocks.append((name, cur, None))
tot = Counter()
for n, c, t in blocks:
    tot.update(c)
    valu = sum(v for k, v in c.items() if k in ("dpp", "xlane", "vf64", "cndmask", "vcmp", "valu_other"))
    back = t is not None and t in label_line and label_line[t] <= blocks.index((n, c, t))
    if sum(c.values()) >= 25:
        print(f"{n:14s} n={sum(c.values()):5d} valu={valu:5d} {'LOOP->' + t if back else (t or ''):14s} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("TOTAL", sum(tot.values()), dict(tot))
