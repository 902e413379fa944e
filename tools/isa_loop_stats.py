"""Opcode statistics of the MFMA-heavy basic blocks of one kernel in a hipcc -save-temps .s file.
Usage: python tools/isa_loop_stats.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if pat in m]
for name in names:
    a = s.index(name + ":")
    b = s.index(".Lfunc_end", a)
    blocks, cur = [], []
    for l in s[a:b].splitlines():
        if re.match(r"^\.LBB", l):
            blocks.append(cur)
            cur = [l]
        else:
            cur.append(l)
    blocks.append(cur)
    print(name)
    for blk in blocks:
        ins = [l.strip() for l in blk if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(i.split()[0] for i in ins)
        nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        if nm < 8:
            continue
        print(f"  block {blk[0][:12]:12s} {len(ins):4d} instr, {nm} mfma;", ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
        print("     vmcnt waits:", [i.split("vmcnt(")[1].split(")")[0] for i in ins if "vmcnt(" in i])
