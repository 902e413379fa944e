"""Opcode statistics of the MFMA-heavy basic blocks of one kernel in a hipcc -save-temps .s file.
Usage: python tools/isa_loop_stats.py file.s <substring of the mangled kernel name>
CAUTION: a "block" here is the text between two .LBB labels — it says nothing about loop membership.  A block that ends in
a backward branch may be the loop's EXIT path (hipcc puts the accumulator shuffles of the epilogue there): check with
tools/isa_mfma_gaps.sh / the branch targets before reading a block's v_accvgpr_* count as per-iteration work."""
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
