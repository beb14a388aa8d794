#!/usr/bin/env python3
"""Instruction mix of one kernel in hipcc's --save-temps assembly: python scripts/isa_stats.py file.s KernelName"""
import collections
import re
import sys
s = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r"^_Z\S*%s\S*:" % name, l))
end = next(i for i in range(start, len(s)) if ".amdhsa_kernel" in s[i])
body = [l.strip() for l in s[start + 1:end]]
ops = [l.split()[0] for l in body if l and not l.startswith((".", ";")) and not l.endswith(":")]
c = collections.Counter(ops)
groups = collections.Counter()
for k, v in c.items():
    g = "s_load" if k.startswith("s_load") else "global_load" if k.startswith("global_load") else "global_store" if k.startswith("global_store") \
        else "atomic" if "atomic" in k else "scratch" if k.startswith("scratch") else "s_waitcnt" if k.startswith("s_waitcnt") \
        else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "other"
    groups[g] += v
print(dict(groups), "total", len(ops))
for l in s[end:end + 60]:
    if re.search(r"next_free_vgpr|next_free_sgpr|scratch|private_segment_fixed", l):
        print(l.strip())
