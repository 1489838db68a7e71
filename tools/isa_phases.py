#!/usr/bin/env python3
"""count ISA instructions of one kernel between s_barrier's (usage: isa_phases.py file.s mangled-name-substring)"""
import re, sys, collections
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(_Z\w*" + re.escape(pat) + r"\w*):.*?\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
body = m.group(2).split("\n")
print(m.group(1), len(body), "lines")
seg = 0
cnt = collections.defaultdict(collections.Counter)
for l in body:
    l = l.strip()
    if not l or l[0] in ";." or l.endswith(":"):
        continue
    op = l.split()[0]
    if op == "s_barrier":
        seg += 1
        continue
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_")) else "salu" if op.startswith("s_") else "other")
    cnt[seg][k] += 1
    cnt[seg]["op:" + op] += 1
for sg in sorted(cnt):
    c = cnt[sg]
    print("segment", sg, {k: v for k, v in c.items() if not k.startswith("op:")})
    print("    ", " ".join("%s:%d" % (k[3:], v) for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("op:"))[:900])
for key in ("vgpr_count", "sgpr_count", "lds_size", "scratch"):
    mm = re.search(r"\.amdhsa_next_free_vgpr (\d+)", s[m.end():m.end() + 6000]) if key == "vgpr_count" else None
    if mm:
        print("next_free_vgpr", mm.group(1))
