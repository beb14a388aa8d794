#!/usr/bin/env python3
"""The device's timeline of ONE steady-state bench step from a rocprofv3 --kernel-trace database: every kernel of the last complete
step traced (a step begins at the anchor call's IndexInsert, the longest IndexInsert dispatch), with the idle time before it where
that exceeds 8 us -- where the device waits for the host.   python scripts/step_timeline.py <dir with *_results.db> [out.txt]"""
import glob, os, re, sqlite3, sys
d = sys.argv[1]
db = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))[0]
c = sqlite3.connect(db)
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = "kernels" if "kernels" in tables else next(x for x in tables if x.startswith("kernels"))
rows = list(c.execute("select name, start, end from %s order by start" % t))
def short(n):
    m = re.search(r"pm_(?:wave_)?kernel<pm::(\w+)>", n)
    if m: return m.group(1)
    m = re.search(r"(pm_fill16|pm_fill_many|gap_align_kernel|radix_sort\w*|onesweep\w*|scan\w*|copyBuffer|fillBuffer\w*)", n)
    return m.group(1) if m else n.split("(")[0][-40:]
idx = [i for i, (n, b, e) in enumerate(rows) if "IndexInsert" in n]
big = max(rows[i][2] - rows[i][1] for i in idx)
starts = [i for i in idx if rows[i][2] - rows[i][1] > 0.6 * big]
a, b = starts[-2], starts[-1]
# the step's launches begin a little before its IndexInsert (set-up copies): walk back over copies / fills
while a > 0 and ("copyBuffer" in rows[a - 1][0] or "fillBuffer" in rows[a - 1][0] or "GatherRegions" in rows[a - 1][0] or "CheckRows" in rows[a - 1][0] or "AlgBytes" in rows[a - 1][0]) and rows[a][1] - rows[a - 1][2] < 200000: a -= 1
while b > 0 and ("copyBuffer" in rows[b - 1][0] or "fillBuffer" in rows[b - 1][0]) and rows[b][1] - rows[b - 1][2] < 200000: b -= 1
out = []
busy = idle = 0
for i in range(a, b):
    n, s, e = rows[i]
    gap = (s - rows[i - 1][2]) / 1e3 if i > a else 0.0
    busy += (e - s) / 1e3
    if i > a: idle += max(gap, 0)
    out.append("%9.1f us  %-28s %8.1f us%s" % ((s - rows[a][1]) / 1e3, short(n), (e - s) / 1e3, ("   <- idle %.1f us before" % gap) if gap > 8 else ""))
span = (rows[b - 1][2] - rows[a][1]) / 1e3
head = "# last complete step of the trace: %d dispatches, span %.1f us, kernels busy %.1f us, idle %.1f us (gaps > 8 us marked)\n" % (b - a, span, busy, idle)
text = head + "\n".join(out) + "\n"
if len(sys.argv) > 2: open(sys.argv[2], "w").write(text)
gaps = sorted(((rows[i][1] - rows[i - 1][2]) / 1e3, short(rows[i - 1][0]), short(rows[i][0])) for i in range(a + 1, b))
print(head, end="")
for g, p, q in gaps[::-1][:25]: print("%8.1f us  %s -> %s" % (g, p, q))
