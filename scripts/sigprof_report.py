"""Turn the samples of scripts/sigprof.c into (library, symbol) counts.

    python scripts/sigprof_report.py gpurun_out/prof.txt [--from-ms A --to-ms B] [--top 40]

Symbols come from `nm` on the mapped files (the same image here and on the GPU box; the in-tree libraries travel with the
snapshot), so the report can be made in the build container from a profile taken on the box.  Prints the samples per
100 ms first (to pick the window of the timed steps), then the table."""
import argparse, bisect, collections, os, subprocess, sys


def symbols(path):
    out = []
    for flags in (["-n", "-C", "--defined-only"], ["-n", "-C", "-D", "--defined-only"]):
        try:
            txt = subprocess.run(["nm"] + flags + [path], capture_output=True, text=True).stdout
        except OSError:
            continue
        for line in txt.splitlines():
            p = line.split(None, 2)
            if len(p) == 3 and p[1] in "TtWwiV":
                out.append((int(p[0], 16), p[2]))
    out.sort()
    return [a for a, _ in out], [s for _, s in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--from-ms", type=int, default=0)
    ap.add_argument("--to-ms", type=int, default=1 << 30)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--remap", default="", help="prefix=replacement for the paths in the map (a repository checked out elsewhere)")
    a = ap.parse_args()
    maps, samples = [], []
    for line in open(a.file):
        if line[0] == "M":
            p = line[2:].split()
            if len(p) < 6:
                continue
            lo, hi = (int(x, 16) for x in p[0].split("-"))
            maps.append((lo, hi, int(p[2], 16), p[5]))
        elif line[0] == "S":
            pc, tid, ms = line[2:].split()
            samples.append((int(pc, 16), int(tid), int(ms)))
    base = {}
    for lo, hi, off, path in maps:
        if off == 0:
            base[path] = min(lo, base.get(path, lo))
    maps.sort()
    starts = [m[0] for m in maps]
    per100 = collections.Counter(ms // 100 for _, _, ms in samples)
    print("samples per 100 ms:", " ".join("%d:%d" % (k, per100[k]) for k in sorted(per100)))
    sel = [s for s in samples if a.from_ms <= s[2] < a.to_ms]
    print("%d samples in the window (%d in all), %d threads" % (len(sel), len(samples), len({s[1] for s in sel})))
    tables, by = {}, collections.Counter()
    bylib = collections.Counter()
    for pc, tid, ms in sel:
        i = bisect.bisect_right(starts, pc) - 1
        if i < 0 or pc >= maps[i][1]:
            by[("?", "?")] += 1
            continue
        path = maps[i][3]
        real = path
        if a.remap:
            k, v = a.remap.split("=")
            if real.startswith(k):
                real = v + real[len(k):]
        if path not in tables:
            tables[path] = symbols(real) if os.path.exists(real) else ([], [])
        addrs, names = tables[path]
        va = pc - base.get(path, maps[i][0])
        j = bisect.bisect_right(addrs, va) - 1
        name = names[j] if j >= 0 else "?"
        lib = os.path.basename(path)
        by[(lib, name)] += 1
        bylib[lib] += 1
    total = max(1, len(sel))
    print("\nby library:")
    for lib, c in bylib.most_common(12):
        print("  %6.2f %%  %s" % (100.0 * c / total, lib))
    print("\nby symbol:")
    for (lib, name), c in by.most_common(a.top):
        print("  %6.2f %%  %-22s %s" % (100.0 * c / total, lib, name[:150]))


if __name__ == "__main__":
    main()
