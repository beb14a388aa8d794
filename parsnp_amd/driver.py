"""The caller's side of the drop-in boundary: write the .ini exactly as the reference's Python driver does and run a
parsnp_core binary on it.

ini text = the reference's template.ini with the placeholder substitutions of `parsnp` write_inifile_1/2
(parsnp:1047-1067, :1097-1133) and the driver's argparse defaults (parsnp:416-469, :554-558)."""
import os
import random
import subprocess

DEFAULTS = dict(anchors="1.1*(Log(S))", mums="1.1*(Log(S))", extend=0, recombfilt=0, threads=1, diagdiff=0.12, aligner=2,
                mincluster=21, clusterd=300, partpos=15000000, unaligned=0, calcmumi=0)

_TEMPLATE = """;Parsnp configuration File
;
[Reference]
file={ref}
reverse=0
[Query]
{files}[MUM]
anchors={anchors}
anchorfile=
anchorsonly=0
calcmumi={calcmumi}
mums={mums}
mumfile=
filter=1
factor=2.0
extendmums={extend}
[LCB]
recombfilter={recombfilt}
cores={threads}
diagdiff={diagdiff}
doalign={aligner}
c={mincluster}
d={clusterd}
q=30
p={partpos}
icr=0
unaligned={unaligned}
[Output]
outdir={outdir}
prefix=parsnp
showbps=1
"""


def driver_order(paths):
    """query order of the reference driver: sorted(), then random.Random(42).shuffle (parsnp:29, :1509-1510)."""
    out = sorted(paths)
    random.Random(42).shuffle(out)
    return out


def ini_text(ref, queries, outdir, **kw):
    p = dict(DEFAULTS, **kw)
    files = "".join("file%d=%s\nreverse%d=0\n" % (i, q, i) for i, q in enumerate(queries, 1))
    return _TEMPLATE.format(ref=ref, files=files, outdir=outdir, **p)


def run_core(core_bin, ref, queries, outdir, timing=None, env=None, timeout=None, **kw):
    """-> (returncode, ini path).  stdout/stderr go to <outdir>/parsnp-aligner.{out,err} like the driver's log dir."""
    os.makedirs(outdir, exist_ok=True)
    ini = os.path.join(outdir, "parsnpAligner.ini")
    with open(ini, "w") as f:
        f.write(ini_text(ref, queries, outdir, **kw))
    e = dict(os.environ if env is None else env)
    if timing:
        e["PARSNP_TIMING"] = timing
    with open(os.path.join(outdir, "parsnp-aligner.out"), "w") as so, open(os.path.join(outdir, "parsnp-aligner.err"), "w") as se:
        rc = subprocess.run([core_bin, ini], stdout=so, stderr=se, cwd=outdir, env=e, timeout=timeout).returncode
    return rc, ini
