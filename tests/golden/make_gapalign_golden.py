"""Golden vectors for the inter-MUM gap aligner (parsnp_amd/csrc/host/gapalign.cpp).

Run in the build container (needs oracle/_ref/muscle_ref = the REFERENCE's MuscleInterface::CallMuscleFast over its
vendored libMUSCLE 3.7, built by `make -C oracle ref`).  Writes tests/golden/gapalign.json: a list of
{"in": [...], "out": [...]} blocks -- inputs from the seeded generator in tests/gapgen.py, outputs from the reference.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gapgen  # noqa: E402


def main():
    blocks = gapgen.golden_blocks()
    outs = gapgen.reference_align(blocks)
    assert len(outs) == len(blocks)
    data = [{"in": b, "out": o} for b, o in zip(blocks, outs)]
    path = os.path.join(HERE, "gapalign.json")
    json.dump(data, open(path, "w"), separators=(",", ":"))
    print(path, len(data), "blocks", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
