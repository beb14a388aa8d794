"""The flows of BASELINE configs 4 and 5 (tests/flows.py) at a reduced size on the CPU checker: 8 partitions of 4 genomes x
150 kb + native merge, and 10 x 400 kb rearranged genomes unsharded and sharded over 2 ranks (gloo).  Same assertions as the
full-size runs of tests/test_gpu_big.py."""
import os

import flows
from parsnp_amd import synth


def test_config4_flow_small(cpu_checkers, tmp_path):
    kw = dict(synth.CONFIGS["bact2000"][1], n=150_000)
    flows.config4_flow(cpu_checkers, str(tmp_path), kw, 4, 8, threads=4, min_lcbs=10)


def test_config5_flow_small(cpu_checkers, emu, tmp_path):
    # the sharded form runs the host code against the sequential kernel emulation (tests/emu): the CPU checker has no sharded session
    core_lib = os.path.join(os.path.dirname(emu[0]), "libparsnp_core_emu.so")
    flows.config5_flow(emu[1], str(tmp_path), "poprearr10x400k", dict(n=120_000, n_genomes=6), threads=4, ranks=2, min_lcbs=10, min_reverse=5,
                       sharded_env={"PARSNP_CORE_LIB": core_lib})
