"""CPU checks of the generated leaf-hash block (tools/gen_keccak_asm.py -> virgo-plus_amd/csrc/vp_keccak_asm.h): the instruction list, interpreted on the CPU,
chains SHA3-256 like hashlib (fri.cpp:96-124 / my_hhash.h:27-33 semantics), for every generator variant; and the committed header is what the
generator writes, so the two cannot drift apart."""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytest

import check_keccak_asm as chk
import gen_keccak_asm as gen


@pytest.mark.parametrize("rot1", ["alignbit", "fast"])
@pytest.mark.parametrize("dce", [True, False])
def test_generated_block_chains_sha3_256(rot1, dce):
    assert chk.check(rot1, dce, blocks=4, seed=7)


def test_bank_placement_of_the_generated_block():
    code, regs = gen.build_body("alignbit", True)
    n = gen.stats(code)
    assert n["bitop3_bank_pairs"] <= 10 * 24          # chi's one forced pair per row and half, nothing else
    assert len(regs["used"]) <= 112 and max(regs["used"]) < gen.BASE + gen.SPAN
    # with a barrier at both ends of a rotation phase: between two barriers either only rotations or no rotation at all
    code, _ = gen.build_body("alignbit", True, bar_mode="both")
    kinds = set()
    for ins in code:
        if ins.text == "s_barrier":
            assert kinds in ({"slow"}, {"fast"}, set()), kinds
            kinds = set()
        elif ins.cls in ("slow", "fast"):
            kinds.add(ins.cls)
    # the default keeps the one at the end of the rotation phase: every barrier is preceded by a rotation and followed by logic
    code, _ = gen.build_body("alignbit", True)
    valu = [i for i in code if i.cls in ("slow", "fast") or i.text == "s_barrier"]
    for k, ins in enumerate(valu):
        if ins.text == "s_barrier":
            assert valu[k - 1].cls == "slow" and valu[k + 1].cls == "fast"
    assert sum(1 for i in code if i.text == "s_barrier") == 48


def test_committed_header_is_the_generators_output():
    buf = io.StringIO()
    gen.emit_header(buf, "alignbit", True)
    with open(os.path.join(ROOT, "virgo-plus_amd", "csrc", "vp_keccak_asm.h")) as f:
        assert f.read() == buf.getvalue()
