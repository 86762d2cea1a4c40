"""Build-time contract of the hand-written DDA loops (tools/check_kernel_contract.py), on the gfx950
assembly of the default build: m0 is touched by nothing but the loops' iteration counter, and the
two live frame kernels keep <= 64 vector / <= 96 scalar registers (8 wavefronts per SIMD) without
spills.  `make gpu` and `make variants` run the same check on every library they produce."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_m0_and_occupancy_contract_of_the_default_build():
    import check_kernel_contract as c
    problems, facts = c.check(c.device_assembly())
    assert not problems, problems
    assert facts["m0_counter_instructions"] > 0 and facts["loop_instructions_on_v47_v55"] > 0
    for k in c.LIVE:
        assert facts[k]["vgpr"] <= 64 and facts[k]["sgpr"] <= 96 and facts[k]["vgpr_spills"] == 0


def test_the_check_notices_a_broken_contract():
    import check_kernel_contract as c
    meta = ("    .name:           _ZN2vk13render_framesILb%dELb0ELi%dEEEvX\n    .private_segment_fixed_size: 0\n    .sgpr_count:     %d\n"
            "    .sgpr_spill_count: 0\n    .symbol: x\n    .vgpr_count:     %d\n    .vgpr_spill_count: 0\n")
    loop = "\t;;#ASMSTART\n\ts_mov_b32 m0, 0\n\tv_add_f32 v47, v1, v2\n\ts_add_u32 m0, m0, 1\n\t;;#ASMEND\n"
    naive = meta % (1, 32, 70, 56) + meta % (0, 32, 78, 56)
    ok = loop + meta % (1, 8, 70, 56) + meta % (0, 8, 78, 56) + naive
    assert c.check(ok)[0] == []
    assert any("m0 is used" in p for p in c.check(ok + "\ts_movrels_b32 s3, m0\n")[0])
    assert any("vector registers" in p for p in c.check(loop + meta % (1, 8, 70, 72) + meta % (0, 8, 78, 56) + naive)[0])
    assert any("scalar registers" in p for p in c.check(loop + meta % (1, 8, 70, 56) + meta % (0, 8, 100, 56) + naive)[0])
    assert any("not found" in p for p in c.check(loop + meta % (1, 8, 70, 56) + naive)[0])
    assert any("loops are gone" in p for p in c.check(meta % (1, 8, 70, 56) + meta % (0, 8, 78, 56) + naive)[0])
