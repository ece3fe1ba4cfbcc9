"""The reason the library is built without packed fp32 arithmetic, kept executable: tools/pkfma_probe (built by __graft_entry__.build())
replays the tail of round 3's faulty build of the forward (tools/pkfma_tail_block.s, verbatim compiler output: 80 v_pk_fma_f32 that form
eight 16-term sums) and compares what reaches memory with exact in-kernel results.  Measured on MI355X (profiles/r04_pkfma_probe.log): with
MFMA running in the other waves of the SIMD the packed stream delivers wrong sums -- always a LOW destination register, always lanes
48..63 -- and never without MFMA neighbours; the same instructions written as v_fma_f32 pairs are never wrong.  The test asserts what must
hold on any box (the scalar control is exact; without MFMA neighbours the packed stream is exact) and REPORTS the packed stream's count
under MFMA (an erratum is allowed to be absent on other silicon)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "_build", "pkfma_probe")


def _run(bg, v0, v1, iters=20000):
    out = subprocess.check_output([PROBE, "replay", str(iters), str(bg), str(v0), str(v1)], timeout=600).decode()
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_packed_fp32_fma_under_mfma_neighbours():
    if not os.path.exists(PROBE):
        import __graft_entry__ as g
        g.build_probe()
    control = _run(1, 15, 16)                         # v_fma_f32 pairs, plain and padded, MFMA neighbours
    assert len(control) == 2 and all(r["lanes_whose_results_differ_from_the_exact_result"] == 0 for r in control), control
    alone = _run(0, 0, 0)                             # the packed block without MFMA neighbours
    assert alone[0]["lanes_whose_results_differ_from_the_exact_result"] == 0, alone
    packed = _run(1, 0, 0) + _run(4, 0, 0)            # the packed block, MFMA / mixed neighbours
    for r in packed:
        print("packed, neighbours bg=%d: %d wrong lane results of %.3g wave executions; by column %s; by 16-lane group %s"
              % (r["bg"], r["lanes_whose_results_differ_from_the_exact_result"], r["wave_executions"], r["wrong_by_column"], r["wrong_by_16_lane_group"]))
        if r["lanes_whose_results_differ_from_the_exact_result"]:
            # the signature: low destination registers (even columns) only, last 16-lane group only
            assert sum(r["wrong_by_column"][1::2]) == 0 and sum(r["wrong_by_16_lane_group"][:3]) == 0, r
