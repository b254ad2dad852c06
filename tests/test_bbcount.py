"""scripts/bbcount.py: the basic-block counters that give profiles/*/dynamic_mix.json."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("bbcount", os.path.join(ROOT, "scripts", "bbcount.py"))
bbcount = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bbcount)

ASM = """\t.file\t1 "/x" "k.hip"
_Z1kv:
; %bb.0:
\t.loc\t1 10 0
\tv_add_f64 v[0:1], v[0:1], v[2:3]
\ts_cmp_eq_u32 s4, 0
\ts_cbranch_scc1 .LBB0_2
; %bb.1:
\t.loc\t1 11 0
\tv_mov_b32_dpp v2, v0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
\tv_cndmask_b32_e32 v3, v2, v1, vcc
.LBB0_2:
\t.loc\t1 12 0
\tds_bpermute_b32 v4, v5, v6
\ts_waitcnt lgkmcnt(0)
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z1kv
\t\t.amdhsa_next_free_vgpr 7
\t\t.amdhsa_next_free_sgpr 6
\t\t.amdhsa_accum_offset 8
\t.end_amdhsa_kernel
"""


def test_classes_and_rewrite_of_a_synthetic_kernel():
    assert [bbcount.klass(o) for o in ("v_fma_f64", "v_fmac_f64_dpp", "v_mov_b32_dpp", "v_readlane_b32", "v_cndmask_b32_e64", "v_cmp_lt_f64_e32",
                                       "v_mov_b64_e32", "v_add_u32_e32", "ds_bpermute_b32", "ds_read_b128", "s_nop", "s_cbranch_vccz", "s_mov_b32")] == \
        ["f64", "dpp_f64", "dpp_mov", "lane_sgpr", "select", "compare", "move", "int_valu", "bpermute", "lds", "wait_nop", "branch", "salu"]
    new, blocks, va = bbcount.rewrite(ASM.split("\n"), "_Z1kv")
    assert [b["label"] for b in blocks] == ["bb.0", "bb.1", ".LBB0_2"] and [b["n"] for b in blocks] == [3, 2, 3]
    assert blocks[0]["classes"] == {"f64": 1, "salu": 1, "branch": 1} and blocks[1]["classes"] == {"dpp_mov": 1, "select": 1}
    assert blocks[2]["lines"] == {"k.hip:12|other": 3}
    text = "\n".join(new)
    # one increment per block, on registers the kernel does not use, at consecutive counters; the descriptor makes room for them
    assert text.count("global_atomic_add") == 3 and f"v[{va}:{va + 1}], v{va + 2}, off offset:-4096" in text and "offset:-4088" in text
    assert va > 6 and ".amdhsa_next_free_sgpr 102" in text and f".amdhsa_next_free_vgpr {va + 3}" in text
    # the original instructions are all still there, in order
    kept = [l.strip() for l in new if l.strip() and l.strip().split()[0] in ("v_add_f64", "s_cmp_eq_u32", "s_cbranch_scc1", "v_mov_b32_dpp", "v_cndmask_b32_e32",
                                                                           "ds_bpermute_b32", "s_waitcnt", "s_endpgm")]
    assert len(kept) == 8 and kept[0].startswith("v_add_f64") and kept[-1] == "s_endpgm"


@pytest.mark.gpu
def test_instrumented_kernel_is_the_same_solver_and_counts_every_pass():
    """The library with the counters solves a small batch to the plain build's bits, and the counters see the work: the evaluation's first block
    runs once per pass (helpers' evaluations included, so at least the owners' passes)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bbcount.py"), "run", "hyb", "cfg1", "96"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "bbcount_hyb_cfg1.json")))
    assert d["same_results_as_plain_build"] is True and d["B"] == 96 and d["passes"] > 0
    res = bbcount.report(os.path.join(ROOT, "gpurun_out", "bbcount_hyb_cfg1.json"), 5, quiet=True)
    assert 600 < res["valu_per_pass"] < 2500 and 0.3 < res["f64_share_of_valu"] < 0.7
    assert max(d["counts"]) >= d["passes"] * 0.9
