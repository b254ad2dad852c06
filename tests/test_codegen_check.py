"""The build gate of DESIGN.md section 5.8 (mpc_trajectory_generator_amd/codegen_check.py) against the machine code that made it necessary:
two excerpts of LLVM MIR dumps of this project's own earlier sources (tests/golden/mir_*.txt, a dozen instructions each) -- the minimal
cases of the two compiler defects -- and their repaired counterparts.  CPU only, no compilation."""
import os
import re

from mpc_trajectory_generator_amd import codegen_check as cc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _read(name):
    return open(os.path.join(GOLDEN, name)).read()


def test_scheduler_check_flags_the_hoisted_lane_copy():
    text = _read("mir_sched_permlane_swap.txt")
    stats, bad, latent = cc.check_scheduler(text)
    assert len(stats) == 1 and stats[0][2] == 7                       # one function, seven instructions
    moved = [b for b in bad if "COPY %26673.sub3" in b[2]]
    assert moved, bad                                                  # the copy reads sub3 from another definition after scheduling
    assert any("V_PERMLANE32_SWAP" in l[3] for l in latent), latent    # and the read-undef flag that permitted it is reported by itself


def test_scheduler_check_accepts_an_order_preserving_schedule():
    text = _read("mir_sched_permlane_swap.txt")
    head, before, after = re.split(r"^# \*\*\* IR Dump (?:Before|After) Machine Instruction Scheduler \(machine-scheduler\) \*\*\*:\n", text, flags=re.M)
    # without the wrong flag on the swap's tied def, and scheduled in the original order: nothing to report
    clean = before.replace("undef %26673.sub0:vreg_128_align2 = V_PERMLANE32_SWAP", "%26673.sub0:vreg_128_align2 = V_PERMLANE32_SWAP")
    mk = lambda when, body: f"# *** IR Dump {when} Machine Instruction Scheduler (machine-scheduler) ***:\n" + body
    stats, bad, latent = cc.check_scheduler(mk("Before", clean) + mk("After", clean))
    assert len(stats) == 1 and not bad and not latent
    # the flag alone (no instruction moved yet) is already a finding: any scheduler may use it
    stats, bad, latent = cc.check_scheduler(mk("Before", before) + mk("After", before))
    assert not bad and latent


def test_exec_restore_check_flags_vector_copies_in_front_of_the_restore():
    text = _read("mir_exec_restore_copies.txt")
    nfun, nrestore, hits = cc.check_exec_restores(text)
    assert nfun == 1 and nrestore == 1
    assert [h[2].split(" = ")[0].split()[-1] for h in hits] == ["$agpr106_agpr107", "$agpr92_agpr93", "$agpr48_agpr49"]
    # the same block with the copies behind the restore (where the allocator belongs) passes; the SGPR copy in front of it is fine
    lines = text.split("\n")
    copies = [l for l in lines if re.search(r"\$agpr\d+_agpr\d+ = COPY", l)]
    rest = [l for l in lines if l not in copies]
    k = next(i for i, l in enumerate(rest) if "S_OR_B64 $exec" in l)
    fixed = "\n".join(rest[:k + 1] + copies + rest[k + 1:])
    nfun, nrestore, hits = cc.check_exec_restores(fixed)
    assert nrestore == 1 and not hits


def test_dpp_hazard_check_counts_wait_states():
    """check 3: a DPP read needs two wait states after a vector write of its source; s_nop N gives N + 1, any instruction gives one."""
    asm = """_Z3foov:
\tv_add_f64 v[6:7], v[6:7], v[2:3]
\ts_mov_b32 exec_hi, s0
\tv_mov_b32_dpp v2, v6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
\tv_add_f64 v[8:9], v[6:7], v[2:3]
\ts_nop 1
\tv_mov_b32_dpp v2, v9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
\tv_add_f64 v[8:9], v[6:7], v[2:3]
\ts_nop 0
\tv_fmac_f64_dpp v[10:11], v[8:9], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf
\tv_add_f64 v[8:9], v[6:7], v[2:3]
\ts_mov_b32 exec_hi, s0
\ts_nop 0
\tv_mov_b32_dpp v2, v8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
.Lfunc_end0:
"""
    n, bad = cc.check_dpp_hazards(asm)
    assert n == 4 and [b[1].split()[0] for b in bad] == ["v_mov_b32_dpp", "v_fmac_f64_dpp"] and "v6" in bad[0][1]


def test_dpp_hazard_check_on_the_pipelined_recurrence_step():
    """The software-pipelined Gram recurrences (nmpc_solve_hyb.h, gram_fwd_step / gram_bwd_step) carry no s_nop: the direction updates of the step
    before fill the two wait states between a coefficient's multiply (add) and its DPP read.  The shape as written passes; with one filler
    missing it is flagged."""
    fwd = """_Z3foov:
\tv_mul_f64 v[20:21], v[30:31], v[2:3]
\tv_fmac_f64_dpp v[6:7], -v[22:23], v[40:41] row_newbcast:2 row_mask:0xf bank_mask:0xf
\tv_fmac_f64_dpp v[8:9], -v[22:23], v[42:43] row_newbcast:2 row_mask:0xf bank_mask:0xf
\tv_fmac_f64_dpp v[2:3], -v[20:21], v[44:45] row_newbcast:3 row_mask:0xf bank_mask:0xf
\tv_fmac_f64_dpp v[4:5], -v[20:21], v[46:47] row_newbcast:3 row_mask:0xf bank_mask:0xf
.Lfunc_end0:
"""
    n, bad = cc.check_dpp_hazards(fwd)
    assert n == 4 and not bad
    bwd_short = """_Z3barv:
\tv_mul_f64 v[20:21], v[30:31], v[4:5]
\tv_add_f64 v[20:21], v[32:33], -v[20:21]
\tv_fmac_f64_dpp v[6:7], v[22:23], v[40:41] row_newbcast:4 row_mask:0xf bank_mask:0xf
\tv_fmac_f64_dpp v[4:5], v[20:21], v[44:45] row_newbcast:3 row_mask:0xf bank_mask:0xf
.Lfunc_end1:
"""
    n, bad = cc.check_dpp_hazards(bwd_short)
    assert n == 2 and len(bad) == 1 and "v[20:21]" in bad[0][1]

