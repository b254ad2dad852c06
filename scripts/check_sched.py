"""Validate the machine scheduler's output for the HIP kernels: every virtual-register lane an instruction reads must come from the same
defining instruction after scheduling as before it.

Why this exists ("the max-ilp build"): ROCm 7.2's LLVM, under -amdgpu-sched-strategy=max-ilp, hoists a lane-to-lane COPY inside one
wide virtual register (`%X.sub1 = COPY %X.sub3`, the copy that feeds the second operand of v_permlane32_swap) ABOVE the `undef %X.sub2_sub3 =
V_ADD_F64` that defines its source: the swap then exchanges a stale register.  The machine verifier does not see it (liveness is recomputed
after the move and is self-consistent); the generated code is simply wrong, and reads a register whose content depends on what ran before.

usage: python scripts/check_sched.py [hipcc flags...]      e.g.  -mllvm -amdgpu-sched-strategy=max-ilp   (default: the Makefile's flags)
Compiles csrc/nmpc_kernels.hip for gfx950 with -print-before/-after=machine-scheduler and compares, basic block by basic block, the reaching
definition of every (vreg, lane) read.  Exit status 1 and a listing if any read changed its definition."""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mpc_trajectory_generator_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-Wno-unused-result"]

OPERAND = re.compile(r"(undef |dead |killed |internal |early-clobber |renamable |implicit-def |implicit |def )*%(\d+)(?:\.(sub[0-9_sub]*))?(?::[A-Za-z0-9_]+)?(\(tied-def \d+\))?")


def lanes_of(sub):
    if not sub:
        return None                     # the whole register
    return frozenset(int(x) for x in re.findall(r"sub(\d+)", sub))


def parse_function(lines):
    """-> list of blocks, each a list of (key, defs, uses): key = the instruction's text without flags and slot index,
    defs / uses = [(vreg, lanes or None, undef_flag)]"""
    blocks, cur = [], None
    for ln in lines:
        m = re.match(r"^\d+B\t(.*)$", ln)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("bb."):
            cur = []
            blocks.append(cur)
            continue
        if cur is None or not body.startswith("  "):
            continue
        text = body.strip()
        if text.startswith(("successors:", "liveins:")) or not text:
            continue
        if " = " in text:
            lhs, rhs = text.split(" = ", 1)
        else:
            lhs, rhs = "", text
        rhs_ops = rhs.split(" :: ")[0]
        defs, uses = [], []
        for mm in OPERAND.finditer(lhs):
            defs.append((int(mm.group(2)), lanes_of(mm.group(3)), "undef " in (mm.group(0) or "")))
        for mm in OPERAND.finditer(rhs_ops):
            flags = mm.group(0)
            if "implicit-def" in flags or re.match(r"(\w+ )*def ", flags):
                defs.append((int(mm.group(2)), lanes_of(mm.group(3)), False))
            elif "undef " not in flags:
                uses.append((int(mm.group(2)), lanes_of(mm.group(3))))
        key = re.sub(r"\b(undef|dead|killed|renamable) ", "", text)
        cur.append((key, defs, uses))
    return blocks


def reaching(block):
    """For every instruction of the block (identity = its text without flags + occurrence number): `src` = the instruction that last WROTE
    each (vreg, lane) it reads (-1: live-in) -- what the hardware will deliver, flags ignored -- and `dead` = the lanes it reads although a
    read-undef sub-register def in between has declared them dead (such a read may legally be scheduled anywhere)."""
    seen, wrote, dead_lanes, out = {}, {}, {}, {}
    for key, defs, uses in block:
        n = seen.get(key, 0)
        seen[key] = n + 1
        ident = (key, n)
        src, dead = [], []
        for reg, lanes in uses:
            w = wrote.get(reg, {})
            dl = dead_lanes.get(reg, None)
            if lanes is None:
                src.append((reg, "*", tuple(sorted((str(k), v) for k, v in w.items()))))
            else:
                for l in sorted(lanes):
                    src.append((reg, l, w.get(l, w.get("*", -1))))
                    if dl is not None and l not in dl[0]:
                        dead.append((reg, l, dl[1]))
        out[ident] = (src, dead)
        for reg, lanes, undef in defs:
            w = wrote.setdefault(reg, {})
            if lanes is None:
                w.clear()
                w["*"] = ident
                dead_lanes.pop(reg, None)
            else:
                for l in lanes:
                    w[l] = ident
                if undef:
                    dead_lanes[reg] = (set(lanes), key)     # every other lane is declared dead from here on
                elif reg in dead_lanes:
                    dead_lanes[reg][0].update(lanes)
    return out


def check(dump_text):
    parts = re.split(r"^# \*\*\* IR Dump (Before|After) Machine Instruction Scheduler \(machine-scheduler\) \*\*\*:\n", dump_text, flags=re.M)
    funcs = {}
    for i in range(1, len(parts), 2):
        when, body = parts[i], parts[i + 1]
        m = re.search(r"^# Machine code for function (\S+):", body, re.M)
        if m:
            funcs.setdefault(m.group(1), {})[when] = body.split("\n")
    bad, latent, stats = [], [], []
    for fn, d in funcs.items():
        if "Before" not in d or "After" not in d:
            continue
        bb, ba = parse_function(d["Before"]), parse_function(d["After"])
        if len(bb) != len(ba):
            bad.append((fn, -1, "block count changed", ""))
            continue
        nins = 0
        for bi, (b0, b1) in enumerate(zip(bb, ba)):
            r0, r1 = reaching(b0), reaching(b1)
            nins += len(b0)
            if set(r0) != set(r1):      # the scheduler neither adds nor deletes instructions; flags are stripped from the key
                bad.append((fn, bi, "instruction set of the block changed", str(list(set(r0) ^ set(r1))[:2])[:300]))
                continue
            for ident, (src, dead) in r0.items():
                if r1[ident][0] != src:
                    diff = [(a[:2], str(a[2])[:90], "->", str(b[2])[:90]) for a, b in zip(src, r1[ident][0]) if a != b][:2]
                    bad.append((fn, bi, ident[0][:200], str(diff)))
                for reg, lane, by in dead:
                    latent.append((fn, bi, ident[0][:160], f"%{reg} lane {lane}, declared dead by: {by[:160]}"))
        stats.append((fn, len(bb), nins))
    return stats, bad, latent


def main():
    flags = [f for f in sys.argv[1:] if f != "--strict"]
    if not flags:
        mk = open(os.path.join(CSRC, "Makefile")).read()
        flags = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if "iterative-ilp" in mk else []
    with tempfile.TemporaryDirectory() as td:
        cmd = [HIPCC] + BASE + flags + ["-mllvm", "-print-before=machine-scheduler", "-mllvm", "-print-after=machine-scheduler",
                                        "-o", os.path.join(td, "x.s"), os.path.join(CSRC, "nmpc_kernels.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(2)
        stats, bad, latent = check(r.stderr)
    for fn, nb, ni in stats:
        print(f"checked {fn[:70]:70s} {nb:4d} blocks {ni:6d} instructions")
    if latent:
        print(f"{len(latent)} reads of lanes that a read-undef sub-register def has declared dead (latent: the scheduler may reorder them):")
        for fn, bi, key, what in latent[:6]:
            print(f"  {fn[:60]} block {bi}: {what}\n      read by: {key}")
    if bad:
        print(f"{len(bad)} reads changed their reaching definition (scheduler bug: the build is WRONG):")
        for fn, bi, key, diff in bad[:16]:
            print(f"  {fn[:60]} block {bi}: {key}\n      {diff}")
        sys.exit(1)
    if latent and "--strict" in sys.argv:
        sys.exit(1)
    print("SCHED_OK: every read keeps its reaching definition")


if __name__ == "__main__":
    main()
