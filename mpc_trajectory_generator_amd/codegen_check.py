"""Build-time validation of the generated gfx950 code: two classes of wrong code that ROCm 7.2's LLVM produces for these kernels and that
neither the compiler nor its machine verifier reports.  `verify()` compiles csrc/nmpc_kernels.hip once more with the
flags of the real build plus machine-code dumps and checks

The third check reads the final assembly: every DPP read keeps its two wait states from the last vector write of its source (check_dpp_hazards).

1. the machine scheduler: every virtual-register lane an instruction reads must come from the same defining instruction after scheduling
   as before it.  (-amdgpu-sched-strategy=max-ilp hoisted the lane copy `%X.sub1 = COPY %X.sub3` that feeds the second operand of a
   v_permlane32_swap above the `V_ADD_F64` producing its source, because the register coalescer had left a read-undef flag on the swap's tied
   second def that declares the other lanes of %X dead; the swap then exchanged a stale register.)  Reads of lanes so declared dead are
   reported too ("latent"): any scheduler may move them.
2. the register allocator: no vector instruction in front of a basic block's EXEC restore.  (Splitting live ranges of the 370-register
   two-stage kernel, the greedy allocator inserted VGPR -> AGPR copies at the top of an `endif` block IN FRONT of `s_or_b64 exec, exec, s[a:b]`
   when an SGPR copy already sat there: the copy saves only the lanes of the `if` side -- none when the block is reached through
   s_cbranch_execz -- and the later reload returns whatever the AGPR held before: results that depend on what ran on the SIMD earlier and
   change from run to run.  Found with tests/scrub + scripts/scrub_bisect.py.)

`build_library()` (_lib.py) calls verify() after every real compilation and refuses the library if either check fails.
CLI:  python -m mpc_trajectory_generator_amd.codegen_check [--src file.hip] [hipcc flags ...]     (default: the library's source, the Makefile's flags)"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.environ.get("NMPC_CSRC") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-Wno-unused-result"]

# ------------------------------------------------------------------------------------------------- 1. the machine scheduler
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


def check_scheduler(dump_text):
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


# ------------------------------------------------------------------------------------------------- 2. the register allocator
VECTOR = re.compile(r"^(V_|DS_|GLOBAL_|BUFFER_|SCRATCH_|FLAT_|SI_SPILL_V|SI_SPILL_A|SI_SPILL_AV)")
IGNORES_EXEC = re.compile(r"^(V_READLANE_B32|V_WRITELANE_B32|V_READFIRSTLANE_B32|SI_SPILL_S)")      # SGPR spill traffic: independent of EXEC
RESTORE = re.compile(r"\$exec = S_OR_B64(_term)? \$exec,")


def check_exec_restores(text):
    """MIR after the register allocator has rewritten virtual registers (-print-after=virtregrewriter): in every basic block, the vector
    instructions (and copies into VGPRs / AGPRs) in front of the block's EXEC restore.  -> [(function, block, instruction)]"""
    hits, nfun, nrestore = [], 0, 0
    fn, block, pending = None, None, None
    for raw in text.split("\n"):
        m = re.match(r"^# Machine code for function (\S+):", raw)
        if m:
            fn, nfun = m.group(1), nfun + 1
            continue
        m = re.match(r"^(?:\d+B\t)?(bb\.\d+)", raw)
        if m:
            block, pending = m.group(1), []
            continue
        if pending is None:
            continue
        m = re.match(r"^(?:\d+B)?\t\s+(.*)$", raw)
        if not m:
            continue
        ins = m.group(1).strip()
        if ins.startswith(("successors:", "liveins:", "; predecessors")):
            continue
        if RESTORE.search(ins):
            nrestore += 1
            for p in pending:
                hits.append((fn, block, p))
            pending = None
            continue
        rhs = ins.split(" = ", 1)[1] if " = " in ins else ins
        rhs = re.sub(r"^((nofpexcept|nnan|ninf|nsz|arcp|contract|afn|reassoc|nuw|nsw|exact|disjoint|samesign|frame-setup|frame-destroy) )+", "", rhs)
        op = rhs.split()[0] if rhs.split() else ""
        if op == "COPY":
            if re.match(r"^(renamable |dead |undef |early-clobber )*\$(vgpr|agpr)", ins):
                pending.append(ins[:160])
        elif VECTOR.match(op) and not IGNORES_EXEC.match(op):
            pending.append(ins[:160])
        elif op in ("S_CBRANCH_EXECZ", "S_CBRANCH_EXECNZ", "S_BRANCH", "S_ENDPGM") or op.startswith("S_CBRANCH"):
            pending = None
    return nfun, nrestore, hits


# ------------------------------------------------------------------------------------------------- 3. DPP reads behind inline asm
def _vregs(tok):
    """vector registers an operand names: 'v5' -> {5}, 'v[4:5]' -> {4, 5}; anything else -> {}"""
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_dpp_hazards(asm_text):
    """A DPP instruction reads its first source from OTHER lanes' registers: the hardware needs two wait states between a vector-ALU write
    of that register and the read, and does not interlock.  The compiler inserts them for its own instructions but cannot see into an asm
    statement -- and this library's DPP arithmetic (v_fmac_f64_dpp row_newbcast) and its EXEC windows live in asm statements.  So every DPP
    read of the FINAL code is checked: no vector write of its DPP source within the two preceding wait states (s_nop N counts N + 1; a
    branch target or a branch ends the look-back: the compiler's own nops cover its control-flow edges).  -> (dpp instructions, violations)"""
    ndpp, bad = 0, []
    fn = None
    hist = []           # (wait states the instruction occupies, registers it writes if it is a vector-ALU instruction)
    for raw in asm_text.split("\n"):
        t = raw.strip()
        m = re.match(r"^(_Z\w+|nmpc_\w+):", t)
        if m:
            fn, hist = m.group(1), []
            continue
        if t.startswith(".Lfunc_end"):
            fn = None
            continue
        if fn is None or not t or t.startswith(";") or t.startswith("."):
            if fn and re.match(r"^\.LBB\d+_\d+:", t):
                hist = []
            continue
        t = re.sub(r"\s*;.*$", "", t)
        op = t.split()[0]
        ops = [x.strip() for x in t[len(op):].split(",")]
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            hist = []
            continue
        if op.endswith("_dpp") or " row_" in t or " quad_perm" in t or " wave_sh" in t or " row_newbcast" in t:
            ndpp += 1
            src = _vregs(ops[1].split()[0]) if len(ops) > 1 else set()
            ws = 0
            for states, wr in reversed(hist):
                if ws >= 2:
                    break
                if wr & src:
                    bad.append((fn, t[:120]))
                    break
                ws += states
        if op == "s_nop":
            hist.append((int(ops[0], 0) + 1, set()))
        elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            wr = _vregs(ops[0].split()[0])
            if op.startswith("v_permlane") and len(ops) > 1:      # the swaps rewrite both operands
                wr |= _vregs(ops[1].split()[0])
            hist.append((1, wr))
        else:
            hist.append((1, set()))
        hist = hist[-4:]
    return ndpp, bad


# ------------------------------------------------------------------------------------------------- driver
def makefile_flags():
    """the code-generation flags csrc/Makefile builds with (its scheduler strategy)"""
    r = subprocess.run(["make", "-s", "-C", CSRC, "print-sched"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("make print-sched failed: " + r.stderr[-500:])
    return r.stdout.split()


def kernel_resources(asm_text):
    """per kernel: registers, LDS, scratch and spill counts from the code object metadata of the assembly"""
    out = {}
    for blk in asm_text.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, None])[1]      # noqa: E731
        name = g("name")
        if name:
            out[name] = {"vgpr_total": int(g("vgpr_count") or 0), "agpr": int(blk.split("\n")[0].strip() or 0), "sgpr": int(g("sgpr_count") or 0),
                         "scratch_bytes": int(g("private_segment_fixed_size") or 0), "static_lds_bytes": int(g("group_segment_fixed_size") or 0),
                         "vgpr_spills": int(g("vgpr_spill_count") or 0), "sgpr_spills": int(g("sgpr_spill_count") or 0)}
    return out


def verify(flags=None, src=None):
    """-> dict(ok, flags, kernels, sched_changed, sched_latent, exec_hits, details, resources).  Two extra compilations side by side (about 25 s).
    `src`: another source file than the library's (the minimal cases under tests/repro/)."""
    flags = makefile_flags() if flags is None else list(flags)
    src = src or os.path.join(CSRC, "nmpc_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        cmds = [[HIPCC] + BASE + flags + ["-mllvm", "-print-before=machine-scheduler", "-mllvm", "-print-after=machine-scheduler", "-o", os.path.join(td, "x.s"), src],
                [HIPCC] + BASE + flags + ["-mllvm", "-print-after=virtregrewriter", "-o", os.path.join(td, "y.s"), src]]
        errs = [open(os.path.join(td, f"err{i}.txt"), "w+") for i in range(2)]
        procs = [subprocess.Popen(c, stdout=subprocess.DEVNULL, stderr=e) for c, e in zip(cmds, errs)]
        rcs = [p.wait() for p in procs]
        texts = []
        for e in errs:
            e.seek(0)
            texts.append(e.read())
            e.close()
        if any(rcs):
            return {"ok": False, "flags": flags, "error": (texts[0][-1500:] + texts[1][-1500:])}
        asm_text = open(os.path.join(td, "x.s")).read()
        resources = kernel_resources(asm_text)
    ndpp, dpp_bad = check_dpp_hazards(asm_text)
    stats, bad, latent = check_scheduler(texts[0])
    nfun, nrestore, hits = check_exec_restores(texts[1])
    # strategies that move instructions between blocks (the default one rematerialises) are outside what check 1 can compare
    moved = [b for b in bad if b[2] == "instruction set of the block changed"]
    bad = [b for b in bad if b[2] != "instruction set of the block changed"]
    return {"ok": not bad and not latent and not hits and not dpp_bad, "flags": flags, "dpp_reads": ndpp, "dpp_hazards": len(dpp_bad), "kernels": len(stats), "instructions": sum(s[2] for s in stats),
            "blocks_not_comparable": len(moved), "sched_changed": len(bad), "sched_latent": len(latent), "exec_restores": nrestore,
            "exec_hits": len(hits),
            "details": [f"{b[0][:60]} block {b[1]}: {b[2]} {b[3]}" for b in bad[:8]] +
                       [f"latent {l[0][:60]} block {l[1]}: {l[3]} read by {l[2]}" for l in latent[:8]] +
                       [f"exec {h[0][:60]} {h[1]}: {h[2]}" for h in hits[:8]] +
                       [f"dpp {h[0][:60]}: {h[1]}" for h in dpp_bad[:8]],
            "resources": resources}


def main():
    argv, src = sys.argv[1:], None
    if "--src" in argv:
        i = argv.index("--src"); src = argv[i + 1]; argv = argv[:i] + argv[i + 2:]
    res = verify(argv or None, src)
    print({k: v for k, v in res.items() if k not in ("details", "resources")})
    for d in res.get("details", []):
        print("  " + d[:400])
    if "error" in res:
        print(res["error"])
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
