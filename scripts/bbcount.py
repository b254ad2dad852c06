"""Dynamic instruction mix of a solve kernel: executions of every basic block, counted on the GPU, times the block's static instruction classes.

There is no thread-trace decoder in this image, so the counters are put into the compiler's own output: `build` compiles
csrc/nmpc_kernels.hip to gfx950 assembly with the flags of the shipped library (+ line tables, which do not change the code: checked),
inserts at the head of every basic block of ONE kernel

    s_mov_b64 s[100:101], exec ; s_mov_b64 exec, 1 ; global_atomic_add v[a:a+1], v(a+2), off offset:4*block-4096 ; s_mov_b64 exec, s[100:101]

(s100/s101 and the three VGPRs are registers the kernel does not use; SCC, VCC and every waitcnt stay valid: extra outstanding
operations only make a wait longer), assembles, links and wraps the code object into csrc/variants/libnmpc_bbcount_<tag>.so
together with the block table (classes and source lines per block).  `run` solves a batch with that library on the GPU and stores
the counters; `report` multiplies the two.

  python scripts/bbcount.py build [hyb|hyb2] [tag]                 (CPU: cross-compiles)
  python scripts/bbcount.py run <tag> <cfg> [B] [instance]         (GPU)  -> gpurun_out/bbcount_<tag>_<cfg>[_i<instance>].json
  python scripts/bbcount.py report <counts.json> [top]             (CPU)
  python scripts/bbcount.py collect <out.json> <counts.json> ...   (CPU)  -> profiles/<round>/dynamic_mix.json (bench.py quotes it by source hash)
"""
import json
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "mpc_trajectory_generator_amd", "csrc")
VAR = os.path.join(CSRC, "variants")
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = "/opt/rocm/bin/hipcc"
KERNELS = {"hyb": "_ZN4nmpc21nmpc_solve_hyb_kernelINS_12ShapeDefaultEEEvNS_5KArgsE",
           "hyb50": "_ZN4nmpc21nmpc_solve_hyb_kernelINS_11ShapeNobs50EEEvNS_5KArgsE",
           "hyb2": "_ZN4nmpc22nmpc_solve_hyb2_kernelINS_8ShapeN40EEEvNS_5KArgsE"}
SYM = "_ZN4nmpc10nmpc_bbcntE"
NSLOT = 2048


def klass(op):
    """instruction class of an opcode (the classes the verdict of round 5 asks for)"""
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
        if op.endswith("dpp"):
            return "dpp_f64" if "_f64" in base else "dpp_mov"
        if base.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "lane_sgpr"
        if base.startswith("v_permlane"):
            return "permlane"
        if base.startswith("v_cndmask"):
            return "select"
        if base.startswith("v_cmp"):
            return "compare"
        if base.startswith(("v_mov_b", "v_accvgpr")):
            return "move"
        if "_f64" in base:
            return "f64"
        return "int_valu"
    if op.startswith("ds_bpermute"):
        return "bpermute"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache", "s_atomic")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


VALU = ("f64", "dpp_f64", "dpp_mov", "lane_sgpr", "permlane", "select", "compare", "move", "int_valu")


def sched_flags():
    return subprocess.run(["make", "-s", "-C", CSRC, "print-sched"], capture_output=True, text=True, check=True).stdout.split()


def base_flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"] + sched_flags() + \
           ["-DNMPC_EXPERIMENTS", "-DNMPC_BBCOUNT"]


def kernel_span(lines, name):
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return start, end


def instruction_stream(lines, name):
    s, e = kernel_span(lines, name)
    out = []
    for l in lines[s + 1:e]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        out.append(re.sub(r"\s*;.*$", "", t))
    return out


def rewrite(lines, name):
    """-> (new lines, blocks): blocks[i] = {"label", "classes", "ops", "lines": {file:line: n}}"""
    s, e = kernel_span(lines, name)
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    body = lines[s + 1:e]
    used = "\n".join(body)
    assert not re.search(r"\bs10[01]\b|s\[100:101\]", used), "the kernel uses s100 / s101"
    vmax = max(int(x) for x in re.findall(r"\bv(\d+)\b", used))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", used):
        vmax = max(vmax, int(m.group(2)))
    va = (vmax + 2) & ~1          # even: a 64-bit address pair
    amax = max([int(x) for x in re.findall(r"\ba(\d+)\b", used)] + [int(m.group(2)) for m in re.finditer(r"\ba\[(\d+):(\d+)\]", used)] + [-1])
    agpr_mode = va + 2 >= 256     # every vector register is in use (the two-stage kernel): lane 0 of v0..v2 is parked in spare AGPRs around each increment
    if agpr_mode:
        assert vmax == 255 and amax + 6 < 256, "no spare registers at all"
        a0 = amax + 1             # a0..a0+2: parking; a0+3, a0+4: the counters' base address
        out = [lines[s],
               "\ts_getpc_b64 s[100:101]", f"\ts_add_u32 s100, s100, {SYM}@rel32@lo+4", f"\ts_addc_u32 s101, s101, {SYM}@rel32@hi+12",
               "\ts_add_u32 s100, s100, 4096", "\ts_addc_u32 s101, s101, 0",
               f"\tv_accvgpr_write_b32 a{a0 + 3}, s100", f"\tv_accvgpr_write_b32 a{a0 + 4}, s101"]
    else:
        out = [lines[s],
               f"\ts_getpc_b64 s[100:101]", f"\ts_add_u32 s100, s100, {SYM}@rel32@lo+4", f"\ts_addc_u32 s101, s101, {SYM}@rel32@hi+12",
               "\ts_add_u32 s100, s100, 4096", "\ts_addc_u32 s101, s101, 0",
               f"\tv_mov_b32_e32 v{va}, s100", f"\tv_mov_b32_e32 v{va + 1}, s101", f"\tv_mov_b32_e32 v{va + 2}, 1"]
    blocks = []
    cur = None
    label = "entry"
    loc = None
    new_block = True

    def open_block():
        nonlocal cur
        bid = len(blocks)
        assert bid < NSLOT, "more basic blocks than counters"
        cur = {"label": label, "classes": Counter(), "ops": Counter(), "n": 0, "lines": Counter()}
        blocks.append(cur)
        if agpr_mode:
            out.extend(["\ts_mov_b64 s[100:101], exec", "\ts_mov_b64 exec, 1",
                        f"\tv_accvgpr_write_b32 a{a0}, v0", f"\tv_accvgpr_write_b32 a{a0 + 1}, v1", f"\tv_accvgpr_write_b32 a{a0 + 2}, v2",
                        f"\tv_accvgpr_read_b32 v0, a{a0 + 3}", f"\tv_accvgpr_read_b32 v1, a{a0 + 4}", "\tv_mov_b32_e32 v2, 1", "\ts_nop 0",
                        f"\tglobal_atomic_add v[0:1], v2, off offset:{4 * bid - 4096}", "\ts_nop 1",
                        f"\tv_accvgpr_read_b32 v0, a{a0}", f"\tv_accvgpr_read_b32 v1, a{a0 + 1}", f"\tv_accvgpr_read_b32 v2, a{a0 + 2}",
                        "\ts_mov_b64 exec, s[100:101]"])
        else:
            out.extend(["\ts_mov_b64 s[100:101], exec", "\ts_mov_b64 exec, 1",
                        f"\tglobal_atomic_add v[{va}:{va + 1}], v{va + 2}, off offset:{4 * bid - 4096}", "\ts_mov_b64 exec, s[100:101]"])

    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            label, new_block = m.group(1), True
            out.append(l)
            continue
        m = re.match(r"^; %bb\.(\d+):", t)
        if m:
            label, new_block = "bb." + m.group(1), True
            out.append(l)
            continue
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = f"{files.get(int(m.group(1)), m.group(1))}:{m.group(2)}"
            out.append(l)
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            out.append(l)
            continue
        if new_block:
            open_block()
            new_block = False
        op = t.split()[0]
        k = klass(op)
        cur["classes"][k] += 1
        cur["ops"][re.sub(r"_(e32|e64)$", "", op)] += 1
        cur["n"] += 1
        if loc:
            cur["lines"][loc + "|" + ("valu" if k in VALU else "other")] += 1
        out.append(l)
        if k == "branch":
            label, new_block = cur["label"] + "+", True
    new = lines[:s] + out + lines[e:]
    # the kernel descriptor: three more vector registers, s100 / s101
    text = "\n".join(new)
    pat = re.compile(r"(\.amdhsa_kernel " + re.escape(name) + r"\n.*?\.end_amdhsa_kernel)", re.S)
    desc = pat.search(text).group(1)
    if agpr_mode:
        d2 = re.sub(r"\.amdhsa_next_free_vgpr \d+", f".amdhsa_next_free_vgpr {256 + a0 + 5}", desc)      # unified file: 256 vector + accumulation registers
        va = -a0
    else:
        d2 = re.sub(r"\.amdhsa_next_free_vgpr \d+", f".amdhsa_next_free_vgpr {va + 3}", desc)
        d2 = re.sub(r"\.amdhsa_accum_offset \d+", f".amdhsa_accum_offset {(va + 3 + 3) & ~3}", d2)
        assert ((va + 3 + 3) & ~3) <= 256
    d2 = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", d2)
    text = text.replace(desc, d2)
    # (the metadata's register counts: what the loader reports, not what it allocates -- left alone)
    for b in blocks:
        b["classes"] = dict(b["classes"])
        b["ops"] = dict(b["ops"])
        b["lines"] = dict(b["lines"])
    return text.split("\n"), blocks, va


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise SystemExit("FAILED: " + " ".join(cmd) + "\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return r


def build(which, tag):
    name = KERNELS[which]
    os.makedirs(VAR, exist_ok=True)
    wd = os.path.join("/tmp", f"bbcount_{tag}")
    os.makedirs(wd, exist_ok=True)
    src = os.path.join(CSRC, "nmpc_kernels.hip")
    fl = base_flags()
    p0 = subprocess.Popen([HIPCC] + fl + ["-S", "--cuda-device-only", "-o", f"{wd}/plain.s", src], stderr=subprocess.DEVNULL)
    run([HIPCC] + fl + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", f"{wd}/lines.s", src])
    assert p0.wait() == 0
    a = open(f"{wd}/plain.s").read().split("\n")
    b = open(f"{wd}/lines.s").read().split("\n")
    same = instruction_stream(a, name) == instruction_stream(b, name)
    print("line tables leave the kernel's code unchanged:", same)
    assert same
    new, blocks, va = rewrite(b, name)
    open(f"{wd}/mod.s", "w").write("\n".join(new))
    run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", f"{wd}/mod.s", "-o", f"{wd}/mod.o"])
    run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", f"{wd}/mod.out", f"{wd}/mod.o"])
    run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", f"-input={wd}/mod.out", f"-output={wd}/mod.hipfb"])
    out = os.path.join(VAR, f"libnmpc_bbcount_{tag}.so")
    run([HIPCC] + fl + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", f"{wd}/mod.hipfb", "-shared", "-o", out, src])
    from mpc_trajectory_generator_amd import _lib
    meta = {"kernel": name, "which": which, "source_hash": _lib.source_hash(), "flags": fl, "spare_vgprs": [va, va + 1, va + 2], "blocks": blocks}
    json.dump(meta, open(out[:-3] + ".json", "w"))
    nins = sum(b["n"] for b in blocks)
    print(f"{out}: {len(blocks)} blocks, {nins} instructions, counters on v{va}..v{va + 2}, s[100:101]")


def gpu_run(tag, cfgname, B, inst):
    import ctypes as C
    import numpy as np
    lib_path = os.path.join(VAR, f"libnmpc_bbcount_{tag}.so")
    from mpc_trajectory_generator_amd import _lib as _l
    if not os.path.exists(lib_path) or json.load(open(lib_path[:-3] + ".json")).get("source_hash") != _l.source_hash():
        build(json.load(open(lib_path[:-3] + ".json"))["which"] if os.path.exists(lib_path[:-3] + ".json") else tag, tag)      # (built for other sources)
    os.environ["NMPC_LIB_PATH"] = lib_path
    from mpc_trajectory_generator_amd import named_config, _lib
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    cfg = named_config(cfgname)
    full = 8192
    sol = BatchSolver(cfg, max_batch=max(full, B))
    P = synthetic_batch(cfg, 11, full, 0, routes=random_routes(cfg, 11, 32, seed=1000))
    if inst is not None:
        P = P[inst:inst + 1]
    else:
        P = P[:B]
    lib = _lib.load_library()
    lib.nmpc_debug_bbcount.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(4096, dtype=np.uint32)
    sol.solve(P)                                            # warm-up (and a check that the instrumented code terminates)
    assert lib.nmpc_debug_bbcount(buf.ctypes.data, 1) == 0
    u, y, st = sol.solve(P)
    ms = sol.last_batch_ms
    assert lib.nmpc_debug_bbcount(buf.ctypes.data, 1) == 0
    # the instrumented kernel must still be the same solver: compare with the plain experiments build
    os.environ.pop("NMPC_LIB_PATH")
    sol2 = BatchSolver(cfg, max_batch=max(full, B), experiments=True)
    u2, y2, st2 = sol2.solve(P)
    ref = bool(np.array_equal(u, u2) and np.array_equal(y, y2) and all(np.array_equal(st[f], st2[f]) for f in
               ("exit_status", "num_inner_iterations", "num_cost_evals", "num_grad_evals", "reserved", "cost", "penalty")))
    out = {"tag": tag, "config": cfgname, "B": int(P.shape[0]), "instance": inst, "kernel": sol.kernel_name, "ms_instrumented": ms,
           "passes": int(st["reserved"].astype(np.int64).sum()), "inner_iterations": int(st["num_inner_iterations"].astype(np.int64).sum()),
           "evals": int((st["num_cost_evals"].astype(np.int64) + st["num_grad_evals"]).sum()),
           "same_results_as_plain_build": ref, "counts": buf[:NSLOT].tolist()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    fn = os.path.join(ROOT, "gpurun_out", f"bbcount_{tag}_{cfgname}" + (f"_i{inst}" if inst is not None else "") + ".json")
    json.dump(out, open(fn, "w"))
    print(fn, "passes", out["passes"], "ms", ms, "same results:", ref, "blocks hit", int((buf > 0).sum()))


def report(fn, top=40, quiet=False):
    d = json.load(open(fn))
    meta = json.load(open(os.path.join(VAR, f"libnmpc_bbcount_{d['tag']}.json")))
    blocks, counts, passes = meta["blocks"], d["counts"], d["passes"]
    tot = Counter()
    ops = Counter()
    by_line = defaultdict(lambda: [0.0, 0.0])
    rows = []
    for b, c in zip(blocks, counts):
        if not c:
            continue
        for k, v in b["classes"].items():
            tot[k] += v * c
        for k, v in b.get("ops", {}).items():
            ops[k] += v * c
        valu = sum(v for k, v in b["classes"].items() if k in VALU)
        rows.append((valu * c, c, b))
        for key, n in b["lines"].items():
            loc, kind = key.split("|")
            by_line[loc][0 if kind == "valu" else 1] += n * c
    per_pass = {k: round(v / passes, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
    valu_pp = sum(v for k, v in tot.items() if k in VALU) / passes
    res = {"config": d["config"], "B": d["B"], "instance": d["instance"], "kernel": d["kernel"], "source_hash": meta["source_hash"],
           "passes": passes, "per_pass": per_pass, "valu_per_pass": round(valu_pp, 1),
           "f64_share_of_valu": round((tot["f64"] + tot["dpp_f64"]) / max(1, sum(v for k, v in tot.items() if k in VALU)), 3),
           "all_instructions_per_pass": round(sum(tot.values()) / passes, 1), "same_results_as_plain_build": d.get("same_results_as_plain_build")}
    res["top_opcodes_per_pass"] = {k: round(v / passes, 2) for k, v in ops.most_common(60)}
    rows.sort(key=lambda r: -r[0])
    res["top_blocks"] = []
    for w, c, b in rows[:top]:
        own = sorted((k.split("|")[0] for k in b["lines"]), key=lambda s: (s.split(":")[0], int(s.split(":")[1])))
        hyb = [s for s in own if s.startswith("nmpc_solve_hyb")] or [s for s in own if s.startswith("nmpc_kernels")] or own
        res["top_blocks"].append({"block": b["label"], "executions_per_pass": round(c / passes, 3), "valu_per_pass": round(w / passes, 1),
                                  "static": b["classes"], "lines": (hyb[0] + " .. " + hyb[-1]) if hyb else ""})
    lines = sorted(by_line.items(), key=lambda kv: -kv[1][0])[:top]
    res["top_lines"] = [{"line": k, "valu_per_pass": round(v[0] / passes, 1), "other_per_pass": round(v[1] / passes, 1)} for k, v in lines]
    if not quiet:
        print(json.dumps({k: v for k, v in res.items() if k not in ("top_blocks", "top_lines", "top_opcodes_per_pass")}, indent=1))
        print("opcodes per pass: " + "  ".join(f"{k} {v}" for k, v in res["top_opcodes_per_pass"].items()))
        for r in res["top_blocks"]:
            print(f"{r['block']:14s} x{r['executions_per_pass']:7.3f} valu/pass {r['valu_per_pass']:7.1f}  {r['lines']:60s} " +
                  " ".join(f"{k}={v}" for k, v in sorted(r["static"].items()) if k in VALU))
        print("--- by innermost source line")
        for r in res["top_lines"]:
            print(f"{r['line']:32s} valu/pass {r['valu_per_pass']:7.1f}  other {r['other_per_pass']:6.1f}")
    return res


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        which = sys.argv[2] if len(sys.argv) > 2 else "hyb"
        build(which, sys.argv[3] if len(sys.argv) > 3 else which)
    elif cmd == "run":
        gpu_run(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 8192, int(sys.argv[5]) if len(sys.argv) > 5 else None)
    elif cmd == "collect":      # collect <out.json> <counts.json> ...: the reports of several runs as one list (profiles/<round>/dynamic_mix.json)
        rows = []
        for fn in sys.argv[3:]:
            r = report(fn, 16, quiet=True)
            r["top_lines"] = r["top_lines"][:16]
            r["how"] = ("scripts/bbcount.py: executions of every basic block counted on the GPU (four instructions at the head of each block of the "
                        "compiler's own assembly, registers the kernel does not use; results checked against the plain build) x the block's static classes; "
                        "per evaluation pass = / sum of nmpc_status.reserved")
            rows.append(r)
        json.dump(rows, open(sys.argv[2], "w"), indent=1)
        print(sys.argv[2], len(rows), "entries")
    elif cmd == "report":
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
