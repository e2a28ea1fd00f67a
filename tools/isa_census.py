#!/usr/bin/env python3
"""Per-block census of a kernel's gfx950 ISA, priced with measured issue costs.

Builds rt_device.hip for the device only with line tables (the code is the same as without),
takes one kernel's assembly, attributes every instruction to a source block and prints, per block:
static instruction counts by class, their issue cycles (prices measured by
tools/microbench/valu_rate on MI355X: cycles one SIMD needs per wave64 instruction once several
waves share it), and -- with a wave profile (tools/wave_profile.py output) -- the estimated
cycles per wave = static cycles x times a wave executes the block.

Source blocks are delimited by comments in rt_device.hip / rt_intersect.h:
    // [census: NAME @ COUNTER]      start of block NAME; it runs COUNTER times per wave
                                      (a key of the wave profile, a number, or key*number)
A block extends to the next marker of the same file.  An instruction belongs to the block of the
innermost frame of its inlined-at chain that lies in one of those two files; instructions inlined
from lines marked `// [census-cold]` (the general-operand fall-backs of the arithmetic cores) are
listed under <block>.cold and priced at zero executions.

  python tools/isa_census.py [--kernel REGEX] [--profile FILE] [--flags "-DRT_FAST ..."] [--keep DIR]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray-tracing_amd", "csrc")
BLOCK_FILES = ("rt_device.hip", "rt_intersect.h")

# issue cycles per wave64 instruction on one SIMD (tools/microbench/valu_rate, >= 4 waves per SIMD)
FULL = {"v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_fmamk_f32",
        "v_fmaak_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32",
        "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def price(op):
    """(class, cycles).  Classes: valu2 (full rate), valu4 (half rate), trans, salu, branch, lds, vmem, wait, nop."""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in ("s_nop",):
        return "nop", 4          # one issue slot of the wave; s_nop N holds the wave N+1 cycles
    if base.startswith("s_waitcnt"):
        return "wait", 0
    if base.startswith("s_cbranch") or base in ("s_branch", "s_endpgm", "s_barrier", "s_setpc_b64"):
        return "branch", 4
    if base.startswith("s_load") or base.startswith("s_buffer_load"):
        return "smem", 4
    if base.startswith("s_"):
        return "salu", 4
    if base.startswith("ds_"):
        return "lds", 4
    if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 4
    if base in TRANS:
        return "trans", 8
    if base in FULL:
        return "valu2", 2
    if base.startswith("v_"):
        return "valu4", 4        # compares, selects, min/max, shifts, conversions, div helpers, packed math
    return "other", 4


def build_asm(flags, out_s):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize",
           "-gline-tables-only", "--cuda-device-only", "-S", "-o", out_s, os.path.join(CSRC, "rt_device.hip")] + flags
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.exit(r.stdout[-3000:])


def load_markers():
    """file -> sorted list of (line, name, counter expr); file -> set of cold lines."""
    marks, cold = {}, {}
    for fn in BLOCK_FILES + ("rt_numerics.h",):
        path = os.path.join(CSRC, fn)
        ms, cs = [], set()
        for i, line in enumerate(open(path), 1):
            m = re.search(r"\[census:\s*([\w.]+)\s*@\s*([^\]]+)\]", line)
            if m:
                ms.append((i, m.group(1), m.group(2).strip()))
            if "[census-cold]" in line:
                cs.add(i)
        marks[fn], cold[fn] = ms, cs
    return marks, cold


def block_of(marks, fn, line):
    name, expr = "prologue", "1"
    for (l, n, e) in marks.get(fn, []):
        if l <= line:
            name, expr = n, e
        else:
            break
    return name, expr


def eval_counter(expr, prof):
    """'cand', '2', 'iter*0.5', 'sphere+torus'."""
    total = 0.0
    for term in expr.split("+"):
        v = 1.0
        for f in term.split("*"):
            f = f.strip()
            try:
                v *= float(f)
            except ValueError:
                if f not in prof:
                    return None
                v *= prof[f]
        total += v
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=r"rt_strict9rt_kernelILi3ELb0ELi1ELb0ELb0ELb0E", help="regex on the mangled kernel name")
    ap.add_argument("--profile", help="wave profile (tools/wave_profile.py output): per-wave block counts")
    ap.add_argument("--flags", default="-ffp-contract=off")
    ap.add_argument("--asm", help="use this .s instead of compiling")
    ap.add_argument("--keep", help="directory to keep the annotated assembly of the kernel in")
    ap.add_argument("--by-op", action="store_true", help="also list opcode counts per block")
    args = ap.parse_args()

    tmp = tempfile.mkdtemp(prefix="census_")
    s_path = args.asm or os.path.join(tmp, "rt_device.s")
    if not args.asm:
        build_asm(args.flags.split(), s_path)
    marks, cold = load_markers()
    prof = {}
    if args.profile:
        for line in open(args.profile):
            m = re.match(r"(\w+)\s+\d+\s+per wave\s+([\d.]+)", line)
            if m:
                prof[m.group(1)] = float(m.group(2))

    kre = re.compile(r"^(_Z\w*" + args.kernel + r"\w*):")
    in_k, cur_chain, name = False, [], None
    last_blk, last_expr = "prologue", "1"
    stats = collections.OrderedDict()        # block -> {"expr":..., "cls": Counter, "cyc": Counter, "ops": Counter}
    kept = []
    for raw in open(s_path):
        if not in_k:
            m = kre.match(raw)
            if m:
                in_k, name = True, m.group(1)
            continue
        kept.append(raw)
        t = raw.strip()
        if t.startswith(".Lfunc_end"):
            break
        if t.startswith(".loc"):
            # "; file:line:col @[ file:line:col @[ ... ] ]"
            cur_chain = re.findall(r"(/[^\s:]+):(\d+)(?::\d+)?", t.split(";", 1)[1] if ";" in t else "")
            continue
        m = re.match(r"([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)\b", t)
        if not m:
            continue
        op = m.group(1)
        blk, expr, is_cold = last_blk, last_expr, False
        for (path, line) in cur_chain:              # innermost first
            fn = os.path.basename(path)
            if int(line) in cold.get(fn, ()):
                is_cold = True
            if fn in BLOCK_FILES:
                # line 0 = compiler-generated (exec mask bookkeeping, branches): stays with the
                # block of the instructions before it
                if int(line) > 0:
                    blk, expr = block_of(marks, fn, int(line))
                    last_blk, last_expr = blk, expr
                break
        if is_cold:
            blk, expr = blk + ".cold", "0"
        st = stats.setdefault(blk, {"expr": expr, "cls": collections.Counter(), "cyc": collections.Counter(), "ops": collections.Counter()})
        cls, cyc = price(op)
        if op == "s_nop":
            n = re.search(r"s_nop\s+(\d+)", t)
            cyc = 4 + (int(n.group(1)) if n else 0)
        st["cls"][cls] += 1
        st["cyc"][cls] += cyc
        st["ops"][re.sub(r"_(e32|e64)$", "", op)] += 1
    if not in_k:
        sys.exit("kernel not found: " + args.kernel)
    if args.keep:
        os.makedirs(args.keep, exist_ok=True)
        open(os.path.join(args.keep, "kernel.s"), "w").writelines(kept)

    classes = ["valu2", "valu4", "trans", "salu", "branch", "lds", "vmem", "smem", "nop", "wait", "other"]
    print(f"kernel {name}")
    print("static instructions per block (cycles = issue cycles on one SIMD at the measured prices: valu2 2, valu4 4, trans 8, salu/branch/lds 4)")
    hdr = f"{'block':26s} {'runs/wave':>9s} " + " ".join(f"{c:>6s}" for c in classes) + f" {'valu cyc':>9s} {'salu cyc':>9s} | {'dyn valu instr':>14s} {'dyn valu cyc':>12s} {'dyn salu cyc':>12s}"
    print(hdr)
    tot = collections.Counter()
    for blk, st in stats.items():
        valu_cyc = st["cyc"]["valu2"] + st["cyc"]["valu4"] + st["cyc"]["trans"]
        salu_cyc = st["cyc"]["salu"] + st["cyc"]["branch"] + st["cyc"]["smem"]
        valu_n = st["cls"]["valu2"] + st["cls"]["valu4"] + st["cls"]["trans"]
        runs = eval_counter(st["expr"], prof) if prof or re.fullmatch(r"[\d.]+", st["expr"]) else None
        dyn = ""
        if runs is not None:
            dyn = f"{valu_n * runs:14.1f} {valu_cyc * runs:12.1f} {salu_cyc * runs:12.1f}"
            tot["dyn_valu_n"] += valu_n * runs
            tot["dyn_valu_cyc"] += valu_cyc * runs
            tot["dyn_salu_cyc"] += salu_cyc * runs
            tot["dyn_salu_n"] += (st["cls"]["salu"] + st["cls"]["branch"] + st["cls"]["smem"]) * runs
            tot["dyn_lds_n"] += st["cls"]["lds"] * runs
            tot["dyn_trans"] += st["cls"]["trans"] * runs
            tot["dyn_valu4"] += st["cls"]["valu4"] * runs
            tot["dyn_valu2"] += st["cls"]["valu2"] * runs
        runs_s = f"{runs:9.3f}" if runs is not None else f"{st['expr'][:9]:>9s}"
        print(f"{blk:26s} {runs_s} " + " ".join(f"{st['cls'][c]:6d}" for c in classes) + f" {valu_cyc:9d} {salu_cyc:9d} | {dyn}")
        for c in classes:
            tot[c] += st["cls"][c]
        if args.by_op:
            print("      " + ", ".join(f"{o} {n}" for o, n in st["ops"].most_common()))
    print(f"{'total static':26s} {'':9s} " + " ".join(f"{tot[c]:6d}" for c in classes))
    if tot["dyn_valu_n"]:
        print(f"estimated per wave: VALU instructions {tot['dyn_valu_n']:.0f} (full rate {tot['dyn_valu2']:.0f}, half rate {tot['dyn_valu4']:.0f}, "
              f"transcendental {tot['dyn_trans']:.0f}), VALU issue cycles {tot['dyn_valu_cyc']:.0f}; SALU+branch instructions {tot['dyn_salu_n']:.0f}, "
              f"cycles {tot['dyn_salu_cyc']:.0f}; LDS instructions {tot['dyn_lds_n']:.0f}")


if __name__ == "__main__":
    main()
