#!/usr/bin/env python3
"""Instruction mix of the solve kernel by loop nest, from the compiler's assembly (CPU only).  For every loop of the chosen build
(LLVM's loop comments give header and depth) the static instructions are classed: FP64 arithmetic, matrix core, moves (plain /
DPP / lane reads and writes), selects, compares, integer + address arithmetic, LDS, scalar.  VERDICT r5 item 7 asks for the
non-FP64 vector share of the throughput build; `profiles/r06_bulk_pmc_summary.csv` has the dynamic totals this is read next to.
    python tools/isa_mix.py [throughput|latency] > profiles/rNN_isa_mix_<build>.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mpc-rl_for_avs_amd", "csrc", "mpc_engine.hip")
BUILDS = {"throughput": "ILb1ELi20ELi3ELi0E", "latency": "ILb1ELi20ELi2ELi7E"}
CLASSES = ["fp64", "mfma", "mov", "dpp", "lane", "select", "cmp", "int", "cvt/other", "lds", "salu", "branch", "wait/nop", "vmem/smem"]


def classify(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch"
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
            return "wait/nop"
        if op.startswith(("s_load", "s_buffer", "s_store")):
            return "vmem/smem"
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem/smem"
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return "lane"
    if "dpp" in line or op.startswith("v_permlane"):
        return "dpp"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith("v_cmp"):
        return "cmp"
    if re.search(r"_f64", op) and not op.startswith("v_cvt"):
        return "fp64"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "mov"
    if op.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshl", "v_lshr", "v_ashr", "v_and", "v_or", "v_xor", "v_not", "v_bfe",
                      "v_bfi", "v_mad_u", "v_mad_i", "v_mul_lo", "v_mul_hi", "v_mul_u", "v_mul_i", "v_add_co", "v_addc", "v_add3", "v_add_lshl",
                      "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_min_u", "v_max_u", "v_min_i", "v_max_i", "v_mbcnt", "v_sub_co", "v_subb",
                      "v_add_i", "v_sub_i", "v_bcnt", "v_ffb", "v_alignbit", "v_perm")):
        return "int"
    return "cvt/other"


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "throughput"
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "engine.s")
        res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", s, SRC] +
                             sys.argv[2:], capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(res.stderr)
        asm = open(s).read().split("\n")
    k, on = [], False
    for l in asm:
        if re.match(r"^_ZN12_GLOBAL__N_121mpc_solve_wave_kernel" + BUILDS[which], l):
            on = True
        if on:
            k.append(l)
        if on and l.startswith(".Lfunc_end"):
            break
    # every basic block (a label or a fall-through `; %bb.N:`) carries LLVM's loop comment: the header it is, or the innermost
    # loop it is in
    loops, order, parent, other = {}, [], {}, {}
    key = (0, "-")
    n = 0
    while n < len(k):
        l = k[n]
        m = re.match(r"^(?:\.L(BB\d+_\d+)|; %bb\.\d+):", l)
        if m:
            q, blk = n, ""
            while q < len(k) and (q == n or k[q].lstrip().startswith(";")) and not (q > n and re.match(r"^; %bb\.\d+:", k[q])):
                blk += k[q] + "\n"
                q += 1
            own = re.search(r"Loop Header: Depth=(\d)", blk)
            inn = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d)", blk)
            par = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d)", blk)
            if own and m.group(1):
                key = (int(own.group(1)), m.group(1))
                if par:
                    parent[key] = par[-1][0]
            elif inn:
                key = (int(inn.group(2)), inn.group(1))
            else:
                key = (0, "-")
            n = q if q > n else n + 1
            continue
        mm = re.match(r"^\t([a-z_0-9]+)\s*(.*)", l)
        n += 1
        if not mm:
            continue
        if key not in loops:
            loops[key] = {c: 0 for c in CLASSES}
            order.append(key)
        c = classify(mm.group(1), l)
        loops[key][c] += 1
        if c == "cvt/other":
            other[mm.group(1)] = other.get(mm.group(1), 0) + 1
    print(f"mpc_solve_wave_kernel<{which}>: static instructions by loop (in program order; depth 1 = the iteration loop's own blocks)")
    print("depth header       parent     total | " + " ".join(f"{c:>9s}" for c in CLASSES) + " | non-FP64 vector")
    tot = {c: 0 for c in CLASSES}
    for key in order:
        t = loops[key]
        n = sum(t.values())
        nf = t["mov"] + t["dpp"] + t["lane"] + t["select"] + t["cmp"] + t["int"] + t["cvt/other"]
        print(f"{key[0]:5d} {key[1]:12s} {str(parent.get(key, '')):10s} {n:5d} | " + " ".join(f"{t[c]:9d}" for c in CLASSES) + f" | {nf:5d}")
        for c in CLASSES:
            tot[c] += t[c]
    n = sum(tot.values())
    nf = tot["mov"] + tot["dpp"] + tot["lane"] + tot["select"] + tot["cmp"] + tot["int"] + tot["cvt/other"]
    print(f"  all {'':12s} {'':10s} {n:5d} | " + " ".join(f"{tot[c]:9d}" for c in CLASSES) + f" | {nf:5d}")
    print("cvt/other opcodes:", ", ".join(f"{o} {c}" for o, c in sorted(other.items(), key=lambda q: -q[1])))


if __name__ == "__main__":
    main()
