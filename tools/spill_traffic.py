#!/usr/bin/env python3
"""Where do the spilled scalar registers of the solve kernel cost instructions?  (CPU only: compiles csrc/mpc_engine.hip to
assembly and reads it.)  A scalar register the allocator cannot keep is parked in a lane of a vector register
(v_writelane_b32) and fetched back with v_readlane_b32 at its use; this lists those instructions by loop nest for the two
builds of the headline kernel, next to the v_readlane the algorithm itself asks for (wave reductions, the 2x2 block of a
Riccati stage).       python tools/spill_traffic.py > profiles/rNN_spill_traffic.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mpc-rl_for_avs_amd", "csrc", "mpc_engine.hip")


def kernel_lines(asm, mangled_part):
    out, on = [], False
    for l in asm:
        if re.match(r"^_ZN12_GLOBAL__N_121mpc_solve_wave_kernel" + mangled_part, l):
            on = True
        if on:
            out.append(l)
        if on and l.startswith(".Lfunc_end"):
            break
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "engine.s")
        res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", s, SRC],
                             capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(res.stderr)
        asm = open(s).read().split("\n")
    for part, label in (("ILb1ELi20ELi2ELi7E", "mpc_solve_wave_kernel<true, 20, 2, 7> (latency build, the headline's)"),
                        ("ILb1ELi20ELi3ELi0E", "mpc_solve_wave_kernel<true, 20, 3, 0> (throughput build)")):
        k = kernel_lines(asm, part)
        spillv = set(m.group(1) for m in (re.match(r"\tv_writelane_b32 (v\d+),", l) for l in k) if m)
        loops, cur = {}, None
        for l in k:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = {"hdr": "", "lab": m.group(1)}
                continue
            if cur is None:
                continue
            if l.strip().startswith(";") and "Loop" in l:
                cur["hdr"] += l
                continue
            mm = re.match(r"^\t([a-z_0-9]+)\s*(.*)", l)
            if not mm:
                continue
            own = re.search(r"Loop Header: Depth=(\d)", cur["hdr"])
            inn = re.findall(r"in Loop: Header=(BB\d+_\d+) Depth=(\d)", cur["hdr"])
            if own:
                key = (int(own.group(1)), cur["lab"][2:])
            elif inn:
                key = (int(inn[-1][1]), inn[-1][0])
            else:
                key = (0, "-")
            t = loops.setdefault(key, dict(n=0, sp_r=0, sp_w=0, alg_r=0, mfma=0, dpp=0))
            t["n"] += 1
            op, args = mm.group(1), mm.group(2)
            if op == "v_writelane_b32":
                t["sp_w"] += 1
            if op == "v_readlane_b32":
                if args.split(",")[1].strip() in spillv:
                    t["sp_r"] += 1
                else:
                    t["alg_r"] += 1
            if op.startswith("v_mfma"):
                t["mfma"] += 1
            if "dpp" in l:
                t["dpp"] += 1
        print(f"{label}: {len(spillv)} vector registers hold spilled scalars")
        print("   loop (depth, header)            static instructions | spill v_readlane | spill v_writelane | other v_readlane | mfma | dpp")
        for key in sorted(loops, key=lambda q: (q[0] > 0, q[1])):
            t = loops[key]
            if key[0] >= 2 and t["n"] < 40:
                continue
            what = {0: "outside the iteration loop", 1: "iteration body (phases)"}.get(key[0], "")
            if t["mfma"] >= 8:
                what = "Riccati stage loop"
            elif key[0] >= 2 and t["dpp"] >= 20:
                what = "rollout stage loop"
            print(f"   {key[0]} {key[1]:10s} {what:28s} {t['n']:6d} | {t['sp_r']:4d} | {t['sp_w']:4d} | {t['alg_r']:4d} | {t['mfma']:3d} | {t['dpp']:3d}")


if __name__ == "__main__":
    main()
