"""Compile the engine with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(registers, scratch, occupancy, LDS) - the table tracked as profiles/rNN_resource_usage.txt."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mpc-rl_for_avs_amd", "csrc", "mpc_engine.hip")


def main():
    res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage", SRC], capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr)
    rows, cur = [], None
    for line in res.stderr.splitlines():
        m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r":\s+(?:Function )?Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("void ", "")
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r":\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    cols = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]",
            "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
    print("kernel | " + " | ".join(cols))
    for r in rows:
        print(r["name"] + " | " + " | ".join(str(r.get(c, "-")) for c in cols))


if __name__ == "__main__":
    main()
