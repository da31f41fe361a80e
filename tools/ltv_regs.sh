#!/bin/bash
# Development aid: register / spill report of the two builds of mpc_ltv_kernel (mpc_engine.hip: launch_ltv).
cd "$(dirname "$0")/../mpc-rl_for_avs_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage -o /tmp/_ltv_regs.so mpc_engine.hip 2>&1 |
    grep -A11 "Function Name.*mpc_ltv_kernel" | grep -E "Function|VGPRs|Scratch|Occupancy|SGPRs" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
rm -f /tmp/_ltv_regs.so
