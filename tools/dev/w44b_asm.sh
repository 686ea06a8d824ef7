#!/bin/bash
# device assembly of dncnn_wino44b.hip -> /tmp/w44b.s, with the register / spill summary
cd /root/repo/pnp_svrg_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -pragma-unroll-threshold=200000 -fno-slp-vectorize $W44B_EXTRA -x hip dncnn_wino44b.hip --cuda-device-only -S -o /tmp/w44b.s 2>&1 | grep -v "unused"
grep -E "^\s+\.(vgpr_spill_count|private_segment_fixed_size)|\.name:" /tmp/w44b.s | head -12
python3 /root/repo/tools/dev/w44b_stat.py | tail -4
