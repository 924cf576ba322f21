#!/bin/bash
# tools/kf_resources.sh [extra hipcc flags] -- registers, spills and occupancy of every k_frame instantiation (compiles k_frame.hip)
cd "$(dirname "$0")/../mlvfs_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -I../../include \
  -mllvm --amdgpu-sched-strategy=max-ilp -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c k_frame.hip -o /tmp/kf_res.o 2>&1 |
python3 -c '
import sys, re
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r" Name: (\S+)", ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", ln)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    t = re.search(r"k_frameILi(\d)ELb(\d)ELi(\d)ELb(\d)", k)
    if not t: continue
    print("k_frame<%s,%s,%s,%s>" % t.groups(), "VGPRs", v.get("VGPRs"), "SGPR spill", v.get("SGPRs Spill"), "VGPR spill", v.get("VGPRs Spill"), "occ", v.get("Occupancy [waves/SIMD]"))
'
