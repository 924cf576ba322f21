#!/bin/bash
# tools/kfp_resources.sh [extra hipcc flags] -- registers, spills, occupancy and static instruction mix of k_frame_p<5,true,1,false>
# and <2,true,1,false> (compiles k_frame_p.hip with -DKFP_ONLY: seconds); leaves the annotated assembly in /tmp/kfp/
mkdir -p /tmp/kfp
cd "$(dirname "$0")/../mlvfs_amd/csrc"
KF=${KF_FLAGS--mllvm --amdgpu-sched-strategy=max-ilp -fno-slp-vectorize}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -I../../include -DKFP_ONLY $KF \
  -Rpass-analysis=kernel-resource-usage -gline-tables-only --save-temps=obj "$@" -c k_frame_p.hip -o /tmp/kfp/kfp.o 2>&1 |
python3 -c '
import sys, re
cur = None; rows = {}
for ln in sys.stdin:
    if "error" in ln: print(ln, end="")
    m = re.search(r" Name: (\S+)", ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"\s([A-Za-z][A-Za-z /\[\]]+): (\d+) \[-Rpass", ln)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    t = re.search(r"k_frame_pILi(\d)ELb(\d)ELi(\d)ELb(\d)", k)
    if not t: continue
    print("k_frame_p<%s,%s,%s,%s>" % t.groups(), "VGPRs", v.get("VGPRs"), "SGPRs", v.get("TotalSGPRs"), "SGPR spill", v.get("SGPRs Spill"), "VGPR spill", v.get("VGPRs Spill"), "scratch", v.get("ScratchSize [bytes/lane]"), "LDS", v.get("LDS Size [bytes/block]"), "occ", v.get("Occupancy [waves/SIMD]"))
'
python3 - <<'PY'
import re, collections
txt = open("/tmp/kfp/k_frame_p-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for M in (5, 2):
    k = "_ZN3mlv9k_frame_pILi%dELb1ELi1ELb0EEEvNS_9FrameArgsE" % M
    a = txt.index("\n" + k + ":"); b = txt.index("s_endpgm", a)
    body = txt[a:b].splitlines()
    open("/tmp/kfp/p%d.s" % M, "w").write("\n".join(body))
    c = collections.Counter()
    for ln in body:
        m = re.match(r"^\t([a-z][a-z0-9_]+)\b", ln)
        if m: c[m.group(1)] += 1
    print("M=%d static: all %d valu %d salu %d lds %d | readlane %d writelane %d scratch %d" % (M, sum(c.values()), sum(v for k, v in c.items() if k.startswith("v_")),
          sum(v for k, v in c.items() if k.startswith("s_")), sum(v for k, v in c.items() if k.startswith("ds_")), c["v_readlane_b32"], c["v_writelane_b32"],
          sum(v for k, v in c.items() if k.startswith("scratch_"))))
PY
