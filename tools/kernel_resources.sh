#!/bin/bash
# Print VGPRs / SGPRs / scratch / LDS / occupancy of the kernels of one scene translation unit.
#   tools/kernel_resources.sh <scene id> [name filter regex]
cd "$(dirname "$0")/../raymarch_algo_compare_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -disable-cgp-select2branch \
  -DRM_SCENE_ID=$1 -c rm_scene_tu.hip -o /tmp/_kr_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else None
cur, rows = None, []
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"rm::", "", name)
    if flt and not flt.search(name): continue
    print("%-100s vgpr %s sgpr %s (spill %s) scratch %s lds %s occ %s" % (name[:100], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "${2:-}"
