#!/bin/bash
# Where do the written bytes of the single-launch Mandelbulb frame come from?  WRITE_SIZE and FETCH_SIZE (separate --pmc passes)
# of the 1920x1080 Mandelbulb / Standard frame under schedule variants (RM_TUNING -> tools/prof_target.py).
#   tools/traffic_variants.sh [out dir]      (on a GPU box, from the repo root)
set -u
OUT=${1:-gpurun_out/traffic_variants}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
i=0
while IFS='|' read -r name tuning; do
  i=$((i + 1))
  export RM_TUNING="$tuning"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w$i" -- python3 tools/prof_target.py 10 0 1920 1080 4 > "$OUT/t$i.json" 2> "$OUT/w$i.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f$i" -- python3 tools/prof_target.py 10 0 1920 1080 4 > /dev/null 2> "$OUT/f$i.err"
  python3 tools/prof_collect.py "$OUT/v$i.json" "$OUT/w$i" "$OUT/f$i" > /dev/null 2>&1
  python3 - "$name" "$tuning" "$OUT/v$i.json" "$OUT/t$i.json" <<'PY'
import json, sys
name, tuning, path, tpath = sys.argv[1:5]
d = json.load(open(path))
ms = json.load(open(tpath))["ms_each"]
rows = {k: v for k, v in d.items() if isinstance(v, dict) and "WRITE_SIZE" in v and not k.startswith("order_tiles")}
w = sum(v["WRITE_SIZE"] for v in rows.values()) / 1024.0
f = sum(v["FETCH_SIZE"] for v in rows.values()) / 1024.0
print(json.dumps({"variant": name, "tuning": tuning, "write_MB": round(w, 1), "fetch_MB_x2": round(2 * f, 1), "frame_ms": round(sorted(ms)[len(ms) // 2], 3),
                  "kernels": {k.split("<")[0]: round(v["WRITE_SIZE"] / 1024.0, 1) for k, v in rows.items()}}), flush=True)
PY
  rm -rf "$OUT/w$i" "$OUT/f$i"
done <<'EOF2'
default|{}
one pass, nothing struck or parked|{"pipeline": 1, "suspend_after": [-1, 0]}
single launch, no filler|{"keep_busy": -1}
struck at 40, handed over at 48|{"suspend_after": [40, 48]}
struck at 24, handed over at 400|{"suspend_after": [24, 400]}
struck at 8, handed over at 48|{"suspend_after": [8, 48]}
three launches (parked at 32 / 128)|{"pipeline": 1}
one team workgroup|{"team_grid": 1}
EOF2
