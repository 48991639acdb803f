#!/usr/bin/env python3
"""Copy a round's collected profile summaries (gpurun_out/prof_rNN, written on the GPU box by tools/prof_round.sh and
the table tools) into profiles/rNN/ and write profiles/pmc_rNN.json: the HBM bytes per launch of the headline kernel
(WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 read-side correction of MI355X_MICROARCH.md) stamped with the fingerprint of
the kernel sources they were measured on -- bench.py reports `roofline.traffic` only while that fingerprint matches.
    python tools/prof_finish.py r03"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_fingerprint          # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for dp, _, files in os.walk(src):
    for f in files:
        if f.endswith((".err",)) and os.path.getsize(os.path.join(dp, f)) == 0:
            continue
        rel = os.path.relpath(os.path.join(dp, f), src)
        os.makedirs(os.path.dirname(os.path.join(dst, rel)), exist_ok=True)
        shutil.copy2(os.path.join(dp, f), os.path.join(dst, rel))
pmc = json.load(open(os.path.join(dst, "mandelbulb_standard", "pmc_per_launch.json")))["pipeline_kernel"]
hbm = (pmc["WRITE_SIZE"] + 2.0 * pmc["FETCH_SIZE"]) * 1024.0
out = {"hbm_bytes_per_launch": int(round(hbm)), "csrc_sha16": csrc_fingerprint(),
       "kernel": "pipeline_kernel<SceneMandelbulb, StratStandard, 4, true, false>", "workload": "Mandelbulb/Standard 1920x1080",
       "source": f"profiles/{tag}/mandelbulb_standard/pmc_per_launch.json: WRITE_SIZE {pmc['WRITE_SIZE']:.0f} KB + 2 x FETCH_SIZE "
                 f"{pmc['FETCH_SIZE']:.0f} KB (read-side correction), counters collected in separate --pmc passes, mean over "
                 f"{pmc['launches_per_counter_pass']} launches; 18.66 MB of it are the 9 B/ray outputs"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"pmc_{tag}.json"), "w"), indent=1)
print(json.dumps(out))
