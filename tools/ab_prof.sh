#!/bin/bash
# A/B of two checkouts on one box: kernel durations (rocprofv3 kernel trace) and instruction counters of one cell.
#   tools/ab_prof.sh <out dir> <scene id> <checkout A> <checkout B>
set -u
OUT=$1; SID=$2; shift 2
export TMPDIR=/tmp
for T in "$@"; do
  name=$(basename $(cd $T && pwd))
  mkdir -p $OUT/$name
  (cd $T && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/$name/stats -- python3 tools/prof_target.py $SID 0 1920 1080 12 > $OLDPWD/$OUT/$name/target.json 2> $OLDPWD/$OUT/$name/stats.err)
  (cd $T && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OLDPWD/$OUT/$name/pmc -- python3 tools/prof_target.py $SID 0 1920 1080 6 > /dev/null 2> $OLDPWD/$OUT/$name/pmc.err)
  find $OUT/$name -name "*kernel_stats.csv" -exec cp {} $OUT/$name/kernel_stats.csv \;
  python3 - $OUT/$name <<'P'
import sys, glob, csv, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("<")[0].split("::")[-1]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "render" in k or "pipeline" in k:
        print(d, k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in sorted(v.items())})
for r in csv.DictReader(open(d + "/kernel_stats.csv")):
    if "render_kernel" in r["Name"] or "pipeline" in r["Name"] or "reduce" in r["Name"] or "fill" in r["Name"]:
        print(d, r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2))
P
  rm -rf $OUT/$name/stats $OUT/$name/pmc
done
