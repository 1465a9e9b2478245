#!/bin/bash
# usage: pmc.sh <tag> <counters...> -- <cmd...>   (one rocprofv3 --pmc pass, prints per-kernel mean of each counter)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $out -- "$@" > $out/log.txt 2>&1)
python - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:70], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for (kn, cn), (n, v) in sorted(agg.items()):
        print(f"{kn:70s} {cn:28s} n={n:5d} mean={v/n:.4g} sum={v:.5g}")
PY
