#!/bin/bash
# SQ counter pass over tools/bench_msda_encoder.py (run on the GPU box): per-launch means for both MSDA kernels.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_enc
timeout 600 rocprofv3 --pmc $1 --output-format csv -d /tmp/pmc_enc -- python tools/bench_msda_encoder.py --iters 2 > /tmp/pmc.log 2>&1
tail -1 /tmp/pmc.log
f=$(find /tmp/pmc_enc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][28:52]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if "msda" in k:
        print(k, {c: "%.3e" % (v / n[k][c]) for c, v in acc[k].items()})
PY
