#!/bin/bash
# SQ counter passes over tools/bench_msda_encoder.py (run on the GPU box): per-launch means for both MSDA kernels.
#   bash tools/pmc_msda_encoder.sh <out.txt> <noise> <windows 0|1> - [passes 1|3] [counts 0|1] ["extra bench arguments", e.g. "--v4 1"]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; noise=$2; win=$3; passes=${5:-1}; counts=${6:-0}; extra=${7:-}
echo "== noise $noise px, windows $win, passes $passes, fp32 reference points $counts $extra (batch 1, 1920x1280 encoder shape) ==" >> $out
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_enc
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_enc -- python3 tools/bench_msda_encoder.py --iters 2 --noise $noise --windows $win --passes $passes --counts $counts $extra > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc_enc -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    k = "encoder_v4" if "msda_encoder_v4" in name else "encoder_v3" if "msda_encoder_v3" in name else "encoder_v2" if "msda_encoder_v2" in name else "encoder_v1" if "msda_encoder_kernel" in name else "general" if "msda_tiled" in name else None
    if k is None:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print("  %-10s" % k, "  ".join("%s %.4g" % (c, v / n[k][c]) for c, v in sorted(acc[k].items())))
PY
done
tail -1 /tmp/pmc.log >> $out
