#!/bin/bash
# Counter passes over the PUBLIC op (tools/bench_msda_op.py --pmc: three launches per shape), run on the GPU box:
#   bash tools/pmc_msda_op.sh <out.txt> [spread px]
# Per-launch means for the windowed kernel (encoder shape), the general kernel's skipped launch behind it and the general
# kernel at the decoder shape.  FETCH_SIZE / WRITE_SIZE are in KiB per launch as rocprofv3 reports them; on gfx950 FETCH_SIZE
# counts wide coalesced reads at HALF their bytes (MI355X_MICROARCH.md, HBM): the summary line doubles it.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; spread=${2:-3}
echo "== public op, 1920x1280 pyramid (S = 204 600), fp16, batch 1; encoder shape: spread $spread px around the query's pixel; decoder shape: 900 queries ==" >> $out
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_op
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_op -- python3 tools/bench_msda_op.py --pmc --spread $spread > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc_op -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "msda_op4" in name:
        k = "windowed(enc)"
    elif "msda_tiled" in name:
        k = "general(dec)" if int(r.get("Grid_Size", "0") or 0) < 300000 * 1 and int(r.get("Grid_Size", "0") or 0) != 2048 * 256 else "general(skipped)"
    else:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print("  %-17s" % k, "  ".join("%s %.5g" % (c, v / n[k][c]) for c, v in sorted(acc[k].items())))
PY
done
grep -v amdgpu.ids /tmp/pmc.log | tail -3 >> $out
