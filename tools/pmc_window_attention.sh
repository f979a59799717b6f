#!/bin/bash
# SQ counters of the window-attention kernel at the 4-image launch shapes (tools/bench_window_attention.py 4), per launch
# means over all launches of the run:   bash tools/pmc_window_attention.sh <out.txt>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1
echo "== window_attention_kernel, python tools/bench_window_attention.py 4 (stages 0-3, shift 0 / 6), means over the run's launches ==" >> $out
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VALU_TRANS_F32"; do
  rm -rf /tmp/pmc_wa
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_wa -- python3 tools/bench_window_attention.py 4 > /tmp/pmc.log 2>&1
  f=$(ls -S /tmp/pmc_wa/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "  (pass failed: $pmc)" >> $out; tail -3 /tmp/pmc.log >> $out; continue; fi
  python3 - "$f" >> $out <<'PY'
import csv, collections, sys
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "window_attention" not in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("  " + "  ".join("%s %.5g" % (c, v / n[c]) for c, v in sorted(acc.items())), " (launches %d)" % (max(n.values()) if n else 0))
PY
done
