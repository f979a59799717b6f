#!/bin/bash
# SQ counters per kernel of one 4-image forward (bench.py --batch 4 --no-graph), two passes, folded per kernel name:
#   bash tools/pmc_sq_model.sh <out.txt>
# Time-like SQ counters are in units of four cycles; per-WAVE means (counter / SQ_WAVES) are printed next to the shares of a
# wave's life spent waiting to issue an instruction (SQ_WAIT_INST_ANY: dependencies, s_waitcnt, barriers, full queues), with vector
# instructions active, and with LDS instructions active.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1
p4="--batch 4 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-roofline --no-host-feed --no-fp8-line --pad 1.0"
rm -rf /tmp/sq1 /tmp/sq2
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/sq1 -- python3 bench.py $p4 > /tmp/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_MFMA --output-format csv -d /tmp/sq2 -- python3 bench.py $p4 > /tmp/sq2.log 2>&1
python3 - "$(ls -S /tmp/sq1/*/*counter_collection.csv | head -1)" "$(ls -S /tmp/sq2/*/*counter_collection.csv | head -1)" > $out <<'PY'
import csv, collections, re, sys
def fold(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    return acc, n
a1, n1 = fold(sys.argv[1]); a2, n2 = fold(sys.argv[2])
print("# rocprofv3 --pmc (two passes) -- python bench.py --batch 4 --streams 1 --steps 2 --warmup 1 --no-graph ... (tools/pmc_sq_model.sh)")
print("# per kernel over the whole run (3 forwards of 4 images); time-like counters in units of 4 cycles; per-wave means")
print("# %-58s %8s %9s %7s %7s %7s %8s %8s %7s %7s %7s %7s" % ("kernel", "launches", "waves", "cyc/wv", "wait%", "valu%", "valu/wv", "vmem/wv", "lds/wv", "ldsw%", "confl%", "mfma/wv"))
rows = []
for k, c in a1.items():
    w = c.get("SQ_WAVES", 0.0)
    if w <= 0 or c.get("SQ_WAVE_CYCLES", 0) <= 0: continue
    d = a2.get(k, {}); w2 = d.get("SQ_WAVES", 0.0) or 1.0
    wc = c["SQ_WAVE_CYCLES"]
    rows.append((wc, k, n1[k], w, wc / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_INSTS_VALU", 0) / w,
                 d.get("SQ_INSTS_VMEM", 0) / w2, d.get("SQ_INSTS_LDS", 0) / w2, 100 * d.get("SQ_WAIT_INST_LDS", 0) / (wc * w2 / w),
                 100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 0), 1.0), d.get("SQ_INSTS_MFMA", 0) / w2))
for r in sorted(rows, reverse=True)[:28]:
    print("  %-58s %8d %9.0f %7.0f %7.1f %7.1f %8.0f %8.1f %7.1f %7.1f %7.1f %7.1f" % (r[1][:58], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
PY
cat $out
