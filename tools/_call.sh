cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02c; mkdir -p $out
timeout 600 python -m pytest tests/test_msda_encoder_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for noise in 0 0.5 2; do
 for cfg in "1 0" "4 0" "1 1" "4 1" "8 1" "2 1"; do
  set -- $cfg
  echo -n "BAND=$1 STATIC=$2: "
  CODETR_MSDA_BAND=$1 CODETR_MSDA_STATIC=$2 timeout 120 python tools/bench_msda_encoder.py --noise $noise --batch 4 2>&1 | tail -1
 done
done | tee $out/msda_ab.txt
for cfg in "1 0" "4 1"; do
 set -- $cfg
 for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_enc
  CODETR_MSDA_BAND=$1 CODETR_MSDA_STATIC=$2 timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_enc -- python tools/bench_msda_encoder.py --iters 2 --noise 2 --batch 4 > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc_enc -name "*counter_collection.csv" | head -1)
  echo "BAND=$1 STATIC=$2 $c (KiB per launch, mean):"
  python - "$f" <<'PY'
import csv, collections, sys
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "msda" in k:
        k = k[28:60]; acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in acc: print("   ", k, "%.4e" % (acc[k] / n[k]), "launches", n[k])
PY
 done
done | tee $out/msda_pmc.txt
