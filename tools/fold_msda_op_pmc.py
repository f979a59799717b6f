#!/usr/bin/env python3
"""Fold the text written by tools/pmc_msda_op.sh into profiles/rNN_msda_op_pmc.json (what bench.py's roofline_msda_op /
roofline_msda_op_dec read their `traffic` from): HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB x 1024; the
factor 2: gfx950 tallies a 128-byte read request as 64 bytes, MI355X_MICROARCH.md, HBM).
    python tools/fold_msda_op_pmc.py profiles/r06_msda_op_pmc.txt > profiles/r06_msda_op_pmc.json"""
import json
import re
import sys

rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"\s+(\S+)\s+(.*)", ln)
    if not m or "(" not in m.group(1):
        continue
    kv = m.group(2).split()
    d = rows.setdefault(m.group(1), {})
    for k, v in zip(kv[0::2], kv[1::2]):
        try:
            d[k] = float(v)
        except ValueError:
            pass
out = {"source": sys.argv[1], "note": "per launch; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}
for key, name in (("windowed(enc)", "encoder"), ("general(dec)", "decoder"), ("general(skipped)", "encoder_skipped_general")):
    r = rows.get(key)
    if not r or "FETCH_SIZE" not in r:
        continue
    rec = {"read_raw": r["FETCH_SIZE"] * 1024, "read_corrected": 2 * r["FETCH_SIZE"] * 1024, "write": r.get("WRITE_SIZE", 0.0) * 1024}
    rec["hbm_bytes_per_launch"] = rec["read_corrected"] + rec["write"]
    if "SQ_LDS_IDX_ACTIVE" in r and r["SQ_LDS_IDX_ACTIVE"]:
        rec["lds_bank_conflict_over_idx_active"] = round(r.get("SQ_LDS_BANK_CONFLICT", 0.0) / r["SQ_LDS_IDX_ACTIVE"], 4)
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES"):
        if k in r:
            rec[k] = r[k]
    out[name] = rec
if "encoder" in out and "encoder_skipped_general" in out:
    out["encoder"]["hbm_bytes_per_launch"] += out["encoder_skipped_general"]["hbm_bytes_per_launch"]
print(json.dumps(out, indent=1))
