"""rocprofv3 --pmc CSVs (one pass per counter set, see tools/profile_round.sh) -> per-kernel JSON with the
gfx950 corrections of MI355X_MICROARCH.md applied:
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs            -> cycles per XCD = value / 8
  * SQ_VALU_MFMA_BUSY_CYCLES is summed over 1024 SIMDs   -> busy cycles per SIMD = value / 1024
  * FETCH_SIZE (KB) reports HALF the bytes of wide reads  -> read bytes = 2 * 1024 * value
  * WRITE_SIZE (KB)                                       -> write bytes = 1024 * value
Usage: python tools/pmc_to_json.py out.json pass_a.csv pass_b.csv pass_c.csv"""
import collections
import csv
import json
import sys

out_path, files = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if "cocos" not in name:
                continue
            name = name.split("(")[0].replace("void ", "").strip()
            if name.startswith("_ZN5cocos"):     # rocprofv3 leaves names with _Float16 parameters mangled
                import re
                m = re.match(r"_ZN5cocos(\d+)", name)
                n = int(m.group(1))
                base = name[len(m.group(0)):len(m.group(0)) + n]
                targs = re.findall(r"L[ib](\d+)E", name.split("E", 1)[0] if "IL" not in name else name[name.index("IL"):name.index("EEv") if "EEv" in name else len(name)])
                name = "cocos::" + base + ("<" + ", ".join(targs) + ">" if targs else "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for name, d in agg.items():
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    rec = {"dispatches": max(len(v) for v in d.values())}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
        rec["mfma_busy_cycles_per_simd"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        rec["gui_active_cycles_per_xcd"] = avg["GRBM_GUI_ACTIVE"] / 8
        rec["MfmaUtil"] = rec["mfma_busy_cycles_per_simd"] / rec["gui_active_cycles_per_xcd"]
    if "SQ_WAIT_ANY" in avg and "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"]:
        rec["wait_any_frac"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
    if "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"]:
        for c, k in (("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_frac"), ("SQ_ACTIVE_INST_VALU", "active_valu_frac"),
                     ("SQ_ACTIVE_INST_LDS", "active_lds_frac"), ("SQ_ACTIVE_INST_VMEM", "active_vmem_frac"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac")):
            if c in avg:
                rec[k] = avg[c] / avg["SQ_WAVE_CYCLES"]
    if "SQ_WAVES" in avg:
        rec["waves"] = avg["SQ_WAVES"]
    if "SQ_LDS_BANK_CONFLICT" in avg:
        rec["lds_bank_conflict_cycles"] = avg["SQ_LDS_BANK_CONFLICT"]
    for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F16"):
        if c in avg:
            rec[c] = avg[c]
    if "FETCH_SIZE" in avg:
        rec["hbm_read_bytes"] = 2.0 * 1024.0 * avg["FETCH_SIZE"]
    if "WRITE_SIZE" in avg:
        rec["hbm_write_bytes"] = 1024.0 * avg["WRITE_SIZE"]
    if "hbm_read_bytes" in rec and "hbm_write_bytes" in rec:
        rec["hbm_bytes"] = rec["hbm_read_bytes"] + rec["hbm_write_bytes"]
    res[name] = rec
json.dump(res, open(out_path, "w"), indent=1)
for k, v in res.items():
    print(k[:90], {a: (round(b, 4) if isinstance(b, float) and b < 10 else int(b)) for a, b in v.items()})
