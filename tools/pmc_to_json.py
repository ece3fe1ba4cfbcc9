"""profiles/<tag>_pmc_busy.txt (tools/gpu_pmc_mfma.sh) -> profiles/<tag>_pmc_kernels.json: per-kernel fractions."""
import json, re, sys
tag = sys.argv[1]
cur, data = None, {}
for ln in open("profiles/%s_pmc_busy.txt" % tag):
    m = re.match(r"^(k_\w+)\s*$", ln)
    if m:
        cur = data.setdefault(m.group(1), {}); continue
    m = re.match(r"^\s+(\w+)\s+([0-9.e+]+)\s+per launch", ln)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
out = {}
for k, v in data.items():
    o = {}
    if "GRBM_GUI_ACTIVE" in v: o["mfma_busy_frac_of_simd_cycles"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if v.get("SQ_LDS_IDX_ACTIVE"): o["lds_conflict_frac"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
    w = v.get("SQ_WAVE_CYCLES")
    if w:
        o["sq_wait_any_frac_of_wave_cycles"] = round(v["SQ_WAIT_ANY"] / w, 3)
        o["sq_wait_inst_any_frac_of_wave_cycles"] = round(v["SQ_WAIT_INST_ANY"] / w, 3)
        o["sq_active_inst_any_frac_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_ANY"] / w, 3)
    if v.get("SQ_INSTS_MFMA"): o["valu_per_mfma_instruction"] = round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 2)
    out[k] = o
out["note"] = ("rocprofv3 --pmc passes of bench.py (tools/gpu_pmc_mfma.sh), summed over the launches of a run; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
               "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), i.e. of the cycles at the clock the chip actually ran; lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
json.dump(out, open("profiles/%s_pmc_kernels.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
