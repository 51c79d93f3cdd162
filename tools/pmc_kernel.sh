#!/bin/bash
# SQ counters of ONE kernel under any command (GPU box): tools/pmc_kernel.sh <kernel-name substring> <tag> <program and arguments>
#   e.g. tools/pmc_kernel.sh k_flash_attn64 flash_mode2 python3 tools/bench_flash.py 2
# Three separate --pmc passes (counter slots) + one --stats pass, kernel trace only (gpurun refuses --pmc with the runtime traces; the program itself
# follows `--`, no shell in between).  Writes the per-counter means over the kernel's launches and the derived fractions to
# gpurun_out/pmc/<tag>/summary.json; copy it to profiles/rNN/.
set -eu
if [ "$#" -lt 3 ] || [ -z "${GRAFT_REPO_ROOT:-}" ] || [ -z "$2" ]; then echo "usage (GPU box, GRAFT_REPO_ROOT set): $0 <kernel substring> <tag> <program> [args...]" >&2; exit 2; fi
case "$2" in */*|.*) echo "tag must be a plain name" >&2; exit 2;; esac
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; K="$1"; TAG="$2"; shift 2
O="$R/gpurun_out/pmc/$TAG"; rm -rf -- "$O"; mkdir -p "$O"
set +e
ARGS=(); for a in "$@"; do if [ -f "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/sq1" -- "${ARGS[@]}" > "$O/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d "$O/sq2" -- "${ARGS[@]}" > "$O/sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_LEVEL_LDS --output-format csv -d "$O/sq3" -- "${ARGS[@]}" > "$O/sq3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- "${ARGS[@]}" > "$O/stats.log" 2>&1
python3 - <<PY
import csv, glob, json
from collections import defaultdict
out = {"kernel": "$K", "command": "${ARGS[*]} under rocprofv3 --kernel-trace --pmc (three passes) / --stats", "counters": {}}
for d in ("sq1", "sq2", "sq3"):
    fs = glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no data"); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "$K" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out["counters"][k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
fs = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "$K" in r["Name"]: out["kernel_avg_ns"] = float(r["AverageNs"]); out["kernel_calls"] = int(r["Calls"])
c = {k: v["mean_per_launch"] for k, v in out["counters"].items()}
der = {}
# units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the 1,024 SIMDs; GRBM_GUI_ACTIVE counts cycles summed over the 8
# XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves
if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
    kcyc = c["GRBM_GUI_ACTIVE"] / 8.0
    der["kernel_shader_cycles"] = kcyc
    if out.get("kernel_avg_ns"): der["effective_clock_ghz"] = kcyc / out["kernel_avg_ns"]
    der["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * kcyc)
    if c.get("SQ_INSTS_MFMA"): der["mfma_cycles_per_instruction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"]
if c.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if c.get(k): der[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_LDS_BANK_CONFLICT") is not None: der["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
if c.get("SQ_INSTS_MFMA"):
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU"):
        if c.get(k): der[k.lower() + "_per_mfma"] = c[k] / c["SQ_INSTS_MFMA"]
out["derived"] = der
json.dump(out, open("$O/summary.json", "w"), indent=1)
print("$TAG", json.dumps({k: round(v, 4) for k, v in der.items()}), "avg_us", round(out.get("kernel_avg_ns", 0) / 1e3, 1))
PY
