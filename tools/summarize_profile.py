"""summarize gpurun_out/profile_<tag> into profiles/<round>/<tag>_*: kernel stats + HBM traffic per kernel (PMC)."""
import csv, glob, json, shutil, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
tag, rnd = sys.argv[1], sys.argv[2]
src = ROOT / "gpurun_out" / f"profile_{tag}"
dst = ROOT / "profiles" / rnd
dst.mkdir(parents=True, exist_ok=True)
shutil.copy(sorted(glob.glob(str(src / "stats/**/*kernel_stats.csv"), recursive=True))[-1], dst / f"{tag}_kernel_stats.csv")
shutil.copy(str(src) + ".bench.json", dst / f"{tag}_bench.json")
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("ncsn::", "").replace("void ", "")
    return n.split("(")[0][:48]
traffic = defaultdict(lambda: dict(launches=0, fetch_kb=0.0, write_kb=0.0))
for kind, key in (("fetch", "fetch_kb"), ("write", "write_kb")):
    f = sorted(glob.glob(str(src / kind / "**/*_counter_collection.csv"), recursive=True))[-1]
    seen = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        traffic[k][key] += float(r["Counter_Value"])
        seen[k] += 1
    for k, n in seen.items():
        traffic[k]["launches"] = n
out = {}
for k, v in sorted(traffic.items(), key=lambda kv: -(kv[1]["fetch_kb"] + kv[1]["write_kb"])):
    n = max(v["launches"], 1)
    # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM);
    # WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Units: KB.
    out[k] = dict(launches=n, fetch_bytes_per_launch=2 * 1024 * v["fetch_kb"] / n, write_bytes_per_launch=1024 * v["write_kb"] / n,
                  hbm_bytes_per_launch=(2 * v["fetch_kb"] + v["write_kb"]) * 1024 / n)
(dst / f"{tag}_hbm_traffic.json").write_text(json.dumps(out, indent=1))
for k, v in list(out.items())[:12]:
    print(f"{k:50s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {v['write_bytes_per_launch']/1e6:9.2f} MB per launch")
