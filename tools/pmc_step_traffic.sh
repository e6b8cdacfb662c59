#!/bin/bash
# HBM-side traffic per launch of every kernel of the C2 step from rocprofv3 PMC counters, one counter per pass (MI355X_MICROARCH.md, HBM section:
# FETCH_SIZE / WRITE_SIZE in KiB-units of 1024 B; gfx950: FETCH_SIZE counts half of a wide streaming read -> doubled below).
# Writes gpurun_out/pmc_step/summary.json (copy to profiles/round4_gemm_pmc_traffic.json: bench.py reads roofline.traffic from it, labelled static).
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_step
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rates --no-extras --no-kernel-legs > $OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $OUT/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_step/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "qv::" not in k:
                continue
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    m = lambda c: sum(v[c]) / max(1, len(v[c]))
    out[k] = {"launches": len(v["FETCH_SIZE"]), "fetch_MB": round(2 * m("FETCH_SIZE") * 1024 / 1e6, 1), "write_MB": round(m("WRITE_SIZE") * 1024 / 1e6, 1)}
# the dominant GEMM classes by their epilogue mode (9th template argument of k_gemm_nt<TA, NS, WM, TM, TB, WN, TNT, BK, PM, I8, F16>), not by a full template string
def pm_of(name):
    if not name.startswith("qv::k_gemm_nt<"):
        return None
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    return int(args[8]) if len(args) >= 9 and args[8].lstrip("-").isdigit() else None
res = {"kernels": out}
found = 0
for kind, modes in (("4", (18, 8)), ("5", (19, 9, 5))):   # (18 / 19: the one-plane forms of modes 8 / 9)
    names = [k for k in out if pm_of(k) in modes]
    if kind == "5" and any(k.startswith("qv::k_f16_strip_gelu_bwd") for k in out):   # ViT-S: the fc2 dgrad + GELU' runs in the A-stationary strip kernel (f16strip.hip)
        names = [k for k in out if k.startswith("qv::k_f16_strip_gelu_bwd")]
    if names:
        name = max(names, key=lambda k: out[k]["launches"])
        o = out[name]
        found += 1
        res[kind] = {"bytes_per_launch": (o["fetch_MB"] + o["write_MB"]) * 1e6, "fetch_MB": o["fetch_MB"], "write_MB": o["write_MB"],
                     "note": f"mean FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch of {name} over {o['launches']} launches of the C2 step at batch 256, "
                             "separate rocprofv3 --pmc passes (tools/pmc_step_traffic.sh)"}
if not found:
    print("no k_gemm_nt launch with epilogue mode 8 / 9 found"); raise SystemExit(1)
for kind, prefix in (("3", "qv::k_gemm_tn_q8<0"), ("6", "qv::k_gemm_tn_q8<1")):   # the byte-X weight gradients (their k_tn_reduce launches are separate rows of "kernels")
    names = [k for k in out if k.startswith(prefix)]
    if names:
        o = out[names[0]]
        res[kind] = {"bytes_per_launch": (o["fetch_MB"] + o["write_MB"]) * 1e6, "fetch_MB": o["fetch_MB"], "write_MB": o["write_MB"],
                     "note": f"mean FETCH_SIZE (x2) + WRITE_SIZE per launch of {names[0]} over {o['launches']} launches (the partial tiles included, k_tn_reduce not)"}
json.dump(res, open("gpurun_out/pmc_step/summary.json", "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_MB"] + kv[1]["write_MB"]) * kv[1]["launches"])[:24]:
    print(f'{o["launches"]:5d} x  fetch {o["fetch_MB"]:8.1f} MB  write {o["write_MB"]:8.1f} MB   {k[:100]}')
PY
