#!/bin/bash
# HBM-side traffic per launch of every kernel of the C2 step from rocprofv3 PMC counters, one counter per pass (MI355X_MICROARCH.md, HBM section:
# FETCH_SIZE / WRITE_SIZE in KiB-units of 1024 B; gfx950: FETCH_SIZE counts half of a wide streaming read -> doubled below).
# Writes gpurun_out/pmc_step/summary.json (copy to profiles/round2_gemm_pmc_traffic.json: bench.py reads roofline.traffic from it, labelled static).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
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
kinds = {"4": "qv::k_gemm_nt<2, 3, 1, 13, 1, 0, 8, 3, 32, 0, 8, false, false>", "5": "qv::k_gemm_nt<2, 3, 1, 13, 1, 0, 8, 3, 32, 0, 5, false, false>"}
res = {"kernels": out}
for kind, name in kinds.items():
    if name in out:
        o = out[name]
        res[kind] = {"bytes_per_launch": (o["fetch_MB"] + o["write_MB"]) * 1e6, "fetch_MB": o["fetch_MB"], "write_MB": o["write_MB"],
                     "note": f"mean FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch of {name} over {o['launches']} launches of the C2 step at batch 256, "
                             "separate rocprofv3 --pmc passes (tools/pmc_step_traffic.sh): profiles/round2_gemm_pmc_traffic.json"}
json.dump(res, open("gpurun_out/pmc_step/summary.json", "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_MB"] + kv[1]["write_MB"]) * kv[1]["launches"])[:24]:
    print(f'{o["launches"]:5d} x  fetch {o["fetch_MB"]:8.1f} MB  write {o["write_MB"]:8.1f} MB   {k[:100]}')
PY
