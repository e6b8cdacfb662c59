"""Overlap report of tools/dp_overlap_trace.sh: for the last traced step of rank 0, every large memory copy (a gradient bucket on its way to /
from the host-side gloo reduction) against the engine kernels that execute while it is in flight."""
import csv
import glob
import sys

out = sys.argv[1]
kf = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
mf = glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True)
if not kf or not mf:
    print("missing traces", kf, mf)
    sys.exit(1)
K = list(csv.DictReader(open(kf[0])))
Mc = list(csv.DictReader(open(mf[0])))
print("kernel trace columns:", list(K[0].keys()))
print("memcpy trace columns:", list(Mc[0].keys()))


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


ks = sorted(((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name")) for r in K), key=lambda t: t[0])
ms = []
for r in Mc:
    size = int(r.get("Size", r.get("Bytes", "0")) or 0) if any(k in r for k in ("Size", "Bytes")) else 0
    ms.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Direction"), size))
ms.sort()
# the last step: from the last k_img_patches launch (start of a forward) on
starts = [s for s, e, n in ks if "k_img_patches" in n]
t0 = starts[-1]
ks = [k for k in ks if k[0] >= t0]
ms = [m for m in ms if m[0] >= t0 and (m[1] - m[0]) > 200_000]       # copies longer than 0.2 ms: the gradient buckets
tend = max(max(e for s, e, n in ks), max((e for s, e, d, b in ms), default=0))
bwd0 = next(s for s, e, n in ks if "k_head_bwd" in n or "head_bwd" in n) if any("head_bwd" in n for s, e, n in ks) else ks[0][0]
print(f"last step of rank 0: {len(ks)} kernels, {(tend - t0) / 1e6:.2f} ms from the first forward kernel to the last event; backward starts at +{(bwd0 - t0) / 1e6:.2f} ms")
tot_ov = 0
for s, e, d, b in ms:
    conc = [(ks_, ke, n) for ks_, ke, n in ks if ke > s and ks_ < e]
    ov = sum(min(ke, e) - max(ks_, s) for ks_, ke, n in conc)
    tot_ov += ov
    names = {}
    for _, _, n in conc:
        short = n.split("(")[0].replace("void qv::", "")[:40]
        names[short] = names.get(short, 0) + 1
    top = ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
    print(f"  copy {d.replace('MEMORY_COPY_', ''):>14s}  +{(s - t0) / 1e6:7.2f} .. +{(e - t0) / 1e6:7.2f} ms  ({(e - s) / 1e6:5.2f} ms): "
          f"{len(conc):3d} engine kernels run meanwhile ({ov / 1e6:5.2f} ms of kernel time)  [{top}]")
last_kernel_end = max(e for s, e, n in ks)
print(f"last engine kernel of the step ends at +{(last_kernel_end - t0) / 1e6:.2f} ms; copies in flight after that: "
      f"{sum(max(0, e - max(s, last_kernel_end)) for s, e, d, b in ms) / 1e6:.2f} ms (exposed), {tot_ov / 1e6:.2f} ms of kernel time ran under a bucket copy")
