import os, sys, runpy
sys.argv = ["bench_attn.py"]
os.environ["BENCH_N"] = "3"
g = runpy.run_path("tools/bench_attn.py")
import torch
d = g["delta"]; torch.cuda.synchronize()
names = ["loads requested", "loads arrived", "images written", "barrier", "dQ tile of step 0", "2nd barrier step 3", "dQ tile of step 3", "sweep done", "epilogue loads arrived", "stores issued", "stores acked"]
for blk in (5, 700):
    for w, off in ((0, 198), (5, 211)):
        v = d[blk, off:off + 11].tolist()
        print(f"block {blk} wave {w}: " + "  ".join(f"{n}={int(x)}" for n, x in zip(names, v)))
