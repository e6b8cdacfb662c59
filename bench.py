#!/usr/bin/env python3
"""bench.py - QAT student step (fwd + bwd + gradient all-reduce) throughput on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1: launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``); rank 0 prints
ONE JSON line.  Workload at N=1 is BASELINE.json configs[1]: ViT-S student + QATWrapper,
batch 256, qnnpack qconfig (per-tensor fake-quant), no teacher; synthetic 224x224x3 images.
A "step" = student forward + label-smoothed CE + backward (+ gradient all-reduce when N>1);
optimizer/clip are outside the metric.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def prepare(wrapper, backend):
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat

    wrapper.train()
    wrapper.qconfig = get_default_qat_qconfig(backend)
    p = prepare_qat(wrapper, inplace=False)
    p.train()
    return p


def roofline_fq(batch, iters=20):
    """Dominant HBM-bound kernel of the student step: the activation fake-quant quantize pass
    on the largest activation (fc1 output [B*197,1536] fp32).  Algorithmic bytes = 8 B/element
    (SURVEY.md section 8(d)); time = HIP events on the launch stream around `iters` launches."""
    from qat_vit_amd import native

    L = native.lib()
    n = batch * 197 * 1536
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    mask = torch.empty((n + 31) // 32 * 4, dtype=torch.uint8, device="cuda")
    mn, mx = torch.tensor([float("inf")], device="cuda"), torch.tensor([float("-inf")], device="cuda")
    sc, zp = torch.ones(1, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    on = torch.ones(1, dtype=torch.int64, device="cuda")
    ws = torch.empty(1 << 12, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call():
        native.check(L.qatvit_fq_forward(x.data_ptr(), y.data_ptr(), mask.data_ptr(), mn.data_ptr(), mx.data_ptr(), sc.data_ptr(),
                                         zp.data_ptr(), on.data_ptr(), on.data_ptr(), 0.01, 0, 255, 1, n, 0, 0, ws.data_ptr(), st), "fq")

    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # one call = min/max pass (reads 4 B/elt) + quantize pass (reads 4, writes 4 B/elt): the
    # algorithmic credit is 8 B/elt for the whole fused op
    achieved = 8.0 * n / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "qatvit_fq_forward (k_minmax_tensor + k_qparams + k_quantize)", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "elements": n, "ms_per_launch": round(ms, 4)}


def cpu_baseline(seconds=15.0):
    """The oracle (CPU restatement of the reference step over torch.ao eager QAT) on this box's
    host cores, BASELINE config C1 shapes (ViT-S, batch 8, qnnpack)."""
    from oracle import step_ref

    # a 1-GPU box owns a 16-core share of the host; more threads than that only oversubscribe
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    p = step_ref.enable_qat(step_ref.build_student("vit_small_patch16_224", seed=0), "qnnpack")
    x = torch.randn(8, 3, 224, 224)
    y = torch.randint(0, 10, (8,))
    step_ref.student_step(p, x, y, None)  # first call initialises observers / allocators
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds and n < 200:
        step_ref.student_step(p, x, y, None)
        n += 1
    dt = time.time() - t0
    return {"value": round(8 * n / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} steps of ViT-S student QAT fwd+bwd at batch 8 (config C1), qnnpack, fp32, torch {torch.__version__} CPU eager"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--backend", default="qnnpack")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import qat_vit_amd
    from qat_vit_amd import functional as F
    from qat_vit_amd import native
    from qat_vit_amd.dp import FQStateSync, GradReducer

    native.lib()  # fail loudly before any timing if the HIP library is missing
    torch.manual_seed(0)
    student = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True).to(dev)
    model = prepare(student, args.backend).to(dev)
    reducer = sync = None
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
        reducer, sync = GradReducer(model), FQStateSync(model)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(args.batch, 3, 224, 224, device=dev, generator=g)
    y = torch.randint(0, 10, (args.batch,), device=dev, generator=g)

    def step():
        for p in model.parameters():
            p.grad = None
        if sync is not None:
            sync.broadcast()
        out = model(x)
        loss, _ = F.kd_ce_loss(out, None, y, 4.0, 0.5, 0.1)
        loss.backward()
        if reducer is not None:
            reducer.wait()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        imgs = args.batch * world * args.steps
        res = {
            "metric": "images/sec QAT student step (fwd+bwd+allreduce)",
            "value": round(imgs / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"vit_small_patch16_224 student + QATWrapper, {args.backend} qconfig (per-tensor fake-quant), no teacher, "
                                   f"batch {args.batch}/GPU, 224x224x3 (BASELINE configs[1])",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
        }
        res["roofline"] = roofline_fq(args.batch)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
