#!/usr/bin/env python3
"""bench.py - QAT student step (fwd + bwd + gradient all-reduce) throughput on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1: launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``); rank 0 prints
ONE JSON line.  Workload at N=1 is BASELINE.json configs[1]: ViT-S student + QATWrapper,
batch 256, qnnpack qconfig (per-tensor fake-quant), no teacher; synthetic 224x224x3 images.
A "step" = student forward + label-smoothed CE + backward (+ bucketed RCCL gradient all-reduce
overlapped with backward and the rank-0 fake-quant-state broadcast when N>1); optimizer/clip are
outside the metric.  `--backend x86 --teacher` gives config C3/C4
(per-channel weights, [0,127] activations, KD against a frozen ViT-B teacher: native forward, fp16 activation pair x fp16 weights on MFMA by default - qat-vit_amd/teacher.py).

Started directly with ``--gpus N`` (N > 1, no WORLD_SIZE in the environment) it launches its own N workers - the reference starts
its workers itself too (scripts/train_final.sh:13, torchrun --standalone) - BEFORE any GPU call; the parent never touches the GPU,
relays rank 0's JSON line and exits non-zero if any worker fails.

The headline steps run with NO profiling hooks; per-kernel HIP-event timing (roofline / mfma_gemms) and the extra configurations
(C3 with its teacher share, C5) run afterwards in separate, untimed-for-the-headline steps.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # multi-process GPU work on this host needs dmabuf IPC (RCCL)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
I8_PEAK_TOPS = 5000.0      # dense int8 MFMA (v_mfma_i32_16x16x64_i8: twice the bf16 rate)


def prepare(wrapper, backend):
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat

    wrapper.train()
    wrapper.qconfig = get_default_qat_qconfig(backend)
    p = prepare_qat(wrapper, inplace=False)
    p.train()
    return p


def hbm_kernel_rates(batch, iters=20):
    """Achieved HBM GB/s of the stand-alone fake-quant and LayerNorm kernels at the step's largest shapes
    (algorithmic bytes: FQ 8 B/elt, LN fwd 8 B/elt, LN bwd 12 B/elt - SURVEY.md section 8(d))."""
    from qat_vit_amd import native

    L = native.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = {}

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    n = batch * 197 * 1536
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    mask = torch.empty((n + 31) // 32 * 4, dtype=torch.uint8, device="cuda")
    mn, mx = torch.tensor([float("inf")], device="cuda"), torch.tensor([float("-inf")], device="cuda")
    sc, zp = torch.ones(1, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    on = torch.ones(1, dtype=torch.int64, device="cuda")
    ws = torch.empty(1 << 12, dtype=torch.uint8, device="cuda")
    ms = timed(lambda: native.check(L.qatvit_fq_forward(x.data_ptr(), y.data_ptr(), mask.data_ptr(), mn.data_ptr(), mx.data_ptr(), sc.data_ptr(),
                                                        zp.data_ptr(), on.data_ptr(), on.data_ptr(), 0.01, 0, 255, 1, n, 0, 0, ws.data_ptr(), st), "fq"))
    out["fake_quant_fwd_GBps"] = round(8.0 * n / (ms * 1e-3) / 1e9, 1)
    ms = timed(lambda: native.check(L.qatvit_fq_backward(x.data_ptr(), mask.data_ptr(), y.data_ptr(), n, st), "fqb"))
    out["fake_quant_bwd_GBps"] = round(8.0 * n / (ms * 1e-3) / 1e9, 1)
    rows, D = batch * 197, 384
    xx = torch.randn(rows, D, device="cuda")
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    yy, mean, rstd = torch.empty_like(xx), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ms = timed(lambda: native.check(L.qatvit_ln_forward(xx.data_ptr(), g.data_ptr(), b.data_ptr(), yy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, 1e-6, st), "ln"))
    out["layernorm_fwd_GBps"] = round(8.0 * rows * D / (ms * 1e-3) / 1e9, 1)
    ms = timed(lambda: native.check(L.qatvit_ln_backward(yy.data_ptr(), xx.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), yy.data_ptr(),
                                                         dg.data_ptr(), db.data_ptr(), rows, D, st), "lnb"))
    out["layernorm_bwd_GBps"] = round(12.0 * rows * D / (ms * 1e-3) / 1e9, 1)
    out["hbm_peak_GBps"] = HBM_PEAK_GBS
    # clip + AdamW over a ViT-S sized parameter set (outside the timed step: SURVEY 8(d)); 4 B/param norm + 28 B/param update
    from qat_vit_amd.optim import ClipAdamW

    ps = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.02) for s in [(384, 768)] + [(1152, 384), (384, 384), (1536, 384), (384, 1536)] * 12]
    for p in ps:
        p.grad = torch.randn_like(p)
    opt = ClipAdamW(ps, lr=1.5e-4, weight_decay=1e-3)
    ms = timed(lambda: opt.step(max_norm=1.0))
    out["clip_adamw_GBps"] = round(32.0 * sum(p.numel() for p in ps) / (ms * 1e-3) / 1e9, 1)
    out["clip_adamw_ms_vit_small"] = round(ms, 3)
    return out


def cpu_baseline(seconds=12.0):
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


def spawn_workers(n, argv):
    """``bench.py --gpus N`` started directly: run N workers (one per GPU, RCCL rendezvous on 127.0.0.1) under the watchdog of
    qat_vit_amd.launch (first failing rank stops the rest; wall-clock limit), relay rank 0's stdout.
    Nothing in this process has touched the GPU (importing torch does not)."""
    from qat_vit_amd.launch import run_workers

    rc, out0 = run_workers(n, [sys.executable, os.path.abspath(__file__)] + argv, wall_limit_s=float(os.environ.get("BENCH_WALL_LIMIT_S", "3000")))
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc == 0 and '"metric"' not in out0:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    raise SystemExit(rc)


def build_model(qat_vit_amd, student, backend, dev):
    if student == "vit_small":
        stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True).to(dev)
    else:
        stu = qat_vit_amd.create_model("vit_base_patch16_224_teacher", pretrained=False, num_classes=10, qat_wrapper=True).to(dev)
    return prepare(stu, backend).to(dev)


def timed_steps(step, steps, warmup, world, dev):
    """W untimed steps, then exactly K timed ones bracketed by barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--backend", default="qnnpack")
    ap.add_argument("--teacher", action="store_true", help="KD against a frozen ViT-B teacher (configs C3/C4)")
    ap.add_argument("--student", default="vit_small", choices=["vit_small", "vit_base"], help="vit_base = config C5 (use --batch 128)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (launch-bound small batches; N=1, no teacher)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rates", action="store_true")
    ap.add_argument("--no-kernel-legs", action="store_true", help="skip the per-GEMM-class HIP-event legs (no roofline object; profiling tools)")
    ap.add_argument("--no-extras", action="store_true", help="skip the C3 / C5 legs that follow the headline measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_workers(args.gpus, sys.argv[1:])          # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal knobs for a ONE-GPU box (tests the N>1 code path, not its speed): all ranks on device 0, gloo as the transport
    if os.environ.get("BENCH_ONE_DEVICE"):
        local = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        # a dead peer ends the run in minutes, not in c10d's default 10 - 30.  The limit also covers the rank-0-only sections (per-kernel legs' bookkeeping,
        # hbm_kernel_rates: seconds) during which the other ranks already wait in the next collective - hence minutes, not seconds
        tmo = datetime.timedelta(seconds=float(os.environ.get("BENCH_DIST_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    import qat_vit_amd
    from qat_vit_amd import functional as F
    from qat_vit_amd.engine import engine_of
    from qat_vit_amd import native

    L = native.lib()  # fail loudly before any timing if the HIP library is missing
    torch.manual_seed(0)

    def make_config(student, qbackend, with_teacher, batch):
        """(step function, engine) for one configuration; synthetic N(0,1) images generated on the device once, seed 1234 + rank."""
        model = build_model(qat_vit_amd, student, qbackend, dev)
        teacher = None
        if with_teacher:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                teacher = qat_vit_amd.create_teacher("vit", num_classes=10).to(dev).eval()
            for p in teacher.parameters():
                p.requires_grad = False
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        x = torch.randn(batch, 3, 224, 224, device=dev, generator=g)
        y = torch.randint(0, 10, (batch,), device=dev, generator=g)
        with torch.no_grad():
            model(x)  # builds the native engine (workspace, FQ arena)
        eng = engine_of(model)
        if world > 1:
            eng.enable_data_parallel()

        def teacher_only():
            with torch.no_grad():
                return teacher(x)   # eval + no_grad + CUDA -> qatvit_teacher_forward

        def step():
            for p in eng.params:
                p.grad = None
            t_out = teacher_only() if teacher is not None else None
            out = model(x)
            loss, _ = F.kd_ce_loss(out, t_out, y, 4.0, 0.5, 0.1)
            loss.backward()

        return step, eng, model, x, y, (teacher_only if teacher is not None else None)

    step, eng, model, x, y, _ = make_config(args.student, args.backend, args.teacher, args.batch)
    if args.graph:
        if world > 1 or args.teacher:
            raise SystemExit("--graph: single GPU, no teacher")
        from qat_vit_amd.graph import GraphedStudentStep

        gstep = GraphedStudentStep(model, x, y)

        def step():  # noqa: F811
            gstep(x, y)

    dt = timed_steps(step, args.steps, args.warmup, world, dev)     # the headline: no profiling hooks are active

    # ---- N > 1: the record proves what ran - group size, transport, library version, one distinct device per rank, the bucket plan and the collective
    # time left exposed behind the last backward kernel (HIP events on the compute stream, extra steps outside the headline)
    rccl = None
    if world > 1:
        pr = torch.cuda.get_device_properties(local)
        ident = ":".join(f"{getattr(pr, a, -1):02x}" for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")) + f" uuid={getattr(pr, 'uuid', '?')}"
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "local_rank": local, "device": ident, "name": pr.name})
        eng.exposed_events = []
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        exp_ms = [a.elapsed_time(b) for a, b in eng.exposed_events]
        eng.exposed_events = None
        t = torch.tensor([sum(exp_ms) / max(1, len(exp_ms))], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        one_device = bool(os.environ.get("BENCH_ONE_DEVICE"))
        if not one_device:
            assert len({d["device"] for d in devs}) == world, f"ranks share a device: {devs}"
        rccl = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                "devices": devs, "one_device_rehearsal": one_device,
                "bucket_bytes": eng.bucket_bytes, "bucket_count": len(eng.layout.buckets(eng.bucket_bytes)),
                "gradient_bytes_per_step": eng.grad_numel * 4, "fq_state_broadcast_bytes_per_step": eng.fq_arena.numel(),
                "exposed_allreduce_ms": round(float(t.item()), 4),
                "exposed_note": "stream time from the end of the last backward kernel to the join of the last bucket's all-reduce, mean of 3 extra steps, MAX over ranks"}

    ws = eng.workspace.data_ptr()

    def profile_kind(kind, nsteps):
        """HIP events around every launch of one GEMM class, on the stream it is launched on, in `nsteps` extra steps (outside the headline)."""
        ms, cnt, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        native.check(L.qatvit_profile_start(ws, kind, (8 * eng.cfg.depth + 16) * nsteps), "profile_start")
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        native.check(L.qatvit_profile_stop(ws, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl)), "profile_stop")
        return ms.value, cnt.value, fl.value

    res = None
    if rank == 0:
        imgs = args.batch * world * args.steps
        res = {
            "metric": "images/sec QAT student step (fwd+bwd+allreduce)",
            "value": round(imgs / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (emulated: int8 MFMA for grid x grid products; forward float operands as fp16 hi+lo pairs; backward gradients as ONE fp16 plane scaled per "
                      "tensor - 2^-12 per element, <= 4e-4 relative L2 per stage against the fp32 reference, bf16-pair fallback on overflow; fp32 / exact-integer accumulate)")
                     if eng.dy16 else
                     "f32 (emulated: int8 MFMA for grid x grid products, 16-bit MFMA on exact-grid / hi+lo split operands; exact-integer or fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{args.student}_patch16_224 student + QATWrapper, {args.backend} qconfig, "
                                   f"{'vit_base teacher KD (native teacher forward inside the timed step)' if args.teacher else 'no teacher'}, "
                                   f"batch {args.batch}/GPU, 224x224x3 (BASELINE configs[{2 if args.teacher else 1}])",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "hipgraph": bool(args.graph),
                       "backward_form": "one fp16 plane per gradient tensor (QATVIT_BWD_DY16)" if eng.dy16 else "bf16 (hi, lo) pairs",
                       "one_plane_fallbacks_in_timed_region": eng.dy16_fallbacks},
        }
        if rccl is not None:
            res["rccl"] = rccl
    # ---- per-kernel legs: every rank runs the same extra steps (collectives stay matched), rank 0 times its own launches
    KINDS = {1: "k_gemm_nt, plain epilogue (proj / fc2 forward on fp16 pairs; proj dgrad on one fp16 plane - bf16 pairs with QATVIT_DY16=0)",
             4: "k_gemm_nt dgrad + LayerNorm backward fused into the epilogue (fc1 / qkv dgrad, mode 8; gradient operand = one fp16 plane, or a bf16 pair with QATVIT_DY16=0)",
             5: "k_gemm_nt fc2 dgrad + GELU backward fused into the epilogue (mode 9: codes + mask bits; gradient operand / output = one fp16 plane, or bf16 pairs with QATVIT_DY16=0)",
             2: "k_gemm_nt grid A on int8 MFMA, plain epilogue (patch embedding; qkv when it runs once)",
             7: "k_i8_strip<3> A-stationary int8 strip kernel, statistics-only pass (qkv and fc1 first passes; csrc/i8strip.hip)",
             8: "k_i8_strip<4> A-stationary int8 strip kernel, fc1 code pass (gelu(fq(.)) as uint8 codes + STE mask bits + two 256-entry tables)",
             9: "k_i8_strip<7> A-stationary int8 strip kernel, qkv code pass (uint8 codes + STE mask bits in the attention layout)",
             3: "k_tn_stream<0> + fix-up: ALL qkv / fc1 weight gradients of the backward call in one persistent stream-K launch (X = the forward's int8 plane, expanded in "
                "registers; QATVIT_TN_STREAM=0: k_gemm_tn_q8<0> + k_tn_reduce per GEMM), + the patch-embedding weight gradient (k_gemm_tn, bf16 pair)",
             6: "k_tn_stream<1> / <2> + fix-ups: all fc2 (X = codes through a bank-replicated table) / proj (X fp16) weight gradients of the backward call, one persistent "
                "stream-K launch each (QATVIT_TN_STREAM=0: one launch + k_tn_reduce per GEMM; QATVIT_DY16=0: 3 bf16 passes)"}
    SHORT = {1: "nt_split_plain", 4: "nt_split_dgrad_fused_layernorm_bwd", 5: "nt_split_dgrad_fused_gelu_bwd", 2: "nt_int8_plain", 7: "nt_int8_stats_pass",
             8: "nt_int8_fc1_store_pass", 9: "nt_int8_qkv_code_pass", 3: "tn_grid_x", 6: "tn_split_x"}
    prof = {}
    nprof = 0 if (args.graph or args.no_kernel_legs) else 3
    for kind in KINDS:
        if nprof == 0:
            break
        if rank == 0:
            prof[kind] = profile_kind(kind, nprof)
        else:
            for _ in range(nprof):
                step()
    if rank == 0 and prof:
        c = eng.cfg
        T = (c.img_size // c.patch_size) ** 2 + 1
        Mr, Dm, Hd, dep = args.batch * T, c.embed_dim, c.mlp_hidden, c.depth
        Kpe, Mpe = c.in_chans * c.patch_size ** 2, args.batch * (T - 1)
        codes = os.environ.get("QATVIT_FC2_CODES", "1") != "0" and os.environ.get("QATVIT_F16", "1") != "0"
        bits = codes and os.environ.get("QATVIT_FC1_BITS", "1") != "0" and os.environ.get("QATVIT_I8", "1") != "0"      # fc1 codes for the backward: byte plane + mask bits
        fc2w = bits and os.environ.get("QATVIT_FC2W_CODES", "1") != "0"                                                # fc2 wgrad from the byte plane: no bf16 pair of gelu(fq(fc1))
        qkv2 = os.environ.get("QATVIT_QKV_2PASS", "1") != "0" and os.environ.get("QATVIT_ATTN_CODES", "1") != "0" and os.environ.get("QATVIT_I8", "1") != "0"
        # ALGORITHMIC HBM bytes per step of each class: every operand once, in the format the kernel reads / writes it;
        # weights once per launch; split-reduction partials, mask bit planes (1/32 of an fp32 plane) and re-reads are NOT counted
        dy = 2 if eng.dy16 else 4                            # bytes per element of a backward gradient operand: one fp16 plane, or a bf16 (hi, lo) pair
        xf = 2 if eng.dy16 else 4                            # ... of the float X operand of the proj weight gradient (fp16 / bf16 pair); fc2's X is codes either way
        xg = 1 if (eng.dy16 and os.environ.get("QATVIT_TN_Q8", "1") != "0" and Dm % 384 == 0) else 2   # ... of the grid X operand of the qkv / fc1 weight gradients: int8 plane, or 16-bit integers
        lnb = 3 * Mr * Dm * 4 + Mr * Dm * dy                 # fused LayerNorm backward: x, dx_in read; dx_out and the masked gradient for the next branch written
        step_bytes = {
            1: dep * ((Mr * Dm * 4 + Dm * Dm * 2 + Mr * Dm * 4)                                  # proj forward: fp16 pair in, fp32 out
                      + (Mr * Hd * (1 if codes else 4) + Dm * Hd * 2 + Mr * Dm * 4)              # fc2 forward: codes (or fp16 pair) in, fp32 out
                      + (Mr * Dm * dy + Dm * Dm * 2 + Mr * Dm * 4)),                             # proj dgrad: gradient in, fp32 out
            4: dep * ((Mr * Hd * dy + Dm * Hd * 2 + lnb) + (Mr * 3 * Dm * dy + 3 * Dm * Dm * 2 + lnb)),   # fc1 dgrad, qkv dgrad (+ LayerNorm backward)
            5: dep * (Mr * Dm * dy + Dm * Hd * 2 + (Mr * Hd * 9 // 8 if bits else Mr * Hd * 2) + Mr * Hd * dy),   # fc2 dgrad: gradient in, codes (+ mask bits) in, gradient out
            2: (Mpe * Kpe + Dm * Kpe + Mpe * Dm * 4) + (0 if qkv2 else dep * (Mr * Dm + 3 * Dm * Dm + Mr * 3 * Dm * 4)),   # patch embedding (+ one-pass qkv: fp32 out)
            7: dep * ((Mr * Dm + Hd * Dm) + ((Mr * Dm + 3 * Dm * Dm) if qkv2 else 0)),          # statistics passes: operands in, nothing stored
            8: dep * (Mr * Dm + Hd * Dm + Mr * Hd * (1 if codes else 4) + (Mr * Hd // 8 if bits else Mr * Hd * 2) + (0 if fc2w else Mr * Hd * 4)),   # fc1 storing pass: codes (or fp16 pair) + mask bits (or uint16 code) [+ bf16 pair]
            9: dep * (Mr * Dm + 3 * Dm * Dm + Mr * 3 * Dm * 9 // 8),                            # qkv code pass: 1 B + 1 bit per element out
            3: (Mpe * Dm * 4 + Mpe * Kpe * 2 + Dm * Kpe * 4)
               + dep * ((Mr * 3 * Dm * dy + Mr * Dm * xg + 3 * Dm * Dm * 4) + (Mr * Hd * dy + Mr * Dm * xg + Hd * Dm * 4)),   # qkv, fc1 wgrad
            6: dep * ((Mr * Dm * dy + Mr * Dm * xf + Dm * Dm * 4) + (Mr * Dm * dy + Mr * Hd * (1 if fc2w else 4) + Hd * Dm * 4)),   # proj, fc2 wgrad (Q as codes)
        }
        gemms = {}
        for kind, (ms, cnt, fl) in prof.items():
            if ms <= 0 or cnt == 0:
                continue
            rate = fl / (ms * 1e-3) / 1e12
            peak = I8_PEAK_TOPS if kind in (2, 7, 8, 9) else BF16_PEAK_TFLOPS
            lps = cnt / nprof
            by = step_bytes[kind] / lps if (args.student in ("vit_small", "vit_base") and lps > 0) else None
            g = {"kernel": KINDS[kind], "ms_per_step": round(ms / nprof, 3), "launches_per_step": round(lps, 1),
                 "avg_us_per_launch": round(1e3 * ms / cnt, 1), "algorithmic_T(FL)OPs": round(rate, 1),
                 "peak": peak, "frac_of_peak": round(rate / peak, 4)}
            if by:
                gbs = by / (ms * 1e-3 / cnt) / 1e9
                ai = (fl / cnt) / by
                g.update({"algorithmic_MB_per_launch": round(by / 1e6, 1), "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                          "flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(peak * 1e12 / (HBM_PEAK_GBS * 1e9), 1),
                          "roofline_bound": "hbm" if ai < peak * 1e12 / (HBM_PEAK_GBS * 1e9) else "mfma"})
            gemms[kind] = g
        if 7 in gemms:
            gemms[7]["note"] = "fc1 and qkv run twice (statistics-only pass + storing pass): the statistics passes count as time, not as algorithmic work"
        for k in (3, 6):
            if k in gemms:
                gemms[k]["note"] = "the bracket holds the GEMM launch and its fix-up / reduction launch (ordered summation of the split tiles)"
        dom = max(gemms, key=lambda k: gemms[k]["ms_per_step"])          # the dominant kernel of the step = the GEMM class with the largest time
        g = gemms[dom]
        hbm_bound = g.get("roofline_bound") == "hbm"
        traffic, traffic_note = None, None
        try:   # HBM-side bytes per launch: NOT measured by this run - offline rocprofv3 --pmc passes at B=256 shapes (tools/pmc_traffic.sh)
            pm = json.load(open(os.path.join(ROOT, "profiles", "round4_gemm_pmc_traffic.json")))
            if args.batch == 256 and args.student == "vit_small" and str(dom) in pm:
                traffic = round(pm[str(dom)]["bytes_per_launch"])
                traffic_note = "STATIC, not measured in this run: " + pm[str(dom)]["note"]
        except Exception:  # noqa: BLE001
            pass
        res["roofline"] = {
            "bound": "hbm" if hbm_bound else "mfma",
            "kernel": "qv::" + g["kernel"] + " - the GEMM kernel with the largest share of the step (208x384 tiles)",
            "achieved": g["algorithmic_GBps"] if hbm_bound else g["algorithmic_T(FL)OPs"],
            "peak": HBM_PEAK_GBS if hbm_bound else g["peak"],
            "unit": "GB/s" if hbm_bound else ("TOP/s" if dom in (2, 7, 8, 9) else "TFLOP/s"),
            "frac": g["frac_of_hbm_peak"] if hbm_bound else g["frac_of_peak"],
            "traffic": traffic, "traffic_note": traffic_note, "launches": int(g["launches_per_step"] * nprof), "avg_us_per_launch": g["avg_us_per_launch"],
            "flop_per_byte": g.get("flop_per_byte"), "ridge_flop_per_byte": g.get("ridge_flop_per_byte"),
            "mfma_side": {"achieved": g["algorithmic_T(FL)OPs"], "peak": g["peak"], "frac": g["frac_of_peak"]},
            "note": f"per launch: algorithmic bytes (every operand once in its stored format: DESIGN.md section 4) and algorithmic FLOPs 2*M*N*K / HIP-event time of that "
                    f"launch on its launch stream, {nprof} steps run right after the timed region (no event is recorded inside the timed region).  The bound is the "
                    "roofline that binds at the kernel's arithmetic intensity (flop_per_byte against the ridge peak_flops / peak_bytes); the other side is in "
                    "mfma_side.  Launches on a (hi, lo) pair issue two 16-bit MFMA passes (issued MFMA work 2x the algorithmic figure); the one-plane backward issues one",
        }
        res["mfma_gemms"] = {SHORT[k]: v for k, v in gemms.items()}
    if rank == 0 and not args.no_kernel_rates:
        res["hbm_kernels"] = hbm_kernel_rates(args.batch)
    # ---- the other single-GPU-sized configurations of BASELINE.json, on the same clock discipline (fewer steps): every rank runs them
    if not args.no_extras and not args.graph and not args.teacher and args.student == "vit_small" and args.batch == 256:
        del step, model, x, y
        eng.workspace = None
        del eng
        torch.cuda.empty_cache()
        extras = {}
        for tag, cfgs in (("C3" if world == 1 else "C4", ("vit_small", "x86", True, 256)), ("C5", ("vit_base", "x86", False, 128))):
            st2, eng2, _m, _x, _y, t_only = make_config(*cfgs)
            k = max(5, args.steps // 4)
            d2 = timed_steps(st2, k, 3, world, dev)
            e = {"value": round(cfgs[3] * world * k / d2, 2), "unit": "images/sec", "ms_per_step": round(1e3 * d2 / k, 3), "steps": k, "warmup": 3,
                 "n_gpus": world, "workload": f"{cfgs[0]}_patch16_224 student, {cfgs[1]} qconfig (per-channel weight fake-quant, [0,127] activations), "
                                              f"{'vit_base teacher KD inside the step' if cfgs[2] else 'no teacher'}, batch {cfgs[3]}/GPU"}
            if t_only is not None:
                d3 = timed_steps(t_only, k, 2, 1, dev)
                e["teacher_forward_ms"] = round(1e3 * d3 / k, 3)
                e["student_step_ms"] = round(1e3 * (d2 - d3) / k, 3)
                from qat_vit_amd import teacher as _teacher
                t_eng = next(iter(_teacher._ENGINES.values()), None)   # the engine that ran (its effective form: the fp16 forms fall back to 3 passes for dims % 384 != 0)
                e["teacher_form"] = {3: "bf16 pairs x bf16 pairs, 3 MFMA passes", 2: "fp16 activation pair x fp16 weights, 2 MFMA passes",
                                     1: "fp16 x fp16, 1 MFMA pass"}[t_eng.passes if t_eng is not None else _teacher.DEFAULT_PASSES]
            extras[tag] = e
            del st2, _m, _x, _y, t_only
            eng2.workspace = None
            del eng2
            torch.cuda.empty_cache()
        if world == 1:
            # the exported integer network (Int8Student: int8 MFMA, frozen qparams) next to the fake-quant forward it equals bit for bit
            from torch.ao.quantization import disable_observer

            mi = build_model(qat_vit_amd, "vit_small", "qnnpack", dev)
            xi = torch.randn(256, 3, 224, 224, device=dev)
            with torch.no_grad():
                mi(xi)                                         # one observation so that every quantizer has a range
                mi.apply(disable_observer)
                mi.eval()
                infer = qat_vit_amd.Int8Student(qat_vit_amd.export_int8(mi))
                d_fq = timed_steps(lambda: mi(xi), 20, 3, 1, dev)
                d_i8 = timed_steps(lambda: infer(xi), 20, 3, 1, dev)
                same = bool(torch.equal(mi(xi), infer(xi)))
            extras["INT8_INFERENCE"] = {"value": round(256 * 20 / d_i8, 2), "unit": "images/sec", "ms_per_forward": round(1e3 * d_i8 / 20, 3),
                                        "fake_quant_forward_ms": round(1e3 * d_fq / 20, 3), "logits_bit_identical_to_fake_quant_forward": same,
                                        "workload": "vit_small_patch16_224 exported to int8 (Int8Student), batch 256, forward only"}
            del mi, infer, xi
            torch.cuda.empty_cache()
        if rank == 0:
            res["extra_configs"] = extras
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
