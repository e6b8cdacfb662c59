"""The native engine's data-parallel path with TWO ranks on one MI355X (both processes on cuda:0, gloo
as the transport because RCCL refuses two ranks on one device).  Everything except the transport is the
code bench.py runs at N>1: parameter broadcast, per-step rank-0 fake-quant state broadcast, staged
backward with bucketed all-reduce, averaging (reference DDP at qat_trainer.py:311)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(embed_dim=128, depth=2, num_heads=2, img_size=32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import qat_vit_amd
        from qat_vit_amd import functional as F
        from qat_vit_amd.engine import engine_of
        from tests.util import prepare, rel_l2

        def make(seed):
            torch.manual_seed(seed)
            stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **TINY)
            return prepare(stu.cuda(), "qnnpack")

        def step(p, x, y):
            for t in p.parameters():
                t.grad = None
            out = p(x)
            loss, _ = F.kd_ce_loss(out, None, y, 4.0, 0.5, 0.1)
            loss.backward()
            torch.cuda.synchronize()

        g = torch.Generator().manual_seed(100 + rank)       # each rank its own shard
        x = torch.randn(4, 3, 32, 32, generator=g).cuda()
        y = torch.randint(0, 10, (4,), generator=g).cuda()
        dp = make(7 + rank)                                  # replicas deliberately start DIFFERENT: enable_data_parallel must fix that
        with torch.no_grad():
            dp(x)
        eng = engine_of(dp)
        eng.enable_data_parallel(bucket_bytes=64 << 10)
        local = make(7)                                      # rank 0's initial weights, no DP
        fresh = {k: v.clone() for k, v in local.state_dict().items()}
        dp.load_state_dict(fresh)                            # drop the state the engine-building forward observed
        dist.broadcast_object_list([None], src=0)            # (barrier)
        for p in eng.params:
            dist.broadcast(p.data, src=0)
        ok, msgs = True, []
        for it in range(2):
            # what the step must equal: rank 0's fake-quant state, this rank's shard, gradients averaged over ranks
            src = [{k: v.cpu() for k, v in local.state_dict().items() if "activation_post_process" in k or "weight_fake_quant" in k}]
            dist.broadcast_object_list(src, src=0)
            local.load_state_dict({k: v.cuda() for k, v in src[0].items()}, strict=False)
            step(local, x, y)
            want = []
            for p in local.parameters():
                t = p.grad.detach().cpu().clone()
                dist.all_reduce(t)
                want.append(t / world)
            step(dp, x, y)
            for (n, p), w in zip(dp.named_parameters(), want):
                e = rel_l2(p.grad.cpu(), w)
                if not e < 1e-5:                              # fp32 atomics order differs run to run
                    ok = False
                    msgs.append(f"it{it} grad {n} {e:.2e}")
            for (n, a), (_, b) in zip(local.named_buffers(), dp.named_buffers()):
                if not torch.equal(a, b):
                    ok = False
                    msgs.append(f"it{it} buffer {n}")
        # the reference's own wrapper, unchanged: DDP(prepared) (qat_trainer.py:311) around the native module.  torch's reducer then does
        # the bucketing / all-reduce and broadcasts the fake-quant buffers from rank 0 before every forward, in place, into the very
        # tensors the engine reads by pointer.
        from torch.nn.parallel import DistributedDataParallel as DDP
        inner = make(11 + rank)
        ddp = DDP(inner, device_ids=[0])                      # construction broadcasts rank 0's parameters and buffers
        local2 = make(11)
        for it in range(2):
            src = [{k: v.cpu() for k, v in local2.state_dict().items() if "activation_post_process" in k or "weight_fake_quant" in k}]
            dist.broadcast_object_list(src, src=0)
            local2.load_state_dict({k: v.cuda() for k, v in src[0].items()}, strict=False)
            step(local2, x, y)
            want = []
            for p in local2.parameters():
                t = p.grad.detach().cpu().clone()
                dist.all_reduce(t)
                want.append(t / world)
            step(ddp, x, y)
            for (n, p), w in zip(inner.named_parameters(), want):
                e = rel_l2(p.grad.cpu(), w)
                if not e < 1e-5:
                    ok = False
                    msgs.append(f"stock-DDP it{it} grad {n} {e:.2e}")
            for (n, a), (_, b) in zip(local2.named_buffers(), inner.named_buffers()):
                if not torch.equal(a, b):
                    ok = False
                    msgs.append(f"stock-DDP it{it} buffer {n}")
        if engine_of(inner) is None:
            ok = False
            msgs.append("stock-DDP: native engine not bound")
        # evaluation on rank 0 only (qat_trainer.py:370-371): must not issue a collective (would hang: rank 1 never joins)
        if rank == 0:
            dp.eval()
            with torch.no_grad():
                out = dp(x)
            torch.cuda.synchronize()
            ok &= bool(torch.isfinite(out).all())
        dist.barrier()
        q.put((rank, ok, msgs[:8]))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, False, [traceback.format_exc()[-1500:]]))


@pytest.mark.timeout(600)
def test_engine_two_ranks_one_gpu(native_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res


def _worker_dy16(rank, world, port, q):
    """Two ranks, a configuration the one-plane backward covers: calibrating step, one-plane step, then the scale history is wrecked on rank 1 ONLY - both ranks must
    repeat that backward in the pair form (they agree on the flag through the c10d store, each from its pinned mirror) and end up with the same averaged gradients."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ctypes
        import warnings
        import qat_vit_amd
        from qat_vit_amd import functional as F
        from qat_vit_amd.engine import engine_of
        from tests.util import prepare

        torch.manual_seed(5)
        dp = prepare(qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, depth=2).cuda(), "qnnpack")
        g = torch.Generator().manual_seed(200 + rank)
        x = torch.randn(4, 3, 224, 224, generator=g).cuda()
        y = torch.randint(0, 10, (4,), generator=g).cuda()
        with torch.no_grad():
            dp(x)
        eng = engine_of(dp)
        eng.enable_data_parallel(bucket_bytes=4 << 20)
        msgs = []
        ok = eng.dy16 and eng._agree_store is not None and eng._mirror_np is not None
        if not ok:
            msgs.append(f"dy16 {eng.dy16} store {eng._agree_store is not None} mirror {eng._mirror_np is not None}")

        def step():
            for t in dp.parameters():
                t.grad = None
            loss, _ = F.kd_ce_loss(dp(x), None, y, 4.0, 0.5, 0.1)
            loss.backward()
            torch.cuda.synchronize()

        step(); step()                                       # calibrate, one-plane
        ok &= eng._fwd_x16 and eng.dy16_fallbacks == 0
        if rank == 1:                                        # every tensor's previous maximum 2^-60 on this rank: its planes overflow
            c = eng.cfg
            off = eng.lib.qatvit_student_tensor_offset(ctypes.byref(c), b"dy16", 0)
            st = eng.workspace[off:off + 4 * (64 + 256 * 4 * c.depth)].view(torch.float32)
            for t in range(4 * c.depth):
                st[64 + 256 * t + 3] = 2.0 ** -60
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            step()
        if eng.dy16_fallbacks != 1:
            ok = False
            msgs.append(f"rank {rank}: fallbacks {eng.dy16_fallbacks} (every rank must repeat the backward)")
        flat = torch.cat([p.grad.detach().flatten() for p in dp.parameters()]).cpu()
        if not torch.isfinite(flat).all():
            ok = False
            msgs.append("non-finite gradients after the fallback")
        other = [None, None]
        dist.all_gather_object(other, flat)
        if not torch.equal(other[0], other[1]):
            ok = False
            msgs.append("the ranks hold different averaged gradients")
        step()                                               # ... and the group is back on the one-plane form
        ok &= eng._fwd_x16 and eng.dy16_fallbacks == 1
        dist.barrier()
        q.put((rank, bool(ok), msgs[:8]))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, False, [traceback.format_exc()[-1500:]]))


@pytest.mark.timeout(600)
def test_two_ranks_agree_on_the_overflow_fallback(native_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dy16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
