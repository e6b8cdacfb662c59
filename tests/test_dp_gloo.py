"""N>1 data-parallel path on CPU: 2 ranks over gloo, driving the ENGINE's own host logic (qat_vit_amd.engine:
FlatGradLayout, staged_backward_allreduce, rehome_fq_state, broadcast_fq_state) - the code bench.py runs over RCCL -
with the oracle as the stand-in for the native compute.  Pinned against torch DDP run on the same shards (what the
reference does, qat_trainer.py:311): averaged gradients and rank-0-authoritative fake-quant buffers."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, backend_cfg):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        from torch.nn.parallel import DistributedDataParallel as DDP

        import qat_vit_amd  # noqa: F401
        from oracle import step_ref
        from oracle.vit_ref import RefVisionTransformer, randomize_
        from qat_vit_amd.engine import FlatGradLayout, broadcast_fq_state, collect_student, rehome_fq_state, staged_backward_allreduce

        def make(seed):
            torch.manual_seed(seed)
            w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), seed))
            return step_ref.enable_qat(w, backend_cfg)

        g = torch.Generator().manual_seed(100 + rank)  # each rank its own shard
        x = torch.randn(4, 3, 32, 32, generator=g)
        y = torch.randint(0, 10, (4,), generator=g)

        ref = DDP(make(3))                      # what the reference does (qat_trainer.py:311)
        ours = make(3 + rank)                   # replicas deliberately start different: the parameter broadcast must fix that
        params, act, wfq = collect_student(ours)
        for p in params:                        # StudentEngine.enable_data_parallel
            dist.broadcast(p.data, src=0)
        _, _, arena, tot = rehome_fq_state(params, act, wfq, torch.device("cpu"))
        assert tot >= len(act) + len(wfq) and arena.numel() == 16 * tot
        layout = FlatGradLayout([p.numel() for p in params], depth=2)
        buckets = layout.buckets(64 << 10)
        assert len(buckets) > 1 and buckets[0][0] == 0 and buckets[-1][1] == layout.last_stage
        assert all(b[2] == a[3] for a, b in zip(buckets, buckets[1:])) and buckets[-1][3] == layout.numel   # contiguous cover
        stage_of = {}
        for slot, pi in enumerate(layout.order):
            stage_of[pi] = 0 if slot < 4 else (layout.last_stage if slot >= len(layout.order) - 4 else 1 + (slot - 4) // 12)
        ok, msgs = True, []
        for step in range(3):
            step_ref.student_step(ref, x, y, None)
            broadcast_fq_state(arena, None)     # StudentEngine.forward under grad mode
            step_ref.student_step(ours, x, y, None)
            flat = torch.zeros(layout.numel)
            views = layout.views(flat, [p.shape for p in params])
            ran = []

            def run_stages(s0, s1):             # stand-in for qatvit_student_backward(stage_from, stage_to): the oracle's gradients
                ran.append((s0, s1))
                for pi, p in enumerate(params):
                    if s0 <= stage_of[pi] <= s1:
                        views[pi].copy_(p.grad)

            staged_backward_allreduce(flat, layout, 64 << 10, None, run_stages)
            if ran != [(b[0], b[1]) for b in buckets]:
                ok = False
                msgs.append(f"stage calls {ran}")
            ref_named = dict(ref.module.named_parameters())
            index_of = {id(p): pi for pi, p in enumerate(params)}
            for n, p in ours.named_parameters():
                pi = index_of.get(id(p))
                if pi is None or not torch.allclose(ref_named[n].grad, views[pi], rtol=1e-5, atol=1e-7):
                    ok = False
                    msgs.append(f"step {step} grad {n}")
            for (n, a), (_, b) in zip(ref.module.named_buffers(), ours.named_buffers()):
                if not torch.equal(a, b.reshape(a.shape)):
                    ok = False
                    msgs.append(f"step {step} buffer {n}")
        # state_dict keys / shapes unchanged by the re-homing
        for (ka, va), (kb, vb) in zip(ref.module.state_dict().items(), ours.state_dict().items()):
            if ka != kb or va.shape != vb.shape:
                ok = False
                msgs.append(f"state_dict {ka} {kb}")
        q.put((rank, bool(ok), msgs[:6]))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, False, [traceback.format_exc()[-2000:]]))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_two_rank_gloo_matches_ddp(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res


def test_flat_grad_layout_vit_small():
    """ViT-S: 152 parameters, 21,669,514 elements; default 16 MiB buckets close at block boundaries, in backward order."""
    import qat_vit_amd  # noqa: F401
    from qat_vit_amd.engine import FlatGradLayout

    D, H, C = 384, 1536, 10
    numels = [D * 768, D, D, 197 * D]
    for _ in range(12):
        numels += [D, D, 3 * D * D, 3 * D, D * D, D, D, D, H * D, H, D * H, D]
    numels += [D, D, C * D, C]
    assert sum(numels) == 21_669_514
    lay = FlatGradLayout(numels, 12)
    b = lay.buckets(16 << 20)
    assert b[0][0] == 0 and b[-1][1] == 13 and b[-1][3] == lay.numel
    assert all((e - s) * 4 >= 16 << 20 for _, _, s, e in b[:-1])
    assert [(s0, s1) for s0, s1, _, _ in b] == [(0, 3), (4, 6), (7, 9), (10, 12), (13, 13)]   # 86.7 MB in four 21.3 MB slices + the embedding
    with pytest.raises(ValueError):
        FlatGradLayout(numels[:-1], 12)
