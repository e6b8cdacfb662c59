"""N>1 data-parallel path on CPU: 2 ranks over gloo.  Pins (against torch DDP run on the same
shards) the two things DP adds to the step: averaged gradients and rank-0-authoritative
fake-quant buffers."""
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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from torch.nn.parallel import DistributedDataParallel as DDP

    from oracle import step_ref
    from oracle.vit_ref import RefVisionTransformer, randomize_
    from qat_vit_amd.dp import FQStateSync, GradReducer

    def make():
        torch.manual_seed(3)
        w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 3))
        return step_ref.enable_qat(w, "qnnpack")

    g = torch.Generator().manual_seed(100 + rank)  # each rank its own shard
    x = torch.randn(4, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (4,), generator=g)

    ref = DDP(make())                       # what the reference does (qat_trainer.py:311)
    ours = make()                           # same compute, our DP glue (compute = oracle: CPU test of the DP logic only)
    red, sync = GradReducer(ours, bucket_bytes=64 << 10), FQStateSync(ours)
    assert len(red.buckets) > 1
    ok = True
    for step in range(3):
        step_ref.student_step(ref, x, y, None)
        sync.broadcast()
        step_ref.student_step(ours, x, y, None)
        red.wait()
        for (n, a), (_, b) in zip(ref.module.named_parameters(), ours.named_parameters()):
            ok &= torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7)
        for (n, a), (_, b) in zip(ref.module.named_buffers(), ours.named_buffers()):
            ok &= torch.equal(a, b)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_ddp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res
