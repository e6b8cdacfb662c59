"""Teacher-forced, stage-level parity of the native QAT student step at FULL model size (ViT-S, batch 8: BASELINE config C1
shapes with the qnnpack qconfig, config C3 semantics with x86).

The whole-network comparison cannot be tighter than the fp32 noise floor (fake-quant is discontinuous and the 76 activation
quantizers compound one-step flips: tests/test_gpu_step.py).  Here every stage is run on the ORACLE's input instead: the oracle
(oracle/step_ref.py: the reference's wrapper / QAT enable / loss over stock torch.ao, CPU) records the tensor entering every stage in
both directions; each native stage is then executed alone through ``qatvit_student_forward_stages`` / ``qatvit_student_forward_part``
/ ``qatvit_student_backward_stages`` with QATVIT_STAGE_INJECT (include/qatvit.h) on that recorded input, so no upstream drift enters.
Every kernel that runs in the real step (NT epilogue variants 0 / 3 / 4 / 5, int8 / fp16-pair / bf16-pair GEMMs, k_resid_fq_lnstats,
k_ln_apply_quant, attention fwd / bwd, k_ln_bwd_fq with its fused mask output, TN weight gradients, k_embed_bwd, k_head_fwd / bwd) is
compared tensor by tensor.

Granularity.  Even INSIDE one block a single flipped code is amplified: one flipped key element perturbs that head's softmax for all
197 queries, one flipped norm2 element flips ~5 % of its row's 1,536 fc1 codes (measured below, and the same for the stock torch tree
on this GPU against the CPU oracle).  The forward is therefore injected at every fake-quantizer input that follows such an amplifier -
four parts per block (x_in -> norm1 -> qkv | qkv -> attention -> proj -> residual | x_mid -> norm2 -> fc1 -> GELU | GELU out -> fc2 -> residual):

  (a) integer codes of every activation quantizer: fewer than 1e-4 of the elements differ, by at most one step;
  (b) relative L2 <= 1e-3 on every pre-fake-quant tensor, every float GEMM operand, the residual stream, the residual-stream
      gradient and every parameter gradient of the stage (backward: one stage per block on the teacher-forced forward state).

A second table repeats the forward with ONE injection per block next to the same experiment on the stock torch tree - on this GPU and on the
host CPU with the reduction order of every Linear layer permuted (three permutations): independent fp32 evaluations, the live floor DISTRIBUTION of that coarser granularity; twice: free-running (every
quantizer observes for itself; reported with every scale difference and the stock samples' own leave-one-out ratios, asserted in the median) and with the
block's quantizers on the oracle's scales on both sides (asserted per row: within twice the largest sample of ITS OWN block + 1e-4).

Reference call sites: forward ``ddp_model(images)`` qat_trainer.py:341, loss :343-349, ``loss.backward()`` :359.
Tables are written to gpurun_out/ (committed copies: profiles/round4_stage_flip_table_*.txt)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from oracle import step_ref  # noqa: E402
from qat_vit_amd import engine as E  # noqa: E402
from tests.util import capture_fq_io, fq_modules, prepare, qkv_ints, qkv_mask, rel_l2  # noqa: E402

CODE_FLIP_FRAC = 1e-4
TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleTrace:
    """One oracle step with everything a stage needs recorded: inputs of every stage (forward and backward), the float GEMM operands
    between the quantizers, every activation quantizer's (input, output), every parameter gradient."""

    def __init__(self, po, x, y, t):
        m = po.model
        self.block_in, self.g_block_in = {}, {}
        self.proj_in, self.fc2_in, self.norm2_in = {}, {}, {}
        depth = len(m.blocks)
        self.depth = depth

        def keep_grad(store, key):
            def h(g):
                store[key] = g.detach().clone()
            return h

        def pre_block(i):
            def h(mod, inp):
                self.block_in[i] = inp[0].detach().clone()
                inp[0].register_hook(keep_grad(self.g_block_in, i))
            return h

        hs = [b.register_forward_pre_hook(pre_block(i)) for i, b in enumerate(m.blocks)]
        hs.append(m.norm.register_forward_pre_hook(pre_block(depth)))
        for i, b in enumerate(m.blocks):
            hs.append(b.attn.proj.register_forward_pre_hook(lambda mod, inp, i=i: self.proj_in.__setitem__(i, inp[0].detach().clone())))
            hs.append(b.mlp.fc2.register_forward_pre_hook(lambda mod, inp, i=i: self.fc2_in.__setitem__(i, inp[0].detach().clone())))
            hs.append(b.norm2.register_forward_pre_hook(lambda mod, inp, i=i: self.norm2_in.__setitem__(i, inp[0].detach().clone())))
        self.fq_io = capture_fq_io(po)
        for q in po.parameters():
            q.grad = None
        out = po(x)
        self.g_logits = {}
        out.register_hook(keep_grad(self.g_logits, 0))
        if t is None:
            loss = torch.nn.CrossEntropyLoss(label_smoothing=0.1)(out, y)
        else:
            loss, _, _ = step_ref.kd_ce_loss(out, t, y, 4.0, 0.5, 0.1)
        loss.backward()
        self.logits = out.detach()
        self.grads = {n: q.grad.detach().clone() for n, q in po.named_parameters()}
        self.fq = fq_modules(po)
        for h in hs:
            h.remove()

    def codes(self, name):
        """q - zero_point of an activation quantizer's output (integers in fp32)."""
        f = self.fq[name]
        return torch.round(self.fq_io[name][1] / f.scale)

    def pre(self, name):
        return self.fq_io[name][0]


class Table:
    """Collects every comparison; `check()` (after the table has been written) fails the test listing all violations."""

    def __init__(self):
        self.rows = []
        self.bad = []

    def codes(self, stage, name, ours, ref, limit=CODE_FLIP_FRAC):
        ours, ref = ours.float().cpu().reshape(-1), ref.float().cpu().reshape(-1)
        d = (ours - ref).abs()
        frac, mx = (d != 0).float().mean().item(), d.max().item()
        self.rows.append((stage, name, "codes", ours.numel(), frac, mx))
        if not (frac < limit and mx <= 1):
            self.bad.append((stage, name, "codes", frac, mx))

    def close(self, stage, name, ours, ref, tol=TOL):
        e = rel_l2(ours.detach().float().cpu().numpy(), ref.detach().float().cpu().numpy())
        self.rows.append((stage, name, "rel_l2", int(np.prod(ref.shape)), e, 0.0))
        if not e <= tol:
            self.bad.append((stage, name, "rel_l2", e, tol))

    def check(self):
        assert not self.bad, self.bad[:12]

    def write(self, path, header):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(header + "\n")
            f.write(f"{'stage':<14}{'tensor':<34}{'kind':<8}{'elements':>12}{'differing frac / rel L2':>26}{'max |code diff|':>17}\n")
            for r in self.rows:
                st, n, k, ne, v, mx = r[:6]
                fl = f"   floor {r[6]:.3e}" if len(r) > 6 else ""
                f.write(f"{st:<14}{n:<34}{k:<8}{ne:>12}{v:>26.3e}{(int(mx) if k == 'codes' else ''):>17}{fl}\n")
            cf = [r for r in self.rows if r[2] == "codes"]
            tot = sum(r[3] for r in cf)
            f.write(f"\nall activation quantizers: {sum(r[3] * r[4] for r in cf):.0f} differing codes of {tot} ({sum(r[3] * r[4] for r in cf) / tot:.2e}), "
                    f"max |diff| {int(max(r[5] for r in cf))}; worst rel L2 {max(r[4] for r in self.rows if r[2] == 'rel_l2'):.2e}\n")


def _fq_of_codes(pre, fqmod, qmin, qmax):
    """q - zp codes our consumer kernels produce from a pre-FQ tensor and the module's (scale, zero_point): x * (1/s), rint, + zp, clamp."""
    inv = (1.0 / fqmod.scale.float()).to(pre.device)
    zp = fqmod.zero_point.float().to(pre.device)
    return torch.clamp(torch.round(pre * inv) + zp, qmin, qmax) - zp


def _run(backend, seed, teacher, golden_tag, arch="vit_small_patch16_224", B=8, blocks=None):
    """arch / B: model and batch of the harness; blocks: the blocks whose stages are executed and compared (None = all; the embedding, head and
    the whole oracle step always run)."""
    T = 197
    w = step_ref.build_student(arch, seed=seed)
    po = step_ref.enable_qat(w, backend)
    po0 = copy.deepcopy(po)                                             # never-observed copy: source of the stock-torch-on-GPU floor blocks
    if arch == "vit_small_patch16_224":
        stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True)
    else:   # the reference's ViT-B factory (model_registry.py:152-175) without the checkpoint fetch
        stu = qat_vit_amd.create_model("vit_base_patch16_224_teacher", pretrained=False, num_classes=10, qat_wrapper=True)
    stu.load_state_dict(w.state_dict())
    p = prepare(stu.cuda(), backend)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    t = torch.randn(B, 10, generator=g) * 2 if teacher else None
    tr = OracleTrace(po, x, y, t)
    depth, D = tr.depth, p.model.embed_dim
    blocks = list(range(depth)) if blocks is None else list(blocks)
    M = B * T
    eng = E.bind(p, B)
    c = eng.cfg
    qa, qb = c.act_qmin, c.act_qmax
    n_act = len(eng.act_fq)
    tot = eng.fq_total
    fresh = eng.fq_arena.clone()

    def reset_act_observers(idxs):
        """The given activation quantizers (C-ABI act_fq order: quant, patch_embed, 6 per block, norm, head) back to 'never observed',
        so that the stage's one observation equals the oracle's first step.  (Weights keep theirs: an identical weight is an EMA fixed point.)"""
        f32, i32 = eng.fq_arena[:12 * tot].view(torch.float32), eng.fq_arena[12 * tot:].view(torch.int32)
        f0, i0 = fresh[:12 * tot].view(torch.float32), fresh[12 * tot:].view(torch.int32)
        ix = torch.tensor(list(idxs), device=eng.device)
        for k in range(3):
            f32[k * tot:k * tot + n_act][ix] = f0[k * tot:k * tot + n_act][ix]
        i32[:n_act][ix] = i0[:n_act][ix]

    # one ordinary step first: engine, weights and every buffer exist; nothing of it is compared here
    xc, yc = x.cuda(), y.cuda()
    from qat_vit_amd import functional as F
    out = p(xc)
    F.kd_ce_loss(out, None if t is None else t.cuda(), yc, 4.0, 0.5, 0.1)[0].backward()
    fqm = fq_modules(p)
    tab = Table()
    A = "activation_post_process"

    # ================================================================= forward, stage by stage
    reset_act_observers([0, 1])
    eng.forward_stages(xc, 0, 0)
    np_ = T - 1
    ref_img = tr.codes(f"quant.{A}").reshape(B, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(B * np_, 768)
    tab.codes("embed", "quant (input image)", eng.tensor("imgq", 0, (B * np_, 768), torch.bfloat16), ref_img)
    y0_ref = tr.pre(f"model.patch_embed.proj.{A}").permute(0, 2, 3, 1).reshape(-1, D)
    y0 = eng.tensor("Y0", 0, (B * np_, D))
    tab.close("embed", "patch_embed.proj pre-FQ", y0, y0_ref)
    tab.codes("embed", "patch_embed.proj", _fq_of_codes(y0, fqm[f"model.patch_embed.proj.{A}"], qa, qb),
              tr.codes(f"model.patch_embed.proj.{A}").permute(0, 2, 3, 1).reshape(-1, D))
    tab.close("embed", "x_in[0] (cls, pos added)", eng.tensor("x_in", 0, (M, D)), tr.block_in[0].reshape(M, D))
    Hd = c.mlp_hidden
    if os.environ.get("QATVIT_ATTN_CODES", "1") == "0":
        pytest.skip("diagnostic knob set: the per-tensor checks below read the code planes of the default path (uint16 fc1 codes, qkv codes)")
    f16 = os.environ.get("QATVIT_F16", "1") != "0"
    fc2_codes = f16 and os.environ.get("QATVIT_FC2_CODES", "1") != "0"
    fc2w_codes = fc2_codes and os.environ.get("QATVIT_FC1_BITS", "1") != "0" and os.environ.get("QATVIT_I8", "1") != "0" and os.environ.get("QATVIT_FC2W_CODES", "1") != "0"
    fc1_bits = fc2_codes and os.environ.get("QATVIT_FC1_BITS", "1") != "0" and os.environ.get("QATVIT_I8", "1") != "0"
    qkv_2pass = os.environ.get("QATVIT_QKV_2PASS", "1") != "0" and os.environ.get("QATVIT_ATTN_CODES", "1") != "0" and os.environ.get("QATVIT_I8", "1") != "0" and qb - qa <= 255

    def cmp_part(tb, st, i, part, lim=CODE_FLIP_FRAC, tol=TOL, split_fc2=True):
        """Everything part `part` of block i left in the workspace against the oracle."""
        pre = f"model.blocks.{i}"
        if part == 0:
            tb.codes(st, "norm1", eng.tensor("h1q", i, (M, D), torch.bfloat16), tr.codes(f"{pre}.norm1.{A}").reshape(M, D), lim)
            if qkv_2pass:   # the qkv GEMM's second pass wrote codes + STE mask bits (no fp32 tensor): the quantised values and the mask it left
                tb.codes(st, "attn.qkv (codes from the GEMM epilogue)", qkv_ints(eng, i, fqm[f"{pre}.attn.qkv.{A}"]), tr.codes(f"{pre}.attn.qkv.{A}").reshape(M, 3 * D), lim)
                oq = tr.fq[f"{pre}.attn.qkv.{A}"]
                ref_in = (tr.pre(f"{pre}.attn.qkv.{A}") * (1.0 / oq.scale)).round() + oq.zero_point
                tb.codes(st, "attn.qkv STE mask", qkv_mask(eng, i).float(), ((ref_in >= qa) & (ref_in <= qb)).reshape(M, 3 * D).float(), lim)
            else:
                tb.close(st, "attn.qkv pre-FQ", eng.tensor("qkv", i, (M, 3 * D)), tr.pre(f"{pre}.attn.qkv.{A}").reshape(M, 3 * D), tol)
        elif part == 1:
            # the codes the attention forward worked from (saved by itself from an injected fp32 qkv, or by the qkv GEMM's second pass)
            tb.codes(st, "attn.qkv", qkv_ints(eng, i, fqm[f"{pre}.attn.qkv.{A}"]), tr.codes(f"{pre}.attn.qkv.{A}").reshape(M, 3 * D), lim)
            o = eng.tensor("O_hi", i, (M, D), torch.bfloat16).float() + eng.tensor("O_lo", i, (M, D), torch.bfloat16).float()
            tb.close(st, "attention out (bf16 pair, bwd)", o, tr.proj_in[i].reshape(M, D), tol)
            if f16:   # the fp16 (hi, lo) pair the proj forward GEMM actually read (per block: the one-plane proj weight gradient reads it again)
                sc = eng.tensor("scal16", i, (2,))
                o16 = (eng.tensor("O16_hi", i, (M, D), torch.float16).float() + eng.tensor("O16_lo", i, (M, D), torch.float16).float()) * sc[0]
                tb.close(st, "attention out (fp16 pair, fwd)", o16, tr.proj_in[i].reshape(M, D), tol)
            yp = eng.tensor("Yproj", i, (M, D))
            tb.close(st, "attn.proj pre-FQ", yp, tr.pre(f"{pre}.attn.proj.{A}").reshape(M, D), tol)
            tb.codes(st, "attn.proj", _fq_of_codes(yp, fqm[f"{pre}.attn.proj.{A}"], qa, qb), tr.codes(f"{pre}.attn.proj.{A}").reshape(M, D), lim)
            tb.close(st, "x_mid", eng.tensor("x_mid", i, (M, D)), tr.norm2_in[i].reshape(M, D), tol)
        elif part == 2:
            tb.codes(st, "norm2", eng.tensor("h2q", i, (M, D), torch.bfloat16), tr.codes(f"{pre}.norm2.{A}").reshape(M, D), lim)
            if fc1_bits:   # the backward's form of the fc1 codes: the byte plane fc2's forward reads + one STE mask bit per element
                mb = eng.tensor("Y1m", i, (M, Hd // 8), torch.uint8).int()
                inr = ((mb.unsqueeze(-1) >> torch.arange(8, device=mb.device)) & 1).reshape(M, Hd)
                code = eng.tensor("G8", i, (M, Hd), torch.uint8).int() | (inr << 15)
            else:
                code = eng.tensor("Y1", i, (M, Hd), torch.int16).int() & 0xffff                 # (q - qmin) | in_range << 15
            f1 = fqm[f"{pre}.mlp.fc1.{A}"]
            tb.codes(st, "mlp.fc1", (code & 0x7fff).float() + qa - f1.zero_point.float(), tr.codes(f"{pre}.mlp.fc1.{A}").reshape(M, Hd), lim)
            of1 = tr.fq[f"{pre}.mlp.fc1.{A}"]
            ref_in = (tr.pre(f"{pre}.mlp.fc1.{A}") * (1.0 / of1.scale)).round() + of1.zero_point
            ref_mask = ((ref_in >= qa) & (ref_in <= qb)).reshape(M, Hd)
            tb.codes(st, "mlp.fc1 STE mask", (code >> 15).float(), ref_mask.float(), lim)
            if fc2w_codes:   # what the fc2 weight gradient reads: the byte plane + the 256-entry table of bf16 (hi, lo) pairs
                lq = eng.tensor("glutq", i, (256,), torch.int32)
                pq = (lq & 0xffff).to(torch.int16).view(torch.bfloat16).float() + ((lq >> 16) & 0xffff).to(torch.int16).view(torch.bfloat16).float()
                gl = pq[eng.tensor("G8", i, (M, Hd), torch.uint8).long()]
                tb.close(st, "gelu out (codes + bf16 pair table, bwd)", gl, tr.fc2_in[i].reshape(M, Hd), tol)
            else:
                gl = eng.tensor("G_hi", i, (M, Hd), torch.bfloat16).float() + eng.tensor("G_lo", i, (M, Hd), torch.bfloat16).float()
                tb.close(st, "gelu out (bf16 pair, bwd)", gl, tr.fc2_in[i].reshape(M, Hd), tol)
            if f16 and fc2_codes:   # what the fc2 forward GEMM reads: one byte per element + the 256-entry table of fp16 (hi, lo) pairs
                sc = eng.tensor("scal16", i, (2,))
                lut = eng.tensor("glut", i, (256,), torch.int32)
                pair = (lut & 0xffff).to(torch.int16).view(torch.float16).float() + ((lut >> 16) & 0xffff).to(torch.int16).view(torch.float16).float()
                g16 = pair[eng.tensor("G8", i, (M, Hd), torch.uint8).long()] * sc[1]
                tb.close(st, "gelu out (codes + fp16 pair table, fwd)", g16, tr.fc2_in[i].reshape(M, Hd), tol)
            elif f16:
                sc = eng.tensor("scal16", i, (2,))
                g16 = (eng.tensor("G16_hi", 0, (M, Hd), torch.float16).float() + eng.tensor("G16_lo", 0, (M, Hd), torch.float16).float()) * sc[1]
                tb.close(st, "gelu out (fp16 pair, fwd)", g16, tr.fc2_in[i].reshape(M, Hd), tol)
        if part == 3 or (part == 2 and not split_fc2):
            y2 = eng.tensor("Y2", i, (M, D))
            tb.close(st, "mlp.fc2 pre-FQ", y2, tr.pre(f"{pre}.mlp.fc2.{A}").reshape(M, D), tol)
            tb.codes(st, "mlp.fc2", _fq_of_codes(y2, fqm[f"{pre}.mlp.fc2.{A}"], qa, qb), tr.codes(f"{pre}.mlp.fc2.{A}").reshape(M, D), lim)
            tb.close(st, f"x_in[{i + 1}] (block output)", eng.tensor("x_in", i + 1, (M, D)), tr.block_in[i + 1].reshape(M, D), tol)

    # ---- (informational) ONE injection per block, next to the same experiment on the stock torch tree on this GPU: the live floor of
    # that granularity (in-block amplification: a flipped key perturbs a whole head; a flipped norm2 element ~5 % of its row's fc1 codes)
    coarse = Table()
    forced = Table()
    coarse_rows, forced_rows = [], []
    n_cmp = n_within = 0
    perm_seeds = (101, 102, 103)
    CHAIN = ["norm1", "attn.qkv", "attn.proj", "norm2", "mlp.fc1", "mlp.fc2"]

    def force_oracle_qparams(mods, pre, on):
        """The six activation quantizers of a block (mods: name -> module) set to the ORACLE's state after its step, observers off (on=True) / observers back on."""
        for nm in CHAIN:
            m, o = mods[nm], tr.fq[f"{pre}.{nm}.{A}"]
            if on:
                with torch.no_grad():
                    m.activation_post_process.min_val.copy_(o.activation_post_process.min_val)
                    m.activation_post_process.max_val.copy_(o.activation_post_process.max_val)
                    m.scale.copy_(o.scale)
                    m.zero_point.copy_(o.zero_point)
            m.observer_enabled[0] = 0 if on else 1

    def floor_sample(i, device, perm_seed=None, forced=False):
        """One independent fp32 evaluation of block i on the oracle's block input by the STOCK torch tree (fresh observers): per tensor the fraction of
        codes (relative L2 for the block output) that differ from the oracle - the deviation between two correct evaluations at this granularity.
        perm_seed (CPU): the block is evaluated in a permuted feature basis - the same mathematical function, every LayerNorm statistic and every dot
        product summed in another order.  (The host CPU at 1 .. 16 threads is NOT an independent evaluation: oneDNN
        splits the rows, every thread count gave the oracle's bits - measured, round 4.)"""
        pre = f"model.blocks.{i}"
        blk = copy.deepcopy(po0.model.blocks[i]).to(device)
        xin, inv = tr.block_in[i], None
        if perm_seed is not None:
            # the same function in a permuted feature basis: the residual stream's D features by pi (LayerNorm affine parameters, the K columns of qkv / fc1,
            # the output rows of proj / fc2 follow), the K dimensions of proj / fc2 by permutations of their own - every sum of the block (LayerNorm mean /
            # variance, every dot product) runs in another order; outputs are brought back by the inverse permutation
            gp = torch.Generator().manual_seed(perm_seed + i)
            pi = torch.randperm(blk.norm1.weight.numel(), generator=gp)
            sig = torch.randperm(blk.attn.proj.weight.shape[1], generator=gp)
            tau = torch.randperm(blk.mlp.fc2.weight.shape[1], generator=gp)
            inv = torch.argsort(pi)
            with torch.no_grad():
                for ln in (blk.norm1, blk.norm2):
                    ln.weight.copy_(ln.weight[pi].clone()); ln.bias.copy_(ln.bias[pi].clone())
                blk.attn.qkv.weight.copy_(blk.attn.qkv.weight[:, pi].clone())
                blk.mlp.fc1.weight.copy_(blk.mlp.fc1.weight[:, pi].clone())
                blk.attn.proj.weight.copy_(blk.attn.proj.weight[pi][:, sig].clone()); blk.attn.proj.bias.copy_(blk.attn.proj.bias[pi].clone())
                blk.mlp.fc2.weight.copy_(blk.mlp.fc2.weight[pi][:, tau].clone()); blk.mlp.fc2.bias.copy_(blk.mlp.fc2.bias[pi].clone())
            blk.attn.proj.register_forward_pre_hook(lambda mod, inp, sig=sig: (inp[0][..., sig],))
            blk.mlp.fc2.register_forward_pre_hook(lambda mod, inp, tau=tau: (inp[0][..., tau],))
            xin = xin[..., pi]
        if forced:
            force_oracle_qparams({nm: dict(blk.named_modules())[f"{nm}.{A}"] for nm in CHAIN}, pre, True)
        caps = capture_fq_io(blk)
        with torch.no_grad():
            bo = blk(xin.to(device))
        fl = {}
        for nm in ("norm1", "attn.qkv", "attn.proj", "norm2", "mlp.fc1", "mlp.fc2"):
            fm = dict(blk.named_modules())[f"{nm}.{A}"]
            cg = torch.round(caps[f"{nm}.{A}"][1] / fm.scale).cpu()
            if inv is not None and nm in ("norm1", "attn.proj", "norm2", "mlp.fc2"):
                cg = cg[..., inv]
            fl[nm] = (cg.reshape(-1) != tr.codes(f"{pre}.{nm}.{A}").reshape(-1)).float().mean().item()
            so = tr.fq[f"{pre}.{nm}.{A}"].scale
            fl["ds " + nm] = ((fm.scale.cpu() - so).abs() / so).item()      # relative difference of this quantizer's scale from the oracle's
        bo = bo.cpu() if inv is None else bo.cpu()[..., inv]
        fl[f"x_in[{i + 1}] (block output)"] = rel_l2(bo.numpy(), tr.block_in[i + 1].numpy())
        return fl

    for i in blocks:
        st, pre = f"block{i}", f"model.blocks.{i}"
        reset_act_observers(range(2 + 6 * i, 8 + 6 * i))
        eng.tensor("x_in", i, (M, D)).copy_(tr.block_in[i].reshape(M, D).cuda())
        eng.forward_stages(None, i + 1, i + 1, inject=True)
        n0 = len(coarse.rows)
        for part in range(3):
            cmp_part(coarse, st, i, part, lim=1.0, tol=1.0, split_fc2=False)
        # the floor as a DISTRIBUTION: stock torch on this GPU and on the host CPU with permuted reduction orders - independent fp32 evaluations of the
        # same block on the same input
        samples = [("gpu", floor_sample(i, "cuda"))] + [(f"cpu-perm{n}", floor_sample(i, "cpu", n)) for n in perm_seeds]
        for k in range(n0, len(coarse.rows)):
            r = coarse.rows[k]
            if r[1] in samples[0][1]:
                vals = [fl[r[1]] for _, fl in samples]
                coarse.rows[k] = r + (max(vals),)
                n_cmp += 1
                n_within += r[4] <= 2 * max(vals) + 1e-4
                ds_ours, ds_vals, allow = 0.0, [0.0] * len(samples), 0.0
                if "ds " + r[1] in samples[0][1]:
                    so = tr.fq[f"{pre}.{r[1]}.{A}"].scale
                    ds_ours = ((fqm[f"{pre}.{r[1]}.{A}"].scale.cpu() - so).abs() / so).item()
                    ds_vals = [fl["ds " + r[1]] for _, fl in samples]
                    # what the quantizer-scale differences of this block explain: a scale (or an input) that is off by a relative d moves every
                    # pre-rounding value x / s by |q - zp| d, i.e. changes about mean|q - zp| d of the codes; d = this quantizer's own scale difference
                    # plus those of the quantizers upstream of it in the block (their de-quantised outputs carry the factor on)
                    chain = CHAIN[:CHAIN.index(r[1]) + 1]
                    dsum = sum(((fqm[f"{pre}.{q}.{A}"].scale.cpu() - tr.fq[f"{pre}.{q}.{A}"].scale).abs() / tr.fq[f"{pre}.{q}.{A}"].scale).item() for q in chain)
                    allow = 2.0 * tr.codes(f"{pre}.{r[1]}.{A}").abs().mean().item() * dsum
                coarse_rows.append((st, r[1], r[4], vals, ds_ours, ds_vals, allow))
        # ---- the ASSERTED coarse run: the same block, one injection, with every quantizer of the block on the ORACLE's scale / zero point (observers off) -
        # in the native tree and in the stock samples alike.  What remains is what chaining the block's stages does to VALUES (codes that sit on a rounding
        # tie, amplified through attention and the MLP); the scale mechanism above is taken out on both sides.
        mods = {nm: fqm[f"{pre}.{nm}.{A}"] for nm in CHAIN}
        force_oracle_qparams(mods, pre, True)
        eng.tensor("x_in", i, (M, D)).copy_(tr.block_in[i].reshape(M, D).cuda())
        eng.forward_stages(None, i + 1, i + 1, inject=True)
        n1 = len(forced.rows)
        for part in range(3):
            cmp_part(forced, st, i, part, lim=1.0, tol=1.0, split_fc2=False)
        force_oracle_qparams(mods, pre, False)
        fsamples = [floor_sample(i, "cuda", forced=True)] + [floor_sample(i, "cpu", n, forced=True) for n in perm_seeds]
        for k in range(n1, len(forced.rows)):
            r = forced.rows[k]
            if r[1] in fsamples[0]:
                vals = [fl[r[1]] for fl in fsamples]
                forced.rows[k] = r + (max(vals),)
                forced_rows.append((st, r[1], r[4], vals))
    # The free-running table (every quantizer observes for itself) is heavy-tailed through ONE mechanism, visible in its scale columns: a flipped code in
    # the row that holds a tensor's extreme element moves that quantizer's SCALE by 1e-5 .. 2e-3, which re-rounds mean|q| times that fraction of ALL its
    # codes and of everything downstream (it happens to the oracle itself: C3 block 5, where the three permuted CPU evaluations and the native one agree
    # on fc1's scale and differ from the oracle's by the same 2.3e-4).  Every quantizer's scale is asserted to 2e-5 of the oracle's in the teacher-forced
    # run, where its input is the oracle's.  That table is reported with the null distribution (each stock sample against the other evaluations of its
    # block) and asserted only in the median (a systematic excess everywhere); the per-row assertion is on the forced-scale run: every native row within
    # twice the largest of ITS OWN block's four independent stock evaluations (+ 1e-4).
    plain2 = [(st, name) for st, name, v, vals, _, _, _ in coarse_rows if v > 2 * max(vals) + 1e-4]
    sig = [(v, vals) for _, _, v, vals, _, _, _ in coarse_rows if max(vals + [v]) > 1e-4]          # rows with a measurable deviation
    ratios = sorted(v / (max(vals) + 1e-12) for v, vals in sig)
    null = sorted(vals[k] / (max(vals[:k] + vals[k + 1:] + [v]) + 1e-12) for v, vals in sig for k in range(len(vals)))
    med = ratios[len(ratios) // 2] if ratios else 0.0
    forced_bad = [(st, name, v, max(vals)) for st, name, v, vals in forced_rows if v > 2 * max(vals) + 1e-4]
    fr = sorted(v / (max(vals) + 1e-12) for _, _, v, vals in forced_rows if max(vals + [v]) > 1e-4)
    forced.write(os.path.join(ROOT, "gpurun_out", f"round4_block_level_forced_scales_{golden_tag}.txt"),
                 f"# ONE injection per block with every quantizer of the block on the oracle's scale / zero point (observers off), native and stock alike: native block on "
                 f"the oracle's block input vs the oracle; last column = the largest deviation of {1 + len(perm_seeds)} independent stock-torch evaluations (this GPU; host CPU "
                 f"with {len(perm_seeds)} permutations of every Linear layer's reduction order); {arch} batch {B}, {backend}; {len(forced_rows) - len(forced_bad)} of "
                 f"{len(forced_rows)} rows within 2x that + 1e-4; ratio native / largest stock sample: median {(fr[len(fr) // 2] if fr else 0):.2f}, max {(fr[-1] if fr else 0):.2f}")
    coarse.write(os.path.join(ROOT, "gpurun_out", f"round4_block_level_vs_floor_{golden_tag}.txt"),
                 f"# ONE injection per block (coarse): native block on the oracle's block input vs the oracle; last column = the largest deviation of {1 + len(perm_seeds)} "
                 f"independent stock-torch evaluations of the same block on the same input (this GPU; host CPU with {len(perm_seeds)} permutations of every Linear layer's reduction order) from the oracle; "
                 f"{arch} batch {B}, {backend}; native within 2x that + 1e-4 in {n_within} of {n_cmp} rows (beyond it: {plain2}); ratio native / largest stock sample: median {med:.2f}, "
                 f"max {(ratios[-1] if ratios else 0):.2f}; the same ratio of each stock sample against the other evaluations of its block (leave one out): "
                 f"median {(null[len(null) // 2] if null else 0):.2f}, above 1 in {sum(r > 1 for r in null)} of {len(null)}, above 2 in {sum(r > 2 for r in null)}, max {(null[-1] if null else 0):.2f}")
    with open(os.path.join(ROOT, "gpurun_out", f"round4_block_level_floor_samples_{golden_tag}.txt"), "w") as f:
        f.write(f"# per block and tensor: native deviation from the oracle, then every stock-torch sample's deviation (gpu; cpu with permuted reduction order x {len(perm_seeds)})\n")
        f.write("# then (| ...): the relative difference of that quantizer's SCALE from the oracle's, native first, then the samples; last: the deviation the "
                "block's scale differences explain (2 mean|q - zp| x their sum up to this quantizer)\n")
        for st, name, v, vals, dso, dsv, al in coarse_rows:
            f.write(f"{st:<10}{name:<28}{v:12.3e}   " + " ".join(f"{x:10.3e}" for x in vals) + f"   | {dso:9.2e}  " + " ".join(f"{x:9.2e}" for x in dsv) + f"   | {al:9.2e}"
                    + ("   <-- beyond 2x max + 1e-4" if v > 2 * max(vals) + 1e-4 else "") + "\n")

    # ---- the asserted run: injection at every fake-quantizer input that follows an amplifier (four parts per block)
    for i in blocks:
        st = f"block{i}"
        reset_act_observers([2 + 6 * i])
        eng.tensor("x_in", i, (M, D)).copy_(tr.block_in[i].reshape(M, D).cuda())
        eng.forward_part(i, 0, inject=True)
        cmp_part(tab, st, i, 0)
        reset_act_observers([3 + 6 * i, 4 + 6 * i])
        eng.tensor("qkv", i, (M, 3 * D)).copy_(tr.pre(f"model.blocks.{i}.attn.qkv.{A}").reshape(M, 3 * D).cuda())
        eng.forward_part(i, 1, inject=True)
        cmp_part(tab, st, i, 1)
        reset_act_observers([5 + 6 * i, 6 + 6 * i])
        eng.tensor("x_mid", i, (M, D)).copy_(tr.norm2_in[i].reshape(M, D).cuda())
        eng.forward_part(i, 2, inject=True)
        cmp_part(tab, st, i, 2)
        # part 3 on the oracle's GELU output: written in every form the fc2 GEMMs read (bf16 pair; fp16 pair + its power-of-two scale)
        g_ref = tr.fc2_in[i].reshape(M, Hd).cuda()
        gh = g_ref.to(torch.bfloat16)
        eng.tensor("G_hi", i, (M, Hd), torch.bfloat16).copy_(gh)
        eng.tensor("G_lo", i, (M, Hd), torch.bfloat16).copy_((g_ref - gh.float()).to(torch.bfloat16))
        if f16:
            kexp = 13 - int(np.floor(np.log2(g_ref.abs().max().item())))
            gs = g_ref * (2.0 ** kexp)
            g16h = gs.to(torch.float16)
            if fc2_codes:   # the oracle's fc1 codes + the table of the oracle's GELU values per code
                of1 = tr.fq[f"model.blocks.{i}.mlp.fc1.{A}"]
                idx = (tr.codes(f"model.blocks.{i}.mlp.fc1.{A}").reshape(M, Hd) + of1.zero_point.float() - qa).round().long().cuda()
                assert int(idx.min()) >= 0 and int(idx.max()) <= 255
                hi_t = torch.zeros(256, dtype=torch.float16, device="cuda")
                lo_t = torch.zeros(256, dtype=torch.float16, device="cuda")
                hi_t[idx.reshape(-1)] = g16h.reshape(-1)
                lo_t[idx.reshape(-1)] = (gs - g16h.float()).to(torch.float16).reshape(-1)
                packed = (hi_t.view(torch.int16).int() & 0xffff) | (lo_t.view(torch.int16).int() << 16)
                eng.tensor("glut", i, (256,), torch.int32).copy_(packed)
                if fc2w_codes:   # ... and the bf16-pair table the fc2 weight gradient expands the same codes through
                    qh_t = torch.zeros(256, dtype=torch.bfloat16, device="cuda")
                    ql_t = torch.zeros(256, dtype=torch.bfloat16, device="cuda")
                    qh_t[idx.reshape(-1)] = gh.reshape(-1)
                    ql_t[idx.reshape(-1)] = (g_ref - gh.float()).to(torch.bfloat16).reshape(-1)
                    eng.tensor("glutq", i, (256,), torch.int32).copy_((qh_t.view(torch.int16).int() & 0xffff) | (ql_t.view(torch.int16).int() << 16))
                eng.tensor("G8", i, (M, Hd), torch.uint8).copy_(idx.to(torch.uint8))
            else:
                eng.tensor("G16_hi", 0, (M, Hd), torch.float16).copy_(g16h)
                eng.tensor("G16_lo", 0, (M, Hd), torch.float16).copy_((gs - g16h.float()).to(torch.float16))
            eng.tensor("scal16", i, (2,))[1] = 2.0 ** -kexp
        reset_act_observers([7 + 6 * i])
        eng.forward_part(i, 3, inject=True)
        cmp_part(tab, st, i, 3)
    reset_act_observers([n_act - 2, n_act - 1])
    eng.tensor("x_in", depth, (M, D)).copy_(tr.block_in[depth].reshape(M, D).cuda())
    logits = eng.forward_stages(None, depth + 1, depth + 1, inject=True)
    tab.codes("head", "norm (cls rows)", eng.tensor("hq", 0, (B, D)), tr.codes(f"model.norm.{A}")[:, 0])
    lp = eng.tensor("logits_pre", 0, (B, 10))
    tab.close("head", "head pre-FQ", lp, tr.pre(f"model.head.{A}"))
    tab.codes("head", "head (logits)", _fq_of_codes(lp, fqm[f"model.head.{A}"], qa, qb), tr.codes(f"model.head.{A}"))
    tab.close("head", "logits", logits, tr.logits, tol=5e-3)            # 80 values on a 256-level grid: one flipped code is 4e-3 here
    # every activation quantizer's state after its one observation: the oracle's, to fp32 rounding of the observed min / max
    ran = tuple(f"model.blocks.{i}." for i in blocks)
    for n, f in fqm.items():
        if "weight_fake_quant" in n:
            assert torch.allclose(f.scale.cpu(), tr.fq[n].scale, rtol=1e-6), n
            assert torch.equal(f.zero_point.cpu(), tr.fq[n].zero_point), n
        elif ".blocks." in n and not n.startswith(ran):
            continue   # (a block this run did not execute teacher-forced: its observers hold the ordinary first step's state)
        else:
            if not (torch.allclose(f.scale.cpu(), tr.fq[n].scale, rtol=2e-5) and (f.zero_point.cpu() - tr.fq[n].zero_point).abs().max().item() <= 1):
                tab.bad.append(("state", n, "scale/zp", f.scale.item(), tr.fq[n].scale.item()))

    # ================================================================= backward, stage by stage (forward state = the teacher-forced one)
    names = [n for n, _ in p.named_parameters()]
    pidx = {id(q): k for k, q in enumerate(eng.params)}
    name_of = {pidx[id(q)]: n for n, q in p.named_parameters()}

    def check_grads(stage, views, which):
        for k in which:
            tab.close(stage, "d " + name_of[k].replace("model.", ""), views[k], tr.grads[name_of[k]])

    n_par = len(eng.params)
    v = eng.backward_stages(tr.g_logits[0].cuda(), 0, 0)
    check_grads("bwd head", v, range(n_par - 4, n_par))
    dxA = eng.tensor("dxA", 0, (M, D))
    tab.close("bwd head", f"d x_in[{depth}]", dxA, tr.g_block_in[depth].reshape(M, D))
    for s in range(1, depth + 1):
        i = depth - s
        if i not in blocks:
            continue
        dxA.copy_(tr.g_block_in[i + 1].reshape(M, D).cuda())
        v = eng.backward_stages(None, s, s, inject=True)
        check_grads(f"bwd block{i}", v, range(4 + 12 * i, 4 + 12 * i + 12))
        tab.close(f"bwd block{i}", f"d x_in[{i}]", dxA, tr.g_block_in[i].reshape(M, D))
    dxA.copy_(tr.g_block_in[0].reshape(M, D).cuda())
    v = eng.backward_stages(None, depth + 1, depth + 1, inject=True)
    check_grads("bwd embed", v, range(0, 4))
    assert len(names) == n_par
    # ---- the same block stages in the ONE-PLANE form (include/qatvit.h QATVIT_BWD_DY16: every dY an fp16 plane, one MFMA pass per GEMM): per stage a
    # calibrating pair-form run on the same injected gradient (records the maxima the scales come from), then the one-plane run, same bars
    tab16 = Table()
    if eng.dy16:
        for s in range(1, depth + 1):
            i = depth - s
            if i not in blocks:
                continue
            for nm in ("h1q", "h2q"):   # the X operand of the qkv / fc1 weight gradients as fp16 integers (what a QATVIT_FWD_X16 forward writes)
                hq = eng.tensor(nm, i, (M, D), torch.bfloat16)
                hq.view(torch.float16).copy_(hq.float().to(torch.float16))
            dxA.copy_(tr.g_block_in[i + 1].reshape(M, D).cuda())
            eng.backward_stages(None, s, s, inject=True, mode=E.BWD_CALIBRATE)   # (its weight gradients read the fp16 planes as bf16: discarded)
            dxA.copy_(tr.g_block_in[i + 1].reshape(M, D).cuda())
            v = eng.backward_stages(None, s, s, inject=True, mode=E.BWD_DY16)
            assert not eng.dy16_overflowed()
            for k in range(4 + 12 * i, 4 + 12 * i + 12):
                tab16.close(f"bwd block{i}", "d " + name_of[k].replace("model.", ""), v[k], tr.grads[name_of[k]])
            tab16.close(f"bwd block{i}", f"d x_in[{i}]", dxA, tr.g_block_in[i].reshape(M, D))
        with open(os.path.join(ROOT, "gpurun_out", f"round4_stage_table_dy16_{golden_tag}.txt"), "w") as f:
            f.write(f"# one-plane backward (QATVIT_BWD_DY16), teacher-forced block stages, {arch} batch {B}, {backend}: native stage on the oracle's input vs the oracle\n")
            for st, n, k, ne, val, _ in tab16.rows:
                f.write(f"{st:<14}{n:<34}{k:<8}{ne:>12}{val:>14.3e}\n")
            f.write(f"\nworst rel L2 {max(r[4] for r in tab16.rows):.2e} (pair form, same stages: {max(r[4] for r in tab.rows if r[0].startswith('bwd block')):.2e})\n")
    path = os.path.join(ROOT, "gpurun_out", f"round4_stage_flip_table_{golden_tag}.txt")
    tab.write(path, f"# teacher-forced stage parity, {arch} batch {B}, {backend} qconfig, {'KD' if teacher else 'CE only'}; native stage on the oracle's input vs the oracle "
                    f"(torch {torch.__version__} CPU eager QAT); produced by tests/test_gpu_stage_parity.py")
    tab.check()
    tab16.check()
    assert not forced_bad, ("one injection per block, oracle's scales: rows beyond 2x the largest of that block's independent stock deviations", forced_bad[:8])
    assert med <= 1.25, ("one injection per block: the native deviation is systematically above the stock evaluations'", med)
    return tab


def test_stage_parity_c1_qnnpack(native_lib):
    """BASELINE config C1 shapes: ViT-S + QATWrapper, batch 8, qnnpack (per-tensor weights, [0,255] activations), CE only."""
    _run("qnnpack", 21, False, "c1_qnnpack")


def test_stage_parity_c3_x86(native_lib):
    """BASELINE config C3 semantics at batch 8: x86 qconfig (per-channel weights, [0,127] activations), KD loss on."""
    _run("x86", 22, True, "c3_x86")


@pytest.mark.timeout(1200)
def test_stage_parity_c5_vitb_x86(native_lib):
    """BASELINE config C5's student (vit_base_patch16_224, /root/reference/src/models/model_registry.py:152-175) at batch 4, x86 qconfig: the D = 768
    kernel variants - K = 768 int8 GEMMs on the general tile (statistics / code / storing passes), N = 768 / 2304 / 3072 tiles, per-channel scales at
    3072 channels, the unfused LayerNorm backward (k_ln_bwd_fq), 12 heads - through the same harness as ViT-S."""
    _run("x86", 23, False, "c5_vitb_x86", arch="vit_base_patch16_224", B=4)


@pytest.mark.timeout(1500)
def test_stage_parity_vits_b64_one_block(native_lib):
    """Batch 64 (M = 12,608 rows: 61 row tiles, ragged last tile) for blocks 0 and 6 + embedding + head: the row-sum length effects the batch-8 harness
    cannot see - bias gradients, LayerNorm dgamma / dbeta, the weight gradients' token splits (qat_trainer.py:337-359 at a realistic batch)."""
    _run("qnnpack", 24, False, "c2_b64_blocks_0_6", B=64, blocks=[0, 6])
