"""Symmetric contrastive loss of /root/reference/CLIP/train.py:162-173 and
/root/reference/CLIP/train_caption.py:125-136,

    label = arange(N); loss = (CE(logits_per_image, label) + CE(logits_per_text, label)) / 2
    accuracy = mean(argmax(logits_per_image, 1) == label)

fused with CLIP.forward's normalise + similarity matmul, and made data-parallel: the reference is
single-GPU; here each rank holds N_loc rows, the L2-normalised image and text
features are exchanged with ONE RCCL all-gather of a packed [N_loc, 2E] fp32 buffer over xGMI,
every rank forms its row blocks  L_i = s I_loc T_all^T  and  L_t = s T_loc I_all^T  ([N_loc, N]),
and the cross-rank part of the feature gradient comes back through ONE reduce-scatter.  The
result equals the single-GPU loss at N = world * N_loc (tests/test_dp_gloo.py).

All arithmetic goes through cclip_hip.ops (HIP kernels); torch.distributed only moves bytes.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from cclip_hip import ops


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _collectives(group) -> bool:
    from .parallel import collectives_active
    return collectives_active(group)


class _Contrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fi, ft, logit_scale, group):
        rank, world = _world(group)
        dp = _collectives(group)              # world > 1, or a forced one-rank RCCL pass (clip.parallel.collectives_active)
        dev = fi.device
        fi, ft = fi.contiguous().float(), ft.contiguous().float()
        ls = logit_scale.detach().float().reshape(1).contiguous()
        nloc, E = fi.shape
        N = nloc * world
        need_grad = any(ctx.needs_input_grad[:3])

        packed = torch.empty(nloc, 2 * E, device=dev, dtype=torch.float32)      # [ In | Tn ]
        i_n, t_n = packed[:, :E], packed[:, E:]
        inv_i = torch.empty(nloc, device=dev, dtype=torch.float32)
        inv_t = torch.empty(nloc, device=dev, dtype=torch.float32)
        ops.l2norm_fwd(fi, i_n, inv_i)
        ops.l2norm_fwd(ft, t_n, inv_t)
        if dp:
            gathered = torch.empty(N, 2 * E, device=dev, dtype=torch.float32)
            dist.all_gather_into_tensor(gathered, packed, group=group)
        else:
            gathered = packed
        i_all, t_all = gathered[:, :E], gathered[:, E:]

        L_i = torch.empty(nloc, N, device=dev, dtype=torch.float32)
        L_t = torch.empty(nloc, N, device=dev, dtype=torch.float32)
        ops.gemm_f32(i_n, t_all, L_i, alpha_log_dev=ls)
        ops.gemm_f32(t_n, i_all, L_t, alpha_log_dev=ls)
        labels = (torch.arange(nloc, device=dev) + rank * nloc).to(torch.int32)
        loss_rows = torch.empty(2, nloc, device=dev, dtype=torch.float32)
        rowdot = torch.empty(2, nloc, device=dev, dtype=torch.float32) if need_grad else None
        pred = torch.empty(nloc, device=dev, dtype=torch.int32)
        gs = 1.0 / (2.0 * N)
        # gradients of the GLOBAL mean loss overwrite the logits in place (nothing else needs them)
        ops.xent_rows(L_i, labels, loss_row=loss_rows[0], pred=pred, dlogits=L_i if need_grad else None, grad_scale=gs,
                      rowdot=rowdot[0] if need_grad else None)
        ops.xent_rows(L_t, labels, loss_row=loss_rows[1], dlogits=L_t if need_grad else None, grad_scale=gs,
                      rowdot=rowdot[1] if need_grad else None)
        out = torch.empty(2, device=dev, dtype=torch.float32)                    # [loss, #correct]
        ops.reduce_dot(loss_rows.view(-1), None, out[0:1], alpha=gs)
        hit = (pred == labels).to(torch.float32)                                 # integer compare (bookkeeping)
        ops.reduce_dot(hit, None, out[1:2])
        if dp:
            dist.all_reduce(out, group=group)
        if need_grad:
            # d/d(normalised features): local rows + the other ranks' rows that used our features
            cross = torch.empty(N, 2 * E, device=dev, dtype=torch.float32)
            ops.gemm_f32(L_t.t(), t_n.t(), cross[:, :E], alpha_log_dev=ls)       # -> d I_all = s dL_t^T T_loc
            ops.gemm_f32(L_i.t(), i_n.t(), cross[:, E:], alpha_log_dev=ls)       # -> d T_all = s dL_i^T I_loc
            if dp:
                d = torch.empty(nloc, 2 * E, device=dev, dtype=torch.float32)
                dist.reduce_scatter_tensor(d, cross, group=group)
            else:
                d = cross
            ops.gemm_f32(L_i, t_all.t(), d[:, :E], alpha_log_dev=ls, beta=1.0)   # += s dL_i T_all
            ops.gemm_f32(L_t, i_all.t(), d[:, E:], alpha_log_dev=ls, beta=1.0)   # += s dL_t I_all
            ctx.saved = (d, packed, inv_i, inv_t, rowdot)
        ctx.mark_non_differentiable(out)
        loss = out[0].clone()
        ctx.stats = out
        return loss, out

    @staticmethod
    def backward(ctx, dloss, _dout):
        d, packed, inv_i, inv_t, rowdot = ctx.saved
        E = packed.shape[1] // 2
        g = dloss.detach().float().reshape(1).contiguous()
        dfi = torch.empty(packed.shape[0], E, device=packed.device, dtype=torch.float32)
        dft = torch.empty_like(dfi)
        ops.l2norm_bwd(d[:, :E], packed[:, :E], inv_i, dfi, mul_dev=g)
        ops.l2norm_bwd(d[:, E:], packed[:, E:], inv_t, dft, mul_dev=g)
        dscale = torch.empty(1, device=packed.device, dtype=torch.float32)
        ops.reduce_dot(rowdot.view(-1), None, dscale, mul_dev=g)
        ctx.saved = None
        return dfi, dft, dscale.reshape(()), None


def contrastive_loss(image_features: torch.Tensor, text_features: torch.Tensor, logit_scale: torch.Tensor,
                     group: Optional["dist.ProcessGroup"] = None):
    """Returns (loss, stats) where stats = tensor([global mean loss, global #correct image->text]).
    `loss` is the GLOBAL mean loss; its gradient w.r.t. this rank's features is exact, so parameter
    gradients must be SUMMED over ranks (clip.parallel.allreduce_gradients does that)."""
    return _Contrastive.apply(image_features, text_features, logit_scale, group)


class ContrastiveLoss(torch.nn.Module):
    def __init__(self, group=None):
        super().__init__()
        self.group = group

    def forward(self, image_features, text_features, logit_scale):
        return contrastive_loss(image_features, text_features, logit_scale, self.group)
