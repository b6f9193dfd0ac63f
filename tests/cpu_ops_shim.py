"""TEST-ONLY stand-in for cclip_hip.ops on CPU tensors, so that the multi-process (gloo, world_size 2)
tests can exercise the data-parallel choreography of clip/loss.py and clip/parallel.py - all-gather layout,
label offsets, reduce-scatter of the cross terms, gradient buckets - in this GPU-less container.
Each function restates the contract of the HIP launcher of the same name (include/cclip_hip.h) with plain torch.
It is never importable from the product package."""
import torch


def l2norm_fwd(x, y, inv_norm):
    inv = 1.0 / x.norm(dim=1)
    y.copy_(x * inv[:, None])
    if inv_norm is not None:
        inv_norm.copy_(inv)


def l2norm_bwd(dy, y, inv_norm, dx, mul_dev=None):
    s = (y * dy).sum(1, keepdim=True)
    m = 1.0 if mul_dev is None else mul_dev.item()
    dx.copy_((dy - y * s) * inv_norm[:, None] * m)


def gemm_f32(A, B, C, *, alpha=1.0, beta=0.0, alpha_log_dev=None):
    a = alpha * (alpha_log_dev.exp().item() if alpha_log_dev is not None else 1.0)
    r = a * (A @ B.t())
    C.copy_(r + beta * C if beta != 0.0 else r)


def xent_rows(logits, labels_i32, *, loss_row=None, pred=None, dlogits=None, grad_scale=1.0, ignore_index=-100, rowdot=None):
    lab = labels_i32.long()
    lse = torch.logsumexp(logits, dim=1)
    ign = lab == ignore_index
    safe = lab.clamp(0, logits.shape[1] - 1)
    if loss_row is not None:
        loss_row.copy_(torch.where(ign, torch.zeros_like(lse), lse - logits.gather(1, safe[:, None])[:, 0]))
    if pred is not None:
        pred.copy_(logits.argmax(1).to(torch.int32))
    if dlogits is not None:
        d = torch.softmax(logits, 1)
        d[torch.arange(len(lab)), safe] -= 1.0
        d = d * grad_scale
        d[ign] = 0
        if rowdot is not None:
            rowdot.copy_((d * logits).sum(1))
        dlogits.copy_(d.to(dlogits.dtype))


def reduce_dot(a, b, out, *, alpha=1.0, mul_dev=None, accumulate=False):
    v = (a * b).sum() if b is not None else a.sum()
    v = v * alpha * (1.0 if mul_dev is None else mul_dev.item())
    out.copy_((out + v) if accumulate else v.reshape(out.shape))


def adamw_step(param, grad, exp_avg, exp_avg_sq, *, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0, step=1,
               correct_bias=True, grad_scale=1.0, mode=0, bf16_shadow=None):
    """transformers.AdamW (mode 0) on flat views, in place - the contract of cclip_adamw_step (include/cclip_hip.h)."""
    assert mode == 0
    g = grad * grad_scale
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    step_size = lr
    if correct_bias:
        step_size = lr * (1 - beta2 ** step) ** 0.5 / (1 - beta1 ** step)
    param.addcdiv_(exp_avg, exp_avg_sq.sqrt().add_(eps), value=-step_size)
    if weight_decay > 0:
        param.add_(param, alpha=-lr * weight_decay)
    if bf16_shadow is not None:
        bf16_shadow.copy_(param)
