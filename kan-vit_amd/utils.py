"""Helpers of the drop-in surface (counterparts of the reference's utils.py).

Only FlashAttentionFunction is on the accelerated path (utils.py:134-295 -> the HIP attention
kernels).  The metric / logging helpers are host-side reporting kept so that train.py behaves like
the reference's; they are not part of the hot path."""
import datetime
import logging
import os

import torch
from torch.autograd.function import Function

from kanvit import ops

EPSILON = 1e-10


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else d


class FlashAttentionFunction(Function):
    """Same call signature as the reference (utils.py:137): apply(q, k, v, mask, causal,
    q_bucket_size, k_bucket_size) with q/k/v of shape (b, h, n, d).  The bucket sizes only chose
    the tiling of the python implementation; on MI355X a whole head fits one workgroup, so they
    do not change anything.  ``mask`` is not supported by the kernel and must be None."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, q, k, v, mask, causal, q_bucket_size, k_bucket_size):
        if mask is not None:
            raise NotImplementedError("key-padding masks are not implemented in the HIP attention kernel")
        if tuple(k.shape) != tuple(q.shape) or tuple(v.shape) != tuple(q.shape):
            # the reference tiles over independent q / k lengths (utils.py:150-160); the HIP kernel holds one head of ONE
            # length in a work-group, so cross-attention (FlashAttention(context=...)) is rejected instead of mis-read
            raise NotImplementedError(f"q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)}: the HIP attention kernel "
                                      "needs q, k and v of one shape (self-attention)")
        q, k, v = (t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v))
        scale = q.shape[-1] ** -0.5
        o = torch.empty(q.shape, device=q.device, dtype=q.dtype)
        lse = ops._attn_fwd(q, k, v, o, causal, scale)
        ctx.args = (causal, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, do):
        causal, scale = ctx.args
        q, k, v, o, lse = ctx.saved_tensors
        do = do.float().contiguous()
        dq, dk, dv = (torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype) for t in (q, k, v))
        ops._attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal, scale)
        return dq, dk, dv, None, None, None, None


# ---------------------------------------------------------------------------------------------
# reporting (off the hot path)
# ---------------------------------------------------------------------------------------------
def calculate_metrics(y_true, y_pred, y_pred_proba, num_classes=100):
    """accuracy, balanced accuracy, weighted F1 and one-vs-rest ROC-AUC (utils.py:13-47)."""
    from sklearn.metrics import accuracy_score, balanced_accuracy_score, f1_score, roc_auc_score
    acc = accuracy_score(y_true, y_pred)
    bal = balanced_accuracy_score(y_true, y_pred)
    f1 = f1_score(y_true, y_pred, average='weighted')
    onehot = torch.nn.functional.one_hot(torch.as_tensor(y_true), num_classes=num_classes).numpy()
    try:
        auc = roc_auc_score(onehot, y_pred_proba, average='weighted', multi_class='ovr')
    except ValueError:          # a class absent from y_true (short synthetic runs)
        auc = float('nan')
    return acc, bal, f1, auc


def save_metrics(filename, epoch, phase, loss, accuracy, balanced_accuracy, f1, roc_auc, flag):
    """Append one block in the reference's text format (utils.py:79-94)."""
    os.makedirs(os.path.dirname(filename) or '.', exist_ok=True)
    head = f"Epoch: {epoch}, Phase: {phase}\n" if flag == 0 else f"Phase: {phase}\n"
    with open(filename, 'a') as f:
        f.write(head)
        f.write(f"  Loss: {loss:.4f}\n  Accuracy: {accuracy:.4f}\n  Balanced Accuracy: {balanced_accuracy:.4f}\n"
                f"  F1 Score: {f1:.4f}\n  ROC AUC: {roc_auc:.4f}\n\n")


def setup_logging(log_dir='logs'):
    """File + console logging; returns the metrics file name (utils.py:298-328)."""
    os.makedirs(log_dir, exist_ok=True)
    stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
    logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(levelname)s - %(message)s',
                        handlers=[logging.FileHandler(os.path.join(log_dir, f'training_{stamp}.log')),
                                  logging.StreamHandler()])
    return os.path.join(log_dir, f'mnist_metrics_{stamp}.txt')
