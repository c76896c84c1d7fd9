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
    q_bucket_size, k_bucket_size) with q (b, h, q_len, d) and k, v (b, h, k_len, d).  The bucket sizes only chose
    the tiling of the python implementation; on MI355X a whole head fits one workgroup, so they
    do not change anything.  Self-attention without a mask (the ViT path) runs the kernels of csrc/attention.hip; a mask
    ((b, k_len) key padding or anything broadcastable to (b, h, q_len, k_len), True = attend; utils.py:156-164) or
    q_len != k_len (cross-attention) runs the general kernels of csrc/attention_x.hip (exact fp32).  `causal` together with
    k_len > q_len raises: the reference shifts the diagonal the wrong way there (utils.py:169: the first k_len - q_len
    queries see no key, and what it returns for them depends on the bucket sizes)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, q, k, v, mask, causal, q_bucket_size, k_bucket_size):
        if q.dim() != 4 or k.dim() != 4 or tuple(v.shape) != tuple(k.shape) or k.shape[:2] != q.shape[:2] or k.shape[3] != q.shape[3]:
            raise ValueError(f"q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)}: need q (b, h, q_len, d) and k, v (b, h, k_len, d)")
        if causal and k.shape[2] > q.shape[2]:
            raise NotImplementedError(f"causal attention with k_len {k.shape[2]} > q_len {q.shape[2]}: the reference shifts the diagonal so that "
                                      "the first k_len - q_len queries see no key (utils.py:169,183) and answers bucket-size dependently; refused")
        q, k, v = (t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v))
        scale = q.shape[-1] ** -0.5
        o = torch.empty(q.shape, device=q.device, dtype=q.dtype)
        general = mask is not None or k.shape[2] != q.shape[2]
        if mask is not None:
            if mask.dim() == 2:
                mask = mask[:, None, None, :]                         # 'b n -> b 1 1 n' (utils.py:156-157)
            mask = mask.to(device=q.device, dtype=torch.bool).expand(q.shape[0], q.shape[1], q.shape[2], k.shape[2])
        if general:
            lse = ops._attn_x_fwd(q, k, v, o, mask, causal, scale)
        else:
            lse = ops._attn_fwd(q, k, v, o, causal, scale)
        ctx.args = (causal, scale, general)
        ctx.save_for_backward(q, k, v, o, lse, mask)
        return o

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, do):
        causal, scale, general = ctx.args
        q, k, v, o, lse, mask = ctx.saved_tensors
        do = do.float().contiguous()
        dq, dk, dv = (torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype) for t in (q, k, v))
        if general:
            ops._attn_x_bwd(q, k, v, o, lse, do, dq, dk, dv, mask, causal, scale)
        else:
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
