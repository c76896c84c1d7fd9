"""Run one kanvit op in a loop (for rocprofv3 --pmc passes on a single kernel).
  python tools/bench_op.py fwd|bwd [amp] [iters] [type] [vits]   one MSA block (ViT-B geometry, ChebyKAN; or another type / ViT-S)
  python tools/bench_op.py layer <type> [amp] [iters]            the fused patch embedding of a ViT-B model of <type>, forward + backward"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
amp = 'amp' in sys.argv[2:]
nums = [int(a) for a in sys.argv[2:] if a.isdigit()]
iters = nums[0] if nums else 5
torch.manual_seed(0)
if what == 'layer':
    from model import VisionTransformer
    t = sys.argv[2]
    m = VisionTransformer((3, 224, 224), 14, 1, 768, 12, 100, type=t).cuda()
    x = torch.randn(128, 3, 224, 224, device='cuda')
    for _ in range(iters):
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out = m._embed_fused(x)
            if out is None:
                out = m.linear_mapper(m.patchify(x, 14))
        out.float().square().sum().backward()
else:
    from attention import MSA
    t = next((a for a in sys.argv[2:] if a in ('cheby', 'vanilla', 'fast', 'efficientkan', 'sine')), 'cheby')
    d, h, b = (384, 6, 256) if 'vits' in sys.argv[2:] else (768, 12, 128)
    m = MSA(d, h, type=t).cuda()
    x = torch.randn(b, 197, d, device='cuda', requires_grad=True)
    for _ in range(iters):
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            y = m(x)
        if what == 'bwd':
            y.float().square().sum().backward()
torch.cuda.synchronize()
print('done')
