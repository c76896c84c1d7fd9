import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kan-vit_amd'))
import torch
from kanvit import dense as D
torch.manual_seed(0)
M, K, N = 25216, 768, 3072
l1 = torch.nn.Linear(K, N).cuda(); l2 = torch.nn.Linear(N, K).cuda()
x = torch.randn(M, K, device='cuda', requires_grad=True); dy = torch.randn(M, K, device='cuda')
def run(mode):
    D.FF_MODE = mode
    x.grad = None; l1.zero_grad(); l2.zero_grad()
    y = D.feed_forward(x, l1, l2); y.backward(dy)
    return [y.detach().clone(), x.grad.clone(), l1.weight.grad.clone(), l1.bias.grad.clone(), l2.weight.grad.clone(), l2.bias.grad.clone()]
a = run("fp32"); b = run("bf16x3")
# fp64 reference
l1d = torch.nn.Linear(K, N).double().cuda(); l2d = torch.nn.Linear(N, K).double().cuda()
l1d.load_state_dict({k: v.double() for k, v in l1.state_dict().items()}); l2d.load_state_dict({k: v.double() for k, v in l2.state_dict().items()})
xd = x.detach().double().requires_grad_(True)
yd = l2d(torch.relu(l1d(xd))); yd.backward(dy.double())
ref = [yd.detach(), xd.grad, l1d.weight.grad, l1d.bias.grad, l2d.weight.grad, l2d.bias.grad]
for name, u, v, r in zip(["y", "dx", "dw1", "db1", "dw2", "db2"], a, b, ref):
    s = float(r.abs().max())
    print(f"{name:4s} fp32 err {float((u.double()-r).abs().max())/s:.2e}   bf16x3 err {float((v.double()-r).abs().max())/s:.2e}")
def t(mode, n=5):
    D.FF_MODE = mode
    for _ in range(2): run(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run(mode)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("fwd+bwd ms: fp32", t("fp32"), " bf16x3", t("bf16x3"))
