import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from bnn_chaos_model_amd import ops
np.set_printoptions(linewidth=220, precision=3, suppress=True)
L = 8
plan = ops.get_plan(0, 0.5, hidden=40, latent=L, depth_in=0, depth_out=0)
d = plan.d
F = 41
offW1 = F + 2 * L; offb1 = offW1 + L * F
B, T = 16, 8
x = torch.zeros(B, T, F, device="cuda")
x[:, :, 0] = 1.0
for k in range(1, F):
    x[:, :, k] = 0.01 * k
eps = torch.zeros(1, B, 2, L, device="cuda")
def run(w):
    out, pre, summ = ops.forward(x, torch.tensor(w[None]).cuda(), eps=eps, plan=plan, debug=True)
    return summ[0].cpu().numpy()
w = np.zeros(d, np.float32); w[offb1:offb1 + L] = np.arange(1, L + 1)
print("A bias only   :", run(w)[0, :L], run(w)[5, :L])
w = np.zeros(d, np.float32)
for n in range(L): w[offW1 + n * F + 0] = n + 1
print("B W[n][0]=n+1 :", run(w)[0, :L])
w = np.zeros(d, np.float32)
for n in range(L): w[offW1 + n * F + n + 1] = 100.0
print("C W[n][n+1]=100 (expect n+1):", run(w)[0, :L])
w = np.zeros(d, np.float32)
for n in range(L): w[offW1 + n * F + 40 - n] = 100.0
print("D W[n][40-n]=100 (expect 40-n):", run(w)[0, :L])
x2 = x.clone(); x2[:, :, 0] = torch.arange(T, device="cuda")[None, :].float()
w = np.zeros(d, np.float32)
for n in range(L): w[offW1 + n * F + 0] = 1.0
out, pre, summ = ops.forward(x2, torch.tensor(w[None]).cuda(), eps=eps, plan=plan, debug=True)
print("E y=t: mean (expect 3.5)", summ[0, 0, :L].cpu().numpy(), "std (expect sqrt(6+1e-5)=2.449)", summ[0, 0, L:].cpu().numpy())
