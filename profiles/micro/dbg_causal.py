import torch, sys
sys.path.insert(0, '.')
import magic_amd
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from oracle.nav_ref import RefVLNBert
kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1, do_back_txt=True)
cfg = make_config(128, role="student", **kw)
torch.manual_seed(5)
o = RefVLNBert(cfg).double().eval()
g = VLNBert(None, role="student", config=cfg, device="cuda", compute_dtype=torch.float32)
g.load_state_dict(o.state_dict()); g.eval()
B,N,H,Nz=4,13,128,11
x = torch.randn(B,N,H, dtype=torch.float64, requires_grad=True)
z = torch.randn(Nz,H,dtype=torch.float64); pz = torch.softmax(torch.randn(Nz,dtype=torch.float64),0)
yo = o.vln_bert.causal["back_txt"](x, z.unsqueeze(0).repeat(B,1,1), pz.reshape(1,Nz,1).repeat(B,1,1))
wt=torch.randn(B,N,H,dtype=torch.float64)
(yo*wt).sum().backward()
xg = x.detach().float().cuda().requires_grad_(True)
g.store.zero_grad()
yg = g.causal_blocks["back_txt"](xg, z.float().cuda().unsqueeze(0).repeat(B,1,1), pz.float().cuda().reshape(1,Nz,1).repeat(B,1,1))
print("fwd err", (yg.detach().cpu().double()-yo.detach()).abs().max().item())
(yg*wt.float().cuda()).sum().backward()
torch.cuda.synchronize()
print("dx err", (xg.grad.cpu().double()-x.grad).abs().max().item(), x.grad.abs().max().item())
P = dict(g.named_parameters())
for n,p in o.named_parameters():
    if 'causal' in n and p.grad is not None:
        print(n, (P[n].grad.cpu().double()-p.grad).abs().max().item(), p.grad.abs().max().item())
