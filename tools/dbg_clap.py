import sys, numpy as np, torch, math
import torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from oracle import clap_text_oracle as C
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.lib import call, ptr
dev = torch.device("cuda:0")
g = np.load("/root/repo/tests/golden/clap_text_tiny.npz")
st = {k[2:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("w/")}
ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
B, L = ids.shape; D = 64; H = 4; eps = float(g["eps"]); M = B*L
_keep = []
def gd(k):
    t = st["model."+k].float().to(dev).contiguous(); _keep.append(t); return t
ids_d, mask_d = ids.to(dev), mask.to(dev)
h = torch.empty(M, D, device=dev)
call("tag_roberta_embed_ln", ptr(ids_d), ptr(gd("embeddings.word_embeddings.weight")), ptr(gd("embeddings.token_type_embeddings.weight")),
     ptr(gd("embeddings.position_embeddings.weight")), ptr(gd("embeddings.LayerNorm.weight")), ptr(gd("embeddings.LayerNorm.bias")), eps, ptr(h), B, L, D, 1)
pos = C.position_ids(ids)
e = st["model.embeddings.word_embeddings.weight"][ids] + st["model.embeddings.token_type_embeddings.weight"][0] + st["model.embeddings.position_embeddings.weight"][pos]
e = F.layer_norm(e, (D,), st["model.embeddings.LayerNorm.weight"], st["model.embeddings.LayerNorm.bias"], eps)
torch.cuda.synchronize(); print("embed err", (h.cpu().double().view(B,L,D) - e).abs().max().item(), e.abs().max().item())
p = "model.encoder.layer.0."
hh = e.float().to(dev).view(M, D).contiguous()
wq = torch.cat([st[p+"attention.self.query.weight"], st[p+"attention.self.key.weight"], st[p+"attention.self.value.weight"]],0).float().to(dev).contiguous()
bq = torch.cat([st[p+"attention.self.query.bias"], st[p+"attention.self.key.bias"], st[p+"attention.self.value.bias"]],0).float().to(dev).contiguous()
print("gemm...", flush=True); qkv = ops.gemm(hh, wq, M, 3*D, D, transB=True, bias=bq)
qkv_ref = F.linear(e, wq.cpu().double(), bq.cpu().double())
torch.cuda.synchronize(); print("qkv err", (qkv.cpu().double().view(B,L,3*D) - qkv_ref).abs().max().item(), qkv_ref.abs().max().item())
att = torch.empty(M, D, device=dev)
call("tag_mha_small", ptr(qkv), ptr(mask_d), ptr(att), B, L, H, D//H)
q, k, v = [t.view(B, L, H, D//H).transpose(1,2) for t in qkv_ref.split(D, dim=-1)]
s = q @ k.transpose(-1,-2) / math.sqrt(D//H) + (1.0 - mask.double())[:,None,None,:] * torch.finfo(torch.float32).min
a = (torch.softmax(s, -1) @ v).transpose(1,2).reshape(B, L, D)
torch.cuda.synchronize(); print("att err", (att.cpu().double().view(B,L,D) - a).abs().max().item(), a.abs().max().item(), "scores max", s[s > -1e30].abs().max().item())
