"""CPU restatement of the LAION-CLAP text tower as the reference uses it for inference
(models/hf_modeling_grounding.py:183-199 `LaionClapEncoder`): TEST INFRASTRUCTURE ONLY -- imported by tests/, never by
the product path.

The arithmetic lives in a third-party dependency, Hugging Face `transformers` (`ClapTextModel` + `ClapProjectionLayer`,
models/clap/modeling_clap.py; the reference pins no version, requirements.txt lists none; this image has 5.x): a
RoBERTa-style post-LayerNorm encoder.  Restated here over the HF state-dict keys:

  position ids   cumsum(input_ids != pad) * (input_ids != pad) + pad            (pad_token_id = 1)
  embeddings     LayerNorm(word[ids] + token_type[0] + position[pos])            eps = layer_norm_eps
  layer x N      a = softmax(q k^T / sqrt(d_head) + (1 - mask) * -inf) v;  h = LN(h + dense(a));
                 h = LN(h + dense2(gelu_erf(dense1(h))))
  pooler         tanh(dense(h[:, 0]))
  projection     linear2(relu(linear1(x)))
  LaionClapEncoder.forward: token_emb = projection(last_hidden_state); seq_emb = normalize(projection(pooler_output))

Pinned by tests/golden/clap_text_tiny.npz, which tests/golden/make_golden_clap.py produced by running the real
`transformers` modules (random weights, tiny config) in this container and asserting this restatement equals them.
Real LAION weights need the network and are NOT pinned (structure only), as SURVEY.md section 8(c) states.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def position_ids(input_ids, pad_id=1):
    mask = input_ids.ne(pad_id).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad_id


def text_model_forward(st, input_ids, attention_mask, n_heads, eps=1e-12, pad_id=1, prefix=""):
    """st: HF ClapTextModel state dict (keys 'embeddings.*', 'encoder.layer.N.*', 'pooler.dense.*').
    Returns (last_hidden_state (B,L,D), pooler_output (B,D))."""
    g = lambda k: st[prefix + k]
    pos = position_ids(input_ids, pad_id)
    h = g("embeddings.word_embeddings.weight")[input_ids] + g("embeddings.token_type_embeddings.weight")[0] \
        + g("embeddings.position_embeddings.weight")[pos]
    D = h.shape[-1]
    h = F.layer_norm(h, (D,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    B, L, _ = h.shape
    dh = D // n_heads
    neg = torch.finfo(h.dtype).min
    add_mask = (1.0 - attention_mask.to(h.dtype))[:, None, None, :] * neg          # (B,1,1,L) on the keys
    n_layers = 1 + max(int(k[len(prefix):].split(".")[2]) for k in st if k.startswith(prefix + "encoder.layer."))
    for i in range(n_layers):
        p = f"encoder.layer.{i}."
        q = F.linear(h, g(p + "attention.self.query.weight"), g(p + "attention.self.query.bias"))
        k = F.linear(h, g(p + "attention.self.key.weight"), g(p + "attention.self.key.bias"))
        v = F.linear(h, g(p + "attention.self.value.weight"), g(p + "attention.self.value.bias"))
        q, k, v = (t.view(B, L, n_heads, dh).transpose(1, 2) for t in (q, k, v))
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + add_mask
        a = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, L, D)
        a = F.linear(a, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"))
        h = F.layer_norm(h + a, (D,), g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"), eps)
        f = F.gelu(F.linear(h, g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias")))
        f = F.linear(f, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
        h = F.layer_norm(h + f, (D,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
    pooled = torch.tanh(F.linear(h[:, 0], g("pooler.dense.weight"), g("pooler.dense.bias")))
    return h, pooled


def projection(st, x, prefix=""):
    x = F.relu(F.linear(x, st[prefix + "linear1.weight"], st[prefix + "linear1.bias"]))
    return F.linear(x, st[prefix + "linear2.weight"], st[prefix + "linear2.bias"])


def laion_clap_encoder_forward(st, input_ids, attention_mask, n_heads, eps=1e-12):
    """models/hf_modeling_grounding.py:192-199.  st keys: 'model.*' (text model), 'projection.*'."""
    h, pooled = text_model_forward(st, input_ids, attention_mask, n_heads, eps, prefix="model.")
    token_emb = projection(st, h, "projection.")
    seq_emb = F.normalize(projection(st, pooled, "projection."), dim=-1)
    return {"seq_emb": seq_emb, "token_emb": token_emb}


def init_text_state(seed, vocab=50265, hidden=768, layers=12, inter=3072, max_pos=514, proj=512, scale=0.05):
    """Seeded random LaionClapEncoder state (keys as the reference's module would have) for structural tests."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * scale
    st = {"model.embeddings.word_embeddings.weight": r(vocab, hidden),
          "model.embeddings.token_type_embeddings.weight": r(1, hidden),
          "model.embeddings.position_embeddings.weight": r(max_pos, hidden),
          "model.embeddings.LayerNorm.weight": 1 + r(hidden), "model.embeddings.LayerNorm.bias": r(hidden)}
    for i in range(layers):
        p = f"model.encoder.layer.{i}."
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            st[p + nm + ".weight"], st[p + nm + ".bias"] = r(hidden, hidden), r(hidden)
        st[p + "intermediate.dense.weight"], st[p + "intermediate.dense.bias"] = r(inter, hidden), r(inter)
        st[p + "output.dense.weight"], st[p + "output.dense.bias"] = r(hidden, inter), r(hidden)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            st[p + nm + ".weight"], st[p + nm + ".bias"] = 1 + r(hidden), r(hidden)
    st["model.pooler.dense.weight"], st["model.pooler.dense.bias"] = r(hidden, hidden), r(hidden)
    st["projection.linear1.weight"], st["projection.linear1.bias"] = r(proj, hidden), r(proj)
    st["projection.linear2.weight"], st["projection.linear2.bias"] = r(proj, proj), r(proj)
    return st


def synthetic_tokens(B, L, seed, vocab=50265, pad_id=1):
    """RoBERTa-style token batch: <s>=0 ... </s>=2, right-padded with pad_id; lengths 3..L."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((B, L), pad_id, dtype=torch.long)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(3, L + 1, (1,), generator=g)) if b else L
        ids[b, 0] = 0
        ids[b, 1:n - 1] = torch.randint(3, vocab, (n - 2,), generator=g)
        ids[b, n - 1] = 2
        mask[b, :n] = 1
    return ids, mask
