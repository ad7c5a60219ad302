"""Inference wrapper (mirror of models/hf_modeling_grounding.py:97-352 in the reference, BASELINE config 5):
Cnn8Rnn audio encoder + LAION-CLAP text tower + audio/text projections + DotProduct, forward only, on the HIP path.

Differences forced by the environment, not by design:
* the reference builds its text tower with ``ClapModel.from_pretrained("laion/clap-htsat-fused")`` and tokenises with
  ``AutoTokenizer`` -- both need the network.  Here ``LaionClapEncoder`` is built from a configuration (RoBERTa-base
  shape by default) with the SAME module / state-dict names as ``ClapModel.text_model`` + ``ClapModel.text_projection``,
  so a downloaded checkpoint loads with ``load_state_dict``; ``forward`` takes ``input_ids`` / ``attention_mask``
  (what the tokenizer would return).
* ``transformers`` is used for the reference's two BASE classes only (``PreTrainedModel`` / ``PretrainedConfig``: the
  ``from_pretrained`` / ``save_pretrained`` surface of models/hf_modeling_grounding.py:305-352) and, when one is available
  locally, the tokenizer; the towers are plain ``nn.Module`` holders of the parameters and the arithmetic runs in libtag_hip.so (tag_roberta_embed_ln, tag_gemm with bias/GELU/tanh/ReLU epilogues, tag_mha_small, tag_add_layernorm,
  tag_l2norm_rows_forward).
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from ..lib import call, ptr
from .audio_encoder import Cnn8Rnn  # noqa: F401  (same encoder as the training path; re-exported like the reference)
from .audio_text_model import BiEncoder  # noqa: F401
from .match import DotProduct  # noqa: F401

#: ClapTextConfig defaults of "laion/clap-htsat-fused" (RoBERTa-base + 512-d projection)
CLAP_TEXT_DEFAULTS = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                          intermediate_size=3072, max_position_embeddings=514, projection_dim=512, layer_norm_eps=1e-12,
                          pad_token_id=1)


class _Holder(nn.Module):
    """Parameter container with free-form children (names mirror the Hugging Face module tree)."""


def _text_model(cfg):
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    m = _Holder()
    m.embeddings = _Holder()
    m.embeddings.word_embeddings = nn.Embedding(cfg["vocab_size"], D, padding_idx=cfg["pad_token_id"])
    m.embeddings.token_type_embeddings = nn.Embedding(1, D)
    m.embeddings.position_embeddings = nn.Embedding(cfg["max_position_embeddings"], D, padding_idx=cfg["pad_token_id"])
    m.embeddings.LayerNorm = nn.LayerNorm(D, eps=cfg["layer_norm_eps"])
    # index buffers some transformers versions keep in the state dict (unused by the HIP path)
    m.embeddings.register_buffer("position_ids", torch.arange(cfg["max_position_embeddings"]).unsqueeze(0))
    m.embeddings.register_buffer("token_type_ids", torch.zeros(1, cfg["max_position_embeddings"], dtype=torch.long))
    m.encoder = _Holder()
    layers = []
    for _ in range(cfg["num_hidden_layers"]):
        lyr = _Holder()
        lyr.attention = _Holder()
        lyr.attention.self = _Holder()
        lyr.attention.self.query, lyr.attention.self.key, lyr.attention.self.value = (nn.Linear(D, D) for _ in range(3))
        lyr.attention.output = _Holder()
        lyr.attention.output.dense = nn.Linear(D, D)
        lyr.attention.output.LayerNorm = nn.LayerNorm(D, eps=cfg["layer_norm_eps"])
        lyr.intermediate = _Holder()
        lyr.intermediate.dense = nn.Linear(D, I)
        lyr.output = _Holder()
        lyr.output.dense = nn.Linear(I, D)
        lyr.output.LayerNorm = nn.LayerNorm(D, eps=cfg["layer_norm_eps"])
        layers.append(lyr)
    m.encoder.layer = nn.ModuleList(layers)
    m.pooler = _Holder()
    m.pooler.dense = nn.Linear(D, D)
    return m


class LaionClapEncoder(nn.Module):
    """models/hf_modeling_grounding.py:183-199.  ``forward(input_dict)`` with keys ``input_ids`` / ``attention_mask``
    (B, L <= 64) -> ``{"seq_emb": (B, P) L2-normalised, "token_emb": (B, L, P)}``.  Inference only."""

    def __init__(self, model_type: Optional[str] = None, config: Optional[dict] = None):
        super().__init__()
        cfg = dict(CLAP_TEXT_DEFAULTS)
        cfg.update(config or {})
        self.config = cfg
        self.model_type = model_type            # kept for interface parity; weights come from load_state_dict
        self.model = _text_model(cfg)
        self.projection = _Holder()
        self.projection.linear1 = nn.Linear(cfg["hidden_size"], cfg["projection_dim"])
        self.projection.linear2 = nn.Linear(cfg["projection_dim"], cfg["projection_dim"])
        self.embed_dim = cfg["projection_dim"]
        self._qkv = None

    def _apply(self, fn, *a, **k):              # .to(device) / .float(): drop the packed q|k|v weights
        self._qkv = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._qkv = None
        state_dict = dict(state_dict)
        for name in ("model.embeddings.position_ids", "model.embeddings.token_type_ids"):   # optional in checkpoints
            state_dict.setdefault(name, self.state_dict()[name])
        return super().load_state_dict(state_dict, *a, **k)

    def _packed_qkv(self):
        if self._qkv is None:
            self._qkv = []
            for lyr in self.model.encoder.layer:
                s = lyr.attention.self
                self._qkv.append((torch.cat([s.query.weight, s.key.weight, s.value.weight], 0).detach().contiguous(),
                                  torch.cat([s.query.bias, s.key.bias, s.value.bias], 0).detach().contiguous()))
        return self._qkv

    @torch.no_grad()
    def forward(self, input_dict: Dict):
        ids = input_dict["input_ids"].long()
        mask = input_dict["attention_mask"].long()
        dev = self.projection.linear1.weight.device
        ids, mask = ids.to(dev).contiguous(), mask.to(dev).contiguous()
        B, L = ids.shape
        if L > 64:
            raise ValueError("tag_mha_small handles phrases of at most 64 tokens")
        cfg = self.config
        D, H = cfg["hidden_size"], cfg["num_attention_heads"]
        eps, M = float(cfg["layer_norm_eps"]), B * L
        emb = self.model.embeddings
        h = torch.empty(M, D, device=dev, dtype=torch.float32)
        call("tag_roberta_embed_ln", ptr(ids), ptr(emb.word_embeddings.weight), ptr(emb.token_type_embeddings.weight),
             ptr(emb.position_embeddings.weight), ptr(emb.LayerNorm.weight), ptr(emb.LayerNorm.bias), eps, ptr(h), B, L,
             D, int(cfg["pad_token_id"]))
        for lyr, (wqkv, bqkv) in zip(self.model.encoder.layer, self._packed_qkv()):
            qkv = ops.gemm(h, wqkv, M, 3 * D, D, transB=True, bias=bqkv)
            att = torch.empty(M, D, device=dev, dtype=torch.float32)
            call("tag_mha_small", ptr(qkv), ptr(mask), ptr(att), B, L, H, D // H)
            ao = lyr.attention.output
            a = ops.gemm(att, ao.dense.weight, M, D, D, transB=True, bias=ao.dense.bias)
            h1 = torch.empty_like(h)
            call("tag_add_layernorm", ptr(a), ptr(h), ptr(ao.LayerNorm.weight), ptr(ao.LayerNorm.bias), eps, ptr(h1), M, D)
            inter = lyr.intermediate.dense
            f = ops.gemm(h1, inter.weight, M, inter.weight.shape[0], D, transB=True, bias=inter.bias, act=3)
            od = lyr.output
            f2 = ops.gemm(f, od.dense.weight, M, D, inter.weight.shape[0], transB=True, bias=od.dense.bias)
            h = torch.empty_like(h1)
            call("tag_add_layernorm", ptr(f2), ptr(h1), ptr(od.LayerNorm.weight), ptr(od.LayerNorm.bias), eps, ptr(h), M, D)
        pd = self.model.pooler.dense
        pooled = ops.gemm(h, pd.weight, B, D, D, transB=True, lda=L * D, bias=pd.bias, act=4)     # rows = h[:, 0]
        p1, p2 = self.projection.linear1, self.projection.linear2
        P = p1.weight.shape[0]

        def proj(x, rows):
            t = ops.gemm(x, p1.weight, rows, P, D, transB=True, bias=p1.bias, act=1)
            return ops.gemm(t, p2.weight, rows, P, P, transB=True, bias=p2.bias)

        token_emb = proj(h, M).view(B, L, P)
        seq = proj(pooled, B)
        seq_emb = torch.empty_like(seq)
        call("tag_l2norm_rows_forward", ptr(seq), ptr(seq_emb), B, P)
        return {"seq_emb": seq_emb, "token_emb": token_emb, "last_hidden_state": h.view(B, L, D), "pooler_output": pooled}


try:                                            # the reference's own base classes when transformers is installed (it is optional
    from transformers import PreTrainedModel as _ModelBase, PretrainedConfig as _ConfigBase   # for the training path)
    _HAVE_TRANSFORMERS = True
except Exception:                               # noqa: BLE001 - any import problem: plain objects, same attributes
    _ModelBase, _ConfigBase, _HAVE_TRANSFORMERS = nn.Module, object, False


class Cnn8RnnLaionClapGroundingConfig(_ConfigBase):
    """models/hf_modeling_grounding.py:296-306: a ``PretrainedConfig`` (``save_pretrained`` / ``from_pretrained`` /
    ``config.json``) with the reference's three fields.  ``text_config`` (optional, not in the reference) overrides entries of
    CLAP_TEXT_DEFAULTS -- the text tower is built from a configuration here, not downloaded (module docstring)."""
    # the reference leaves model_type empty and reaches its classes through config.json's auto_map + trust_remote_code; a model_type
    # lets AutoConfig / AutoModel resolve a LOCAL directory saved by this class without remote code (registered below)
    model_type = "cnn8rnn_laionclap_grounding"

    def __init__(self, sample_rate: int = 32000, shared_dim: int = 512, text_encoder_name: str = "laion/clap-htsat-fused",
                 text_config: Optional[dict] = None, **kwargs):
        self.sample_rate = sample_rate
        self.shared_dim = shared_dim
        self.text_encoder_name = text_encoder_name
        self.text_config = text_config
        if _HAVE_TRANSFORMERS:
            super().__init__(**kwargs)


class Cnn8RnnLaionClapGroundingModel(_ModelBase):
    """models/hf_modeling_grounding.py:305-352: a ``PreTrainedModel`` with ``config_class`` set, so ``save_pretrained`` /
    ``from_pretrained`` / ``AutoModel.from_pretrained(..., trust_remote_code=True)`` (README.md:7-39) work on it, called like the
    reference: ``model(audio, audio_len, text)`` -> frame_sim (B, T').

    ``text`` is a ``List[str]`` (the reference's call) when a tokenizer is available -- ``model.text_tokenizer`` is loaded from
    ``config.text_encoder_name`` if that resolves LOCALLY (a directory, or the Hugging Face cache; there is no network on the
    build / GPU boxes) and can be injected (``model.text_tokenizer = tok``: any callable with the tokenizer's
    ``(text, padding=True, return_tensors="pt", truncation=True)`` signature) -- or the tokenizer's output itself (a mapping with
    ``input_ids`` and ``attention_mask``).  Strings without a tokenizer raise a clear error instead of guessing."""
    config_class = Cnn8RnnLaionClapGroundingConfig
    base_model_prefix = "model"
    # the audio tower's activations at 30 s x 64 clips are GBs: keep the module on one device
    _no_split_modules = ["BiEncoder"]

    def __init__(self, config: Optional[Cnn8RnnLaionClapGroundingConfig] = None, max_clips_per_pass: int = 64):
        config = config or Cnn8RnnLaionClapGroundingConfig()
        if _HAVE_TRANSFORMERS:
            super().__init__(config)
        else:
            super().__init__()
            self.config = config
        self.model = BiEncoder(audio_encoder=Cnn8Rnn(sample_rate=config.sample_rate),
                               text_encoder=LaionClapEncoder(config.text_encoder_name, config.text_config),
                               match_fn=DotProduct(), shared_dim=config.shared_dim, add_proj=True)
        self.max_clips_per_pass = max_clips_per_pass
        self.text_tokenizer = self._local_tokenizer(config.text_encoder_name)
        if _HAVE_TRANSFORMERS:
            self._tag_constructing = True       # post_init() must not redo the sub-modules' own initialisation (see _init_weights)
            self.post_init()
            self._tag_constructing = False

    @staticmethod
    def _local_tokenizer(name):
        """AutoTokenizer of ``name`` when it can be had WITHOUT the network (models/hf_modeling_grounding.py:326-328 downloads it)."""
        if not _HAVE_TRANSFORMERS or not name:
            return None
        try:
            from transformers import AutoTokenizer
            return AutoTokenizer.from_pretrained(name, local_files_only=True)
        except Exception:                       # noqa: BLE001 - not cached / no such directory: the caller injects one
            return None

    def _init_weights(self, module):
        """Called by ``from_pretrained`` for modules whose weights are MISSING from the checkpoint (the constructors' own
        initialisation is skipped when the model is materialised from a checkpoint): PyTorch's default for the standard layer types,
        so that a partial checkpoint never leaves uninitialised memory behind.  The reference defines no ``_init_weights`` either."""
        if getattr(self, "_tag_constructing", False):
            return                              # fresh construction: the constructors' initialisation stands, as in the reference
        if isinstance(module, (nn.Linear, nn.Embedding, nn.LayerNorm, nn.BatchNorm2d, nn.GRU, nn.Conv2d)):
            # a module with ONE missing key (a LayerNorm bias, a BatchNorm counter) is handed over whole: what the checkpoint did
            # deliver (transformers marks loaded tensors ``_is_hf_initialized``) survives the reset
            own = list(module.parameters(recurse=False)) + [b for b in module.buffers(recurse=False) if b is not None]
            loaded = [(t, t.detach().clone()) for t in own if getattr(t, "_is_hf_initialized", False)]
            module.reset_parameters()
            with torch.no_grad():
                for t, v in loaded:
                    t.copy_(v)

    if not _HAVE_TRANSFORMERS:
        @property
        def device(self):
            return next(self.parameters()).device

    def tokenize(self, text: List[str]) -> Dict:
        if self.text_tokenizer is None:
            raise RuntimeError(
                "Cnn8RnnLaionClapGroundingModel got strings but has no tokenizer: the tokenizer of "
                f"{self.config.text_encoder_name!r} is not available locally (no network here).  Set model.text_tokenizer = "
                "AutoTokenizer.from_pretrained(<local path>), or pass the tokenizer output (input_ids, attention_mask)")
        return self.text_tokenizer(list(text), padding=True, return_tensors="pt", truncation=True)

    @torch.no_grad()
    def forward(self, audio: torch.Tensor, audio_len, text):
        dev = self.device
        if isinstance(text, (list, tuple)) and (not text or isinstance(text[0], str)):
            text = self.tokenize(text)          # models/hf_modeling_grounding.py:340-343
        audio = audio.to(dev)
        B = audio.shape[0]
        ids, mask = text["input_ids"], text["attention_mask"]
        audio_len = torch.as_tensor(audio_len)
        outs = []
        # batches are processed in passes of max_clips_per_pass (64): the first conv output of 64 x 30 s clips is 3.15 GB, and the
        # persistent GRU holds <= 128 sequences per launch (the kernels themselves take any batch since round 4)
        frames = audio.shape[1] // self.model.audio_encoder.hop_length + 1
        per_pass = max(1, min(self.max_clips_per_pass, ops.max_clips_per_pass(frames)))
        for b0 in range(0, B, per_pass):
            sl = slice(b0, min(B, b0 + per_pass))
            d = {"waveform": audio[sl], "waveform_len": audio_len[sl], "input_ids": ids[sl], "attention_mask": mask[sl],
                 "text_len": mask[sl].sum(-1), "specaug": False}
            outs.append(self.model(d)["frame_sim"])
        return torch.cat(outs, 0)


if _HAVE_TRANSFORMERS:
    try:                                        # AutoModel.from_pretrained(<directory saved by save_pretrained>) resolves to this class
        from transformers import AutoConfig, AutoModel
        AutoConfig.register(Cnn8RnnLaionClapGroundingConfig.model_type, Cnn8RnnLaionClapGroundingConfig)
        AutoModel.register(Cnn8RnnLaionClapGroundingConfig, Cnn8RnnLaionClapGroundingModel)
    except Exception:                           # noqa: BLE001 - already registered (module re-import) or an older transformers
        pass
