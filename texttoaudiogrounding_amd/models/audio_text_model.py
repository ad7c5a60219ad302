"""Model composition (mirror of BiEncoder, models/audio_text_model.py:16-98 in the reference)."""
from typing import Optional

import torch
import torch.nn as nn

from .. import ops


class BiEncoder(nn.Module):
    def __init__(self, audio_encoder: nn.Module, text_encoder: nn.Module, match_fn: nn.Module, shared_dim: int,
                 cross_encoder: Optional[nn.Module] = None, add_proj: bool = False, upsample: bool = False,
                 freeze_audio_encoder: bool = False, freeze_text_encoder: bool = False,
                 pretrained: Optional[str] = None):
        super().__init__()
        self.audio_encoder = audio_encoder
        self.text_encoder = text_encoder
        self.match_fn = match_fn
        self.cross_encoder = cross_encoder
        if audio_encoder.embed_dim != text_encoder.embed_dim or add_proj:
            self.audio_proj = nn.Linear(audio_encoder.embed_dim, shared_dim)
            self.text_proj = nn.Linear(text_encoder.embed_dim, shared_dim)
        self.interpolate_ratio = self.audio_encoder.downsample_ratio
        self.upsample = upsample
        if pretrained is not None and type(self) is BiEncoder:
            self.load_pretrained(pretrained)
        if freeze_audio_encoder:
            for p in self.audio_encoder.parameters():
                p.requires_grad = False
        if freeze_text_encoder:
            for p in self.text_encoder.parameters():
                p.requires_grad = False

    def load_pretrained(self, ckpt_path, output_fn=print):
        state = torch.load(ckpt_path, map_location="cpu")
        state = state.get("model", state)
        own = self.state_dict()
        matched = {k: v for k, v in state.items() if k in own and own[k].shape == v.shape}
        output_fn(f"BiEncoder: loading {len(matched)}/{len(own)} tensors from {ckpt_path}")
        own.update(matched)
        self.load_state_dict(own)

    def forward(self, input_dict):
        audio_output = self.audio_encoder(input_dict)
        audio_emb = audio_output["embedding"]
        text_emb = self.text_encoder(input_dict)
        forward_dict = {"audio_emb": audio_emb, "text_emb": text_emb, "audio_len": audio_output["length"]}
        if "text_len" in input_dict:
            forward_dict["text_len"] = input_dict["text_len"]
        if self.cross_encoder is not None:
            forward_dict.update(self.cross_encoder(forward_dict))
            text_emb = forward_dict["text_emb"]
        if hasattr(self, "audio_proj"):
            forward_dict["audio_emb"] = ops.LinearFunction.apply(forward_dict["audio_emb"], self.audio_proj.weight,
                                                                 self.audio_proj.bias)
            if "seq_emb" in text_emb:
                text_emb["seq_emb"] = ops.LinearFunction.apply(text_emb["seq_emb"], self.text_proj.weight,
                                                               self.text_proj.bias)
            if self.cross_encoder is not None and "token_emb" in text_emb:
                text_emb["token_emb"] = ops.LinearFunction.apply(text_emb["token_emb"], self.text_proj.weight,
                                                                 self.text_proj.bias)
            # without a cross-encoder token_emb is not projected: no head consumes it (text_level='seq')
        frame_sim = self.match_fn(forward_dict)
        length = audio_output["length"]
        if self.interpolate_ratio != 1 and self.upsample:
            # F.interpolate(mode="linear", align_corners=False) x interpolate_ratio (models/audio_text_model.py:90-97)
            frame_sim = ops.UpsampleLinearFunction.apply(frame_sim, self.interpolate_ratio)
            length = length * self.interpolate_ratio
        return {"frame_sim": frame_sim, "length": length}


class MultiTextBiEncoder(BiEncoder):
    """Weakly supervised variant (mirror of models/audio_text_model.py:101-229 in the reference): every clip comes with N
    phrases; frame_sim (B,T',N) is pooled over time into clip_sim (B,N).  The audio embedding is NOT expanded to
    (B*N,T',D): the grouped head (ops.MatchGroupFunction) scores the N phrases of a clip against the same rows."""

    def __init__(self, audio_encoder: nn.Module, text_encoder: nn.Module, match_fn: nn.Module, shared_dim: int,
                 text_forward_keys: "list[str]", cross_encoder: Optional[nn.Module] = None, pooling: str = "linear_softmax",
                 add_proj: bool = False, upsample: bool = False, freeze_audio_encoder: bool = False,
                 freeze_text_encoder: bool = False, safe_size: Optional[int] = None, pretrained: Optional[str] = None,
                 output_fn=print):
        super().__init__(audio_encoder=audio_encoder, text_encoder=text_encoder, match_fn=match_fn, shared_dim=shared_dim,
                         cross_encoder=cross_encoder, add_proj=add_proj, upsample=upsample,
                         freeze_audio_encoder=freeze_audio_encoder, freeze_text_encoder=freeze_text_encoder)
        self.text_forward_keys = list(text_forward_keys)
        if "text_len" not in self.text_forward_keys:
            self.text_forward_keys.append("text_len")
        if pooling not in ops.POOL_MODES:
            raise Exception(f"Unsupported pooling {pooling}")         # the reference raises at forward time (:215)
        self.pooling = pooling
        self.safe_size = safe_size          # chunking knob of the reference: unnecessary here (no expansion)
        if pretrained is not None and type(self) is MultiTextBiEncoder:
            self.load_pretrained(pretrained, output_fn)

    def _pool(self, sim, B, N, length):
        len_dev = torch.as_tensor(length).long().to(sim.device).contiguous()
        # linear_softmax / max / mean / exp_softmax _with_lens over the valid frames (models/audio_text_model.py:205-215)
        clip_sim = ops.SimPoolFunction.apply(sim.view(B * N, -1, 1), len_dev, None, N, 1, ops.POOL_MODES[self.pooling],
                                             -1).view(B, N)
        if self.interpolate_ratio != 1 and self.upsample:                                    # models/audio_text_model.py:216-224
            sim = ops.UpsampleLinearFunction.apply(sim, self.interpolate_ratio)
            length = length * self.interpolate_ratio
        frame_sim = sim.view(B, N, -1).transpose(1, 2)                                       # (B, T', N)
        return {"frame_sim": frame_sim, "clip_sim": clip_sim, "length": length}

    def _forward_general(self, input_dict):
        """The reference's own data flow (models/audio_text_model.py:148-203): the audio embedding repeated for the N phrases
        of its clip, (B*N)-row cross-encoder and head.  Taken with a cross-encoder or a head other than the grouped
        DotProduct; ``safe_size`` chunking is unnecessary (the heads are row kernels, nothing is materialised per chunk)."""
        audio_output = self.audio_encoder(input_dict)
        audio_emb = audio_output["embedding"]
        if hasattr(self, "audio_proj"):
            audio_emb = ops.LinearFunction.apply(audio_emb, self.audio_proj.weight, self.audio_proj.bias)
        B = audio_emb.size(0)
        N = input_dict[self.text_forward_keys[0]].shape[1]
        text_forward_dict = {}
        for key in self.text_forward_keys:
            x = torch.as_tensor(input_dict[key])
            text_forward_dict[key] = x.reshape(x.shape[0] * x.shape[1], *x.shape[2:])
        text_emb = self.text_encoder(text_forward_dict)
        length = audio_output["length"]
        forward_dict = {"audio_emb": ops.GroupExpandFunction.apply(audio_emb, N), "text_emb": text_emb,
                        "audio_len": torch.as_tensor(length).repeat_interleave(N),
                        "text_len": text_forward_dict["text_len"]}
        if self.cross_encoder is not None:
            forward_dict.update(self.cross_encoder(forward_dict))
        if hasattr(self, "text_proj"):
            text_emb = forward_dict["text_emb"]
            for k in ("seq_emb", "token_emb"):
                if k in text_emb:
                    text_emb[k] = ops.LinearFunction.apply(text_emb[k], self.text_proj.weight, self.text_proj.bias)
        sim = self.match_fn(forward_dict)                                                     # (B*N, T')
        return self._pool(sim.contiguous(), B, N, length)

    def forward(self, input_dict):
        from .match import DotProduct
        if (self.cross_encoder is not None or not isinstance(self.match_fn, DotProduct) or self.match_fn.l2norm
                or self.match_fn.text_level != "seq"):
            return self._forward_general(input_dict)
        audio_output = self.audio_encoder(input_dict)
        audio_emb = audio_output["embedding"]
        if hasattr(self, "audio_proj"):
            audio_emb = ops.LinearFunction.apply(audio_emb, self.audio_proj.weight, self.audio_proj.bias)
        B = audio_emb.size(0)
        N = input_dict[self.text_forward_keys[0]].shape[1]
        text_forward_dict = {}
        for key in self.text_forward_keys:
            x = torch.as_tensor(input_dict[key])
            text_forward_dict[key] = x.reshape(x.shape[0] * x.shape[1], *x.shape[2:])
        text_emb = self.text_encoder(text_forward_dict)
        seq = text_emb["seq_emb"]
        if hasattr(self, "text_proj"):
            seq = ops.LinearFunction.apply(seq, self.text_proj.weight, self.text_proj.bias)
        sim = ops.MatchGroupFunction.apply(audio_emb, seq, N, self.match_fn.scale)            # (B*N, T')
        return self._pool(sim, B, N, audio_output["length"])


class AudioTextAlignByWord(nn.Module):
    """Word-level weak alignment (mirror of models/audio_text_model.py:843-904): every clip's frames against every clip's
    WORD embeddings -- (projected) token_emb -> align.DotProduct (B,B,T',n_word) -> sim_pooling -> (B,B)."""

    def __init__(self, audio_encoder, text_encoder, match_fn, sim_pooling, shared_dim, add_proj=False,
                 freeze_audio_encoder=False, freeze_text_encoder=False):
        super().__init__()
        self.audio_encoder, self.text_encoder, self.match_fn, self.sim_pooling = audio_encoder, text_encoder, match_fn, sim_pooling
        if audio_encoder.embed_dim != text_encoder.embed_dim or add_proj:
            self.audio_proj = nn.Linear(audio_encoder.embed_dim, shared_dim)
            self.text_proj = nn.Linear(text_encoder.embed_dim, shared_dim)
        if freeze_audio_encoder:
            for p in self.audio_encoder.parameters():
                p.requires_grad = False
        if freeze_text_encoder:
            for p in self.text_encoder.parameters():
                p.requires_grad = False

    def forward(self, input_dict):
        audio_output = self.audio_encoder(input_dict)
        audio_emb = audio_output["embedding"]
        if hasattr(self, "audio_proj"):
            audio_emb = ops.LinearFunction.apply(audio_emb, self.audio_proj.weight, self.audio_proj.bias)
        word_emb = self.text_encoder(input_dict)["token_emb"]
        if hasattr(self, "text_proj"):
            word_emb = ops.LinearFunction.apply(word_emb, self.text_proj.weight, self.text_proj.bias)
        sim_matrix = self.match_fn(audio_emb, word_emb.contiguous())
        sim = self.sim_pooling({"sim": sim_matrix, "audio_len": audio_output["length"], "text_len": input_dict["text_len"]})
        output = {"sim": sim}
        if input_dict.get("output_matrix", False):
            output["sim_matrix"] = sim_matrix
        return output


class AudioTextAlignByPhrase(nn.Module):
    """Weakly supervised alignment (mirror of models/audio_text_model.py:907-976 in the reference): every clip against
    every clip's phrases -- align.DotProduct (B,B,T',N) -> sim_pooling -> (B,B) for MaxMarginRankingLoss."""

    def __init__(self, audio_encoder, text_encoder, match_fn, sim_pooling, shared_dim, cross_encoder=None, add_proj=False,
                 freeze_audio_encoder=False, freeze_text_encoder=False):
        super().__init__()
        self.audio_encoder, self.text_encoder, self.match_fn = audio_encoder, text_encoder, match_fn
        self.cross_encoder, self.sim_pooling = cross_encoder, sim_pooling        # stored and never applied, as in the reference (:925, :936-976)
        if audio_encoder.embed_dim != text_encoder.embed_dim or add_proj:
            self.audio_proj = nn.Linear(audio_encoder.embed_dim, shared_dim)
            self.text_proj = nn.Linear(text_encoder.embed_dim, shared_dim)
        if freeze_audio_encoder:
            for p in self.audio_encoder.parameters():
                p.requires_grad = False
        if freeze_text_encoder:
            for p in self.text_encoder.parameters():
                p.requires_grad = False

    def forward(self, input_dict):
        audio_output = self.audio_encoder(input_dict)
        audio_emb = audio_output["embedding"]
        text_key = input_dict["text_key"]
        phrases_emb = self.text_encoder({"text": input_dict[text_key], "text_len": input_dict[f"{text_key}_len"]})
        phrases_num = [int(v) for v in input_dict[f"{text_key}_num"]]
        seq_emb = torch.split(phrases_emb["seq_emb"], phrases_num, dim=0)
        seq_emb = nn.utils.rnn.pad_sequence(seq_emb, batch_first=True)          # (B, max_num, D)
        if hasattr(self, "audio_proj"):                                         # the reference declares but never applies them
            pass
        sim_matrix = self.match_fn(audio_emb, seq_emb.contiguous())
        sim = self.sim_pooling({"sim": sim_matrix, "audio_len": audio_output["length"], "text_len": phrases_num})
        output = {"sim": sim}
        if input_dict.get("output_matrix", False):
            output["sim_matrix"] = sim_matrix
        return output
