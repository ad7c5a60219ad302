"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/tag_hip.h declares;
the host-side mirror keeps the reference's names, constructor arguments and state-dict keys; and the
product package never touches the oracle."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tag_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from texttoaudiogrounding_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = lib.load()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/tag_hip.h but not exported"
    assert sorted(lib.declared_symbols()) == syms, "lib.py signature table and the header disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (tag_[a-z0-9_]+)", out))
    assert set(syms) <= exported
    assert handle.tag_abi_version() == lib.ABI_VERSION == 3


def test_binary_attests_its_sources(tmp_path):
    """tag_build_id() = sha256 of csrc/ at compile time: equal to lib.csrc_sha256() for a current build, and lib.load() refuses a
    library whose id differs from the checked-out sources (a stale .so beside edited kernels)."""
    import sys
    from texttoaudiogrounding_amd import lib
    lib.load()
    assert re.fullmatch(r"[0-9a-f]{64}", lib.build_id())
    assert lib.build_id() == lib.csrc_sha256()
    code = ("import texttoaudiogrounding_amd.lib as L\n"
            "L.csrc_sha256 = lambda: '0' * 64\n"
            "try:\n    L.load()\nexcept RuntimeError as e:\n    print('REFUSED' if 'other kernel sources' in str(e) else e)\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode()
    assert "REFUSED" in out, out


def test_code_object_is_gfx950_only():
    from texttoaudiogrounding_amd import lib
    blob = open(lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_missing_gpu_fails_loudly():
    """No silent CPU fallback: a CPU tensor must raise, not compute."""
    from texttoaudiogrounding_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.logmel(torch.zeros(1, 4000), 1024, 1024, 320, torch.zeros(1024), torch.zeros(513, 64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "texttoaudiogrounding_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_reference_shaped_interface():
    from texttoaudiogrounding_amd.models import align, audio_encoder, audio_text_model, match, text_encoder
    from texttoaudiogrounding_amd import losses
    from oracle import tag_oracle as O
    ae = audio_encoder.Cnn8Rnn(sample_rate=32000, freeze_cnn=False, freeze_bn=False, pretrained=None)
    assert (ae.embed_dim, ae.downsample_ratio, ae.time_resolution, ae.hop_length) == (512, 4, 0.04, 320)
    assert audio_encoder.Cnn8_Rnn is audio_encoder.Cnn8Rnn
    te = text_encoder.EmbeddingAgg(vocab_size=5221, embed_dim=256, pretrained_embedding=None, freeze_embedding=False,
                                   aggregation="mean")
    model = audio_text_model.BiEncoder(ae, te, match.ExpNegL2(l2norm=True, text_level="seq"), shared_dim=256,
                                       cross_encoder=None, add_proj=False, upsample=False,
                                       freeze_audio_encoder=False, freeze_text_encoder=False, pretrained=None)
    keys = set(model.state_dict())
    want = set(O.init_state(text_dim=256, shared_dim=256))
    assert want <= keys
    assert keys - want == {"audio_encoder.melspec_extractor.spectrogram.window",
                           "audio_encoder.melspec_extractor.mel_scale.fb"}     # torchaudio's persistent buffers
    assert sum(p.numel() for p in model.parameters()) == 7_665_344
    m2 = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512),
                                    match.DotProduct(l2norm=False, scale=True, text_level="seq"), 512)
    assert not hasattr(m2, "audio_proj") and sum(p.numel() for p in m2.parameters()) == 8_804_800
    assert isinstance(losses.FrameBceLoss(), torch.nn.Module) and align.DotProduct(l2norm=False, scaled=False)
    # the strong eg_config verbatim (cdur_w2vmean.yaml:45-70): CrnnEncoder(256) + EmbeddingAgg(256) + ExpNegL2
    crnn = audio_text_model.BiEncoder(audio_encoder.CrnnEncoder(sample_rate=32000, embed_dim=256),
                                      text_encoder.EmbeddingAgg(embed_dim=256, vocab_size=5221, aggregation="mean"),
                                      match.ExpNegL2(text_level="seq"), shared_dim=256)
    assert sum(p.numel() for p in crnn.parameters()) == 2_015_074 and not hasattr(crnn, "audio_proj")   # appendix B
    ck = set(crnn.audio_encoder.state_dict())
    assert {"cnn.0.0.weight", "cnn.0.1.weight", "cnn.6.1.weight", "gru.weight_hh_l0_reverse"} <= ck
    assert tuple(crnn.audio_encoder.cnn[0][0].weight.shape) == (1,)                    # 1-channel BatchNorm
    assert (crnn.audio_encoder.hop_length, crnn.audio_encoder.embed_dim) == (640, 256)
    # length arithmetic (row A5) is host-side integer work
    length = torch.div(torch.div(torch.as_tensor([320000, 160000, 1]), 320, rounding_mode="floor") + 1, 4,
                       rounding_mode="floor")
    assert length.tolist() == [250, 125, 0]
    # frozen-CNN / frozen-BN switches
    fz = audio_encoder.Cnn8Rnn(32000, freeze_cnn=True, freeze_bn=True).train()
    assert not fz.fc1.weight.requires_grad and fz.rnn.weight_hh_l0.requires_grad and not fz.bn0.training


def test_yaml_style_construction_with_aliases():
    import texttoaudiogrounding_amd as pkg
    from texttoaudiogrounding_amd.runner import build_model
    pkg.install_aliases(force=True)
    cfg = {"audio_encoder": {"type": "models.audio_encoder.Cnn8Rnn", "args": {"sample_rate": 32000}},
           "text_encoder": {"type": "models.text_encoder.EmbeddingAgg",
                            "args": {"embed_dim": 512, "vocab_size": 5221, "aggregation": "mean"}},
           "match_fn": {"type": "models.match.DotProduct", "args": {"text_level": "seq"}},
           "type": "models.audio_text_model.BiEncoder",
           "args": {"shared_dim": 512, "add_proj": False, "upsample": False}}
    model = build_model(cfg)
    assert type(model).__name__ == "BiEncoder" and type(model).__module__.startswith("texttoaudiogrounding_amd")
    import sys
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "losses", "utils")]:
        del sys.modules[k]


def test_host_resident_token_ids_out_of_range_raise_at_once():
    """ids handed over on the host (the reference's collate output) are range-checked before anything is launched
    and raise IndexError immediately like nn.Embedding (models/text_encoder.py:39); no GPU needed to see it."""
    import torch
    from texttoaudiogrounding_amd.models.text_encoder import EmbeddingAgg
    enc = EmbeddingAgg(50, 16)
    with pytest.raises(IndexError, match="out of range"):
        enc({"text": torch.tensor([[1, 2, 50]]), "text_len": torch.tensor([3])})
    with pytest.raises(IndexError, match="out of range"):
        enc({"text": torch.tensor([[1, -1, 3]]), "text_len": torch.tensor([3])})


def test_pass_size_limit_is_reported_before_any_launch():
    """A batch beyond the kernels' index range raises a RuntimeError that names the limit (round-2 review: it used to surface as
    TAG_EINVAL from the first conv).  Round 4: the conv kernels take a 64-bit image base, so the limit is the 31-bit pixel
    index of the BatchNorm / pool passes (33 520 clips of 10 s), no longer 261 clips of 10 s."""
    from texttoaudiogrounding_amd import ops
    assert ops.max_clips_per_pass(1001) == 33520 and ops.max_clips_per_pass(3001) == 11181
    ops.check_pass_size(300, 1001)
    with pytest.raises(RuntimeError, match="at most 33520 clips"):
        ops.check_pass_size(33521, 1001)


def test_winograd_host_queries_and_dispatch_rule(monkeypatch):
    """Host side of the fused Winograd path (no GPU): which shapes the C side serves, its row / workspace queries, and the dispatch
    rule of dispatch.py (models/panns.py:29-38,49-50 run through it at the benched size)."""
    from texttoaudiogrounding_amd import ops
    q = ops.query
    # channel counts that are multiples of 64: fused form (forward, dgrad and weight gradient) -- no planes, one row per 64-tile block
    for (B, H, W, Ci, Co) in [(64, 250, 8, 512, 512), (64, 1001, 64, 64, 64), (3, 9, 7, 128, 192), (1, 1, 2, 64, 64)]:
        assert q("tag_conv3x3_wino_ok", B, H, W, Ci, Co) == 1
        T = B * ((H + 1) // 2) * ((W + 1) // 2)
        assert q("tag_conv3x3_wino_stats_rows", B, H, W, Co) == (T + 63) // 64
        assert q("tag_conv3x3_wino_ws_bytes", B, H, W, Ci, Co) <= 64
        assert q("tag_conv3x3_wino_wgrad_can_reuse_v", B, H, W, Ci, Co) == 0
        ws = q("tag_conv3x3_wino_wgrad_ws_bytes", B, H, W, Ci, Co)
        shares, rem = divmod(ws, Ci * Co * 9 * 4)
        assert rem == 0 and shares % 2 == 0 and 2 <= shares and (shares // 2) * (Ci // 64) * (Co // 64) <= 256   # S slices: <= one workgroup per CU
    # the plane form keeps the other channel counts it always took; neither form takes these
    assert q("tag_conv3x3_wino_ok", 2, 10, 8, 32, 32) == 1 and q("tag_conv3x3_wino_ws_bytes", 2, 10, 8, 32, 32) > 64
    assert q("tag_conv3x3_wino_ok", 2, 10, 8, 48, 64) == 0 and q("tag_conv3x3_wino_ok", 2, 10, 8, 64, 2048) == 0
    # a tensor of 2^31 bytes or more is refused (buffer descriptors): the caller cuts the batch
    assert q("tag_conv3x3_wino_ok", 256, 750, 16, 256, 256) == 0 and q("tag_conv3x3_wino_ok", 32, 750, 16, 256, 256) == 1
    cuts = ops._batch_chunks(256, 750 * 16 * 256 * 4)
    assert cuts == [(0, 174), (174, 256)] and all(q("tag_conv3x3_wino_ok", b1 - b0, 750, 16, 256, 256) == 1 for b0, b1 in cuts)
    # dispatch rule: fp32, widths 8 .. 64, both channel counts >= 64
    assert ops._wino_shape(8, 512, 512) and ops._wino_shape(64, 64, 64) and ops._wino_shape(32, 64, 128)
    assert not ops._wino_shape(4, 128, 128) and not ops._wino_shape(16, 32, 64) and not ops._wino_shape(128, 64, 64)
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    assert not ops._wino_shape(8, 512, 512)
    monkeypatch.setattr(ops, "CONV_MATH", "fp32")
    monkeypatch.setattr(ops, "CONV_WINOGRAD", False)
    assert not ops._wino_shape(8, 512, 512)


def test_developer_options_are_set_through_the_abi_not_the_environment():
    """tag_set_option: known names are accepted, unknown ones refused with a message; no kernel source reads the environment."""
    from texttoaudiogrounding_amd import lib
    h = lib.load()
    assert h.tag_set_option(b"gemm_big_min", 2048) == 0
    assert h.tag_set_option(b"no_such_switch", 1) != 0 and b"no_such_switch" in h.tag_last_error()
    csrc = os.path.join(ROOT, "texttoaudiogrounding_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    assert set(v[0] for v in lib._ENV_OPTIONS.values()) <= {"conv_impl", "halo_lds_pad", "halo_bn256", "wgrad_wgs", "conv_rows",
                                                             "wgrad_dma", "x3_products", "gemm_big_min", "gru_tile4", "gru_xcd",
                                                             "gru_coop", "mha_mfma"}
