"""Is the fc1/top-layer gradient error of the full-length step a discrete ReLU-flip event?  HIP step on waveforms perturbed at
the 1e-7 level (the fp64 oracle's gradients move by 1e-8 under the same perturbation) against ONE fp64 oracle step."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import tag_oracle as O
from tests.test_gpu_path import build_hip_model
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.runner import StrongRunner
dev = torch.device("cuda:0")
st = O.init_state(seed=5, logit_gain=120.0)
batch = O.synthetic_batch(6, 320000, seed=99, ragged=True)
names = ["audio_encoder.fc1.weight", "audio_encoder.fc1.bias", "audio_encoder.conv_block4.bn2.weight",
         "audio_encoder.conv_block3.bn2.bias", "audio_encoder.rnn.weight_ih_l0"]
ref = None
for trial in range(7):
    torch.manual_seed(7)
    model = build_hip_model(st, "dot", dev).train()
    runner = StrongRunner(model, lr=1e-3, max_grad_norm=1.0, device=str(dev))
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    if trial:
        gen = torch.Generator().manual_seed(trial)
        b["waveform"] = (b["waveform"].double() * (1 + 6e-8 * torch.randn(b["waveform"].shape, generator=gen, dtype=torch.float64))).float()
    loss = runner.forward_backward(b)
    if ref is None:
        info = model.audio_encoder._last_dropout
        shapes = [(6, 500, 32, 64), (6, 250, 16, 128), (6, 250, 8, 256), (6, 250, 4, 512)]
        masks = {f"drop{i + 1}": ops.dropout_mask(info["seeds"][i], shp, 0.2, dev).cpu().permute(0, 3, 1, 2).double()
                 for i, shp in enumerate(shapes)}
        masks["drop5"] = ops.dropout_mask(info["seeds"][4], (6, 250, 512), 0.5, dev).cpu().double()
        st_o = O.state_to(st, torch.float64, requires_grad=True)
        bo = dict(batch); bo["waveform"], bo["label"] = batch["waveform"].double(), batch["label"].double()
        oloss, _ = O.train_step_loss(st_o, bo, "dot", "cnn8rnn", True, None, masks)
        oloss.backward()
        ref = {n: st_o[n].grad for n in names}
    g = dict(model.named_parameters())
    print(trial, " | ".join(f"{n.split('audio_encoder.')[-1]} {(g[n].grad.cpu().double() - ref[n]).abs().max().item() / ref[n].abs().max().item():.1e}" for n in names), flush=True)
