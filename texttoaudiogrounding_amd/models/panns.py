"""PANNs conv block parameters (mirror of models/panns.py:20-44 in the reference).

The block owns the same parameters / buffers / state-dict keys as the reference ConvBlock
(conv1, conv2 without bias; bn1, bn2).  Its arithmetic is executed inside the fused
Cnn8Rnn HIP engine (texttoaudiogrounding_amd.ops.Cnn8RnnFunction), which reads these
parameters directly; ``ConvBlock.forward`` is the block on its own (two conv3x3_bn_relu[_pool] HIP stages).
"""
import torch
import torch.nn as nn

from .. import ops
from .. import torch_ops  # noqa: F401  (registers torch.ops.tag.*)


def init_layer(layer):
    nn.init.xavier_uniform_(layer.weight)
    if getattr(layer, "bias", None) is not None:
        layer.bias.data.fill_(0.0)


def init_bn(bn):
    bn.bias.data.fill_(0.0)
    bn.weight.data.fill_(1.0)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                               bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                               bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.init_weight()

    def init_weight(self):
        init_layer(self.conv1)
        init_layer(self.conv2)
        init_bn(self.bn1)
        init_bn(self.bn2)

    def forward(self, input, pool_size=(2, 2), pool_type="avg"):
        """Standalone ConvBlock.forward (models/panns.py:46-62): input NCHW (B,C,T,F) -> (B,C_out,T/ph,F/pw) through two
        ``torch.ops.tag.conv3x3_bn_relu_pool`` stages.  Inside Cnn8Rnn the fused engine
        (ops.Cnn8RnnFunction) reads the same parameters and never materialises the intermediate activations; this method is
        the block on its own, for callers that compose PANNs blocks themselves."""
        if pool_type not in ops.POOL_TYPES:
            raise Exception("Incorrect argument!")
        ph, pw = (pool_size, pool_size) if isinstance(pool_size, int) else (int(pool_size[0]), int(pool_size[1]))
        x = input.permute(0, 2, 3, 1).contiguous()               # channels-last (plumbing copy)
        for m in (self.bn1, self.bn2):
            if m.training:
                m.num_batches_tracked += 1
        for conv, bn, (kh, kw), pool in ((self.conv1, self.bn1, (1, 1), "avg"), (self.conv2, self.bn2, (ph, pw), pool_type)):
            r = torch.ops.tag.conv3x3_bn_relu_pool(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                   bn.training, bn.momentum, bn.eps, kh, kw, ops.POOL_TYPES[pool])
            x = r[0]
            if bn.training:
                with torch.no_grad():
                    bn.running_mean.copy_(r[6])
                    bn.running_var.copy_(r[7])
        return x.permute(0, 3, 1, 2).contiguous()
