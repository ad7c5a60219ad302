"""PANNs conv block parameters (mirror of models/panns.py:20-44 in the reference).

The block owns the same parameters / buffers / state-dict keys as the reference ConvBlock
(conv1, conv2 without bias; bn1, bn2).  Its arithmetic is executed inside the fused
Cnn8Rnn HIP engine (texttoaudiogrounding_amd.ops.Cnn8RnnFunction), which reads these
parameters directly; the block is a parameter container, not a compute graph.
"""
import torch.nn as nn


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
        raise RuntimeError("ConvBlock is executed by the fused Cnn8Rnn HIP engine; call Cnn8Rnn.forward")
