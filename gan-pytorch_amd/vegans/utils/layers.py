"""Helper layers users put inside their generator / adversary definitions (reference: vegans/utils/layers.py:4-33)."""
import torch
from torch import nn


class LayerPrintSize(nn.Module):
    """Debug layer: prints the shape of what flows through it and passes it on unchanged."""

    def forward(self, x):
        print(tuple(x.shape))
        return x


class LayerReshape(nn.Module):
    """Reshape [batch, features] to [batch, *shape] (metadata only: no kernel runs)."""

    def __init__(self, shape):
        super().__init__()
        self.shape = list(shape) if not isinstance(shape, int) else [shape]

    def forward(self, x):
        return torch.reshape(x, (-1, *self.shape))

    def extra_repr(self):
        return "shape={}".format(self.shape)
