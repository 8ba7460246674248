"""Wrappers around the user's generator / adversary / encoder modules (reference: vegans/utils/networks.py).

Same checks and attributes as the reference (`input_size`, `output_size`, `input_type`, `name`, first-layer width
validation, last-layer type rules, the batch-2 probe forward), but the wrapped network is bound to the HIP
operators instead of being handed to stock torch, and `ngpu` does not create an `nn.DataParallel`: data
parallelism is one process per GPU over RCCL (see vegans.engine.parallel).
"""
import numpy as np
import torch
from torch import nn

from ..engine import modules as hip_modules
from ..engine import ops as hip_ops


class NeuralNetwork(nn.Module):
    def __init__(self, network, name, input_size, device, ngpu):
        super().__init__()
        assert isinstance(network, nn.Module), "network must inherit from nn.Module."
        self.name = name
        if isinstance(input_size, int):
            input_size = (input_size,)
        elif isinstance(input_size, list):
            input_size = tuple(input_size)
        self.input_size = input_size
        self.device = device
        self.ngpu = ngpu
        self.input_type = "Sequential" if _is_indexable(network) else "Object"
        if device != "cuda":
            raise RuntimeError(
                "{}: device={!r} requested, but this build of vegans runs on MI355X only (device='cuda'); "
                "there is no CPU path.".format(name, device))
        self.network = network.to(device)
        self._validate_input()
        hip_modules.bind(self.network)
        hip_modules.prepare_parameters(self.network)
        self.output_size = self._get_output_shape()[1:]

    def forward(self, x):
        # what leaves a network is f32 in every precision mode (in the bf16 mode a feature-shaped output is widened
        # here): losses, the gradient penalty and the caller only ever see f32
        return hip_ops.as_f32(self.network(x))

    # -- checks -------------------------------------------------------------------------------------------
    def _get_iterative_layers(self):
        if self.input_type == "Sequential":
            return list(self.network)
        layers = []
        for child in self.network._modules.values():
            if _is_indexable(child) or isinstance(child, (nn.Sequential, nn.ModuleList)):
                layers.extend(list(child))
            else:
                layers.append(child)
        return layers

    def _validate_input(self):
        first_input = None
        for layer in self._get_iterative_layers():
            for attr in ("in_features", "in_channels", "num_features"):
                if attr in layer.__dict__:
                    first_input = layer.__dict__[attr]
                    break
            if first_input is not None:
                break
        if first_input is None:
            raise UnboundLocalError("no layer with in_features / in_channels / num_features found in {}".format(self.name))
        if first_input == int(np.prod(self.input_size)):
            return True
        if len(self.input_size) > 1 and self.input_size[0] == first_input:
            return True
        raise TypeError(
            "\n\tInput mismatch for {}:\n".format(self.name) +
            "\t\tFirst input layer 'in_features' or 'in_channels': {}. self.input_size: {}.\n".format(
                first_input, self.input_size) +
            "\t\tMaybe forgot to adjust size of input layer for y_dim.")

    def _get_output_shape(self):
        # A batch of two uniform samples drawn on the host, pushed through the network in its current (train)
        # mode: like the reference (networks.py:92-94) this advances the host RNG and the BatchNorm running
        # statistics once per wrapped network.
        sample_input = torch.rand([2, *self.input_size]).to(self.device)
        with torch.no_grad():
            out = self.network(sample_input)
        return tuple(out.shape)

    def summary(self):
        print("Input shape: ", self.input_size)
        print(self.network)
        n_params = sum(p.numel() for p in self.network.parameters())
        print("Total params: {:,}".format(n_params))

    def __str__(self):
        return self.name


def _is_indexable(network):
    try:
        type(network[-1])
        return True
    except (TypeError, IndexError, KeyError):
        return False


def _last_layer_type(network):
    if _is_indexable(network):
        return type(network[-1])
    return type(network._modules["output"])


def _stock(t):
    return hip_modules._STOCK.get(t, t)


class Generator(NeuralNetwork):
    def __init__(self, network, input_size, device, ngpu):
        super().__init__(network, name="Generator", input_size=input_size, device=device, ngpu=ngpu)


class Adversariat(NeuralNetwork):
    """Discriminator (outputs in [0, 1], must end in Sigmoid) or Critic (unbounded, must end in Linear/Identity)."""

    def __init__(self, network, input_size, adv_type, device, ngpu):
        valid_types = ["Discriminator", "Critic"]
        if adv_type == "Discriminator":
            valid_last_layer = [nn.Sigmoid]
        elif adv_type == "Critic":
            valid_last_layer = [nn.Linear, nn.Identity]
        else:
            raise TypeError("adv_type must be one of {}.".format(valid_types))
        self._type = adv_type
        assert _stock(_last_layer_type(network)) in valid_last_layer, (
            "Last layer activation function of {} needs to be one of '{}'.".format(adv_type, valid_last_layer))
        super().__init__(network, name="Adversariat", input_size=input_size, device=device, ngpu=ngpu)

    def predict(self, x):
        return self(x)


class Encoder(NeuralNetwork):
    def __init__(self, network, input_size, device, ngpu):
        valid_last_layer = [nn.Linear, nn.Identity]
        assert _stock(_last_layer_type(network)) in valid_last_layer, (
            "Last layer activation function of Encoder needs to be one of '{}'.".format(valid_last_layer))
        super().__init__(network, name="Encoder", input_size=input_size, device=device, ngpu=ngpu)
