"""Counterpart of the reference's models/model_utils.py: activation lookup by
name (model_utils.py:5-34) and the PNA degree histogram (model_utils.py:37-58)."""
import torch
import torch.nn as nn

_ACTIVATIONS = {
    "relu": nn.ReLU,
    "leaky_relu": lambda: nn.LeakyReLU(0.01),
    "tanh": nn.Tanh,
    "sigmoid": nn.Sigmoid,
    "gelu": nn.GELU,
    "elu": nn.ELU,
    "selu": nn.SELU,
    "swish": nn.SiLU,
    "silu": nn.SiLU,
    "none": nn.Identity,
}


def _select_activation(activation):
    """String -> activation module; modules pass through unchanged."""
    if isinstance(activation, nn.Module):
        return activation
    key = str(activation).lower()
    if key not in _ACTIVATIONS:
        raise ValueError(f"Activation function {activation} not recognized as a string. "
                         "You can pass the module directly as an argument instead of a string.")
    return _ACTIVATIONS[key]()


def activation_slope(act):
    """Negative-side slope when `act` is ReLU-like (what the fused kernels
    implement), else None."""
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    if isinstance(act, nn.Identity):
        return 1.0
    return None


def _calc_PNA_degrees(in_ds, for_type="molecule"):
    """In-degree histogram over a dataset of (protein, molecule, ...) graph pairs."""
    which = 1 if for_type == "molecule" else 0
    hist = torch.zeros(1, dtype=torch.long)
    for item in in_ds:
        g = item[which]
        deg = torch.bincount(g.edge_index[1], minlength=g.num_nodes)
        h = torch.bincount(deg)
        if h.numel() > hist.numel():
            hist = torch.cat([hist, hist.new_zeros(h.numel() - hist.numel())])
        hist[:h.numel()] += h
    return hist
