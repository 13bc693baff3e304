"""Diagonal-Gaussian variational distribution parameters.

Mirrors /root/reference/src/variationaldist.py:4-13 and
/root/reference/src/vardist/diag_gauss.py:6-23 (parameter container only: sampling and
log-density of q happen inside the HIP trajectory kernel)."""
import math

import torch


def encode_params(mean, logdiag):
    return {"mean": mean, "logdiag": logdiag}


def decode_params(params):
    return params["mean"], params["logdiag"]


def initialize(dim, init_sigma=1.0):
    mean = torch.zeros(dim, dtype=torch.float32)
    logdiag = torch.ones(dim, dtype=torch.float32) * math.log(init_sigma)
    return encode_params(mean, logdiag)
