"""Score-network parameter construction (the forward passes live in csrc/cmcd_kernels.hip).

Mirrors /root/reference/src/nn.py:21-72 ("geffner") and /root/reference/src/nn_dds.py:55-70,91-192
("dds" PISNet).  Parameter trees use the reference's own naming and ordering so that a flattened
parameter vector has the same layout as the reference's `ravel_pytree` output
(/root/reference/src/mcdboundingmachine.py:122):
  geffner: {"emb": [K,e], "factor_sn": [], "nn": [(W1,b1),(W2,b2),(W3,b3)]}
  dds:     haiku naming, "drift_net" {timestep_phase[1,64]}, "drift_net/~/linear{,_1,_2,_3}" {b,w},
           "drift_net/~/linear_zero" {b,w}
Initial values follow the same distributions as stax / haiku defaults, drawn from a torch
generator (the reference's exact jax.random init streams are not reproduced).
"""
import collections
import math

import torch

ScoreNet = collections.namedtuple("ScoreNet", ["arch", "x_dim", "emb_dim", "nbridges", "width", "rho_dim"],
                                  defaults=(0,))
ScoreNet.__doc__ = ("Static description of apply_fun_sn (the 4th entry of params_fixed).  rho_dim = x_dim for the "
                    "momentum modes (network input concat(z, rho), /root/reference/src/mcdboundingmachine.py:82-98), else 0.")

DDS_WIDTH = 64  # PISNet overwrites fully_connected_units with [64, 64]  (nn_dds.py:95)


def _trunc_normal(gen, shape, std):
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return t * std


def _glorot_normal(gen, fan_in, fan_out):
    # jax.nn.initializers.glorot_normal = variance_scaling(1, "fan_avg", "truncated_normal")
    std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    return _trunc_normal(gen, (fan_in, fan_out), std)


def initialize_embedding(gen, nbridges, emb_dim, factor=0.05):
    """/root/reference/src/nn.py:17-18"""
    return torch.randn(nbridges, emb_dim, generator=gen, dtype=torch.float32) * factor


def initialize_mcd_network(x_dim, emb_dim, nbridges, rho_dim=0, nlayers=4):
    """/root/reference/src/nn.py:42-72 (nlayers is ignored there as well)."""
    if rho_dim not in (0, x_dim):
        raise NotImplementedError("rho_dim must be 0 or x_dim")   # the reference only ever passes dim (mcdboundingmachine.py:92)
    in_dim = x_dim + rho_dim + emb_dim                                # nn.py:43
    spec = ScoreNet("geffner", x_dim, emb_dim, nbridges, in_dim, rho_dim)

    def init_fun(seed, input_shape=None):
        gen = torch.Generator().manual_seed(int(seed))
        dense = []
        for out in (in_dim, in_dim, x_dim):
            w = _glorot_normal(gen, in_dim, out)
            b = torch.randn(out, generator=gen, dtype=torch.float32) * 1e-2  # stax normal(1e-2)
            dense.append((w, b))
        params = {"nn": dense, "emb": initialize_embedding(gen, nbridges, emb_dim),
                  "factor_sn": torch.zeros((), dtype=torch.float32)}
        return (x_dim,), params

    return init_fun, spec


def initialize_pis_network(x_dim, fully_connected_units=None, rho_dim=0):
    """/root/reference/src/nn_dds.py:55-70,91-127 — always 64 wide."""
    if rho_dim not in (0, x_dim):
        raise NotImplementedError("rho_dim must be 0 or x_dim")
    h = DDS_WIDTH
    spec = ScoreNet("dds", x_dim, h, 0, h, rho_dim)

    def _linear(gen, fan_in, fan_out):
        # haiku Linear: w ~ TruncatedNormal(1/sqrt(fan_in)), b = 0
        return {"b": torch.zeros(fan_out, dtype=torch.float32),
                "w": _trunc_normal(gen, (fan_in, fan_out), 1.0 / math.sqrt(fan_in))}

    def init_fun(seed, input_shape=None):
        gen = torch.Generator().manual_seed(int(seed))
        params = {
            "drift_net": {"timestep_phase": torch.zeros(1, h, dtype=torch.float32)},
            "drift_net/~/linear": _linear(gen, 2 * h, h),
            "drift_net/~/linear_1": _linear(gen, h, h),
            "drift_net/~/linear_2": _linear(gen, x_dim + rho_dim + h, h),   # concat(x, t_net) with x = [z; rho]
            "drift_net/~/linear_3": _linear(gen, h, h),
            # LinearZero (nn_dds.py:179-192)
            "drift_net/~/linear_zero": {"b": torch.zeros(x_dim, dtype=torch.float32),
                                        "w": torch.zeros(h, x_dim, dtype=torch.float32)},
        }
        return None, params

    return init_fun, spec


def initialize_network(x_dim, emb_dim, nbridges, rho_dim=0, nlayers=4, nn_arch="geffner",
                       fully_connected_units=None):
    """/root/reference/src/nn.py:21-39"""
    if nn_arch == "geffner":
        return initialize_mcd_network(x_dim, emb_dim, nbridges, rho_dim=rho_dim, nlayers=nlayers)
    if nn_arch == "dds":
        return initialize_pis_network(x_dim, fully_connected_units, rho_dim=rho_dim)
    # "dds_grad" is broken in the reference itself (undefined LinearConsInit, nn_dds.py:245)
    raise NotImplementedError(f"nn_arch {nn_arch!r} not implemented.")
