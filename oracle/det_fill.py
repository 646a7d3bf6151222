"""Deterministic, construction-order-independent parameter fill.   *** TEST INFRASTRUCTURE ***

Large models (the TCM codec: ~20 M parameters) cannot ship their weights in a fixture.  Instead the fixture's generator and
the tests both fill a model's state_dict with values that depend only on each tensor's KEY and SHAPE (a generator seeded by
crc32(key)), so the reference-side model and the mirror get identical parameters without sharing an RNG stream.  Scales are
chosen per kind of tensor so that a deep random network stays in a sane numeric range."""
import zlib

import torch


def det_fill_(state_dict) -> None:
    with torch.no_grad():
        for key, v in state_dict.items():
            if not v.is_floating_point():
                continue
            leaf = key.rsplit(".", 1)[-1]
            g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
            r = lambda *shape: torch.randn(*shape, generator=g)
            if leaf in ("pedestal", "bound", "target", "scale_bound", "scale_table"):
                continue                                                        # fixed buffers of the re-parametrisations / entropy models
            if leaf == "beta":                                                  # GDN: stored value = sqrt(effective + pedestal)
                new = torch.sqrt(1.0 + 0.2 * torch.rand(v.shape, generator=g) + 2.0 ** -36)
            elif leaf == "gamma" and ".igdn." in key:                            # inverse GDN multiplies: keep its gain near 1
                new = torch.sqrt(0.01 * torch.eye(v.shape[0]) + 0.0005 * torch.rand(v.shape, generator=g) + 2.0 ** -36)
            elif leaf == "gamma":
                new = torch.sqrt(0.1 * torch.eye(v.shape[0]) + 0.02 * torch.rand(v.shape, generator=g) + 2.0 ** -36)
            elif leaf == "running_mean":                                         # BatchNorm statistics of the GroupMix aggregator
                new = 0.1 * r(*v.shape)
            elif leaf == "running_var":
                new = 1.0 + 0.2 * torch.rand(v.shape, generator=g)
            elif leaf == "quantiles":
                new = v.clone(); new[:, 0, 1] = 0.5 * r(v.shape[0])
            elif leaf.startswith("_matrix"):
                new = v + 0.2 * r(*v.shape)
            elif leaf.startswith("_factor"):
                new = 0.3 * r(*v.shape)
            elif leaf.startswith("_bias"):
                new = 0.3 * r(*v.shape)
            elif leaf == "relative_position_params":
                new = 0.5 * r(*v.shape)
            elif leaf == "bias":
                new = 0.1 * r(*v.shape)
            elif leaf == "weight" and v.dim() == 1:                              # LayerNorm gain
                new = 1.0 + 0.1 * r(*v.shape)
            elif leaf == "weight":
                fan_in = v[0].numel()
                new = r(*v.shape) * 0.6 * (1.0 / fan_in) ** 0.5
            else:
                raise KeyError(f"det_fill_: no rule for {key} {tuple(v.shape)}")
            v.copy_(new.to(v.dtype))
