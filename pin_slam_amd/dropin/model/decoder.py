"""Drop-in for the reference's ``model.decoder.Decoder`` (model/decoder.py:14): same
constructor, ``layers.{i}.weight/bias`` + ``lout.weight/bias`` state_dict keys, ``sdf`` /
``mlp`` methods.  All parameters are views of ONE flat fp32 buffer in state_dict order --
exactly the layout libpinhip's fused kernels read (pin_field.dec) and the flat buffer the
optimiser / RCCL all-reduce work on."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops


class Decoder(nn.Module):
    def __init__(self, config, hidden_dim, hidden_level, out_dim, is_time_conditioned=False):
        super().__init__()
        if is_time_conditioned:
            raise NotImplementedError("time-conditioned decoder is unused by the reference (decoder.py:40)")
        if config.mlp_leaky_relu or not config.mlp_bias_on:
            raise NotImplementedError("libpinhip decoders are Linear+bias+ReLU (config.py:139-140 defaults)")
        self.out_dim = out_dim
        self.hidden_dim, self.hidden_level = int(hidden_dim), int(hidden_level)
        self.use_leaky_relu = False
        if config.use_gaussian_pe:
            position_dim = config.pos_input_dim + 2 * config.pos_encoding_band
        else:
            position_dim = config.pos_input_dim * (2 * config.pos_encoding_band + 1)
        input_dim = config.feature_dim + position_dim
        if input_dim != 11:
            raise NotImplementedError("libpinhip is built for feature_dim 8 + 3 position inputs")
        layers = []
        for i in range(hidden_level):
            layers.append(nn.Linear(input_dim if i == 0 else hidden_dim, hidden_dim, True))
        self.layers = nn.ModuleList(layers)
        self.lout = nn.Linear(hidden_dim, out_dim, True)
        self.sdf_scale = 1.0
        if config.main_loss_type == "bce":
            self.sdf_scale = config.logistic_gaussian_ratio * config.sigma_sigmoid_m
        self.to(config.device)
        self._flat = None
        self.flat_params()

    def _ordered(self):
        return [p for _, p in self.named_parameters()]  # state_dict order: layers.i.weight, .bias, lout.*

    def flat_params(self) -> torch.Tensor:
        """The flat parameter buffer; parameters are (re)pointed at it if something (e.g.
        .to(), load of a pickled module) detached them."""
        ps = self._ordered()
        flat = self._flat
        ok = flat is not None and flat.device == ps[0].device
        if ok:
            off = 0
            for p in ps:
                if p.data.data_ptr() != flat.data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.cat([p.data.reshape(-1).float() for p in ps]).contiguous()
            off = 0
            for p in ps:
                p.data = flat[off:off + p.numel()].view_as(p.data)
                off += p.numel()
            self._flat = flat
        return self._flat

    def _field(self):
        return ops.FieldState(feats=self.flat_params(), dec=self.flat_params(), k=1, hidden=self.hidden_dim,
                              levels=self.hidden_level, weighted_first=True, sdf_scale=self.sdf_scale,
                              out_dim=self.out_dim)

    def mlp(self, features):
        if self.out_dim > 3:  # the semantic decoder's raw outputs (sem_class_count + 1 heads)
            return self._sem(features, raw=True)
        if self.out_dim != 1:
            raise NotImplementedError("raw multi-head outputs are only reachable through regress_color (3 heads)")
        shape = features.shape[:-1]
        f = features.detach().reshape(-1, 11).to(torch.float32).contiguous()
        fs = self._field()
        fs.sdf_scale = 1.0
        return ops.decoder_sdf(fs, f).reshape(*shape, 1)

    def sdf(self, features):
        return self.mlp(features).squeeze(1) * self.sdf_scale

    def occupancy(self, features):
        return torch.sigmoid(self.sdf(features) / -self.sdf_scale)

    def regress_color(self, features):
        """sigmoid(mlp(features)) with 3 heads (decoder.py:112-114); [..., 11] -> [..., 3]."""
        if self.out_dim != 3:
            raise NotImplementedError("libpinhip colour decoders have 3 heads (color_channel = 3)")
        shape = features.shape[:-1]
        f = features.detach().reshape(-1, 11).to(torch.float32).contiguous()
        return ops.decoder_color(self._field(), f).reshape(*shape, 3)

    def _sem(self, features, raw: bool):
        if not 2 <= self.out_dim <= 32:
            raise NotImplementedError("libpinhip semantic decoders have 2..32 heads (sem_class_count + 1)")
        shape = features.shape[:-1]
        f = features.detach().reshape(-1, 11).to(torch.float32).contiguous()
        return ops.decoder_sem(self._field(), f, self.out_dim, raw=raw).reshape(*shape, self.out_dim)

    def sem_label_prob(self, features):
        """F.log_softmax(mlp(features), dim=-1) (decoder.py:100-103); [..., 11] -> [..., heads]."""
        return self._sem(features, raw=False)

    def sem_label(self, features):
        """decoder.py:105-107."""
        return torch.argmax(self.sem_label_prob(features), dim=1)
