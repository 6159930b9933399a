"""Pose regression network: stacked range-image pair -> (translation[B,3], quaternion[B,4] as x,y,z,w).
Mirror of the reference's OdometryModel (src/models/model.py:16-116) with identical parameter names."""
import torch

from . import model_parts, resnet_modified


def _act(name):
    return torch.nn.ReLU() if name == "relu" else torch.nn.Tanh()


def _mlp(name, sizes):
    layers = []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        layers += [_act(name), torch.nn.Linear(fan_in, fan_out)]
    return torch.nn.Sequential(*layers)


class OdometryModel(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.device = config["device"]
        self.config = config
        act = config["activation_fct"]
        if act not in ("relu", "tanh"):
            raise Exception('The specified activation function must be either "relu" or "tanh".')
        self.pre_feature_extraction = config["pre_feature_extraction"]
        in_channels, n_pre = 8, 5
        if self.pre_feature_extraction:                  # model.py:31-48: per-image conv tower before stacking
            tower = []
            for i in range(n_pre):
                c_in = in_channels // 2 if i == 0 else i * in_channels
                tower += [model_parts.CircularPad(padding=(1, 1, 0, 0)),
                          torch.nn.Conv2d(c_in, (i + 1) * in_channels, kernel_size=3, padding=(1, 0), bias=False),
                          torch.nn.ReLU(inplace=True) if act == "relu" else torch.nn.Tanh()]
            self.feature_extractor = torch.nn.Sequential(*tower)
        self.resnet = resnet_modified.ResNetModified(
            in_channels=in_channels if not self.pre_feature_extraction else 2 * n_pre * in_channels,
            num_outputs=config["resnet_outputs"], use_dropout=config["use_dropout"], layers=config["layers"],
            factor_fewer_resnet_channels=config["factor_fewer_resnet_channels"], activation_fct=act,
            impl=config.get("cnn_impl", "auto"))
        n_feat = config["resnet_outputs"]
        if config["use_single_mlp_at_output"]:           # model.py:59-72
            self.fully_connected_rot_trans = _mlp(act, [n_feat, 512, 512, 256, 64, 3 + 4])
        else:                                            # model.py:74-83
            self.fully_connected_rotation = _mlp(act, [n_feat, 100, 4])
            self.fully_connected_translation = _mlp(act, [n_feat, 100, 3])
        self.geometry_handler = model_parts.GeometryHandler(config=config)

    def forward_features(self, image_1, image_2):
        if self.pre_feature_extraction:
            x = torch.cat((self.feature_extractor(image_1), self.feature_extractor(image_2)), dim=1)
        else:
            x = torch.cat((image_1, image_2), dim=1)
        return self.resnet(x)

    def forward_stacked(self, stacked):
        """Same as forward() for an already channel-stacked ``[B,8,H,W]`` pair (no concatenation copy).  When the CNN runs on the HIP
        stem + trunk, fc and the two heads run as ``model_parts.FusedHeads`` (csrc/heads.hip: 3 + 4 launches instead of ~45)."""
        if self._fused_heads_ok(stacked):
            pooled, _ = self.resnet.pooled_features(stacked)
            if pooled is not None:
                act = 2 if self.config["activation_fct"] == "relu" else 1
                fr, ft = self.fully_connected_rotation, self.fully_connected_translation
                with torch.autocast("cuda", enabled=False):
                    translation, rotation = model_parts.FusedHeads.apply(
                        pooled.float(), act, self.resnet.fc.weight, self.resnet.fc.bias, fr[1].weight, fr[1].bias, fr[3].weight, fr[3].bias,
                        ft[1].weight, ft[1].bias, ft[3].weight, ft[3].bias)
                return translation, rotation
        feat = self.resnet(stacked)[-1]
        return self._heads(feat)

    def _fused_heads_ok(self, x):
        """fc + heads as one fused Function: CUDA input, the default two-head architecture, no active dropout, a batch of at most 16."""
        return (x.is_cuda and not self.config["use_single_mlp_at_output"] and x.shape[0] <= 16 and self.config.get("fused_heads", True)
                and not (self.resnet.use_dropout and self.training) and self.config.get("cnn_impl", "auto") != "modules")

    def _heads(self, feat):
        if self.config["use_single_mlp_at_output"]:
            y = self.fully_connected_rot_trans(feat)
            rotation, translation = y[:, :4], y[:, 4:]
        else:
            rotation = self.fully_connected_rotation(feat)
            translation = self.fully_connected_translation(feat)
        # model.py:114: one norm over the WHOLE batch of quaternions (the per-row normalisation happens later,
        # inside quaternion -> R); kept because it is part of the gradient path
        rotation = rotation / torch.norm(rotation)
        return translation, rotation

    def forward(self, image_1, image_2=None):
        """``forward(image_1, image_2)`` as the reference; ``forward(stacked)`` with a ``[B,8,H,W]`` tensor skips the
        concatenation copy (only without the per-image feature tower)."""
        if image_2 is None:
            if self.pre_feature_extraction:
                image_1, image_2 = image_1[:, :4], image_1[:, 4:]
            else:
                return self.forward_stacked(image_1)
        return self._heads(self.forward_features(image_1=image_1, image_2=image_2)[-1])
