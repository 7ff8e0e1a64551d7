"""ResNet-50/101 backbone, parameter-compatible with radet/models/backbones/resnet.py:303-648
(Bottleneck style='pytorch', BN in eval mode, stem + stages <= frozen_stages frozen; frozen_stages=-1 trains the stem too).
Arithmetic runs in the HIP engine; this class owns parameters, init and the config surface."""
from torch import nn

from .builder import BACKBONES
from .shells import BNShell, ConvShell, kaiming_normal_


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = ConvShell(inplanes, planes, 1)
        self.bn1 = BNShell(planes)
        self.conv2 = ConvShell(planes, planes, 3, stride=stride, padding=1)   # style='pytorch': stride on the 3x3
        self.bn2 = BNShell(planes)
        self.conv3 = ConvShell(planes, planes * 4, 1)
        self.bn3 = BNShell(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(ConvShell(inplanes, planes * 4, 1, stride=stride), BNShell(planes * 4))
        else:
            self.downsample = None


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth, in_channels=3, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, style="pytorch",
                 zero_init_residual=True, **unsupported):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f"invalid depth {depth} for the MI355X ResNet (50 / 101 are built)")
        if num_stages != 4 or style != "pytorch" or in_channels != 3 or not norm_eval or norm_cfg.get("type") != "BN":
            raise NotImplementedError("only the configuration used by configs/bop/*.py is implemented: "
                                      "num_stages=4, style='pytorch', norm_eval=True, BN")
        for k, v in unsupported.items():
            if v not in (None, False, (False, False, False, False), 1, (1, 2, 2, 2), (1, 1, 1, 1), -1):
                raise NotImplementedError(f"ResNet option {k}={v!r} is outside the hot-path scope")
        self.depth, self.out_indices, self.frozen_stages = depth, tuple(out_indices), frozen_stages
        self.norm_eval, self.zero_init_residual = norm_eval, zero_init_residual
        self.conv1 = ConvShell(3, 64, 7, stride=2, padding=3)
        self.bn1 = BNShell(64)
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(self.arch_settings[depth]):
            planes = 64 * 2 ** i
            blocks = []
            for b in range(nb):
                blocks.append(Bottleneck(inplanes, planes, stride=2 if (b == 0 and i > 0) else 1, downsample=b == 0))
                inplanes = planes * 4
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self.init_weights()
        self._freeze_stages()

    def init_weights(self, pretrained=None):
        """resnet.py:590-620: kaiming-normal(fan_out) convs, BN gamma=1/beta=0, zero-init of every
        Bottleneck.bn3.weight. `pretrained` checkpoints are loaded by the caller via load_state_dict."""
        for m in self.modules():
            if isinstance(m, ConvShell):
                kaiming_normal_(m)
            elif isinstance(m, BNShell):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, f"layer{i}").parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    def forward(self, x):
        from ..runtime import standalone_forward
        return standalone_forward(self, "backbone", x)
