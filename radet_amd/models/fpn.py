"""FPN neck, parameter-compatible with radet/models/necks/fpn.py:12-221 for the configuration the
BOP configs use (start_level=1, add_extra_convs='on_output', num_outs=5, no norm, no activation)."""
from torch import nn

from .builder import NECKS
from .shells import ConvModuleShell, ConvShell, xavier_uniform_


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode="nearest")):
        super().__init__()
        if (list(in_channels) != [256, 512, 1024, 2048] or out_channels != 256 or num_outs != 5 or start_level != 1
                or end_level != -1 or add_extra_convs != "on_output" or relu_before_extra_convs or norm_cfg is not None
                or act_cfg is not None or conv_cfg is not None or upsample_cfg.get("mode") != "nearest"):
            raise NotImplementedError("FPN: only the r50/r101 BOP configuration is implemented on MI355X "
                                      "(in_channels=[256,512,1024,2048], out=256, start_level=1, "
                                      "add_extra_convs='on_output', num_outs=5)")
        self.in_channels, self.out_channels, self.num_outs, self.start_level = list(in_channels), out_channels, num_outs, 1
        self.lateral_convs = nn.ModuleList(ConvModuleShell(c, out_channels, 1) for c in in_channels[1:])
        self.fpn_convs = nn.ModuleList(ConvModuleShell(out_channels, out_channels, 3, stride=1 if i < 3 else 2, padding=1)
                                       for i in range(5))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, ConvShell):
                xavier_uniform_(m)

    def forward(self, inputs):
        from ..runtime import standalone_forward
        return standalone_forward(self, "neck", inputs)
