"""MobileInvertedResidualBlock (reference ofa/imagenet_codebase/networks/proxyless_nets.py:36-72):
`mobile_inverted_conv(x) + shortcut(x)`.  The ImageNet ProxylessNAS networks are out of scope."""
from ...layers import IdentityLayer, ZeroLayer, set_layer_from_config
from ...utils import MyModule


class MobileInvertedResidualBlock(MyModule):

    def __init__(self, mobile_inverted_conv, shortcut):
        super().__init__()
        self.mobile_inverted_conv = mobile_inverted_conv
        self.shortcut = shortcut

    def forward(self, x):
        conv, skip = self.mobile_inverted_conv, self.shortcut
        if conv is None or isinstance(conv, ZeroLayer):
            return x
        if skip is None or isinstance(skip, ZeroLayer):
            return conv(x)
        if isinstance(skip, IdentityLayer) and not skip._modules and getattr(conv, "accepts_residual", False):
            return conv(x, residual=x)      # shortcut add fused into the block's last BN pass
        return conv(x) + skip(x)

    def stackable(self, x):
        """this block can be one item of ops.FusedMBStackFn: elastic MB layer on the composite path + identity shortcut"""
        conv, skip = self.mobile_inverted_conv, self.shortcut
        return (isinstance(skip, IdentityLayer) and not skip._modules and getattr(conv, "accepts_residual", False)
                and hasattr(conv, "composite_eligible") and conv.composite_eligible(x))

    @property
    def module_str(self):
        return "(%s, %s)" % (
            self.mobile_inverted_conv.module_str if self.mobile_inverted_conv is not None else None,
            self.shortcut.module_str if self.shortcut is not None else None)

    @property
    def config(self):
        return {
            "name": MobileInvertedResidualBlock.__name__,
            "mobile_inverted_conv": None if self.mobile_inverted_conv is None else self.mobile_inverted_conv.config,
            "shortcut": None if self.shortcut is None else self.shortcut.config,
        }

    @staticmethod
    def build_from_config(config):
        return MobileInvertedResidualBlock(set_layer_from_config(config["mobile_inverted_conv"]),
                                           set_layer_from_config(config["shortcut"]))
