"""Model configurations: drop-in for reference TransUnet/vit_seg_configs.py (same factory names and values).

The reference builds `ml_collections.ConfigDict`s (a third-party package that is not a dependency here); `ConfigDict`
below provides the subset the model code uses: attribute access, item access, `.get`, nested dicts.
"""


class ConfigDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _vit(hidden, mlp, heads, layers, patch, classifier, **extra):
    c = ConfigDict(patches={"size": (patch, patch)}, hidden_size=hidden,
                   transformer=dict(mlp_dim=mlp, num_heads=heads, num_layers=layers, attention_dropout_rate=0.0,
                                    dropout_rate=0.1),
                   classifier=classifier, representation_size=None)
    for k, v in extra.items():
        c[k] = v
    return c


_SEG = dict(decoder_channels=(256, 128, 64, 16), n_classes=2, activation="softmax")
_CKPT = "./model/vit_checkpoint/imagenet21k/"


def get_b16_config():
    """ViT-B/16 (reference vit_seg_configs.py:3-24)."""
    return _vit(768, 3072, 12, 12, 16, "seg", resnet_pretrained_path=None, pretrained_path=_CKPT + "ViT-B_16.npz",
                patch_size=16, **_SEG)


def get_testing():
    """Minimal configuration (reference :27-40)."""
    return _vit(1, 1, 1, 1, 16, "token")


def _add_r50(c, pretrained_key, path):
    c.patches.grid = (16, 16)
    c.resnet = ConfigDict(num_layers=(3, 4, 9), width_factor=1)
    c.classifier = "seg"
    c[pretrained_key] = path
    c.decoder_channels = (256, 128, 64, 16)
    c.skip_channels = [512, 256, 64, 16]
    c.n_classes = 2
    c.activation = "softmax"
    return c


def get_r50_b16_config():
    """ResNet50 + ViT-B/16 (reference :42-58) -- the configuration the hot path uses."""
    c = _add_r50(get_b16_config(), "pretrained_path", _CKPT + "R50+ViT-B_16.npz")
    c.n_skip = 3
    return c


def get_b32_config():
    c = get_b16_config()
    c.patches.size = (32, 32)
    c.pretrained_path = _CKPT + "ViT-B_32.npz"
    return c


def get_l16_config():
    return _vit(1024, 4096, 16, 24, 16, "seg", resnet_pretrained_path=None, pretrained_path=_CKPT + "ViT-L_16.npz", **_SEG)


def get_r50_l16_config():
    return _add_r50(get_l16_config(), "resnet_pretrained_path", _CKPT + "R50+ViT-B_16.npz")


def get_l32_config():
    c = get_l16_config()
    c.patches.size = (32, 32)
    return c


def get_h14_config():
    return _vit(1280, 5120, 16, 32, 14, "token")
