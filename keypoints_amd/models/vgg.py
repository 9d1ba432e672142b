"""Layer-list factory with the reference's names (keypoints/models/vgg.py:16-70).

`make_layers` returns an ``nn.Sequential`` with the same module kinds at the same indices as the
reference builds (so ``state_dict`` keys / checkpoints are interchangeable); the modules are only
parameter holders -- `knn.Unit` executes them through the fused HIP kernels, never through
``nn.Sequential.forward``.
"""
import torch.nn as nn

# architecture tables: data, identical to vgg.py:48-70 by necessity (they define the checkpoints)
decoder_cfg = {
    'A': [512, 512, 'U', 256, 256, 'U', 256, 256, 'U', 128, 'U', 64, 'U'],
    'F': [512, 512, 'U', 256, 256, 'U', 256, 256, 'U', 128, 64],
    'VGG_PONG': [32, 'U', 16, 'U', 16],
    'VGG_PONG_TRIVIAL': [16, 16],
    'VGG_PONG_LAYERNECK': [32, 32, 16, 16],
    'VGG_PACMAN': [16, 32, 32, 16],
    'VGG_PACMAN_2': [64, 'U', 32, 32, 16],
}

vgg_cfg = {
    'A': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'B': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'D': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'E': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
    'F': [64, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512],
    'VGG_PONG': [16, 'M', 16, 'M', 32],
    'VGG_PONG_TRIVIAL': [16, 16],
    'VGG_PONG_LAYERNECK': [16, 32],
    'VGG_PACMAN': [16, 32, 32, 16],
    'VGG_PACMAN_2': [16, 32, 32, 'M', 64],
    'MAPPER': [8, 8],
}


def make_layers(cfg, batch_norm=True, extra_in_channels=0, nonlinearity=None, nonlinearity_kwargs=None,
                co_ord_conv=False):
    """cfg[0] = input width; int -> Pad(1)+Conv3x3(+BN)+act, 'M' -> MaxPool2d(2,2), 'U' -> bilinear x2."""
    if co_ord_conv:
        raise NotImplementedError('co_ord_conv is unused by every shipped config (and broken in the reference, '
                                  'vgg.py:2,31-32); no HIP kernel')
    act = nn.ReLU(inplace=True) if nonlinearity is None else nonlinearity(**(nonlinearity_kwargs or {}))
    width = cfg[0] + extra_in_channels
    mods = []
    for token in cfg[1:]:
        if token == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
        elif token == 'U':
            mods.append(nn.UpsamplingBilinear2d(scale_factor=2))
        elif token == 'L':
            raise NotImplementedError("'L' (activation capture) is unused by every shipped config")
        else:
            mods.append(nn.ReplicationPad2d(1))
            mods.append(nn.Conv2d(width, token, kernel_size=3))
            if batch_norm:
                mods.append(nn.BatchNorm2d(token))
            mods.append(act)   # one shared activation instance, like the reference
            width = token
    return nn.Sequential(*mods)
