"""AutoEncoder with the reference's surface (keypoints/models/autoencoder.py:4-24): encoder Unit -> decoder Unit, used for
the pre-training that `--transfer_load` / `load_from_autoencoder` (transporter.py:72-76, keynet.py:38-42) start from.
Both Units run on the HIP engine (unit_engine.py); checkpoints are `<dir>/encoder/*.mdl` and `<dir>/decoder/*.mdl`."""
from torch import nn

from . import knn, vgg


class AutoEncoder(knn.Container):
    def __init__(self, encoder, decoder, init_weights=True):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        if init_weights:
            self._initialize_weights()

    def forward(self, x):
        z = self.encoder(x)
        x = self.decoder(z)
        return z, x

    def load(self, root_path):
        self.encoder.load(root_path + '/encoder')
        self.decoder.load(root_path + '/decoder')

    def save(self, root_path):
        self.encoder.save(root_path + '/encoder')
        self.decoder.save(root_path + '/decoder')


def make(model_type, in_channels, z_channels, load=None):
    """The model autoencode.py:59-69 builds: LeakyReLU cores for BOTH Units (unlike transporter.make's ReLU decoder)."""
    leaky = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True})
    encoder = knn.Unit(in_channels, z_channels, vgg.make_layers(vgg.vgg_cfg[model_type], **leaky))
    decoder = knn.Unit(z_channels, in_channels, vgg.make_layers(vgg.decoder_cfg[model_type], **leaky))
    net = AutoEncoder(encoder, decoder, init_weights=load is None)
    if load is not None:
        net.load(load)
    return net
