"""KeyNet (Jakab et al. conditional image generation) with the reference's surface
(keypoints/models/keynet.py:6-66).  `make` works here (the reference's raises NameError, keynet.py:3,51)."""
import torch
import torch.nn as nn

from .. import runtime
from ..unit_engine import prepack, shared_packs
from . import knn, vgg


class KeyNet(knn.Container):
    def __init__(self, encoder, keypoint, key2map, decoder, init_weights=True):
        super().__init__()
        self.encoder = encoder
        self.keypoint = keypoint
        self.ssm = knn.SpatialLogSoftmax()
        self.key2map = key2map
        self.decoder = decoder
        if init_weights:
            self._initialize_weights()

    def forward(self, x, x_t):
        with shared_packs():   # the three Units' weight packs in ONE launch, up front (the decoder's off the critical path)
            if torch.is_tensor(x) and all(isinstance(u, knn.Unit) for u in (self.encoder, self.keypoint, self.decoder)):
                prepack([(self.encoder, x.size(1)), (self.keypoint, x_t.size(1)),
                         (self.decoder, self.encoder.out_block[0].out_channels + self.keypoint.out_block[0].out_channels)])
            with runtime.fork('chain', x_t) as side:   # two independent encoders: the keypoint one on a side stream
                with side:
                    heatmap = self.keypoint(x_t)
                    side.outputs(heatmap)
                z = self.encoder(x)
            k, p = self.ssm(heatmap, probs=True)
            m = self.key2map(k, height=z.size(2), width=z.size(3))
            if isinstance(self.decoder, knn.Unit) and z.dim() == 4 and m.dim() == 4:
                x_t = self.decoder.forward_cat(z, m)   # decoder(torch.cat((z, m), dim=1)), keynet.py:29, without the concatenated tensor
            else:
                x_t = self.decoder(torch.cat((z, m), dim=1))
        return x_t, z, k, m, p, heatmap

    def load(self, directory):
        self.encoder.load(directory + '/encoder')
        self.keypoint.load(directory + '/keypoint')
        self.decoder.load(directory + '/decoder')

    def load_from_autoencoder(self, directory):
        self._initialize_weights()
        self.encoder.load(directory + '/encoder', out_block=False)
        self.keypoint.load(directory + '/encoder', in_block=True, core=True, out_block=False)
        self.decoder.load(directory + '/decoder', in_block=False, core=True, out_block=True)

    def save(self, directory):
        self.encoder.save(directory + '/encoder')
        self.keypoint.save(directory + '/keypoint')
        self.decoder.save(directory + '/decoder')


def make(args):
    """args: model_type, model_in_channels, model_z_channels, model_keypoints, load, transfer_load."""
    leaky = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True})
    encoder = knn.Unit(args.model_in_channels, args.model_z_channels,
                       vgg.make_layers(vgg.vgg_cfg[args.model_type], **leaky))
    decoder = knn.Unit(args.model_z_channels + args.model_keypoints, args.model_in_channels,
                       vgg.make_layers(vgg.decoder_cfg[args.model_type]))
    keypoint = knn.Unit(args.model_in_channels, args.model_keypoints,
                        vgg.make_layers(vgg.vgg_cfg[args.model_type], **leaky))
    net = KeyNet(encoder, keypoint, knn.GaussianLike(sigma=0.1), decoder, init_weights=True)
    if getattr(args, 'load', None) is not None:
        net.load(args.load)
    if getattr(args, 'transfer_load', None) is not None:
        net.load_from_autoencoder(args.transfer_load)
    return net
