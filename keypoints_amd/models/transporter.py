"""TransporterNet / make with the reference's surface (keypoints/models/transporter.py:8-138)."""
import torch
import torch.nn as nn

from . import functional as MF
from . import knn, vgg
from ..unit_engine import shared_packs


class TransporterNet(knn.Container):
    def __init__(self, feature_cnn, keypoint_cnn, key2map, decoder, init_weights=True, combine_method='loop'):
        super().__init__()
        self.feature = feature_cnn
        self.keypoint = keypoint_cnn
        self.ssm = knn.SpatialLogSoftmax()
        self.key2map = key2map
        self.decoder = decoder
        self.combine_method = combine_method
        if init_weights:
            self._initialize_weights()

    def extract(self, x):
        phi = self.feature(x)
        heatmap = self.keypoint(x)
        k, p = self.ssm(heatmap, probs=True)
        m = self.key2map(k, height=phi.size(2), width=phi.size(3))
        return phi, heatmap, k, p, m

    def forward(self, xs, xt):
        with shared_packs():   # both branches run on the same weights: pack them once
            with torch.no_grad():   # source branch carries no gradient (transporter.py:36-37)
                phi_xs, _, _, _, m_xs = self.extract(xs)
            phi_xt, heatmap_xt, k_xt, p_xt, m_xt = self.extract(xt)

        mode = self.combine_method
        if mode == 'max':
            phi, mask_xs, mask_xt = MF.transport_max(phi_xs, phi_xt, m_xs, m_xt)
        elif mode == 'sum_and_clamp':
            mask_xs = m_xs.sum(dim=1, keepdim=True).clamp(0.0, 1.0)
            mask_xt = m_xt.sum(dim=1, keepdim=True).clamp(0.0, 1.0)
            phi = phi_xs * (1 - mask_xs) * (1 - mask_xt) + phi_xt * mask_xt
        elif mode == 'pretrained_network':
            mask_xs, mask_xt = m_xs, m_xt
            phi = phi_xs * (1 - mask_xs) * (1 - mask_xt) + phi_xt * mask_xt
        elif mode == 'loop':
            phi = phi_xs
            for i in range(k_xt.size(1)):
                mask_xs, mask_xt = m_xs[:, i:i + 1], m_xt[:, i:i + 1]
                phi = phi * (1 - mask_xs) * (1 - mask_xt) + phi_xt * mask_xt
        else:
            raise ValueError(f'unknown combine_method {mode!r}')

        x_t = self.decoder(phi)
        return x_t, phi, k_xt, m_xt, p_xt, heatmap_xt, mask_xs, mask_xt

    def load(self, directory, map_device=None):
        self.feature.load(directory + '/encoder', map_device=map_device)
        self.keypoint.load(directory + '/keypoint', map_device=map_device)
        self.decoder.load(directory + '/decoder', map_device=map_device)

    def load_from_autoencoder(self, directory):
        self._initialize_weights()
        self.feature.load(directory + '/encoder', out_block=False)
        self.keypoint.load(directory + '/encoder', in_block=True, core=True, out_block=False)
        self.decoder.load(directory + '/decoder', in_block=False, core=True, out_block=True)

    def save(self, directory):
        self.feature.save(directory + '/encoder')
        self.keypoint.save(directory + '/keypoint')
        self.decoder.save(directory + '/decoder')


class MaskMaker(nn.Module):
    """Keypoints -> point map -> pretrained mapper network ('pretrained_network' mode, transporter.py:83-92)."""

    def __init__(self, map_f):
        super().__init__()
        self.map = map_f

    def __call__(self, k, height, width):
        with torch.no_grad():
            return self.map(MF.point_map(k, height, width))


class TransporterMap(knn.Container):
    def __init__(self, mapper, init_weights=True):
        super().__init__()
        self.map = mapper
        if init_weights:
            self._initialize_weights()

    def forward(self, x):
        return self.map(x)

    def load(self, directory):
        self.map.load(directory + '/map')

    def save(self, directory):
        self.map.save(directory + '/map')


def make(type, in_channels, z_channels, keypoints, combine_mode='max', load=None, transfer_load=None,
         map_device=None):
    """Same wiring and construction order as transporter.py:112-138 (so one torch seed gives the
    same initial weights as the reference): LeakyReLU encoder / keypoint cores, ReLU decoder core."""
    leaky = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True})
    encoder = knn.Unit(in_channels, z_channels, vgg.make_layers(vgg.vgg_cfg[type], **leaky))
    decoder = knn.Unit(z_channels, in_channels, vgg.make_layers(vgg.decoder_cfg[type]))
    keypoint = knn.Unit(in_channels, keypoints, vgg.make_layers(vgg.vgg_cfg[type], **leaky))
    net = TransporterNet(encoder, keypoint, knn.GaussianLike(sigma=0.1), decoder, init_weights=True,
                         combine_method=combine_mode)
    if load is not None:
        net.load(load, map_device)
    if transfer_load is not None:
        net.load_from_autoencoder(transfer_load)
    return net
