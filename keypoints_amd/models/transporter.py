"""TransporterNet / make with the reference's surface (keypoints/models/transporter.py:8-138)."""
import torch
import torch.nn as nn

from . import functional as MF
from . import knn, vgg
from .. import _lib, runtime
from ..unit_engine import _first_cp, head_dims, prepack, shared_packs


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

    def _fused_bottleneck_ok(self, phi):
        return (self.combine_method == 'max' and type(self.ssm) is knn.SpatialLogSoftmax
                and type(self.key2map) is knn.GaussianLike
                and MF.bottleneck_fusable(phi.size(2), phi.size(3), phi.size(1)))

    def _t_route(self, xt):
        """The storage type's dtype code if feature encoder -> 'max' bottleneck -> decoder can exchange 16-bit NHWC ("T") tensors
        (include/kp_hip.h: kp_head_fwd_t, kp_keypoint_bottleneck_fwd_t; the Pong-class shapes), else None: the fp32 NCHW tensors of
        the reference's Unit interface are then used between them.  Same arithmetic either way; the T route rounds the encoder
        head's output once more (to the storage type the decoder's first layer would round the blend to anyway)."""
        if not (runtime.t_interchange() and torch.is_tensor(xt) and xt.is_cuda and xt.dim() == 4 and self.combine_method == 'max'
                and type(self.ssm) is knn.SpatialLogSoftmax and type(self.key2map) is knn.GaussianLike
                and all(isinstance(u, knn.Unit) for u in (self.feature, self.keypoint, self.decoder))
                and self.feature.precision == self.decoder.precision and self.feature.precision in ('bf16', 'fp16')):
            return None
        try:
            fplan, dplan = self.feature.plan(), self.decoder.plan()
        except NotImplementedError:
            return None
        N, _, H, W = xt.shape
        code = runtime.precision_dtype(self.feature.precision)[1]
        Hh, Wh, Ch, Kf = head_dims(self.feature, H, W)
        Kp = self.keypoint.out_block[0].out_channels
        L = _lib.lib()
        if (dplan[0]['kind'] == 'conv3' and dplan[0]['conv'].in_channels == Kf and _first_cp(dplan, Kf, self.decoder.precision) == Kf
                and L.kp_head_t_supported(N, Hh, Wh, Ch, Kf, code) == 1
                and L.kp_keypoint_bottleneck_t_supported(N, Kf, Kp, Hh, Wh, code) == 1):
            return code
        return None

    def forward(self, xs, xt):
        with shared_packs():   # both branches run on the same weights: pack them once
            return self._forward(xs, xt)

    def _forward(self, xs, xt):
        # ... and all three Units in ONE launch, up front: the decoder's pack is then off the critical path behind the bottleneck
        if torch.is_tensor(xs) and all(isinstance(u, knn.Unit) for u in (self.feature, self.keypoint, self.decoder)):
            prepack([(self.feature, xs.size(1)), (self.keypoint, xs.size(1)),
                     (self.decoder, self.feature.out_block[0].out_channels)])
        # the keypoint encoder's two calls are independent of the feature encoder's: they go to a side stream (and,
        # the autograd engine running a node's backward on the stream of its forward, so does their backward)
        with runtime.fork('chain', xs, xt) as side:
            units = isinstance(self.keypoint, knn.Unit) and isinstance(self.feature, knn.Unit)
            with side:
                if units:   # source frame under no_grad + target frame: one batch where the pair kernels exist (Unit.forward_pair)
                    heatmap_xs, heatmap_xt = self.keypoint.forward_pair(xs, xt)
                else:
                    with torch.no_grad():
                        heatmap_xs = self.keypoint(xs)
                    heatmap_xt = self.keypoint(xt)
                side.outputs(heatmap_xs, heatmap_xt)
            tcode = self._t_route(xt)
            if units:
                phi_xs, phi_xt = self.feature.forward_pair(xs, xt, out_t=tcode is not None)
            else:
                with torch.no_grad():   # source branch carries no gradient (transporter.py:36-37)
                    phi_xs = self.feature(xs)
                phi_xt = self.feature(xt)

        if tcode is not None:
            # the bottleneck reads the heads' 16-bit NHWC output and writes the decoder's padded input; the fp32 NCHW copy of the
            # blend that forward() also returns is made beside the decoder, on a side stream
            a, k_xt, p_xt, m_xt, mask_xs, mask_xt = MF.keypoint_bottleneck_t(
                heatmap_xs, heatmap_xt, phi_xs, phi_xt, tcode, sigma=self.key2map.sigma)
            if runtime.lazy_outputs():
                # ... made when the caller first looks at it (MF.LazyNCHW): inside a train step nobody does (transporter.py:80-89)
                phi = MF.LazyNCHW(a, 1, tcode)
                x_t = self.decoder.forward_t(a, in_t=True)
                return x_t, phi, k_xt, m_xt, p_xt, heatmap_xt, mask_xs, mask_xt
            with runtime.fork('aux', a) as side:
                with side:
                    phi = MF.padded_t_to_nchw(a, 1, tcode)
                    side.outputs(phi)
                x_t = self.decoder.forward_t(a, in_t=True)
            return x_t, phi, k_xt, m_xt, p_xt, heatmap_xt, mask_xs, mask_xt

        if self._fused_bottleneck_ok(phi_xt):
            # ssm + key2map of both branches and the 'max' blend: one kernel forward, one backward
            phi, k_xt, p_xt, m_xt, mask_xs, mask_xt = MF.keypoint_bottleneck(
                heatmap_xs, heatmap_xt, phi_xs, phi_xt, sigma=self.key2map.sigma)
            x_t = self.decoder(phi)
            return x_t, phi, k_xt, m_xt, p_xt, heatmap_xt, mask_xs, mask_xt

        with torch.no_grad():
            k_xs, _ = self.ssm(heatmap_xs, probs=True)
            m_xs = self.key2map(k_xs, height=phi_xs.size(2), width=phi_xs.size(3))
        k_xt, p_xt = self.ssm(heatmap_xt, probs=True)
        m_xt = self.key2map(k_xt, height=phi_xt.size(2), width=phi_xt.size(3))

        # 'max' on shapes the fused kernel does not take, and the three other combine modes: one blend kernel each
        # way (kp_transport_max_* / kp_transport_*), forward and backward
        phi, mask_xs, mask_xt = MF.transport(phi_xs, phi_xt, m_xs, m_xt, self.combine_method)
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
        self._fused_unit, self._fused_key = None, None

    def _fused(self):
        """The mapper as ONE kernel (keypoints_amd.inference.FusedUnit: eval-mode BatchNorm folded into the weights, fp32 arithmetic)
        where it has that form -- a frozen, eval-mode `TransporterMap` over a small-channel Unit on the device, which is what
        transporter.make(combine_mode='pretrained_network') builds.  The layer chain of the 8-channel MAPPER runs on the generic VALU
        kernels (~0.2 ms per layer and call at 256 x 84 x 84); fused, the whole mask costs one launch.  The folded weights are a snapshot
        keyed on the parameters' / buffers' versions: a changed mapper is folded again.  None: run the Unit itself."""
        unit = getattr(self.map, 'map', None)
        if not isinstance(unit, knn.Unit) or unit.training or torch.is_grad_enabled():
            return None
        tensors = list(unit.parameters()) + list(unit.buffers())
        if not tensors or not tensors[0].is_cuda:
            return None
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._fused_key:
            from ..inference import FusedUnit
            try:
                self._fused_unit = FusedUnit(unit)
            except NotImplementedError:
                self._fused_unit = None
            self._fused_key = key
        return self._fused_unit

    def __call__(self, k, height, width):
        with torch.no_grad():
            pm = MF.point_map(k, height, width)
            fused = self._fused()
            return fused(pm) if fused is not None else self.map(pm)


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


def make_mapper(keypoints, model_type='MAPPER', load=None):
    """The keypoint-map -> mask network that mapper.py:50-52 trains: Unit(keypoints -> 1) over vgg_cfg[model_type],
    wrapped as TransporterMap (checkpoint layout `<dir>/map/*.mdl`)."""
    mapper = TransporterMap(knn.Unit(keypoints, 1, vgg.make_layers(vgg.vgg_cfg[model_type])), init_weights=load is None)
    if load is not None:
        mapper.load(load)
    return mapper


def make(type, in_channels, z_channels, keypoints, combine_mode='max', load=None, transfer_load=None,
         map_device=None, mapper_load=None, mapper_type='MAPPER'):
    """Same wiring and construction order as transporter.py:112-138 (so one torch seed gives the
    same initial weights as the reference): LeakyReLU encoder / keypoint cores, ReLU decoder core.

    combine_mode='pretrained_network' wires what the reference defines but never connects (MaskMaker,
    transporter.py:83-92 + the mapper of mapper.py:50-52): key2map = MaskMaker(pretrained mapper) turns the keypoints
    into ONE mask channel through point_map and the frozen mapper network; `mapper_load` names its checkpoint
    directory (None: a freshly initialised mapper, for tests).  Built after the three Units, so the seed-for-seed
    weight reproduction of the reference's construction order is unaffected."""
    leaky = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True})
    encoder = knn.Unit(in_channels, z_channels, vgg.make_layers(vgg.vgg_cfg[type], **leaky))
    decoder = knn.Unit(z_channels, in_channels, vgg.make_layers(vgg.decoder_cfg[type]))
    keypoint = knn.Unit(in_channels, keypoints, vgg.make_layers(vgg.vgg_cfg[type], **leaky))
    net = TransporterNet(encoder, keypoint, knn.GaussianLike(sigma=0.1), decoder, init_weights=True,
                         combine_method=combine_mode)
    if combine_mode == 'pretrained_network':
        mapper = make_mapper(keypoints, mapper_type, mapper_load)
        for p in mapper.parameters():
            p.requires_grad_(False)
        net.key2map = MaskMaker(mapper.eval())
    if load is not None:
        net.load(load, map_device)
    if transfer_load is not None:
        net.load_from_autoencoder(transfer_load)
    return net
