"""Algorithmic FLOP accounting of a train step (conv MACs x 2) from the model's own layer plans --
used by bench.py for `step_tflops` (the per-kernel figures come from the C ABI's kp_prof_* instrumentation)."""
from .models.knn import Unit


def unit_macs(unit: Unit, h: int, w: int):
    """(total MACs of one forward pass of `unit` on an h x w input, MACs of its first layer, output h, output w)."""
    total, first = 0.0, None
    for step in unit.plan():
        conv = step['conv']
        taps = 9 if step['kind'] == 'conv3' else 1
        macs = float(h) * w * conv.in_channels * conv.out_channels * taps
        if first is None:
            first = macs
        total += macs
        tail = step.get('tail')
        if tail == 'M':
            h, w = h // 2, w // 2
        elif tail == 'U':
            h, w = 2 * h, 2 * w
    return total, first, h, w


def conv_flops_per_pair(net, height: int, width: int) -> float:
    """Forward + backward conv FLOPs of one image pair for a TransporterNet (feature / keypoint / decoder) or a
    KeyNet (encoder / keypoint / decoder).  Transporter: the source branch runs forward only (transporter.py:36-37),
    so forward = 2 (feature + keypoint) + decoder; backward = dgrad + wgrad of the three nets, minus the data-gradient
    of the two image-side first layers (nothing upstream needs it)."""
    enc = net.feature if hasattr(net, 'feature') else net.encoder
    e_tot, e_first, zh, zw = unit_macs(enc, height, width)
    k_tot, k_first, _, _ = unit_macs(net.keypoint, height, width)
    d_tot, _, _, _ = unit_macs(net.decoder, zh, zw)
    tot = e_tot + k_tot + d_tot
    fwd = 2 * (e_tot + k_tot) + d_tot if hasattr(net, 'feature') else tot
    bwd = 2 * tot - e_first - k_first
    return 2.0 * (fwd + bwd)
