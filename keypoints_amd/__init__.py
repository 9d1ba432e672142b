"""keypoints_amd -- MI355X-native (gfx950) Transporter / KeyNet train-step hot path behind the
`keypoints.models` surface of DuaneNielsen/keypoints.  See DESIGN.md / INTEGRATION.md."""
from .runtime import default_precision, set_default_precision, set_multi_stream, set_fuse_bn_input, set_fuse_bn_reduce, set_t_interchange, set_pair_calls, set_lazy_outputs, set_fuse_first_layer, set_fold_finalize  # noqa: F401
from . import models  # noqa: F401
from . import data_augments, tps  # noqa: F401
from .optim import FusedAdam, l2_reconstruction_loss  # noqa: F401
from . import inference  # noqa: F401
from . import amp  # noqa: F401
from . import graph  # noqa: F401


def install_as_keypoints():
    """Alias this package as `keypoints.models.*` so the reference's train scripts import it unchanged."""
    import sys
    import types
    pkg = sys.modules.setdefault('keypoints', types.ModuleType('keypoints'))
    pkg.models = models
    sys.modules['keypoints.models'] = models
    for name in ('functional', 'knn', 'vgg', 'transporter', 'keynet', 'autoencoder'):
        sys.modules[f'keypoints.models.{name}'] = getattr(models, name)
    # the train scripts import these two as top-level modules (transporter.py:6,11 with keypoints/ on sys.path)
    sys.modules.setdefault('data_augments', data_augments)
    sys.modules.setdefault('tps', tps)
