"""CPU: host-side logic of the drop-in surface and the C-ABI library (load + exported symbols only; no
compute call is made without a GPU)."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'kp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kp_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from keypoints_amd import _lib
    names = header_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert set(names) == bound, (set(names) ^ bound)
    assert _lib.lib().kp_version() >= 100


def test_no_oracle_import_in_product():
    """The product path must never route through the oracle / a CPU fallback."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'keypoints_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no oracle', ''), f


def test_cpu_tensors_fail_loudly():
    from keypoints_amd.models import functional as MF, transporter
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3)
    with pytest.raises(RuntimeError, match='HIP device'):
        net(torch.zeros(2, 1, 32, 32), torch.zeros(2, 1, 32, 32))
    with pytest.raises(RuntimeError, match='HIP device'):
        MF.spacial_logsoftmax(torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError, match='HIP device'):
        MF.gaussian_like_function(torch.zeros(1, 2, 2), 8, 8)


def test_same_seed_same_init_and_keys_as_reference(golden):
    """transporter.make draws the reference's RNG stream: same torch seed -> identical initial weights,
    and the state_dict keys / parameter count are the reference's (checkpoint compatibility)."""
    from keypoints_amd.models import transporter
    z = golden('init_check.npz')
    for tag, args in (('pong', ('VGG_PONG_LAYERNECK', 1, 16, 3)), ('F', ('F', 3, 64, 10))):
        torch.manual_seed(1234)
        net = transporter.make(*args)
        assert list(net.state_dict().keys()) == list(z[f'{tag}.keys'])
        assert sum(p.numel() for p in net.parameters()) == int(z[f'{tag}.nparams'])
        for n, p in net.named_parameters():
            assert abs(p.detach().double().sum().item() - float(z[f'{tag}.sum.{n}'])) < 1e-9
            assert abs(p.detach().double().abs().sum().item() - float(z[f'{tag}.abs.{n}'])) < 1e-9
    assert sum(p.numel() for p in transporter.make('F', 3, 64, 10).parameters()) == 24468813


def test_plan_of_model_F():
    from keypoints_amd.models import transporter
    net = transporter.make('F', 3, 64, 10)
    enc = net.feature.plan()
    assert [s['kind'] for s in enc] == ['conv3'] * 8 + ['head']
    assert [s.get('tail') for s in enc[:-1]] == [None, 'M', None, 'M', None, 'M', None, None]
    assert all(abs(s['slope'] - 0.01) < 1e-9 for s in enc)
    dec = net.decoder.plan()
    assert [s.get('tail') for s in dec[:-1]] == [None, 'U', None, 'U', None, 'U', None, None]
    assert dec[0]['slope'] == 0.01 and all(s['slope'] == 0.0 for s in dec[1:-1]) and dec[-1]['slope'] == 0.01
    assert [s['conv'].out_channels for s in dec] == [512, 512, 256, 256, 256, 256, 128, 64, 3]


def test_keynet_make_and_decoder_width():
    from keypoints_amd.models import keynet
    args = types.SimpleNamespace(model_type='F', model_in_channels=3, model_z_channels=64, model_keypoints=10,
                                 load=None, transfer_load=None)
    net = keynet.make(args)
    assert net.decoder.in_block[1].in_channels == 74
    assert sum(p.numel() for p in net.parameters()) == 24514893


def test_unsupported_structures_raise():
    from keypoints_amd.models import knn, vgg
    with pytest.raises(NotImplementedError):
        vgg.make_layers(vgg.vgg_cfg['F'], co_ord_conv=True)
    core = nn.Sequential(nn.ReplicationPad2d(1), nn.Conv2d(8, 8, 3), nn.BatchNorm2d(8), nn.Tanh())
    with pytest.raises(NotImplementedError):
        knn.Unit(3, 4, core).plan()


def test_checkpoint_layout_roundtrip(tmp_path):
    """save()/load() write the reference's files: <dir>/{encoder,keypoint,decoder}/{in_block,core,out_block}.mdl
    holding nn.Sequential state_dicts (knn.py:143-167, transporter.py:66-80)."""
    from keypoints_amd.models import transporter
    torch.manual_seed(1)
    a = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3)
    a.save(str(tmp_path / 'ck'))
    for unit in ('encoder', 'keypoint', 'decoder'):
        for blk in ('in_block', 'core', 'out_block'):
            assert (tmp_path / 'ck' / unit / f'{blk}.mdl').exists()
    sd = torch.load(str(tmp_path / 'ck' / 'encoder' / 'in_block.mdl'))
    assert list(sd.keys()) == ['1.weight', '1.bias', '2.weight', '2.bias', '2.running_mean', '2.running_var',
                               '2.num_batches_tracked']
    torch.manual_seed(2)
    b = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3, load=str(tmp_path / 'ck'))
    for (n1, p1), (_, p2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p1, p2), n1
    c = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3)
    c.load_from_autoencoder(str(tmp_path / 'ck'))   # re-uses encoder in_block+core for the keypoint net
    assert torch.equal(c.keypoint.in_block[1].weight, a.feature.in_block[1].weight)


def test_install_as_keypoints_alias():
    import sys
    import keypoints_amd as ka
    saved = {k: v for k, v in sys.modules.items() if k == 'keypoints' or k.startswith('keypoints.')}
    try:
        ka.install_as_keypoints()
        from keypoints.models import transporter as T   # noqa: the reference's import line
        assert T is ka.models.transporter
    finally:
        for k in [k for k in sys.modules if k == 'keypoints' or k.startswith('keypoints.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_grad_arena_views_and_attach():
    from keypoints_amd.models import transporter
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3)
    u = net.decoder
    arena = u.grad_arena()
    params = list(u.parameters())
    assert arena.flat.numel() == sum(p.numel() for p in params)
    arena.flat.fill_(3.0)
    arena.attach(params)                      # nothing attached yet -> whole arena zeroed
    assert float(arena.flat.abs().max()) == 0 and all(p.grad is arena.views[id(p)] for p in params)
    arena.views[id(params[0])].fill_(2.0)
    params[1].grad = None
    arena.attach(params)                      # only the detached slot is re-zeroed
    assert float(params[0].grad.min()) == 2.0 and float(params[1].grad.abs().max()) == 0


def test_flop_accounting_matches_survey():
    """keypoints_amd.flops (what bench.py reports as FLOP/pair) reproduces SURVEY.md 8(d): 235.23 GFLOP per pair and
    train step for Transporter F 128x128 K=10, 196.5 for KeyNet, and agrees with the oracle's independent accounting."""
    import types
    from keypoints_amd.flops import conv_flops_per_pair
    from keypoints_amd.models import keynet, transporter
    from oracle import kp_oracle as O
    net = transporter.make('F', 3, 64, 10)
    f = conv_flops_per_pair(net, 128, 128)
    assert abs(f - 235.23e9) < 0.01e9
    assert f == O.conv_flops_per_pair(O.transporter_plans('F', 3, 64, 10), 128, 128)
    kn = keynet.make(types.SimpleNamespace(model_type='F', model_in_channels=3, model_z_channels=64, model_keypoints=10))
    assert conv_flops_per_pair(kn, 128, 128) == O.conv_flops_per_pair(O.keynet_plans('F', 3, 64, 10), 128, 128, transporter=False)


def test_side_stream_helpers_are_inert_without_a_device():
    """runtime.fork / join_side_streams touch no HIP API when nothing lives on a device: the CPU-side uses of
    the host engine (plan building, reducer bookkeeping over gloo) never initialise HIP through them."""
    import torch
    from keypoints_amd import runtime
    x = torch.zeros(2, 3)
    order = []
    with runtime.fork('chain', x) as side:
        with side:
            order.append('side')
            side.outputs(x)
        order.append('main')
    runtime.join_side_streams()
    assert order == ['side', 'main'] and not runtime._used and not runtime._held
    runtime.set_multi_stream(False)
    assert not runtime.multi_stream() and not runtime.multi_stream('wgrad')
    runtime.set_multi_stream(True)
    assert runtime.multi_stream('chain')


def test_overlap_report_on_a_synthetic_trace(tmp_path, capsys):
    """tools/overlap_report.py (the concurrency evidence of DESIGN 3.8) on a hand-made kernel trace: two steps delimited by
    loss kernels, one MFMA kernel overlapping an HBM pass for half of its duration."""
    import csv, importlib.util, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = [('l2_loss_k', 0, 10), ('l2_loss_k', 1000, 1010), ('l2_loss_k', 2000, 2010), ('conv3x3_patch_k<128, true, false>(x)', 2100, 2500),
            ('v4::bn_bwd_fast_k<0, 1, 1>(y)', 2300, 2700), ('l2_loss_k', 3000, 3010)]
    f = tmp_path / 't_kernel_trace.csv'
    with open(f, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        for r in rows:
            w.writerow(r)
    spec = importlib.util.spec_from_file_location('overlap_report', os.path.join(root, 'tools', 'overlap_report.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    sys.argv = ['overlap_report.py', str(f)]
    try:
        mod.main()
    finally:
        sys.argv = old
    out = capsys.readouterr().out
    assert '1 steps' in out                       # window: second half of the loss kernels = [2000, 3000]
    lines = {ln.split()[0]: float(ln.split()[-2]) for ln in out.splitlines() if ln.split() and ln.split()[0] in ('mfma', 'hbm', 'mfma+hbm', 'idle')}
    assert lines['mfma+hbm'] == 20.0 and lines['mfma'] == 20.0 and abs(lines['hbm'] - 21.0) < 1e-6 and abs(lines['idle'] - 39.0) < 1e-6
    sys.argv = old


def test_freeze_gc_parks_the_live_objects():
    """runtime.freeze_gc(): everything alive moves to the permanent generation (a later full collection does not walk it),
    the collector stays enabled."""
    import gc
    from keypoints_amd import runtime
    junk = [[i] for i in range(1000)]
    before = gc.get_freeze_count()
    try:
        runtime.freeze_gc()
        assert gc.isenabled()
        assert gc.get_freeze_count() >= before + len(junk)
    finally:
        gc.unfreeze()


def test_loss_scaler_refuses_an_optimizer_that_would_ignore_it():
    """amp.initialize / GraphedTrainStep attach a loss scaler; on torch.optim.Adam (what transporter.py:47 builds) nothing would
    ever divide the 2^16-scaled gradients again: TypeError instead of silent mis-training.  FusedAdam refuses added groups."""
    import torch
    import keypoints_amd as ka
    from keypoints_amd import amp
    lin = torch.nn.Linear(2, 2)
    with pytest.raises(TypeError):
        amp.initialize(lin, torch.optim.Adam(lin.parameters(), lr=1e-4), opt_level='O2')
    amp._require_scaling_optimizer(ka.FusedAdam(lin.parameters()), 'test')        # accepted
    opt = ka.FusedAdam(lin.parameters())
    opt.param_groups.append(dict(params=[], lr=1e-3, betas=(0.9, 0.999), eps=1e-8))
    with pytest.raises(RuntimeError, match='param_groups'):
        opt.step()


def test_round4_support_queries_answer_without_a_device():
    """The *_supported / row-count queries behind TransporterNet's route selection (16-bit NHWC interchange, pair calls) are pure
    functions of the shape: BASELINE config 2 (Pong 84x84, 16 feature channels, K = 4, 256 pairs) takes both routes in bf16 / fp16,
    config 4 (model F, 64 feature channels at 16x16) and fp32 storage take neither -- and TransporterNet asks the same questions."""
    import torch
    from keypoints_amd import _lib
    from keypoints_amd.models import transporter
    from keypoints_amd.unit_engine import head_dims, pair_ok
    L = _lib.lib()
    BF, FP16, F32 = _lib.KP_BF16, _lib.KP_F16, _lib.KP_F32
    for code in (BF, FP16):
        assert L.kp_head_t_supported(256, 84, 84, 32, 16, code) == 1
        assert L.kp_keypoint_bottleneck_t_supported(256, 16, 4, 84, 84, code) == 1
        assert L.kp_conv_pair_supported(512, 84, 84, 1, 16, 1, code) == 1          # first layer, both frames
        assert L.kp_conv_pair_supported(512, 84, 84, 16, 32, 0, code) == 1
        assert L.kp_head_pair_supported(512, 84, 84, 32, 16, 1, code) == 1 and L.kp_head_pair_supported(512, 84, 84, 32, 4, 0, code) == 1
    assert L.kp_head_t_supported(256, 84, 84, 32, 16, F32) == 0 and L.kp_keypoint_bottleneck_t_supported(256, 16, 4, 84, 84, F32) == 0
    assert L.kp_head_t_supported(128, 16, 16, 512, 64, FP16) == 0 and L.kp_keypoint_bottleneck_t_supported(128, 64, 10, 16, 16, FP16) == 0
    assert L.kp_conv_pair_supported(256, 128, 128, 3, 64, 1, FP16) == 0 and L.kp_conv_pair_supported(511, 84, 84, 1, 16, 1, BF) == 0
    # a pair launch writes the rows of the separate call, per half
    # (first-layer row kernel: one row per workgroup of <= 768; band kernel: one row per workgroup of 512)
    assert L.kp_conv_fwd_stats_rows(256, 86, 86, 1, 16, 3, BF) == 768 and L.kp_conv_fwd_stats_rows(256, 86, 86, 16, 32, 3, BF) == 512
    torch.manual_seed(0)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4, combine_mode='max').set_precision('bf16')
    assert head_dims(net.feature, 84, 84) == (84, 84, 32, 16) and head_dims(net.keypoint, 84, 84) == (84, 84, 32, 4)
    assert pair_ok(net.feature, 256, 1, 84, 84, True) and pair_ok(net.keypoint, 256, 1, 84, 84, False)
    assert not pair_ok(net.feature.eval(), 256, 1, 84, 84, True)
    netF = transporter.make('F', 3, 64, 10).set_precision('fp16')
    assert head_dims(netF.feature, 128, 128)[:2] == (16, 16) and not pair_ok(netF.feature, 128, 3, 128, 128, False)
    assert netF._t_route(torch.zeros(2, 3, 128, 128)) is None            # (a CPU tensor: no route; the forward itself raises loudly)


def test_package_reads_no_environment_variable_outside_probe():
    """Measurement knobs live in keypoints_amd/_probe.py (called by bench.py / tools, never by the package): no other module of the
    product path reads os.environ / getenv."""
    import pathlib
    import re
    import keypoints_amd
    root = pathlib.Path(keypoints_amd.__file__).parent
    bad = []
    for f in list(root.glob('*.py')) + list((root / 'models').glob('*.py')):
        if f.name in ('_probe.py', 'hip_env.py'):   # (hip_env WRITES one variable of the HIP runtime for the application; see its docstring)
            continue
        for i, ln in enumerate(f.read_text().splitlines(), 1):
            code = ln.split('#', 1)[0]
            if re.search(r'os\.environ|getenv|_os\.environ', code):
                bad.append(f'{f.name}:{i}: {ln.strip()}')
    for f in list((root / 'csrc').glob('*.hip')) + list((root / 'csrc').glob('*.h')) + list((root / 'csrc').glob('*.inc')):
        for i, ln in enumerate(f.read_text().splitlines(), 1):
            if re.search(r'\bgetenv\s*\(', ln.split('//', 1)[0]):
                bad.append(f'csrc/{f.name}:{i}: {ln.strip()}')     # (library knobs are kp_config keys: include/kp_hip.h)
    assert not bad, bad
    # hip_env is reachable only through an explicit call (runtime.configure_hip_runtime imports it lazily): importing the package must
    # not touch the environment
    import subprocess
    import sys
    code = ("import os; before = dict(os.environ); import keypoints_amd, keypoints_amd.runtime, keypoints_amd.models.transporter; "
            "assert dict(os.environ) == before; import sys; assert 'keypoints_amd.hip_env' not in sys.modules; "
            "assert keypoints_amd.runtime.configure_hip_runtime() is True and os.environ['HSA_KERNARG_POOL_SIZE'] == str(64 << 20)")
    env = {k: v for k, v in __import__('os').environ.items() if k != 'HSA_KERNARG_POOL_SIZE'}
    subprocess.run([sys.executable, '-c', code], check=True, cwd=str(root.parent), env=env)


def test_probe_apply_env_turns_variables_into_setters():
    import keypoints_amd as ka
    from keypoints_amd import _probe, runtime, unit_engine
    try:
        used = _probe.apply_env({'KP_NO_PAIR': '1', 'KP_NO_ROLES': 'wgrad,aux', 'KP_NO_PREPACK': '1', 'KP_SMALL_SIDE': '1'})
        assert set(used) == {'KP_NO_PAIR', 'KP_NO_ROLES=wgrad,aux', 'KP_NO_PREPACK', 'KP_SMALL_SIDE'}
        assert not runtime.pair_calls() and not runtime.multi_stream('wgrad') and runtime.multi_stream('chain')
        assert unit_engine.PREPACK is False and unit_engine.SMALL_INLINE is False
        assert _probe.apply_env({}) == []
    finally:
        runtime.set_pair_calls(True); runtime._disabled_roles.clear()
        unit_engine.PREPACK = True; unit_engine.SMALL_INLINE = True


def test_the_16bit_contract_is_frozen():
    """oracle.TOL16 and the rules around it (per-network factor, the emulation anchor) as frozen in round 5 and re-frozen in round 6: any edit of a constant
    changes this hash and has to be argued in DESIGN.md section 4 together with the new hash -- the contract does not follow the
    measurements."""
    import hashlib
    import json
    from oracle import kp_oracle as O
    blob = json.dumps(dict(TOL16=O.TOL16, per_network=O.TOL16_PER_NETWORK, vs_emulation=O.TOL16_VS_EMULATION, floor=O.TOL16_EMULATION_FLOOR,
                           survey=dict(xt_max=O.SURVEY_XT_MAX, xt_near=O.SURVEY_XT_NEAR, quorum=O.SURVEY_KEYPOINT_QUORUM)), sort_keys=True)
    # round 6 re-freeze: per-network factor 1.5 -> 1.25 (worst network over 26 trajectories:
    # 1.20 x its clause), + the premise constants (SURVEY's as-written x_t clause and keypoint quorum).  Round 5's hash: b6d8cbaf...eae87
    assert hashlib.sha256(blob.encode()).hexdigest() == 'e42a2a8088937529b154c0195f4d1b1ce086fcb25762ed7ffeb762e331b3a985', blob
