"""CPU-only tests of the host side: the C-ABI library exports what include/esr_hip.h declares, the product refuses to run without a
GPU (no CPU fallback), the reference-compatible module tree / state_dict / CEM construction / options / checkpoint loader behave
like the reference (golden data from tests/golden), and the Z-optimisation loop logic."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle.check_golden import load
from oracle.weights import seeded_uniform
from oracle.gen_golden import aniso_gaussian_kernel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    from esr_hip import _lib
    hdr = open(os.path.join(ROOT, 'include', 'esr_hip.h')).read()
    declared = set(re.findall(r'^(?:int|size_t|int64_t)\s+(esr_\w+)\s*\(', hdr, flags=re.M))
    assert declared, 'no entry points parsed from the header'
    h = _lib.load_library()
    for name in declared:
        assert hasattr(h, name), 'libesr_hip.so does not export %s' % name
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert h.esr_version() >= 100


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors instead of silently computing somewhere else."""
    from esr_hip import EsrError
    import CEM.CEMnet as C
    import models.modules.architecture as arch
    net = arch.RRDBNet(3, 3, 64, 1, upscale=4, num_latent_channels=0)
    with pytest.raises(EsrError):
        with torch.no_grad():
            net(torch.rand(1, 3, 8, 8))
    cem = C.CEMnet(C.Get_CEM_Conf(4)).WrapArchitecture_PyTorch(generated_image=None)
    with pytest.raises(EsrError):
        cem.DownscaleOP(torch.rand(1, 3, 16, 16))


TAP_CASES = [('cubic_x2', 2, None, None), ('cubic_x3', 3, None, None), ('cubic_x4', 4, None, None), ('cubic_x8', 8, None, None),
             ('blurry1.0_x4', 4, 'blurry_cubic_1.0', None), ('blurry2.0_x8', 8, 'blurry_cubic_2.0', None),
             ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1), ('aniso_x8', 8, aniso_gaussian_kernel(17, 4.0, 1.8, 0.6), 0.1)]


@pytest.mark.parametrize('name,sf,kernel,bound', TAP_CASES, ids=[c[0] for c in TAP_CASES])
def test_product_cem_construction_matches_reference(name, sf, kernel, bound):
    import CEM.CEMnet as C
    from CEM.imresize_CEM import imresize, calc_strides
    imresize.kernels = {}
    g = load('cem_taps.npz')
    conf = C.Get_CEM_Conf(sf)
    if bound:
        conf.lower_magnitude_bound = bound
    cem = C.CEMnet(conf, upscale_kernel=kernel)
    np.testing.assert_allclose(cem.ds_kernel, g[name + '/ds_kernel'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(cem.inv_hTh, g[name + '/inv_hTh'], rtol=0, atol=1e-13)
    pre, post = calc_strides(None, sf)
    ints = np.array([sf, cem.ds_kernel_invalidity_half_size_LR, cem.inv_hTh_invalidity_half_size, cem.invalidity_margins_LR,
                     cem.invalidity_margins_HR, pre[0], post[0]])
    assert (ints == g[name + '/ints']).all()


def test_imresize_kernel_cache_semantics():
    """A custom kernel replaces the per-scale cached kernel process-wide until 'reset_2_default' (reference imresize_CEM.py:23-43)."""
    from CEM.imresize_CEM import imresize
    imresize.kernels = {}
    default = imresize(None, [4, 4], return_upscale_kernel=True)
    custom = imresize(None, [4, 4], return_upscale_kernel=True, kernel=aniso_gaussian_kernel())
    assert custom.shape != default.shape
    again = imresize(None, [4, 4], return_upscale_kernel=True)           # still the custom one
    np.testing.assert_array_equal(again, custom)
    reset = imresize(None, [4, 4], return_upscale_kernel=True, kernel='reset_2_default')
    np.testing.assert_array_equal(reset, default)
    with pytest.raises(AssertionError):
        imresize(np.ones((8, 8)), scale_factor=[1.5])
    with pytest.raises(AssertionError):
        imresize(None, [4, 4], return_upscale_kernel=True, kernel=np.ones((5, 5)))     # does not sum to 1
    imresize.kernels = {}


def test_numpy_projections_match_reference():
    import CEM.CEMnet as C
    from CEM.imresize_CEM import imresize
    imresize.kernels = {}
    g = load('cem_forward.npz')
    cem = C.CEMnet(C.Get_CEM_Conf(4))
    rng = np.random.Generator(np.random.PCG64(23))
    hr_np = rng.random((48, 48, 3))
    lr_np = rng.random((12, 12, 3))
    np.testing.assert_allclose(cem.Project_2_ortho_2_NS(hr_np), g['numpy/Project_2_ortho_2_NS'], atol=1e-12)
    np.testing.assert_allclose(cem.DT_Satisfying_Upscale(lr_np), g['numpy/DT_Satisfying_Upscale'], atol=1e-12)
    np.testing.assert_allclose(cem.Enforce_DT_on_Image_Pair(lr_np, hr_np), g['numpy/Enforce_DT_on_Image_Pair'], atol=1e-12)
    np.testing.assert_allclose(imresize(hr_np, scale_factor=[0.25]), g['numpy/imresize_down4'], atol=1e-12)
    np.testing.assert_allclose(imresize(lr_np, scale_factor=[4]), g['numpy/imresize_up4'], atol=1e-12)


def _opt(nb=1, lat=0, cem=True, is_train=False):
    from options.options import dict_to_nonedict
    return dict_to_nonedict({
        'model': 'srragan', 'scale': 4, 'gpu_ids': None, 'range': [0, 1], 'is_train': is_train,
        'path': {'models': '/tmp/esr_models', 'log': '/tmp/esr_log', 'pretrain_model_G': None},
        'network_G': {'which_model_G': 'RRDB_net', 'CEM_arch': 1 if cem else 0, 'sigmoid_range_limit': 0, 'latent_input': 'all_layers' if lat else 'None',
                      'latent_input_domain': 'HR_downscaled', 'latent_channels': lat, 'norm_type': None, 'mode': 'CNA', 'nf': 64, 'nb': nb,
                      'in_nc': 3, 'out_nc': 3, 'gc': 32, 'scale': 4},
        'network_D': None, 'test': {'kernel': None}, 'datasets': {'train': {'patch_size': 128}},
        'train': {'pixel_weight': 1, 'pixel_criterion': 'l1', 'lr_G': 1e-4, 'pixel_domain': 'HR', 'grad_accumulation_steps_G': 1,
                  'lr_scheme': 'MultiStepLR', 'lr_steps': [1000], 'lr_gamma': 0.5}})


def test_state_dict_keys_and_parameter_counts_match_reference():
    from models import create_model
    g = load('c1_end_to_end.npz')
    m = create_model(_opt(nb=3))
    sd = m.netG.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['c1/keys']]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g['c1/key_shapes']]
    assert sum(p.numel() for n, p in m.netG.named_parameters() if 'Filter_OP' not in n) == 2309507      # RRDB-3
    m = create_model(_opt(nb=2, lat=3))
    assert list(m.netG.state_dict().keys()) == [str(k) for k in g['c1_lat3/keys']]
    assert [str(tuple(v.shape)) for v in m.netG.state_dict().values()] == [str(s) for s in g['c1_lat3/key_shapes']]
    # callers written against nn.DataParallel reach through `.module`
    assert m.netG.module is m.netG and hasattr(m.netG.module, 'DownscaleOP')
    # .train()/.eval() toggles the CEM pre-padding flag (reference CEMnet.py:313-315)
    m.netG.eval(); assert m.netG.pre_pad
    m.netG.train(); assert not m.netG.pre_pad


def test_rrdb23_parameter_count():
    import models.modules.architecture as arch
    n = sum(p.numel() for p in arch.RRDBNet(3, 3, 64, 23, upscale=4, num_latent_channels=0).parameters())
    assert n == 16697987
    n = sum(p.numel() for p in arch.RRDBNet(3, 3, 64, 23, upscale=4, latent_input='all_layers_HR_downscaled', num_latent_channels=3).parameters())
    assert n == 17060948              # explorable RRDB-23, 3 Z channels


def test_init_weights_skips_cem_filters():
    import CEM.CEMnet as C
    import models.modules.architecture as arch
    import models.networks as networks
    cem = C.CEMnet(C.Get_CEM_Conf(4))
    G = cem.WrapArchitecture_PyTorch(arch.RRDBNet(3, 3, 64, 1, upscale=4, num_latent_channels=0))
    taps = {k: v.clone() for k, v in G.state_dict().items() if 'Filter_OP' in k}
    networks.init_weights(G, init_type='kaiming', scale=0.1)
    for k, v in G.state_dict().items():
        if 'Filter_OP' in k:
            assert torch.equal(v, taps[k])
        elif k.endswith('bias'):
            assert float(v.abs().max()) == 0
    w = G.generated_image_model.model[0].weight
    assert abs(float(w.detach().std()) - 0.1 * np.sqrt(2.0 / 27)) < 0.02 * np.sqrt(2.0 / 27)
    assert not any(p.requires_grad for n, p in G.named_parameters() if 'Filter_OP' in n)


def test_options_parser(tmp_path):
    from options import options as option
    txt = '''{
  "name": "exp1" // experiment name
  , "model": "srragan", "scale": 4, "gpu_ids": [0]
  , "datasets": {"train": {"name": "DIV2K", "mode": "LRHR", "dataroot_HR": "DIV2K_train", "dataroot_LR": null, "n_workers": 2, "batch_size_per_GPU": 4, "patch_size": 208}}
  , "path": {"root": "%s", "datasets": "/data"}
  , "network_G": {"which_model_G": "RRDB_net", "CEM_arch": 1, "nf": 64, "nb": 23, "in_nc": 3, "out_nc": 3, "gc": 32}
  , "train": {"lr_G": {"PhaseInit": 1e-4, "PhaseGAN": 1e-5}, "gan_type": "wgan-gp", "resume": 0}
}''' % str(tmp_path)
    p = tmp_path / 'opt.json'
    p.write_text(txt)
    opt = option.parse(str(p), is_train=True, batch_size_multiplier=8)
    assert opt['train']['lr_G'] == 1e-5                                # PhaseGAN value chosen
    assert option.parse(str(p), is_train=True, initialization=True)['train']['lr_G'] == 1e-4
    assert opt['datasets']['train']['batch_size'] == 32 and opt['datasets']['train']['n_workers'] == 16
    assert opt['train']['grad_accumulation_steps_G'] == 1
    assert opt['network_G']['latent_input'] == 'None' and opt['network_G']['latent_channels'] == 0 and opt['network_G']['scale'] == 4
    assert opt['path']['models'] == os.path.join(str(tmp_path), 'experiments', 'exp1', 'models')
    assert opt['datasets']['train']['dataroot_HR'] == '/data/DIV2K_train' and opt['datasets']['train']['data_type'] == 'img'
    nd = option.dict_to_nonedict(opt)
    assert nd['train']['no_such_key'] is None and nd['network_G']['nb'] == 23
    option.save(opt)
    assert json.load(open(os.path.join(opt['path']['experiments_root'], 'options.json')))['name'] == 'exp1'
    t = option.parse(str(p), is_train=False)
    assert t['path']['results_root'].endswith(os.path.join('results', 'exp1'))


def test_checkpoint_positional_load_with_latent_zero_extension(tmp_path):
    """A plain ESRGAN-style checkpoint (different key names, no latent channels, no CEM prefix) initialises a CEM-wrapped explorable
    generator: positional match, 'generated_image_model.' prefix, zero weights for the new leading Z channels, CEM taps untouched."""
    import models.modules.architecture as arch
    from models import create_model
    torch.manual_seed(0)
    plain = arch.RRDBNet(3, 3, 64, 1, upscale=4, num_latent_channels=0)
    renamed = {k.replace('convs.', 'conv').replace('.0.weight', '.weight').replace('.0.bias', '.bias'): v for k, v in plain.state_dict().items()}
    path = tmp_path / 'esrgan.pth'
    torch.save(renamed, str(path))
    opt = _opt(nb=1, lat=3)
    opt['path']['pretrain_model_G'] = str(path)
    m = create_model(opt)
    sd = m.netG.state_dict()
    src = list(plain.state_dict().items())
    dst = [(k, v) for k, v in sd.items() if 'Filter_OP' not in k]
    assert len(src) == len(dst)
    n_ext = 0
    for (ks, vs), (kd, vd) in zip(src, dst):
        assert kd.startswith('generated_image_model.')
        if vd.shape == vs.shape:
            assert torch.equal(vd, vs), kd
        else:
            extra = vd.shape[1] - vs.shape[1]
            assert extra == 3 and float(vd[:, :extra].abs().max()) == 0 and torch.equal(vd[:, extra:], vs), kd
            n_ext += 1
    assert n_ext == 1 + 15 + 1 + 2           # fea, 15 RDB convs, LR_conv, HR_conv0/1 get Z channels; the upconvs do not
    # save -> load round trip in the reference's format
    m2 = create_model(_opt(nb=1, lat=3, is_train=True))
    m2.save_dir = str(tmp_path / 'models')
    saved = m2.save(7)
    ck = torch.load(saved)
    assert set(ck.keys()) == {'model_state_dict', 'optimizer_state_dict'}
    assert list(ck['model_state_dict'].keys()) == list(m2.netG.state_dict().keys())


class _ToyG(torch.nn.Module):
    """Stand-in generator for the Z-loop logic on CPU: output depends smoothly on Z."""
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(0.5))

    def forward(self, x):
        z, lr = x[:, :-3], x[:, -3:]
        z = z.reshape(z.size(0), 1, 4 * z.size(2), 4 * z.size(3))
        up = torch.nn.functional.interpolate(lr, scale_factor=4, mode='nearest')
        return up * 0.5 + 0.25 + self.w * 0.3 * z * torch.sin(torch.arange(up.shape[-1]).float())[None, None, None, :]


class _ToyModel:
    def __init__(self):
        self.netG = _ToyG()
        self.device = torch.device('cpu')
        self.num_latent_channels = 1
        self.opt = {'scale': 4}
    Output_Batch = lambda self, within_0_1: torch.clamp(self.fake_H, 0, 1) if within_0_1 else self.fake_H

    def feed_data(self, data, need_GT=True, **kw):
        z = data['Z']
        lr = data['LR']
        self.model_input = torch.cat([z.contiguous().view(z.size(0), 16, lr.size(2), lr.size(3)), lr], 1)

    def test(self, prevent_grads_calc=True, **kw):
        self.fake_H = self.netG(self.model_input)


def test_z_optimizer_loop_logic():
    from Z_optimization import Z_optimizer, Optimizable_Z
    torch.manual_seed(0)
    model = _ToyModel()
    lr = torch.rand(1, 3, 4, 4)
    model.feed_data({'LR': lr.expand(3, -1, -1, -1), 'Z': torch.zeros(3, 1, 16, 16)})
    model.test()
    zo = Z_optimizer(objective='max_STD', Z_size=[16, 16], model=model, Z_range=1, max_iters=15, data={'LR': lr}, initial_LR=0.1, batch_size=3)
    std0 = float(zo.initial_STD.mean())
    Z = zo.optimize()
    assert Z.shape == (3, 1, 16, 16) and float(Z.abs().max()) <= 1.0          # tanh-bounded
    assert all(p.requires_grad for p in model.netG.parameters())               # grad flags restored
    model.feed_data({'LR': lr.expand(3, -1, -1, -1), 'Z': Z}); model.test()
    assert float(torch.std(model.Output_Batch(True), dim=(1, 2, 3)).mean()) > std0 + 1e-3
    assert zo.loss_values == sorted(zo.loss_values, reverse=True) or min(zo.loss_values) == zo.loss_values[-1]
    with pytest.raises(NotImplementedError):
        Z_optimizer(objective='scribble', Z_size=[16, 16], model=model, Z_range=1, max_iters=1, initial_LR=0.1)
    oz = Optimizable_Z([2, 1, 4, 4], Z_range=0.5, device='cpu')
    oz.Z.data.fill_(100.)
    assert float(oz().max()) <= 0.5


def test_cabi_argument_validation_without_a_gpu():
    """Error behaviour of the C-ABI (include/esr_hip.h: 0 on success, negative ESR_E_* otherwise, never throws): bad descriptors are
    rejected by the host-side checks before anything touches a device, so this runs on the CPU-only box."""
    import ctypes as C
    from esr_hip import _lib
    h = _lib.load_library()
    E_ARG, E_UNSUPPORTED = -1, -2
    assert h.esr_conv3x3(None, None) == E_ARG
    d = _lib.Conv3x3Desc()                                   # all-zero descriptor: no input view, no weights
    assert h.esr_conv3x3(C.byref(d), None) == E_ARG
    assert h.esr_pack_conv_weights(None, 32, 64, None, 8, None, 1, 0, 1, 1.0, None, None) == E_ARG
    assert h.esr_pack_nchw(None, 0, 1, 3, 8, 8, 0, 3, 0, 1, None, None) < 0
    assert h.esr_cem_downscale(None, 1, 3, 8, 8, 4, 1, None, 17, None, 0, None, None) == E_ARG
    assert h.esr_cem_lrfilter(None, 1, 3, 8, 8, None, 27, None, None) == E_ARG
    assert h.esr_cem_upscale(None, None, 1, 3, 8, 8, 4, 1, None, 17, None, 0, 0, 0.0, None, None, None) == E_ARG
    w = _lib.WgradDesc()
    assert h.esr_conv3x3_wgrad(C.byref(w), None) == E_ARG
    assert h.esr_conv3x3_wgrad_workspace_floats(C.byref(w)) == E_ARG
    assert h.esr_conv3x3_wgrad_batch(None, 0, None, 0, None) == E_ARG
    assert h.esr_pack_batch_run(None, 0, 0, None) == E_ARG
    # sizes are pure host arithmetic
    assert h.esr_conv_wpack_bytes(24, 64, 1) == 12 * 9 * 2 * 2 * 1024 and h.esr_conv_wpack_bytes(8, 32, 0) == 4 * 9 * 1 * 1 * 1024
    w.B, w.H, w.W, w.cout, w.cin_main = 2, 16, 40, 64, 192   # 2 images x (2 x 2) tiles of 8x32 pixels; 6 input tiles x 2 output tiles
    n = h.esr_conv3x3_wgrad_workspace_floats(C.byref(w))
    assert n == 12 * 8 * 9 * 1024 + 2 * 8 * 32               # every tile its own slice (8), 9 x 32 x 32 partial block + bias partials


def test_unsupported_generator_configurations_fail_loudly():
    """What the fused kernels do not implement must raise at construction, never run something else silently."""
    import models.modules.architecture as arch
    base = dict(in_nc=3, out_nc=3, nf=64, nb=1, upscale=4, num_latent_channels=0)
    arch.RRDBNet(**base)
    for bad in (dict(norm_type='batch'), dict(act_type='relu'), dict(mode='NAC'), dict(nf=96), dict(nf=36), dict(nf=24), dict(nf=32, upsample_mode='pixelshuffle'),
                dict(upsample_mode='nearest')):
        with pytest.raises(NotImplementedError):
            arch.RRDBNet(**dict(base, **bad))
    assert arch.RRDBNet(**dict(base, nf=32)).nf == 32          # 16, 32, 48, 64 (round 6)
    net = arch.RRDBNet(**dict(base, upsample_mode='pixelshuffle'))          # constructible (state_dict parity), not executable
    with pytest.raises(Exception):
        net(torch.zeros(1, 3, 8, 8))


def test_bench_launches_its_own_ranks_for_several_gpus():
    """`python bench.py --gpus N` (how the driver starts the N = 1 run) must work for N > 1 too: when no launcher set WORLD_SIZE it
    re-executes itself under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1; under a launcher it runs as a rank."""
    import bench
    assert not bench.needs_self_launch(1, {}) and bench.needs_self_launch(8, {}) and not bench.needs_self_launch(8, {'WORLD_SIZE': '8'})
    cmd = bench.launch_command(8, ['--gpus', '8', '--steps', '5', '--warmup', '2'], port=29511)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '5', '--warmup', '2'] and cmd[-7].endswith('bench.py')
    a = bench.parse_args(['--gpus', '4'])
    assert a.gpus == 4 and a.workload == 'c2' and a.precision is None
    assert bench.free_port() > 1024
    # RCCL ('nccl') is the backend of every multi-GPU run; gloo only under the test-only switch that puts several ranks on one GPU
    assert bench.select_backend({}) == 'nccl' and bench.select_backend({'ESR_BENCH_SHARE_GPU': '0'}) == 'nccl'
    assert bench.select_backend({'ESR_BENCH_SHARE_GPU': '1'}) == 'gloo'
    src = open(bench.__file__).read()
    assert "init_process_group(backend='nccl', device_id=dev)" in src and "'HSA_ENABLE_IPC_MODE_LEGACY', '0'" in src


@pytest.mark.parametrize('code', ['SVDinNormedOut_structure_tensor', 'structure_tensor'])
def test_filter_loss_matches_reference(code):
    """The product's FilterLoss (structure-tensor latent-control loss of the explorable training, reference loss.py:27-209) against values the
    reference's class produced over three consecutive calls (fixture F10: the percentile history accumulates), and its gradient."""
    from models.modules.loss import FilterLoss
    g = load('filter_loss.npz')
    fl = FilterLoss(latent_channels=code)
    assert fl.num_channels == 3
    for call in range(3):
        sr = seeded_uniform((4, 3, 24, 20), 1000 + call).requires_grad_(True)
        hr = seeded_uniform((4, 3, 24, 20), 1010 + call)
        z = seeded_uniform((4, 3, 1, 1), 1020 + call, -1.0, 1.0) * torch.ones(4, 3, 24, 20)
        loss = fl({'SR': sr, 'HR': hr, 'Z': z})
        np.testing.assert_allclose(loss.detach().numpy(), g['%s/call%d' % (code, call)], rtol=1e-5, atol=1e-7)
    loss.sum().backward()
    np.testing.assert_allclose(sr.grad.numpy(), g[code + '/dSR'], rtol=1e-4, atol=1e-8)


def test_recorder_inserts_host_steps_at_earlier_positions():
    """act.Recorder.position() / insert_host(): the two-stream backward places its side-stream launches where a group's last dy was written, after
    the whole pass has been recorded — the segment a position falls into is split, positions at a segment boundary or behind a host step insert
    without splitting, and inserting last-to-first keeps the earlier positions valid."""
    import torch
    from esr_hip import act as A
    rec = A.Recorder({'x': torch.zeros(4)})
    pos = [rec.position()]
    for i in range(5):
        rec.emit(100 + i, None)
        pos.append(rec.position())
    rec.host('h0')
    pos.append(rec.position())
    for i in range(3):
        rec.emit(200 + i, None)
        pos.append(rec.position())
    assert pos == [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (2, 0), (2, 1), (2, 2), (2, 3)]
    for p, name in reversed([(pos[0], 'a'), (pos[2], 'b'), (pos[5], 'c'), (pos[6], 'd'), (pos[8], 'e'), (pos[9], 'f')]):
        rec.insert_host(p, name)
    flat = []
    for kind, payload in rec.items:
        flat += [payload] if kind == 'host' else [op for op, _, _ in payload]
    assert flat == ['a', 100, 101, 'b', 102, 103, 104, 'c', 'h0', 'd', 200, 201, 'e', 202, 'f'], flat
    assert all(kind == 'host' or payload for kind, payload in rec.items)        # no empty segments


def test_adjoint_tables_of_rank_one_taps_factor_per_axis():
    """esr_cem_adjoint_sep's premise (esr_hip/autograd.py): for taps = outer(tv, th) every one of the nine prefix / plain / suffix tables of the 2-D
    adjoint is the outer product of the per-axis tables."""
    import torch
    from esr_hip import autograd as AG
    g = torch.Generator().manual_seed(5)
    tv, th = torch.rand(17, generator=g, dtype=torch.float64) - 0.3, torch.rand(17, generator=g, dtype=torch.float64) - 0.3
    full = AG.tap_tables(torch.outer(tv, th).float())
    v, h = AG.tap_tables_1d(tv.float()), AG.tap_tables_1d(th.float())
    assert full.shape == (3, 3, 17, 17) and v.shape == h.shape == (3, 17)
    for ry in range(3):
        for rx in range(3):
            ref = torch.outer(v[ry].double(), h[rx].double())
            assert float((full[ry, rx].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
