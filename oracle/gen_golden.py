"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by importing the
read-only reference at /root/reference (build container only; see
oracle/_refshim.py for the shims).  Run:  python -m oracle.gen_golden [F1 F2 ...]

Fixture ids follow SURVEY.md §8(c).  Inputs are regenerated on the test side
from the same seeds / formula weights (oracle/weights.py), so the files hold
only expected outputs (and small inputs where convenient).
"""
import os
import sys
import numpy as np
import torch

from oracle import _refshim
from oracle.weights import fill_formula_weights, seeded_uniform

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def aniso_gaussian_kernel(size=13, s1=2.2, s2=0.9, theta=0.6):
    """An anisotropic Gaussian *downscaling* kernel (sums to 1), standing in for a
    KernelGAN-estimated kernel (reference: CEMnet(conf, upscale_kernel=ndarray))."""
    ax = np.arange(size) - (size - 1) / 2.0
    xx, yy = np.meshgrid(ax, ax)
    c, s = np.cos(theta), np.sin(theta)
    u, v = c * xx + s * yy, -s * xx + c * yy
    k = np.exp(-0.5 * ((u / s1) ** 2 + (v / s2) ** 2))
    return k / k.sum()


def _reset_kernel_cache():
    from CEM.imresize_CEM import imresize
    imresize.kernels = {}


def _cem(sf, kernel=None, bound=None, **conf_over):
    import CEM.CEMnet as C
    _reset_kernel_cache()
    conf = C.Get_CEM_Conf(sf)
    if bound is not None:
        conf.lower_magnitude_bound = bound
    for k, v in conf_over.items():
        setattr(conf, k, v)
    return C.CEMnet(conf, upscale_kernel=kernel)


def gen_F1():
    """CEM taps / margins / strides (pins A6)."""
    from CEM.imresize_CEM import calc_strides
    out = {}
    cases = [('cubic_x2', 2, None, None), ('cubic_x3', 3, None, None), ('cubic_x4', 4, None, None),
             ('cubic_x8', 8, None, None), ('blurry1.0_x4', 4, 'blurry_cubic_1.0', None),
             ('blurry2.0_x8', 8, 'blurry_cubic_2.0', None),
             ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1), ('aniso_x8', 8, aniso_gaussian_kernel(17, 4.0, 1.8, 0.6), 0.1)]
    for name, sf, kernel, bound in cases:
        cem = _cem(sf, kernel, bound)
        pre, post = calc_strides(None, sf)
        out[name + '/ds_kernel'] = np.asarray(cem.ds_kernel, dtype=np.float64)
        out[name + '/inv_hTh'] = np.asarray(cem.inv_hTh, dtype=np.float64)
        out[name + '/ints'] = np.array([sf, cem.ds_kernel_invalidity_half_size_LR, cem.inv_hTh_invalidity_half_size,
                                        cem.invalidity_margins_LR, cem.invalidity_margins_HR, pre[0], post[0]], dtype=np.int64)
        print(name, cem.ds_kernel.shape, cem.inv_hTh.shape, out[name + '/ints'])
    np.savez_compressed(os.path.join(GOLDEN, 'cem_taps.npz'), **out)


def gen_F2():
    """The three Filter_Layer ops + CEM_downsampler individually (pins A7 / index conventions)."""
    import CEM.CEMnet as C
    out = {}
    for name, sf, kernel, bound in [('cubic_x4', 4, None, None), ('cubic_x2', 2, None, None), ('cubic_x3', 3, None, None),
                                    ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1)]:
        cem = _cem(sf, kernel, bound)
        net = cem.WrapArchitecture_PyTorch(generated_image=None)
        lr = seeded_uniform((2, 3, 20, 24), 11)
        hr = seeded_uniform((2, 3, 20 * sf, 24 * sf), 12)
        with torch.no_grad():
            out[name + '/DownscaleOP'] = net.DownscaleOP(hr).numpy()
            out[name + '/Conv_LR_with_Inv_hTh_OP'] = net.Conv_LR_with_Inv_hTh_OP(lr).numpy()
            out[name + '/Upscale_OP'] = net.Upscale_OP(lr).numpy()
        if kernel is None:
            _reset_kernel_cache()
            ds = C.CEM_downsampler(sf)
            with torch.no_grad():
                out[name + '/CEM_downsampler'] = ds(hr).numpy()
            g = C.CEM_downsampler(sf, grayscale=True)
            with torch.no_grad():
                out[name + '/CEM_downsampler_gray'] = g(hr[:, :1]).numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'cem_filter_ops.npz'), **out)


def gen_F3():
    """CEM_PyTorch with generated_image=None (pair input): train/eval, sigmoid-range and
    decomposed variants; plus the NumPy projections (pins A8)."""
    out = {}
    for name, sf, kernel, bound in [('cubic_x4', 4, None, None), ('cubic_x2', 2, None, None), ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1)]:
        lr = seeded_uniform((2, 3, 12, 16), 21)
        gen = seeded_uniform((2, 3, 12 * sf, 16 * sf), 22)
        cem = _cem(sf, kernel, bound)
        net = cem.WrapArchitecture_PyTorch(generated_image=None)
        with torch.no_grad():
            net.train()
            out[name + '/train'] = net([lr, gen]).numpy()
            net.eval()
            out[name + '/eval'] = net([lr, gen]).numpy()
        # consistency of the train-mode output: downscale(out) vs lr, whole frame
        with torch.no_grad():
            net.train()
            y = net([lr, gen])
            d = net.DownscaleOP(y)
            out[name + '/train_consistency_interior_rmse'] = np.array(
                float(((d - lr)[:, :, cem.invalidity_margins_LR // 2:-(cem.invalidity_margins_LR // 2) or None,
                                cem.invalidity_margins_LR // 2:-(cem.invalidity_margins_LR // 2) or None] ** 2).mean().sqrt()))
    # option variants on cubic x4
    lr = seeded_uniform((2, 3, 12, 16), 21)
    gen = seeded_uniform((2, 3, 48, 64), 22)
    cem = _cem(4, sigmoid_range_limit=True, input_range=np.array([0, 1]))
    net = cem.WrapArchitecture_PyTorch(generated_image=None)
    with torch.no_grad():
        net.train()
        out['cubic_x4_sigmoid/train'] = net([lr, gen]).numpy()
    cem = _cem(4, decomposed_output=True)
    net = cem.WrapArchitecture_PyTorch(generated_image=None)
    with torch.no_grad():
        net.train()
        o = net([lr, gen])
        out['cubic_x4_decomposed/train_ortho'] = o[0].numpy()
        out['cubic_x4_decomposed/train_NS'] = o[1].numpy()
        net.eval()
        out['cubic_x4_decomposed/eval'] = net([lr, gen]).numpy()
    # NumPy projections
    cem = _cem(4)
    rng = np.random.Generator(np.random.PCG64(23))
    hr_np = rng.random((48, 48, 3))
    lr_np = rng.random((12, 12, 3))
    out['numpy/Project_2_ortho_2_NS'] = cem.Project_2_ortho_2_NS(hr_np)
    out['numpy/DT_Satisfying_Upscale'] = cem.DT_Satisfying_Upscale(lr_np)
    out['numpy/Enforce_DT_on_Image_Pair'] = cem.Enforce_DT_on_Image_Pair(lr_np, hr_np)
    from CEM.imresize_CEM import imresize
    out['numpy/imresize_down4'] = imresize(hr_np, scale_factor=[0.25])
    out['numpy/imresize_up4'] = imresize(lr_np, scale_factor=[4])
    np.savez_compressed(os.path.join(GOLDEN, 'cem_forward.npz'), **out)


def _param_digest(t):
    f = t.detach().reshape(-1).double()
    n = f.numel()
    idx = torch.linspace(0, n - 1, steps=24).long()
    return np.concatenate([[float(f.sum()), float(f.norm())], f[idx].numpy()])


def gen_F4():
    """RRDBNet forward + input-grad + weight-grad digests (pins A1-A5 fwd/bwd)."""
    import models.modules.architecture as arch
    out = {}
    cases = [('nb1_x4', 1, 4, 0), ('nb3_x4', 3, 4, 0), ('nb1_x8', 1, 8, 0), ('nb1_x2', 1, 2, 0),
             ('nb1_x4_lat3', 1, 4, 3), ('nb2_x4_lat3', 2, 4, 3), ('nb1_x2_lat1', 1, 2, 1)]
    for name, nb, sf, lat in cases:
        torch.manual_seed(0)
        net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, gc=32, upscale=sf, norm_type=None, act_type='leakyrelu',
                           mode='CNA', upsample_mode='upconv',
                           latent_input='all_layers_HR_downscaled' if lat else None, num_latent_channels=lat)
        n = fill_formula_weights(net, gain=1.0)
        h, w = (12, 16) if sf != 8 else (8, 8)
        x = seeded_uniform((1, 3 + lat * sf * sf, h, w), 31 + nb + sf + lat, -1.0 if lat else 0.0, 1.0)
        if lat:
            x[:, -3:] = x[:, -3:] * 0.5 + 0.5
        x.requires_grad_(True)
        y = net(x)
        cot = seeded_uniform(tuple(y.shape), 41 + nb + sf + lat, -1.0, 1.0)
        (y * cot).sum().backward()
        out[name + '/out'] = y.detach().numpy()
        out[name + '/dx'] = x.grad.numpy()
        out[name + '/dparams'] = np.stack([_param_digest(p.grad) for _, p in net.named_parameters()])
        out[name + '/nparams'] = np.array([n, sum(p.numel() for p in net.parameters())])
        print(name, tuple(y.shape), float(y.detach().abs().mean()), float(x.grad.abs().mean()))
    np.savez_compressed(os.path.join(GOLDEN, 'rrdb_fwd_bwd.npz'), **out)


def gen_F8():
    """RRDBNet with the latent fed to the FIRST layer only ('first_layer_HR_downscaled', architecture.py:245-246,288-299): forward,
    input gradient and weight-gradient digests.  A separate fixture file so that F4's stays byte-identical."""
    import models.modules.architecture as arch
    out = {}
    for name, nb, sf, lat in [('nb1_x4_lat3_first', 1, 4, 3), ('nb2_x2_lat1_first', 2, 2, 1)]:
        torch.manual_seed(0)
        net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, gc=32, upscale=sf, norm_type=None, act_type='leakyrelu',
                           mode='CNA', upsample_mode='upconv', latent_input='first_layer_HR_downscaled', num_latent_channels=lat)
        n = fill_formula_weights(net, gain=1.0)
        x = seeded_uniform((1, 3 + lat * sf * sf, 12, 16), 61 + nb + sf + lat, -1.0, 1.0)
        x[:, -3:] = x[:, -3:] * 0.5 + 0.5
        x.requires_grad_(True)
        y = net(x)
        cot = seeded_uniform(tuple(y.shape), 71 + nb + sf + lat, -1.0, 1.0)
        (y * cot).sum().backward()
        out[name + '/out'] = y.detach().numpy()
        out[name + '/dx'] = x.grad.numpy()
        out[name + '/dparams'] = np.stack([_param_digest(p.grad) for _, p in net.named_parameters()])
        out[name + '/nparams'] = np.array([n, sum(p.numel() for p in net.parameters())])
        print(name, tuple(y.shape), float(y.detach().abs().mean()), float(x.grad.abs().mean()))
    np.savez_compressed(os.path.join(GOLDEN, 'rrdb_first_layer.npz'), **out)


NF_CASES = [('nf32_nb2_x4', 32, 2, 4, 0), ('nf32_nb1_x4_lat3', 32, 1, 4, 3), ('nf48_nb1_x2', 48, 1, 2, 0), ('nf16_nb1_x4_lat1', 16, 1, 4, 1),
            ('nf128_nb1_x4_lat3', 128, 1, 4, 3), ('nf128_nb1_x2', 128, 1, 2, 0)]


def gen_F13():
    """RRDBNet with nf != 64 (architecture.py:228-230 takes any nf; growth channels stay 32): forward, input gradient, weight-gradient digests.
    Its own fixture file (round 6): F4's stays byte-identical."""
    import models.modules.architecture as arch
    out = {}
    for name, nf, nb, sf, lat in NF_CASES:
        torch.manual_seed(0)
        net = arch.RRDBNet(in_nc=3, out_nc=3, nf=nf, nb=nb, gc=32, upscale=sf, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                           latent_input='all_layers_HR_downscaled' if lat else None, num_latent_channels=lat)
        n = fill_formula_weights(net, gain=1.0)
        x = seeded_uniform((1, 3 + lat * sf * sf, 12, 16), 131 + nf + nb + sf + lat, -1.0 if lat else 0.0, 1.0)
        if lat:
            x[:, -3:] = x[:, -3:] * 0.5 + 0.5
        x.requires_grad_(True)
        y = net(x)
        cot = seeded_uniform(tuple(y.shape), 141 + nf + nb + sf + lat, -1.0, 1.0)
        (y * cot).sum().backward()
        out[name + '/out'] = y.detach().numpy()
        out[name + '/dx'] = x.grad.numpy()
        out[name + '/dparams'] = np.stack([_param_digest(p.grad) for _, p in net.named_parameters()])
        out[name + '/nparams'] = np.array([n, sum(p.numel() for p in net.parameters())])
        print(name, tuple(y.shape), float(y.detach().abs().mean()), float(x.grad.abs().mean()))
    np.savez_compressed(os.path.join(GOLDEN, 'rrdb_nf.npz'), **out)


def gen_F9():
    """RRDBNet(upsample_mode='pixelshuffle') (architecture.py:254-259 -> block.py:278-291): forward, input gradient, weight-gradient digests."""
    import models.modules.architecture as arch
    out = {}
    for name, nb, sf, lat in [('nb1_x4_ps', 1, 4, 0), ('nb2_x2_ps', 2, 2, 0)]:      # (with a latent input the reference's forward fails: it concatenates Z in front of the shuffle block)
        torch.manual_seed(0)
        net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, gc=32, upscale=sf, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='pixelshuffle',
                           latent_input='all_layers_HR_downscaled' if lat else None, num_latent_channels=lat)
        n = fill_formula_weights(net, gain=1.0)
        x = seeded_uniform((1, 3 + lat * sf * sf, 12, 16), 81 + nb + sf + lat, -1.0 if lat else 0.0, 1.0)
        if lat:
            x[:, -3:] = x[:, -3:] * 0.5 + 0.5
        x.requires_grad_(True)
        y = net(x)
        cot = seeded_uniform(tuple(y.shape), 91 + nb + sf + lat, -1.0, 1.0)
        (y * cot).sum().backward()
        out[name + '/out'] = y.detach().numpy()
        out[name + '/dx'] = x.grad.numpy()
        out[name + '/dparams'] = np.stack([_param_digest(p.grad) for _, p in net.named_parameters()])
        out[name + '/keys'] = np.array(list(net.state_dict().keys()))
        out[name + '/nparams'] = np.array([n, sum(p.numel() for p in net.parameters())])
        print(name, tuple(y.shape), float(y.detach().abs().mean()), float(x.grad.abs().mean()))
    np.savez_compressed(os.path.join(GOLDEN, 'rrdb_pixelshuffle.npz'), **out)


def gen_F10():
    """FilterLoss (loss.py:27-209) in its model-training form for two structure-tensor latent codes: three consecutive calls (the
    percentile history of the ratios accumulates), loss values [B, 3] of each call and d(sum of the last loss)/d(SR)."""
    from models.modules.loss import FilterLoss
    out = {}
    for code in ('SVDinNormedOut_structure_tensor', 'structure_tensor'):
        fl = FilterLoss(latent_channels=code)
        for call in range(3):
            sr = seeded_uniform((4, 3, 24, 20), 1000 + call).requires_grad_(True)
            hr = seeded_uniform((4, 3, 24, 20), 1010 + call)
            z = seeded_uniform((4, 3, 1, 1), 1020 + call, -1.0, 1.0) * torch.ones(4, 3, 24, 20)
            loss = fl({'SR': sr, 'HR': hr, 'Z': z})
            out['%s/call%d' % (code, call)] = loss.detach().numpy()
        loss.sum().backward()
        out[code + '/dSR'] = sr.grad.numpy()
        print(code, loss.detach().numpy()[0])
    np.savez_compressed(os.path.join(GOLDEN, 'filter_loss.npz'), **out)


def gen_F11():
    """SoftHistogramLoss (Z_optimization.py:24-230), gray scale / patch size 1 / fixed temperature: KL loss of a batch of two images against a
    desired image's histogram, with and without an image mask, and its gradient w.r.t. the images."""
    from Z_optimization import SoftHistogramLoss
    out = {}
    desired = seeded_uniform((1, 3, 40, 36), 1101)
    for name, mask in (('plain', None), ('masked', (seeded_uniform((40, 36), 1103) > 0.4).float())):
        loss_fn = SoftHistogramLoss(bins=64, min=0, max=1, desired_hist_image=[desired], desired_hist_image_mask=[None], input_im_HR_mask=mask, gray_scale=True,
                                    patch_size=1, temperature=2e-3)
        cur = (seeded_uniform((2, 3, 40, 36), 1102) ** 2).requires_grad_(True)
        loss = loss_fn(cur)
        loss.backward()
        out[name + '/loss'] = np.array(float(loss))
        out[name + '/grad'] = cur.grad.numpy().copy()
        out[name + '/desired_hist'] = loss_fn.desired_hists_list[0].numpy().copy()
        if mask is not None:
            out[name + '/mask'] = mask.numpy()
        print(name, float(loss), float(cur.grad.abs().max()))
    np.savez_compressed(os.path.join(GOLDEN, 'soft_histogram.npz'), **out)


def _wrapped_G(nb, sf, lat=0, kernel=None, gain=1.0):
    import models.modules.architecture as arch
    cem = _cem(sf, kernel)
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, gc=32, upscale=sf, norm_type=None, act_type='leakyrelu',
                       mode='CNA', upsample_mode='upconv',
                       latent_input='all_layers_HR_downscaled' if lat else None, num_latent_channels=lat)
    G = cem.WrapArchitecture_PyTorch(net)
    fill_formula_weights(G, gain=gain)
    return cem, G


def gen_F5():
    """Config 1 exactly: RRDB-3 x4 + CEM eval on [1,3,32,32] (+ an explorable lat=3 variant)."""
    out = {}
    cem, G = _wrapped_G(3, 4)
    x = seeded_uniform((1, 3, 32, 32), 51)
    G.eval()
    with torch.no_grad():
        y = G(x)
        d = G.LR_unpadder(G.DownscaleOP(G.HR_padder(y)))
    m = cem.invalidity_margins_LR
    out['c1/out'] = y.numpy()
    out['c1/consistency_interior_rmse'] = np.array(float(((d - x)[:, :, m:-m, m:-m] ** 2).mean().sqrt()))
    out['c1/keys'] = np.array(list(G.state_dict().keys()))
    out['c1/key_shapes'] = np.array([str(tuple(v.shape)) for v in G.state_dict().values()])
    print('c1', tuple(y.shape), float(y.mean()), out['c1/consistency_interior_rmse'])
    # train mode (no pre-pad)
    G.train()
    with torch.no_grad():
        out['c1/out_train_mode'] = G(x).numpy()
    # explorable: Z given HR-res, packed by view as SRRaGANModel.Prepare_Input does (SRRaGAN_model.py:230-236)
    cem, G = _wrapped_G(2, 4, lat=3)
    z = seeded_uniform((1, 3, 128, 128), 52, -1.0, 1.0)
    xin = torch.cat([z.contiguous().view(1, 3 * 16, 32, 32), x], 1)
    G.eval()
    with torch.no_grad():
        y = G(xin)
    out['c1_lat3/out'] = y.numpy()
    out['c1_lat3/keys'] = np.array(list(G.state_dict().keys()))
    out['c1_lat3/key_shapes'] = np.array([str(tuple(v.shape)) for v in G.state_dict().values()])
    print('c1_lat3', tuple(y.shape), float(y.mean()))
    np.savez_compressed(os.path.join(GOLDEN, 'c1_end_to_end.npz'), **out)


def gen_F6():
    """RRDB-23 x4 + CEM on one [1,3,128,128] input: statistics + crops (depth accumulation)."""
    out = {}
    cem, G = _wrapped_G(23, 4, gain=0.6)
    x = seeded_uniform((1, 3, 128, 128), 61)
    G.eval()
    with torch.no_grad():
        y = G(x)
    out['stats'] = np.stack([y.mean((0, 2, 3)).numpy(), y.std((0, 2, 3)).numpy(), y.amin((0, 2, 3)).numpy(), y.amax((0, 2, 3)).numpy()])
    out['crop64'] = y[:, :, 200:264, 300:364].numpy()
    out['stride8'] = y[:, :, 3::8, 5::8].numpy()
    out['nparams'] = np.array([sum(p.numel() for n, p in G.named_parameters() if 'Filter_OP' not in n),
                               sum(p.numel() for p in G.parameters())])
    print('rrdb23', out['stats'], out['nparams'])
    np.savez_compressed(os.path.join(GOLDEN, 'c2_rrdb23_probe.npz'), **out)


def _ref_opt(is_train, nb=1, lat=3, gan=False, batch=2, root='/tmp/esr_ref_f7', train_extra=None):
    """Options of the reference's model wrapper (codes/options/train/train_explorable_SR.json, reduced to what SRRaGANModel reads)."""
    from options.options import dict_to_nonedict
    os.makedirs(os.path.join(root, 'models'), exist_ok=True)
    train = {'resume': 0, 'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_D': 1e-4, 'weight_decay_D': 0, 'beta1_D': 0.9, 'lr_scheme': 'MultiStepLR',
             'lr_steps': [100000], 'lr_gamma': 0.5, 'pixel_domain': 'HR', 'pixel_criterion': 'l1', 'pixel_weight': 1, 'range_weight': 5000, 'CEM_exp': 1,
             'grad_accumulation_steps_G': 1, 'grad_accumulation_steps_D': 1, 'D_verification': None, 'D_update_ratio': 1, 'D_init_iters': 0}
    if gan:
        train.update({'gan_type': 'wgan-gp', 'gan_weight': 1, 'gp_weight': 10})
    train.update(train_extra or {})
    return dict_to_nonedict({
        'name': 'f7', 'model': 'srragan', 'scale': 4, 'gpu_ids': None, 'range': [0, 1], 'is_train': is_train,
        'path': {'root': root, 'models': os.path.join(root, 'models'), 'log': root, 'experiments_root': root, 'val_images': root},
        'network_G': {'which_model_G': 'RRDB_net', 'CEM_arch': 1, 'sigmoid_range_limit': 0, 'latent_input': 'all_layers' if lat else 'None',
                      'latent_input_domain': 'HR_downscaled', 'latent_channels': lat, 'norm_type': None, 'mode': 'CNA', 'nf': 64, 'nb': nb,
                      'in_nc': 3, 'out_nc': 3, 'gc': 32, 'group': 1, 'scale': 4},
        'network_D': {'which_model_D': 'discriminator_vgg_128', 'relativistic': 0, 'decomposed_input': 0, 'pre_clipping': 0, 'add_quantization_noise': 0,
                      'norm_type': 'batch', 'act_type': 'leakyrelu', 'mode': 'CNA', 'n_layers': 10, 'nf': 64, 'in_nc': 3},
        'datasets': {'train': {'patch_size': 208, 'batch_size': batch}}, 'train': train if is_train else None, 'test': {'kernel': None}})


def _norms(params):
    return np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in params])


def _train_batch(batch=2, seed=900):
    return {'LR': seeded_uniform((batch, 3, 52, 52), seed), 'HR': seeded_uniform((batch, 3, 208, 208), seed + 1),
            'Z': seeded_uniform((batch, 3, 208, 208), seed + 2, -1.0, 1.0)}


def gen_F7():
    """The callers, pinned by the reference itself:
      g_only/*  SRRaGANModel.optimize_parameters() without a discriminator (RRDB-1, lat 3, batch 2, patch 208): the reference idles on its
                first call and steps G on the second: l_g_pix / l_g_range, per-parameter gradient norms, weight deltas of the Adam step
      gd/*      the same with Discriminator_VGG_128 + WGAN-GP (D_verification off, D_update_ratio 1): call 1 = D step only, call 2 = D and G
                steps: D losses, penalty, interpolation points drawn, D / G gradient norms, l_g_gan, D logits, BatchNorm running means
      z_<obj>/* Z_optimizer.optimize() on the eval-mode model (RRDB-1, lat 3, B=3, 4 iterations, lr 0.1): loss trajectory, final Z digest"""
    import contextlib
    import io
    import models
    out = {}
    quiet = contextlib.redirect_stdout(io.StringIO())
    # ---- (i) generator-only step
    with quiet:
        m = models.create_model(_ref_opt(True))
    fill_formula_weights(m.netG, gain=0.5)
    data = _train_batch()
    gp = [p for n, p in m.netG.named_parameters() if p.requires_grad]
    before = [p.detach().clone() for p in gp]
    for _ in range(2):
        m.feed_data({k: v.clone() for k, v in data.items()})
        m.optimize_parameters()
    log = m.get_current_log()
    out['g_only/l_g_pix'], out['g_only/l_g_range'] = np.array(log['l_g_pix']), np.array(log['l_g_range'])
    out['g_only/grad_norms'] = _norms(gp)
    out['g_only/delta_norms'] = np.array([float((p.detach() - b).double().norm()) for p, b in zip(gp, before)])
    out['g_only/fea_weight_after'] = _param_digest(gp[0])
    out['g_only/fake_H_digest'] = _param_digest(m.fake_H)
    print('g_only', log)
    # ---- (ii) G + D step
    with quiet:
        m = models.create_model(_ref_opt(True, gan=True))
    fill_formula_weights(m.netG, gain=0.5)
    fill_formula_weights(m.netD, gain=1.0)
    out['gd/D_keys'] = np.array(list(m.netD.state_dict().keys()))
    out['gd/D_key_shapes'] = np.array([str(tuple(v.shape)) for v in m.netD.state_dict().values()])
    m.netD.eval()
    with torch.no_grad():
        out['gd/D_eval_out'] = m.netD(seeded_uniform((2, 3, 128, 128), 910)).numpy()
    m.netD.train()
    gp = [p for n, p in m.netG.named_parameters() if p.requires_grad]
    dp = list(m.netD.parameters())
    for call in range(2):
        torch.manual_seed(1234 + call)
        m.feed_data({k: v.clone() for k, v in data.items()})
        m.optimize_parameters()
        out['gd/call%d/random_pt' % call] = m.random_pt.detach().numpy().copy()
        out['gd/call%d/D_grad_norms' % call] = _norms(dp)
        log = m.get_current_log()
        for k in ('l_d_real', 'l_d_fake', 'l_d_gp', 'D_real', 'D_fake', 'D_logits_diff', 'Correctly_distinguished'):
            out['gd/call%d/%s' % (call, k)] = np.array(log[k])
        print('gd call', call, {k: v for k, v in log.items()})
    for k in ('l_g_gan', 'l_g_pix', 'l_g_range'):
        out['gd/call1/' + k] = np.array(log[k])
    out['gd/call1/G_grad_norms'] = _norms(gp)
    out['gd/bn_running_mean_first'] = m.netD.state_dict()['features.3.running_mean'].numpy().copy()
    # ---- (iii) latent search
    from Z_optimization import Z_optimizer
    with quiet:
        m = models.create_model(_ref_opt(False))
    fill_formula_weights(m.netG, gain=0.5)
    lr = seeded_uniform((1, 3, 24, 28), 920)
    B = 3
    for obj in ('STD_increase', 'max_STD', 'TV'):
        z0 = seeded_uniform((B, 3, 96, 112), 921, -0.3, 0.3)
        m.feed_data({'LR': lr.expand(B, -1, -1, -1).clone(), 'Z': z0.clone()}, need_GT=False)
        m.test()
        with quiet:
            zo = Z_optimizer(objective=obj, Z_size=[96, 112], model=m, Z_range=1, max_iters=4, data={'LR': lr.expand(B, -1, -1, -1).clone(), 'STD_increment': 0.01}, initial_Z=z0.clone(),
                             initial_LR=0.1, batch_size=B)
            z = zo.optimize()
        out['z_%s/loss' % obj] = np.array(zo.loss_values, dtype=np.float64)
        out['z_%s/final_Z_digest' % obj] = _param_digest(z)
        out['z_%s/final_Z_sub' % obj] = z[:, :, ::16, ::16].numpy().copy()
        out['z_%s/initial_STD' % obj] = zo.initial_STD.detach().numpy().copy()
        print(obj, zo.loss_values)
    # a user-marked region (GUI.py:1925-2057): image_mask limits the objective, Z_mask limits which latent entries may move
    for obj in ('max_STD', 'TV'):
        im_mask = np.zeros([96, 112], dtype=np.float32); im_mask[24:72, 32:96] = 1
        z_mask = np.zeros([96, 112], dtype=np.float32); z_mask[16:80, 24:104] = 1
        z0 = seeded_uniform((B, 3, 96, 112), 921, -0.3, 0.3)
        m.feed_data({'LR': lr.expand(B, -1, -1, -1).clone(), 'Z': z0.clone()}, need_GT=False)
        m.test()
        with quiet:
            zo = Z_optimizer(objective=obj, Z_size=[96, 112], model=m, Z_range=1, max_iters=3, data={'LR': lr.expand(B, -1, -1, -1).clone(), 'STD_increment': 0.01},
                             initial_Z=z0.clone(), initial_LR=0.1, batch_size=B, image_mask=im_mask, Z_mask=z_mask)
            z = zo.optimize()
        out['zmask_%s/loss' % obj] = np.array(zo.loss_values, dtype=np.float64)
        out['zmask_%s/final_Z_sub' % obj] = z[:, :, ::8, ::8].numpy().copy()
        out['zmask_%s/initial_STD' % obj] = zo.initial_STD.detach().numpy().copy()
        print('masked', obj, zo.loss_values)
    np.savez_compressed(os.path.join(GOLDEN, 'callers_f7.npz'), **out)


def gen_F12():
    """The optimal-Z dual pass of the training step (SRRaGAN_model.py:314-338, the reference's own train_explorable_SR.json: wgan-gp +
    optimalZ_loss_type 'l1', weight 100): once the generator has stepped, every call runs twice — first on Z found by a 10-iteration
    Z_optimizer('l1') search through the frozen generator, then on the assigned Z — and the D backward of the first pass must leave the graph of
    pred_d_real alive for the second (:400).  Calls 0 (D only), 1 (D + G; one pass), 2 (D + G, two passes): critic losses, gradient norms of D
    and G after call 2, l_g_optimalZ."""
    import contextlib
    import io
    import models
    out = {}
    quiet = contextlib.redirect_stdout(io.StringIO())
    with quiet:
        m = models.create_model(_ref_opt(True, gan=True, train_extra={'optimalZ_loss_type': 'l1', 'optimalZ_loss_weight': 100, 'Num_Z_iterations': [4]}))
    fill_formula_weights(m.netG, gain=0.5)
    fill_formula_weights(m.netD, gain=1.0)
    data = _train_batch()
    gp = [p for n, p in m.netG.named_parameters() if p.requires_grad]
    dp = list(m.netD.parameters())
    draws = {'pt': [], 'z': []}
    zm = m.Z_optimizer.Z_model
    orig_rand, orig_unif = zm.Randomize_Z, m.random_pt.uniform_

    def rec_rand(*a, **kw):                 # the search starts from a fresh xavier-uniform Z (Z_optimization.py:654-655): record the draw
        orig_rand(*a, **kw)
        draws['z'].append(zm.Z.data.clone().numpy())

    def rec_unif(*a, **kw):                 # one draw of interpolation points per dual pass (:366)
        r = orig_unif(*a, **kw)
        draws['pt'].append(m.random_pt.detach().clone().numpy())
        return r
    zm.Randomize_Z, m.random_pt.uniform_ = rec_rand, rec_unif
    for call in range(3):
        torch.manual_seed(4321 + call)
        draws['pt'], draws['z'] = [], []
        m.feed_data({k: v.clone() for k, v in data.items()})
        with quiet:
            m.optimize_parameters()
        log = m.get_current_log()
        out['call%d/random_pts' % call] = np.stack(draws['pt'])                     # [passes, B, 1, 1, 1]
        if draws['z']:
            out['call%d/initial_pre_tanh_Z' % call] = np.stack(draws['z'])[0]
            out['call%d/z_search_losses' % call] = np.array(m.Z_optimizer.loss_values, dtype=np.float64)
            out['call%d/optimal_Z_sub' % call] = zm.Return_Detached_Z()[:, :, ::16, ::16].numpy().copy()
        out['call%d/D_grad_norms' % call] = _norms(dp)
        for k in ('l_d_real', 'l_d_fake', 'l_d_gp', 'D_real', 'D_fake', 'D_logits_diff'):
            out['call%d/%s' % (call, k)] = np.array(log[k])
        if call:
            out['call%d/G_grad_norms' % call] = _norms(gp)
            for k in ('l_g_gan', 'l_g_pix', 'l_g_range', 'l_g_optimalZ'):
                if k in log:
                    out['call%d/%s' % (call, k)] = np.array(log[k])
        print('dual call', call, {k: v for k, v in log.items()})
    out['optimal_Z_digest'] = _param_digest(zm.Z.detach())
    np.savez_compressed(os.path.join(GOLDEN, 'callers_dual.npz'), **out)


ALL = {'F12': gen_F12, 'F1': gen_F1, 'F2': gen_F2, 'F3': gen_F3, 'F4': gen_F4, 'F5': gen_F5, 'F6': gen_F6, 'F7': gen_F7, 'F8': gen_F8, 'F9': gen_F9, 'F10': gen_F10, 'F11': gen_F11, 'F13': gen_F13}

if __name__ == '__main__':
    _refshim.install()
    torch.set_num_threads(os.cpu_count())
    os.makedirs(GOLDEN, exist_ok=True)
    for k in (sys.argv[1:] or list(ALL)):
        print('==', k)
        ALL[k]()
