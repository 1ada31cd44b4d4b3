"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself
(oracle/gen_golden.py, run in the build container where /root/reference exists).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cem_oracle as co
from oracle import rrdb_oracle as ro
from oracle.check_golden import load, rel_l2, rel_max
from oracle.weights import formula_tensor, seeded_uniform
from oracle.gen_golden import aniso_gaussian_kernel

TAP_CASES = [('cubic_x2', 2, None, None), ('cubic_x3', 3, None, None), ('cubic_x4', 4, None, None),
             ('cubic_x8', 8, None, None), ('blurry1.0_x4', 4, 'blurry_cubic_1.0', None),
             ('blurry2.0_x8', 8, 'blurry_cubic_2.0', None),
             ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1), ('aniso_x8', 8, aniso_gaussian_kernel(17, 4.0, 1.8, 0.6), 0.1)]


@pytest.mark.parametrize('name,sf,kernel,bound', TAP_CASES, ids=[c[0] for c in TAP_CASES])
def test_taps_match_reference(name, sf, kernel, bound):
    g = load('cem_taps.npz')
    t = co.CEMTaps(sf, kernel, lower_magnitude_bound=bound or 0.01)
    assert t.ds_kernel.shape == g[name + '/ds_kernel'].shape
    assert t.inv_hTh.shape == g[name + '/inv_hTh'].shape
    np.testing.assert_allclose(t.ds_kernel, g[name + '/ds_kernel'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(t.inv_hTh, g[name + '/inv_hTh'], rtol=0, atol=1e-13)
    ints = np.array([sf, t.ds_half, t.inv_half, t.margins_LR, t.margins_HR, t.pre, t.post])
    assert (ints == g[name + '/ints']).all()          # integer conventions: bit-exact


def test_known_answers_survey():
    """Structural known answers: supports, sums, margins, strides."""
    t = co.CEMTaps(4)
    assert t.up_kernel.shape == (17, 17) and abs(t.up_kernel.sum() - 16.0) < 1e-6
    assert t.ds_kernel.shape == (17, 17) and abs(t.ds_kernel.sum() - 1.0) < 1e-7
    assert t.inv_hTh.shape == (27, 27) and abs(t.inv_hTh[13, 13] - 1.5548) < 1e-3
    assert (t.ds_half, t.inv_half, t.margins_LR, t.margins_HR) == (2, 6, 10, 40)
    assert [co.calc_strides(s)[0] for s in (2, 3, 4, 8)] == [0, 1, 1, 3]
    s = np.linalg.svd(t.ds_kernel, compute_uv=False)
    assert s[1] / s[0] < 1e-6      # bicubic taps are separable (rank 1), float32-rounded


F2_CASES = [('cubic_x4', 4, None, None), ('cubic_x2', 2, None, None), ('cubic_x3', 3, None, None), ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1)]


@pytest.mark.parametrize('name,sf,kernel,bound', F2_CASES, ids=[c[0] for c in F2_CASES])
def test_filter_ops_match_reference(name, sf, kernel, bound):
    g = load('cem_filter_ops.npz')
    t = co.CEMTaps(sf, kernel, lower_magnitude_bound=bound or 0.01)
    lr = seeded_uniform((2, 3, 20, 24), 11)
    hr = seeded_uniform((2, 3, 20 * sf, 24 * sf), 12)
    np.testing.assert_allclose(co.downscale_op(hr, t).numpy(), g[name + '/DownscaleOP'], atol=2e-6)
    np.testing.assert_allclose(co.conv_lr_with_inv_hTh(lr, t).numpy(), g[name + '/Conv_LR_with_Inv_hTh_OP'], atol=5e-6)
    np.testing.assert_allclose(co.upscale_op(lr, t).numpy(), g[name + '/Upscale_OP'], atol=2e-6)
    if kernel is None:
        np.testing.assert_allclose(co.cem_downsampler(hr, sf).numpy(), g[name + '/CEM_downsampler'], atol=2e-6)
        np.testing.assert_allclose(co.cem_downsampler(hr[:, :1], sf).numpy(), g[name + '/CEM_downsampler_gray'], atol=2e-6)


def test_cem_forward_matches_reference():
    g = load('cem_forward.npz')
    for name, sf, kernel, bound in [('cubic_x4', 4, None, None), ('cubic_x2', 2, None, None), ('aniso_x4', 4, aniso_gaussian_kernel(), 0.1)]:
        t = co.CEMTaps(sf, kernel, lower_magnitude_bound=bound or 0.01)
        lr = seeded_uniform((2, 3, 12, 16), 21)
        gen = seeded_uniform((2, 3, 12 * sf, 16 * sf), 22)
        np.testing.assert_allclose(co.cem_project(lr, gen, t, pre_pad=False).numpy(), g[name + '/train'], atol=1e-5)
        np.testing.assert_allclose(co.cem_project(lr, gen, t, pre_pad=True).numpy(), g[name + '/eval'], atol=1e-5)
    t = co.CEMTaps(4)
    lr = seeded_uniform((2, 3, 12, 16), 21)
    gen = seeded_uniform((2, 3, 48, 64), 22)
    np.testing.assert_allclose(co.cem_project(lr, gen, t, sigmoid_range=(0, 1)).numpy(), g['cubic_x4_sigmoid/train'], atol=1e-5)
    o = co.cem_project(lr, gen, t, decomposed=True)
    np.testing.assert_allclose(o[0].numpy(), g['cubic_x4_decomposed/train_ortho'], atol=1e-5)
    np.testing.assert_allclose(o[1].numpy(), g['cubic_x4_decomposed/train_NS'], atol=1e-5)
    np.testing.assert_allclose(co.cem_project(lr, gen, t, pre_pad=True, decomposed=True).numpy(), g['cubic_x4_decomposed/eval'], atol=1e-5)


def test_numpy_projections_match_reference():
    g = load('cem_forward.npz')
    t = co.CEMTaps(4)
    rng = np.random.Generator(np.random.PCG64(23))
    hr_np = rng.random((48, 48, 3))
    lr_np = rng.random((12, 12, 3))
    np.testing.assert_allclose(co.imresize_np(hr_np, sf_down=4, kernel_up=t.up_kernel), g['numpy/imresize_down4'], atol=1e-12)
    np.testing.assert_allclose(co.imresize_np(lr_np, sf_up=4, kernel_up=t.up_kernel), g['numpy/imresize_up4'], atol=1e-12)


def formula_state_dict(keys, shapes, gain=1.0):
    """Formula weights laid out by state_dict order, skipping CEM filter taps (as fill_formula_weights does)."""
    sd, l = {}, 0
    for k, shp in zip(keys, shapes):
        if 'Filter_OP' in k:
            continue
        sd[k] = formula_tensor(tuple(shp), l, is_bias=k.endswith('bias'), gain=gain)
        l += 1
    return sd


def rrdb_keys(nb, sf, lat, prefix='model', nf=64):
    """state_dict keys/shapes of RRDBNet in reference order."""
    keys, shapes = [], []

    def add(k, cout, cin):
        keys.extend([k + '.weight', k + '.bias']); shapes.extend([(cout, cin, 3, 3), (cout,)])
    add(prefix + '.0', nf, 3 + lat)
    for r in range(nb):
        for d in (1, 2, 3):
            for i in range(5):
                add('%s.1.sub.%d.RDB%d.convs.%d.0' % (prefix, r, d, i), 32 if i < 4 else nf, nf + 32 * i + lat)
    add('%s.1.sub.%d' % (prefix, nb), nf, nf + lat)
    idx = 2
    for _ in range(1 if sf == 3 else int(np.log2(sf))):
        add('%s.%d.1' % (prefix, idx), nf, nf); idx += 1
    add('%s.%d' % (prefix, idx), nf, nf + lat)
    add('%s.%d' % (prefix, idx + 2), 3, nf + lat)
    return keys, shapes


F4_CASES = [('nb1_x4', 1, 4, 0), ('nb3_x4', 3, 4, 0), ('nb1_x8', 1, 8, 0), ('nb1_x2', 1, 2, 0),
            ('nb1_x4_lat3', 1, 4, 3), ('nb2_x4_lat3', 2, 4, 3), ('nb1_x2_lat1', 1, 2, 1)]


def f4_input(nb, sf, lat):
    h, w = (12, 16) if sf != 8 else (8, 8)
    x = seeded_uniform((1, 3 + lat * sf * sf, h, w), 31 + nb + sf + lat, -1.0 if lat else 0.0, 1.0)
    if lat:
        x[:, -3:] = x[:, -3:] * 0.5 + 0.5
    return x


@pytest.mark.parametrize('name,nb,sf,lat', F4_CASES, ids=[c[0] for c in F4_CASES])
def test_rrdb_fwd_bwd_matches_reference(name, nb, sf, lat):
    g = load('rrdb_fwd_bwd.npz')
    keys, shapes = rrdb_keys(nb, sf, lat)
    sd = formula_state_dict(keys, shapes)
    assert int(g[name + '/nparams'][1]) == sum(int(np.prod(s)) for s in shapes)
    for v in sd.values():
        v.requires_grad_(True)
    x = f4_input(nb, sf, lat).requires_grad_(True)
    y = ro.rrdb_forward(sd, x, nb, sf, lat)
    assert rel_l2(y.detach().numpy(), g[name + '/out']) < 2e-6
    cot = seeded_uniform(tuple(y.shape), 41 + nb + sf + lat, -1.0, 1.0)
    (y * cot).sum().backward()
    assert rel_l2(x.grad.numpy(), g[name + '/dx']) < 5e-6
    dig = g[name + '/dparams']
    for j, k in enumerate(keys):
        f = sd[k].grad.reshape(-1).double()
        idx = torch.linspace(0, f.numel() - 1, steps=24).long()
        mine = np.concatenate([[float(f.sum()), float(f.norm())], f[idx].numpy()])
        assert abs(mine[1] - dig[j][1]) <= 1e-5 * max(dig[j][1], 1e-6), k
        np.testing.assert_allclose(mine[2:], dig[j][2:], atol=2e-5 * max(dig[j][1] / np.sqrt(f.numel()), 1e-6) + 1e-6, err_msg=k)


NF_CASES = [('nf32_nb2_x4', 32, 2, 4, 0), ('nf32_nb1_x4_lat3', 32, 1, 4, 3), ('nf48_nb1_x2', 48, 1, 2, 0), ('nf16_nb1_x4_lat1', 16, 1, 4, 1),
            ('nf128_nb1_x4_lat3', 128, 1, 4, 3), ('nf128_nb1_x2', 128, 1, 2, 0)]


def nf_input(nf, nb, sf, lat):
    x = seeded_uniform((1, 3 + lat * sf * sf, 12, 16), 131 + nf + nb + sf + lat, -1.0 if lat else 0.0, 1.0)
    if lat:
        x[:, -3:] = x[:, -3:] * 0.5 + 0.5
    return x


@pytest.mark.parametrize('name,nf,nb,sf,lat', NF_CASES, ids=[c[0] for c in NF_CASES])
def test_rrdb_with_other_stream_widths_matches_reference(name, nf, nb, sf, lat):
    """F13: the reference's RRDBNet takes any nf (architecture.py:228-230; growth channels stay 32): forward, dx and weight-gradient digests of
    nf = 16 / 32 / 48 generators generated from the reference itself, against the oracle."""
    g = load('rrdb_nf.npz')
    keys, shapes = rrdb_keys(nb, sf, lat, nf=nf)
    sd = formula_state_dict(keys, shapes)
    assert int(g[name + '/nparams'][1]) == sum(int(np.prod(s)) for s in shapes)
    for v in sd.values():
        v.requires_grad_(True)
    x = nf_input(nf, nb, sf, lat).requires_grad_(True)
    y = ro.rrdb_forward(sd, x, nb, sf, lat)
    assert rel_l2(y.detach().numpy(), g[name + '/out']) < 2e-6
    cot = seeded_uniform(tuple(y.shape), 141 + nf + nb + sf + lat, -1.0, 1.0)
    (y * cot).sum().backward()
    assert rel_l2(x.grad.numpy(), g[name + '/dx']) < 5e-6
    dig = g[name + '/dparams']
    for j, k in enumerate(keys):
        f = sd[k].grad.reshape(-1).double()
        idx = torch.linspace(0, f.numel() - 1, steps=24).long()
        mine = np.concatenate([[float(f.sum()), float(f.norm())], f[idx].numpy()])
        assert abs(mine[1] - dig[j][1]) <= 1e-5 * max(dig[j][1], 1e-6), k
        np.testing.assert_allclose(mine[2:], dig[j][2:], atol=2e-5 * max(dig[j][1] / np.sqrt(f.numel()), 1e-6) + 1e-6, err_msg=k)


def test_c1_end_to_end_matches_reference():
    """BASELINE config 1: RRDB-3 x4 + CEM (eval) on [1,3,32,32]; state_dict key names/order included."""
    g = load('c1_end_to_end.npz')
    keys, shapes = rrdb_keys(3, 4, 0, prefix='generated_image_model.model')
    ref_keys = [str(k) for k in g['c1/keys']]
    assert ref_keys[:len(keys)] == keys
    assert ref_keys[len(keys):] == ['Conv_LR_with_Inv_hTh_OP.Filter_OP.weight', 'Upscale_OP.Filter_OP.weight', 'DownscaleOP.Filter_OP.weight']
    sd = formula_state_dict(keys, shapes)
    t = co.CEMTaps(4)
    x = seeded_uniform((1, 3, 32, 32), 51)
    with torch.no_grad():
        xp = torch.nn.functional.pad(x, (t.margins_LR,) * 4, mode='replicate')
        gen = ro.rrdb_forward(sd, xp, 3, 4, 0, prefix='generated_image_model.model')
        y = co.cem_combine(xp, gen, t, crop=True)
        assert rel_l2(y.numpy(), g['c1/out']) < 2e-6 and rel_max(y.numpy(), g['c1/out']) < 1e-5
        gen = ro.rrdb_forward(sd, x, 3, 4, 0, prefix='generated_image_model.model')
        y = co.cem_combine(x, gen, t, crop=False)
        assert rel_l2(y.numpy(), g['c1/out_train_mode']) < 2e-6


def test_c1_explorable_matches_reference():
    g = load('c1_end_to_end.npz')
    keys, shapes = rrdb_keys(2, 4, 3, prefix='generated_image_model.model')
    assert [str(k) for k in g['c1_lat3/keys']][:len(keys)] == keys
    sd = formula_state_dict(keys, shapes)
    t = co.CEMTaps(4)
    x = seeded_uniform((1, 3, 32, 32), 51)
    z = seeded_uniform((1, 3, 128, 128), 52, -1.0, 1.0)
    with torch.no_grad():
        xp = torch.nn.functional.pad(x, (t.margins_LR,) * 4, mode='replicate')
        zp = torch.nn.functional.pad(z, (t.margins_HR,) * 4, mode='replicate')            # CEMnet.py:289-293
        xin = torch.cat([zp.reshape(1, 48, xp.shape[2], xp.shape[3]), xp], 1)
        gen = ro.rrdb_forward(sd, xin, 2, 4, 3, prefix='generated_image_model.model')
        y = co.cem_combine(xp, gen, t, crop=True)
    assert rel_l2(y.numpy(), g['c1_lat3/out']) < 2e-6


@pytest.mark.slow
def test_c2_rrdb23_probe_matches_reference():
    g = load('c2_rrdb23_probe.npz')
    keys, shapes = rrdb_keys(23, 4, 0, prefix='generated_image_model.model')
    assert sum(int(np.prod(s)) for s in shapes) == int(g['nparams'][0]) == 16697987
    sd = formula_state_dict(keys, shapes, gain=0.6)
    t = co.CEMTaps(4)
    x = seeded_uniform((1, 3, 128, 128), 61)
    with torch.no_grad():
        xp = torch.nn.functional.pad(x, (t.margins_LR,) * 4, mode='replicate')
        gen = ro.rrdb_forward(sd, xp, 23, 4, 0, prefix='generated_image_model.model')
        y = co.cem_combine(xp, gen, t, crop=True)
    assert rel_l2(y[:, :, 200:264, 300:364].numpy(), g['crop64']) < 5e-6
    assert rel_l2(y[:, :, 3::8, 5::8].numpy(), g['stride8']) < 5e-6


F7_CASES = [('nb1_x4_lat3_first', 1, 4, 3), ('nb2_x2_lat1_first', 2, 2, 1)]


def f7_input(nb, sf, lat):
    x = seeded_uniform((1, 3 + lat * sf * sf, 12, 16), 61 + nb + sf + lat, -1.0, 1.0)
    x[:, -3:] = x[:, -3:] * 0.5 + 0.5
    return x


def first_layer_keys(nb, sf, lat):
    """RRDBNet('first_layer_*'): only fea_conv sees the latent channels."""
    keys, shapes = rrdb_keys(nb, sf, 0)
    shapes[0] = (64, 3 + lat, 3, 3)
    return keys, shapes


@pytest.mark.parametrize('name,nb,sf,lat', F7_CASES, ids=[c[0] for c in F7_CASES])
def test_rrdb_first_layer_latent_matches_reference(name, nb, sf, lat):
    g = load('rrdb_first_layer.npz')
    keys, shapes = first_layer_keys(nb, sf, lat)
    sd = formula_state_dict(keys, shapes)
    assert int(g[name + '/nparams'][1]) == sum(int(np.prod(s)) for s in shapes)
    x = f7_input(nb, sf, lat).requires_grad_(True)
    y = ro.rrdb_forward(sd, x, nb, sf, lat, first_layer_only=True)
    assert rel_l2(y.detach().numpy(), g[name + '/out']) < 2e-6
    cot = seeded_uniform(tuple(y.shape), 71 + nb + sf + lat, -1.0, 1.0)
    (y * cot).sum().backward()
    assert rel_l2(x.grad.numpy(), g[name + '/dx']) < 5e-6


# ---- F9: RRDBNet(upsample_mode='pixelshuffle') (architecture.py:254-259 -> block.py:278-291)
F9_CASES = [('nb1_x4_ps', 1, 4, 0), ('nb2_x2_ps', 2, 2, 0)]


def pixelshuffle_keys(nb, sf):
    """The pixel-shuffle block is Sequential(conv 64 -> 64*r^2, PixelShuffle, act): its conv is child 0 (upconv: child 1) and has r^2 x the rows."""
    keys, shapes = rrdb_keys(nb, sf, 0)
    r = 3 if sf == 3 else 2
    n_up = 1 if sf == 3 else int(np.log2(sf))
    for j in range(n_up):
        for i, k in enumerate(keys):
            if k.startswith('model.%d.1.' % (2 + j)):
                keys[i] = k.replace('model.%d.1.' % (2 + j), 'model.%d.0.' % (2 + j))
                shapes[i] = (64 * r * r,) + tuple(shapes[i][1:])
    return keys, shapes


@pytest.mark.parametrize('name,nb,sf,lat', F9_CASES, ids=[c[0] for c in F9_CASES])
def test_rrdb_pixelshuffle_matches_reference(name, nb, sf, lat):
    g = load('rrdb_pixelshuffle.npz')
    keys, shapes = pixelshuffle_keys(nb, sf)
    assert keys == [str(k) for k in g[name + '/keys']]
    sd = formula_state_dict(keys, shapes)
    assert int(g[name + '/nparams'][1]) == sum(int(np.prod(s)) for s in shapes)
    for v in sd.values():
        v.requires_grad_(True)
    x = seeded_uniform((1, 3, 12, 16), 81 + nb + sf + lat).requires_grad_(True)
    y = ro.rrdb_forward(sd, x, nb, sf, lat, upsample_mode='pixelshuffle')
    assert rel_l2(y.detach().numpy(), g[name + '/out']) < 2e-6
    cot = seeded_uniform(tuple(y.shape), 91 + nb + sf + lat, -1.0, 1.0)
    (y * cot).sum().backward()
    assert rel_l2(x.grad.numpy(), g[name + '/dx']) < 5e-6
    dig = g[name + '/dparams']
    for j, k in enumerate(keys):
        f = sd[k].grad.reshape(-1).double()
        assert abs(float(f.norm()) - dig[j][1]) <= 1e-5 * max(dig[j][1], 1e-6), k
