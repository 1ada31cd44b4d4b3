"""CPU tests: the oracle's restatement of the hot path's CALLERS (oracle/callers_oracle.py) against fixture F7, which the reference's own
SRRaGANModel / Discriminator_VGG_128 / Z_optimizer produced (oracle/gen_golden.py::gen_F7).  Same formula weights, same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import callers_oracle as cao
from oracle import cem_oracle as co
from oracle.weights import formula_tensor, seeded_uniform

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NB, LAT, SF = 1, 3, 4


def f7():
    return np.load(os.path.join(GOLDEN, 'callers_f7.npz'))


def generator_state(gain, nb=NB, lat=LAT):
    """state_dict of the CEM-wrapped RRDB generator with the formula weights, keys in the reference's order (tests/golden/c1_end_to_end.npz
    holds that order for lat 3 / nb 2; built here for any nb by the same naming rule)."""
    keys, shapes = [], []

    def add(name, cout, cin):
        keys.extend(['generated_image_model.model.%s.weight' % name, 'generated_image_model.model.%s.bias' % name])
        shapes.extend([(cout, cin, 3, 3), (cout,)])
    add('0', 64, 3 + lat)
    for r in range(nb):
        for k in (1, 2, 3):
            for i in range(5):
                add('1.sub.%d.RDB%d.convs.%d.0' % (r, k, i), 32 if i < 4 else 64, 64 + 32 * i + lat)
    add('1.sub.%d' % nb, 64, 64 + lat)
    add('2.1', 64, 64); add('3.1', 64, 64)
    add('4', 64, 64 + lat); add('6', 3, 64 + lat)
    return {k: formula_tensor(s, l, is_bias=k.endswith('bias'), gain=gain) for l, (k, s) in enumerate(zip(keys, shapes))}


def discriminator_state(g, gain=1.0):
    sd, l = {}, 0
    for k, s in zip(g['gd/D_keys'], g['gd/D_key_shapes']):
        k, shape = str(k), eval(str(s))
        if k.endswith('running_mean'):
            sd[k] = torch.zeros(shape)
        elif k.endswith('running_var'):
            sd[k] = torch.ones(shape)
        elif k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        else:
            sd[k] = formula_tensor(shape, l, is_bias=k.endswith('bias'), gain=gain)      # named_parameters() order == state_dict order without buffers
            l += 1
    return sd


def batch(seed=900, n=2):
    return seeded_uniform((n, 3, 52, 52), seed), seeded_uniform((n, 3, 208, 208), seed + 1), seeded_uniform((n, 3, 208, 208), seed + 2, -1.0, 1.0)


def leaf(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}


def grad_norms(sd, names):
    return np.array([float(sd[k].grad.double().norm()) if sd[k].grad is not None else 0.0 for k in names])


def close(got, ref, tol):
    floor = 1e-4 * ref.max()
    return float((np.abs(got - ref) / np.maximum(ref, floor)).max()) < tol


@pytest.mark.slow
def test_generator_step_of_the_oracle_matches_the_reference():
    g = f7()
    taps = co.CEMTaps(SF)
    sd = leaf(generator_state(0.5))
    lr, hr, z = batch()
    m = taps.margins_HR
    fake = cao.crop(cao.generator_output(sd, lr, z, NB, LAT, SF, taps, eval_mode=False), m)
    l_pix, l_range = (fake - cao.crop(hr, m)).abs().mean(), cao.range_loss(fake)
    (1 * l_pix + 5000 * l_range).backward()
    assert abs(float(l_pix) - float(g['g_only/l_g_pix'])) < 1e-5 * float(g['g_only/l_g_pix'])
    assert abs(float(l_range) - float(g['g_only/l_g_range'])) < 1e-4 * float(g['g_only/l_g_range'])
    assert close(grad_norms(sd, list(sd)), g['g_only/grad_norms'], 1e-3)


@pytest.mark.slow
def test_discriminator_and_wgan_gp_steps_of_the_oracle_match_the_reference():
    g = f7()
    taps = co.CEMTaps(SF)
    sdG, sdD = leaf(generator_state(0.5)), leaf(discriminator_state(g))
    with torch.no_grad():
        out = cao.d_forward({k: v.detach() for k, v in sdD.items()}, seeded_uniform((2, 3, 128, 128), 910), train=False)
    np.testing.assert_allclose(out.numpy(), g['gd/D_eval_out'], rtol=1e-4, atol=1e-5)
    lr, hr, z = batch()
    m = taps.margins_HR
    real = cao.crop(hr, m)
    dnames = [k for k, v in sdD.items() if v.requires_grad]
    optD = torch.optim.Adam([sdD[k] for k in dnames], lr=1e-4, betas=(0.9, 0.999))
    for call in range(2):
        fake = cao.crop(cao.generator_output(sdG, lr, z, NB, LAT, SF, taps, eval_mode=False), m)
        optD.zero_grad()
        l_real, l_fake, l_gp, total, p_real, p_fake = cao.d_losses(sdD, real, fake.detach(), torch.from_numpy(g['gd/call%d/random_pt' % call]), 10)
        total.backward()
        for name, val in (('l_d_real', l_real), ('l_d_fake', l_fake), ('l_d_gp', l_gp), ('D_real', p_real.mean()), ('D_fake', p_fake.mean())):
            ref = float(g['gd/call%d/%s' % (call, name)])
            assert abs(float(val) - ref) < 1e-3 * abs(ref), (call, name, float(val), ref)
        assert close(grad_norms(sdD, dnames), g['gd/call%d/D_grad_norms' % call], 2e-3)
        optD.step()
    # the generator step of call 1 (after D's second update): pixel + range + GAN terms (SRRaGAN_model.py:418-472)
    l_gan = -cao.d_forward(sdD, fake).mean()
    l_pix, l_range = (fake - real).abs().mean(), cao.range_loss(fake)
    for v in sdD.values():
        if v.is_floating_point():
            v.requires_grad_(False)
    (l_pix + 5000 * l_range + l_gan).backward()
    assert abs(float(l_gan) - float(g['gd/call1/l_g_gan'])) < 1e-3 * abs(float(g['gd/call1/l_g_gan']))
    assert close(grad_norms(sdG, list(sdG)), g['gd/call1/G_grad_norms'], 2e-3)


@pytest.mark.slow
@pytest.mark.parametrize('objective', ['STD_increase', 'max_STD', 'TV'])
def test_z_search_of_the_oracle_matches_the_reference(objective):
    g = f7()
    taps = co.CEMTaps(SF)
    sd = generator_state(0.5)
    B = 3
    lr = seeded_uniform((1, 3, 24, 28), 920).expand(B, -1, -1, -1)
    z0 = seeded_uniform((B, 3, 96, 112), 921, -0.3, 0.3)
    losses, z, std0 = cao.z_search(sd, lr, z0, NB, LAT, SF, taps, objective, 4, 0.1, std_increment=0.01)
    ref = g['z_%s/loss' % objective]
    assert len(losses) == len(ref)
    np.testing.assert_allclose(losses, ref, rtol=1e-3, atol=1e-3 * abs(ref[0]))
    d = np.abs(z[:, :, ::16, ::16].numpy() - g['z_%s/final_Z_sub' % objective])
    assert np.median(d) < 1e-4 and np.mean(d > 1e-2) < 0.01
    expect0 = g['z_%s/initial_STD' % objective] - (0.01 if objective == 'STD_increase' else 0)     # the reference's desired_STD aliases initial_STD
    np.testing.assert_allclose(std0.numpy(), expect0, rtol=1e-4)


@pytest.mark.slow
@pytest.mark.parametrize('objective', ['max_STD', 'TV'])
def test_masked_z_search_of_the_oracle_matches_the_reference(objective):
    """The GUI's region tools: the objective is evaluated on output * image_mask, latent entries outside Z_mask stay where they were."""
    g = f7()
    taps = co.CEMTaps(SF)
    sd = generator_state(0.5)
    B = 3
    lr = seeded_uniform((1, 3, 24, 28), 920).expand(B, -1, -1, -1)
    z0 = seeded_uniform((B, 3, 96, 112), 921, -0.3, 0.3)
    im_mask = torch.zeros(96, 112); im_mask[24:72, 32:96] = 1
    z_mask = torch.zeros(96, 112); z_mask[16:80, 24:104] = 1
    losses, z, std0 = cao.z_search(sd, lr, z0, NB, LAT, SF, taps, objective, 3, 0.1, std_increment=0.01, image_mask=im_mask, z_mask=z_mask)
    np.testing.assert_allclose(losses, g['zmask_%s/loss' % objective], rtol=1e-3)
    np.testing.assert_allclose(std0.numpy(), g['zmask_%s/initial_STD' % objective], rtol=1e-4)
    d = np.abs(z[:, :, ::8, ::8].numpy() - g['zmask_%s/final_Z_sub' % objective])
    assert np.median(d) < 1e-4 and np.mean(d > 1e-2) < 0.01
    assert float((z[:, :, :16] - z0[:, :, :16]).abs().max()) < 1e-6          # outside Z_mask nothing moved (tanh(arctanh(z0)) round trip)


@pytest.mark.parametrize('name', ['plain', 'masked'])
def test_soft_histogram_loss_of_the_oracle_matches_the_reference(name):
    g = np.load(os.path.join(GOLDEN, 'soft_histogram.npz'))
    desired = seeded_uniform((1, 3, 40, 36), 1101)
    cur = (seeded_uniform((2, 3, 40, 36), 1102) ** 2).requires_grad_(True)
    mask = torch.from_numpy(g['masked/mask']) if name == 'masked' else None
    loss = cao.soft_hist_loss(cur, desired[0], 64, 0.0, 1.0, 2e-3, mask)
    loss.backward()
    assert abs(float(loss) - float(g[name + '/loss'])) < 1e-5 * float(g[name + '/loss'])
    np.testing.assert_allclose(cur.grad.numpy(), g[name + '/grad'], rtol=1e-3, atol=1e-9)
