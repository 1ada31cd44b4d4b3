"""GPU tests (-m gpu) of the callers around the hot path: the SRRaGANModel generator step (forward, CEM, loss crop, backward,
all-reduce hook, Adam) and the Z-optimisation loop on the real generator, both against autograd through the CPU oracle run with
the same weights and inputs (reference: models/SRRaGAN_model.py:280-499 G side, Z_optimization.py:566-801)."""
import numpy as np
import pytest
import torch

from oracle import cem_oracle as co
from oracle import rrdb_oracle as ro
from oracle.check_golden import rel_l2
from oracle.weights import fill_formula_weights, seeded_uniform
from test_host_api import _opt

pytestmark = pytest.mark.gpu


def _model(nb, lat, is_train):
    import models
    opt = _opt(nb=nb, lat=lat, cem=True, is_train=is_train)
    opt['gpu_ids'] = [0]
    m = models.create_model(opt)
    fill_formula_weights(m.netG, gain=1.0)
    return m


class _OracleModel:
    """The same wrapper semantics on the CPU oracle (differentiable torch ops), for running the product's Z_optimizer against."""
    def __init__(self, sd, nb, lat, sf=4):
        self.sd, self.nb, self.lat, self.sf = sd, nb, lat, sf
        self.t = co.CEMTaps(sf)
        self.device = torch.device('cpu')
        self.num_latent_channels = lat
        self.opt = {'scale': sf}
        self.netG = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in sd.values()])   # for the grad-flag bookkeeping

    def Output_Batch(self, within_0_1):
        return torch.clamp(self.fake_H, 0, 1) if within_0_1 else self.fake_H

    def feed_data(self, data, need_GT=True, **kw):
        self.lr, self.z = data['LR'], data['Z']
        if self.lr.size(0) == 1 and self.z.size(0) > 1:
            self.lr = self.lr.expand(self.z.size(0), -1, -1, -1)

    def forward(self, eval_mode):
        t, x, z = self.t, self.lr, self.z
        if eval_mode:
            x = torch.nn.functional.pad(x, (t.margins_LR,) * 4, mode='replicate')
            z = torch.nn.functional.pad(z, (t.margins_HR,) * 4, mode='replicate')
        xin = torch.cat([z.reshape(z.size(0), self.lat * self.sf ** 2, x.shape[2], x.shape[3]), x], 1)
        gen = ro.rrdb_forward(self.sd, xin, self.nb, self.sf, self.lat, prefix='generated_image_model.model')
        return co.cem_combine(x, gen, t, crop=eval_mode)

    def test(self, prevent_grads_calc=True, **kw):
        if prevent_grads_calc:
            with torch.no_grad():
                self.fake_H = self.forward(True)
        else:
            self.fake_H = self.forward(True)


def test_generator_step_matches_oracle_autograd():
    nb, lat = 1, 1
    m = _model(nb, lat, is_train=True)
    sd = {k: v.detach().cpu().clone() for k, v in m.netG.state_dict().items()}
    lr = seeded_uniform((2, 3, 24, 26), 201)
    hr = seeded_uniform((2, 3, 96, 104), 202)
    z = seeded_uniform((2, lat, 96, 104), 203, -1, 1)
    for _ in range(2):       # without a discriminator the first call is idle, as in the reference (SRRaGAN_model.py:338-339)
        m.feed_data({'LR': lr, 'HR': hr, 'Z': z})
        m.optimize_parameters()
    loss_gpu = m.get_current_log()['l_g_pix']
    # oracle: same graph on CPU
    names = [k for k, v in m.netG.named_parameters() if v.requires_grad]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    sd_o = dict(sd); sd_o.update(params)
    t = co.CEMTaps(4)
    xin = torch.cat([z.reshape(2, lat * 16, 24, 26), lr], 1)
    gen = ro.rrdb_forward(sd_o, xin, nb, 4, lat, prefix='generated_image_model.model')
    out = co.cem_combine(lr, gen, t, crop=False)
    mh = t.margins_HR
    assert mh == int(m.CEM_net.invalidity_margins_HR)
    loss = (out[..., mh:-mh, mh:-mh] - hr[..., mh:-mh, mh:-mh]).abs().mean()
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, betas=(0.9, 0.999))
    loss.backward()
    opt.step()
    assert abs(loss_gpu - loss.item()) < 1e-4 * abs(loss.item())
    # (like the reference, :329-333, the wrapper keeps the generator output with the CEM's invalidity frame cropped)
    assert rel_l2(m.fake_H.detach().cpu().numpy(), out[..., mh:-mh, mh:-mh].detach().numpy()) < 1e-4
    got = dict(m.netG.named_parameters())
    gmax = max(float(np.abs(params[k].grad.numpy()).max()) for k in names)
    checked = 0
    for k in names:
        g_ref = params[k].grad.numpy()
        g = got[k].grad.cpu().numpy()
        if np.abs(g_ref).max() < 1e-5 * gmax:
            # e.g. the last conv's bias: the CEM projects a constant offset of the generator output away, so this gradient is
            # analytically zero and both sides hold rounding noise (and Adam normalises that noise to +-lr)
            assert np.abs(g).max() < 1e-4 * gmax, k
            continue
        checked += 1
        # a weight gradient sums over every pixel, so each LeakyReLU sign flip (see test_gpu_backward.assert_grad_close) perturbs
        # ALL its elements a little instead of a few elements a lot: bound the relative L2 error
        assert rel_l2(g, g_ref) < 1e-2, (k, rel_l2(g, g_ref))
        # Adam's first step moves every weight by ~lr*sign(g): both sides must have moved the same way almost everywhere
        d = np.abs(got[k].detach().cpu().numpy() - params[k].detach().numpy())
        assert np.mean(d > 2e-5) < 2e-2, (k, np.mean(d > 2e-5))
    assert checked >= len(names) - 2
    assert m.step == 2 and m.generator_changed


def test_model_test_eval_path_and_scalar_z():
    nb, lat = 1, 1
    m = _model(nb, lat, is_train=False)
    sd = {k: v.detach().cpu().clone() for k, v in m.netG.state_dict().items()}
    lr = seeded_uniform((1, 3, 13, 17), 211)
    m.feed_data({'LR': lr, 'Z': 0.25}, need_GT=False)           # scalar Z broadcasts to B x lat x 4h x 4w (reference :254-262)
    m.test()
    om = _OracleModel(sd, nb, lat)
    om.feed_data({'LR': lr, 'Z': 0.25 * torch.ones(1, lat, 52, 68)})
    om.test()
    assert m.fake_H.shape == (1, 3, 52, 68)
    assert rel_l2(m.fake_H.cpu().numpy(), om.fake_H.numpy()) < 1e-4
    assert tuple(m.GetLatent().shape) == (1, lat, 52, 68)
    # CEM guarantee on the product output: downscaling it gives back the LR image (interior)
    lr_back = co.downscale_op(m.fake_H.cpu(), co.CEMTaps(4))
    scale = max(1.0, float(m.fake_H.abs().max()))
    assert float((lr_back - lr)[..., 4:-4, 4:-4].abs().max()) < 1e-5 * scale


@pytest.mark.parametrize('objective', ['max_STD', 'TV'])
def test_z_optimizer_on_real_generator_matches_oracle_run(objective):
    from Z_optimization import Z_optimizer
    nb, lat, B, iters = 1, 3, 2, 4
    m = _model(nb, lat, is_train=False)
    sd = {k: v.detach().cpu().clone() for k, v in m.netG.state_dict().items()}
    lr = seeded_uniform((1, 3, 10, 12), 221)
    z0 = seeded_uniform((B, lat, 40, 48), 222, -0.5, 0.5)
    res, flags0 = [], [p.requires_grad for p in m.netG.parameters()]
    for model in (m, _OracleModel(sd, nb, lat)):
        dev = model.device
        model.feed_data({'LR': lr.expand(B, -1, -1, -1).to(dev), 'Z': z0.to(dev)}, need_GT=False)
        model.test()
        zo = Z_optimizer(objective=objective, Z_size=[40, 48], model=model, Z_range=1, max_iters=iters, data={'LR': lr.to(dev)},
                         initial_LR=0.02, batch_size=B, initial_Z=z0.to(dev))
        Z = zo.optimize()
        res.append((Z.cpu().numpy(), np.array(zo.loss_values)))
    flags1 = [p.requires_grad for p in m.netG.parameters()]
    (Zg, Lg), (Zc, Lc) = res
    assert len(Lg) == len(Lc)
    np.testing.assert_allclose(Lg, Lc, rtol=2e-3, atol=1e-6)
    assert np.median(np.abs(Zg - Zc)) < 2e-4 and rel_l2(Zg, Zc) < 5e-2
    assert float(np.abs(Zg - z0.numpy()).max()) > 1e-3          # it did move
    assert flags0 == flags1 and any(flags1)                     # the model's grad flags are restored after the search


def test_z_optimizer_training_mode_leaves_a_differentiable_forward():
    """The train-time use of the Z search (train.py's optimized-Z step; Z_optimization.py:603,788-795): the model is in training mode,
    the loss is taken on the HR_unpadder crop, Z starts random, and after the search one more forward with the FOUND Z is left on the
    model with the generator's parameters attached to the graph, ready for optimize_parameters()-style backward."""
    from Z_optimization import Z_optimizer
    m = _model(1, 3, is_train=True)
    lr = seeded_uniform((2, 3, 24, 26), 231)
    m.feed_data({'LR': lr, 'HR': seeded_uniform((2, 3, 96, 104), 232), 'Z': torch.zeros(2, 3, 96, 104)})
    desired = seeded_uniform((2, 3, 96, 104), 233)
    unpad = m.CEM_net.HR_unpadder
    zo = Z_optimizer(objective='l1', Z_size=[96, 104], model=m, Z_range=1, max_iters=3, data={'LR': lr, 'desired': unpad(desired)},
                     initial_LR=0.05, batch_size=2, HR_unpadder=unpad)
    Z = zo.optimize()
    assert Z.shape == (2, 3, 96, 104) and float(Z.abs().max()) <= 1.0 and not Z.requires_grad
    assert len(zo.loss_values) == 3 and all(np.isfinite(zo.loss_values))
    assert m.fake_H.requires_grad                                   # the final forward carries the generator's graph
    assert all(p.requires_grad for n, p in m.netG.named_parameters() if 'Filter_OP' not in n)
    m.fake_H.mean().backward()
    assert m.netG.generated_image_model.model[0].weight.grad is not None


def test_z_search_in_mixed_precision_follows_the_fp32_path():
    """'mixed' (fp16 forward) supports the Z search: the data gradient runs in the same fp16 format (power-of-two scaled, gradient of the
    residual stream stored hi+lo), the saved fp16 activations serve as LeakyReLU' masks.  The search must track the fp32-path search (same losses to 1e-3, same Z to Adam-step accuracy)."""
    from Z_optimization import Z_optimizer
    nb, lat, B, iters = 2, 3, 2, 4
    res = []
    for prec in ('split', 'mixed'):
        m = _model(nb, lat, is_train=False)
        m.netG.generated_image_model.set_precision(prec)
        lr = seeded_uniform((1, 3, 10, 12), 241)
        z0 = seeded_uniform((B, lat, 40, 48), 242, -0.5, 0.5)
        dev = m.device
        m.feed_data({'LR': lr.expand(B, -1, -1, -1).to(dev), 'Z': z0.to(dev)}, need_GT=False)
        m.test()
        zo = Z_optimizer(objective='max_STD', Z_size=[40, 48], model=m, Z_range=1, max_iters=iters, data={'LR': lr.to(dev)}, initial_LR=0.02,
                         batch_size=B, initial_Z=z0.to(dev))
        Z = zo.optimize()
        res.append((Z.cpu().numpy(), np.array(zo.loss_values)))
    (Zs, Ls), (Zm, Lm) = res
    np.testing.assert_allclose(Lm, Ls, rtol=2e-3, atol=1e-6)
    # Adam normalises every component's step to ~lr whatever the gradient's size, so 1e-4-level gradient differences (these are the
    # high-gain formula weights, mixed's worst case) move individual Z entries by a fraction of a step: 4 steps of 0.02 -> median 5e-4
    assert np.median(np.abs(Zm - Zs)) < 2e-3 and rel_l2(Zm, Zs) < 5e-2


def test_data_edits_need_invalidate_packs_and_get_it():
    """Writes through `.data` do not bump torch's version counter, which the engine's weight packs watch: the documented
    remedy RRDBNet.invalidate_packs() makes the next forward see them; in-place ops on the parameter itself are picked up automatically."""
    import models.modules.architecture as arch
    net = arch.RRDBNet(3, 3, 64, 1, upscale=2, num_latent_channels=0).cuda()
    fill_formula_weights(net, gain=1.0)
    x = seeded_uniform((1, 3, 8, 8), 251).cuda()
    with torch.no_grad():
        y0 = net(x).clone()
        w = net.model[0].weight
        w.data.mul_(2.0)
        net.invalidate_packs()
        y1 = net(x).clone()
        w.mul_(0.5)                          # versioned in-place op: no call needed
        y2 = net(x).clone()
    assert not torch.allclose(y0, y1) and torch.equal(y0, y2)
