"""GPU tests (-m gpu) of the Z-objective kernels: the soft-histogram HIP kernels behind SoftHistogramLoss (csrc/esr_zobj.hip) against values the
reference's class produced (fixture F11, oracle/gen_golden.py::gen_F11 from codes/Z_optimization.py:24-230), and the 'hist' objective of
Z_optimizer end to end."""
import os

import numpy as np
import pytest
import torch

from oracle.weights import fill_formula_weights, seeded_uniform

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('name', ['plain', 'masked'])
def test_soft_histogram_loss_matches_reference(name):
    from Z_optimization import SoftHistogramLoss
    g = np.load(os.path.join(GOLDEN, 'soft_histogram.npz'))
    desired = seeded_uniform((1, 3, 40, 36), 1101).cuda()
    mask = torch.from_numpy(g['masked/mask']).cuda() if name == 'masked' else None
    loss_fn = SoftHistogramLoss(bins=64, min=0, max=1, desired_hist_image=[desired], desired_hist_image_mask=[None], input_im_HR_mask=mask, gray_scale=True,
                                patch_size=1, temperature=2e-3)
    np.testing.assert_allclose(loss_fn.desired_hists_list[0].cpu().numpy(), g[name + '/desired_hist'], rtol=1e-4, atol=1e-8)
    cur = (seeded_uniform((2, 3, 40, 36), 1102) ** 2).cuda().requires_grad_(True)
    loss = loss_fn(cur)
    loss.backward()
    assert abs(float(loss) - float(g[name + '/loss'])) < 1e-3 * float(g[name + '/loss'])
    np.testing.assert_allclose(cur.grad.cpu().numpy(), g[name + '/grad'], rtol=2e-3, atol=2e-7)


def test_soft_histogram_kernel_at_gui_size():
    """512 x 512 pixels x 256 bins (the GUI's setting, Z_optimization.py:536): the reference's n x K float64 matrix would be 0.5 GB; the kernel
    result must equal a chunked float64 evaluation of the same formula, forward and backward."""
    from esr_hip import zobj
    v = torch.rand(512 * 512, generator=torch.Generator().manual_seed(3)).cuda().requires_grad_(True)
    h = zobj.soft_histogram(v, 256, 0.0, 1.0, 5e-4)
    w = torch.linspace(0.5, 1.5, 256, dtype=torch.float64, device='cuda')
    (h * w).sum().backward()
    c = torch.linspace(0, 1, 256, dtype=torch.float64, device='cuda').view(1, -1)
    vd = v.detach().double().clone().requires_grad_(True)
    ref = torch.zeros(256, dtype=torch.float64, device='cuda')
    for chunk in vd.split(32768):
        x = chunk.view(-1, 1)
        d = torch.min(torch.min((x - c).abs(), (x - c - 1).abs()), (x - c + 1).abs())
        ref = ref + torch.exp(-((d + 1e-7) ** 2) / 5e-4).sum(0)
    ref = ref / vd.numel()
    (ref * w).sum().backward()
    np.testing.assert_allclose(h.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5)
    gref = vd.grad.float().cpu().numpy()
    np.testing.assert_allclose(v.grad.cpu().numpy(), gref, rtol=1e-3, atol=1e-5 * np.abs(gref).max())      # (a few entries are differences of nearly equal bin terms)


def test_hist_objective_of_z_optimizer_reduces_the_divergence():
    import models
    from Z_optimization import Z_optimizer
    from test_gpu_callers_f7 import product_opt
    m = models.create_model(product_opt(False))
    fill_formula_weights(m.netG, gain=0.5)
    lr = seeded_uniform((1, 3, 24, 28), 920).to(m.device)
    z0 = seeded_uniform((2, 3, 96, 112), 921, -0.3, 0.3).to(m.device)
    m.feed_data({'LR': lr.expand(2, -1, -1, -1).clone(), 'Z': z0.clone()}, need_GT=False)
    m.test()
    desired = (m.fake_H[:1].detach().clamp(0, 1) * 0.8 + 0.1)                     # a lower-contrast version of the current output
    zo = Z_optimizer(objective='hist', Z_size=[96, 112], model=m, Z_range=1, max_iters=4, data={'LR': lr.expand(2, -1, -1, -1).clone(), 'desired': [desired]},
                     initial_Z=z0.clone(), initial_LR=0.1, batch_size=2)
    z = zo.optimize()
    assert all(np.isfinite(zo.loss_values)) and zo.loss_values[-1] < zo.loss_values[0] and float((z - z0).abs().max()) > 1e-3


def _ref_stats(x, mask, clamp01, kind):
    v = x.double()
    if clamp01:
        v = torch.clamp(v, 0, 1)
    if mask is not None:
        v = v * mask.double()
    if kind == 0:
        return torch.std(v, dim=(1, 2, 3))
    if kind == 1:
        return (v[:, :, :, :-1] - v[:, :, :, 1:]).abs().mean(dim=(1, 2, 3)) + (v[:, :, :-1, :] - v[:, :, 1:, :]).abs().mean(dim=(1, 2, 3))
    ix = (v[..., :, 1:] - v[..., :, :-1])[..., :-1, :]
    iy = (v[..., 1:, :] - v[..., :-1, :])[..., :, :-1]
    return torch.stack([(ix * ix).mean(dim=(1, 2, 3)), (iy * iy).mean(dim=(1, 2, 3)), (ix * iy).mean(dim=(1, 2, 3))], 0)


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('masked,clamp01', [(False, False), (True, True)])
def test_image_statistic_kernels_match_the_torch_expressions(kind, masked, clamp01):
    """Masked STD (Z_optimization.py:383-388), TV_Loss (:324-326) and the structure tensor of FilterLoss (loss.py:49-62,141-151): value and
    gradient of the single-pass kernels against the reference's torch expressions evaluated in float64."""
    from esr_hip import zobj
    from oracle.weights import seeded_uniform
    x = (seeded_uniform((3, 3, 37, 45), 1200 + kind) * 1.6 - 0.3).cuda()                 # some values outside [0, 1]: the clamp's gradient mask matters
    mask = (seeded_uniform((37, 45), 1210) > 0.35).float().cuda() if masked else None
    if kind == 2 and masked:
        pytest.skip('the structure tensor is taken of the whole image')
    fn = {0: zobj.image_std, 1: zobj.tv_loss}.get(kind)
    xa = x.clone().requires_grad_(True)
    got = fn(xa, mask, clamp01) if fn else zobj.structure_tensor(xa)
    xr = x.clone().requires_grad_(True)
    ref = _ref_stats(xr, mask, clamp01, kind)
    torch.testing.assert_close(got.double(), ref, rtol=2e-6, atol=1e-9)
    w = seeded_uniform(tuple(ref.shape), 1220).cuda().double() + 0.5
    (got.double() * w).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(xa.grad.double(), xr.grad.double(), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('code', ['SVDinNormedOut_structure_tensor', 'structure_tensor'])
def test_filter_loss_on_the_gpu_matches_the_reference(code):
    """FilterLoss (reference loss.py:27-209) on GPU tensors — structure tensor and its gradient by the HIP kernels — against fixture F10
    (values of three consecutive calls of the reference's class, and d loss / d SR)."""
    import numpy as np
    import os
    from models.modules.loss import FilterLoss
    from oracle.weights import seeded_uniform
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'filter_loss.npz'))
    fl = FilterLoss(latent_channels=code)
    for call in range(3):
        sr = seeded_uniform((4, 3, 24, 20), 1000 + call).cuda().requires_grad_(True)
        hr = seeded_uniform((4, 3, 24, 20), 1010 + call).cuda()
        z = (seeded_uniform((4, 3, 1, 1), 1020 + call, -1.0, 1.0) * torch.ones(4, 3, 24, 20)).cuda()
        loss = fl({'SR': sr, 'HR': hr, 'Z': z})
        # |measured - target| of O(0.1) quantities summed in fp32: absolute agreement 5e-7 (the reference's own CPU sums differ by as much)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g['%s/call%d' % (code, call)], rtol=1e-4, atol=5e-7)
    loss.sum().backward()
    np.testing.assert_allclose(sr.grad.cpu().numpy(), g[code + '/dSR'], rtol=2e-4, atol=1e-7)
