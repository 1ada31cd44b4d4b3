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
