"""Size-independent properties of the HIP path at BASELINE.json's FULL configs[1] size (RRDB-23 x4 + CEM, 32 x 3 x 128 x 128), where the
CPU oracle is too slow to be the checker: consistency, idempotence and linearity of the CEM projection,
identity of zero-weight residual blocks, exactness of batch sharding (the data-parallel decomposition), adjointness of the data
gradient.  Run with -m gpu."""
import numpy as np
import pytest
import torch

from oracle import cem_oracle as co
from oracle.weights import seeded_uniform

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def full():
    import contextlib
    import io
    import CEM.CEMnet as C
    import models.modules.architecture as arch
    import models.networks as networks
    torch.manual_seed(0)
    cem = C.CEMnet(C.Get_CEM_Conf(4))
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                       latent_input=None, num_latent_channels=0)
    G = cem.WrapArchitecture_PyTorch(net)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)
    G = G.to(DEV).eval()
    x = torch.rand(32, 3, 128, 128, generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad():
        y = G(x)
    return cem, G, x, y


def test_full_size_output_is_consistent_with_the_lr_input(full):
    """D(out) == x in the interior (target < 1e-5), with the HIP downsampler on the whole batch and with the CPU oracle's on one image."""
    cem, G, x, y = full
    assert y.shape == (32, 3, 512, 512) and bool(torch.isfinite(y).all())
    m = int(cem.invalidity_margins_LR)
    with torch.no_grad():
        d = G.DownscaleOP(y)
    assert float(((d - x)[..., m:-m, m:-m] ** 2).mean().sqrt()) < 1e-5
    d0 = co.downscale_op(y[:1].cpu(), co.CEMTaps(4))
    assert float((d0 - x[:1].cpu())[..., m:-m, m:-m].abs().max()) < 1e-5


def test_batch_sharding_is_exact(full):
    """The data-parallel decomposition: any shard of the batch, processed alone, gives bit-identical images (no cross-image state)."""
    cem, G, x, y = full
    with torch.no_grad():
        assert torch.equal(G(x[8:16]), y[8:16])
        assert torch.equal(G(x[31:32]), y[31:32])


def test_full_size_mixed_precision_tracks_the_fp32_class_path(full):
    """The benchmark's headline precision on the benchmark's workload: every one of the 32 images within 1e-4 (relative L2) of the
    split-bf16 path (both are ~3e-5 from the fp32 oracle, bench.py prints those on one image), consistent with its LR input to < 1e-5,
    finite, and sharding-exact.  Bar: 1e-3."""
    cem, G, x, y = full
    net = G.generated_image_model
    net.set_precision('mixed')
    try:
        with torch.no_grad():
            ym = G(x)
            assert bool(torch.isfinite(ym).all())
            per_image = ((ym - y).flatten(1).norm(dim=1) / y.flatten(1).norm(dim=1)).cpu().numpy()
            assert per_image.max() < 1e-4, per_image.max()
            m = int(cem.invalidity_margins_LR)
            assert float(((G.DownscaleOP(ym) - x)[..., m:-m, m:-m] ** 2).mean().sqrt()) < 1e-5
            assert torch.equal(G(x[5:9]), ym[5:9])
    finally:
        net.set_precision('split')


def test_projection_is_idempotent_and_affine_in_g(full):
    """CEM(x, .) of an image that is already consistent with x returns it (interior); out(x, g1 + g2) - out(x, g1) is the null-space
    component of g2, independent of x and g1 (the projector is affine in g)."""
    cem, G, x, y = full
    import CEM.CEMnet as C
    P = C.CEMnet(C.Get_CEM_Conf(4)).WrapArchitecture_PyTorch(generated_image=None).to(DEV).train()       # train mode: no pad / crop
    xs, ys = x[:4], y[:4]
    with torch.no_grad():
        again = P([xs, ys])
        m = int(cem.invalidity_margins_HR)
        assert float((again - ys)[..., m:-m, m:-m].abs().max()) < 2e-5
        g1, g2 = torch.rand_like(ys), torch.rand_like(ys)
        lhs = P([xs, g1 + g2]) - P([xs, g1])
        rhs = P([torch.zeros_like(xs), g2])
        assert float((lhs - rhs).abs().max()) < 2e-5


def test_zero_weight_blocks_are_identities():
    """An RDB with zero conv weights is the identity (block.py:235), so an RRDB of three such RDBs is x -> 0.2*x + x = 1.2*x
    (block.py:270) and the trunk of nb such RRDBs reduces to fea + LR_conv(1.2^nb * fea)."""
    import models.modules.architecture as arch
    from oracle import rrdb_oracle as ro
    from oracle.weights import fill_formula_weights
    net = arch.RRDBNet(3, 3, 64, 4, upscale=4, num_latent_channels=0)
    fill_formula_weights(net, gain=1.0)
    for r in range(4):
        for p in net.model[1].sub[r].parameters():
            p.data.zero_()
    x = seeded_uniform((2, 3, 20, 24), 701)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        y = net.to(DEV)(x.to(DEV)).cpu()
    # oracle on the same weights (zero blocks included) and the analytic statement about the trunk
    yo = ro.rrdb_forward(sd, x, 4, 4, 0, prefix='model')
    assert float((y - yo).norm() / yo.norm()) < 1e-4
    cap = {}
    ro.rrdb_forward(sd, x, 4, 4, 0, prefix='model', capture=cap)
    fea = torch.nn.functional.conv2d(x, sd['model.0.weight'], sd['model.0.bias'], padding=1)
    trunk = fea + torch.nn.functional.conv2d(1.2 ** 4 * fea, sd['model.1.sub.4.weight'], sd['model.1.sub.4.bias'], padding=1)
    assert float((cap['trunk'] - trunk).abs().max()) < 1e-4 * float(trunk.abs().max())


def test_data_gradient_is_the_adjoint_of_the_forward_conv():
    """<conv(a), b> == <a, conv^T(b)> for the linear part (a conv without activation), at a mid-size shape, both sides on the GPU."""
    from esr_hip import act as A
    a = seeded_uniform((4, 96, 37, 53), 801, -1, 1).to(DEV)
    b = seeded_uniform((4, 64, 37, 53), 802, -1, 1).to(DEV)
    w = (seeded_uniform((64, 96, 3, 3), 803, -1, 1) * 0.05).to(DEV)
    La = A.conv3x3_nchw(a, w, None, 1.0)
    Ltb = A.conv3x3_dgrad_nchw(b, w)
    lhs, rhs = float((La.double() * b.double()).sum()), float((a.double() * Ltb.double()).sum())
    assert abs(lhs - rhs) < 2e-5 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


def test_full_size_c5_x8_non_bicubic_is_consistent():
    """BASELINE configs[4] at its full size on one GPU: RRDB-23 x8, 'blurry_cubic_2.0' CEM kernel, batch 16 of 256x256 -> 2048x2048
    (the generator runs on 280x280 padded frames; ~60 GB of activation buffers): finite, and downsample-consistent with its input.
    Exercises the >2^31-element buffers and the 12-pixel margin / 45-tap kernels."""
    import contextlib
    import io
    import CEM.CEMnet as C
    from CEM.imresize_CEM import imresize
    import models.modules.architecture as arch
    import models.networks as networks
    if torch.cuda.get_device_properties(0).total_memory < 120 * 2 ** 30:
        pytest.skip('needs ~70 GB of device memory')
    imresize.kernels = {}
    torch.manual_seed(0)
    cem = C.CEMnet(C.Get_CEM_Conf(8), upscale_kernel='blurry_cubic_2.0')
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, upscale=8, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                       latent_input=None, num_latent_channels=0)
    G = cem.WrapArchitecture_PyTorch(net)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)
    G = G.to(DEV).eval()
    x = torch.rand(16, 3, 256, 256, generator=torch.Generator().manual_seed(6)).to(DEV)
    with torch.no_grad():
        y = G(x)
        assert y.shape == (16, 3, 2048, 2048) and bool(torch.isfinite(y).all())
        d = G.DownscaleOP(y)
    m = int(cem.invalidity_margins_LR)
    assert m == 12
    assert float(((d - x)[..., m:-m, m:-m] ** 2).mean().sqrt()) < 1e-5
    # the precision configs[4] names: fp16 operands, one MFMA per product.  Same size, same checks, plus the distance of the GENERATOR's
    # output from the fp32-class one on the same weights (stated tolerance for the one-plane fp16 mode: 5e-3 relative L2; measured 1.6e-3 on
    # this init, DESIGN.md 5.5 — the CEM output itself is dominated by the LR content and would hide it)
    with torch.no_grad():
        gs = net(x, pad=m)
    net.set_precision('f16')
    with torch.no_grad():
        yh = G(x)
        assert bool(torch.isfinite(yh).all())
        dh = G.DownscaleOP(yh)
        assert float(((dh - x)[..., m:-m, m:-m] ** 2).mean().sqrt()) < 1e-5
        assert torch.equal(G(x[4:8]), yh[4:8])                                   # shards of the 4-GPU run are exact
        gh = net(x, pad=m)
        err = float((gh - gs).norm() / gs.norm())
    assert err < 5e-3, err
    net.set_precision('split')
    del G, y, d, yh, dh, gs, gh
    torch.cuda.empty_cache()
    imresize.kernels = {}


def _scaled_weights(G, kind, gain):
    """Weight sets for the robustness sweep: the closed-form 'formula' weights at a given gain (gain 1.0 keeps activations O(1) through the
    stack, 2.0 lets them grow to ~1e9, 0.1 is the scale of the reference's training init), or a heavy-tailed set (the training init with 1 %
    of the weights multiplied by 30)."""
    import contextlib
    import io
    import models.networks as networks
    from oracle.weights import fill_formula_weights
    if kind == 'formula':
        fill_formula_weights(G, gain=gain)
        return
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in G.named_parameters():
            if n.endswith('.weight') and p.dim() == 4 and 'Filter_OP' not in n:
                mask = (torch.rand(p.shape, generator=g) < 0.01).to(p.device)
                p.mul_(torch.where(mask, torch.full_like(p, 30.0), torch.ones_like(p)))


@pytest.mark.parametrize('kind,gain', [('formula', 0.1), ('formula', 0.6), ('formula', 1.0), ('formula', 2.0), ('heavy', None)])
def test_headline_precision_against_the_oracle_over_weight_scales(full, kind, gain):
    """The benchmark's headline precision (split-bf16) at the benchmark's size — RRDB-23 x4 on one padded 148 x 148 frame — against the fp32
    CPU oracle over weight scales: every product carries 16-bit operands and the error is relative to each product, so the margin does not
    depend on the scale of the weights (the judge's sweep for any precision that wants to be the headline: rel-max <= 3e-4).  'mixed' is run
    on the same weights for the record: its one-plane dense-block operands cost it accuracy as the gain grows, and fp16 cannot hold the
    activations of the gain-2.0 set at all (they reach 1e9)."""
    from oracle import rrdb_oracle as ro
    from esr_hip import EsrError
    cem, G, x, _ = full
    net = G.generated_image_model
    backup = {k: v.detach().clone() for k, v in G.state_dict().items()}
    try:
        _scaled_weights(G, kind, gain)
        sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
        xp = torch.nn.functional.pad(x[:1].cpu(), (10,) * 4, mode='replicate')
        torch.set_num_threads(16)
        with torch.no_grad():
            ref = ro.rrdb_forward(sd, xp, 23, 4, 0, prefix='generated_image_model.model')
        res = {}
        for prec in ('split', 'mixed'):
            net.set_precision(prec)
            with torch.no_grad():
                got = net(x[:1], pad=10).cpu()
            if prec == 'mixed' and kind == 'formula' and gain == 2.0:
                # fp16 cannot hold these activations: the kernels' range watch names the first layer that left the range
                with pytest.raises(EsrError, match='reached fp16'):
                    net.check_range()
                net.check_range()                 # the verdict is consumed: nothing pending
            else:
                net.check_range()                 # everything else stays in range
            res[prec] = (float((got - ref).abs().max() / ref.abs().max()), float((got - ref).norm() / ref.norm()))
        print('%s gain %s: |out|max %.3g; split rel-max %.2e rel-l2 %.2e; mixed rel-max %.2e rel-l2 %.2e' % ((kind, gain, float(ref.abs().max())) + res['split'] + res['mixed']))
        assert res['split'][0] <= 3e-4 and res['split'][1] <= 1e-4, res['split']
        if not (kind == 'formula' and gain == 2.0):
            assert res['mixed'][0] <= 1e-3, res['mixed']          # the 1e-3 bar of the north star, with less margin as the gain grows
    finally:
        net.set_precision('split')
        G.load_state_dict(backup)
