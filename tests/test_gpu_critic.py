"""GPU tests (-m gpu) of the critic on the library's kernels (esr_hip/critic.py, csrc/esr_critic.hip) against the SAME nn.Module executed
by torch (fp32, MIOpen) — which tests/test_gpu_callers_f7.py pins to the reference's Discriminator_VGG_128: logits, first-order gradients
(parameters and input) and the WGAN-GP step's parameter gradients, which differentiate through the backward pass."""
import numpy as np
import pytest
import torch

from oracle.weights import fill_formula_weights, seeded_uniform

pytestmark = pytest.mark.gpu


def make_D(size=64, gain=1.0):
    import models.modules.architecture as arch
    torch.manual_seed(0)
    netD = arch.Discriminator_VGG_128(in_nc=3, base_nf=64, norm_type='batch', act_type='leakyrelu', mode='CNA', input_patch_size=size)
    fill_formula_weights(netD, gain=gain)
    return netD.cuda().train()


def s2d(x):
    """NCHW -> the space-to-depth channel order of esr_bn_apply (csrc/esr_critic.hip): channel ((16 (g // 4) + 4 s + g % 4) * 8 + e)."""
    B, Cc, H, W = x.shape
    return x.view(B, Cc // 32, 4, 8, H // 2, 2, W // 2, 2).permute(0, 1, 5, 7, 2, 3, 4, 6).reshape(B, 4 * Cc, H // 2, W // 2).contiguous()


def d2s(xs):
    B, C4, H2, W2 = xs.shape
    return xs.view(B, C4 // 128, 2, 2, 4, 8, H2, W2).permute(0, 1, 4, 5, 6, 2, 7, 3).reshape(B, C4 // 4, 2 * H2, 2 * W2)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_strided_and_sliced_convs_match_torch():
    """One 4x4 stride-2 block (space-to-depth + embedded 3x3) and one wide 3x3 block (output slices) against F.conv2d in float64."""
    from esr_hip import critic as K
    netD = make_D()
    eng = K.CriticEngine(netD, 'split')
    eng.refresh()
    x = seeded_uniform((3, 3, 32, 48), 5).cuda()
    t = K._PackIn.apply(x, 2)
    with torch.no_grad():
        L0, L1, L2 = eng.layers[:3]
        y0 = eng.conv_fwd(L0, t)
        ref0 = torch.nn.functional.conv2d(x.double(), L0.conv.weight.double(), L0.conv.bias.double(), padding=1)
        got0 = K._UnpackOut.apply(y0, 64)
        assert rel(got0, ref0) < 1e-4
        z0 = K._BNAct.apply(eng, L0, y0, None, None, True, True)                       # LeakyReLU, stored space-to-depth
        y1 = eng.conv_fwd(L1, z0)
        a0 = torch.nn.functional.leaky_relu(got0.double(), 0.2)
        ref1 = torch.nn.functional.conv2d(a0, L1.conv.weight.double(), L1.conv.bias.double(), stride=2, padding=1)
        got1 = K._UnpackOut.apply(y1, 64)
        assert got1.shape == ref1.shape and rel(got1, ref1) < 1e-4
        z1 = K._BNAct.apply(eng, L1, y1, L1.bn.weight, L1.bn.bias, False, True)        # BatchNorm (batch statistics) + LeakyReLU
        bn = torch.nn.functional.batch_norm(got1.double(), None, None, L1.bn.weight.double(), L1.bn.bias.double(), training=True, eps=L1.bn.eps)
        a1 = torch.nn.functional.leaky_relu(bn, 0.2)
        assert rel(K._UnpackOut.apply(z1, 64), a1) < 1e-4
        y2 = eng.conv_fwd(L2, z1)                                                     # 64 -> 128: two output slices in one launch
        ref2 = torch.nn.functional.conv2d(a1, L2.conv.weight.double(), L2.conv.bias.double(), padding=1)
        assert rel(K._UnpackOut.apply(y2, 128), ref2) < 1e-4


def test_backward_and_double_backward_kernels_match_float64_autograd():
    """Every piece of the critic's backward pass and of the backward of that pass, one at a time, against torch autograd in float64 — the
    strong statement (<= 2e-5, the operands carry 16 significand bits); the end-to-end test below can only make robust ones."""
    import torch.nn.functional as F
    from esr_hip import critic as K
    netD = make_D()
    eng = K.CriticEngine(netD, 'split')
    eng.refresh()
    L1, L2 = eng.layers[1], eng.layers[2]
    B, H, W = 3, 32, 48
    tol = 2e-5
    # 3x3 stride-1 block, 64 -> 128 (two output slices): forward, data gradient, weight / bias gradient
    x, dy = seeded_uniform((B, 64, H, W), 1).cuda() - 0.5, seeded_uniform((B, 128, H, W), 2).cuda() - 0.5
    xa, dya = K._PackIn.apply(x, 2), K._PackIn.apply(dy, 2)
    xr, w = x.double().requires_grad_(True), L2.conv.weight.double().detach().requires_grad_(True)
    F.conv2d(xr, w, None, padding=1).backward(dy.double())
    with torch.no_grad():
        assert rel(K._UnpackOut.apply(eng.conv_dgrad(L2, dya), 64), xr.grad) < tol
        dw, db = eng.conv_wgrad(L2, dya, xa)
        assert rel(dw, w.grad) < tol and rel(db, dy.double().sum((0, 2, 3))) < tol
    # 4x4 stride-2 block on the space-to-depth input
    x, dy = seeded_uniform((B, 64, H, W), 3).cuda() - 0.5, seeded_uniform((B, 64, H // 2, W // 2), 4).cuda() - 0.5
    xs = s2d(x)
    xa, dya = K._PackIn.apply(xs, 2), K._PackIn.apply(dy, 2)
    xr, w = x.double().requires_grad_(True), L1.conv.weight.double().detach().requires_grad_(True)
    F.conv2d(xr, w, None, stride=2, padding=1).backward(dy.double())
    with torch.no_grad():
        gx = d2s(K._UnpackOut.apply(eng.conv_dgrad(L1, dya), 256))
        assert rel(gx, xr.grad) < tol
        assert rel(eng.conv_wgrad(L1, dya, xa)[0], w.grad) < tol
    # BatchNorm (batch statistics) + LeakyReLU: gradient, and gradient of the gradient (what the penalty differentiates through)
    y, dz, u = seeded_uniform((B, 128, H, W), 5).cuda() * 3 - 1, seeded_uniform((B, 128, H, W), 6).cuda() - 0.5, seeded_uniform((B, 128, H, W), 7).cuda() - 0.5
    yr = y.double().requires_grad_(True)
    gr, br = L2.bn.weight.double().detach().requires_grad_(True), L2.bn.bias.double().detach().requires_grad_(True)
    zr = F.leaky_relu(F.batch_norm(yr, None, None, gr, br, training=True, eps=L2.bn.eps), 0.2)
    dzr = dz.double().requires_grad_(True)
    dyr, dgr, dbr = torch.autograd.grad(zr, [yr, gr, br], dzr, create_graph=True)
    gy, gg, gdz = torch.autograd.grad(dyr, [yr, gr, dzr], u.double())
    ya = K._PackIn.apply(y, 2).requires_grad_(True)
    gp_, bp_ = L2.bn.weight.detach().clone().requires_grad_(True), L2.bn.bias.detach().clone().requires_grad_(True)
    za = K._BNAct.apply(eng, L2, ya, gp_, bp_, False, True)
    dza = K._PackIn.apply(dz, 2).requires_grad_(True)
    dya, dga, dba = torch.autograd.grad(za, [ya, gp_, bp_], dza, create_graph=True)
    assert rel(K._UnpackOut.apply(dya.detach(), 128), dyr.detach()) < tol and rel(dga.detach(), dgr.detach()) < tol and rel(dba.detach(), dbr.detach()) < tol
    gya, gga, gdza = torch.autograd.grad(dya, [ya, gp_, dza], K._PackIn.apply(u, 2))
    assert rel(K._UnpackOut.apply(gya, 128), gy) < tol and rel(gga, gg) < tol and rel(K._UnpackOut.apply(gdza, 128), gdz) < tol


@pytest.mark.parametrize('precision,tol', [('split', 2e-5), ('bf16', 1e-4)])
def test_first_layer_weight_gradient_runs_the_gathered_tile_form(precision, tol):
    """The critic's first conv has a 3-channel MAIN input: its weight gradient runs the 27 (tap, channel) columns as one MFMA tile (csrc/esr_bwd.hip,
    main_as_latk) and the same workgroups own the bias gradient.  Against float64 autograd, as a single launch (partial sums + fold) and inside
    the batched launch of all layers' gradients (which takes the space-to-depth kernel flavour)."""
    import torch.nn.functional as F
    from esr_hip import critic as K
    netD = make_D()
    eng = K.CriticEngine(netD, precision)
    eng.refresh()
    L0 = eng.layers[0]
    B, H, W = 5, 40, 72                       # ragged 8 x 32 pixel tiles
    x, dy = seeded_uniform((B, 3, H, W), 11).cuda() - 0.5, seeded_uniform((B, 64, H, W), 12).cuda() - 0.5
    xa, dya = K._PackIn.apply(x, eng.planes), K._PackIn.apply(dy, eng.planes)
    xs, dys = K._UnpackOut.apply(xa, 3).double(), K._UnpackOut.apply(dya, 64).double()          # the operands as stored
    w = L0.conv.weight.double().detach().requires_grad_(True)
    F.conv2d(xs, w, None, padding=1).backward(dys)
    with torch.no_grad():
        dw, db = eng.conv_wgrad(L0, dya, xa)
        assert dw.shape == (64, 3, 3, 3) and rel(dw, w.grad) < tol and rel(db, dys.sum((0, 2, 3))) < tol
    # the batched launch: this layer next to a space-to-depth one (sizes the critic's five stride-2 convs accept)
    B, H, W = 2, 64, 96
    x, dy = seeded_uniform((B, 3, H, W), 15).cuda() - 0.5, seeded_uniform((B, 64, H, W), 16).cuda() - 0.5
    xa, dya = K._PackIn.apply(x, eng.planes), K._PackIn.apply(dy, eng.planes)
    xs, dys = K._UnpackOut.apply(xa, 3).double(), K._UnpackOut.apply(dya, 64).double()
    w = L0.conv.weight.double().detach().requires_grad_(True)
    F.conv2d(xs, w, None, padding=1).backward(dys)
    bs = K._BufSet(eng, B, 3, H, W, x.device)
    bs.t0.copy_(xa); bs.dy[0].copy_(dya)
    z0 = seeded_uniform((B, 256, H // 2, W // 2), 13).cuda() - 0.5
    dy1 = seeded_uniform((B, 64, H // 2, W // 2), 14).cuda() - 0.5
    bs.z[0].copy_(K._PackIn.apply(z0, eng.planes)); bs.dy[1].copy_(K._PackIn.apply(dy1, eng.planes))
    with torch.no_grad():
        out = K._WgradSet(eng, bs, [(eng.layers[0], bs.dy[0], bs.t0), (eng.layers[1], bs.dy[1], bs.z[0])]).run()
        assert rel(out[0][0], w.grad) < tol and rel(out[0][1], dys.sum((0, 2, 3))) < tol
        dw1, _ = eng.conv_wgrad(eng.layers[1], bs.dy[1], bs.z[0])
        assert rel(out[1][0], dw1) < 1e-6


def stock_losses(netD, real, fake, pt, gp_w=10.0):
    pr, pf = netD(real), netD(fake)
    interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
    crit = netD(interp)
    g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
    gp = gp_w * ((g.reshape(g.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
    return pr, pf, gp, g


@pytest.mark.parametrize('size,batch', [(64, 4), (128, 2)])
def test_critic_matches_the_torch_module_to_second_order(size, batch):
    from esr_hip import critic as K
    netD = make_D(size)
    eng = K.CriticEngine(netD, 'split')
    real, fake = seeded_uniform((batch, 3, size, size), 11).cuda(), seeded_uniform((batch, 3, size, size), 12).cuda()
    pt = seeded_uniform((batch, 1, 1, 1), 13).cuda()
    params = list(netD.parameters())
    run_stock = lambda x: netD(x)
    run_hip = lambda x: K.critic_forward(eng, x)

    def step(run):
        for p in params:
            p.grad = None
        for m in netD.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        pr, pf = run(real), run(fake)
        interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
        crit = run(interp)
        with K.input_grad_only():
            g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
        gp = 10.0 * ((g.reshape(g.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
        loss = pf.mean() - pr.mean() + gp
        loss.backward()
        return (pr.detach().clone(), pf.detach().clone(), g.detach().clone(), gp.detach().clone(), [p.grad.clone() for p in params],
                netD.features[3].running_mean.clone(), netD.features[3].running_var.clone())
    ref = step(run_stock)
    got = step(run_hip)
    assert rel(got[0], ref[0]) < 1e-3 and rel(got[1], ref[1]) < 1e-3                  # logits
    # Gradients: LeakyReLU makes them piecewise continuous.  The two executions agree on the features to ~3e-5 (16 vs 24 significand bits),
    # so a handful of the ~10^6 activations that lie within that distance of zero take the other branch, and EACH such flip moves the
    # gradient by O(1) in its receptive field: 2 flips of 10^6 already cost 1e-3 in relative L2 (tools/experiments/critic_probe_depth.py
    # counts them; every kernel alone agrees with float64 to 4e-6, tools/experiments/critic_probe_bwd.py).  Hence the robust statements:
    # norms, the penalty (a function of the gradient's norm per image) and the bulk of the elements.
    assert robust_close(got[2], ref[2])                                               # d critic / d input (first-order backward)
    assert abs(float(got[3]) - float(ref[3])) < 2e-3 * abs(float(ref[3]))            # the penalty
    scale = max(float(g.norm()) for g in ref[4])
    for (name, _), a, b in zip(netD.named_parameters(), got[4], ref[4]):
        if float(b.norm()) < 1e-3 * scale:
            assert float(a.norm()) < 3e-3 * scale, name                               # analytically ~0 (a conv bias in front of BatchNorm)
            continue
        # measured (tools/experiments/critic_probe.py, float64 module with its weights perturbed by 3e-5 relative noise — the size of the
        # 16-bit-operand feature error): parameter gradients of this loss move by 9e-3 (median) to 2e-2, d critic / d input by 1.8e-2
        assert abs(float(a.norm()) - float(b.norm())) < 2e-2 * float(b.norm()), (name, float(a.norm()), float(b.norm()))
        assert rel(a, b) < 6e-2, (name, rel(a, b))
    torch.testing.assert_close(got[5], ref[5], rtol=1e-3, atol=1e-5)                  # running statistics: three calls, in order
    torch.testing.assert_close(got[6], ref[6], rtol=1e-3, atol=1e-6)


def robust_close(a, b, tol=2e-2, bulk=0.95, l2=0.2):
    """`bulk` of the elements within tol * rms(b) of each other, and the whole within l2 in relative L2."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    rms = float(b.norm()) / b.numel() ** 0.5
    frac = float(((a - b).abs() <= tol * rms).double().mean())
    return frac >= bulk and rel(a, b) < l2


def test_critic_first_order_input_gradient_for_the_generator_step():
    """The GAN term of the G step: d(-mean critic(fake)) / d fake with the critic's parameters frozen (SRRaGAN_model.py:462-472)."""
    from esr_hip import critic as K
    netD = make_D(64)
    eng = K.CriticEngine(netD, 'split')
    for p in netD.parameters():
        p.requires_grad_(False)
    # (batch 16: with 4 images the last BatchNorm averages 16 values and the input gradient of EVERY implementation — torch fp32 against torch
    # float64 included — moves by 0.4-2 % with the summation order, through LeakyReLU sign flips; measured: 0.946-0.98 of the elements inside
    # the bound at batch 4-8 depending on which launches split their K axis, 0.974-0.979 at batch 16)
    fake = seeded_uniform((16, 3, 64, 64), 21).cuda()
    fa, fb = fake.clone().requires_grad_(True), fake.clone().requires_grad_(True)
    (-netD(fa).mean()).backward()
    (-K.critic_forward(eng, fb).mean()).backward()
    assert robust_close(fb.grad, fa.grad, bulk=0.93) and abs(float(fb.grad.norm()) - float(fa.grad.norm())) < 1e-2 * float(fa.grad.norm())


def test_critic_eval_mode_uses_running_statistics():
    from esr_hip import critic as K
    netD = make_D(64)
    eng = K.CriticEngine(netD, 'split')
    x = seeded_uniform((2, 3, 64, 64), 31).cuda()
    with torch.no_grad():
        for _ in range(2):
            netD(seeded_uniform((4, 3, 64, 64), 32).cuda())          # move the running statistics away from (0, 1)
        netD.eval()
        assert rel(K.critic_forward(eng, x), netD(x)) < 1e-3


def test_bf16_critic_tracks_the_fp32_class_one():
    from esr_hip import critic as K
    netD = make_D(64)
    eng = K.CriticEngine(netD, 'split')
    x = seeded_uniform((4, 3, 64, 64), 41).cuda()
    with torch.no_grad():
        a = K.critic_forward(eng, x)
        for m in netD.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        eng.set_precision('bf16')
        b = K.critic_forward(eng, x)
    assert rel(b, a) < 5e-2


def test_fused_passes_equal_the_per_layer_graph_bit_for_bit(monkeypatch):
    """Three launch lists (forward, backward, backward of the backward) against one autograd node per layer: same kernels in the same order,
    so logits, the input gradient, the penalty and every parameter gradient of the WGAN-GP loss must be identical."""
    from esr_hip import critic as K
    monkeypatch.setattr(K, 'SPLITK', False)       # (the lists may split the K axis of the deep layers' launches: another fp32 summation order; next test)
    netD = make_D(64)
    eng = K.CriticEngine(netD, 'split')
    real, fake = seeded_uniform((4, 3, 64, 64), 11).cuda(), seeded_uniform((4, 3, 64, 64), 12).cuda()
    pt = seeded_uniform((4, 1, 1, 1), 13).cuda()
    params = list(netD.parameters())

    def step(fused):
        monkeypatch.setattr(K, 'FUSED', fused)
        for p in params:
            p.grad = None
        for m in netD.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        pr, pf = K.critic_forward(eng, real), K.critic_forward(eng, fake)
        interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
        crit = K.critic_forward(eng, interp)
        with K.input_grad_only():
            g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
        gp = 10.0 * ((g.reshape(g.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
        (pf.mean() - pr.mean() + gp).backward()
        return [pr.detach().clone(), pf.detach().clone(), g.detach().clone(), gp.detach().clone()] + [p.grad.clone() for p in params] + \
            [netD.features[3].running_var.clone(), netD.features[3].num_batches_tracked.clone()]
    a, b = step(False), step(True)
    names = ['pred_real', 'pred_fake', 'dD/dx', 'gp'] + [n for n, _ in netD.named_parameters()] + ['running_var', 'num_batches_tracked']
    scale = max(float(g.norm()) for g in a[4:-2])
    for name, u, v in zip(names, a, b):
        if name in ('pred_real', 'pred_fake', 'dD/dx', 'gp', 'running_var', 'num_batches_tracked'):
            assert torch.equal(u, v), (name, float((u.double() - v.double()).abs().max()))
        else:
            # parameter gradients: where the penalty's second-order cotangent and the first-order one meet on a conv output, the fused
            # pass adds them in fp32 (esr_act_combine) while autograd adds the per-layer graph's bf16 planes — plus another summation
            # order in the batched weight-gradient launch
            if max(float(u.norm()), float(v.norm())) < 1e-3 * scale:
                continue                      # analytically zero (conv bias in front of BatchNorm): noise on both sides
            assert float((u.double() - v.double()).norm()) < 3e-2 * float(u.norm()), name
    # and the G step's use: input gradient with frozen parameters
    for p in params:
        p.requires_grad_(False)
    outs = []
    for fused in (False, True):
        monkeypatch.setattr(K, 'FUSED', fused)
        f = fake.clone().requires_grad_(True)
        (-K.critic_forward(eng, f).mean()).backward()
        outs.append(f.grad.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('precision', ['split'])      # (one-plane bf16: a reordered sum flips bf16 roundings layer after layer — logits move by 0.7 %)
def test_split_k_launches_of_the_deep_layers_only_reorder_the_sum(monkeypatch, precision):
    """esr_conv3x3_desc.k_split_ws: with the workspace lent, the 256- / 512-channel layers' launches (forward, data gradient, forward of the double
    backward; plain and tap-masked kernels) run as 2-8 workgroup sets over slices of the K axis + one ordered sum.  Same WGAN-GP step with and
    without: equal up to fp32 summation order (then one more rounding to the planes' 16 / 8 significand bits)."""
    from esr_hip import critic as K
    netD = make_D(128)
    real, fake = seeded_uniform((8, 3, 128, 128), 21).cuda(), seeded_uniform((8, 3, 128, 128), 22).cuda()
    pt = seeded_uniform((8, 1, 1, 1), 23).cuda()
    params = list(netD.parameters())

    def step(splitk):
        monkeypatch.setattr(K, 'SPLITK', splitk)
        eng = K.CriticEngine(netD, precision)
        for p in params:
            p.grad = None
        for m in netD.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        pr, pf = K.critic_forward(eng, real), K.critic_forward(eng, fake)
        interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
        crit = K.critic_forward(eng, interp)
        with K.input_grad_only():
            g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
        gp = 10.0 * ((g.reshape(g.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
        (pf.mean() - pr.mean() + gp).backward()
        sets = [bs for v in eng._free_sets.values() for bs in v]
        return [pr.detach().clone(), pf.detach().clone(), g.detach().clone(), gp.detach().clone()] + [p.grad.clone() for p in params], sets
    (a, sets_a), (b, sets_b) = step(False), step(True)
    assert all(bs.ksw is None for bs in sets_a)
    names = ['pred_real', 'pred_fake', 'dD/dx', 'gp'] + [n for n, _ in netD.named_parameters()]
    # bf16 planes: a reordered fp32 sum can round to the neighbouring bf16 value (2^-8 relative) at a few outputs of a layer
    # (the critic amplifies: a last-bit change of a deep layer's output moves the logits by 3e-5 and, through LeakyReLU sign flips behind
    # BatchNorms that average few values, the gradients by ~1 % — the same spread as torch fp32 against torch float64; the launch-level
    # statement is test_split_k_convs_match_float64_like_the_unsplit_launch)
    tol = {'split': (2e-4, 5e-2), 'bf16': (5e-3, 2e-1)}[precision]
    scale = max(float(g.norm()) for g in a[4:])
    for name, u, v in zip(names, a, b):
        if max(float(u.norm()), float(v.norm())) < 1e-3 * scale and name not in ('pred_real', 'pred_fake', 'dD/dx', 'gp'):
            continue                          # analytically zero gradients (conv bias in front of BatchNorm)
        t = tol[0] if name in ('pred_real', 'pred_fake') else tol[1]
        assert rel(v, u) < t, (name, rel(v, u))
    assert not torch.equal(a[0], b[0]) or precision == 'bf16'          # the split launches really ran (split precision: some last bit moves)


@pytest.mark.parametrize('precision', ['split', 'bf16'])
def test_split_k_convs_match_float64_like_the_unsplit_launch(precision):
    """Launch level: the deep layers' forward (plain 3x3 and the tap-masked space-to-depth form) and data-gradient convs with the split-K
    workspace lent vs without, both against torch float64: the same error (the split launch only reorders an fp32 sum), and the workspace was
    really used (4 images of 8x8: 32 workgroups -> the library splits)."""
    import torch.nn.functional as F
    from esr_hip import act as A
    from esr_hip import critic as K
    netD = make_D(128)
    eng = K.CriticEngine(netD, precision)
    eng.refresh()
    P = eng.planes
    ws = torch.full((4 << 20,), float('nan'), device='cuda')
    B, h = 4, 8
    for li in (5, 6, 7, 8, 9):
        L = eng.layers[li]
        if L.strided:
            x = seeded_uniform((B, L.cin, 2 * h, 2 * h), 3 + li).cuda() - 0.5
            xin = s2d(x)
            ref = F.conv2d(x.double(), L.conv.weight.double(), L.conv.bias.double(), stride=2, padding=1)
            kw = dict(tap_mask_k=K.MASK_FWD, tap_mask_k_shift=1)
        else:
            x = seeded_uniform((B, L.cin, h, h), 3 + li).cuda() - 0.5
            xin = x
            ref = F.conv2d(x.double(), L.conv.weight.double(), L.conv.bias.double(), padding=1)
            kw = {}
        xa = K._PackIn.apply(xin, P)
        err = []
        for w in (None, ws):
            y = K.new_at(P, B, L.cout // 8, h, h, 'cuda')
            A.conv3x3(L.fwd, K.view_of(xa), B, h, h, L.cout, out=K.view_of(y), reverse=False, k_split_ws=w, **kw)
            err.append(rel(K._UnpackOut.apply(y, L.cout), ref))
        assert bool(torch.isfinite(ws[:4096]).all()), ('forward did not split', li)
        assert err[1] < 1.05 * err[0] + 1e-7 and err[0] < (1e-5 if precision == 'split' else 5e-3), (li, err)
        ws.fill_(float('nan'))
        dy = seeded_uniform((B, L.cout, h, h), 30 + li).cuda() - 0.5
        dya = K._PackIn.apply(dy, P)
        if L.strided:
            refdx = s2d(torch.nn.grad.conv2d_input((B, L.cin, 2 * h, 2 * h), L.conv.weight.double(), dy.double(), stride=2, padding=1).float()).double()
            kw = dict(tap_mask_m=K.MASK_FLIPPED)
        else:
            refdx = torch.nn.grad.conv2d_input((B, L.cin, h, h), L.conv.weight.double(), dy.double(), padding=1)
            kw = {}
        err = []
        for w in (None, ws):
            dx = K.new_at(P, B, L.cin_e // 8, h, h, 'cuda')
            A.conv3x3(L.tr, K.view_of(dya), B, h, h, L.cin_e, out=K.view_of(dx), use_bias=False, reverse=False, k_split_ws=w, **kw)
            err.append(rel(K._UnpackOut.apply(dx, L.cin_e), refdx))
        assert bool(torch.isfinite(ws[:4096]).all()), ('data gradient did not split', li)
        assert err[1] < 1.05 * err[0] + 1e-7 and err[0] < (1e-5 if precision == 'split' else 5e-3), (li, err)
        ws.fill_(float('nan'))


@pytest.mark.parametrize('splitk', [False, True])
def test_grouped_call_equals_separate_calls(monkeypatch, splitk):
    """critic_forward_group([real, fake, interp]): one pass over the 3 x B images with per-batch statistics, the penalty's backward and double
    backward restricted to the interpolated images (input_grad_only(group=2)), ONE weight-gradient launch for the three batches — against
    three separate calls (the reference's form, SRRaGAN_model.py:345-368).  Same kernels on the same numbers: only fp32 / double summation
    orders differ (batched weight gradient; with split K also the deep layers' sums, which the critic amplifies — see the split-K test)."""
    from esr_hip import critic as K
    monkeypatch.setattr(K, 'SPLITK', splitk)
    netD = make_D(64)
    real, fake = seeded_uniform((8, 3, 64, 64), 31).cuda(), seeded_uniform((8, 3, 64, 64), 32).cuda()
    pt = seeded_uniform((8, 1, 1, 1), 33).cuda()
    params = list(netD.parameters())
    bns = [m for m in netD.modules() if isinstance(m, torch.nn.BatchNorm2d)]

    def step(grouped):
        eng = K.CriticEngine(netD, 'split')
        for p in params:
            p.grad = None
        for m in bns:
            m.reset_running_stats()
        interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
        if grouped:
            pr, pf, crit = K.critic_forward_group(eng, [real, fake, interp])
        else:
            pr, pf, crit = K.critic_forward(eng, real), K.critic_forward(eng, fake), K.critic_forward(eng, interp)
        with K.input_grad_only(group=2 if grouped else None):
            g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
        gp = 10.0 * ((g.reshape(g.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
        (pf.mean() - pr.mean() + gp).backward()
        return [pr.detach().clone(), pf.detach().clone(), crit.detach().clone(), g.detach().clone(), gp.detach().clone()] + [p.grad.clone() for p in params] + \
            [bns[0].running_mean.clone(), bns[-1].running_var.clone(), bns[3].num_batches_tracked.clone()]
    a, b = step(False), step(True)
    names = ['pred_real', 'pred_fake', 'pred_interp', 'dD/dx', 'gp'] + [n for n, _ in netD.named_parameters()] + ['running_mean', 'running_var', 'num_batches_tracked']
    scale = max(float(g.norm()) for g in a[5:-3])
    assert int(b[-1]) == 3
    for name, u, v in zip(names, a, b):
        if name == 'num_batches_tracked':
            assert torch.equal(u, v)
            continue
        if name.startswith('pred'):
            tol = 2e-4 if splitk else 2e-6
        elif name in ('dD/dx', 'gp'):
            tol = 5e-2 if splitk else 1e-4
        elif name.startswith('running'):
            tol = 1e-5
        else:
            if max(float(u.norm()), float(v.norm())) < 1e-3 * scale:
                continue                      # analytically zero (conv bias in front of BatchNorm)
            tol = 5e-2 if splitk else 2e-3
        assert rel(v, u) < tol, (name, rel(v, u))


def test_finalize_folded_into_the_normalise_launch_equals_the_separate_launch(monkeypatch):
    """esr_bn_finalize_apply (the forward's default: statistics -> ONE launch that derives the affine, stores mean / rstd / scale / shift,
    moves the running statistics and normalises) against esr_bn_finalize followed by esr_bn_apply: same fp64 arithmetic on the same sums
    (the sums themselves are fp64 atomics: their order may move the last bit of a double), for a grouped call of three batches."""
    from esr_hip import critic as K
    netD = make_D(64)
    xs = [seeded_uniform((4, 3, 64, 64), 41 + i).cuda() for i in range(3)]
    bns = [m for m in netD.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    params = list(netD.parameters())

    def run(fused):
        monkeypatch.setattr(K, 'FUSE_FINALIZE', fused)
        eng = K.CriticEngine(netD, 'split')
        for m in bns:
            m.reset_running_stats()
        for p in params:
            p.grad = None
        outs = K.critic_forward_group(eng, xs)
        (outs[0].mean() - 2 * outs[1].mean() + outs[2].square().mean()).backward()
        return [o.detach().clone() for o in outs] + [p.grad.clone() for p in params] + [m.running_mean.clone() for m in bns] + [m.running_var.clone() for m in bns]
    a, b = run(False), run(True)
    assert all(int(m.num_batches_tracked) == 3 for m in bns)
    scale = max(float(g.norm()) for g in a[3:3 + len(params)])
    for i, (u, v) in enumerate(zip(a, b)):
        if max(float(u.norm()), float(v.norm())) < 1e-6 * scale:
            continue
        assert rel(v, u) < 1e-6, (i, rel(v, u))
