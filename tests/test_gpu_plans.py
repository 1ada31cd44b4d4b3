"""GPU tests (-m gpu) of the launch lists (esr_cmd / esr_run, include/esr_hip.h): a pass of the generator recorded once and replayed with
one C-ABI call must produce bit-identical results to the same launches issued one FFI call at a time (ESR_PLANS=0 path: the recording
code itself), across replays with NEW input / output tensors (the patched pointers), in inference and in training (dx and every dW)."""
import ctypes as C
import os
import sys

import pytest
import torch

from oracle.weights import fill_formula_weights, seeded_uniform

pytestmark = pytest.mark.gpu


def make_net(nb=2, lat=3, sf=4, precision='split'):
    import models.modules.architecture as arch
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, gc=32, upscale=sf, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                       latent_input='all_layers_HR_downscaled' if lat else None, num_latent_channels=lat)
    fill_formula_weights(net, gain=0.5)
    net = net.cuda()
    net.set_precision(precision)
    return net


def inputs(B, lat, sf, h, w, seed):
    return seeded_uniform((B, 3 + lat * sf * sf, h, w), seed).cuda()


@pytest.mark.parametrize('precision', ['split', 'bf16', 'mixed'])
def test_replayed_forward_is_bit_identical_to_direct_launches(precision):
    net = make_net(precision=precision)
    eng = net.engine
    xs = [inputs(2, 3, 4, 20, 24, 70 + i) for i in range(3)]
    with torch.no_grad():
        eng.use_plans = False
        ref = [net(x, pad=2).clone() for x in xs]
        eng.use_plans = True
        got = [net(x, pad=2) for x in xs]           # call 0 records, calls 1-2 replay into new output tensors from new inputs
    for r, g in zip(ref, got):
        assert torch.equal(r, g)
    plans = [p for b in eng._bufs.values() for p in b['_plans'].values()]        # (no_grad forwards: one inference buffer set)
    assert len(plans) == 1 and plans[0].n_cmds >= 3 + 15 * 2 + 5      # one list for the whole pass, not one per call
    assert len({g.data_ptr() for g in got}) == 3


@pytest.mark.parametrize('precision', ['split', 'bf16'])
def test_replayed_training_pass_is_bit_identical_to_direct_launches(precision):
    def run(use_plans):
        net = make_net(precision=precision)
        net.engine.use_plans = use_plans
        outs = []
        for i in range(3):
            x = inputs(2, 3, 4, 16, 20, 80 + i).requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            y = net(x)
            (y * seeded_uniform(tuple(y.shape), 90 + i).cuda()).sum().backward()
            outs.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()]))
            del y                                    # frees the saved buffer set: the next forward reuses it (and its recorded lists)
        return net, outs
    _, ref = run(False)
    net, got = run(True)
    for (y0, dx0, dw0), (y1, dx1, dw1) in zip(ref, got):
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
        assert all(torch.equal(a, b) for a, b in zip(dw0, dw1))
    kinds = sorted(k[0] for b in net.engine._bufs.values() for k in b['_plans'])
    assert kinds == ['bwd', 'fwd'], kinds                   # one list per pass kind, recorded on the first step and replayed on the other two


@pytest.mark.parametrize('groups', [2, 3, 5])
def test_weight_gradients_under_the_data_gradient_chain_are_bit_identical(groups):
    """RRDBEngine.wgrad_overlap: the recorded backward launches the weight gradients of the layers whose dy are final on a second stream
    (esr_conv3x3_wgrad_batch_run_side) while the data-gradient chain goes on, and joins at the end.  Same slicing as the single launch:
    every gradient bit-identical, on the recording step and on replays with new tensors; the gradients are consumed on the main stream right
    behind backward() (a missing join would read them early)."""
    def run(g):
        net = make_net(nb=3, precision='bf16')
        net.engine.wgrad_overlap = g
        outs = []
        for i in range(4):
            x = inputs(4, 3, 4, 20, 16, 180 + i).requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            y = net(x)
            (y * seeded_uniform(tuple(y.shape), 190 + i).cuda()).sum().backward()
            total = torch.stack([p.grad.double().abs().sum() for p in net.parameters()]).sum()       # main stream, no synchronisation in between
            outs.append((x.grad.clone(), [p.grad.clone() for p in net.parameters()], total.clone()))
            del y
        return net, outs
    _, ref = run(0)
    net, got = run(groups)
    plans = [v for b in net.engine._bufs.values() for k, v in b['_plans'].items() if k[0] == 'bwd']
    assert len(plans) == 1 and sum(1 for it in plans[0][0].items if callable(it)) == groups          # groups - 1 side launches + the join
    for (dx0, dw0, t0), (dx1, dw1, t1) in zip(ref, got):
        assert torch.equal(dx0, dx1) and torch.equal(t0, t1)
        assert all(torch.equal(a, b) for a, b in zip(dw0, dw1))


def test_replay_follows_weight_updates_and_gradient_accumulation():
    """The lists point at the weight PACKS, which are refreshed before every replay; .grad accumulation over two backward passes must add
    into the first pass's gradients (each pass gets its own flat dW buffer)."""
    net = make_net(nb=1)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    ref = make_net(nb=1)
    ref.engine.use_plans = False
    opt_ref = torch.optim.SGD(ref.parameters(), lr=1e-3)
    for i in range(3):
        for n_, o_ in ((net, opt), (ref, opt_ref)):
            o_.zero_grad()
            for j in range(2):                       # two accumulation passes per step
                x = inputs(1, 3, 4, 12, 12, 100 + 2 * i + j)
                n_(x).square().mean().backward()
            o_.step()
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.equal(a, b) and torch.equal(a.grad, b.grad)


def test_esr_run_reports_the_failing_command():
    from esr_hip import _lib
    cmds = (_lib.Cmd * 2)()
    cmds[0].op = _lib.OP_ZERO
    buf = torch.ones(64, dtype=torch.int32, device='cuda')
    cmds[0].u.zero.p, cmds[0].u.zero.n16 = buf.data_ptr(), 4
    cmds[1].op = 99
    failed = C.c_int(-7)
    rc = _lib.lib.esr_run(cmds, 2, C.byref(failed), None)
    assert rc == _lib.ESR_E_ARG and failed.value == 1
    torch.cuda.synchronize()
    assert int(buf[:16].abs().sum()) == 0 and int(buf[16:].sum()) == 48        # command 0 ran, the list stopped at command 1
    assert _lib.lib.esr_run(cmds, 1, C.byref(failed), None) == 0 and failed.value == -1
    assert _lib.lib.esr_run(None, 3, None, None) == _lib.ESR_E_ARG


@pytest.mark.parametrize('weight_decay', [0.0, 1e-2])
def test_multi_tensor_adam_matches_torch_adam(weight_decay):
    """esr_hip.optim.Adam (one launch for all tensors) against torch.optim.Adam, which the reference uses (SRRaGAN_model.py:147-160): same
    parameters after 6 steps with a learning-rate change in between, gradients as views of one flat buffer (the engine's layout) whose
    address changes between steps; state_dict round trip into torch's optimizer."""
    from esr_hip.optim import Adam
    torch.manual_seed(0)
    shapes = [(64, 51, 3, 3), (64,), (32, 67, 3, 3), (32,), (3, 64, 3, 3), (3,), (5001,)]
    pa = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = Adam(pa, lr=1e-2, betas=(0.9, 0.99), weight_decay=weight_decay)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), weight_decay=weight_decay)
    sched_a = torch.optim.lr_scheduler.MultiStepLR(oa, [3], 0.5)
    sched_b = torch.optim.lr_scheduler.MultiStepLR(ob, [3], 0.5)
    n = sum(p.numel() for p in pa)
    hold = []
    for it in range(6):
        flat = torch.randn(n, device='cuda') * (10.0 ** -(it % 3))
        hold.append(flat)                                   # keeps the old buffers allocated: every step sees new gradient addresses
        off = 0
        for p, q in zip(pa, pb):
            p.grad = flat[off:off + p.numel()].view_as(p)
            q.grad = p.grad.clone()
            off += p.numel()
        oa.step(); ob.step(); sched_a.step(); sched_b.step()
        for p, q in zip(pa, pb):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()), it
    assert oa.param_groups[0]['lr'] == ob.param_groups[0]['lr'] == 5e-3
    # the pointer tables go to the device through pinned memory + an asynchronous copy (no host synchronisation per step), up to four cached
    # per group by pointer fingerprint: six new gradient addresses = six tables built; coming back to the last two buffers builds none
    assert oa.table_uploads == 6
    for it in (4, 5, 4):
        off = 0
        for p, q in zip(pa, pb):
            p.grad = hold[it][off:off + p.numel()].view_as(p)
            q.grad = p.grad.clone()
            off += p.numel()
        oa.step(); ob.step()
        for p, q in zip(pa, pb):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()), it
    assert oa.table_uploads == 6
    assert all(p._version == q._version for p, q in zip(pa, pb))      # the step is visible to everything that watches version counters (weight packs)
    for p, q in zip(pa, pb):
        assert float(oa.state[p]['step']) == float(ob.state[q]['step']) == 9
        torch.testing.assert_close(oa.state[p]['exp_avg_sq'], ob.state[q]['exp_avg_sq'], rtol=1e-5, atol=1e-12)
    oc = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), weight_decay=weight_decay)
    oc.load_state_dict(oa.state_dict())                     # same state layout: checkpoints are interchangeable
    assert float(oc.state[pb[0]]['step']) == 9


@pytest.mark.parametrize('precision', ['split', 'mixed'])
def test_mask_stash_gives_the_same_input_gradient_with_a_third_of_the_memory(precision):
    """A differentiable forward through a FROZEN generator (the Z search) keeps three rotating dense-block buffers and a one-plane copy of
    each RDB's intermediate activations instead of every 24-group hi+lo buffer: the data gradient only reads their signs (LeakyReLU').  Same
    dx bit for bit; weight gradients are refused loudly."""
    net = make_net(nb=3, precision=precision)
    for p in net.parameters():
        p.requires_grad_(False)
    eng = net.engine
    x0 = inputs(2, 3, 4, 16, 20, 130)
    cot = None
    out = {}
    for mode in ('full', 'masks'):
        eng.stash = mode
        x = x0.clone().requires_grad_(True)
        y = net(x, pad=2)
        cot = seeded_uniform(tuple(y.shape), 131).cuda() if cot is None else cot
        (y * cot).sum().backward()
        bufs = [b for k, b in eng._bufs.items() if k[-1] == (True if mode == 'full' else 'masks')][0]
        nbytes = sum(t.nbytes() for t in bufs['rdb']) + sum(t.nbytes() for t in bufs.get('stash', []))
        out[mode] = (y.detach().clone(), x.grad.clone(), nbytes)
        del y
    assert torch.equal(out['full'][0], out['masks'][0]) and torch.equal(out['full'][1], out['masks'][1])
    assert out['masks'][2] < 0.67 * out['full'][2]            # nb = 3: (3 x 24 x 2 + 9 x 16) / (9 x 24 x 2) planes; -> 1/3 for deep nets
    # a parameter that wants a gradient switches the forward back to keeping everything
    next(net.parameters()).requires_grad_(True)
    assert eng.keep_mode() is True


@pytest.mark.parametrize('precision', ['split', 'bf16', 'mixed'])
def test_small_launch_output_slices_are_bit_identical_to_the_64_channel_form(precision):
    """A 64-channel conv of a small launch (no more tiles than CUs) runs as two 32-channel output slices of the 32-channel kernel out of the same
    weight pack (esr_conv3x3, csrc/esr_conv.hip: two workgroups per CU instead of one): same products, same order per output channel, so the
    forward, the input gradient and every weight gradient must not change by a bit against the unsliced form (esr_conv3x3_desc.lds_stages = 2
    keeps every launch in its 64-channel two-stage form).  Covers the RDB's closing conv (residual from the staged tile), the RRDB's (second
    residual), the trunk / HR convs and the mirrored data gradients (LeakyReLU masks)."""
    from esr_hip import act as A

    def run(stages):
        keep = A.LDS_STAGES
        A.LDS_STAGES = stages
        try:
            net = make_net(nb=2, precision=precision)
            x = inputs(3, 3, 4, 24, 20, 77).requires_grad_(True)
            y = net(x)
            (y * seeded_uniform(tuple(y.shape), 78).cuda()).sum().backward()
            return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()]
        finally:
            A.LDS_STAGES = keep
    y0, dx0, dw0 = run(2)
    y1, dx1, dw1 = run(0)
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    assert all(torch.equal(a, b) for a, b in zip(dw0, dw1))


@pytest.mark.parametrize('split', [True, False, 'f16x2', 'f16x3'])
def test_batched_weight_pack_is_bit_identical_to_the_single_tensor_pack(split):
    """esr_pack_batch_run (one launch for every pack of a step: each (K chunk, M tile) staged through LDS) against esr_pack_conv_weights (one
    thread per output vector) on the pack kinds the engines make: forward packs with a latent group / ragged channel counts / permuted rows
    (pixel-shuffle convs), data-gradient packs over a channel slice, the latent slice and permuted K rows, 1 and 2 M tiles, 1..6 K chunks."""
    from esr_hip import act as A
    torch.manual_seed(3)
    dev = 'cuda'
    w = lambda co, ci: torch.randn(co, ci, 3, 3, device=dev) * (1 + torch.arange(co, device=dev).view(-1, 1, 1, 1) * 0.01)
    perm = [int(i) for i in torch.randperm(64)]
    kinds = [dict(weight=w(32, 64), lat=0), dict(weight=w(64, 192), lat=0), dict(weight=w(64, 4), lat=1), dict(weight=w(32, 163), lat=3),
             dict(weight=w(3, 64), lat=0), dict(weight=w(64, 64), lat=0, rows=perm),
             dict(weight=w(32, 160), lat=0, transposed=True, m_slice=(64, 128)), dict(weight=w(64, 67), lat=3, transposed=True, m_slice='latent'),
             dict(weight=w(64, 67), lat=3, transposed=True, m_slice=(0, 64)), dict(weight=w(64, 32), lat=0, transposed=True, rows=perm),
             dict(weight=w(3, 64), lat=0, transposed=True)]
    single = [A.PackedConv(k['weight'], None, k['lat'], split=split, transposed=k.get('transposed', False), m_slice=k.get('m_slice'), rows=k.get('rows')) for k in kinds]
    want = [pk.get().wpack.clone() for pk in single]
    batch = [A.PackedConv(k['weight'], None, k['lat'], split=split, transposed=k.get('transposed', False), m_slice=k.get('m_slice'), rows=k.get('rows')) for k in kinds]
    for pk in batch:
        pk.prepare()
        pk.wpack.fill_(0x5A)
    A.PackBatch().run(batch)
    torch.cuda.synchronize()
    for i, (pk, ref) in enumerate(zip(batch, want)):
        assert torch.equal(pk.wpack, ref), (i, kinds[i]['weight'].shape, int((pk.wpack != ref).sum()))
