#!/usr/bin/env python
"""Secondary measurements of the callers around the hot path (not the headline bench): the generator training step at the C3
per-GPU shape and Z-search iterations at the C4 shape, both through the product's model wrapper / Z_optimizer.

    python tools/bench_paths.py c3 [--batch 32] [--steps 10]        RRDB-23 x4 + CEM, lat 3, 52x52 LR crops, L1 pixel loss, Adam
    python tools/bench_paths.py c4 [--batch 64] [--steps 10]        Z search: STD_increase on one 128x128 LR image, Adam on Z
    python tools/bench_paths.py c5 [--batch 16] [--precision f16]   inference: RRDB-23 x8, 'blurry_cubic_2.0' CEM kernel, 256x256 -> 2048x2048
Under torchrun (one process per GPU) c3 all-reduces the gradients over RCCL and c4 shards the Z batch.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def make_opt(is_train, lat=3, nb=23, with_D=False):
    from options.options import dict_to_nonedict
    return dict_to_nonedict({
        'model': 'srragan', 'scale': 4, 'gpu_ids': [0], 'range': [0, 1], 'is_train': is_train,
        'path': {'models': '/tmp/esr_models', 'log': '/tmp/esr_log', 'pretrain_model_G': None},
        'network_G': {'which_model_G': 'RRDB_net', 'CEM_arch': 1, 'sigmoid_range_limit': 0, 'latent_input': 'all_layers',
                      'latent_input_domain': 'HR_downscaled', 'latent_channels': lat, 'norm_type': None, 'mode': 'CNA', 'nf': 64, 'nb': nb,
                      'in_nc': 3, 'out_nc': 3, 'gc': 32, 'scale': 4},
        # with_D: the discriminator half of configs[2] as codes/options/train/train_explorable_SR.json sets it (Discriminator_VGG_128, n_layers 10,
        # BatchNorm, WGAN-GP with gp_weight 10, non-relativistic), D-verification off and one G step per D step so that every step does both
        'network_D': {'which_model_D': 'discriminator_vgg_128', 'relativistic': 0, 'decomposed_input': 0, 'pre_clipping': 0, 'add_quantization_noise': 0,
                      'norm_type': 'batch', 'act_type': 'leakyrelu', 'mode': 'CNA', 'n_layers': 10, 'nf': 64, 'in_nc': 3,
                      'miopen_find': 1} if with_D else None,       # MIOpen find mode: D step 30.7 -> 20.0 ms with the bf16 critic
        'test': {'kernel': None}, 'datasets': {'train': {'patch_size': 208, 'batch_size': 32}},
        'train': dict({'pixel_weight': 1, 'pixel_criterion': 'l1', 'lr_G': 1e-5, 'lr_D': 1e-5, 'pixel_domain': 'HR', 'grad_accumulation_steps_G': 1,
                       'grad_accumulation_steps_D': 1, 'range_weight': 5000, 'CEM_exp': 1, 'lr_scheme': 'MultiStepLR', 'lr_steps': [100000], 'lr_gamma': 0.5},
                      **({'gan_type': 'wgan-gp', 'gan_weight': 1, 'gp_weight': 10, 'D_update_ratio': 1, 'D_init_iters': 0, 'D_verification': None} if with_D else {}))})


def c5(a, D):
    import contextlib
    import io
    import CEM.CEMnet as C
    import models.modules.architecture as arch
    import models.networks as networks
    torch.manual_seed(0)
    dev = torch.device('cuda', torch.cuda.current_device())
    cem = C.CEMnet(C.Get_CEM_Conf(8), upscale_kernel='blurry_cubic_2.0')
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=a.nb, gc=32, upscale=8, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                       latent_input=None, num_latent_channels=0)
    G = cem.WrapArchitecture_PyTorch(net)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)
    G = G.to(dev).eval()
    net.set_precision(a.precision)
    B = a.batch or 16
    lo, hi = D.shard_range(B)
    x = torch.rand(hi - lo, 3, 256, 256, device=dev)
    with torch.no_grad():
        for _ in range(2):
            y = G(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            y = G(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        d = G.DownscaleOP(y)
    m = int(cem.invalidity_margins_LR)
    if D.rank() == 0:
        print('c5 inference [%s] (RRDB-%d x8 + CEM blurry_cubic_2.0, %d x 256x256 -> 2048x2048 over %d GPU(s)): %.1f ms/batch, %.3e HR pixels/s, '
              'consistency rmse %.2e, peak %.1f GB' % (a.precision, a.nb, B, D.world_size(), dt * 1e3, B * 2048 * 2048 / dt,
                                                       float(((d - x)[..., m:-m, m:-m] ** 2).mean().sqrt()), torch.cuda.max_memory_allocated() / 2 ** 30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('which', choices=['c3', 'c4', 'c5'])
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--nb', type=int, default=23)
    ap.add_argument('--with-D', dest='with_D', action='store_true', help='c3: add Discriminator_VGG_128 + WGAN-GP (the full configs[2] step)')
    ap.add_argument('--precision', default='split', choices=['split', 'mixed', 'f16x2', 'f16', 'bf16'], help="'mixed': fp16 planes, fp32-class forward (inference, Z search, training); 'bf16': single-MFMA operands (C3 names bf16); 'f16x2' / 'f16' are inference-only (c5)")
    a = ap.parse_args()
    from esr_hip import dist as D
    D.init_from_env()
    if a.which == 'c5':
        return c5(a, D)
    import contextlib
    import io
    import models
    torch.manual_seed(0)
    dev = torch.device('cuda', torch.cuda.current_device())
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.create_model(make_opt(a.which == 'c3', nb=a.nb, with_D=a.with_D))
    if a.precision != 'split':
        model.netG.generated_image_model.set_precision(a.precision)
    sync = torch.cuda.synchronize
    if a.which == 'c3':
        B = a.batch or 32
        data = {'LR': torch.rand(B, 3, 52, 52, device=dev), 'HR': torch.rand(B, 3, 208, 208, device=dev),
                'Z': torch.rand(B, 3, 208, 208, device=dev) * 2 - 1}
        for _ in range(3):
            model.feed_data(data); model.optimize_parameters()
        if os.environ.get('ESR_GC_OFF') == '1':
            import gc
            gc.collect(); gc.disable()
        model.timing = None if os.environ.get('ESR_NO_PHASES') == '1' else {}      # the phase timers synchronise every step: ESR_NO_PHASES=1 measures the free-running step
        sync(); t0 = time.perf_counter()
        for _ in range(a.steps):
            model.feed_data(data); model.optimize_parameters()
        sync(); dt = (time.perf_counter() - t0) / a.steps
        if D.rank() == 0:
            print('c3 generator step [' + a.precision + '] (RRDB-%d x4 + CEM, lat 3, %d x 52x52 per GPU, %d GPU(s)): %.1f ms/step, %.0f LR crops/s, l_g_pix %.4f, peak %.1f GB'
                  % (a.nb, B, D.world_size(), dt * 1e3, B * D.world_size() / dt, model.get_current_log()['l_g_pix'], torch.cuda.max_memory_allocated() / 2 ** 30))
            if model.timing is not None:
                print('   phases (GPU ms per step, step synchronised for the timers): ' + ', '.join('%s %.1f' % (k, v / a.steps) for k, v in model.timing.items()))
    else:
        from Z_optimization import Z_optimizer
        B = a.batch or 64
        lr = torch.rand(1, 3, 128, 128, device=dev)
        lo, hi = D.shard_range(B)
        model.feed_data({'LR': lr.expand(hi - lo, -1, -1, -1), 'Z': torch.zeros(hi - lo, 3, 512, 512, device=dev)}, need_GT=False)
        model.test()
        zo = Z_optimizer(objective='STD_increase', Z_size=[512, 512], model=model, Z_range=1, max_iters=2, data={'LR': lr, 'STD_increment': 0.01},
                         initial_LR=0.1, batch_size=B)
        zo.optimize()                       # warm-up (2 iterations)
        zo.max_iters = a.steps
        sync(); t0 = time.perf_counter()
        zo.optimize()
        sync(); dt = (time.perf_counter() - t0) / a.steps
        if D.rank() == 0:
            print('c4 Z search [' + a.precision + '] (RRDB-%d x4 + CEM, %d Z samples of 512x512 over %d GPU(s)): %.1f ms/iteration, %.2f Z-iterations/s, loss %.3e -> %.3e, peak %.1f GB'
                  % (a.nb, B, D.world_size(), dt * 1e3, B / dt, zo.loss_values[0], zo.loss_values[-1], torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == '__main__':
    main()
