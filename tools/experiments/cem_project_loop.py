"""The configs[1] CEM projection (x4 bicubic, 32 x 128x128 -> 512x512, G output on the 592x592 padded frame) 20 times, the three kernels per
chunk of <arg> images (0: whole batch) — the command tools/pmc_workload.sh profiles for the HBM traffic of the two readings of `g`.
    python tools/experiments/cem_project_loop.py [chunk images]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as C
from esr_hip import cem_ops

cem_ops.PROJECT_CHUNK_IMAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cem = C.CEMnet(C.Get_CEM_Conf(4))
net = cem.WrapArchitecture_PyTorch(generated_image=None).cuda().eval()
m = int(cem.invalidity_margins_LR)
lr = torch.rand(32, 3, 128, 128, device='cuda')
g = torch.rand(32, 3, 4 * (128 + 2 * m), 4 * (128 + 2 * m), device='cuda')
td, ti, tu = net.DownscaleOP.taps(), net.Conv_LR_with_Inv_hTh_OP.taps(), net.Upscale_OP.taps()
with torch.no_grad():
    for _ in range(20):
        out = cem_ops.project(lr, g, td, ti, tu, 4, 4 - 4 // 2 - 1, lr_pad=m, crop=4 * m)
torch.cuda.synchronize()
print(out.shape, cem_ops.PROJECT_CHUNK_IMAGES)
