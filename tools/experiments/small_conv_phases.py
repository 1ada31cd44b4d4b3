"""Where a small conv launch spends its time: phase stamps of the instrumented build (make -C explorable-super-resolution_amd/csrc trace),
written by tools/experiments/trace_conv.py / trace_conv_mixed.py into gpurun_out/trace_<cin>_<cout>.npy.

    python tools/experiments/small_conv_phases.py 128_32:8[:lead] [64_32:4 ...]          # name:chunks[:lead]

lead = stamps between the entry stamp and chunk 0's first stamp (0: the round-4 kernels; 3: the round-5 kernels, which stamp the end of the tile
decode, of the first chunk's copy issue and of the accumulator seed).

Per workgroup (wave 0's stamps, shader-clock cycles): entry -> first step stamp (tile set-up, and the prologue copies of the two-stage
form), per chunk [copy issue, landing wait, barrier, MFMAs, barrier], the epilogue; plus the 100 MHz wall stamps at entry / exit."""
import sys
import numpy as np

for arg in sys.argv[1:]:
    name, ncp, *lead = arg.split(':')
    ncp, lead = int(ncp), int(lead[0]) if lead else 0
    t = np.load('gpurun_out/trace_%s.npy' % name).astype(np.int64)
    t = t[t[:, 2] != 0]
    rt0, rt1 = t[:, 126], t[:, 127]
    ts = t[:, 2:126]
    n = 5 * ncp + 3 + lead                # entry, [lead], 5 per chunk, after the K loop, after the epilogue
    T = ts[:, :n]
    tot = T[:, n - 1] - T[:, 0]
    wall = (rt1 - rt0) / 100.0            # us
    clk = tot / wall / 1e3                # GHz
    print('%s: %d workgroups; stamped span per workgroup %.2f us (median), launch span %.2f us, shader clock %.2f GHz' %
          (name, len(t), np.median(wall), (rt1.max() - rt0.min()) / 100.0, np.median(clk)))
    c = np.median(clk) * 1e3              # cycles per us
    pro = T[:, 1 + lead] - T[:, 0]
    print('  set-up (entry -> first step stamp)   %6.0f cycles  %.2f us' % (pro.mean(), pro.mean() / c))
    if lead == 3:                         # round-5 kernels: entry | tile decode + slot offsets | first copies issued | seed + epilogue coordinates
        for i, nm in enumerate(['tile decode, slot offsets', "first chunk's copies issued", 'seed + epilogue coordinates']):
            d = T[:, i + 1] - T[:, i]
            print('    %-30s %6.0f cycles' % (nm, d.mean()))
    names = ['issue', 'wait', 'bar1', 'mfma', 'bar2']
    tot_k = 0
    for i, nm in enumerate(names):
        a = np.array([T[:, 1 + lead + 5 * ch + i + 1] - T[:, 1 + lead + 5 * ch + i] for ch in range(ncp)]).T
        tot_k += a.sum(axis=1).mean()
        print('  %-6s per chunk mean %6.0f  chunk0 %6.0f  last %6.0f   sum over chunks %.2f us' % (nm, a.mean(), a[:, 0].mean(), a[:, -1].mean(), a.sum(axis=1).mean() / c))
    epi = T[:, n - 1] - T[:, n - 2]
    print('  K loop total %.2f us;  epilogue %6.0f cycles  %.2f us;  late start of the last workgroup %.2f us' %
          (tot_k / c, epi.mean(), epi.mean() / c, (rt0.max() - rt0.min()) / 100.0))
