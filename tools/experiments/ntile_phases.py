"""(Runs against the tree with tools/experiments/patches/r06_three_experiments.patch applied = commit bc221d6: the experiment lost and is not in the shipping tree.)
Phase stamps of the one-MFMA 32-channel conv kernel with ONE and with TWO tiles per workgroup (csrc/esr_conv.hip: NTILE_ONE_MFMA)
at the configs[1] shape in 'mixed' — written by tools/experiments/trace_conv_mixed.py (instrumented build) into gpurun_out/trace_<cin>_32.npy.

    python tools/experiments/ntile_phases.py <trace.npy> <chunks>

Per workgroup (wave 0): entry | tile decoded | first copies issued | per tile: seeded, per chunk [start, copies issued, landed, barrier, MFMAs], K loop
done, stored.  With two tiles per workgroup the second tile's first copies are issued in front of the first tile's stores; its chunk 0 then waits
for `vmcnt(0)`, i.e. for those stores as well."""
import sys
import numpy as np

path, ncp = sys.argv[1], int(sys.argv[2])
t = np.load(path).astype(np.int64)
t = t[t[:, 2] != 0]
ns = (t[:, 2:126] != 0).sum(1)
per_tile = 1 + 5 * ncp + 2
for ntl in (1, 2):
    sel = t[ns == 3 + ntl * per_tile]
    if not len(sel):
        continue
    T = sel[:, 2:2 + 3 + ntl * per_tile]
    wall = (sel[:, 127] - sel[:, 126]) / 100.0
    c = np.median((T[:, -1] - T[:, 0]) / wall)            # cycles per us
    print('%d workgroups with %d tile(s): life %.2f us (median) = %.2f us per tile; launch span %.2f us; shader clock %.2f GHz' % (
        len(sel), ntl, np.median(wall), np.median(wall) / ntl, (t[:, 127].max() - t[:, 126].min()) / 100.0, c / 1e3))
    print('  set-up: entry -> decoded %5.0f, -> first copies issued %5.0f cycles (%.2f us)' % ((T[:, 1] - T[:, 0]).mean(), (T[:, 2] - T[:, 1]).mean(), (T[:, 2] - T[:, 0]).mean() / c))
    for k in range(ntl):
        o = 3 + k * per_tile
        seed = T[:, o] - T[:, o - 1]
        ch = lambda j, a, b: T[:, o + 1 + 5 * j + b] - T[:, o + 1 + 5 * j + a]
        wait0 = ch(0, 1, 2)
        waits = np.array([ch(j, 1, 2) for j in range(1, ncp)]).mean(0)
        mfma = np.array([ch(j, 3, 4) for j in range(ncp)]).mean(0)
        issue = np.array([ch(j, 0, 1) for j in range(ncp)]).mean(0)
        kloop = T[:, o + 1 + 5 * ncp] - T[:, o]
        epi = T[:, o + 2 + 5 * ncp] - T[:, o + 1 + 5 * ncp]
        print('  tile %d: seed %5.0f | chunk 0 landing wait %5.0f (%.2f us), later chunks %5.0f | copy issue per chunk %5.0f | MFMAs per chunk %5.0f | K loop %6.0f (%.2f us) | '
              'next tile set-up + stores %5.0f (%.2f us)' % (k, seed.mean(), wait0.mean(), wait0.mean() / c, waits.mean(), issue.mean(), mfma.mean(), kloop.mean(), kloop.mean() / c,
                                                            epi.mean(), epi.mean() / c))
