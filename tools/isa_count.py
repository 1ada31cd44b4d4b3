#!/usr/bin/env python
"""Static instruction counts of conv3x3_tile_kernel instantiations (gfx950 ISA out of hipcc -S): total, in front of the first MFMA, behind the
last MFMA (K-loop tail + epilogue), registers and scratch.  VERDICT r4 item 1 asks for <= 500 / <= 700 in the <1,1,*> instantiations.

    python tools/isa_count.py [--only NPL,MT,FMT,NPW] [--filter "1,1,"] [--src path/to/esr_conv.hip] [--keep out.s]

--only compiles a scratch copy of the translation unit whose launch() dispatches just that operand form (seconds instead of minutes); without
it every instantiation of the shipping file is compiled (about three minutes)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'explorable-super-resolution_amd', 'csrc')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None, help='NPL,MT,FMT,NPW of the one operand form to compile')
    ap.add_argument('--filter', default='', help='prefix of the template argument list to print, e.g. "1,1,"')
    ap.add_argument('--src', default=os.path.join(CSRC, 'esr_conv.hip'))
    ap.add_argument('--keep', default=None, help='write the assembly here')
    a = ap.parse_args()
    src = open(a.src).read()
    tmp = tempfile.mkdtemp()
    if a.only:
        npl, mt, fmt, npw = [int(v) for v in a.only.split(',')]
        # restrict launch<>() — where every launch_nst<> instantiation comes from — to the requested form
        head = 'int launch(const ConvArgs& a, hipStream_t s) {\n'
        assert head in src
        src = src.replace(head, head + '    if constexpr (!(NPL == %d && MT == %d && FMT == %d && NPW == %d && !PARTLO && TMODE == 0)) return ESR_E_UNSUPPORTED; else {\n' % (npl, mt, fmt, npw))
        tail = "        return two ? launch_nst<NPL, MT, EPI, 2, FMT, NPW, PARTLO, TMODE>(a, s) : launch_nst<NPL, MT, EPI, 1, FMT, NPW, PARTLO, TMODE>(a, s);\n    }\n}\n"
        assert tail in src
        src = src.replace(tail, tail[:-2] + '    }\n}\n')
    cp = os.path.join(tmp, 'esr_conv_isa.hip')
    open(cp, 'w').write(src.replace('#include "esr_common.h"', '#include "%s/esr_common.h"' % CSRC))
    out = a.keep or os.path.join(tmp, 'conv.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-function', '-Wno-unused-command-line-argument',
                           '--cuda-device-only', '-S', cp, '-o', out])
    txt = open(out).read()
    rows = []
    for m in re.finditer(r'^(_Z[\w]*conv3x3_tile_kernel[\w]*):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        t = re.search(r'conv3x3_tile_kernel<(.*?)>', dem).group(1).replace(' ', '')
        if a.filter and not t.startswith(a.filter.replace(' ', '')):
            continue
        ins = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';')) and l.strip()]
        mf = [i for i, l in enumerate(ins) if l.startswith('v_mfma')]
        meta = re.search(r'\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel' % re.escape(name), txt, re.S).group(1)
        vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta).group(1)
        sg = re.search(r'\.amdhsa_next_free_sgpr (\d+)', meta).group(1)
        scr = re.search(r'; ScratchSize: (\d+)', txt[m.end():m.end() + 3000])
        lanes = sum(1 for l in ins if l.startswith(('v_writelane', 'v_readlane')))
        rows.append((t, len(ins), mf[0], len(ins) - 1 - mf[-1], len(mf), vg, sg, scr.group(1) if scr else '?', lanes))
    print('# conv3x3_tile_kernel<NPL, MT, EPI, NST, FMT, NPW, PARTLO, TMODE>: instructions total / before the first MFMA / after the last MFMA')
    print('%-34s %6s %7s %6s %5s %5s %5s %8s %10s' % ('instantiation', 'total', 'before', 'after', 'mfma', 'vgpr', 'sgpr', 'scratch', 'sgpr-spill'))
    for r in sorted(rows):
        print('%-34s %6d %7d %6d %5d %5s %5s %8s %10d' % r)


if __name__ == '__main__':
    main()
