#!/usr/bin/env python3
"""ISA resource lines and static instruction histogram of the MOG batch
kernels as shipped -> profiles/<tag>_mog_isa.txt (no GPU needed: hipcc -S).

    python3 tools/mog_isa_summary.py [tag]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'hk_mog.hip')
FLAGS = ('--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off '
         '-fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math '
         '-fno-slp-vectorize').split()
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'

with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, 'mog.s')
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS +
                   ['-S', '--cuda-device-only', '-o', asm, SRC], check=True,
                   stderr=subprocess.DEVNULL)
    text = open(asm).read()

name_of = lambda m: 'mog_batch2_k<%s, %s, %s, %s, %s>' % m.groups()
sym = re.compile(r'mog_batch2_kILi(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)E')
lines = []
for block in re.findall(r'  - \.agpr_count:.*?\.wavefront_size:\s+\d+', text,
                        re.S):
    m = sym.search(block)
    if not m:
        continue
    f = dict(re.findall(r'\.(\w+):\s+(\S+)', block))
    lines.append('%s: vgpr_count %s  sgpr_count %s  vgpr_spill_count %s  '
                 'sgpr_spill_count %s  private_segment_fixed_size %s  '
                 'group_segment_fixed_size %s' % (
                     name_of(m), f['vgpr_count'], f['sgpr_count'],
                     f['vgpr_spill_count'], f['sgpr_spill_count'],
                     f['private_segment_fixed_size'],
                     f['group_segment_fixed_size']))

# body of the kernel the pipeline launches
start = re.search(r'^_ZN\S*mog_batch2_kILi4ELi4ELb1ELb0ELb1E\S*:', text, re.M)
body = text[start.end():]
body = body[:body.index('s_endpgm')]
hist = collections.Counter()
for ln in body.splitlines():
    ln = ln.strip()
    if not ln or ln[0] in ';.' or ln.endswith(':'):
        continue
    op = ln.split()[0]
    if re.match(r'^[a-z_0-9]+$', op):
        hist[re.sub(r'_e(32|64)$', '', op)] += 1

out = os.path.join(ROOT, 'profiles', tag + '_mog_isa.txt')
with open(out, 'w') as fh:
    fh.write(
        'ISA summary of the MOG batch kernels as shipped (tools/mog_isa_summary.py: '
        'hipcc -S --cuda-device-only,\ngfx950, flags of hawkeye_amd/csrc/Makefile).  '
        'The pipeline launches mog_batch2_k<4, 4, true, false, true>\n(4 waves per '
        'SIMD, groups of 4 frames, tiled pixel order, no byte masks, 1-bit tiles '
        'out).\n\n')
    fh.write('\n'.join(lines) + '\n\n')
    fh.write('Static instruction histogram of mog_batch2_k<4, 4, true, false, '
             'true> (%d instructions; the\ndynamic mix per wave-frame is in '
             'profiles/mog_issue.json):\n' % sum(hist.values()))
    for op, n in hist.most_common():
        fh.write('  %-28s %5d\n' % (op, n))
print(out)
