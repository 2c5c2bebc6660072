#!/bin/bash
# Registers, scratch and occupancy of every render kernel build (here, on the CPU): tools/regs.sh [extra hipcc flags]
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function "$@" -c -Rpass-analysis=kernel-resource-usage \
  -o /tmp/rt_device_report.o rt_device.hip 2>&1 | python3 -c "
import sys, re, subprocess
rows = []; cur = None
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key, pat in (('sgpr', r'TotalSGPRs: (\d+)'), ('vgpr', r' VGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, l)
        if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = n.split('(')[0].replace('void ', '')
    if 'render_' in n or 'trace' in n:
        print('%-44s vgpr %3d  sgpr %3d  scratch %4d  occupancy %d' % (n, r.get('vgpr', -1), r.get('sgpr', -1), r.get('scratch', -1), r.get('occ', -1)))
"
