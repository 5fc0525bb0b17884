#!/bin/bash
# tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt -- VGPR / SGPR / scratch / occupancy / LDS of every kernel from the code
# object itself (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed)
cd "$(dirname "$0")/../gpu-icp-slam_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Rpass-analysis=kernel-resource-usage -c pfslam_hip.hip -o /tmp/kres.o 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    m = re.search(r'remark:\s+(TotalSGPRs|VGPRs|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)', l)
    if m and cur is not None and m.group(1) not in cur: cur[m.group(1)] = int(m.group(2))
print('# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage, csrc/pfslam_hip.hip')
print('%-64s %5s %5s %8s %9s %8s' % ('kernel', 'VGPR', 'SGPR', 'scratch', 'waves/SIMD', 'LDS B'))
seen = set()
for r in rows:
    if 'VGPRs' not in r: continue
    d = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'\(.*', '', d).replace('void ', '')
    if d in seen: continue
    seen.add(d)
    print('%-64s %5d %5d %8d %9d %8d' % (d[:64], r.get('VGPRs', 0), r.get('TotalSGPRs', 0), r.get('ScratchSize', 0), r.get('Occupancy', 0), r.get('LDS Size', 0)))
"
