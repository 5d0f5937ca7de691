#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile.sh into the committed summaries
under profiles/: the kernel stats table and mog_traffic.json (HBM bytes per
launch of mog_batch2_k = 2 x FETCH_SIZE + WRITE_SIZE, in KiB units, the 2x
being the gfx950 wide-read correction of MI355X_MICROARCH.md, §HBM)."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join('gpurun_out', 'prof_' + tag)
os.makedirs('profiles', exist_ok=True)


def counter(name, counter_name, frames=128):
    path = os.path.join(src, name, 'pmc_counter_collection.csv')
    rows = [r for r in csv.DictReader(open(path))
            if 'mog_batch2_k' in r['Kernel_Name'] and
            r['Counter_Name'] == counter_name]
    vals = [float(r['Counter_Value']) for r in rows]
    return vals


fetch = counter('pmc_fetch', 'FETCH_SIZE')
write = counter('pmc_write', 'WRITE_SIZE')
fetch_kib = sum(fetch) / len(fetch)
write_kib = sum(write) / len(write)
px = 1920 * 1080
per_launch = int(round((2 * fetch_kib + write_kib) * 1024))
out = {
    'kernel': 'mog_batch2_k', 'frames_per_launch': 128,
    'launches_sampled': len(fetch),
    'FETCH_SIZE_KiB': fetch_kib, 'WRITE_SIZE_KiB': write_kib,
    'bytes_per_launch': per_launch,
    'bytes_per_pixel_frame': per_launch / (px * 128.0),
    # frames in, masks out as 1-bit tiles, the 5 weight planes and component 0 read and
    # written; the planes of components never created are not touched
    'expected_bytes_per_launch': (3 * 128 + 128 // 8 + 2 * 4 * (5 + 7)) * px,
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate '
              'passes), profiles/{}_pmc_*.csv; FETCH_SIZE doubled per '
              'MI355X_MICROARCH.md (gfx950 wide reads count half)'.format(tag),
}
with open('profiles/mog_traffic.json', 'w') as fh:
    json.dump(out, fh, indent=1)
shutil.copy(os.path.join(src, 'trace', 'bench_kernel_stats.csv'),
            'profiles/{}_kernel_stats.csv'.format(tag))
shutil.copy(os.path.join(src, 'bench_under_trace.json'),
            'profiles/{}_bench_under_trace.json'.format(tag))
for name in ('pmc_fetch', 'pmc_write'):
    path = os.path.join(src, name, 'pmc_counter_collection.csv')
    rows = [r for r in csv.DictReader(open(path))]
    keep = ['Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count',
            'Accum_VGPR_Count', 'SGPR_Count', 'Counter_Name', 'Counter_Value',
            'Start_Timestamp', 'End_Timestamp']
    with open('profiles/{}_{}.csv'.format(tag, name), 'w', newline='') as fh:
        w = csv.DictWriter(fh, keep)
        w.writeheader()
        for r in rows:
            if 'rocclr' in r['Kernel_Name']:
                continue
            r = {k: r[k] for k in keep}
            r['Kernel_Name'] = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0]
            w.writerow(r)
print(json.dumps(out, indent=1))
