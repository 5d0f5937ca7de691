#!/usr/bin/env python3
"""Turn the rocprofv3 counter passes of tools/prof_mog_sq.sh (the shipped MOG
batch kernel alone on the GPU) and tools/prof_ubench_sq.sh (calibration
streams) into the committed summaries under profiles/:

  profiles/<tag>_mog_counters.csv     every counter, per scene / pass /
                                      launch, summed over the chip
  profiles/<tag>_ubench_counters.csv  the calibration: VALU instructions per
                                      cycle per SIMD of pure instruction
                                      streams, from the same counters
  profiles/mog_issue.json             what bench.py's roofline reads: VALU
                                      instructions and quad-cycles per launch
                                      of the benchmark clip, shader cycles per
                                      launch, the class mix
  profiles/mog_traffic.json           HBM bytes per launch (2 x FETCH_SIZE +
                                      WRITE_SIZE, KiB units; the 2 x is the
                                      gfx950 wide-read correction of
                                      MI355X_MICROARCH.md, HBM section)

    python3 tools/mog_profile_summary.py r02 [sq_dir_tag] [ubench_dir_tag]

HK_PROF_CHUNK = frames per launch the passes were taken with (120: what the
pipeline launches for a 1800-frame clip; tools/prof_mog_sq.sh's default).
"""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
sq_tag = sys.argv[2] if len(sys.argv) > 2 else tag
ub_tag = sys.argv[3] if len(sys.argv) > 3 else tag
SQ = os.path.join('gpurun_out', 'sq_' + sq_tag)
UB = os.path.join('gpurun_out', 'sq_ubench_' + ub_tag, 'p1',
                  'pmc_counter_collection.csv')
SIMDS = 1024                    # 256 CUs x 4
XCDS = 8                        # GRBM_GUI_ACTIVE is summed over the XCDs
PX = 1920 * 1080
WAVE_PIXELS = 128               # 64 lanes x 2 pixels
FRAMES = int(os.environ.get('HK_PROF_CHUNK', '120'))   # frames per profiled launch
os.makedirs('profiles', exist_ok=True)


def passes(scene):
    """{launch index: {counter: value summed over the chip}} of the kernel's
    launches, merged over the passes (same launches in every pass); the first
    launch (cold model, launch order not yet known) is left out."""
    merged = collections.defaultdict(dict)
    rows = []
    for p in range(1, 6):
        path = os.path.join(SQ, '{}_p{}'.format(scene, p),
                            'pmc_counter_collection.csv')
        if not os.path.exists(path):
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            if 'mog_batch2_k' not in r['Kernel_Name']:
                continue
            per[int(r['Dispatch_Id'])][r['Counter_Name']] += float(
                r['Counter_Value'])
        for k, disp in enumerate(sorted(per)):
            for name, v in sorted(per[disp].items()):
                rows.append((scene, p, k, name, v))
                if k > 0 and (name != 'GRBM_GUI_ACTIVE' or p == 1):
                    merged[k][name] = v
    return merged, rows


def mean(launches, name):
    vals = [c[name] for c in launches.values() if name in c]
    return sum(vals) / len(vals) if vals else None


all_rows = []
issue = {
    'kernel': 'mog_batch2_k', 'frames_per_launch': FRAMES, 'simds': SIMDS,
    'wave_frames_per_launch': PX // WAVE_PIXELS * FRAMES,
    'source': 'rocprofv3 --pmc passes of tools/prof_mog_sq.sh, summed over '
              'the chip, averaged over the launches after the first; '
              'profiles/{}_mog_counters.csv'.format(tag),
    'units': 'SQ_ACTIVE_INST_* and SQ_WAIT_* count quad-cycles; '
             'shader_cycles_per_launch = GRBM_GUI_ACTIVE / 8 XCDs',
    'scenes': {},
}
traffic = None
for scene in ('static', 'clip'):
    launches, rows = passes(scene)
    all_rows += rows
    if not launches:
        continue
    wf = issue['wave_frames_per_launch']
    cyc = mean(launches, 'GRBM_GUI_ACTIVE') / XCDS
    sc = {'launches': len(launches),
          'shader_cycles_per_launch': round(cyc),
          'simd_cycles_per_wave_frame': round(cyc * SIMDS / wf, 1)}
    for name in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_SALU',
                 'SQ_ACTIVE_INST_SCA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD',
                 'SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32',
                 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32',
                 'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_INT32',
                 'SQ_INSTS_VALU_INT64', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY',
                 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        v = mean(launches, name)
        if v is not None:
            sc[name] = {'per_launch': round(v),
                        'per_wave_frame': round(v / wf, 2)}
    # VALU pipe occupancy: one quad-cycle issue slot per VALU instruction
    # (two for the transcendental ones), SIMDs x shader cycles available
    sc['valu_busy'] = round(
        mean(launches, 'SQ_ACTIVE_INST_VALU') * 4 / (SIMDS * cyc), 4)
    f, w = mean(launches, 'FETCH_SIZE'), mean(launches, 'WRITE_SIZE')
    if f is not None and w is not None:
        sc['hbm_bytes_per_launch'] = int(round((2 * f + w) * 1024))
        if scene == 'clip':
            traffic = {
                'kernel': 'mog_batch2_k', 'frames_per_launch': FRAMES,
                'launches_sampled': len(launches),
                'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
                'bytes_per_launch': sc['hbm_bytes_per_launch'],
                'bytes_per_pixel_frame':
                    sc['hbm_bytes_per_launch'] / (PX * float(FRAMES)),
                'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                          '(separate passes of tools/prof_mog_sq.sh over the '
                          'benchmark clip), profiles/{}_mog_counters.csv; '
                          'FETCH_SIZE doubled per MI355X_MICROARCH.md '
                          '(gfx950 wide reads count half)'.format(tag)}
    issue['scenes'][scene] = sc

with open(os.path.join('profiles', tag + '_mog_counters.csv'), 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['scene', 'pass', 'launch', 'counter', 'value_summed_over_chip'])
    w.writerows(all_rows)
with open(os.path.join('profiles', 'mog_issue.json'), 'w') as fh:
    json.dump(issue, fh, indent=1)
if traffic is not None:
    with open(os.path.join('profiles', 'mog_traffic.json'), 'w') as fh:
        json.dump(traffic, fh, indent=1)

# -- calibration streams -------------------------------------------------
KINDS = {0: 'v_fma_f32', 1: 'v_pk_fma_f32', 2: 'v_pk_mul_f32',
         3: 'v_pk_add_f32', 11: 'v_pk_mul_f32 (SGPR pair)', 4: 'v_add_f32',
         5: 'v_max_f32', 6: 'v_cvt_f32_ubyte1', 7: 'v_rcp_f32',
         8: 'v_mov_b64', 9: 'v_cmp_lt_f32 + v_cndmask_b32',
         10: 'MOG unit-step mix', 12: 'v_mul_f32', 13: 'v_cndmask_b32 (vcc)',
         14: 'v_cmp_lt_f32 (vcc)', 15: 'v_pk_fma_f32 + v_add_f32',
         16: 'v_pk_fma_f32 + v_max_f32', 17: '2 x v_fma_f32 (dependent pair)',
         18: 's_add_u32', 19: 's_nop 0', 20: 'v_pk_fma_f32 + s_add_u32',
         21: 'v_pk_fma_f32 + 2 s_add_u32', 22: 'v_add_f32 + s_add_u32',
         23: 's_cmp + s_cbranch'}
if os.path.exists(UB):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(UB)):
        m = re.search(r'issue_k<(\d+)>', r['Kernel_Name'])
        if not m:
            continue
        key = (int(r['Dispatch_Id']), int(m.group(1)),
               int(r['Workgroup_Size']), int(r['Grid_Size']))
        per.setdefault(key, collections.defaultdict(float))[
            r['Counter_Name']] += float(r['Counter_Value'])
    best = {}
    for (disp, kind, wg, grid), c in per.items():
        waves_per_simd = grid // 64 // SIMDS
        cyc = c['GRBM_GUI_ACTIVE'] / XCDS
        row = (cyc, c['SQ_INSTS_VALU'], c['SQ_INSTS_SALU'],
               c['SQ_ACTIVE_INST_VALU'])
        k = (kind, waves_per_simd)
        if k not in best or cyc < best[k][0]:
            best[k] = row
    with open(os.path.join('profiles', tag + '_ubench_counters.csv'),
              'w') as fh:
        w = csv.writer(fh)
        w.writerow(['stream', 'waves_per_simd', 'shader_cycles',
                    'valu_insts', 'salu_insts', 'cycles_per_valu_per_simd',
                    'cycles_per_salu_per_simd', 'valu_busy'])
        for (kind, wps), (cyc, valu, salu, act) in sorted(best.items()):
            w.writerow([KINDS.get(kind, kind), wps, round(cyc), int(valu),
                        int(salu),
                        round(cyc * SIMDS / valu, 2) if valu > 1e6 else '',
                        round(cyc * SIMDS / salu, 2) if salu > 1e6 else '',
                        round(act * 4 / (SIMDS * cyc), 3)])
print(json.dumps(issue['scenes'], indent=1)[:3000])
