"""Gaps between consecutive MOG launches in a kernel trace, and what ran in them."""
import csv
import sys
import collections
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
k = [(int(r['Start_Timestamp']) / 1e3, int(r['End_Timestamp']) / 1e3,
      r['Kernel_Name']) for r in rows]
k.sort()
mog = [(a, b) for a, b, n in k if 'mog_batch2_k' in n]
m = np.array(mog)
dur = m[:, 1] - m[:, 0]
gap = m[1:, 0] - m[:-1, 1]
per_clip_launches = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if per_clip_launches:     # equal chunks: every n-th launch ends a clip
    short = (np.arange(len(m)) % per_clip_launches) == per_clip_launches - 1
else:                     # a short last launch ends a clip
    short = dur < 150
print('MOG launches', len(m), 'duration pct 5/50/95', np.percentile(dur, [5, 50, 95]))
idx = np.where(short)[0]
print('clips', len(idx))
# the batch: the longest run of clip periods < 15 ms
starts = m[idx[:-1] + 1, 0]
per = np.diff(starts)
inb = per < 15000
print('clip period in batch: n', inb.sum(), 'median', np.median(per[inb]) / 1e3, 'mean', per[inb].mean() / 1e3)
t_lo, t_hi = starts[:-1][inb].min(), (starts[:-1][inb] + per[inb]).max()
sel = (m[:-1, 0] >= t_lo) & (m[1:, 0] <= t_hi)
w = sel & ~short[:-1]
b = sel & short[:-1]
print('in batch: within-clip gaps n', w.sum(), 'median', np.median(gap[w]), 'mean', gap[w].mean(),
      'pct 10/90/99', np.percentile(gap[w], [10, 90, 99]))
print('in batch: clip-boundary gaps n', b.sum(), 'median', np.median(gap[b]), 'mean', gap[b].mean())
busy = dur[:-1][sel].sum() / (t_hi - t_lo)
print('MOG busy fraction of the batch', busy, ' MOG duration in batch median', np.median(dur[:-1][sel & ~short[:-1]]))
per_clip = (t_hi - t_lo) / max(1, b.sum())
print('per clip: period %.2f ms = MOG %.2f + within-clip gaps %.2f + boundary gap %.2f' % (
    per_clip / 1e3, dur[:-1][sel].sum() / b.sum() / 1e3, gap[w].sum() / b.sum() / 1e3,
    gap[b].sum() / b.sum() / 1e3))
# what ran inside the large within-clip gaps
big = np.where(w & (gap > 100))[0]
names = collections.Counter()
for i in big[:200]:
    a, e = m[i, 1], m[i + 1, 0]
    for s0, s1, n in k:
        if s1 > a and s0 < e and 'mog_batch2_k' not in n:
            names[n[:50]] += 1
print('large within-clip gaps:', len(big), names.most_common(8))

# along the batch: per clip (launch groups), period, MOG time, gaps
if per_clip_launches:
    n = per_clip_launches
    first = np.arange(0, len(m) - n, n)
    print('clip  period  mog  gaps(within)  boundary   (ms)')
    for c, i in enumerate(first):
        period = m[i + n, 0] - m[i, 0]
        if period > 30000:
            print('  --- pause %.1f ms' % (period / 1e3)); continue
        if c % 4 == 0:
            print('%4d %7.2f %6.2f %7.2f %7.2f' % (c, period / 1e3, dur[i:i + n].sum() / 1e3,
                  gap[i:i + n - 1].sum() / 1e3, gap[i + n - 1] / 1e3))
