"""Long HIP API calls in a rocprofv3 --hip-trace csv: per function count, total, and the longest."""
import csv
import sys
import collections
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), 'calls; columns', list(rows[0].keys()))
by = collections.defaultdict(list)
for r in rows:
    by[r['Function']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('%-40s %7s %10s %8s %8s %8s' % ('function', 'calls', 'total ms', 'p50 us', 'p99 us', 'max us'))
for f, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    v = np.array(v)
    print('%-40s %7d %10.1f %8.1f %8.1f %8.1f' % (f, len(v), v.sum() / 1e3, np.median(v), np.percentile(v, 99), v.max()))
# the thread with most hipLaunchKernel / module launches of MOG = detection thread: its long calls
tid = collections.Counter(r['Thread_Id'] for r in rows if 'Launch' in r['Function'])
print('launches per thread', tid.most_common(5))
det = tid.most_common(1)[0][0]
long_calls = [(int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Function'])
              for r in rows if r['Thread_Id'] == det and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 800000]
print('calls > 800 us on thread', det, ':', len(long_calls))
rows_t = sorted(rows, key=lambda r: int(r['Start_Timestamp']))
for t0, d, f in long_calls[:12]:
    print('  %-28s %8.0f us; other threads inside HIP meanwhile:' % (f, d))
    seen = collections.Counter()
    for r in rows_t:
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if b > t0 and a < t0 + d * 1e3 and r['Thread_Id'] != det and b - a > 200000:
            seen[(r['Function'], round((b - a) / 1e3))] += 1
    print('     ', seen.most_common(8))
