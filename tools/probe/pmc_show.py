import csv, collections, sys, glob
for d in sys.argv[1:]:
    print('==', d)
    for f in sorted(glob.glob(d + '/*.csv')):
        rows = list(csv.DictReader(open(f)))
        per = collections.defaultdict(dict)
        for r in rows:
            if 'mog_batch2' in r['Kernel_Name']:
                per[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        ids = sorted(per)[1:]     # skip the cold first launch
        if not ids: continue
        keys = sorted(per[ids[0]])
        print(f.split('/')[-1], {k: round(sum(per[i][k] for i in ids) / len(ids), 1) for k in keys})
