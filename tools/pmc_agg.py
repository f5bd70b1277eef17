#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per kernel over the launches of each pass directory.
    python tools/pmc_agg.py <out.json> <pass_dir>...        (values as rocprofv3 reports them: summed over XCDs;
FETCH_SIZE / WRITE_SIZE in KiB -- bench.py applies the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md)"""
import collections, csv, glob, json, sys
out = collections.OrderedDict()
for dname in sys.argv[2:]:
    for f in glob.glob(dname + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'node::' not in k:
                continue
            a = acc[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
        for k, cs in acc.items():
            for c, (s, n) in cs.items():
                out.setdefault(k, collections.OrderedDict())[c] = s / n
json.dump(out, open(sys.argv[1], 'w'), indent=1)
for k, cs in out.items():
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        print('%-40s HBM bytes/launch: read %.2f MB (2 x FETCH_SIZE) write %.2f MB' % (k[:40], 2 * cs['FETCH_SIZE'] * 1024 / 1e6, cs['WRITE_SIZE'] * 1024 / 1e6))
