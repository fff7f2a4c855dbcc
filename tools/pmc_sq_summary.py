import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    dur = collections.defaultdict(list)
    for fn in kt:
        for r in csv.DictReader(open(fn)):
            dur[r['Kernel_Name'].split('(')[0][-44:]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name'].split('(')[0][-44:]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
    print(d)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
        n = cnt[k]
        if v.get('SQ_INSTS_VALU', 0) / n < 1e5: continue
        us = sum(dur[k]) / max(len(dur[k]), 1) / 1000
        iv = v['SQ_INSTS_VALU'] / n
        # VALU-bound time: each wave64 VALU instruction holds a SIMD for 4 cycles; 1024 SIMDs; ~2.4 GHz
        print("  %-44s n=%d us=%7.1f valuM=%7.1f valu_floor_us=%6.1f activeM=%7.1f waveMcyc=%8.1f waitM=%8.1f waitinstM=%7.1f saluM=%6.1f waves=%d" % (
            k, n, us, iv / 1e6, iv * 4 / 1024 / 2400, v['SQ_ACTIVE_INST_VALU'] / n / 1e6, v['SQ_WAVE_CYCLES'] / n / 1e6,
            v['SQ_WAIT_ANY'] / n / 1e6, v['SQ_WAIT_INST_ANY'] / n / 1e6, v.get('SQ_INSTS_SALU', 0) / n / 1e6, v['SQ_WAVES'] / n))
