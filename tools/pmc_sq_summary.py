"""Per-kernel summary of the counters tools/pmc_sq.sh collects.  usage: pmc_sq_summary.py <tag>..."""
import csv, glob, collections, sys
for tag in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    dur = collections.defaultdict(list)
    for part in "ab":
        d = "gpurun_out/pmc_%s_%s" % (tag, part)
        for fn in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
            if part == "a":
                for r in csv.DictReader(open(fn)):
                    dur[r['Kernel_Name'].split('(')[0][-34:] + ':' + str(r.get('Grid_Size', r.get('Grid_Size_X', '')))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r['Kernel_Name'].split('(')[0][-34:] + ':' + str(r.get('Grid_Size', r.get('Grid_Size_X', '')))
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[k][r['Counter_Name']] += 1
    print("==", tag)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
        g = lambda name: v.get(name, 0.0) / max(cnt[k].get(name, 1), 1)
        if g('SQ_INSTS_VALU') < 1e6 or 'fill' in k or k.strip() == '': continue
        us = sum(dur[k]) / max(len(dur[k]), 1) / 1000
        iv = g('SQ_INSTS_VALU')
        # VALU floor: a wave64 VALU instruction holds its SIMD for 4 cycles; 1024 SIMDs; ~2.4 GHz
        print("  %-40s us=%7.1f valuM=%6.1f floor_us=%6.1f transM=%5.1f saluM=%5.1f vmemrdM=%5.2f | wave-cyc M=%7.1f wait=%4.0f%% waitinst=%4.0f%% | icache reqM=%6.1f miss=%5.2f%% dup=%5.2f%%" % (
            k, us, iv / 1e6, iv * 4 / 1024 / 2400, g('SQ_INSTS_VALU_TRANS_F32') / 1e6, g('SQ_INSTS_SALU') / 1e6, g('SQ_INSTS_VMEM_RD') / 1e6,
            g('SQ_WAVE_CYCLES') / 1e6, 100 * g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1), 100 * g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1),
            g('SQC_ICACHE_REQ') / 1e6, 100 * g('SQC_ICACHE_MISSES') / max(g('SQC_ICACHE_REQ'), 1), 100 * g('SQC_ICACHE_MISSES_DUPLICATE') / max(g('SQC_ICACHE_REQ'), 1)))
