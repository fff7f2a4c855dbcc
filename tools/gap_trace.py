"""Joins a rocprofv3 --hip-trace --kernel-trace capture of bench.py per dispatch: when the host
submitted each kernel of a step (API call, thread) against when the queue started it."""
import csv, glob, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
api = list(csv.DictReader(open(glob.glob(d + '/**/*hip_api_trace.csv', recursive=True)[0])))
by_corr = {r['Correlation_Id']: r for r in api}
def short(n):
    n = n.replace('void ', '').replace('ftrl_dev::', '').replace('ftrl::', '')
    if 'rocprim' in n:
        return 'sort:' + n.split('detail::')[-1][:18]
    return n.split('(')[0][:28]
kt.sort(key=lambda r: int(r['Start_Timestamp']))
ref = [i for i, r in enumerate(kt) if 'ffm_refresh_kernel' in r['Kernel_Name']]
threads = sorted({r['Thread_Id'] for r in api})
print('api threads:', threads, ' calls:', len(api))
lo = ref[len(ref) // 2]
hi = ref[min(len(ref) - 1, len(ref) // 2 + 4)]
t0 = int(kt[lo]['Start_Timestamp'])
print('%9s %9s %7s  %9s %7s  q thread kernel' % ('start', 'end', 'dur', 'submit', 'queued'))
for r in kt[lo:hi + 1]:
    a = by_corr.get(r['Correlation_Id'])
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    sub = (int(a['End_Timestamp']) - t0) / 1e3 if a else float('nan')
    print('%9.1f %9.1f %7.1f  %9.1f %7.1f  %s %s %s' % (s, e, e - s, sub, s - sub, r['Queue_Id'],
          threads.index(a['Thread_Id']) if a else '?', short(r['Kernel_Name'])))
# the host's view of the same window: every API call of the two submitting threads
print()
print('host calls in the window (start, dur us, thread, function):')
w0, w1 = int(kt[lo]['Start_Timestamp']) - 1500000, int(kt[hi]['Start_Timestamp'])
for a in sorted(api, key=lambda r: int(r['Start_Timestamp'])):
    s = int(a['Start_Timestamp'])
    if w0 <= s <= w1:
        dur = (int(a['End_Timestamp']) - s) / 1e3
        if dur >= 15:
            print('%9.1f %7.1f  %s %s' % ((s - t0) / 1e3, dur, threads.index(a['Thread_Id']), a['Function']))
# how far ahead of the queue the host runs: refresh submit vs start, per step
print()
print('refresh kernels: submit -> start (us), previous update end -> start')
prev_end = None
for i in ref:
    r = kt[i]
    a = by_corr.get(r['Correlation_Id'])
    ends = [int(x['End_Timestamp']) for x in kt[max(0, i - 40):i] if 'ffm_update' in x['Kernel_Name']]
    gap = (int(r['Start_Timestamp']) - max(ends)) / 1e3 if ends else float('nan')
    q = (int(r['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 if a else float('nan')
    print('  queued %9.1f   after update end %7.1f' % (q, gap))
