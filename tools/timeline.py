"""Prints the kernel timeline of the last full step in a rocprofv3 --kernel-trace csv."""
import csv, sys, glob
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1][:28], r.get('Queue_Id', '?')) for r in rows]
ev.sort()
# steps begin at row kernels (ffm_row_kernel<true...) ; take the third from the end
starts = [i for i, e in enumerate(ev) if e[2].startswith('ffm_refresh') or e[2].startswith('ffm_row_kernel<true')]
ref = [i for i in starts if ev[i][2].startswith('ffm_refresh')] or starts
i0, i1 = ref[-4], ref[-3]
t0 = ev[i0][0]
print("step span us:", (ev[i1][0] - t0) / 1000)
for s, e, n, q in ev[i0:i1 + 3]:
    print("%9.1f %9.1f %7.1f  q=%s %s" % ((s - t0) / 1000, (e - t0) / 1000, (e - s) / 1000, q, n))
