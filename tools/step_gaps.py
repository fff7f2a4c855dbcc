"""Per-step spans and idle gaps of the main stream in a rocprofv3 kernel trace of bench.py."""
import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1][:24]) for r in rows)
main = [e for e in ev if not e[2].startswith(('group', 'default', '__amd', 'void rocprim', 'rocprim'))]
ref = [i for i, e in enumerate(main) if e[2].startswith('ffm_refresh')]
spans, gaps = [], []
for a, b in zip(ref[:-1], ref[1:]):
    spans.append((main[b][0] - main[a][0]) / 1e3)
    gaps.append((main[b][0] - max(e[1] for e in main[a:b])) / 1e3)
print("spans us:", [round(x) for x in spans])
print("idle gap before next refresh us:", [round(x) for x in gaps])
cp = glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)
if cp:
    c = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(cp[0]))]
    print("copies: n=%d mean %.1f us max %.1f us" % (len(c), sum(e - s for s, e in c) / len(c) / 1e3, max(e - s for s, e in c) / 1e3))
