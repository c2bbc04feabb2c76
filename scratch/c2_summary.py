"""Per-kernel duration distribution of a C2 rocprofv3 kernel trace + the bench line beside it.
usage: python scratch/c2_summary.py gpurun_out/<tag>"""
import csv, collections, json, sys
root = sys.argv[1]
b = json.load(open(root + '/bench_c2.json')); print(b['ms_per_step'], b['stage_ms'], b['window'])
d = collections.defaultdict(list)
for r in csv.DictReader(open(root + '/stats/c2_kernel_trace.csv')):
    d[r['Kernel_Name'][:30]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in d.items():
    if len(v) > 100:
        v2 = sorted(v)
        print('%-32s %4d  min %6.1f med %6.1f mean %6.1f p90 %6.1f max %6.1f' % (k, len(v), v2[0], v2[len(v2) // 2], sum(v) / len(v), v2[int(len(v2) * .9)], v2[-1]))
        tot += sum(v) / len(v)
print('sum of means %.1f us' % tot)
