"""Per-kernel averages of the counters collected by tools/pmc_run.sh (developer tool): python tools/pmc_table.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
dur = defaultdict(list)
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
        dur[name].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
for name, cs in sorted(acc.items()):
    av = {k: sum(v) / len(v) for k, v in cs.items()}
    line = '%-62s n=%d' % (name, max(len(v) for v in cs.values()))
    if name in dur:
        line += '  %.1f us' % (sum(dur[name]) / len(dur[name]))
    if 'TCC_HIT_sum' in av:
        line += '  L2 hit %.3f (req %.3g)' % (av['TCC_HIT_sum'] / max(1.0, av['TCC_HIT_sum'] + av['TCC_MISS_sum']), av['TCC_HIT_sum'] + av['TCC_MISS_sum'])
    if 'FETCH_SIZE' in av:
        line += '  fetch %.1f MB (x2 %.1f)' % (av['FETCH_SIZE'] / 1024, av['FETCH_SIZE'] / 512)
    if 'SQ_BUSY_CYCLES' in av:
        line += '  mfma_busy %.3f' % (av['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, av['SQ_BUSY_CYCLES']) / 4 if False else av['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, av['SQ_BUSY_CYCLES']))
        w = max(1.0, av.get('SQ_WAVE_CYCLES', 1.0))
        line += '  parked %.2f stalled %.2f issuing %.2f' % (av.get('SQ_WAIT_ANY', 0) / w, av.get('SQ_WAIT_INST_ANY', 0) / w, av.get('SQ_ACTIVE_INST_ANY', 0) / w)
        line += '  gui %.3g' % av.get('GRBM_GUI_ACTIVE', 0)
    print(line)
