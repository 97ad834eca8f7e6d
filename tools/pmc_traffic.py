"""HBM traffic of the dominant kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
as MI355X_MICROARCH.md (HBM section) prescribes.  Usage (on the GPU box, from the repo root):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json

Corrections applied (gfx950): FETCH_SIZE is in KiB and reports exactly half of the bytes of a wide
coalesced 16-B-per-lane stream -> doubled; WRITE_SIZE is in KiB and is exact for 16-B-per-lane
stores but UNCALIBRATED for the conv epilogue's 4-B-per-lane 128-byte row segments -> reported raw,
next to the algorithmic store bytes.  Launches are matched by (kernel, grid size) and order.
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        m = re.match(r'(void )?([\w:]+)', n)
        out[(m.group(2), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return out


def main():
    fetch, write, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    F, W = load(fetch, 'FETCH_SIZE'), load(write, 'WRITE_SIZE')
    report = {'units': 'bytes per launch', 'corrections': 'FETCH_SIZE KiB x2 (gfx950 half-count), WRITE_SIZE KiB raw',
              'kernels': {}}
    tot_b, tot_n = 0.0, 0
    for key in sorted(F):
        name, grid = key
        if not name.startswith('conv3d') or key not in W:
            continue
        n = min(len(F[key]), len(W[key]))
        fb = sum(F[key][:n]) / n * 1024 * 2
        wb = sum(W[key][:n]) / n * 1024
        report['kernels']['%s grid=%d' % (name, grid)] = {'launches': n, 'fetch_bytes': fb, 'write_bytes': wb}
        if name in ('conv3d_gather_pw', 'conv3d_gather_pf'):      # persistent / classic launch of the same unit body
            tot_b += (fb + wb) * n
            tot_n += n
    report['conv3d_gather_pw_hbm_bytes_per_launch'] = tot_b / max(1, tot_n)
    with open(dst, 'w') as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
