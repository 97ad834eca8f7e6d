"""Summary of the counter passes of tools/profile_round.sh for profiles/ (developer tool).

usage: pmc_summary.py gpurun_out/prof_<tag> profiles/<tag>_pmc_summary.json
Derived figures (gfx950, MI355X_MICROARCH.md): cycles per XCD = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs);
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles per XCD x 1024 SIMDs); executed matrix FLOPs = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512;
HBM bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 counts a wide coalesced read at half its bytes) + WRITE_SIZE KiB x 1024 (exact for
16-B-per-lane stores, uncalibrated for the 4-B-per-lane epilogue stores of the convolution)."""
import collections, csv, glob, json, re, sys

root, dst = sys.argv[1], sys.argv[2]
WANT = ('conv3d_gather_pw', 'conv3d_gather_splitT', 'conv3d_wgrad4', 'linear_fwd', 'linear_wgrad', 'rowgemm_wgrad_pre', 'rowgemm_fwd_pre', 'rowgemm_k128', 'bn_apply_tiles_bev', 'split_rows_kernel', 'bn_apply', 'bn_bwd_apply',
        'bn_bwd_reduce', 'bnb_tiles', 'sparse_conv_output', 'vox_gather', 'vox_insert', 'vox_scan', 'crop_write', 'feature_sample_rows')


def short(name):
    n = name.replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)\(', n)
    return m.group(2) if m else n[:40]


def load(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (root, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if any(k.startswith(w) for w in WANT):
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


sq, sq2, sq3, fe, wr = load('pmc_sq'), load('pmc_sq2'), load('pmc_sq3'), load('pmc_fetch'), load('pmc_write')
out = {'source': root, 'kernels': {}}
for k in sorted(sq):
    c = {n: sum(v) / len(v) for n, v in sq[k].items()}
    c.update({n: sum(v) / len(v) for n, v in sq2.get(k, {}).items()})
    c.update({n: sum(v) / len(v) for n, v in sq3.get(k, {}).items()})
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
    e = {'launches_in_pass': len(sq[k].get('GRBM_GUI_ACTIVE', [])), 'cycles_per_xcd': cyc,
         'mfma_busy_frac': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) if cyc else None,
         'executed_matrix_gflop': (c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) + c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) +
                                   c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0)) * 512 / 1e9,
         'mops_f32_raw': c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0), 'mops_bf16_raw': c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0), 'mops_f16_raw': c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0),
         'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT', 0), 'lds_active_cycles': c.get('SQ_LDS_IDX_ACTIVE', 0),
         'wave_cycles': c.get('SQ_WAVE_CYCLES', 0)}
    if k in fe and 'FETCH_SIZE' in fe[k]:
        e['hbm_fetch_bytes'] = sum(fe[k]['FETCH_SIZE']) / len(fe[k]['FETCH_SIZE']) * 1024 * 2
    if k in wr and 'WRITE_SIZE' in wr[k]:
        e['hbm_write_bytes'] = sum(wr[k]['WRITE_SIZE']) / len(wr[k]['WRITE_SIZE']) * 1024
    out['kernels'][k] = e
json.dump(out, open(dst, 'w'), indent=1)
for k, e in out['kernels'].items():
    print('%-28s n=%-3d mfma busy %-6s  matrix GFLOP %-8.2f fetch %-9s write %-9s' % (
        k, e['launches_in_pass'], '%.3f' % e['mfma_busy_frac'] if e['mfma_busy_frac'] is not None else '-', e['executed_matrix_gflop'],
        '%.1f MB' % (e.get('hbm_fetch_bytes', 0) / 1e6), '%.1f MB' % (e.get('hbm_write_bytes', 0) / 1e6)))

# profiles/traffic.json: measured HBM bytes per launch of the dominant kernel, tagged with the hash of the kernel's sources so
# that bench.py prints `roofline.traffic` only for the kernel it was measured on
import hashlib, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(REPO, 'mvxnet-makise_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


# the dominant gather of the run's arithmetic: fp16x3 <2, 32, 2, WIN, 1>, bf16x6 <3, 16, 2, WIN, 0>, bf16x3 <2, 32, 2, WIN, 0>, exact f32
dom, math = None, None
for pat, m in ((r'conv3d_gather_splitT<2, 32, 2, \w+, 1>', 'fp16x3'), (r'conv3d_gather_splitT<3, 16, 2, \w+, 0>', 'bf16x6'),
               (r'conv3d_gather_splitT<2, 32, 2, \w+, 0>', 'bf16x3'), (r'conv3d_gather_pw', 'f32')):
    hit = [k for k in out['kernels'] if re.match(pat, k)]
    if hit:
        dom, math = sorted(hit, key=lambda k: -out['kernels'][k]['launches_in_pass']), m
        break
if dom and len(sys.argv) > 3:
    e = out['kernels'][dom[0]]
    split = dom[0].startswith('conv3d_gather_splitT')
    files = ['conv3d_split.hip', 'split_common.h', 'common.h'] if split else ['conv3d.hip', 'common.h']
    tj = {'kernel': dom[0], 'convmath': math, 'source_files': files, 'source_sha16': source_sha(files),
          'hbm_bytes_per_launch': e.get('hbm_fetch_bytes', 0) + e.get('hbm_write_bytes', 0),
          'fetch_bytes': e.get('hbm_fetch_bytes'), 'write_bytes': e.get('hbm_write_bytes'), 'launches_in_pass': e['launches_in_pass'],
          'corrections': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads), WRITE_SIZE KiB x 1024 raw; separate --pmc passes',
          'from': root}
    json.dump(tj, open(sys.argv[3], 'w'), indent=1)
    print('traffic ->', sys.argv[3], tj['hbm_bytes_per_launch'] / 1e6, 'MB per launch')
