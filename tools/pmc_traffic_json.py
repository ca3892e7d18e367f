"""profiles/rNN_bench_pmc.md (tools/pmc_summary.py) -> profiles/rNN_pmc_traffic.json: per-kernel fabric / HBM bytes per launch
and MFMA busy fraction, the numbers bench.py looks up for its roofline object."""
import glob, hashlib, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
lines = [l for l in open(src) if l.startswith('|')]
cols = [c.strip() for c in lines[0].strip().strip('|').split('|')]
out = {'note': 'rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE and the SQ groups in separate runs; tools/run_pmc_bench.sh) of '
               '`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras`; per-dispatch averages. bytes = (2*FETCH_SIZE + '
               'WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); this is '
               'L2<->fabric traffic, i.e. HBM plus Infinity-Cache (MALL) hits. mfma_busy_frac = 32*SQ_INSTS_MFMA / (GRBM_GUI_ACTIVE/8 XCDs '
               '* 1024 SIMDs).',
       'kernels': {}}
def num(s):
    try:
        return float(s)
    except ValueError:
        return None
for l in lines[2:]:
    cells = [c.strip() for c in l.strip().strip('|').split('|')]
    rec = dict(zip(cols, cells))
    name = rec['kernel'].strip('`')
    f, w = num(rec.get('FETCH_SIZE', '-')), num(rec.get('WRITE_SIZE', '-'))
    if f is None or w is None:
        continue
    mf, ga = num(rec.get('SQ_INSTS_MFMA', '-')) or 0.0, num(rec.get('GRBM_GUI_ACTIVE', '-'))
    b = (2 * f + w) * 1024
    out['kernels'][name] = {'fetch_kb': f, 'write_kb': w, 'l2_fabric_bytes_per_launch': b, 'hbm_bytes_per_launch': b,
                            'SQ_LDS_BANK_CONFLICT': num(rec.get('SQ_LDS_BANK_CONFLICT', '-')) or 0.0, 'SQ_INSTS_MFMA': mf,
                            'GRBM_GUI_ACTIVE': ga, 'mfma_busy_frac': round(32 * mf / (ga / 8 * 1024), 4) if ga else None}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, 'wdno_amd', 'csrc', '*.h*'))):      # *.hip and *.h: what the profiled library was built from
    h.update(open(f, 'rb').read())
out['csrc_sha16'] = h.hexdigest()[:16]          # bench.py marks the looked-up traffic stale when the sources have changed since
json.dump(out, open(dst, 'w'), indent=1)
print('wrote', dst, len(out['kernels']), 'kernels')
