"""Condense the per-op PMC passes of profiles/pmc_probe.sh (gpurun_out/pmc_<op>/) into one table: MFMA busy as a fraction of the
kernel's duration per SIMD, VALU issue / dependency wait / LDS conflict fractions of resident-wave time, FETCH / WRITE bytes.
usage: python profiles/pmc_summary.py <op> [<op> ...] > profiles/rN/pmc_top_kernels.txt"""
import csv, glob, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print('# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): fraction of the kernel duration in which a SIMD\'s')
print('# matrix pipe executes (32 cycles per v_mfma_f32_32x32x16_bf16).  VALU / wait-inst / wait-any / LDS-conflict: fractions of SQ_WAVE_CYCLES.')
print('# FETCH_SIZE with the gfx950 x2 correction (MI355X_MICROARCH.md), both from their own passes, per dispatch.')
print('# %-14s %-46s %8s %9s %8s %9s %9s %9s %9s %9s' % ('op', 'kernel', 'Mcycles', 'MFMA busy', 'VALU', 'wait-inst', 'wait-any', 'LDS confl', 'fetch GB', 'write GB'))
for op in sys.argv[1:]:
    out = os.path.join(ROOT, 'gpurun_out', 'pmc_' + op)
    agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'rvt' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
            if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'): agg[k][r['Counter_Name'] + '_n'] += 1
    for k, c in agg.items():
        if c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) == 0: continue          # reduction / fold helpers
        d = max(n[k], 1); wc = c['SQ_WAVE_CYCLES']; gui = c['GRBM_GUI_ACTIVE'] / d / 8
        name = k.replace('_ZN3rvt', '').replace('rvt::', '')[:46]
        fe = c.get('FETCH_SIZE', 0) / max(c.get('FETCH_SIZE_n', 1), 1) * 1024 * 2 / 1e9
        wr = c.get('WRITE_SIZE', 0) / max(c.get('WRITE_SIZE_n', 1), 1) * 1024 / 1e9
        print('  %-14s %-46s %8.2f %8.1f %% %6.1f %% %7.1f %% %7.1f %% %7.1f %% %9.2f %9.2f' % (
            op, name, gui / 1e6, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / d / 1024 / gui, 100 * c['SQ_ACTIVE_INST_VALU'] / wc,
            100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_LDS_BANK_CONFLICT'] / wc, fe, wr))
