#!/usr/bin/env python3
"""Print the figures DESIGN.md section 5 / README / BASELINE.md quote from the artefacts of a round:
python tools/profile_numbers.py [profiles/r5]"""
import csv, json, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r5'


def line(name):
    with open(os.path.join(d, name)) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


for f in ('bench_k12_n1.json', 'bench_k12_n1_under_rocprof.json', 'bench_k12_n1_after_profile.json', 'bench_k15_n1.json', 'bench_k15_n1_under_rocprof.json',
          'bench_k9_n1.json', 'bench_k11_n1.json', 'bench_k13_n1.json', 'bench_k14_n1.json', 'matrix_k12_P64_prod_bench.json',
          'matrix_k12_P64_euclidean_bench.json', 'matrix_k12_P64_sum_bench.json'):
    try:
        b = line(f)
    except OSError:
        continue
    r = b['roofline']
    print('%-40s %8.1f %s  %7.3f ms/step  src_sha %s' % (f, b['value'], b['unit'], b['ms_per_step'], b.get('src_sha')))
    print('    dominant %s %.3f ms  frac %.4f  pipeline_frac %s  traffic %s  traffic_step %s' % (r.get('kernel'), r.get('avg_launch_ms', 0.0), r.get('frac', 0.0),
                                                                                 r.get('pipeline_frac'), r.get('traffic'), r.get('traffic_step')))
    for name, e in (b.get('extra') or {}).items():
        er = e.get('roofline') or {}
        print('    extra %-16s %s' % (name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k in (
            'ms_per_step', 'value', 'checksum_ok', 'parity_max_rel_vs_oracle', 'parity_pairs', 'h2d_s', 'kernel_s', 'd2h_s', 'overlapped_host_feed_s',
            'serial_Gbases_per_s', 'overlapped_Gbases_per_s', 'error')}))
        if er:
            print('          dominant %s frac %.4f kernels %s' % (er.get('kernel'), er.get('frac', 0.0), {k: round(v, 2) for k, v in er.get('kernels_ms_per_step', {}).items()}))
    print('    kernels', {k: round(v, 2) for k, v in r.get('kernels_ms_per_step', {}).items()})
    c = b.get('cpu_baseline')
    if c and 'all_cores' in c:
        print('    cpu 1 thread %.3f  all cores %.3f (%d threads; %s)  python loop %.4f' % (
            c['value'], c['all_cores']['value'], c['all_cores']['cores'], c['all_cores']['seconds_by_threads'], c['python_reference_loop']['value']))
for f in ('pmc_hbm_traffic.json', 'pmc_hbm_traffic_k15.json'):
    try:
        p = json.load(open(os.path.join(d, f)))
    except OSError:
        continue
    print(f, 'head', p.get('head'), 'src_sha', p.get('src_sha'), 'input bytes per launch', p['input_bytes_per_launch_avg'])
    for k, v in p['kernels'].items():
        if 'quad' in k or 'balance' in k:
            print('    %-60s read %.3f GB  written %.3f GB  (per input byte %.3f)' % (
                k[:60], 2 * v.get('FETCH_SIZE', 0) * 1024 / 1e9, v.get('WRITE_SIZE', 0) * 1024 / 1e9,
                v['hbm_bytes_per_dispatch_corrected'] / p['input_bytes_per_launch_avg']))
for f in ('pmc_lds_quad.json', 'pmc_lds_quad_k15.json'):
  try:
    c = json.load(open(os.path.join(d, f)))
  except OSError:
    continue
  print(f)
  kib = (6.04e9 if 'k15' in f else 3.02e9) / 1024.0     # input per launch of the counter passes (tools/profile_round.sh: 40 M / 20 M reads)
  for k, v in c['kernels'].items():
    if 'quad' in k and v.get('SQ_WAVE_CYCLES') and v.get('SQ_LDS_IDX_ACTIVE'):
        cyc = v['SQ_BUSY_CYCLES'] / 32
        print('    %-44s VALU/KiB %.0f  LDS/KiB %.1f  VALU issue %.1f %% of wave cycles  waiting %.0f %%  LDS busy %.0f %% (%.0f %% conflicts)' % (
            k[:44], v['SQ_INSTS_VALU'] / kib, v['SQ_INSTS_LDS'] / kib, 100 * v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES'],
            100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], 100 * v['SQ_LDS_IDX_ACTIVE'] / 256 / cyc, 100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']))
for f in ('bench_k12_kernel_stats.csv', 'bench_k15_kernel_stats.csv', 'matrix_k12_P64_prod_kernel_stats.csv', 'matrix_k12_P64_sum_kernel_stats.csv', 'matrix_k12_P64_euclidean_kernel_stats.csv'):
    if not os.path.exists(os.path.join(d, f)):
        continue
    print(f)
    for r in csv.DictReader(open(os.path.join(d, f))):
        if float(r['Percentage']) > 0.4:
            print('    %-60s calls %4s  avg %.3f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e6))
for f in ('skewbench_k12.log', 'skewbench_k13.log'):
    if not os.path.exists(os.path.join(d, f)):
        continue
    print(f)
    for ln in open(os.path.join(d, f)):
        if 'Gbases/s' in ln:
            print('    ' + ln.rstrip())
