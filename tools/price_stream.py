#!/usr/bin/env python
"""Price a kernel's vector-instruction stream with the issue costs tools/valu_bench.hip measured (review of round 5, item 2).

    python tools/price_stream.py --unit kpal_quads --kernel 'quad_scatter_kernelILi12ELi16ELi8ELi8ENS_9TableOnlyELb0' \
        --from-barrier 0 --to-barrier 1 --prices gpurun_out/r6/valu_bench.log [--dynamic 186 --steps 8]

Compiles the translation unit to gfx950 assembly (here, no GPU needed), takes the instructions between two s_barrier
instructions of the named kernel (the scatter's tile: the eight wave-steps between the barrier behind the row flush and the one
before the next), counts the VALU opcodes and multiplies by the measured cycles per wave-instruction at four waves per SIMD.
--dynamic N: the PMC count of VALU instructions per wave-step (SQ_INSTS_VALU / wave-steps): the static mix is scaled to it (the
blocks a wave skips -- spill list, riders, the direct path -- are in the static count)."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument('--unit', default='kpal_quads')
ap.add_argument('--kernel', required=True, help='substring of the mangled kernel name')
ap.add_argument('--from-barrier', type=int, default=0, help='start behind this s_barrier (0-based) of the kernel')
ap.add_argument('--to-barrier', type=int, default=1)
ap.add_argument('--prices', required=True, help='output of tools/valu_bench.hip')
ap.add_argument('--dynamic', type=float, default=0.0, help='measured VALU instructions per wave-step')
ap.add_argument('--steps', type=int, default=8, help='wave-steps the region holds')
a = ap.parse_args()

# measured cycles per wave-instruction per SIMD with 4 waves per SIMD
price = {}
for ln in open(a.prices):
    m = re.search(r'4/SIMD:\s+([0-9.]+) cyc', ln)
    if m and '1/SIMD' in ln:
        name = ln[:ln.index('1/SIMD')].strip()
        key = 'v_mov_b32 dpp' if ' dpp' in name else name.split()[0]
        if name.startswith('v_cndmask_b32') or '+' in name or name.startswith('mix') or name.startswith('s_nop'):
            key = name                      # (the select is priced from the compare + select pair, below)
        price[key] = float(m.group(1))
if 'v_cmp + v_cndmask' in price and 'v_cmp_eq_u32' in price:
    price['v_cndmask_b32'] = 2.0 * price['v_cmp + v_cndmask'] - price['v_cmp_eq_u32']    # what the select adds to the pair
default_price = price.get('v_perm_b32', 4.25)
alias = {'v_mov_b64': 'v_mov_b32', 'v_cmp_gt_u32': 'v_cmp_eq_u32', 'v_cmp_ne_u32': 'v_cmp_eq_u32', 'v_cmp_lt_u32': 'v_cmp_eq_u32',
         'v_cmp_lt_u64': 'v_cmp_gt_u64', 'v_cmp_gt_i64': 'v_cmp_gt_u64', 'v_cmp_lt_i64': 'v_cmp_gt_u64', 'v_subrev_u32': 'v_sub_u32',
         'v_readfirstlane_b32': 'v_readlane_b32', 'v_add_co_u32': 'v_add_u32', 'v_addc_co_u32': 'v_add_u32'}

asm = os.path.join('/tmp', a.unit + '.price.s')
subprocess.check_call(['hipcc'] + g.HIPCC_FLAGS + ['-S', '--cuda-device-only', '-o', asm, os.path.join(g.CSRC, a.unit + '.hip')], cwd=ROOT,
                      stderr=subprocess.DEVNULL)
lines = open(asm).read().splitlines()
start = next(i for i, ln in enumerate(lines) if ln.startswith('_Z') and a.kernel in ln and ':' in ln)
body = []
for ln in lines[start + 1:]:
    body.append(ln)
    if 's_endpgm' in ln:
        break
barriers = [i for i, ln in enumerate(body) if ln.strip().startswith('s_barrier')]
lo, hi = barriers[a.from_barrier], barriers[a.to_barrier]
ops = collections.Counter()
others = collections.Counter()
for ln in body[lo:hi]:
    t = ln.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.') or t[0].endswith(':'):
        continue
    op = re.sub(r'_(e32|e64|dpp|sdwa)$', '', t[0])
    if t[0].endswith('_dpp'):
        op += ' dpp'
    (ops if op.startswith('v_') else others)[op] += 1
static = sum(ops.values())
scale = a.dynamic * a.steps / static if a.dynamic else 1.0
print('%s: %d VALU instructions between barriers %d and %d (%d wave-steps)%s' % (a.kernel, static, a.from_barrier, a.to_barrier, a.steps,
      '; scaled to the measured %.0f per wave-step (x %.3f)' % (a.dynamic, scale) if a.dynamic else ''))
print('%-24s %6s %8s %10s' % ('opcode', 'count', 'cycles', 'cycles x n'))
total = 0.0
for op, n in ops.most_common():
    key = 'v_mov_b32 dpp' if op.endswith(' dpp') else op
    p = price.get(alias.get(key, key))
    note = ''
    if p is None:
        p, note = default_price, ' (not measured: priced as v_perm_b32)'
    total += n * p
    print('%-24s %6d %8.2f %10.1f%s' % (op, n, p, n * p, note))
print('other instruction classes in the region:', dict(collections.Counter(k.split('_')[0] for k in others.elements())))
per_step = total * scale / a.steps
print('VALU issue: %.0f cycles per region, %.0f per wave-step%s' % (total * scale, per_step, ' (dynamic)' if a.dynamic else ' (static)'))
