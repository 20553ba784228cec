"""Per-kernel register / spill / LDS summary from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
Usage: python tools/resusage.py UNIT [name-substring ...] [-DMACRO ...]
(compiles kpal_amd/csrc/UNIT.hip -- kpal_quads, kpal_quads2, kpal_count, kpal_vec --, no GPU needed)"""
import re
import subprocess
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ['hipcc'] + list(g.HIPCC_FLAGS) + ['-Rpass-analysis=kernel-resource-usage', '-c', '-o', '/dev/null',
                                         os.path.join(root, 'kpal_amd/csrc/%s.hip' % sys.argv[1])] + \
      [a for a in sys.argv[2:] if a.startswith('-D')]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
want = [a for a in sys.argv[2:] if not a.startswith('-D')]
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = subprocess.run(['c++filt', t.split(':', 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(.*', '', cur).replace('void ', '').replace('kpal::', '')
        rows[cur] = {}
    elif cur and ':' in t:
        a, b = t.split(':', 1)
        rows[cur][a.strip()] = b.strip()
print('%-44s %5s %5s %6s %7s %4s %7s' % ('kernel', 'VGPR', 'AGPR', 'spill', 'scratch', 'occ', 'LDS'))
for name, r in rows.items():
    if want and not any(w in name for w in want):
        continue
    print('%-44s %5s %5s %6s %7s %4s %7s' % (name[:44], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'),
                                           r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'),
                                           r.get('LDS Size [bytes/block]')))
