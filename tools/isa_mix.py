"""Instruction mix of the MFMA-carrying basic blocks of every kernel in a `-save-temps` .s file (static counts)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r'\n(?=_Z\w+:)', s)
for f in funcs:
    m = re.match(r'(_Z\w+):', f)
    if not m or pat not in m.group(1):
        continue
    blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', f)
    for b in blocks:
        ins = [l.strip().split()[0] for l in b.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        nm = sum(i.startswith('v_mfma') for i in ins)
        if nm < 4:
            continue
        c = Counter()
        for i in ins:
            if i.startswith('v_mfma'): c['mfma'] += 1
            elif i.startswith('v_'): c['valu'] += 1
            elif i.startswith('ds_'): c['ds'] += 1
            elif i.startswith('s_waitcnt'): c['wait'] += 1
            elif i.startswith('s_'): c['salu'] += 1
            elif i.startswith(('buffer', 'global', 'scratch')): c['vmem'] += 1
            else: c['other'] += 1
        print(m.group(1)[:44], b.split(':')[0][:12], len(ins), dict(c))
        vc = Counter(i for i in ins if i.startswith('v_') and not i.startswith('v_mfma'))
        print('     ', vc.most_common(16))
        dc = Counter(i for i in ins if i.startswith('ds_'))
        print('     ', dict(dc))
