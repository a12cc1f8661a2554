"""Instruction mix of the MFMA-carrying basic blocks of every kernel in a `-save-temps` .s file (static counts).

    python tools/isa_mix.py file.s [name filter] [--per-score N]
--per-score N: N = scores one lane handles per trip of the block (attention main loops: 64 x 32 scores per wave step =
32 per lane); prints the block's vector issue slots per score with a transcendental counted as 5/3 of a plain VALU
instruction (MI355X_MICROARCH.md, "Two waves per SIMD" item 3) -- the measured counterpart of bench.py's `valu` roof,
which prices only the instructions no formulation of the step can drop."""
import re
import sys
from collections import Counter

per_score = 0
if "--per-score" in sys.argv:
    i_ = sys.argv.index("--per-score")
    per_score = int(sys.argv[i_ + 1])
    del sys.argv[i_:i_ + 2]
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
funcs = re.split(r'\n(?=_Z\w+:)', s)
for f in funcs:
    m = re.match(r'(_Z\w+):', f)
    if not m or pat not in m.group(1):
        continue
    # basic blocks: cut at labels AND behind branches (a rare-path body that follows a conditional branch is its own block)
    blocks = []
    for chunk in re.split(r'\n(?=\.LBB\d+_\d+:)', f):
        name = chunk.split(':')[0] if chunk.startswith('.LBB') else 'entry'
        parts = re.split(r'(?<=\n)(?=\ts_cbranch|\ts_branch)', chunk)
        acc = ''
        k = 0
        for part in parts:
            lines_ = part.split('\n')
            if lines_ and lines_[0].startswith(('\ts_cbranch', '\ts_branch')):
                acc += lines_[0] + '\n'
                blocks.append((name + ('' if k == 0 else f'+{k}'), acc))
                k += 1
                acc = '\n'.join(lines_[1:])
            else:
                acc += part
        blocks.append((name + ('' if k == 0 else f'+{k}'), acc))
    for bname, b in blocks:
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
        print(m.group(1)[:44], bname[:14], len(ins), dict(c))
        vc = Counter(i for i in ins if i.startswith('v_') and not i.startswith('v_mfma'))
        print('     ', vc.most_common(16))
        dc = Counter(i for i in ins if i.startswith('ds_'))
        print('     ', dict(dc))
        if per_score and nm >= 16:
            ntr = sum(v for k, v in vc.items() if k.startswith(TRANS))
            slots = (c['valu'] - ntr) + ntr * 5.0 / 3.0
            print(f"      per score ({per_score} per lane and trip): {c['valu'] / per_score:.2f} VALU instructions = {slots / per_score:.2f} issue slots "
                  f"(transcendental 5/3), {c['ds'] / per_score:.2f} LDS, {c['mfma'] / per_score:.3f} MFMA, {len(ins) / per_score:.2f} instructions in all")
