"""Instruction mix of the longest loop body of every kernel whose name contains PATTERN in a hipcc .s file (-save-temps).
python tools/count_loop_isa.py FILE.s [PATTERN]"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_Z[\w]+:)', src):
    m = re.match(r'(_Z\w+):', f)
    if not m or pat not in m.group(1):
        continue
    lines = f.split('\n')
    labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'(\.LBB\d+_\d+):', l)] if mm}
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if not best:
        print(m.group(1), 'no loop')
        continue
    body = [l.strip() for l in lines[best[0] + 1:best[1]] if l.strip() and not l.strip().startswith((';', '.'))]
    ops = collections.Counter(l.split()[0] for l in body)
    valu = sum(c for o, c in ops.items() if o.startswith('v_') and not o.startswith('v_mfma'))
    print(m.group(1), '| loop instructions', len(body), '| VALU', valu, '| MFMA', sum(c for o, c in ops.items() if o.startswith('v_mfma')))
    print('    ' + ', '.join(f'{o} {c}' for o, c in ops.most_common(24)))
    for key in ('NumVgprs', 'NumAgprs', 'TotalNumVgprs', 'Occupancy', 'ScratchSize'):
        mm = re.search(r'; %s: (\d+)' % key, f)
        if mm:
            print('    %s %s' % (key, mm.group(1)), end='')
    print()
