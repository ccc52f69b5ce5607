#!/usr/bin/env python3
"""Compare two `hipcc -S --cuda-device-only` listings kernel by kernel (comments, directives and label numbers ignored).
    python tools/isa_diff.py before.s after.s
Used in round 4 to show that moving the experiments out of rt_trace_wave.h left the production kernels' ISA unchanged."""
import re
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
        lines = []
        for l in m.group(2).split('\n'):
            t = l.split(';')[0].rstrip()
            if not t.strip() or t.strip().startswith('.'):
                continue
            lines.append(re.sub(r'\.LBB\d+_', '.LBB_', t))
        out[m.group(1)] = lines
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = [k for k in a if k in b and a[k] == b[k]]
diff = [k for k in a if k in b and a[k] != b[k]]
print("kernels: %d before, %d after; identical %d, different %d" % (len(a), len(b), len(same), len(diff)))
for k in diff:
    print("  DIFFERENT %s: %d -> %d instructions" % (k[:110], len(a[k]), len(b[k])))
for k in a:
    if k not in b:
        print("  only before:", k[:110])
for k in b:
    if k not in a:
        print("  only after:", k[:110])
