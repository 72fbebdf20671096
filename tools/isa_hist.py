#!/usr/bin/env python3
"""Instruction histogram of the kernels of one HIP translation unit (device ISA via hipcc -S):
   python3 tools/isa_hist.py kvazaar_amd/csrc/FILE.hip [kernel-substring]"""
import collections
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                      stderr=subprocess.DEVNULL)
s = open(out).read()
for name in re.findall(r'^(_Z\S+):', s, re.M):
    if sub not in name or "__device_stub" in name:
        continue
    a = s.index(name + ":")
    b = s.find("s_endpgm", a)
    if b < 0:
        continue
    body = s[a:b]
    ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    v = sum(n for i, n in c.items() if i.startswith('v_'))
    meta = s[b:b + 6000]
    vg = re.search(r'; NumVgprs: (\d+)', meta); oc = re.search(r'; Occupancy: (\d+)', meta); sc = re.search(r'; ScratchSize: (\d+)', meta)
    print("%s\n  instructions %d, vector %d, vgprs %s, occupancy %s, scratch %s" % (name, len(ins), v, vg and vg.group(1), oc and oc.group(1), sc and sc.group(1)))
    print("  " + ", ".join("%s:%d" % t for t in sorted(c.items(), key=lambda t: -t[1])[:45]))
    # the hottest loop: the longest span from a label to a backward branch to it
    lines = body.splitlines()
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(lines) if re.match(r'^\.?\w+:', l)}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r'\s+s_cbranch_\w+\s+(\S+)|\s+s_branch\s+(\S+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i and (best is None or i - labels[t] > best[1] - best[0]):
                best = (labels[t], i)
    if best:
        li = [l.split()[0] for l in lines[best[0]:best[1] + 1] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        lc = collections.Counter(li)
        print("  longest loop: %d instructions, %d vector, %d LDS, %d vmem" % (len(li), sum(n for i, n in lc.items() if i.startswith('v_')),
              sum(n for i, n in lc.items() if i.startswith('ds_')), sum(n for i, n in lc.items() if i.startswith(('global_', 'buffer_', 'flat_')))))
        print("  " + ", ".join("%s:%d" % t for t in sorted(lc.items(), key=lambda t: -t[1])[:45]))
