#!/bin/bash
# usage: tools/isa.sh <file.hip>   -> /tmp/isa/<file>.s + resource table of its kernels (developer tool)
set -e
F="$1"; B=$(basename "$F" .hip)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function --cuda-device-only -S \
    -o /tmp/isa/$B.s /root/repo/mvxnet-makise_amd/csrc/$B.hip -I/root/repo/mvxnet-makise_amd/csrc -I/root/repo/include 2>&1 | grep -v "hip-link" || true
python3 - "$B" <<'P'
import re, sys
s = open('/tmp/isa/%s.s' % sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    body = m.group(2)
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [0, '?'])[1]
    print('%-110s vgpr %s agpr %s sgpr %s lds %s scratch %s' % (m.group(1)[:110], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
P
