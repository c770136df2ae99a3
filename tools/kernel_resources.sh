#!/bin/bash
# Registers, scratch, occupancy of every kernel of one translation unit (clang's kernel-resource-usage remarks).
#   tools/kernel_resources.sh frames_2048 [pattern] [extra hipcc flags...]
cd "$(dirname "$0")/../watersurfacerendering_amd/csrc" || exit 1
tu=${1:-frames_2048}; pat=${2:-k_}; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -S --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -o /dev/null $tu.hip 2> /tmp/$tu.rem
python3 - "$tu" "$pat" <<'PY'
import re, subprocess, sys
tu, pat = sys.argv[1], sys.argv[2]
txt = open(f'/tmp/{tu}.rem').read()
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()
    if pat not in name:
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    g = lambda k: re.search(re.escape(k) + r': (\d+)', b).group(1)
    print("%-110s VGPR %3s AGPR %3s scratch %4s occ %s" % (dem[:110], g('VGPRs'), g('AGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]')))
PY
