#!/bin/bash
# usage: ru.sh file.hip [extra flags]  -> per-kernel resource usage
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I /root/repo/include -I /root/repo/medical-transformer_amd/csrc -c /root/repo/medical-transformer_amd/csrc/$f -o /tmp/ru/out.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r':(\d+):\d+: remark: +(.*?) \[-Rpass', l)
    if not m:
        if 'error' in l: print(l, end='')
        continue
    line,txt=m.group(1),m.group(2)
    if txt.startswith('Function Name'): print(); print(line, txt[15:105], end=' | ')
    elif any(k in txt for k in ('VGPRs:','TotalSGPRs','Spill','Scratch','Occupancy')): print(txt.replace(' [bytes/lane]','').replace(' [waves/SIMD]',''), end=' | ')
"
echo
