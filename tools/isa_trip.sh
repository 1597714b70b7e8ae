#!/bin/bash
# Static instruction mix of the traversal trip of render_wavefront_kernel<false,true,false> (the shipped
# specialisation): the kernel is VALU-issue bound, so VALU instructions per trip are what every edit is judged by.
# usage: tools/isa_trip.sh [extra hipcc flags]   -> /tmp/isa/k.s plus a summary
set -e
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude \
  -Iopencl_pathtracer_amd/csrc "$@" --cuda-device-only -S opencl_pathtracer_amd/csrc/kernel_wavefront.hip -o /tmp/isa/wf.s 2>/dev/null
python3 - <<'PY'
import re
t=open('/tmp/isa/wf.s').read().split('\n')
start=[i for i,l in enumerate(t) if l.startswith('_ZN8ptmi_dev23render_wavefront_kernelILb0ELb1ELb0E')][0]
end=[i for i,l in enumerate(t) if i>start and l.startswith('.Lfunc_end')][0]
k=t[start:end]
open('/tmp/isa/k.s','w').write('\n'.join(k))
# the trip: from the loop header to the end of the block that holds the first run of four dwordx4 loads
loads=[i for i,l in enumerate(k) if 'global_load_dwordx4' in l]
first=loads[0]
hdr=max(i for i,l in enumerate(k[:first]) if 'Loop Header' in l)
# end: first label after the loads whose block starts the path logic: heuristic = next 'global_load_dwordx4' run minus nothing
nxt=[i for i,l in enumerate(k) if i>first+3 and 's_cbranch_vccnz' in l][0]
target=k[nxt].split()[1]
tl=[i for i,l in enumerate(k) if l.startswith(target+':')][0]
te=[i for i,l in enumerate(k) if i>tl and 's_branch' in l][0]
body=k[hdr:nxt]+k[tl:te]
ins=[l.split()[0] for l in body if re.match(r'\s+[a-z]',l)]
from collections import Counter
c=Counter(ins)
valu=sum(n for i,n in c.items() if i.startswith('v_'))
mov=sum(n for i,n in c.items() if i.startswith('v_mov'))
print('lines %d..%d of /tmp/isa/k.s: VALU %d (v_mov %d, v_cndmask %d, v_cmp %d)  SALU %d  LDS %d  VMEM %d'%(hdr,nxt,valu,mov,
      sum(n for i,n in c.items() if i.startswith('v_cndmask')),sum(n for i,n in c.items() if i.startswith('v_cmp')),
      sum(n for i,n in c.items() if i.startswith('s_')),sum(n for i,n in c.items() if i.startswith('ds_')),
      sum(n for i,n in c.items() if i.startswith('global_') or i.startswith('scratch_'))))
PY
grep -A12 "Function Name: _ZN8ptmi_dev23render_wavefront_kernelILb0ELb1ELb0E" /dev/null 2>/dev/null || true
