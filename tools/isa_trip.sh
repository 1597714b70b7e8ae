#!/bin/bash
# Static instruction mix of the traversal loop (node trips + leaf passes) of render_wavefront_kernel<false,true,false,true>, the
# specialisation BASELINE's scenes run (precomputed triangles, plain shading).  Both the vector and the scalar instructions of a trip are paid for (DESIGN.md 5 "What binds":
# plain 32-bit VALU operations cost 2 cycles, packed / 64-bit / compare-to-SGPR ones 4, and the one scalar unit of a CU is
# ~45 % busy), so every edit of the loop is judged by both counts - and by a same-box A/B, the compiler's register
# allocation being what it is.
# usage: tools/isa_trip.sh [extra hipcc flags]   -> /tmp/isa/k.s plus a summary
set -e
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude \
  -Iopencl_pathtracer_amd/csrc "$@" --cuda-device-only -S opencl_pathtracer_amd/csrc/kernel_wavefront.hip -o /tmp/isa/wf.s 2>/dev/null
python3 - <<'PY'
import re
from collections import Counter
t=open('/tmp/isa/wf.s').read().split('\n')
start=[i for i,l in enumerate(t) if l.startswith('_ZN8ptmi_dev23render_wavefront_kernelILb0ELb1ELb0ELb1E')][0]
end=[i for i,l in enumerate(t) if i>start and l.startswith('.Lfunc_end')][0]
k=t[start:end]
open('/tmp/isa/k.s','w').write('\n'.join(k))
# the traversal trips are the only depth-2 loop of the kernel: count the instructions of its blocks
depth2=False; ins=[]
for l in k:
    if re.match(r'(\.LBB|; %bb\.)',l):
        depth2 = 'Depth=2' in l
    elif depth2 and re.match(r'\s+[a-z]',l):
        ins.append(l.split()[0])
c=Counter(ins)
tot=lambda p: sum(n for i,n in c.items() if i.startswith(p))
print('traversal loop: VALU %d (v_mov %d, v_cndmask %d, v_cmp %d)  SALU %d  LDS %d  VMEM %d'%(tot('v_'),tot('v_mov'),tot('v_cndmask'),tot('v_cmp'),tot('s_'),tot('ds_'),tot('global_')+tot('scratch_')))
PY
grep -A12 "Function Name: _ZN8ptmi_dev23render_wavefront_kernelILb0ELb1ELb0ELb1E" /dev/null 2>/dev/null || true
