#!/usr/bin/env python3
"""Static instruction mix of kernels in a built libtha4_hip.so (tuning aid; runs in the build container, no GPU):
unbundles the gfx950 code object, disassembles it and counts opcodes per kernel - VALU / MFMA / SALU / LDS totals, 64-bit address
adds, moves - and lists the most frequent VALU opcodes.  The straight-line epilogues of the SIREN kernels make static counts a good
proxy for the dynamic ones (round 3: the 8-instruction hi/lo split and the 9-instruction upsample taps were found this way).

  python tools/isa_mix.py talking-head-anime-4-demo_amd/csrc/libtha4_hip.so /tmp/lib.dis [kernel-name-substring ...]"""
import collections
import re
import subprocess
import sys

lib=sys.argv[1]; out=sys.argv[2]
names=sys.argv[3:] or ['level1_16_kernel','front16_kernel','level2_16p_kernel']
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy","-O","binary","--only-section=.hip_fatbin",lib,"/tmp/m.fat"],check=True)
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler","--type=o","--unbundle","--input=/tmp/m.fat","--output=/tmp/m.co","--targets=hipv4-amdgcn-amd-amdhsa--gfx950"],check=True)
dis=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump","-d","--mcpu=gfx950","/tmp/m.co"],capture_output=True,text=True,check=True).stdout
open(out,'w').write(dis)
L=dis.split('\n')
def kern(name):
    o=[];on=False
    for l in L:
        if re.match(r'^[0-9a-f]+ <.*>:',l): on = name in l; continue
        if on and l.strip(): o.append(l.split('//')[0].strip().split()[0])
    return o
for k in names:
    ins=kern(k); c=collections.Counter(ins)
    valu=sum(v for o,v in c.items() if o.startswith('v_') and not o.startswith('v_mfma'))
    print(k,'total',len(ins),'valu',valu,'salu',sum(v for o,v in c.items() if o.startswith('s_')),'u64',c['v_lshl_add_u64'],'addc',c['v_addc_co_u32_e32'],'mov',c['v_mov_b32_e32'])
    print('   ',[(o,v) for o,v in c.most_common(40) if o.startswith('v_') and not o.startswith('v_mfma')][:16])
