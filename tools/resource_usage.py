#!/usr/bin/env python3
"""Registers, spills, occupancy and LDS of every kernel at HEAD (hipcc -Rpass-analysis=kernel-resource-usage): python tools/resource_usage.py > profiles/rN_resource_usage.txt"""
import subprocess, re, os
root=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fuif_amd', 'csrc')
rows=[]
for f in ['maniac_decode.hip','transforms.hip','maniac_encode.hip']:
    r=subprocess.run(['hipcc','--offload-arch=gfx950','-O3','-std=c++17','-ffp-contract=off','-Wno-unused-value','-c',os.path.join(root,f),'-o','/tmp/ru.o','-Rpass-analysis=kernel-resource-usage'],capture_output=True,text=True)
    cur=None
    for line in r.stderr.split('\n'):
        m=re.search(r'remark: Function Name: (\S+)',line)
        if m:
            name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
            name=re.sub(r'\(.*','',name).replace('fuifgpu::','').replace('(anonymous namespace)::','')
            cur={'name':name}; rows.append(cur); continue
        for key,pat in [('sgpr','TotalSGPRs: (\d+)'),('vgpr',' VGPRs: (\d+)'),('occ','Occupancy \[waves/SIMD\]: (\d+)'),('scratch','ScratchSize \[bytes/lane\]: (\d+)'),('ss','SGPRs Spill: (\d+)'),('vs','VGPRs Spill: (\d+)'),('lds','LDS Size \[bytes/block\]: (\d+)')]:
            m=re.search(pat,line)
            if m and cur is not None: cur[key]=int(m.group(1))
print("# hipcc -Rpass-analysis=kernel-resource-usage, --offload-arch=gfx950 -O3 -ffp-contract=off, the sources at HEAD (round 4)")
print("%-58s %5s %5s %5s %7s %6s %6s %6s"%("kernel","SGPR","VGPR","occ","scratch","sSpill","vSpill","LDS"))
for r in rows:
    if 'sgpr' in r: print("%-58s %5d %5d %5d %7d %6d %6d %6d"%(r['name'][:58],r['sgpr'],r['vgpr'],r['occ'],r['scratch'],r['ss'],r['vs'],r['lds']))
