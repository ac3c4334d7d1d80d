"""Loads, stores, atomics and s_waitcnt lines of one kernel, in program order, from a disassembled code object:
where does the compiler wait, i.e. how many DEPENDENT memory round trips does the kernel make?

  hipcc --offload-arch=gfx950 -O3 ... --offload-device-only -c kernels.hip -o k.co
  clang-offload-bundler --unbundle --type=o --input=k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=k.elf
  llvm-objdump -d k.elf > k.s
  python tools/isa_skeleton.py k.s 12k_visibilityILi8

prints `<instruction number>:<op>` - LD1/LD2/LD4 = global_load_dword/x2/x4, ST = any global store."""
import re,sys
src=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
out=[];on=False
for l in src:
    m=re.match(r'^[0-9a-f]+ <(.*)>:',l)
    if m:
        on = pat in m.group(1)
        if on: out.append('== '+m.group(1)[:80]); n=0
        continue
    if on:
        n+=1
        t=l.split('//')[0].strip()
        if re.match(r'(global_load|global_store|global_atomic|s_waitcnt vmcnt|s_waitcnt lgkmcnt\(0\)$|s_barrier|s_endpgm|buffer_|flat_|scratch_)',t):
            t=t.replace('global_load_dwordx4','LD4').replace('global_load_dwordx2','LD2').replace('global_load_dwordx3','LD3').replace('global_load_dword','LD1').replace('global_load_ushort','LDu16').replace('global_load_ubyte','LDu8').replace('s_waitcnt ','')
            t=re.sub(r' v\[?[0-9:]+\]?,.*','',t)
            t=re.sub(r'global_store_\w+.*','ST',t)
            out.append('%d:%s'%(n,t))
print(' '.join(out))
