#!/usr/bin/env python3
"""Linear-scan checker over a kernel's ISA (hipcc -S --cuda-device-only): flags a VGPR that is read or overwritten while a VMEM / LDS
operation that writes it is still outstanding according to the s_waitcnt instructions in program order (branches ignored: a
straight-line approximation).  Used in round 4 to rule out a missing wait in the nondeterministic fast-path build
(profiles/NOTES.md 4.1).      python tools/isa_waitcheck.py kernel.s"""
# crude linear-scan checker: flags a VGPR read (or overwrite) while a memory op that writes it is still outstanding
src = open(sys.argv[1]).read().split('\n')
def regs(tok):
    out=[]
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2))+1))
        else: out.append(int(m.group(3)))
    return out
vm=[]; lg=[]   # lists of (line, [dest regs])
flags=0
for ln,l in enumerate(src,1):
    s=l.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
        if s.endswith(':') : pass
        continue
    s=s.split(';')[0].strip()
    op=s.split()[0]
    args=s[len(op):]
    if op=='s_waitcnt':
        m=re.search(r'vmcnt\((\d+)\)',args)
        if m:
            n=int(m.group(1)); vm=vm[max(0,len(vm)-n):] if n< len(vm) else vm
            if n==0: vm=[]
        m=re.search(r'lgkmcnt\((\d+)\)',args)
        if m:
            n=int(m.group(1)); lg=lg[max(0,len(lg)-n):] if n<len(lg) else lg
            if n==0: lg=[]
        continue
    if op in ('s_barrier',): continue
    parts=[a.strip() for a in args.split(',')]
    is_vmem = op.startswith(('global_load','buffer_load','scratch_load','flat_load'))
    is_vst = op.startswith(('global_store','buffer_store','scratch_store','global_atomic','flat_store'))
    is_ds = op.startswith('ds_')
    is_smem = op.startswith('s_load') or op.startswith('s_buffer_load')
    dest = regs(parts[0]) if parts and (is_vmem or (is_ds and not op.startswith(('ds_write','ds_add')))) else []
    reads=[]
    start = 1 if (is_vmem or (is_ds and dest) ) else (1 if op.startswith('v_') and not op.startswith(('v_cmp','v_cmpx')) else 0)
    for a in parts[start:]: reads += regs(a)
    if op.startswith('v_') and ('fmac' in op or 'mac' in op or 'v_mfma' in op and False): reads += regs(parts[0])
    writes = regs(parts[0]) if (op.startswith('v_') and not op.startswith(('v_cmp','v_cmpx'))) else []
    pend_vm = {r:x[0] for x in vm for r in x[1]}
    pend_lg = {r:x[0] for x in lg for r in x[1]}
    for r in set(reads):
        if r in pend_vm: print(f"line {ln}: READ v{r} pending VMEM from line {pend_vm[r]}: {s}"); flags+=1
        if r in pend_lg: print(f"line {ln}: READ v{r} pending LDS from line {pend_lg[r]}: {s}"); flags+=1
    for r in set(writes):
        if r in pend_vm: print(f"line {ln}: WRITE v{r} pending VMEM from line {pend_vm[r]}: {s}"); flags+=1
        if r in pend_lg: print(f"line {ln}: WRITE v{r} pending LDS from line {pend_lg[r]}: {s}"); flags+=1
    if is_vmem: vm.append((ln,dest))
    elif is_vst: vm.append((ln,[]))
    elif is_ds: lg.append((ln,dest))
    elif is_smem: lg.append((ln,[]))
print("flags",flags)
