import re,sys,glob
# VALU writes SGPR -> VMEM (in an asm block) reads that SGPR: needs 5 wait states (gfx9 family); s_nop N = N+1 states, each other instruction 1
VM=re.compile(r'^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)')
for f in sorted(glob.glob('/tmp/mm_*.s')):
    L=open(f).read().split('\n'); func=''; hits={}
    inasm=False
    for i,l in enumerate(L):
        m=re.match(r'^(_Z\w+):',l)
        if m: func=m.group(1)
        if 'ASMSTART' in l: inasm=True; continue
        if 'ASMEND' in l: inasm=False; continue
        if inasm and VM.match(l):
            sregs=set()
            for a,b in re.findall(r's\[(\d+):(\d+)\]',l): sregs|=set(range(int(a),int(b)+1))
            for a in re.findall(r'\bs(\d+)\b',l): sregs.add(int(a))
            if not sregs: continue
            # walk back counting wait states
            states=0; k=i-1
            while k>0 and states<5:
                t=L[k].strip()
                if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
                    k-=1; continue
                mm=re.match(r's_nop (\d+)',t)
                if mm: states+=int(mm.group(1))+1; k-=1; continue
                # VALU writing sgpr?
                w=None
                m2=re.match(r'v_readfirstlane_b32 s(\d+)|v_readlane_b32 s(\d+)',t)
                if m2: w={int(m2.group(1) or m2.group(2))}
                m3=re.match(r'v_cmp\w* s\[(\d+):(\d+)\]',t)
                if m3: w=set(range(int(m3.group(1)),int(m3.group(2))+1))
                if w and (w&sregs):
                    hits.setdefault(func,[]).append((i+1,l.strip(),k+1,t,states)); break
                states+=1; k-=1
    n=sum(len(v) for v in hits.values())
    print(f, 'hazards:',n)
    for fn,v in hits.items():
        print('  ',fn[:70],len(v))
        for h in v[:2]: print('      ',h)
