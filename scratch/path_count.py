import re,sys,collections
lines=[l.rstrip('\n') for l in open(sys.argv[1])]
ins=[]  # (addr, op, text)
for l in lines:
    m=re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):',l)
    if m: ins.append((int(m.group(3),16),m.group(1),m.group(2)))
addr2i={a:i for i,(a,_,_) in enumerate(ins)}
def target(i):
    a,op,t=ins[i]; off=int(t.split()[0]); 
    if off>=32768: off-=65536
    return addr2i[a+4+off*4]
# find loop head: smallest backward-branch target among the last 300 instrs
heads=[target(i) for i,(a,op,t) in enumerate(ins) if op.startswith(('s_cbranch','s_branch')) and target(i)<i-1000]
head=min(heads); print('loop head idx',head,'of',len(ins), 'back-edges', len(heads))
BIG=int(sys.argv[2]) if len(sys.argv)>2 else 100
cnt=collections.Counter(); i=head; n=0; path=[]
while n<20000:
    a,op,t=ins[i]; cnt[op]+=1; n+=1; path.append(i)
    if op=='s_endpgm': break
    if op=='s_branch':
        j=target(i)
        if j<=head+5 and j<i: break
        i=j; continue
    if op.startswith('s_cbranch'):
        j=target(i)
        if j<i:   # backward: loop back-edge of main loop => stop; inner loops: not taken
            if j<=head+60: break
            i+=1; continue
        dist=j-i
        if op in('s_cbranch_vccnz','s_cbranch_vccz','s_cbranch_scc0','s_cbranch_scc1') and dist>BIG:
            i=j; continue
        i+=1; continue
    i+=1
tot=sum(cnt.values())
valu=sum(v for k,v in cnt.items() if k.startswith('v_'))
f64=sum(v for k,v in cnt.items() if re.match(r'v_(add|mul|fma|fmac)_f64',k))
print('path instrs',tot,'VALU',valu,'f64 arith',f64,'SALU',sum(v for k,v in cnt.items() if k.startswith('s_')))
for k,v in cnt.most_common(40): print(f'{v:6d} {k}')
