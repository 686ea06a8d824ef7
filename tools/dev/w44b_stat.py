"""Instruction-class counts of k_mid_wino44b (production instantiation) in /tmp/w44b.s, split at the first / last MFMA: prologue, main loop, epilogue.  Any argument: also list the scratch accesses."""
import re,sys
s=open('/tmp/w44b.s').read()
i=s.index('_ZN3pnp4w44b13k_mid_wino44bILb0ELb0ELi0EEEvPKfPfPK15HIP_vector_typeIjLj4EES3_iiifPy:')
j=s.index('.Lfunc_end',i)
body=s[i:j].split('\n')
mf=[k for k,l in enumerate(body) if 'v_mfma' in l]
def cnt(a,b):
    c={}
    for l in body[a:b]:
        m=re.match(r'\s+([a-z_0-9]+)',l)
        if m: c[m.group(1)]=c.get(m.group(1),0)+1
    return c
for name,(a,b) in {'before':(0,mf[0]),'main':(mf[0],mf[-1]+1),'after':(mf[-1]+1,len(body))}.items():
    c=cnt(a,b); tot=sum(c.values())
    va=sum(v for k,v in c.items() if k.startswith('v_') and 'mfma' not in k and 'accvgpr' not in k)
    print(name,tot,'valu',va,{k:v for k,v in c.items() if 'scratch' in k or 'accvgpr' in k or 's_barrier' in k or k.startswith('ds_') or 'global' in k})
if len(sys.argv)>1:
    for k,l in enumerate(body):
        if 'scratch_' in l: print(k,l.strip())
