"""Development aid (CPU): the wildcard filter's false positives on the bench workload, simulated - for 2 / 3 / 4 bits per key in the
same 64-bit part, and by how full the part is that answers.  Inputs (made once, in a scratch directory):
    g++ -O2 -std=c++17 -ffp-contract=off -o mc_emul tests/emul/mc_emul.cpp
    zcat microbecensus_amd/data/markers.faa.gz > markers.faa
    python -c "import sys; sys.path.insert(0, '.'); from microbecensus_amd import synth; r = synth.GenomeReads(device='cpu', seed=20261001).single(20000, 150).numpy(); f = open('reads.fa', 'wb'); [f.write(b'>%d\\n' % i + row.tobytes() + b'\\n') for i, row in enumerate(r)]"
    MC_DUMP_STAGES=emul ./mc_emul markers.faa reads.fa e.m8          # -> emul.frames (the six translated, SEG-masked frames)
    python tools/wild_filter_sim.py                                   # in that directory
Round 5: 2 bits per key (the filter as built) 77.1 positive groups per read of 150 bp (17.5 true, 59.6 false - the kernel counts
77.3), 3 bits 81.9, 4 bits 87.6: the parts of the popular contexts are overloaded, more bits per key only fill them faster; the
16 % of the asks that meet a part with more than half of its bits set make 68 % of the false positives."""
import numpy as np
AA="ARNDCQEGHILKMFPSTWYV"; G=["A","KR","EDNQ","C","G","H","ILVM","FYW","P","ST"]
dense=np.full(256,20,np.int64)
for i,c in enumerate(AA): dense[ord(c)]=i; dense[ord(c.lower())]=i
grp=np.full(32,10,np.int64)
for g,s in enumerate(G):
    for c in s: grp[AA.index(c)]=g
seqs=[];cur=[]
for line in open('markers.faa'):
    if line.startswith('>'):
        if cur: seqs.append(''.join(cur)); cur=[]
    else: cur.append(line.strip())
if cur: seqs.append(''.join(cur))
def tenmers(garr):
    n=len(garr)-9
    if n<=0: return np.zeros((0,10),np.int64)
    return np.stack([garr[k:k+n] for k in range(10)],1)
idx=[]
for s in seqs:
    g=grp[dense[np.frombuffer(s.encode(),np.uint8)]]
    t=tenmers(g); t=t[(t<10).all(1)]
    idx.append(t)
idx=np.concatenate(idx); print("index 10-mers", len(idx))
n=20000; FP=172; fr=np.fromfile('emul.frames',np.uint8).reshape(n,6,FP)
q=[]
for f in range(6):
    L=(150-f%3)//3
    g=grp[np.minimum(fr[:,f,:L],31)]
    t=np.stack([g[:,k:k+L-9] for k in range(10)],2).reshape(-1,10)
    q.append(t[(t<10).all(1)])
q=np.concatenate(q); print("query positions (all 10 valid)", len(q), "per read", len(q)/n)
P=10**np.arange(10)[::-1]
def ctx(t): return t[:,0]*100000+t[:,1]*10000+t[:,2]*1000+t[:,7]*100+t[:,8]*10+t[:,9]
mid_off=[3,4,5,6]
def mix(x):
    x=(x*0x9E3779B1)&0xFFFFFFFF; x^=x>>15; x=(x*0x85EBCA77)&0xFFFFFFFF; x^=x>>13; x=(x*0xC2B2AE3D)&0xFFFFFFFF; x^=x>>16; return x
def keys(t,g):
    others=[o for o in mid_off if o!=mid_off[g]]
    c=ctx(t); k=c*1000+t[:,others[0]]*100+t[:,others[1]]*10+t[:,others[2]]
    return c,k
true_any=None
for LOG2L in (19,):
  for K,FBITS in ((2,32),(3,21),(4,16)):
    nl=1<<LOG2L
    tot_fp=0; tot_tp=0
    for g in range(4):
        ci,ki=keys(idx,g); cq,kq=keys(q,g)
        line_i=mix(ci.astype(np.uint64)&0xFFFFFFFF)>>(32-LOG2L); line_q=mix(cq.astype(np.uint64)&0xFFFFFFFF)>>(32-LOG2L)
        # exact truth
        exact=set(np.unique(ki).tolist())
        truth=np.fromiter((k in exact for k in kq.tolist()),bool,len(kq))
        filt=np.zeros((nl,K),np.uint32)
        hi=mix((ki*7+g+1).astype(np.uint64)&0xFFFFFFFF); hq=mix((kq*7+g+1).astype(np.uint64)&0xFFFFFFFF)
        ok=np.ones(len(kq),bool)
        for j in range(K):
            bi=((hi>>(5*j))%FBITS).astype(np.uint32); bq=((hq>>(5*j))%FBITS).astype(np.uint32)
            np.bitwise_or.at(filt[:,j],line_i,(np.uint32(1)<<bi))
            ok&=((filt[line_q,j]>>bq)&1).astype(bool)
        assert ok[truth].all()
        tot_fp+=(ok&~truth).sum(); tot_tp+=truth.sum()
    print("lines 2^%d k=%d fieldbits=%d: per read: positive groups %.1f (true %.1f, false %.1f)"%(LOG2L,K,FBITS,(tot_fp+tot_tp)/n,tot_tp/n,tot_fp/n))
# skew: for k=2 / 2^19: FP by how full the part is
LOG2L=19; nl=1<<LOG2L
fp_by_fill=np.zeros(65); q_by_fill=np.zeros(65)
for g in range(4):
    ci,ki=keys(idx,g); cq,kq=keys(q,g)
    line_i=mix(ci.astype(np.uint64)&0xFFFFFFFF)>>(32-LOG2L); line_q=mix(cq.astype(np.uint64)&0xFFFFFFFF)>>(32-LOG2L)
    exact=set(np.unique(ki).tolist()); truth=np.fromiter((k in exact for k in kq.tolist()),bool,len(kq))
    filt=np.zeros((nl,2),np.uint32)
    hi=mix((ki*7+g+1).astype(np.uint64)&0xFFFFFFFF); hq=mix((kq*7+g+1).astype(np.uint64)&0xFFFFFFFF)
    ok=np.ones(len(kq),bool)
    for j in range(2):
        bi=((hi>>(5*j))%32).astype(np.uint32); bq=((hq>>(5*j))%32).astype(np.uint32)
        np.bitwise_or.at(filt[:,j],line_i,(np.uint32(1)<<bi))
    for j in range(2):
        bq=((hq>>(5*j))%32).astype(np.uint32); ok&=((filt[line_q,j]>>bq)&1).astype(bool)
    pc=np.array([bin(int(x)).count('1') for x in range(1<<16)],np.uint8)
    fill=(pc[filt[:,0]&0xFFFF]+pc[filt[:,0]>>16]+pc[filt[:,1]&0xFFFF]+pc[filt[:,1]>>16]).astype(np.int64)
    fq=fill[line_q]
    np.add.at(q_by_fill,fq,1); np.add.at(fp_by_fill,fq[ok&~truth],1)
cq=np.cumsum(q_by_fill)/q_by_fill.sum(); cf=np.cumsum(fp_by_fill)/fp_by_fill.sum()
for b in (8,16,24,32,40,48,56,64): print("parts with <= %2d bits set: %.3f of the asks, %.3f of the false positives"%(b,cq[b],cf[b]))
