"""Model of the LDS-DMA issue order of the deep-pipelined full-row K loop (pv_gemm.hip, pv_gemm_fullrow_kernel<NT, true>): prints, per K-tile and
phase, how many pieces are issued after the ones the NEXT phase reads - the immediates of the counted s_waitcnt vmcnt (NP = NT / 2 phases)."""
import collections
def sim(NP, nk):
    # issue sequence of (name) per wave; each piece group = 2 ops (A: 2, Wg: 2)
    seq=[]  # list of op labels
    def issue(lbl): seq.extend([lbl,lbl])
    # prologue
    issue(('A',0));
    for g in range(NP): issue(('W',g,0))
    if nk>=2:
        issue(('A',1))
        for g in range(NP-1): issue(('W',g,1))
    waits={}
    # wait for phase 0 of tile 0: placed after prologue
    def need_count(labels):
        last=max(i for i,l in enumerate(seq) if l in labels)
        return len(seq)-1-last
    waits[('pro',)]=need_count([('A',0),('W',0,0)])
    for s in range(nk):
        for j in range(NP):
            # issues of phase j of K-tile s
            if j==0:
                if s+1<nk: issue(('W',NP-1,s+1))
            elif j==1:
                if s+2<nk: issue(('A',s+2)); issue(('W',0,s+2))
            else:
                if s+2<nk: issue(('W',j-1,s+2))
            # wait: data for next phase
            if j<NP-1: labels=[('W',j+1,s)]
            else:
                if s+1>=nk: continue
                labels=[('A',s+1),('W',0,s+1)]
            waits[(s,j)]=need_count(labels)
    return waits
for NP in (2,3,4):
    for nk in (1,2,3,4,6,8):
        w=sim(NP,nk)
        print(NP,nk,' '.join(f"{k}:{v}" for k,v in w.items()))
