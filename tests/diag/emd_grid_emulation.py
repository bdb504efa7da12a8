"""CPU emulation (numpy, fp16 operands, exact products, exact accumulation) of the r06 operand records of csrc/emd.hip (point_records,
level_vectors): the algebra and the scales of the slot table, checked against float64 with and without flush-to-zero of fp16
subnormals.   python tests/diag/emd_grid_emulation.py"""
import numpy as np
F16=np.float16
def f16(v): return np.asarray(v,np.float64).astype(np.float32).astype(F16)
def split2(v):
    h=f16(v); l=f16(np.asarray(v,np.float64)-h.astype(np.float64)); return h,l
def records(pts, cen, G, T):
    S=1.2011224087864498
    n=len(pts); a=np.zeros((n,32),F16); b=np.zeros((n,32),F16)
    v=(pts.astype(np.float64)-cen.astype(np.float64))*S*G
    ip=np.rint(v); fp=v-ip
    N=(ip*ip).sum(1); R=(2*ip*fp+fp*fp).sum(1)
    Nh=np.floor(N/2048); Nl=N-2048*Nh
    sI=2.0**(T-4); sC=2.0**(T-3); sS=2.0**(T+14); sH=2.0**(T//2)
    a[:,0]=f16(-16*Nh); a[:,1]=f16(-16*Nl); a[:,2]=f16(-32768.0); a[:,3]=f16(-16.0)
    a[:,8],a[:,16]=split2(-8*R); a[:,9]=a[:,17]=f16(-sC)
    b[:,0]=f16(2048*sI); b[:,1]=f16(sI); b[:,2]=f16(Nh*sI); b[:,3]=f16(Nl*sI)
    b[:,8]=b[:,16]=f16(sC); b[:,9],b[:,17]=split2(8*R)
    for u in range(3):
        a[:,4+u]=f16(32*ip[:,u]); a[:,10+u]=a[:,18+u]=f16(2*ip[:,u]*2.0**-14)
        a[:,13+u],a[:,21+u]=split2(fp[:,u]*sS); a[:,24+u]=f16(2*fp[:,u]*sH)
        b[:,4+u]=f16(ip[:,u]*sI); b[:,10+u],b[:,18+u]=split2(fp[:,u]*sS)
        b[:,13+u]=b[:,21+u]=f16(2*ip[:,u]*2.0**-14); b[:,24+u]=f16(fp[:,u]*sH)
    return a,b
def level_vec(j,rows):
    e=2*(j-7); eF=max(e,-14); F=2.0**eF; f=2.0**(e-eF); h=2.0**(j-7)
    own,oth=(F,f) if rows else (f,F)
    ints=[own]*8; mixed=[own,oth,oth,oth,oth,own,own,own]; both=[h]*8
    return np.array(ints+mixed+mixed+both)
def exponents(x1,x2,j,ftz=False,mask=0xffffffff):
    cen=x1.astype(np.float32).mean(0).astype(np.float32)
    S=1.2011224087864498
    mx=max(np.abs(x1-cen).max(),np.abs(x2-cen).max())
    lim=1000.0/(1.2012*max(mx,1e-30)); g=min(11,int(np.floor(np.log2(lim))))
    G=2.0**g; T=14-2*g
    _,B=records(x1,cen,G,T); A,_=records(x2,cen,G,T)
    va=level_vec(j,True); vb=level_vec(j,False)
    As=(A.astype(np.float64)*va).astype(F16); Bs=(B.astype(np.float64)*vb).astype(F16)
    if ftz:
        As=np.where(np.abs(As.astype(np.float64))<2.0**-14,0,As.astype(np.float64)); Bs=np.where(np.abs(Bs.astype(np.float64))<2.0**-14,0,Bs.astype(np.float64))
    keep=np.array([(mask>>k)&1 for k in range(32)],np.float64)
    return (As.astype(np.float64)*keep)@Bs.astype(np.float64).T, g, T     # (m rows, n cols) exact accumulate
if __name__=="__main__":
    rng=np.random.default_rng(5)
    a=(rng.random((700,3),dtype=np.float32)-0.5); b=(rng.random((333,3),dtype=np.float32)-0.5)
    d2=((b.astype(np.float64)[:,None,:]-a.astype(np.float64)[None,:,:])**2).sum(2)
    for ftz in (False,True):
      for j in (7,5,3,0,-1):
        e,g,T=exponents(a,b,j,ftz); ref=-(4.0**j)*1.4426950408889634*d2
        live=ref>-150
        print("ftz",ftz,"j",j,"g",g,"T",T,"max err", np.abs(e-ref)[live].max())
