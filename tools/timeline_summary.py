#!/usr/bin/env python3
"""Summarises a work-item timeline written by a library built with
tools/timeline_build.py (RT_TIMELINE_OUT=<file>): item durations, waves in flight
over time, waves per SIMD in steady state by XCD / shader engine / CU / SIMD, and the
idle time between two items on one hardware wave slot.

  python tools/timeline_summary.py /tmp/tl.bin
"""
import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,3)
st=a[:,0].astype(np.int64); en=a[:,1].astype(np.int64)
t0=st.min(); st=(st-t0)/100.0; en=(en-t0)/100.0   # us (100 MHz)
dur=en-st
print("items",len(a),"span us",en.max(),"dur mean %.1f min %.1f max %.1f p50 %.1f p99 %.1f"%(dur.mean(),dur.min(),dur.max(),np.percentile(dur,50),np.percentile(dur,99)))
# concurrency over time
T=en.max(); nb=100
edges=np.linspace(0,T,nb+1)
conc=np.zeros(nb)
for i in range(nb):
    lo,hi=edges[i],edges[i+1]
    ov=np.clip(np.minimum(en,hi)-np.maximum(st,lo),0,None).sum()/(hi-lo)
    conc[i]=ov
print("concurrency (waves in flight) per 1% of time:", np.round(conc[:5]),"...",np.round(conc[45:50]),"...",np.round(conc[-8:]))
print("mean concurrency %.0f; slots if 7/SIMD: 7168, 8/SIMD: 8192"%(dur.sum()/T))
hw=a[:,2]; xcc=(hw>>np.uint64(32)).astype(np.int64)&0xf
hwid=(hw&np.uint64(0xffffffff)).astype(np.int64)
print("xcc counts",np.bincount(xcc))
# per-xcc end times
for x in range(8):
    m=xcc==x
    if m.any(): print("xcc",x,"items",m.sum(),"last end %.0f"%en[m].max(),"busy sum %.0f"%dur[m].sum())
# gap between an item's start and the preceding end on the same slot cannot be known; look at start-time distribution
order=np.argsort(st)
print("first 8192 starts within us:", st[order[8191]] if len(st)>8192 else None, " first 7168:", st[order[7167]])
# ---- per-SIMD occupancy and the static part of the placement
se=(hwid>>13)&7; cu=(hwid>>8)&15; simd=(hwid>>4)&3; wave=hwid&15
key=((xcc*8+se)*16+cu)*4+simd
uk=np.unique(key)
print("distinct SIMDs seen:",len(uk), "distinct (xcc,se):",len(np.unique(xcc*8+se)), "cu ids",np.unique(cu), "se ids",np.unique(se), "wave ids", np.unique(wave))
busy=np.bincount(key,weights=dur,minlength=key.max()+1)[uk]/T
print("per-SIMD mean waves: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f"%(busy.min(),np.percentile(busy,10),np.median(busy),np.percentile(busy,90),busy.max()))
cnt=np.bincount(key,minlength=key.max()+1)[uk]
print("items per SIMD: min %d median %d max %d"%(cnt.min(),np.median(cnt),cnt.max()))
# steady-state window 20%..80% of the span
lo,hi=0.2*T,0.8*T
ov=np.clip(np.minimum(en,hi)-np.maximum(st,lo),0,None)
b2=np.bincount(key,weights=ov,minlength=key.max()+1)[uk]/(hi-lo)
print("steady-state per-SIMD waves: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f mean %.3f"%(b2.min(),np.percentile(b2,10),np.median(b2),np.percentile(b2,90),b2.max(),b2.mean()))
# gap between consecutive items on the same hardware wave slot
slot=key*16+wave
o=np.lexsort((st,slot))
s_sorted=slot[o]; st_s=st[o]; en_s=en[o]
same=s_sorted[1:]==s_sorted[:-1]
gaps=(st_s[1:]-en_s[:-1])[same]
print("gap between items on one wave slot (us): mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f, negative %d"%(gaps.mean(),np.percentile(gaps,50),np.percentile(gaps,90),np.percentile(gaps,99),gaps.max(),(gaps<0).sum()))
print("slots used:",len(np.unique(slot)))
# ---- which units run below their share
def by(name, ids):
    out=[]
    for v in np.unique(ids):
        m=np.isin(key, np.unique(key[ids==v]))
        sel=np.isin(uk, np.unique(key[ids==v]))
        out.append("%d:%.2f"%(v, b2[sel].mean()))
    print("steady-state waves per SIMD by", name, " ".join(out))
by("simd", simd); by("cu", cu); by("se", se); by("xcc", xcc)
low=uk[b2<5.5]
print("SIMDs below 5.5 waves:", len(low), "as (xcc,se,cu,simd):", [((k>>9)&7,(k>>6)&7,(k>>2)&15,k&3) for k in low[:24]])
