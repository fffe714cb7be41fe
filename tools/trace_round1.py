import csv,gzip,sys,collections
rows=list(csv.DictReader(gzip.open(sys.argv[1],'rt')))
ev=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("mzk::","")[:40]) for r in rows))
idx=[i for i,e in enumerate(ev) if "poly_degree_kernel" in e[2]]
seg=ev[idx[-3]:idx[-2]]
# proof order: degree check is in r3; rotate so the segment starts at round 1 (first wire_gather or first nttx after degree+r4/r5)
names=[e[2] for e in seg]
# find start of round 1: the first nttx_pass_kernel after the last msm_collect of the segment's r5 -> simply locate 'poly_mask_kernel' first occurrence after a collect
i0=next(i for i,e in enumerate(seg) if "poly_mask" in e[2])
i1=next(i for i,e in enumerate(seg) if i>i0 and "plonk_perm_terms" in e[2])
w=seg[i0:i1]
t0=w[0][0]
print("round-1 window: %.3f ms, %d kernels"%((w[-1][1]-t0)/1e6,len(w)))
kt=collections.Counter(); kc=collections.Counter()
for s,e,n in w: kt[n]+=e-s; kc[n]+=1
for k,v in kt.most_common(22): print("  %8.3f ms %4d  %s"%(v/1e6,kc[k],k))
for s,e,n in w:
    if e-s>60000: print("%8.3f +%7.1f us %s"%((s-t0)/1e6,(e-s)/1e3,n))
