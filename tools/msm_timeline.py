import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
rows = [(s, e, re.sub(r"\(.*", "", n)) for n, s, e in c.execute("select name,start,end from kernels")]
rows.sort()
# last commit: from the last k_msm_hist_lds<0> cluster start to k_msm_finalize
fin = [i for i,r in enumerate(rows) if "k_msm_finalize" in r[2]]
# find the commit's finalize: the one preceded by most accum0 launches; take the second last finalize (commit of last step) heuristically
import collections
def window(end_idx):
    # walk back to previous finalize
    prev = max([i for i in fin if i < end_idx], default=0)
    return rows[prev+1:end_idx+1]
best=None
for i in fin[-4:]:
    w=window(i)
    n=sum(1 for r in w if "accum0_f9" in r[2])
    if best is None or n>=best[0]: best=(n,i,w)
n,i,w=best
t0=w[0][0]
busy=[]
for s,e,nm in w:
    if any(k in nm for k in ("accum0_f9","accumN","reduce_","finalize","gather_buckets","scatter_lds","hist_lds")):
        print("%-34s start %8.3f dur %7.3f"%(nm[:34],(s-t0)/1e6,(e-s)/1e6))
print("window span %.3f ms, accum0 launches %d"%((w[-1][1]-t0)/1e6,n))
# gaps between consecutive accum0
acc=[(s,e) for s,e,nm in w if "accum0_f9" in nm]
for a,b in zip(acc,acc[1:]): print("gap %.3f ms"%((b[0]-a[1])/1e6))
