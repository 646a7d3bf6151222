import itertools, random, sys
groups=[list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)),
        list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(sw, PXQ=8):
    worst=1; tot=0
    for kx in range(3):
        for h in range(2):
            for g in groups:
                cnt={}
                for lane in g:
                    n=lane&15; q=lane>>4
                    col=n+kx
                    quad=(PXQ*col + ((4*h+q)^sw[col]))%16
                    cnt[quad]=cnt.get(quad,0)+1
                m=max(cnt.values()); worst=max(worst,m); tot+=m
    return worst,tot
# backtracking over sw[0..17]
best=None
def search():
    random.seed(1)
    cur=[(c>>1)&7 for c in range(18)]
    bw,bt=conflicts(cur)
    for it in range(200000):
        i=random.randrange(18); old=cur[i]; cur[i]=random.randrange(8)
        w,t=conflicts(cur)
        if (w,t)<=(bw,bt): bw,bt=w,t
        else: cur[i]=old
        if bw==1: break
    return cur,bw,bt
cur,bw,bt=search()
print(cur,bw,bt)
# structured candidates
for name,f in [("col>>1",lambda c:(c>>1)&7),("col",lambda c:c&7),("col^col>>1",lambda c:(c^(c>>1))&7),("(col>>1)^(col>>3)",lambda c:((c>>1)^(c>>3))&7),("col*3>>1",lambda c:(c*3>>1)&7),("(col+1)>>1",lambda c:((c+1)>>1)&7)]:
    print(name, conflicts([f(c) for c in range(18)]))
