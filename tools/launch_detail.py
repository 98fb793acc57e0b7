import json,sys
L=json.load(open(sys.argv[1]))
# two profiled steps: fold by (name, tag)
agg={}
order=[]
for e in L:
    k=(e['name'],e['tag'])
    if k not in agg: agg[k]=[0,0.0,0.0]; order.append(k)
    agg[k][0]+=1; agg[k][1]+=e['ms']; agg[k][2]+=e['flops']
rows=[]
for k in order:
    n,ms,fl=agg[k]
    rows.append((ms/2, n/2, fl/max(ms,1e-9)/1e9, k[0], k[1]))
tot=sum(r[0] for r in rows)
print("total ms/step (bracketed, incl. ~4.7us each): %.3f" % tot)
for r in sorted(rows, key=lambda r:-r[0])[:int(sys.argv[2]) if len(sys.argv)>2 else 60]:
    print("%7.3f ms x%4.1f %7.1f TF  %-52s %s" % r)
