"""Splits the compact per-launch trace of tools/profile_round5_mixed_trace.sh (trace_compact.csv.gz + trace_names.csv in the current directory) into proofs --
a proof of a stream ends at its k_fri_query -- and prints per proof: launches, summed kernel time, span, the kernels with the largest share.
Usage (in gpurun_out/r5mixed19_trace): python3 tools/trace_split_proofs.py [kernels per line = 9]"""
import gzip, csv, collections, sys
names={}
for l in open('trace_names.csv'):
    k,n=l.rstrip('\n').split(',',1); names[int(k)]=n
rows=[]
for r in csv.DictReader(gzip.open('trace_compact.csv.gz','rt')):
    rows.append((int(r['kid']),int(r['start_ns']),int(r['dur_ns']),r['queue'],r['stream']))
print(len(rows))
def short(k):
    n=names[k].replace('zk::','').replace('void ','').replace('(anonymous namespace)::','')
    return n.split('(')[0][:36]
bystream=collections.defaultdict(list)
for r in rows: bystream[r[4]].append(r)
for k,v in sorted(bystream.items()):
    g=[x for x in v if 'k_fri_query' in names[x[0]]]
    print('stream',k,len(v),'sum ms %.1f'%(sum(x[2] for x in v)/1e6),'first %.1f last %.1f'%(v[0][1]/1e6,v[-1][1]/1e6),'proofs',len(g))
for k,v in sorted(bystream.items()):
    proofs=[];cur=[]
    for x in v:
        cur.append(x)
        if 'k_fri_query' in names[x[0]]:
            proofs.append(cur);cur=[]
    for i,p in enumerate(proofs):
        tot=sum(x[2] for x in p)/1e6
        span=(p[-1][1]+p[-1][2]-p[0][1])/1e6
        agg=collections.Counter()
        for x in p: agg[short(x[0])]+=x[2]
        top=', '.join('%s %.1f'%(n,d/1e6) for n,d in agg.most_common(int(sys.argv[1]) if len(sys.argv)>1 else 9))
        print('s%s proof %d: launches %d kernel_ms %.1f span %.1f | %s'%(k,i,len(p),tot,span,top))
