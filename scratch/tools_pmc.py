import csv, collections, sys
def summarize(path, skip_first=True):
    rows=list(csv.DictReader(open(path)))
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    dur=collections.defaultdict(list)
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if not k.startswith('k_'): continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[(k,r['Dispatch_Id'])]= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    out={}
    for k,cs in acc.items():
        out[k]={c:(sum(v[1:])/max(1,len(v[1:])) if skip_first and len(v)>1 else sum(v)/len(v)) for c,v in cs.items()}
        ds=[v for (kk,_),v in dur.items() if kk==k]
        out[k]['dur_us']=sum(ds[1:])/max(1,len(ds[1:]))/1e3 if len(ds)>1 else ds[0]/1e3
    return out
if __name__=='__main__':
    for p in sys.argv[1:]:
        o=summarize(p)
        for k in ['k_phi_partial<true>','k_distance<true>','k_hist<0>','k_hist<1>','k_hist<2>']:
            if k in o:
                print(k, {c:round(v,1) for c,v in o[k].items()})
