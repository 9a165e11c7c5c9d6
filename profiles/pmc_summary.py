import csv,glob,collections,sys
pat=sys.argv[1]; minus=float(sys.argv[2]); nodes=float(sys.argv[3])
out={}
for d in sorted(glob.glob(pat+'*/runc/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    per=collections.defaultdict(dict)
    for r in rows:
        dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        if 'cs_propagate_' in r['Kernel_Name'] and 'sweeps' not in r['Kernel_Name']:
            if dur>=minus: per[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value']); per[r['Dispatch_Id']]['_dur_us']=dur
    keys=set(k for v in per.values() for k in v)
    for k in keys:
        vals=[v[k] for v in per.values() if k in v]
        out[k]=sum(vals)/len(vals)
for k in sorted(out): print(f"{k:28s} {out[k]:16.0f} per-node {out[k]/nodes:10.1f}")
