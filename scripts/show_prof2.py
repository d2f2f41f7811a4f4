import csv, glob, collections, sys
vals=collections.defaultdict(dict); dur={}
for f in sorted(glob.glob('gpurun_out/prof2/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        if 'conv_' not in k or 'pack' in k: continue
        vals[k][r['Counter_Name']]=float(r['Counter_Value'])
for f in sorted(glob.glob('gpurun_out/prof2/p7/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        if k in vals: dur[k]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
names=sorted({c for v in vals.values() for c in v})
ks=list(vals)
print('%-34s'%'counter'+' '.join('%18s'%k[-18:] for k in ks))
print('%-34s'%'duration_us'+' '.join('%18.1f'%dur.get(k,0) for k in ks))
for c in names:
    print('%-34s'%c+' '.join('%18.4g'%vals[k].get(c,float('nan')) for k in ks))
