import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
shade=[r for r in rows if 'wf_shade' in r['Kernel_Name'] and 'true>' not in r['Kernel_Name']]
trace=[r for r in rows if 'wf_trace' in r['Kernel_Name']]
shade.sort(key=lambda r:int(r['Start_Timestamp']))
# group by stream: each stream's shade launches come in groups of 8 (kinds 0..7) after each trace
bystream=collections.defaultdict(list)
for r in shade: bystream[r.get('Stream_Id', r.get('Queue_Id'))].append(r)
tot=[0]*8; n=0
for st,lst in bystream.items():
    for i,r in enumerate(lst):
        tot[i%8]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
names=['MISS','NEE','LIGHT','LAMBERT','MODPHONG','GGX','EXPLICIT','RGL']
T=sum(tot)
for k in range(8): print("%-9s %9.2f ms %5.1f %%"%(names[k],tot[k]/1e6,100*tot[k]/T))
print("trace total %.2f ms, shade total %.2f ms"%(sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in trace)/1e6,T/1e6))
