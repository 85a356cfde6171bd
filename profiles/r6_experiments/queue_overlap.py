import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rec = [r for r in rows if 'k_rec_mfma' in r['Kernel_Name'] or 'k_gi_gemm' in r['Kernel_Name']]
rec.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rec[0]['Start_Timestamp'])
byq = collections.Counter((r['Queue_Id'], r['Kernel_Name'][:30]) for r in rec)
print(byq)
# last 40 launches: queue, start, end
for r in rec[-40:]:
    print(r['Queue_Id'], r.get('Stream_Id'), r['Kernel_Name'][5:28], (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6)
