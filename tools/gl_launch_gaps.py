import csv,glob,sys
for tag in sys.argv[1:]:
    f=glob.glob('gpurun_out/%s/**/*kernel_trace.csv'%tag,recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'gl_stream_kernel<0' in r['Kernel_Name'] and 'false, 3, false' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
    gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(rows,rows[1:]) if 0 <= int(b['Start_Timestamp'])-int(a['End_Timestamp']) < 200000]
    n=len(durs)
    import statistics as st
    print(tag, 'launches',n,'dur mean %.1f median %.1f us'%(sum(durs)/n, st.median(durs)), 'gap mean %.1f median %.1f us (n=%d)'%(sum(gaps)/len(gaps), st.median(gaps), len(gaps)))
