import csv,glob,sys,os
csv.field_size_limit(1<<30)
for tag in sys.argv[1:]:
    path=glob.glob(f'gpurun_out/{tag}_fetch/**/*counter_collection.csv',recursive=True)
    if not path: print(tag,'no csv'); continue
    per={}
    for r in csv.DictReader(open(path[0],newline='')):
        if 'fwd_mfma_' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            k='dual' if 'dual' in r['Kernel_Name'] else ('p16' if 'true}' in r['Kernel_Name'][-40:] else 'single')
            per.setdefault(k,[]).append(float(r['Counter_Value']))
    print(tag,{k:(len(v),round(2*1024*sum(v)/len(v)/1e6,1)) for k,v in per.items()})
