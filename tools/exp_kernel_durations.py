import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        print("%10.1f ms  %8.1f us" % ((int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
