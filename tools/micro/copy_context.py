import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def sh(n):
    n=re.sub(r"\(anonymous namespace\)::|at::native::|void ","",n); return n[:60]
n=len(rows)
# last forward only: take last 460 kernels
sel=rows[-460:]
for i,r in enumerate(sel):
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    nm=r["Kernel_Name"]
    if ("copy" in nm or "CUDAFunctor_add" in nm or "elementwise" in nm) and d>15:
        ctx=[ (sh(x["Kernel_Name"]), round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3)) for x in sel[max(0,i-2):i+3]]
        print(round(d), ctx)
