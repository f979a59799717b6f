"""Non-native kernels of the last forward in a rocprofv3 --kernel-trace CSV, each with its neighbours (to find the
Python line that launched it).  python tools/micro/copy_context.py <kernel_trace.csv>"""
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
    native = any(k in nm for k in ("linear_", "msda_", "ffn_fused", "window_attention", "layernorm_kernel", "gn_", "sine_pos", "patch_im2col", "level_", "encoder_geometry", "row_max", "query_sine", "splitk"))
    if not native and d>12:
        ctx=[ (sh(x["Kernel_Name"]), round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3)) for x in sel[max(0,i-2):i+3]]
        print(round(d), ctx)
