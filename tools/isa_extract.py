"""Cut one kernel out of a hipcc -S --cuda-device-only listing and summarise it: instruction counts by kind, spills,
and (optionally) the longest loop bodies.  Usage: python tools/isa_extract.py file.s <substring of the mangled name> [out.s]"""
import re
import sys


def kernels(path):
    out, name, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                out[name] = buf
            name, buf = m.group(1), []
        elif line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            if name:
                out[name] = buf
                name, buf = None, []
        if name:
            buf.append(line)
    return out


def main():
    ks = kernels(sys.argv[1])
    key = sys.argv[2]
    for n, body in ks.items():
        if key not in n:
            continue
        text = "".join(body)
        cnt = lambda pat: len(re.findall(pat, text))
        print(n)
        print(f"  lines {len(body)}  mfma {cnt(r'v_mfma')}  ds_read {cnt(r'ds_read')}  ds_write {cnt(r'ds_write')}  "
              f"lds_dma {cnt(r'global_load_lds')}  scratch_st {cnt(r'scratch_store')}  scratch_ld {cnt(r'scratch_load')}  "
              f"accvgpr_rd {cnt(r'v_accvgpr_read')}  accvgpr_wr {cnt(r'v_accvgpr_write')}  v_mov {cnt(r'v_mov_b32')}  "
              f"s_waitcnt {cnt(r's_waitcnt')}  s_barrier {cnt(r's_barrier')}  s_nop {cnt(r's_nop')}")
        if len(sys.argv) > 3:
            open(sys.argv[3], "w").write(text)


if __name__ == "__main__":
    main()
