"""Instruction mix per basic block of one kernel in a hipcc -S listing: python tools/isa_block_mix.py file.s kernel_name_substring"""
import collections, sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
cur, blocks = "entry", collections.OrderedDict(entry=collections.Counter())
for l in lines[start + 1:end]:
    if l.startswith(".LBB"):
        cur = l.split(":")[0]; blocks[cur] = collections.Counter(); continue
    t = l.strip().split()[0] if l.strip() else ""
    if not t or t[0] in ";.":
        continue
    cls = ("mfma" if t.startswith("v_mfma") else "ds_read" if t.startswith("ds_read") else "ds_write" if t.startswith("ds_write")
           else "ds_perm" if t.startswith(("ds_bpermute", "ds_swizzle", "ds_permute")) else "gload" if t.startswith(("global_load", "buffer_load"))
           else "gstore" if t.startswith(("global_store", "buffer_store")) else "v_trans" if t.startswith(("v_exp", "v_rcp", "v_sqrt", "v_log", "v_rsq"))
           else "waitcnt" if t.startswith("s_waitcnt") else "barrier" if t.startswith("s_barrier") else "s_nop" if t.startswith("s_nop")
           else "accmov" if t.startswith(("v_accvgpr", )) else "valu" if t.startswith("v_") else "salu" if t.startswith("s_") else "other")
    blocks[cur][cls] += 1
for k, v in blocks.items():
    if sum(v.values()) >= int(sys.argv[3]) if len(sys.argv) > 3 else 40:
        print(k, sum(v.values()), dict(v))
