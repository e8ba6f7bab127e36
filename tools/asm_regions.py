"""where a kernel's scratch traffic, MFMAs and buffer loads sit in its assembly: python tools/asm_regions.py file.s kernel_substring"""
import collections, sys
txt = open(sys.argv[1]).read()
i = txt.index(sys.argv[2] + "")
i = txt.index(":\n", i)
lines = txt[i:txt.index('.Lfunc_end', i)].split('\n')
print("lines", len(lines), "s_endpgm at", [n for n, l in enumerate(lines) if 's_endpgm' in l])
for key in ('scratch_', 'v_mfma', 'buffer_load_dwordx4', 'ds_read_b128', 'ds_write_b128', 's_barrier', 's_waitcnt vmcnt'):
    at = [n for n, l in enumerate(lines) if key in l]
    h = collections.Counter(n // 400 for n in at)
    print("%-22s %5d  per 400 lines: %s" % (key, len(at), sorted(h.items())))
