"""Static check of the compiled kernels (hipcc -S output): global loads that are waited for ONE AT A TIME -- a load, then
`s_waitcnt vmcnt(0)` before the next load is issued.  Each such pair is a dependent global round trip (~700-900 cycles);
copy loops written as `lds[i] = global[i]` compile to exactly that (load, wait, ds_write, branch).
usage: python tools/scan_serial_loads.py /tmp/linattn_fused.s [...]"""
import re, sys
for path in sys.argv[1:]:
    name, out = None, []
    for l in open(path):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, pending, singles, loads = m.group(1), 0, 0, 0
        elif name:
            if re.search(r"\b(global|buffer)_load", l) and "lds" not in l:
                pending += 1; loads += 1
            elif "s_waitcnt" in l and "vmcnt(0)" in l:
                if pending == 1:
                    singles += 1
                pending = 0
            elif l.startswith(".Lfunc_end"):
                if singles:
                    out.append((name, loads, singles))
                name = None
    print(path, len(out), "kernels with single-load drains")
    for o in sorted(out, key=lambda t: -t[2]):
        print("   %-100s loads %3d  single-load drains %2d" % (o[0][14:114], o[1], o[2]))
