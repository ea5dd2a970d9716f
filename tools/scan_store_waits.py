"""Static check of the compiled kernels (hipcc -S output): full vector-memory drains (`s_waitcnt vmcnt(0)`) that sit BEHIND
a global store inside a kernel -- each one exposes a store round trip (~700-1,000 cycles) to the wave that executes it.
usage: python tools/scan_store_waits.py /tmp/conv3x3.s [...]"""
import re, sys
for path in sys.argv[1:]:
    name, out = None, []
    for l in open(path):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, seen, n, stores = m.group(1), False, 0, 0
        elif name:
            if "global_store" in l or "buffer_store" in l:
                seen, stores = True, stores + 1
            elif seen and "s_waitcnt" in l and "vmcnt(0)" in l:
                n += 1
            elif "s_endpgm" in l:
                if n:
                    out.append((name, stores, n))
                name = None
    print(path, len(out), "kernels with a drain behind a store")
    for o in out:
        print("   %-110s stores %3d  drains behind a store %2d" % (o[0][14:124], o[1], o[2]))
