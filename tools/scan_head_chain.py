"""Static view of a kernel's HEAD (hipcc -S output): the dependent load stages in front of the first MFMA / store.
Walks the listing in text order; every `s_waitcnt` that has scalar (S) or vector (V) loads outstanding closes a stage --
each stage is one dependent memory round trip (~0.5-1 us under load) on the workgroup's critical path.  Prints e.g.
`+s8 S6 S2 +v23 V7 V16 | mfma` = 8 scalar loads issued, waited for in two batches (ONE round trip: nothing was issued
between the waits), 23 vector loads issued, 7 waited for, then 16.  A wait that follows a `+` token is a new round trip.  Branches are not followed: read the
listing before acting on a line.  usage: python tools/scan_head_chain.py file.s [name-substring ...]"""
import re, sys
path, pats = sys.argv[1], sys.argv[2:]
name = None
for l in open(path):
    m = re.match(r"^(_Z\S+):", l)
    if m:
        name, s_out, v_out, stages, done, s_new, v_new = m.group(1), 0, 0, [], False, 0, 0
        continue
    if not name or done is None:
        continue
    if l.startswith(".Lfunc_end"):
        if not pats or any(p in name for p in pats):
            print("%-110s %s" % (name[:110], " ".join(stages)))
        name = None
        continue
    if done:
        continue
    t = l.strip()
    if t.startswith("s_load") or t.startswith("s_buffer_load"):
        s_out += 1; s_new += 1
    elif re.match(r"(global|buffer|flat)_load", t):
        v_out += 1; v_new += 1
    elif t.startswith("s_waitcnt"):
        if s_new:
            stages.append("+s%d" % s_new); s_new = 0
        if v_new:
            stages.append("+v%d" % v_new); v_new = 0
        mv = re.search(r"vmcnt\((\d+)\)", t)
        ml = re.search(r"lgkmcnt\((\d+)\)", t)
        if ml and s_out > int(ml.group(1)):
            stages.append("S%d" % (s_out - int(ml.group(1)))); s_out = int(ml.group(1))
        if mv and v_out > int(mv.group(1)):
            stages.append("V%d" % (v_out - int(mv.group(1)))); v_out = int(mv.group(1))
    elif t.startswith("s_barrier"):
        stages.append("B")
    elif t.startswith("v_mfma") or re.match(r"(global|buffer|flat)_(store|atomic)", t):
        stages.append("| " + t.split()[0]); done = True
