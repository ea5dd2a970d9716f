#!/usr/bin/env python3
"""Every inline-asm write-through store (`global_store_dwordx4 ... sc1`, common.hip.h: store16_out) of the SHIPPED library:
where do its data registers come from?

hipcc's hazard recogniser does not look inside inline asm.  The store's trailing `s_nop 1` covers the overwrite of its data
registers BEHIND it; nothing covers the hazard IN FRONT of it: a VGPR written by an MFMA and read by a vector-memory
instruction needs up to ~18 wait states on gfx942 / gfx950, which the compiler inserts for its own stores and cannot insert
for an asm one.  The precondition of store16_out is therefore: the data (and address) registers' LAST WRITER is
never a matrix instruction -- every call site packs, adds a bias or converts in between (VALU), and the compiler-handled
MFMA -> VALU wait protects that.  This tool checks the precondition on the machine code: the device code object is taken out
of liblocaldiff_hip.so (llvm-objdump --offloading), disassembled, and for every sc1 store the nearest previous writer of each
of its registers is found (inside the function, walking back over at most LOOKBACK instructions).  A `v_mfma*` there is
reported.  usage: scan_asm_store_sources.py [liblocaldiff_hip.so]   (exit code 1 if any store reads a raw accumulator)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
LOOKBACK = 4000


def regs(op):
    """'v[4:7]' / 'v5' -> set of VGPR numbers; anything else -> empty"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()


def written(mnem, ops):
    """VGPRs an instruction writes (a conservative reading of the operand list)"""
    if not ops or mnem.startswith(("global_store", "buffer_store", "ds_write", "ds_store", "flat_store", "scratch_store", "s_", "global_atomic", "buffer_inv", "buffer_wbl2")):
        if mnem.startswith("global_atomic") and "sc0" in ops:                # returning atomic: first operand
            return regs(ops[0])
        return set()
    w = regs(ops[0])
    if mnem.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap_b32")) and len(ops) > 1:
        w |= regs(ops[1])
    return w


def scan(code_object):
    return scan_text(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", code_object], check=True, capture_output=True, text=True).stdout)


def scan_text(txt):
    """disassembly text -> (sc1 16-byte stores, stores whose data / address comes straight from a matrix instruction, report lines)"""
    func, out = None, []
    funcs = {}
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            func = m.group(1)
            funcs[func] = []
            continue
        if func is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*(//.*)?$", line)
        if not m or not m.group(1) or m.group(1).endswith(":"):
            continue
        mnem, rest = m.group(1), m.group(2)
        ops = [o.strip() for o in re.split(r",\s*", rest)] if rest else []
        funcs[func].append((mnem, ops, line.strip()))
    stores = bad = 0
    for func, ins in funcs.items():
        for i, (mnem, ops, line) in enumerate(ins):
            if mnem != "global_store_dwordx4" or "sc1" not in " ".join(ops):
                continue
            stores += 1
            need = set()
            for o in ops[:2]:
                # address and data; modifiers (off, sc1, offset:..) parse to nothing
                need |= regs(o.split(" ")[0])
            for j in range(i - 1, max(-1, i - LOOKBACK), -1):
                pm, pops, pline = ins[j]
                w = written(pm, [o.split(" ")[0] for o in pops]) & need
                if not w:
                    continue
                if pm.startswith(("v_mfma", "v_smfmac")):
                    bad += 1
                    out.append(f"{func}: `{line}` reads v{sorted(w)} straight from `{pline}` ({i - j} instructions earlier)")
                need -= w
                if not need:
                    break
    return stores, bad, out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "localdiffusion-hallucination_amd", "csrc", "liblocaldiff_hip.so")
    tmp = tempfile.mkdtemp(prefix="ld_scan_")
    try:
        cp = os.path.join(tmp, "lib.so")
        shutil.copy(lib, cp)
        subprocess.run([OBJDUMP, "--offloading", cp], check=True, capture_output=True, text=True, cwd=tmp)
        cos = [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "amdgcn" in f]
        if not cos:
            print("no device code object found in", lib)
            return 2
        stores = bad = 0
        lines = []
        for co in cos:
            s, b, o = scan(co)
            stores, bad, lines = stores + s, bad + b, lines + o
        print(f"{stores} write-through (sc1) 16-byte stores in {len(cos)} code object(s); {bad} read a register whose last writer is a matrix instruction")
        for l in lines:
            print("  " + l)
        return 1 if bad else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
