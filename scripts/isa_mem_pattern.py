"""The memory instructions of a kernel in the order the hardware sees them.

Compiles one .hip source of the package to gfx950 assembly (hipcc cross-compiles: no GPU needed) and prints, per kernel whose mangled
name contains one of the given substrings, one character per instruction:

    L  vector load        S  vector store / atomic        s  scalar load
    W  s_waitcnt vmcnt(0) w  s_waitcnt vmcnt(n > 0)       .  s_waitcnt lgkmcnt(0) (runs of them collapsed)

A kernel whose source says "fetch these four things, then decide" and whose line reads LWLWLWLW has three dependent round trips too
many: the compiler sinks a load into the branch that holds its only use, and it does not make a load scalar when an earlier store of
the kernel might alias it ("the memory instructions in the order the hardware sees them").

usage: python scripts/isa_mem_pattern.py hem.hip k_select k_mstepILi4 k_spans
       python scripts/isa_mem_pattern.py icp.hip k_icp_nn
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def assembly(source: str) -> str:
    src = os.path.join(ge.CSRC, source)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = [f for f in ge.HIPCC_FLAGS if f not in ("-fPIC",)] + ge.EXTRA_FLAGS.get(source, [])
        subprocess.run([ge._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def pattern(body: str):
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    pat = []
    for l in lines:
        if l.startswith(("global_load", "buffer_load")):
            pat.append("L")
        elif l.startswith(("global_store", "global_atomic", "buffer_store")):
            pat.append("S")
        elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            pat.append("W")
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            pat.append("w")
        elif l.startswith("s_load"):
            pat.append("s")
        elif l.startswith("s_waitcnt") and "lgkmcnt(0)" in l:
            pat.append(".")
    return len(lines), re.sub(r"\.+", ".", "".join(pat))


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    s = assembly(sys.argv[1])
    want = sys.argv[2:]
    for m in re.finditer(r"^(_ZN[^\n:]+):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name = m.group(1)
        if want and not any(w in name for w in want):
            continue
        regs = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", m.group(2))
        n, p = pattern(m.group(2))
        print(f"{name[:90]}  ({n} instructions, {regs.group(1) if regs else '?'} VGPRs)")
        for i in range(0, len(p), 160):
            print("    " + p[i:i + 160])
    return 0


if __name__ == "__main__":
    sys.exit(main())
