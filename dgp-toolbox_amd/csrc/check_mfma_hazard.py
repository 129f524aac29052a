"""ISA gate for the kernels whose MFMAs are inline asm (gemm_gram.h, gemm_tall.h, gemm_tallu.h): hipcc does not know that
the asm statement is a matrix instruction and inserts no wait states between a VALU write of a register and an MFMA that
reads it (found the hard way: NOTES.md, "a VALU write right in front of an inline-asm MFMA").  Reports every v_mfma_f64
whose accumulator / A / B registers were written by a VALU instruction fewer than WAIT instruction slots earlier (an
`s_nop n` counts as n + 1 slots).  WAIT = 2 is what LLVM's hazard recogniser keeps between a VALU write and a
compiler-visible MFMA on gfx90a and later; the case that produced wrong results had 0.

Part of the build: the Makefile keeps the gfx950 assembly of those three files (-save-temps=obj) and runs
    python3 check_mfma_hazard.py build/<file>-hip-amdgcn-amd-amdhsa-gfx950.s ...
as the rule of build/.hazard_ok, which `all` depends on: a build with a hazard fails.  tests/test_host_cpu.py checks the
scanner itself on a synthetic hazard and that the stamp exists after `make`.

usage: python check_mfma_hazard.py [--wait N] file.s [file.s ...]      (exit code 1 if any hazard is found)"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(lines, wait=2):
    """-> (number of MFMAs seen, [(writer text, slots between, mfma text)])"""
    recent = []          # (slots ago, written registers, text) of the latest VALU writes
    n_mfma, found = 0, []
    for line in lines:
        t = line.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest else []
        if op.startswith("v_mfma"):
            n_mfma += 1
            used = set().union(*[regs(x) for x in ops[:4]]) if ops else set()
            for age, w, text in recent:
                if age < wait and (w & used):
                    found.append((text, age, t))
            slots, written = 1, set()
        elif op == "s_nop":
            slots, written = int(ops[0], 0) + 1, set()
        elif op.startswith("v_") and ops:
            slots, written = 1, regs(ops[0])
        else:
            slots, written = 1, set()
        recent = [(a + slots, w, x) for a, w, x in recent if a + slots < 8]
        if written:
            recent.append((0, written, t))
    return n_mfma, found


def main(argv):
    wait = 2
    if argv and argv[0] == "--wait":
        wait, argv = int(argv[1]), argv[2:]
    if not argv:
        print(__doc__)
        return 2
    bad = 0
    for path in argv:
        with open(path) as f:
            n, found = scan(f, wait)
        for text, age, mfma in found:
            print(f"{path}: `{text}` {age} slot(s) before `{mfma}`")
        print(f"{path}: {n} MFMAs scanned, {len(found)} hazard(s)")
        if n == 0:
            print(f"{path}: no MFMA found - wrong file?")
            bad += 1
        bad += len(found)
    print("hazards:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
