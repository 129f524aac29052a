"""ISA check for the kernels whose MFMAs are inline asm (gemm_gram.h, gemm_tall.h, gemm_tallu.h): hipcc does not know
that the asm statement is a matrix instruction and inserts no wait states between a VALU write of a register and an
MFMA that reads it (found the hard way: DESIGN.md par. 10).  Compiles each file with -save-temps and reports every
v_mfma_f64 whose accumulator / A / B registers were written by a VALU instruction fewer than WAIT instruction slots
earlier (an `s_nop n` counts as n + 1 slots).  Exit code 1 if any is found.  WAIT = 2 is what LLVM's hazard recogniser
keeps between a VALU write and a compiler-visible MFMA on gfx90a and later; the case that produced wrong results had 0.
With 3 the weighted Gram kernel shows three places at exactly 2 (a scale multiply, one instruction, its MFMA).
usage: python tools/check_mfma_hazard.py [wait_slots=2]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dgp-toolbox_amd", "csrc")
WAIT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FILES = ["gemm_gram.hip", "gemm_tall.hip", "gemm_tallu.hip"]


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for f in FILES:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-save-temps", "-c",
                        os.path.join(SRC, f), "-o", os.path.join(tmp, f + ".o")], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        asm = os.path.join(tmp, f.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        recent = []          # (slots ago, written registers, text) of the latest VALU writes
        n_mfma = 0
        for line in open(asm):
            t = line.split(";")[0].strip()
            if not t or t.endswith(":") or t.startswith("."):
                continue
            op, _, rest = t.partition(" ")
            ops = [x.strip() for x in rest.split(",")] if rest else []
            if op.startswith("v_mfma"):
                n_mfma += 1
                used = set().union(*[regs(x) for x in ops[:4]]) if ops else set()
                for age, w, text in recent:
                    if age < WAIT and (w & used):
                        bad += 1
                        print(f"{f}: `{text}` {age} slot(s) before `{t}`")
                slots = 1
                written = set()
            elif op == "s_nop":
                slots, written = int(ops[0], 0) + 1, set()
            elif op.startswith("v_") and ops:
                slots, written = 1, regs(ops[0])
            else:
                slots, written = 1, set()
            recent = [(a + slots, w, x) for a, w, x in recent if a + slots < 8]
            if written:
                recent.append((0, written, t))
        print(f"{f}: {n_mfma} MFMAs scanned")
print("hazards:", bad)
sys.exit(1 if bad else 0)
