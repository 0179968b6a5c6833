"""Inline-asm MFMAs are invisible to the compiler's hazard recogniser: it does not insert the wait states the ISA requires
between a VALU instruction that writes a VGPR and a v_mfma that reads it as SrcA/SrcB (2 on gfx90a+; LLVM's
GCNHazardRecognizer::checkMAIHazards90A, "LegacyVALUWritesVGPRWaitStates").  conv_h3q / conv_h3g / conv_h2q issue their MFMAs
through asm statements (AGPR accumulators updated in place), and build some A operands with a VALU select ([0 | w hi]).  This
script compiles nbe_kernels_h3.hip to assembly and fails if any v_mfma reads a VGPR written by a VALU instruction fewer than
three instructions earlier without s_nop in between.

    python tools/check_mfma_hazards.py          # exit code 1 and a listing if a hazard is found
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "jax_nbody_emulator_with_dj_amd", "csrc", "nbe_kernels_h3.hip")
NEED = 2                                            # wait states between the VALU write and the MFMA read


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


NKERN = [0]


def scan(asm):
    found = []
    for m in re.finditer(r"^(_ZN3nbe\w+):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.M | re.S):
        name, body = m.group(1), m.group(2)
        NKERN[0] += 1
        lines = [l.strip() for l in body.split("\n")]
        lines = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
        for k, l in enumerate(lines):
            if not l.startswith("v_mfma"):
                continue
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
            src = regs(ops[1]) | regs(ops[2])
            waited = 0
            for back in range(1, NEED + 1):
                if k - back < 0:
                    break
                p = lines[k - back]
                if p.startswith("s_nop"):
                    waited += int(p.split()[1])             # s_nop N: N + 1 wait states, one of them counted by its position
                    continue
                if p.startswith("v_") and not p.startswith(("v_mfma", "v_accvgpr_read")):
                    dst = p.split(None, 1)[1].split(",")[0].strip()
                    if regs(dst) & src and (back - 1) + waited < NEED:
                        found.append((name, p, l))
    return found


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "h3.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-w",
                               "-o", out, SRC])
        found = scan(open(out).read())
    for name, p, l in found:
        print("%s:\n    %s\n    %s" % (name, p, l))
    print("%d hazard(s) in %d kernels" % (len(found), NKERN[0]))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
